// wdma: the Generator's resblock convolutions at 64 / 128 / 256 channels (ResBlock1, module/modules.py:190-203 of the
// reference) -- wconv.h's weights-in-registers kernel with its per-tile staging and residual traffic moved OFF the
// vector ALU: rows and residual travel global -> LDS by `global_load_lds` (LDS-DMA), no staging registers, no commit pass.
//
// What wconv.h's stamps said (profiles/r04_wconv_commit_in_loop.txt, 128 channels, cycles per 64-row tile): issue 1120 |
// MFMA 6232 | epilogue 2060 | commit 2780 | barrier 156 -- with one wave per SIMD nothing overlaps, so the 3.9k cycles that
// request the next tile's rows into registers and later push them (leaky-ReLU'd) into LDS sit beside the 5.6k of matrix work.
// Here:
//   * the conv's INPUT is already in the form it contracts: the producer of a resblock state x (the transposed conv, the
//     second conv of the previous pair) writes lrelu(x) NEXT TO x (`A*`: the activated copy, computed from the ROUNDED bf16
//     output, which is what the staging pass computed -- bit-identical), so staging is a pure copy and can be a DMA;
//   * a DMA instruction fills 1 KiB of LDS linearly (lane l -> byte 16 l), so the 16-byte row skew of wconv.h is impossible;
//     rows sit unpadded and the 16-byte pieces of a row are XOR-SWIZZLED instead: piece p of tile row r lives in slot
//     p ^ f(r), f(r) = (r / rows-per-256-B) mod min(pieces, 16).  The lane's GLOBAL address carries the permutation, the
//     32-row B-fragment read (16 lanes per LDS cycle group, 16 different r mod 16) touches 16 different 16-byte bank
//     groups: conflict-free, no padding bytes;
//   * the residual goes straight into the wave's (swizzled) output patch by DMA as well;
//   * rows beyond the sequence read a zero page, stores of rows beyond it land in a sink: every wave issues the same number
//     of memory instructions per tile, so all waits are COUNTED (`s_waitcnt vmcnt(n)`): a tile's stores are still in flight
//     when the next tile's MFMAs start, only the DMAs issued before them have to have landed.
// hipcc does not track LDS-DMA writes against later ds_reads (checked in the ISA of cgemm.h): every such dependency below is
// an explicit `s_waitcnt` + (for other waves' data) `s_barrier`.
#pragma once
#include "wconv.h"
#include "cgemm.h"

namespace gsv {

struct WDmaArgs {
    const bf16_t *X0, *X1, *X2;   // inputs [n_rows][ld], already activated by their producer
    const uint4 *W0, *W1, *W2;    // tapgemm fragment-packed weights ([tap][mtile][kstep][lane])
    const float *b0, *b1, *b2;    // bias [C]
    const bf16_t *R0, *R1, *R2;   // residual [n_rows][ld] or null
    bf16_t *Y0, *Y1, *Y2;         // outputs [n_rows][ld]
    bf16_t *A0, *A1, *A2;         // null, or lrelu(Y, act_slope) of the rounded output (the next conv's input)
    int k0, k1, k2;               // taps (3, 7 or 11)
    int d0, d1, d2;               // dilation (<= 5), "same" padding
    int nb0, nb1, nb2;            // blocks dealt to each branch
    int ld, n_rows;
    float out_slope;              // leaky-ReLU on the output (1 = none)
    float act_slope;              // slope of the activated copy
    const void* zeros;            // >= 16 readable zero bytes (rows outside the sequence)
    void* sink;                   // >= 4 KiB scratch: stores of rows beyond the sequence
    long long* dbg;               // null, or cycle stamps of block 0 / wave 0 (tools/tg_bench)
};

// One LDS-DMA instruction (16 bytes per lane, the wave's 1 KiB lands at LDS byte `lds_addr` + 16 lane), as inline asm: through
// `__builtin_amdgcn_global_load_lds` hipcc books the instruction as a FLAT access that may touch LDS, and while one is pending every
// LDS wait it places is `lgkmcnt(0)` -- the MFMA loop's counted B-fragment waits (three groups in flight) became full drains
// (7.7k instead of 6.2k cycles per 128-channel tile).  Hidden in asm, the DMA is invisible to that bookkeeping; the waits on it
// are the explicit ones below.  One wait state between the M0 write and its use; five in front of the statement, because its
// scalar base may come fresh from a v_readfirstlane and hipcc pads nothing inside an asm string.
__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds_addr) {   // scalar base + per-lane byte offset
    unsigned keep;     // M0 is the compiler's: saved and restored inside the statement (a clobber would not be honoured)
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void dma16(const void* vaddr, unsigned lds_addr) {                  // per-lane 64-bit address
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(vaddr) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {   // wave-uniform LDS byte address of a __shared__ pointer
    typedef __attribute__((address_space(3))) const unsigned char lds_uc;
    return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_uc*)p);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" : : : "memory");
}

template <int C, int MS, int BN, int NT, int KSP, int MSP>
__device__ __forceinline__ void wdma_body(const bf16_t* __restrict__ X, const uint4* __restrict__ W, const float* __restrict__ bias,
                                          const bf16_t* R, bf16_t* Y, bf16_t* A, int dil, int blk, int nblk, const WDmaArgs& a,
                                          unsigned char* lds, long long* dbg) {
    static_assert(C % 64 == 0 && C <= 256, "64, 128, 192*, 256 channels (* needs a non-power-of-two swizzle: not built)");
    constexpr int KSTEPS = C / 16;
    constexpr int MT = C / 32;
    constexpr int RG = 4 / (MS * KSP);
    constexpr int KSW = KSTEPS / KSP;
    constexpr int WN = BN / 32 / RG;
    constexpr int RS = C * 2;                      // LDS bytes per staged row (no padding)
    constexpr int VPR = C / 8;                     // 16-byte pieces per row
    constexpr int RPI = 64 / VPR > 0 ? 64 / VPR : 1;   // rows per DMA instruction (1 KiB)
    static_assert(VPR <= 64, "a row is at most one DMA instruction");
    constexpr int RPP = 4 * RPI;                   // rows per pass of the block's four waves
    constexpr int RPB = VPR >= 16 ? 1 : 16 / VPR;  // rows per 256 bytes of LDS
    constexpr int SM = (VPR < 16 ? VPR : 16) - 1;  // swizzle mask
    constexpr int XROWS = BN + (NT - 1) * 5;       // rows staged at the largest dilation
    constexpr int NPASS = (XROWS + RPP - 1) / RPP; // DMA instructions per wave per tile (fixed: the waits are counted)
    constexpr int XBYTES = ((BN + 50 + RPP - 1) / RPP) * RPP * RS;   // sized for 11 taps whatever NT is (one LDS map per launch)
    constexpr int NVR = WN * 32 * 4 / 64;          // patch vectors per lane
    constexpr int ROBYTES = WN * 32 * 64;          // wave-private residual / output patch: [rows][4 pieces], swizzled

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = lane & 31, hf = lane >> 5;
    const int ms = wid % MS, kh = (wid / MS) % KSP, rg = wid / (MS * KSP);
    const int mg = MSP > 1 ? blk % MSP : 0;
    const int gs = mg * MS + ms;                  // the wave's 32-channel slice of the conv
    if (MSP > 1) { blk /= MSP; nblk /= MSP; }
    const int wrow = rg * WN * 32;
    unsigned char* xbuf0 = lds;
    unsigned char* xbuf1 = lds + XBYTES;
    unsigned char* ro = lds + 2 * XBYTES + wid * ROBYTES;
    float* bl = reinterpret_cast<float*>(lds + 2 * XBYTES + 4 * ROBYTES);
    float* kred = reinterpret_cast<float*>(lds + 2 * XBYTES + 4 * ROBYTES + MS * 32 * sizeof(float)) + (size_t)(rg * MS + ms) * WN * 16 * 64;

    const int n_rows = a.n_rows, ld = a.ld;
    const int pad = (NT - 1) / 2 * dil;
    const int rows = BN + (NT - 1) * dil;
    const int ntiles = (n_rows + BN - 1) / BN;
    if (blk >= ntiles) return;

    u32x4 w[NT][KSW];
    auto load_weights = [&]() {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks)
                w[t][ks] = __builtin_bit_cast(u32x4, W[(((size_t)t * MT + gs) * KSTEPS + kh * KSW + ks) * 64 + lane]);
    };
    const float bias_raw = bias[mg * MS * 32 + (tid < MS * 32 ? tid : 0)];

    // rows: lane -> (row of the instruction's 1 KiB, slot); the piece it fetches is slot ^ f(row).  An interior tile (every row it
    // touches exists) takes the scalar-base form: ONE per-lane byte offset serves all passes (f(row) does not depend on the pass
    // for C <= 128; for 256 it alternates), the pass moves the SGPR base.  Edge tiles select the zero page per lane.
    const int lr = lane / VPR, ls = lane % VPR;
    const size_t ld2 = (size_t)ld * 2;
    unsigned lane_off[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int r = (u * 4 + wid) * RPI + lr;
        lane_off[u] = (unsigned)((wid * RPI + lr) * (int)ld2 + (ls ^ ((r / RPB) & SM)) * 16);
    }
    auto issue_x = [&](int tile, unsigned char* xb) {
        int gbase = tile * BN - pad;
        const unsigned xl = lds_addr_of(xb + wid * 1024);
        asm volatile("" : "+s"(gbase));           // per-tile value: nothing below is hoisted out of the tile loop (it spilled)
        if (gbase >= 0 && gbase + NPASS * RPP <= n_rows) {
            const unsigned char* base = cg_uniform(reinterpret_cast<const unsigned char*>(X) + (size_t)gbase * ld2);
#pragma unroll
            for (int v = 0; v < NPASS; ++v)
                dma16(base + (size_t)(v * RPP) * ld2, lane_off[(RPP % 16 != 0) ? (v & 1) : 0], xl + v * 4096);
        } else {
#pragma unroll
            for (int v = 0; v < NPASS; ++v) {
                const int r = (v * 4 + wid) * RPI + lr;
                const int p = ls ^ ((r / RPB) & SM);
                const int grow = gbase + r;
                const bool ok = grow >= 0 && grow < n_rows;
                const bf16_t* src = ok ? X + (size_t)grow * ld + p * 8 : reinterpret_cast<const bf16_t*>(a.zeros);
                dma16(src, xl + v * 4096);
            }
        }
    };
    const bool epi = kh == 0;                     // the wave that owns the tile's epilogue
    // Two epilogues.  KSP > 1 (256 channels: 88 fragments + the partial-tile exchange leave no registers) keeps the PATCH form:
    // residual -> the wave's patch: vector q of lane l is patch bytes [16 (64 q + l), +16) = (row, slot); piece = slot ^ ((row >> 2) & 3)
#ifdef WDMA_FORCE_PATCH
    constexpr bool DIRECT = false;                  // A/B build
#else
    constexpr bool DIRECT = KSP == 1;
#endif
    const unsigned rl = lds_addr_of(ro);
    auto issue_r_patch = [&](int nb0) {
#pragma unroll
        for (int q = 0; q < NVR; ++q) {
            const int idx = q * 64 + lane, row = idx >> 2, pc = (idx & 3) ^ ((row >> 2) & 3);
            const int n = min(nb0 + wrow + row, n_rows - 1);
            dma16(R + (size_t)n * ld + gs * 32 + pc * 8, rl + q * 1024);
        }
    };
    // KSP = 1: the epilogue without the LDS transposition (round 5; the patch form above cost 2.2k cycles per tile, most of it the dependent
    // chain ds_write -> ds_read -> store at one wave per SIMD).  A lane of the D fragment owns 16 consecutive channels of one row =
    // two 16-byte pieces (p0, p1 for hf = 0; p2, p3 for hf = 1).  Stored as they are, an instruction would write 16 bytes and skip
    // 16; ONE v_permlane32_swap per dword hands lane (j, 1) piece p1 and lane (j, 0) piece p2, so that the first store instruction
    // writes bytes [0, 32) of the slice's 64-byte row and the second [32, 64): 32-byte runs, no LDS, and the residual comes in by
    // plain loads in the same swapped order (the swap is its own inverse).  Arithmetic and rounding points are the patch form's.
    u32x4 rF[WN], rS[WN];
    auto issue_r_direct = [&](int nb0) {
#pragma unroll
        for (int k = 0; k < WN; ++k) {
            const int n = min(nb0 + wrow + k * 32 + j, n_rows - 1);
            const bf16_t* rp = R + (size_t)n * ld + gs * 32 + hf * 8;
            rF[k] = *reinterpret_cast<const u32x4*>(rp);
            rS[k] = *reinterpret_cast<const u32x4*>(rp + 16);
        }
    };
    float bb[16];                                   // the lane's 16 biases (registers: the staging vectors of wconv.h are gone)
    auto issue_r = [&](int nb0) { if constexpr (DIRECT) issue_r_direct(nb0); else issue_r_patch(nb0); };

    issue_x(blk, xbuf0);
    load_weights();
    if (tid < MS * 32) bl[tid] = bias_raw;
    // Only the first tile's rows (and the bias) have to have landed here: they are OLDER than the NT * KSW fragment loads, which were
    // issued in the order the MFMA loop consumes them -- hipcc's own counted waits in front of each fragment's first MFMA let the first
    // tile's loop start behind the first fragments while the rest of the block's 90-360 KB of weights is still streaming in (round 6;
    // the drain here cost every launch the whole weight pull, ~6 us at 128 channels, before its first MFMA).  vmcnt holds 6 bits.
    wait_vm<(NT * KSW < 63 ? NT * KSW : 63)>();
    lds_barrier();
    if constexpr (DIRECT) {
#pragma unroll
        for (int q = 0; q < 16; ++q) bb[q] = bl[ms * 32 + 16 * hf + q];
    }
    int cur = 0, nst = 0;
    auto stamp = [&]() { if (dbg && blk == 0 && tid == 0 && nst < 60) dbg[nst] = (long long)__builtin_readcyclecounter(); ++nst; };
    const unsigned c0 = (unsigned)(kh * KSW * 32 + hf * 16);
    for (int tile = blk; tile < ntiles; tile += nblk) {
        stamp();
        const int tn = tile + nblk;
        const bool has_next = tn < ntiles;
        const int nb0 = tile * BN;
        // in flight from the previous tile: its stores (the patch was read into registers for them: free)
        if (R && epi) issue_r(nb0);
        if (has_next) issue_x(tn, cur ? xbuf0 : xbuf1);

        stamp();
        f32x16 acc[WN];
        {
            const unsigned char* xb = cur ? xbuf1 : xbuf0;
            const unsigned rowj = (unsigned)(wrow + j);
            int dl = dil;
            asm volatile("" : "+s"(dl));          // the tap addresses are recomputed per tile, not kept in 11+ registers across the loop
#pragma unroll
            for (int k = 0; k < WN; ++k)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[k][q] = 0.f;
            constexpr int NIT = NT * KSW;
            constexpr int DEPTH = 3;
            u32x4 bf[DEPTH + 1][WN];
            auto ldb = [&](int it, u32x4 (&dst)[WN]) {
                const int t = it / KSW, ks = it % KSW;
                const unsigned rt = rowj + (unsigned)(t * dl);
                const unsigned pre = rt * RS + (c0 ^ (((rt / RPB) & SM) << 4));
                const unsigned ad = pre ^ (unsigned)(ks * 32);
#pragma unroll
                for (int k = 0; k < WN; ++k) dst[k] = *reinterpret_cast<const u32x4*>(xb + ad + k * 32 * RS);
            };
#pragma unroll
            for (int it = 0; it < DEPTH && it < NIT; ++it) ldb(it, bf[it % (DEPTH + 1)]);
            __builtin_amdgcn_sched_group_barrier(0x100, DEPTH * WN, 0);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (it + DEPTH < NIT) ldb(it + DEPTH, bf[(it + DEPTH) % (DEPTH + 1)]);
#pragma unroll
                for (int k = 0; k < WN; ++k) Mma<bf16_t>::run(acc[k], w[it / KSW][it % KSW], bf[it % (DEPTH + 1)][k]);
                __builtin_amdgcn_sched_group_barrier(0x008, WN, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, WN, 0);
            }
        }
        if constexpr (KSP > 1) {   // the second half hands its partial tiles over
            if (kh == 1) {
#pragma unroll
                for (int k = 0; k < WN; ++k)
#pragma unroll
                    for (int q = 0; q < 16; ++q) kred[(k * 16 + q) * 64 + lane] = acc[k][q];
            }
            lds_barrier();
            if (epi) {
#pragma unroll
                for (int k = 0; k < WN; ++k)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[k][q] += kred[(k * 16 + q) * 64 + lane];
            }
        }

        stamp();
        if constexpr (!DIRECT) {
        if (epi) {
            // the residual rows have landed when only the next tile's row DMAs (issued behind them) are outstanding
            if (R) { if (has_next) wait_vm<NPASS>(); else wait_vm<0>(); }
#pragma unroll
            for (int k = 0; k < WN; ++k) {
                const int row = k * 32 + j, sw = (row >> 2) & 3;
                unsigned char* pa = ro + row * 64 + ((2 * hf) ^ sw) * 16;
                unsigned char* pb = ro + row * 64 + ((2 * hf + 1) ^ sw) * 16;
                float v[16];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + ms * 32 + 16 * hf + 4 * q4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * q4 + e] = acc[k][4 * q4 + e] + b4[e];
                }
                if (a.out_slope != 1.0f) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) v[q] = lrelu(v[q], a.out_slope);
                }
                if (R) {
                    const u32x4 ra = *reinterpret_cast<const u32x4*>(pa), rb = *reinterpret_cast<const u32x4*>(pb);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] += __uint_as_float(ra[e] << 16);
                        v[2 * e + 1] += __uint_as_float(ra[e] & 0xffff0000u);
                        v[8 + 2 * e] += __uint_as_float(rb[e] << 16);
                        v[8 + 2 * e + 1] += __uint_as_float(rb[e] & 0xffff0000u);
                    }
                }
                u32x4 oa, ob;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    oa[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                    ob[e] = pack_bf16x2(v[8 + 2 * e], v[8 + 2 * e + 1]);
                }
                *reinterpret_cast<u32x4*>(pa) = oa;
                *reinterpret_cast<u32x4*>(pb) = ob;
            }
            unsigned char* snk = reinterpret_cast<unsigned char*>(a.sink) + tid * 16;
#pragma unroll
            for (int q = 0; q < NVR; ++q) {
                const int idx = q * 64 + lane, row = idx >> 2, pc = (idx & 3) ^ ((row >> 2) & 3);
                const u32x4 o = *reinterpret_cast<const u32x4*>(ro + idx * 16);
                const int n = nb0 + wrow + row;
                const size_t off = (size_t)n * ld + gs * 32 + pc * 8;
                *reinterpret_cast<u32x4*>(n < n_rows ? reinterpret_cast<unsigned char*>(Y + off) : snk) = o;
                if (A) {
                    u32x4 oa;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        oa[e] = pack_bf16x2(lrelu(__uint_as_float(o[e] << 16), a.act_slope), lrelu(__uint_as_float(o[e] & 0xffff0000u), a.act_slope));
                    *reinterpret_cast<u32x4*>(n < n_rows ? reinterpret_cast<unsigned char*>(A + off) : snk) = oa;
                }
            }
        }

        } else {
        if (epi) {
            unsigned char* snk = reinterpret_cast<unsigned char*>(a.sink) + tid * 16;
#pragma unroll
            for (int k = 0; k < WN; ++k) {
                float v[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = acc[k][q] + bb[q];
                if (a.out_slope != 1.0f) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) v[q] = lrelu(v[q], a.out_slope);
                }
                if (R) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {    // back from the stored order: ra = this lane's channels 0-7, rb = 8-15
                        const auto t = __builtin_amdgcn_permlane32_swap(rF[k][e], rS[k][e], false, false);
                        const unsigned ra = t[0], rb = t[1];
                        v[2 * e] += __uint_as_float(ra << 16);
                        v[2 * e + 1] += __uint_as_float(ra & 0xffff0000u);
                        v[8 + 2 * e] += __uint_as_float(rb << 16);
                        v[8 + 2 * e + 1] += __uint_as_float(rb & 0xffff0000u);
                    }
                }
                u32x4 oF, oS;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto t = __builtin_amdgcn_permlane32_swap(pack_bf16x2(v[2 * e], v[2 * e + 1]), pack_bf16x2(v[8 + 2 * e], v[8 + 2 * e + 1]), false, false);
                    oF[e] = t[0]; oS[e] = t[1];
                }
                const int n = nb0 + wrow + k * 32 + j;
                const bool ok = n < n_rows;
                const size_t off = ((size_t)n * ld + gs * 32 + hf * 8) * 2;
                *reinterpret_cast<u32x4*>(ok ? reinterpret_cast<unsigned char*>(Y) + off : snk) = oF;
                *reinterpret_cast<u32x4*>(ok ? reinterpret_cast<unsigned char*>(Y) + off + 32 : snk) = oS;
                if (A) {
                    u32x4 aF, aS;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        aF[e] = pack_bf16x2(lrelu(__uint_as_float(oF[e] << 16), a.act_slope), lrelu(__uint_as_float(oF[e] & 0xffff0000u), a.act_slope));
                        aS[e] = pack_bf16x2(lrelu(__uint_as_float(oS[e] << 16), a.act_slope), lrelu(__uint_as_float(oS[e] & 0xffff0000u), a.act_slope));
                    }
                    *reinterpret_cast<u32x4*>(ok ? reinterpret_cast<unsigned char*>(A) + off : snk) = aF;
                    *reinterpret_cast<u32x4*>(ok ? reinterpret_cast<unsigned char*>(A) + off + 32 : snk) = aS;
                }
            }
        }
        }   // DIRECT

        stamp();
        // next tile's rows: this wave's DMAs are older than its stores of this tile.  The counts are the stores the epilogue above issues per
        // wave: NVR (= 2 WN in the DIRECT form: two per 32-row tile) for Y, as many again for A.  An instruction more than counted (hipcc
        // splitting a store, a scratch access) only over-waits; one FEWER would under-wait and read stale rows with no error -- so the
        // relation is pinned here, and -DWDMA_DRAIN builds every counted wait as vmcnt(0) (the A/B build behind the bit-identity test's claim).
        static_assert(NVR == 2 * WN, "the counted waits assume NVR = 2 WN stores per output tensor per tile");
        if (has_next) {
#ifdef WDMA_DRAIN
            wait_vm<0>();
#else
            if (!epi) wait_vm<0>();
            else if (A) wait_vm<2 * NVR>();
            else wait_vm<NVR>();
#endif
        }
        lds_barrier();
        stamp();
        cur ^= 1;
    }
}

template <int C, int MS, int BN, int KSP = 1, int MSP = 1>
__global__ __launch_bounds__(256, 1) void wdma_kernel(WDmaArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int b = blockIdx.x;
    const int br = b < a.nb0 ? 0 : (b < a.nb0 + a.nb1 ? 1 : 2);
    const int blk = br == 0 ? b : (br == 1 ? b - a.nb0 : b - a.nb0 - a.nb1);
    const int nblk = br == 0 ? a.nb0 : (br == 1 ? a.nb1 : a.nb2);
    const bf16_t* X = br == 0 ? a.X0 : (br == 1 ? a.X1 : a.X2);
    const uint4* W = br == 0 ? a.W0 : (br == 1 ? a.W1 : a.W2);
    const float* bias = br == 0 ? a.b0 : (br == 1 ? a.b1 : a.b2);
    const bf16_t* R = br == 0 ? a.R0 : (br == 1 ? a.R1 : a.R2);
    bf16_t* Y = br == 0 ? a.Y0 : (br == 1 ? a.Y1 : a.Y2);
    bf16_t* A = br == 0 ? a.A0 : (br == 1 ? a.A1 : a.A2);
    const int k = br == 0 ? a.k0 : (br == 1 ? a.k1 : a.k2);
    const int dil = br == 0 ? a.d0 : (br == 1 ? a.d1 : a.d2);
    long long* dbg = br == 0 ? a.dbg : nullptr;
    if (k == 11) wdma_body<C, MS, BN, 11, KSP, MSP>(X, W, bias, R, Y, A, dil, blk, nblk, a, lds, dbg);
    else if (k == 7) wdma_body<C, MS, BN, 7, KSP, MSP>(X, W, bias, R, Y, A, dil, blk, nblk, a, lds, dbg);
    else if (k == 3) wdma_body<C, MS, BN, 3, KSP, MSP>(X, W, bias, R, Y, A, dil, blk, nblk, a, lds, dbg);
}

template <int C, int MS, int BN, int KSP = 1, int MSP = 1>
constexpr size_t wdma_lds_bytes() {
    constexpr int RPI = 64 / (C / 8) > 0 ? 64 / (C / 8) : 1;
    constexpr int RPP = 4 * RPI;
    return (size_t)2 * ((BN + 50 + RPP - 1) / RPP) * RPP * (C * 2) + (size_t)4 * (BN / (4 / (MS * KSP))) * 64 + MS * 32 * sizeof(float) +
           (KSP > 1 ? (size_t)(4 / KSP) * (BN / (4 / (MS * KSP)) / 32) * 16 * 64 * sizeof(float) : 0);
}

}  // namespace gsv
