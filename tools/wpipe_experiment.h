// EXPERIMENT (tools/tg_bench only; NOT part of the library -- measured and not adopted, profiles/r06_wpipe_notes.txt).
// wpipe: the Generator's resblock convolutions at 64 / 128 channels (ResBlock1, module/modules.py:190-203 of the reference) with TWO
// waves per SIMD -- wdma.h's weights-in-registers / rows-by-LDS-DMA kernel re-cut so that a tile's non-matrix work runs beside matrix work.
//
// What wdma.h's stamps say (profiles/r05_wdma_notes.txt, 128 channels, cycles per 64-row tile): issue 900 | MFMA 6400 | epilogue 1800 |
// barrier 180 -- one wave per SIMD (a 32-channel slice of 11 taps is 352 weight registers), so nothing hides the 2.9k cycles beside the
// 5.6k the matrix pipe needs, and hipcc cannot pipeline an epilogue into the MFMA loop (notes, section 3).  Here a block is EIGHT waves;
// the two waves of a SIMD (wave s and wave s + 4: checked with HW_ID, profiles/r06_wpipe_notes.txt) share one output tile and split its
// CONTRACTION: the (tap, k-step) walk is cut in the middle, the FRONT wave holds the first half of the slice's fragments (176 registers
// at 128 channels x 11 taps), the BACK wave the second half.
//
// Measured on the way (same notes): a wave's vector instructions issued beside its partner's MFMA stream cost ~10 cycles each instead
// of 4 (the matrix pipe takes its share of the SIMD's vector issue), whatever `s_setprio` says -- an epilogue of ~450 instructions in ONE
// wave needs 5k cycles there, longer than the partner's 2.8k of MFMAs.  So the tile's vector work is cut in two as well, and the two
// waves alternate on the matrix pipe:
//   interval p      front                                                   back
//   after B1        MFMAs of its half of tile p, ON TOP OF bias + residual      + the front's partial tile (p - 1), leaky-ReLU, round to
//                   (its accumulators were initialised with them)               bf16 -> packed rows to LDS; their activated copy -> LDS
//   B2 -------------------------------------------------------------------------------------------------------------------------------
//                   partial tile p -> LDS; residual DMA + row DMAs of tile      MFMAs of its half of tile p
//                   p + 1; packed rows of tile p - 1 (and the copy) -> global
//                   in 64-byte runs; accumulators := bias + residual (p + 1)
//   B1 -------------------------------------------------------------------------------------------------------------------------------
// Each wave's MFMAs run beside <= 200 vector instructions of the other; the matrix pipe idles only around the two barriers.
// Rows outside the sequence: the DMAs and the stores are BUFFER instructions whose descriptor carries the tensor's size -- a lane whose
// row is out of range reads zeros (the conv's "same" padding) / is not written; no zero page, no sink, no edge path, 32-bit offsets.
// Numerics: out = lrelu(((bias + residual) + front half) + back half): the fp32 terms of wdma.h's (sum + bias) + residual in another
// order (as wdma<256>'s K split already is); everything stored is rounded at the same points, the activated copy is computed from the
// rounded output.
#pragma once
#include "../gsv-tts-lite_amd/csrc/wdma.h"

#ifndef WPIPE_NOP
#define WPIPE_NOP 0
#endif

namespace gsv {

// The LDS-DMA instruction in its BUFFER form: `rsrc` = {base, base_hi (stride 0), bytes, 0x00020000}; a lane whose 32-bit byte offset is >= bytes
// is out of range and delivers zeros (a negative offset wraps above it).  Checked on MI355X: rows before the first / beyond the last row of the
// tensor arrive as zeros in LDS (tg_bench compares every row incl. the edges).
__device__ __forceinline__ u32x4 buffer_rsrc(const void* base, unsigned bytes) {
    const unsigned long long v = (unsigned long long)base;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)v);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}
__device__ __forceinline__ void dma16_buf(const u32x4& rsrc, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t store_rsrc(void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
}

template <int C, int MS, int BN, int NT, int ROLE>
__device__ __forceinline__ void wpipe_wave(const bf16_t* __restrict__ X, const uint4* __restrict__ W, const float* __restrict__ bias,
                                           const bf16_t* R, bf16_t* Y, bf16_t* A, int dil, int blk, int nblk, const WDmaArgs& a,
                                           unsigned char* lds, long long* dbg) {
    static_assert(C == 64 || C == 128, "64 or 128 channels");
    constexpr int KSTEPS = C / 16;
    constexpr int MT = C / 32;
    constexpr int RG = 4 / MS;                     // row groups of the block
    constexpr int WN = BN / 32 / RG;               // 32-row tiles per wave pair
    constexpr int RS = C * 2;                      // LDS bytes per staged row
    constexpr int VPR = C / 8;                     // 16-byte pieces per row
    constexpr int RPI = 64 / VPR;                  // rows per DMA instruction (1 KiB)
    constexpr int RPP = 4 * RPI;                   // rows per pass of the block's four FRONT waves (they own the DMAs)
    constexpr int RPB = VPR >= 16 ? 1 : 16 / VPR;  // rows per 256 bytes of LDS
    constexpr int SM = (VPR < 16 ? VPR : 16) - 1;  // swizzle mask
    static_assert((RPP / RPB) % (SM + 1) == 0, "the swizzle of a lane's row must not depend on the pass");
    constexpr int XROWS = BN + (NT - 1) * 5;
    constexpr int NPASS = (XROWS + RPP - 1) / RPP; // row DMA instructions per front wave per tile (fixed: the waits are counted)
    constexpr int XBYTES = ((BN + 50 + RPP - 1) / RPP) * RPP * RS;
    constexpr int NIT = NT * KSTEPS;
    constexpr int NA = (NIT + 1) / 2;              // the front wave's share of the (tap, k-step) walk
    constexpr int I0 = ROLE == 0 ? 0 : NA;
    constexpr int NF = ROLE == 0 ? NA : NIT - NA;  // fragments this wave holds
    constexpr int HBYTES = WN * 16 * 64 * 4;       // a partial tile (fp32 accumulators of a wave)
    constexpr int PBYTES = WN * 32 * 64;           // a wave pair's patch: [rows][4 pieces of its 32 channels] bf16, piece p of row r in slot p ^ ((r >> 2) & 3)
    constexpr int NVR = WN * 2;                    // patch vectors per lane

    const int tid = threadIdx.x, lane = tid & 63;
    const int sub = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);   // wave-uniform: every address built from it stays scalar
    const int j = lane & 31, hf = lane >> 5;
    const int ms = sub % MS, rg = sub / MS;
    const int wrow = rg * WN * 32;
    unsigned char* xbuf0 = lds;
    unsigned char* xbuf1 = lds + XBYTES;
    unsigned char* hand = lds + 2 * XBYTES + sub * HBYTES;                   // front -> back: partial tile (+ bias + residual)
    unsigned char* outp = lds + 2 * XBYTES + 4 * HBYTES + sub * PBYTES;      // back -> front: the rounded output rows
    unsigned char* outa = outp + 4 * PBYTES;                                 // back -> front: their activated copy
    unsigned char* resp = outp + 8 * PBYTES;                                 // front's own: the residual rows of the NEXT tile
    float* bl = reinterpret_cast<float*>(lds + 2 * XBYTES + 4 * HBYTES + 12 * PBYTES);

    const int n_rows = a.n_rows, ld = a.ld, ld2 = ld * 2;
    const int pad = (NT - 1) / 2 * dil;
    const int ntiles = (n_rows + BN - 1) / BN;
    const unsigned nbytes = (unsigned)n_rows * (unsigned)ld2;

    u32x4 w[NF];
    auto load_weights = [&]() {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int t = (I0 + i) / KSTEPS, ks = (I0 + i) % KSTEPS;
            w[i] = __builtin_bit_cast(u32x4, W[(((size_t)t * MT + ms) * KSTEPS + ks) * 64 + lane]);
        }
    };
    const float bias_raw = bias[tid < MS * 32 ? tid : 0];

    f32x16 acc[WN];
    auto mfma_half = [&](const unsigned char* xb) {
        const unsigned rowj = (unsigned)(wrow + j);
        const unsigned c0 = (unsigned)(hf * 16);
        int dl = dil;
        asm volatile("" : "+s"(dl));              // tap addresses are recomputed per tile, not kept across the loop
        if constexpr (ROLE == 1) {                  // the front wave's accumulators start from bias + residual (set_acc below)
#pragma unroll
            for (int k = 0; k < WN; ++k)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[k][q] = 0.f;
        }
        constexpr int DEPTH = 2;
        u32x4 bf[DEPTH + 1][WN];
        auto ldb = [&](int i, u32x4 (&dst)[WN]) {
            const int t = (I0 + i) / KSTEPS, ks = (I0 + i) % KSTEPS;
            const unsigned rt = rowj + (unsigned)(t * dl);
            const unsigned pre = rt * RS + (c0 ^ (((rt / RPB) & SM) << 4));
            const unsigned ad = pre ^ (unsigned)(ks * 32);
#pragma unroll
            for (int k = 0; k < WN; ++k) dst[k] = *reinterpret_cast<const u32x4*>(xb + ad + k * 32 * RS);
        };
#pragma unroll
        for (int i = 0; i < DEPTH && i < NF; ++i) ldb(i, bf[i % (DEPTH + 1)]);
        __builtin_amdgcn_sched_group_barrier(0x100, DEPTH * WN, 0);
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            if (i + DEPTH < NF) ldb(i + DEPTH, bf[(i + DEPTH) % (DEPTH + 1)]);
#pragma unroll
            for (int k = 0; k < WN; ++k) {
                Mma<bf16_t>::run(acc[k], w[i], bf[i % (DEPTH + 1)][k]);
#if WPIPE_NOP > 0
                asm volatile("s_nop %0" : : "n"(WPIPE_NOP - 1));
#endif
            }
            __builtin_amdgcn_sched_group_barrier(0x008, WN, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, WN, 0);
        }
    };

    int nst = 0;
    auto stamp = [&]() { if (dbg && blk == 0 && lane == 0 && sub == 0 && nst < 30) dbg[ROLE * 30 + nst] = (long long)__builtin_readcyclecounter(); ++nst; };

    if constexpr (ROLE == 0) {
        // =========================== front ===========================
        // rows: lane -> (row of the instruction's 1 KiB, slot); the piece it fetches is slot ^ f(row) (wdma.h)
        const int lr = lane / VPR, ls = lane % VPR;
        const int lane_off = (sub * RPI + lr) * ld2 + (ls ^ (((sub * RPI + lr) / RPB) & SM)) * 16;
        const u32x4 rsx = buffer_rsrc(X, nbytes);
        const u32x4 rsr = buffer_rsrc(R ? R : X, nbytes);
        const __amdgpu_buffer_rsrc_t rsy = store_rsrc(Y, nbytes);
        const __amdgpu_buffer_rsrc_t rsa = store_rsrc(A ? A : Y, nbytes);
        auto issue_x = [&](int tile, unsigned char* xb) {
            const unsigned xl = lds_addr_of(xb + sub * 1024);
            // the per-instruction part of the offset is SCALAR and opaque: left to itself hipcc keeps one hoisted vector offset per DMA / store
            // of the tile loop (24 registers this kernel does not have; two of them spilled, each reload a vmcnt(0) drain)
#pragma unroll
            for (int v = 0; v < NPASS; ++v) {
                int so = (tile * BN - pad + v * RPP) * ld2;
                asm volatile("" : "+s"(so));
                dma16_buf(rsx, (unsigned)(lane_off + so), xl + v * 4096);
            }
        };
        // patch vector q of lane l = patch bytes [16 (64 q + l), +16) = (row, slot); piece = slot ^ ((row >> 2) & 3): a DMA / store
        // instruction moves 16 rows x 64 bytes
        const int prow = lane >> 2, ppc = (lane & 3) ^ ((lane >> 4) & 3);
        const int patch_off = (wrow + prow) * ld2 + (ms * 32 + ppc * 8) * 2;
        auto issue_res = [&](int nb0) {
            const unsigned rl = lds_addr_of(resp);
#pragma unroll
            for (int q = 0; q < NVR; ++q) {
                int so = (nb0 + q * 16) * ld2;
                asm volatile("" : "+s"(so));
                dma16_buf(rsr, (unsigned)(patch_off + so), rl + q * 1024);
            }
        };
        // the rounded rows the back wave left in `outp` (and their activated copy in `outa`) -> Y, A: LDS read + buffer store, no arithmetic
        auto store_out = [&](int nb0) {
#pragma unroll
            for (int q = 0; q < NVR; ++q) {
                int so = (nb0 + q * 16) * ld2;
                asm volatile("" : "+s"(so));
                const int off = patch_off + so;
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(outp + (q * 64 + lane) * 16), rsy, off, 0, 0);
                if (A) __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(outa + (q * 64 + lane) * 16), rsa, off, 0, 0);
            }
        };
        auto hand_over = [&]() {
#pragma unroll
            for (int k = 0; k < WN; ++k)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 t = {acc[k][4 * q4], acc[k][4 * q4 + 1], acc[k][4 * q4 + 2], acc[k][4 * q4 + 3]};
                    *reinterpret_cast<f32x4*>(hand + ((k * 4 + q4) * 64 + lane) * 16) = t;
                }
        };
        // accumulators := bias (+ the residual rows waiting in `resp`): the lane's 16 channels of its row are pieces 2 hf, 2 hf + 1
        auto set_acc = [&]() {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + ms * 32 + 16 * hf + 4 * q4);
#pragma unroll
                for (int k = 0; k < WN; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[k][4 * q4 + e] = b4[e];
            }
            if (R) {
#pragma unroll
                for (int k = 0; k < WN; ++k) {
                    const int row = k * 32 + j, sw = (row >> 2) & 3;
                    const u32x4 ra = *reinterpret_cast<const u32x4*>(resp + row * 64 + ((2 * hf) ^ sw) * 16);
                    const u32x4 rc = *reinterpret_cast<const u32x4*>(resp + row * 64 + ((2 * hf + 1) ^ sw) * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[k][2 * e] += __uint_as_float(ra[e] << 16);
                        acc[k][2 * e + 1] += __uint_as_float(ra[e] & 0xffff0000u);
                        acc[k][8 + 2 * e] += __uint_as_float(rc[e] << 16);
                        acc[k][8 + 2 * e + 1] += __uint_as_float(rc[e] & 0xffff0000u);
                    }
                }
            }
        };

        if (R) issue_res(blk * BN);
        issue_x(blk, xbuf0);
        load_weights();
        if (tid < MS * 32) bl[tid] = bias_raw;
        wait_vm<0>();
        lds_barrier();                              // (bias and residual visible to this wave's own reads below; the rows to everyone)
        set_acc();
        lds_barrier();                              // B1
        int cur = 0, prev_nb0 = -1;
        for (int tile = blk; tile < ntiles; tile += nblk) {
            stamp();
            const int tn = tile + nblk;
            const bool has_next = tn < ntiles;
            // memory instructions first (they issue beside the back wave's MFMAs): DMAs of the next tile, the previous tile's rows out
            if (has_next) {
                if (R) issue_res(tn * BN);
                issue_x(tn, cur ? xbuf0 : xbuf1);
            }
            if (prev_nb0 >= 0) store_out(prev_nb0);
            stamp();
            mfma_half(cur ? xbuf1 : xbuf0);         // beside the back wave's half: the matrix pipe takes them alternately
            stamp();
            hand_over();
            lds_barrier();                          // B2: both halves of the tile are done, the front's is in LDS
            stamp();
            if (has_next) {                         // (vector work: cannot run beside MFMAs of this SIMD, so it sits between the barriers)
                if (prev_nb0 < 0) wait_vm<0>();     // DMAs landed (they are older than the stores)
                else if (A) wait_vm<2 * NVR>();
                else wait_vm<NVR>();
                set_acc();
            }
            lds_barrier();                          // B1: the back wave has left the tile's rounded rows; the next tile's rows are in LDS
            stamp();
            prev_nb0 = tile * BN;
            cur ^= 1;
        }
        store_out(prev_nb0);
    } else {
        // =========================== back ===========================
        load_weights();
        lds_barrier();
        lds_barrier();                              // B1
        // + the front's partial tile (which carries bias and residual), leaky-ReLU, round; the lane's 16 channels = pieces 2 hf, 2 hf + 1
        auto finish = [&]() {
#pragma unroll
            for (int k = 0; k < WN; ++k) {
                float v[16];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(hand + ((k * 4 + q4) * 64 + lane) * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * q4 + e] = t[e] + acc[k][4 * q4 + e];
                }
                if (a.out_slope != 1.0f) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) v[q] = lrelu(v[q], a.out_slope);
                }
                u32x4 oa, ob;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    oa[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                    ob[e] = pack_bf16x2(v[8 + 2 * e], v[8 + 2 * e + 1]);
                }
                const int row = k * 32 + j, sw = (row >> 2) & 3;
                *reinterpret_cast<u32x4*>(outp + row * 64 + ((2 * hf) ^ sw) * 16) = oa;
                *reinterpret_cast<u32x4*>(outp + row * 64 + ((2 * hf + 1) ^ sw) * 16) = ob;
                if (A) {                            // the activated copy, from the ROUNDED output
                    u32x4 ca, cb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ca[e] = pack_bf16x2(lrelu(__uint_as_float(oa[e] << 16), a.act_slope), lrelu(__uint_as_float(oa[e] & 0xffff0000u), a.act_slope));
                        cb[e] = pack_bf16x2(lrelu(__uint_as_float(ob[e] << 16), a.act_slope), lrelu(__uint_as_float(ob[e] & 0xffff0000u), a.act_slope));
                    }
                    *reinterpret_cast<u32x4*>(outa + row * 64 + ((2 * hf) ^ sw) * 16) = ca;
                    *reinterpret_cast<u32x4*>(outa + row * 64 + ((2 * hf + 1) ^ sw) * 16) = cb;
                }
            }
        };
        int cur = 0;
        for (int tile = blk; tile < ntiles; tile += nblk) {
            stamp();
            mfma_half(cur ? xbuf1 : xbuf0);
            stamp();
            lds_barrier();                          // B2
            stamp();
            finish();
            stamp();
            lds_barrier();                          // B1
            cur ^= 1;
        }
    }
}

template <int C, int MS, int BN>
__global__ __launch_bounds__(512, 1) void wpipe_kernel(WDmaArgs a) {
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
    if (blk >= (a.n_rows + BN - 1) / BN) return;
    if (dbg && b == 0 && (threadIdx.x & 63) == 0) dbg[64 + (threadIdx.x >> 6)] = (long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    const bool front = threadIdx.x < 256;           // waves s and s + 4 sit on one SIMD
#define GSV_WPIPE(NT)                                                                                         \
    if (front) wpipe_wave<C, MS, BN, NT, 0>(X, W, bias, R, Y, A, dil, blk, nblk, a, lds, dbg);                  \
    else wpipe_wave<C, MS, BN, NT, 1>(X, W, bias, R, Y, A, dil, blk, nblk, a, lds, dbg);
    if (k == 11) { GSV_WPIPE(11) }
    else if (k == 7) { GSV_WPIPE(7) }
    else if (k == 3) { GSV_WPIPE(3) }
#undef GSV_WPIPE
}

template <int C, int MS, int BN>
constexpr size_t wpipe_lds_bytes() {
    constexpr int RPP = 4 * (64 / (C / 8));
    constexpr int WN = BN / 32 / (4 / MS);
    return (size_t)2 * ((BN + 50 + RPP - 1) / RPP) * RPP * (C * 2) + (size_t)4 * WN * 16 * 64 * 4 + (size_t)12 * WN * 32 * 64 + MS * 32 * sizeof(float);
}

}  // namespace gsv
