// wconv: the Generator's resblock convolutions (ResBlock1, module/modules.py:190-203 of the reference:
// Conv1d(C, C, k, dilation d, "same" padding) on [time][channel] bf16 activations) as a persistent
// MFMA kernel for gfx950 whose WEIGHTS LIVE IN REGISTERS.
//
// Why: measured on MI355X, the generic tapgemm spends as long fetching weight fragments through the
// CU's 64 B/clk L1 path as it spends in the matrix pipe, and its staging / MFMA / epilogue phases
// do not overlap across co-resident blocks (they queue on the same memory path).  A resblock conv
// has few weights (C*C*k <= 128*128*11) and very many rows (40k-320k), so here
//   * one block per CU, 4 waves, 512 registers each: a wave owns one 32-channel output slice and
//     keeps that slice's whole weight set (k * C/16 fragments, <= 352 VGPR/AGPRs) for its lifetime;
//   * the block walks row tiles; the MFMA loop touches only LDS (B fragments) and registers -- no
//     global load, no vmcnt wait inside it;
//   * the next tile's rows are prefetched global -> registers while the current tile computes, and
//     committed (leaky-ReLU applied once per element) into the other half of a double-buffered LDS tile;
//   * residual and output move through a wave-private LDS patch so that global loads and stores are
//     16-byte lanes in 64-byte runs instead of the D fragment's scattered 8-byte pieces.
// Up to 3 convolutions of one shape class (the three resblock branches, k = 3 / 7 / 11) run in one
// launch; blocks are dealt to branches in proportion to their cost.
// Wider convs (C = 192, 256: a 32-channel slice of 11 taps is 132 / 176 fragments, more than a wave's
// registers) split the contraction between two waves (KSP = 2: each holds half of the slice's k-steps, the
// partial tiles meet in LDS) and the output slices between MSP blocks that walk the same row tiles.
#pragma once
#include "tapgemm.h"

namespace gsv {

struct WConvArgs {
    const bf16_t *X0, *X1, *X2;   // inputs  [n_rows][ld]
    const uint4 *W0, *W1, *W2;    // tapgemm fragment-packed weights ([tap][mtile][kstep][lane])
    const float *b0, *b1, *b2;    // bias [C] or null
    const bf16_t *R0, *R1, *R2;   // residual [n_rows][ld] or null
    bf16_t *Y0, *Y1, *Y2;         // outputs [n_rows][ld]
    int k0, k1, k2;               // taps (3, 7 or 11)
    int d0, d1, d2;               // dilation (<= 5), "same" padding
    int nb0, nb1, nb2;            // blocks dealt to each branch (grid.x = nb0 + nb1 + nb2)
    int ld, n_rows;
    float in_slope;               // leaky-ReLU on the input (1 = none)
    float out_slope;              // leaky-ReLU on the output (1 = none)
    int cout;                     // real output channels when rows are padded to C (bias has only `cout` entries); 0 = C
    long long* dbg;               // null, or cycle stamps of block 0 / wave 0 per tile phase (tools/tg_bench)
};

// C: channels (cin == cout), MS: 32-channel output slices per block (waves along channels),
// BN: rows per tile.  Waves: MS slices x (4/MS) row groups, each wave 32 channels x (WN*32) rows.
template <int C, int MS, int BN, int NT, int KSP = 1, int MSP = 1>
__device__ __forceinline__ void wconv_body(const bf16_t* __restrict__ X, const uint4* __restrict__ W,
                                           const float* __restrict__ bias, const bf16_t* R, bf16_t* Y, int dil, int blk,
                                           int nblk, int ld, int n_rows, float in_slope, float out_slope, int cout,
                                           unsigned char* lds, long long* dbg = nullptr) {
    constexpr int KSTEPS = C / 16;
    constexpr int MT = (C + 31) / 32;             // m-tiles in the packed weights
    constexpr int RG = 4 / (MS * KSP);            // row groups
    constexpr int KSW = KSTEPS / KSP;             // k-steps of a slice held by one wave
    constexpr int WN = BN / 32 / RG;              // 32-row tiles per wave
    constexpr int XRS = C * 2 + 16;               // LDS bytes per staged row (16-byte skew: conflict-free 32-row reads)
    constexpr int XROWS = BN + (NT - 1) * 5;      // rows staged at the largest dilation
    constexpr int XBYTES = XROWS * XRS;
    constexpr int VPR = C / 8;                    // 16-byte vectors per row
    constexpr int RPP = 256 / VPR;                // rows per staging pass
    constexpr int NVX = (XROWS + RPP - 1) / RPP;  // staging vectors per thread
    constexpr int CW = C < 32 ? C : 32;           // channels of a full slice
    constexpr int PCS = CW / 8;                   // 16-byte pieces per row of the slice (the last slice of C = 48 has
                                                  // fewer that exist: `pv` below; C = 96 runs as 4 slices, the 4th idle)
    constexpr int RORS = 32 * 2 + 16;             // residual/output patch: bytes per row
    constexpr int NVR = WN * 32 * PCS / 64;       // patch vectors per lane
    constexpr int ROBYTES = WN * 32 * RORS;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = lane & 31, hf = lane >> 5;
    const int ms = wid % MS, kh = (wid / MS) % KSP, rg = wid / (MS * KSP);
    const int mg = MSP > 1 ? blk % MSP : 0;       // this block's group of MS output slices
    const int gs = mg * MS + ms;                  // the wave's slice of the conv
    if (MSP > 1) { blk /= MSP; nblk /= MSP; }     // the MSP blocks of a walker visit the same tiles
    const bool live = gs < MT;                    // a wave whose slice does not exist only helps staging
    const int wrow = rg * WN * 32;
    unsigned char* xbuf0 = lds;
    unsigned char* xbuf1 = lds + XBYTES;
    unsigned char* ro = lds + 2 * XBYTES + wid * ROBYTES;

    const int pad = (NT - 1) / 2 * dil;
    const int rows = BN + (NT - 1) * dil;
    const int ntiles = (n_rows + BN - 1) / BN;
    if (blk >= ntiles) return;

    // the wave's weights: every (tap, k-step) fragment of its 32-channel slice.  Issued BEHIND the first tile's rows and the bias
    // (issue_x below): in front of them, the bias store to LDS drained the load counter -- all of the block's 90-350 KB of weights --
    // before a single row of the first tile was requested (hipcc places the wait at the first use of the LAST load issued)
    u32x4 w[NT][KSW];
    const int msw = live ? gs : 0;
    auto load_weights = [&]() {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks)
                w[t][ks] = __builtin_bit_cast(u32x4, W[(((size_t)t * MT + msw) * KSTEPS + kh * KSW + ks) * 64 + lane]);
    };
    // bias sits in LDS (registers are for weights): [MS*32] floats behind the patches
    float* bl = reinterpret_cast<float*>(lds + 2 * XBYTES + 4 * ROBYTES);
    float bias_raw = 0.f;                       // loaded from a clamped address now, masked at its use (a select here is a use)
    if (bias != nullptr) bias_raw = bias[min(mg * MS * 32 + (tid < MS * 32 ? tid : 0), cout - 1)];
    // KSP = 2: the second half's partial tiles, [MS*RG waves][WN][16][64] floats behind the bias
    float* kred = reinterpret_cast<float*>(lds + 2 * XBYTES + 4 * ROBYTES + MS * 32 * sizeof(float)) + (size_t)(rg * MS + ms) * WN * 16 * 64;

    // staging map: thread -> (first row, 16-byte column)
    const int cv = tid % VPR, r0 = tid / VPR;
    u32x4 xraw[NVX];
    auto issue_x = [&](int tile) {
        const int gbase = tile * BN - pad;
#pragma unroll
        for (int v = 0; v < NVX; ++v) {
            if (v * RPP < rows) {
                const int grow = gbase + r0 + v * RPP;
                const bool ok = grow >= 0 && grow < n_rows;
                xraw[v] = *reinterpret_cast<const u32x4*>(X + (size_t)(ok ? grow : 0) * ld + cv * 8);
            }
        }
    };
    auto commit_x = [&](int tile, unsigned char* xb) {
        const int gbase = tile * BN - pad;
#pragma unroll
        for (int v = 0; v < NVX; ++v) {
            if (v * RPP < rows) {
                const int r = r0 + v * RPP;
                const int grow = gbase + r;
                const bool ok = grow >= 0 && grow < n_rows;
                if (r < rows) *reinterpret_cast<u32x4*>(xb + (size_t)r * XRS + cv * 16) = Stage16<bf16_t, bf16_t>::finish(xraw[v], ok, in_slope);
            }
        }
    };
    // wave-private residual / output patch: lane -> (row, 16-byte piece) in 64-byte runs
    u32x4 rraw[NVR];
    auto patch_rc = [&](int p, int& row, int& pc) {
        const int idx = p * 64 + lane;
        row = idx / PCS;
        pc = idx % PCS;
    };
    auto piece_ok = [&](int pc) { return gs * 32 + pc * 8 < C; };
    const bool epi = live && kh == 0;             // the wave that owns the tile's epilogue

    issue_x(blk);
    load_weights();
    if (tid < MS * 32) bl[tid] = mg * MS * 32 + tid < cout ? bias_raw : 0.f;
    commit_x(blk, xbuf0);
    __syncthreads();
    int cur = 0, nst = 0;
    auto stamp = [&]() { if (dbg && blk == 0 && tid == 0 && nst < 60) dbg[nst] = (long long)__builtin_readcyclecounter(); ++nst; };
    for (int tile = blk; tile < ntiles; tile += nblk) {
        stamp();
        const int tn = tile + nblk;
        const bool has_next = tn < ntiles;
        const int nb0 = tile * BN;
        if (R && epi) {
#pragma unroll
            for (int p = 0; p < NVR; ++p) {
                int row, pc;
                patch_rc(p, row, pc);
                const int n = min(nb0 + wrow + row, n_rows - 1);
                rraw[p] = *reinterpret_cast<const u32x4*>(R + (size_t)n * ld + gs * 32 + (piece_ok(pc) ? pc : 0) * 8);
            }
        }
        if (has_next) issue_x(tn);

        stamp();
        f32x16 acc[WN];
        if (live) {
        // ---- MFMA loop: LDS + registers only
        const unsigned char* xb = cur ? xbuf1 : xbuf0;
        const unsigned lb = (unsigned)(wrow + j) * XRS + hf * 16 + kh * KSW * 32;
#pragma unroll
        for (int k = 0; k < WN; ++k)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[k][q] = 0.f;
        {
            // flat (tap, k-step) walk; B fragments are read DEPTH iterations ahead of their MFMAs
            constexpr int NIT = NT * KSW;
            constexpr int DEPTH = 3;
            u32x4 bf[DEPTH + 1][WN];
            auto ldb = [&](int it, u32x4 (&dst)[WN]) {
                const unsigned tb = lb + (unsigned)((it / KSW) * dil) * XRS + (it % KSW) * 32;
#pragma unroll
                for (int k = 0; k < WN; ++k) dst[k] = *reinterpret_cast<const u32x4*>(xb + tb + k * 32 * XRS);
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

        }   // live: MFMA
        if constexpr (KSP > 1) {   // the second half hands its partial tiles over
            if (live && kh == 1) {
#pragma unroll
                for (int k = 0; k < WN; ++k)
#pragma unroll
                    for (int q = 0; q < 16; ++q) kred[(k * 16 + q) * 64 + lane] = acc[k][q];
            }
            __syncthreads();
            if (epi) {
#pragma unroll
                for (int k = 0; k < WN; ++k)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[k][q] += kred[(k * 16 + q) * 64 + lane];
            }
        }

        stamp();
        if (epi) {
        // ---- epilogue, wave-private
        if (R) {
#pragma unroll
            for (int p = 0; p < NVR; ++p) {
                int row, pc;
                patch_rc(p, row, pc);
                *reinterpret_cast<u32x4*>(ro + row * RORS + pc * 16) = rraw[p];
            }
        }
#pragma unroll
        for (int k = 0; k < WN; ++k) {
            unsigned char* pp = ro + (k * 32 + j) * RORS + hf * 32;
            float v[16];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + ms * 32 + 16 * hf + 4 * q4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * q4 + e] = acc[k][4 * q4 + e] + b4[e];
            }
            if (out_slope != 1.0f) {
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = lrelu(v[q], out_slope);
            }
            if (R) {
                const u32x4 ra = *reinterpret_cast<const u32x4*>(pp), rb = *reinterpret_cast<const u32x4*>(pp + 16);
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
            *reinterpret_cast<u32x4*>(pp) = oa;
            *reinterpret_cast<u32x4*>(pp + 16) = ob;
        }
#pragma unroll
        for (int p = 0; p < NVR; ++p) {
            int row, pc;
            patch_rc(p, row, pc);
            const u32x4 o = *reinterpret_cast<const u32x4*>(ro + row * RORS + pc * 16);
            const int n = nb0 + wrow + row;
            if (n < n_rows && piece_ok(pc)) *reinterpret_cast<u32x4*>(Y + (size_t)n * ld + gs * 32 + pc * 8) = o;
        }
        }   // epi

        stamp();
        if (has_next) commit_x(tn, cur ? xbuf0 : xbuf1);
        stamp();
        __syncthreads();   // next tile's rows visible; everyone is done with this tile's
        cur ^= 1;
    }
}

template <int C, int MS, int BN, int KSP = 1, int MSP = 1>
__global__ __launch_bounds__(256, 1) void wconv_kernel(WConvArgs a) {
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
    const int k = br == 0 ? a.k0 : (br == 1 ? a.k1 : a.k2);
    const int dil = br == 0 ? a.d0 : (br == 1 ? a.d1 : a.d2);
    const int cout = a.cout > 0 ? a.cout : C;
    if (k == 11) wconv_body<C, MS, BN, 11, KSP, MSP>(X, W, bias, R, Y, dil, blk, nblk, a.ld, a.n_rows, a.in_slope, a.out_slope, cout, lds, br == 0 ? a.dbg : nullptr);
    else if (k == 7) wconv_body<C, MS, BN, 7, KSP, MSP>(X, W, bias, R, Y, dil, blk, nblk, a.ld, a.n_rows, a.in_slope, a.out_slope, cout, lds, br == 0 ? a.dbg : nullptr);
    else if (k == 3) wconv_body<C, MS, BN, 3, KSP, MSP>(X, W, bias, R, Y, dil, blk, nblk, a.ld, a.n_rows, a.in_slope, a.out_slope, cout, lds, br == 0 ? a.dbg : nullptr);
}

// LDS bytes of a wconv_kernel<C, MS, BN> launch (sized for 11 taps at dilation 5)
template <int C, int MS, int BN, int KSP = 1, int MSP = 1>
constexpr size_t wconv_lds_bytes() {
    // staged rows (two buffers) + 4 wave patches of WN*32 rows + bias + (KSP = 2) the partial tiles
    return (size_t)2 * (BN + 50) * (C * 2 + 16) + (size_t)4 * (BN / (4 / (MS * KSP))) * (32 * 2 + 16) + MS * 32 * sizeof(float) +
           (KSP > 1 ? (size_t)(4 / KSP) * (BN / (4 / (MS * KSP)) / 32) * 16 * 64 * sizeof(float) : 0);
}

}  // namespace gsv
