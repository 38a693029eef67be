// EXPERIMENT (not part of the product): fused conv pair of a resblock, kept for tools/tg_bench.hip.
// Result on MI355X (bit-identical to the two-launch path): C=64 59.6 -> 55.6 us per pair, C=32 40.2 -> 41.1,
// C=16 41.3 -> 46.2 -- the small-channel convs are bound by per-row VALU/LDS work in staging and epilogues, not by
// the t1 round trip, and splitting the waves by role halves the parallelism of exactly that work.  Rejected.
#pragma once
#include <type_traits>
#include "../gsv-tts-lite_amd/csrc/wconv.h"

namespace gsv {

// ---- fused conv PAIR of a resblock (C <= 64): y = c2(lrelu(c1(lrelu(x)))) + x without t1 ever leaving the CU --------
// (ResBlock1.forward, modules.py:195-202: xt = lrelu(x); xt = c1(xt); xt = lrelu(xt); xt = c2(xt); x = xt + x).
// Weights in registers as above, but the block's 4 waves split by ROLE: waves 0-1 run c1, waves 2-3 run c2, each with
// ONE conv's weights, as a two-stage pipeline over the block's row tiles: in iteration i the c1 waves turn x(tile i)
// (LDS, staged with the leaky-ReLU) into lrelu(t1)(tile i) (LDS, double-buffered, rows outside the sequence zeroed --
// c2 pads t1, not x), while the c2 waves turn t1(tile i-1) into out(tile i-1) + residual.  One barrier per iteration.
// Both stages run on different SIMDs at the same time, global traffic is x in + out, and the result is bit-identical
// to the two-launch path (t1 is rounded to bf16 at the same point).
struct WPairArgs {
    const bf16_t *X0, *X1, *X2;   // resblock state x [n_rows][ld] (input of c1 AND the residual)
    const uint4 *Wa0, *Wa1, *Wa2; // c1 weights (fragment-packed), dilation d
    const uint4 *Wb0, *Wb1, *Wb2; // c2 weights, dilation 1
    const float *ba0, *ba1, *ba2, *bb0, *bb1, *bb2;
    bf16_t *Y0, *Y1, *Y2;
    int k0, k1, k2;               // taps of both convs of the branch (3, 7 or 11)
    int d0, d1, d2;               // dilation of c1
    int nb0, nb1, nb2;            // blocks per branch
    int ld, n_rows;
    float slope;                  // leaky-ReLU slope (0.1)
};

template <int C, int BN, int NT>
__device__ __forceinline__ void wpair_body(const bf16_t* __restrict__ X, const uint4* __restrict__ Wa, const uint4* __restrict__ Wb,
                                           const float* __restrict__ ba, const float* __restrict__ bb, bf16_t* Y, int dil, int blk,
                                           int nblk, int ld, int n_rows, float slope, unsigned char* lds) {
    constexpr int KSTEPS = C / 16;
    constexpr int MS = C >= 64 ? 2 : 1;            // 32-channel slices (waves along channels inside a role)
    constexpr int RG = 2 / MS;                     // row groups inside a role
    constexpr int MT = (C + 31) / 32;
    constexpr int P2 = (NT - 1) / 2;               // halo of c2 (dilation 1)
    constexpr int R1 = ((BN + 2 * P2 + 32 * RG - 1) / (32 * RG)) * (32 * RG);   // t1 rows computed per tile
    constexpr int WN1 = R1 / 32 / RG, WN2 = BN / 32 / RG;
    constexpr int XRS = C * 2 + 16;
    constexpr int XROWS = R1 + (NT - 1) * 5;       // x rows staged at the largest c1 dilation
    constexpr int XBYTES = XROWS * XRS;
    constexpr int TBYTES = R1 * XRS;               // one t1 buffer
    constexpr int VPR = C / 8, RPP = 256 / VPR;
    constexpr int NVX = (XROWS + RPP - 1) / RPP;
    constexpr int CW = C < 32 ? C : 32;
    constexpr int PCS = CW / 8;
    constexpr int RORS = 32 * 2 + 16;
    constexpr int NVR = WN2 * 32 * PCS / 64;
    constexpr int ROBYTES = WN2 * 32 * RORS;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = lane & 31, hf = lane >> 5;
    const int role = wid >> 1, sub = wid & 1;      // role 0: c1, role 1: c2
    const int ms = sub % MS, rg = sub / MS;
    unsigned char* xbuf0 = lds;
    unsigned char* xbuf1 = lds + XBYTES;
    unsigned char* tbuf0 = lds + 2 * XBYTES;
    unsigned char* tbuf1 = tbuf0 + TBYTES;
    unsigned char* ro = tbuf1 + TBYTES + sub * ROBYTES;
    float* bl = reinterpret_cast<float*>(tbuf1 + TBYTES + 2 * ROBYTES);      // [2][MS*32]: c1 bias, c2 bias

    const int p1 = P2 * dil;
    const int xrows = R1 + (NT - 1) * dil;         // x rows a tile needs: t1 row i reads x rows i + t*dil
    const int ntiles = (n_rows + BN - 1) / BN;
    if (blk >= ntiles) return;
    const int mytiles = (ntiles - blk + nblk - 1) / nblk;

    // the wave's weights: c1's or c2's slice
    const uint4* W = role == 0 ? Wa : Wb;
    u32x4 w[NT][KSTEPS];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
            w[t][ks] = __builtin_bit_cast(u32x4, W[(((size_t)t * MT + ms) * KSTEPS + ks) * 64 + lane]);
    if (tid < 2 * MS * 32) {
        const int which = tid / (MS * 32), c = tid % (MS * 32);
        const float* bsrc = which == 0 ? ba : bb;
        bl[tid] = (bsrc && c < C) ? bsrc[c] : 0.f;
    }

    // x tile of `tile`: LDS row r <-> global row tile*BN - P2 - p1 + r
    const int cv = tid % VPR, r0 = tid / VPR;
    u32x4 xraw[NVX];
    auto issue_x = [&](int tile) {
        const int gbase = tile * BN - P2 - p1;
#pragma unroll
        for (int v = 0; v < NVX; ++v) {
            if (v * RPP < xrows) {
                const int grow = gbase + r0 + v * RPP;
                const bool ok = grow >= 0 && grow < n_rows;
                xraw[v] = *reinterpret_cast<const u32x4*>(X + (size_t)(ok ? grow : 0) * ld + cv * 8);
            }
        }
    };
    auto commit_x = [&](int tile, unsigned char* xb) {
        const int gbase = tile * BN - P2 - p1;
#pragma unroll
        for (int v = 0; v < NVX; ++v) {
            if (v * RPP < xrows) {
                const int r = r0 + v * RPP;
                const int grow = gbase + r;
                const bool ok = grow >= 0 && grow < n_rows;
                if (r < xrows) *reinterpret_cast<u32x4*>(xb + (size_t)r * XRS + cv * 16) = Stage16<bf16_t, bf16_t>::finish(xraw[v], ok, slope);
            }
        }
    };
    u32x4 rraw[NVR];
    auto patch_rc = [&](int p, int& row, int& pc) {
        const int idx = p * 64 + lane;
        row = idx / PCS;
        pc = idx % PCS;
    };
    // MFMA sweep of one conv over LDS rows: WNx 32-row tiles starting at `row0`, taps stepping `tstep` rows
    auto sweep = [&](auto& acc, const unsigned char* src, int row0, int tstep, auto wn_tag) {
        constexpr int WNX = decltype(wn_tag)::value;
        constexpr int NIT = NT * KSTEPS;
        constexpr int DEPTH = 3;
        const unsigned lb = (unsigned)(row0 + j) * XRS + hf * 16;
#pragma unroll
        for (int k = 0; k < WNX; ++k)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[k][q] = 0.f;
        u32x4 bf[DEPTH + 1][WNX];
        auto ldb = [&](int it, u32x4 (&dst)[WNX]) {
            const unsigned tb = lb + (unsigned)((it / KSTEPS) * tstep) * XRS + (it % KSTEPS) * 32;
#pragma unroll
            for (int k = 0; k < WNX; ++k) dst[k] = *reinterpret_cast<const u32x4*>(src + tb + k * 32 * XRS);
        };
#pragma unroll
        for (int it = 0; it < DEPTH && it < NIT; ++it) ldb(it, bf[it % (DEPTH + 1)]);
        __builtin_amdgcn_sched_group_barrier(0x100, DEPTH * WNX, 0);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (it + DEPTH < NIT) ldb(it + DEPTH, bf[(it + DEPTH) % (DEPTH + 1)]);
#pragma unroll
            for (int k = 0; k < WNX; ++k) Mma<bf16_t>::run(acc[k], w[it / KSTEPS][it % KSTEPS], bf[it % (DEPTH + 1)][k]);
            __builtin_amdgcn_sched_group_barrier(0x008, WNX, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, WNX, 0);
        }
    };

    issue_x(blk);
    commit_x(blk, xbuf0);
    __syncthreads();
    // iteration i: c1 on tile(i) (i < mytiles), c2 on tile(i-1) (i >= 1)
    for (int i = 0; i <= mytiles; ++i) {
        const int tile1 = blk + i * nblk;                  // c1's tile
        const int tile2 = tile1 - nblk;                    // c2's tile
        const bool has_next_x = i + 1 < mytiles;
        if (has_next_x) issue_x(tile1 + nblk);
        unsigned char* xb = (i & 1) ? xbuf1 : xbuf0;
        unsigned char* tw = (i & 1) ? tbuf1 : tbuf0;       // t1 written this iteration
        unsigned char* tr = (i & 1) ? tbuf0 : tbuf1;       // t1 of the previous iteration
        if (role == 0) {
            if (i < mytiles) {
                f32x16 acc[WN1];
                sweep(acc, xb, rg * WN1 * 32, dil, std::integral_constant<int, WN1>{});
                // t1 = lrelu(acc + b1), zero outside the sequence, bf16 -> LDS rows of this wave's 32 channels
                const int g0 = tile1 * BN - P2;
#pragma unroll
                for (int k = 0; k < WN1; ++k) {
                    const int row = rg * WN1 * 32 + k * 32 + j;
                    const int grow = g0 + row;
                    const bool ok = grow >= 0 && grow < n_rows;
                    float v[16];
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + ms * 32 + 16 * hf + 4 * q4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * q4 + e] = ok ? lrelu(acc[k][4 * q4 + e] + b4[e], slope) : 0.f;
                    }
                    u32x4 oa, ob;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        oa[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                        ob[e] = pack_bf16x2(v[8 + 2 * e], v[8 + 2 * e + 1]);
                    }
                    if (ms * 32 + 16 * hf < C) {           // C = 16: the upper lane half holds channels that do not exist
                        unsigned char* pp = tw + (size_t)row * XRS + (ms * 32 + 16 * hf) * 2;
                        *reinterpret_cast<u32x4*>(pp) = oa;
                        *reinterpret_cast<u32x4*>(pp + 16) = ob;
                    }
                }
            }
        } else {
            if (i >= 1) {
                const int nb0 = tile2 * BN;
                const int wrow = rg * WN2 * 32;
#pragma unroll
                for (int p = 0; p < NVR; ++p) {            // residual = the raw state rows of the tile
                    int row, pc;
                    patch_rc(p, row, pc);
                    const int n = min(nb0 + wrow + row, n_rows - 1);
                    rraw[p] = *reinterpret_cast<const u32x4*>(X + (size_t)n * ld + ms * 32 + pc * 8);
                }
                f32x16 acc[WN2];
                sweep(acc, tr, wrow, 1, std::integral_constant<int, WN2>{});
#pragma unroll
                for (int p = 0; p < NVR; ++p) {
                    int row, pc;
                    patch_rc(p, row, pc);
                    *reinterpret_cast<u32x4*>(ro + row * RORS + pc * 16) = rraw[p];
                }
#pragma unroll
                for (int k = 0; k < WN2; ++k) {
                    unsigned char* pp = ro + (k * 32 + j) * RORS + hf * 32;
                    float v[16];
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + MS * 32 + ms * 32 + 16 * hf + 4 * q4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * q4 + e] = acc[k][4 * q4 + e] + b4[e];
                    }
                    const u32x4 ra = *reinterpret_cast<const u32x4*>(pp), rb = *reinterpret_cast<const u32x4*>(pp + 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] += __uint_as_float(ra[e] << 16);
                        v[2 * e + 1] += __uint_as_float(ra[e] & 0xffff0000u);
                        v[8 + 2 * e] += __uint_as_float(rb[e] << 16);
                        v[8 + 2 * e + 1] += __uint_as_float(rb[e] & 0xffff0000u);
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
                    if (n < n_rows) *reinterpret_cast<u32x4*>(Y + (size_t)n * ld + ms * 32 + pc * 8) = o;
                }
            }
        }
        if (has_next_x) commit_x(tile1 + nblk, (i & 1) ? xbuf0 : xbuf1);
        __syncthreads();
    }
}

template <int C, int BN>
constexpr size_t wpair_lds_bytes() {
    constexpr int MS = C >= 64 ? 2 : 1, RG = 2 / MS, P2 = 5;
    constexpr int R1 = ((BN + 2 * P2 + 32 * RG - 1) / (32 * RG)) * (32 * RG);
    return (size_t)2 * (R1 + 50) * (C * 2 + 16) + (size_t)2 * R1 * (C * 2 + 16) + (size_t)2 * (BN / 32 / RG) * 32 * 80 + 2 * MS * 32 * 4;
}

template <int C, int BN>
__global__ __launch_bounds__(256, 1) void wpair_kernel(WPairArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int b = blockIdx.x;
    const int br = b < a.nb0 ? 0 : (b < a.nb0 + a.nb1 ? 1 : 2);
    const int blk = br == 0 ? b : (br == 1 ? b - a.nb0 : b - a.nb0 - a.nb1);
    const int nblk = br == 0 ? a.nb0 : (br == 1 ? a.nb1 : a.nb2);
    const bf16_t* X = br == 0 ? a.X0 : (br == 1 ? a.X1 : a.X2);
    const uint4* Wa = br == 0 ? a.Wa0 : (br == 1 ? a.Wa1 : a.Wa2);
    const uint4* Wb = br == 0 ? a.Wb0 : (br == 1 ? a.Wb1 : a.Wb2);
    const float* ba = br == 0 ? a.ba0 : (br == 1 ? a.ba1 : a.ba2);
    const float* bb = br == 0 ? a.bb0 : (br == 1 ? a.bb1 : a.bb2);
    bf16_t* Y = br == 0 ? a.Y0 : (br == 1 ? a.Y1 : a.Y2);
    const int k = br == 0 ? a.k0 : (br == 1 ? a.k1 : a.k2);
    const int dil = br == 0 ? a.d0 : (br == 1 ? a.d1 : a.d2);
    if (k == 11) wpair_body<C, BN, 11>(X, Wa, Wb, ba, bb, Y, dil, blk, nblk, a.ld, a.n_rows, a.slope, lds);
    else if (k == 7) wpair_body<C, BN, 7>(X, Wa, Wb, ba, bb, Y, dil, blk, nblk, a.ld, a.n_rows, a.slope, lds);
    else if (k == 3) wpair_body<C, BN, 3>(X, Wa, Wb, ba, bb, Y, dil, blk, nblk, a.ld, a.n_rows, a.slope, lds);
}

}  // namespace gsv
