// flowfuse: one ResidualCouplingLayer of the SoVITS flow in reverse mode (reference
// SoVITS/module/modules.py:482-501 with WN modules.py:80-104 and Flip :504-511) as ONE kernel on gfx950.
//
// The unfused path runs 19 launches per coupling layer (pre, cond, 4x{k5 conv, gate, skip 1x1, res 1x1},
// post, flip), each a few microseconds of latency around microseconds of work at T = 500 frames.
// Rows are independent through a whole coupling layer except for the k=5 convolutions' +-2 halo per
// WN layer, so a block takes 64 consecutive rows (48 valid + 8 halo each side), keeps h / acts in LDS
// and the skip sum in fp32 registers, and streams the layer's 3.5 MB of fragment-packed weights once.
//
// Flip is folded away: the activation tensor P is never permuted; parity-1 layers read their conv
// input half / write their updated half at swapped offsets with channel-reversed pre / post weights.
//
// bf16 activations (h, acts, out as in the unfused bf16 path), fp32 accumulation and fp32 skip sum.
#pragma once
#include "tapgemm.h"

namespace gsv {

enum {
    FF_H = 192, FF_HALF = 96, FF_NR = 64, FF_VR = 48, FF_HALO = 8, FF_KSH = 12, FF_KSP = 6,
    // weight arena offsets in fragments (1 fragment = 64 lanes x 16 B)
    FF_W_PRE = 0,                                  // [6 mt][6 ks]
    FF_W_IN = FF_W_PRE + 6 * 6,                    // [4 layers][5 taps][12 mt][12 ks]
    FF_W_IN_L = 5 * 12 * 12,
    FF_W_RES = FF_W_IN + 4 * FF_W_IN_L,            // [3 layers][6 mt][12 ks]
    FF_W_SKIP = FF_W_RES + 3 * 6 * 12,             // [4 layers][6 mt][12 ks]
    FF_W_POST = FF_W_SKIP + 4 * 6 * 12,            // [3 mt][12 ks]   (negated)
    FF_W_TOTAL = FF_W_POST + 3 * 12,
    // bias arena = the kernel's LDS table, in floats: in-layer biases, res biases, SUM of the four skip
    // biases (filled by ff_skip_bias_sum_kernel), post (negated), pre
    FF_T_IN = 0, FF_T_RES = 4 * 384, FF_T_SKIP = FF_T_RES + 3 * 192, FF_T_POST = FF_T_SKIP + 192, FF_T_PRE = FF_T_POST + 96,
    FF_T_TOTAL = FF_T_PRE + 192,
    // LDS layout (bytes)
    FF_HRS = FF_H * 2 + 16,                        // row stride of h / acts (16-byte skew)
    FF_XRS = FF_HALF * 2 + 16,
    FF_LDS_H = 0,                                  // 68 rows: tile rows -2 .. 65
    FF_LDS_A = FF_LDS_H + 68 * FF_HRS,
    FF_LDS_X = FF_LDS_A + 64 * FF_HRS,
    FF_LDS_M = FF_LDS_X + 64 * FF_XRS,
    FF_LDS_B = FF_LDS_M + 64 * 4,                  // float tables: in-layer bias (+ broadcast conditioning), res, skip sum, post, pre
    FF_LDS_TOTAL = FF_LDS_B + FF_T_TOTAL * 4
};

struct FlowFuseArgs {
    bf16_t* P;            // [T][192] activations, the updated half is rewritten in place
    const float* mask;    // [T]
    const float* gc;      // conditioning for this layer's WN: [1 or T][ldg], layer l at +l*384
    int ldg;              // 0 = one broadcast row
    const int* seg;       // null, or: frame g takes conditioning row seg[g] (few distinct ge columns: voc_kernels.h)
    const uint4* W;       // weight arena (FF_W_*)
    const float* B;       // bias arena (FF_T_*)
    int T;
    int xin_off, xup_off; // physical channel offsets of the conv-input half and of the updated half
    int per_xcd;          // row tiles per XCD: block b works on tile (b % 8) * per_xcd + b / 8 (grid = 8 * per_xcd)
    long long* dbg;       // null, or 32 cycle stamps of block 0 / wave 0 (GSV_FF_DEBUG)
    int rot_k;            // 1 = tile t walks the k-steps rotated by t % 12, 0 = every tile in the same order (A/B switch)
};

// gate non-linearities on the hardware exp2 / rcp units (1 ulp each; the result is rounded to bf16)
__device__ __forceinline__ float ff_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float ff_tanh(float x) {
    // 1 - 2 / (1 + e^{2x}); e^{2x} -> inf gives 1, -> 0 gives -1
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}
__device__ __forceinline__ void ff_unpack16(const u32x4& a, const u32x4& b, float (&v)[16]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[2 * e] = __uint_as_float(a[e] << 16);
        v[2 * e + 1] = __uint_as_float(a[e] & 0xffff0000u);
        v[8 + 2 * e] = __uint_as_float(b[e] << 16);
        v[8 + 2 * e + 1] = __uint_as_float(b[e] & 0xffff0000u);
    }
}
__device__ __forceinline__ void ff_pack16(const float (&v)[16], u32x4& a, u32x4& b) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        a[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
        b[e] = pack_bf16x2(v[8 + 2 * e], v[8 + 2 * e + 1]);
    }
}

static __global__ __launch_bounds__(256, 1) void flowfuse_kernel(FlowFuseArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* Hb = lds + FF_LDS_H;
    unsigned char* Ab = lds + FF_LDS_A;
    unsigned char* Xb = lds + FF_LDS_X;
    float* mk = reinterpret_cast<float*>(lds + FF_LDS_M);
    float* tb = reinterpret_cast<float*>(lds + FF_LDS_B);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = lane & 31, hf = lane >> 5;
    // Consecutive row tiles go to ONE XCD (workgroup b is dispatched to XCD b % 8): they all stream the same
    // weights, so the XCD's L2 fetches each line once and serves its CUs at L2 rate; a CU on its own gets
    // only ~10 B/clk from the fabric (measured: 156 us per layer at 11 blocks spread over 8 XCDs).
    // Placement is a speed matter only -- any mapping of tiles to blocks is correct.
    const int tile = ((int)blockIdx.x % 8) * a.per_xcd + (int)blockIdx.x / 8;
    if ((int)blockIdx.x / 8 >= a.per_xcd || tile * FF_VR >= a.T) return;
    const int g0 = tile * FF_VR - FF_HALO;                  // global row of tile row 0
    const uint4* Wl = a.W + lane;
    int nstamp = 0;
    auto stamp = [&]() {
        if (a.dbg && tile == 0 && tid == 0) a.dbg[nstamp] = (long long)__builtin_readcyclecounter();
        ++nstamp;
    };
    stamp();

    // the wave's three (channel tile m, row tile n) units: m = mA for both row tiles, m = mB for row tile nB
    const int mA = wid == 0 ? 0 : (wid == 1 ? 2 : (wid == 2 ? 3 : 5));
    const int mB = wid < 2 ? 1 : 4;
    const int nB = wid & 1;

    // ---- phase 0: stage the conv-input half, the mask; clear the guard rows of h
    {
        u32x4 xr[3];
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const int idx = tid + v * 256;                 // 64 rows x 12 vectors
            const int r = idx / 12, c = idx % 12;
            const int g = g0 + r;
            const bool ok = g >= 0 && g < a.T;
            xr[v] = *reinterpret_cast<const u32x4*>(a.P + (size_t)(ok ? g : 0) * FF_H + a.xin_off + c * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) xr[v][e] = ok ? xr[v][e] : 0u;
        }
        if (tid < 64) {
            const int g = g0 + tid;
            mk[tid] = (g >= 0 && g < a.T) ? a.mask[g] : 0.f;
        }
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const int idx = tid + v * 256;
            *reinterpret_cast<u32x4*>(Xb + (idx / 12) * FF_XRS + (idx % 12) * 16) = xr[v];
        }
        // bias table (arena layout == LDS layout); a broadcast conditioning row (ldg == 0) is folded into
        // the in-layer biases.  All loads go out before the first store.
        constexpr int NTB = (FF_T_TOTAL + 255) / 256;
        float tv[NTB], gv[6];
#pragma unroll
        for (int i = 0; i < NTB; ++i) tv[i] = a.B[min(tid + i * 256, FF_T_TOTAL - 1)];
#pragma unroll
        for (int i = 0; i < 6; ++i) gv[i] = a.ldg == 0 ? a.gc[tid + i * 256] : 0.f;
#pragma unroll
        for (int i = 0; i < NTB; ++i)
            if (tid + i * 256 < FF_T_TOTAL) tb[tid + i * 256] = tv[i] + (i < 6 ? gv[i] : 0.f);
        if (tid < 4 * 25) {                                // 4 guard rows x 25 vectors of 16 B
            const int gr = tid / 25, c = tid % 25;
            const int row = gr < 2 ? gr : 64 + gr;         // h rows 0,1 and 66,67
            *reinterpret_cast<u32x4*>(Hb + row * FF_HRS + c * 16) = u32x4{0u, 0u, 0u, 0u};
        }
    }
    __syncthreads();
    stamp();

    // ---- phase 1: h = (pre(x0) + b) * mask      M = 192 (6 tiles), K = 96, N = 64
    {
        const int nmt = wid < 2 ? 2 : 1;                   // waves 0,1 take tiles w and w+4
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][n][q] = 0.f;
        u32x4 wa[2][FF_KSP];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < FF_KSP; ++ks)
                wa[i][ks] = __builtin_bit_cast(u32x4, Wl[(size_t)(FF_W_PRE + (min(wid + 4 * i, 5)) * FF_KSP + ks) * 64]);
#pragma unroll
        for (int ks = 0; ks < FF_KSP; ++ks) {
            u32x4 bf[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) bf[n] = *reinterpret_cast<const u32x4*>(Xb + (n * 32 + j) * FF_XRS + ks * 32 + hf * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int n = 0; n < 2; ++n) Mma<bf16_t>::run(acc[i][n], wa[i][ks], bf[n]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i < nmt) {
                const int ch = (wid + 4 * i) * 32 + 16 * hf;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int row = n * 32 + j;
                    const float m = mk[row];
                    float v[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) v[q] = (acc[i][n][q] + tb[FF_T_PRE + ch + q]) * m;
                    u32x4 oa, ob;
                    ff_pack16(v, oa, ob);
                    unsigned char* pp = Hb + (row + 2) * FF_HRS + ch * 2;
                    *reinterpret_cast<u32x4*>(pp) = oa;
                    *reinterpret_cast<u32x4*>(pp + 16) = ob;
                }
            }
        }
    }
    __syncthreads();
    stamp();

    // ---- phase 2: the four WN layers; skip sum lives in registers
    f32x16 skip[3];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int q = 0; q < 16; ++q) skip[u][q] = 0.f;
    const unsigned lbH = (unsigned)j * FF_HRS + hf * 16;     // lane's row / k-half inside a 32-row tile

    // ONE weight stream for the whole kernel: a 12-slot register ring (4 fragments per slot, 48 KiB per
    // wave in flight) that never drains -- the last 12 in-layer steps of a layer refill it with that
    // layer's res/skip fragments, and the 12 res/skip steps refill it with the next layer's first
    // in-layer steps.  (A CU alone pulls ~8-10 B/clk from the fabric in bursts; a cold 48-fragment burst
    // at the top of 2b cost 24k cycles per layer before this.)
    constexpr int NST = 5 * FF_KSH;                          // 60 (tap, k-step) in-layer steps
    constexpr int RING = FF_KSH;
    u32x4 ring[RING][4];
    const size_t oA = (size_t)mA * FF_KSH * 64, oAb = (size_t)(6 + mA) * FF_KSH * 64;
    const size_t oB = (size_t)mB * FF_KSH * 64, oBb = (size_t)(6 + mB) * FF_KSH * 64;
    // Every tile streams the same weights in the same order at the same time, so all co-resident blocks hit the
    // same L2 lines -- one channel -- in lockstep.  Each tile therefore walks the k-steps of a tap rotated by its own
    // offset (weights and B fragments alike; only the fp32 accumulation order changes), which spreads the
    // simultaneous requests over the channels (measured A/B in one build: flow + Generator 1.39-1.44 -> 1.33-1.37 ms; against
    // the build without the rotation arithmetic the net gain is ~2 %).
    // Rotating the taps as well, or the whole 60-step walk, measured slower: a run-time tap index puts a multiply and
    // selects into every step of a loop whose issue order is pinned.
    const int rot = a.rot_k ? tile % FF_KSH : 0;
    auto rk = [&](int ks) { const int k = ks + rot; return k >= FF_KSH ? k - FF_KSH : k; };
    auto fetch_in = [&](int l, int s, u32x4 (&dst)[4]) {
        const uint4* wi = Wl + (size_t)(FF_W_IN + l * FF_W_IN_L) * 64 + ((size_t)(s / FF_KSH) * 12 * FF_KSH + rk(s % FF_KSH)) * 64;
        dst[0] = __builtin_bit_cast(u32x4, wi[oA]);
        dst[1] = __builtin_bit_cast(u32x4, wi[oAb]);
        dst[2] = __builtin_bit_cast(u32x4, wi[oB]);
        dst[3] = __builtin_bit_cast(u32x4, wi[oBb]);
    };
    auto fetch_rs = [&](int l, int ks, u32x4 (&dst)[4]) {   // layer 3 has no res conv: its slots reload layer 0's (unused)
        const uint4* wr = Wl + (size_t)(FF_W_RES + (l < 3 ? l : 0) * 6 * FF_KSH + rk(ks)) * 64;
        const uint4* wsk = Wl + (size_t)(FF_W_SKIP + l * 6 * FF_KSH + rk(ks)) * 64;
        dst[0] = __builtin_bit_cast(u32x4, wr[oA]);
        dst[1] = __builtin_bit_cast(u32x4, wsk[oA]);
        dst[2] = __builtin_bit_cast(u32x4, wr[oB]);
        dst[3] = __builtin_bit_cast(u32x4, wsk[oB]);
    };
#pragma unroll
    for (int s = 0; s < RING; ++s) fetch_in(0, s, ring[s]);

    for (int l = 0; l < 4; ++l) {
        // 2a: x_in = in_layer(h) (k = 5, 192 -> 384), gate with the conditioning -> acts
        {
            f32x16 acc[6];   // [a_mA n0, a_mA n1, b_mA n0, b_mA n1, a_mB nB, b_mB nB]
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
            constexpr int BD = 2;                            // B fragments are read BD steps ahead
            u32x4 bfr[BD + 1][2];
            auto ldb = [&](int s, u32x4 (&dst)[2]) {
                const unsigned so = (unsigned)(s / FF_KSH) * FF_HRS + rk(s % FF_KSH) * 32;   // tap = row shift, k-step = 32 B
                dst[0] = *reinterpret_cast<const u32x4*>(Hb + lbH + so);
                dst[1] = *reinterpret_cast<const u32x4*>(Hb + lbH + so + 32 * FF_HRS);
            };
#pragma unroll
            for (int s = 0; s < BD; ++s) ldb(s, bfr[s]);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * BD, 0);
#pragma unroll
            for (int s = 0; s < NST; ++s) {
                if (s + BD < NST) ldb(s + BD, bfr[(s + BD) % (BD + 1)]);
                const u32x4 b0 = bfr[s % (BD + 1)][0], b1 = bfr[s % (BD + 1)][1];
                const u32x4 bB = nB ? b1 : b0;
                u32x4 (&w)[4] = ring[s % RING];
                Mma<bf16_t>::run(acc[0], w[0], b0);
                Mma<bf16_t>::run(acc[1], w[0], b1);
                Mma<bf16_t>::run(acc[2], w[1], b0);
                Mma<bf16_t>::run(acc[3], w[1], b1);
                Mma<bf16_t>::run(acc[4], w[2], bB);
                Mma<bf16_t>::run(acc[5], w[3], bB);
                if (s + RING < NST) fetch_in(l, s + RING, ring[s % RING]);
                else fetch_rs(l, s + RING - NST, ring[s % RING]);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            stamp();
            // gate: acts = tanh(a + ba + ga) * sigmoid(b + bb + gb)
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int m = u < 2 ? mA : mB;
                const int n = u < 2 ? u : nB;
                const f32x16& va = u < 2 ? acc[u] : acc[4];
                const f32x16& vb = u < 2 ? acc[2 + u] : acc[5];
                const int row = n * 32 + j;
                const int ch = m * 32 + 16 * hf;
                const int g = min(max(g0 + row, 0), a.T - 1);
                const float* bp = tb + FF_T_IN + l * 384 + ch;
                float ga[16], gb[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) { ga[q] = bp[q]; gb[q] = bp[192 + q]; }
                if (a.ldg != 0) {                          // per-frame conditioning
                    const float* gp = a.gc + (size_t)(a.seg ? a.seg[g] : g) * a.ldg + l * 384 + ch;
#pragma unroll
                    for (int q = 0; q < 16; ++q) { ga[q] += gp[q]; gb[q] += gp[192 + q]; }
                }
                float v[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float xa = va[q] + ga[q];
                    const float xb = vb[q] + gb[q];
                    v[q] = ff_tanh(xa) * ff_sigmoid(xb);
                }
                u32x4 oa, ob;
                ff_pack16(v, oa, ob);
                unsigned char* pp = Ab + row * FF_HRS + ch * 2;
                *reinterpret_cast<u32x4*>(pp) = oa;
                *reinterpret_cast<u32x4*>(pp + 16) = ob;
            }
        }
        __syncthreads();
        stamp();
        // 2b: res / skip 1x1 convs on acts; h = (h + res) * mask, skip accumulates in place
        {
            f32x16 res[3];
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int q = 0; q < 16; ++q) res[u][q] = 0.f;
            const bool has_res = l < 3;
            const int ln = l < 3 ? l + 1 : 3;                // after the last layer the refill is a harmless re-read
#pragma unroll
            for (int ks = 0; ks < FF_KSH; ++ks) {
                const u32x4 b0 = *reinterpret_cast<const u32x4*>(Ab + lbH + rk(ks) * 32);
                const u32x4 b1 = *reinterpret_cast<const u32x4*>(Ab + lbH + rk(ks) * 32 + 32 * FF_HRS);
                const u32x4 bB = nB ? b1 : b0;
                u32x4 (&w4)[4] = ring[ks];
                Mma<bf16_t>::run(res[0], w4[0], b0);
                Mma<bf16_t>::run(res[1], w4[0], b1);
                Mma<bf16_t>::run(skip[0], w4[1], b0);
                Mma<bf16_t>::run(skip[1], w4[1], b1);
                Mma<bf16_t>::run(res[2], w4[2], bB);
                Mma<bf16_t>::run(skip[2], w4[3], bB);
                fetch_in(ln, ks, ring[ks]);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
            }
            stamp();
            if (has_res) {
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int m = u < 2 ? mA : mB;
                    const int n = u < 2 ? u : nB;
                    const int row = n * 32 + j;
                    const int ch = m * 32 + 16 * hf;
                    unsigned char* pp = Hb + (row + 2) * FF_HRS + ch * 2;
                    float h[16];
                    ff_unpack16(*reinterpret_cast<const u32x4*>(pp), *reinterpret_cast<const u32x4*>(pp + 16), h);
                    const float mkr = mk[row];
                    const float* bp = tb + FF_T_RES + l * 192 + ch;
#pragma unroll
                    for (int q = 0; q < 16; ++q) h[q] = (h[q] + res[u][q] + bp[q]) * mkr;
                    u32x4 oa, ob;
                    ff_pack16(h, oa, ob);
                    *reinterpret_cast<u32x4*>(pp) = oa;
                    *reinterpret_cast<u32x4*>(pp + 16) = ob;
                }
            }
        }
        __syncthreads();
        stamp();
    }

    // ---- phase 3: out = (skip + sum of skip biases) * mask -> acts buffer; m = post(out) ; x1 update
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int m = u < 2 ? mA : mB;
        const int n = u < 2 ? u : nB;
        const int row = n * 32 + j;
        const int ch = m * 32 + 16 * hf;
        const float mkr = mk[row];
        const float* bp = tb + FF_T_SKIP + ch;
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = (skip[u][q] + bp[q]) * mkr;
        u32x4 oa, ob;
        ff_pack16(v, oa, ob);
        unsigned char* pp = Ab + row * FF_HRS + ch * 2;
        *reinterpret_cast<u32x4*>(pp) = oa;
        *reinterpret_cast<u32x4*>(pp + 16) = ob;
    }
    __syncthreads();
    {
        // 6 (tile, row-tile) units over 4 waves: unit u -> (mt = u % 3, n = u / 3); wave w takes u = w and w + 4
        const int nun = wid < 2 ? 2 : 1;
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
        u32x4 wp[2][FF_KSH];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int u = min(wid + 4 * i, 5);
#pragma unroll
            for (int ks = 0; ks < FF_KSH; ++ks)
                wp[i][ks] = __builtin_bit_cast(u32x4, Wl[(size_t)(FF_W_POST + (u % 3) * FF_KSH + ks) * 64]);
        }
#pragma unroll
        for (int ks = 0; ks < FF_KSH; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int u = min(wid + 4 * i, 5);
                const u32x4 bf = *reinterpret_cast<const u32x4*>(Ab + lbH + ks * 32 + (u / 3) * 32 * FF_HRS);
                Mma<bf16_t>::run(acc[i], wp[i][ks], bf);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i < nun) {
                const int u = wid + 4 * i;
                const int row = (u / 3) * 32 + j;
                const int g = g0 + row;
                const int ch = (u % 3) * 32 + 16 * hf;
                if (row >= FF_HALO && row < FF_HALO + FF_VR && g < a.T) {
                    bf16_t* xp = a.P + (size_t)g * FF_H + a.xup_off + ch;
                    float x1[16];
                    ff_unpack16(*reinterpret_cast<const u32x4*>(xp), *reinterpret_cast<const u32x4*>(xp + 8), x1);
                    const float mkr = mk[row];
                    const float* bp = tb + FF_T_POST + ch;
                    // post is packed negated: x1 - m = x1 + (-(W out + b))
#pragma unroll
                    for (int q = 0; q < 16; ++q) x1[q] = (x1[q] + (acc[i][q] + bp[q]) * mkr) * mkr;
                    u32x4 oa, ob;
                    ff_pack16(x1, oa, ob);
                    *reinterpret_cast<u32x4*>(xp) = oa;
                    *reinterpret_cast<u32x4*>(xp + 8) = ob;
                }
            }
        }
    }
    stamp();
}

}  // namespace gsv
