// flowstage: one ResidualCouplingLayer of the SoVITS flow in reverse mode (reference SoVITS/module/modules.py:482-501, WN
// modules.py:80-104) as TEN short launches that each use many CUs -- the form for FEW frames.
//
// flowfuse.h runs a coupling layer as one kernel in which a block owns 48 frames for the whole layer and streams the layer's
// 3.5 MB of weights through its registers: ~57 us per layer whatever the frame count, because one CU pulls ~25 B/clk from L2 and
// 10 s of audio are 11 blocks (245 CUs idle).  The layer is a chain of GEMMs with an all-channel dependency between them, so the
// way to put more CUs on few frames is to cut it at those dependencies and give every launch a (frame tile, channel tile) grid:
//     pre (96 -> 192)  |  4 x { in_layer k = 5 (192 -> 384) + gate ;  res / skip 1x1 (192 -> 384) }  |  post (192 -> 96) + update
// A block then pulls 12-120 KB of weights instead of 3.5 MB.  Same operands, rounding points and fragment-packed weight / bias
// arenas as flowfuse.h (FF_W_*, FF_T_*): bf16 activations (h, acts, out), fp32 accumulation, the skip sum carried in fp32 (here
// through a [T][192] fp32 tensor instead of registers; each layer's skip GEMM starts from the running sum, as the fused kernel's
// accumulator does).  The in_layer's 60 (tap, k-step) steps are split over the block's four waves and met in LDS in wave order.
// Frames are independent except for the k = 5 halo (+-2 frames of h), which a block reads from its neighbours' rows.
#pragma once
#include "flowfuse.h"

namespace gsv {

enum { FS_PRE = 0, FS_IN = 1, FS_RS = 2, FS_POST = 3, FS_ROWS = 32 };

struct FlowStageArgs {
    bf16_t* P;            // [T][192] the flow's activations; POST rewrites the updated half in place
    const float* mask;    // [T]
    const float* gc;      // this flow's conditioning [1 or T][ldg], WN layer l at + l * 384
    int ldg;              // 0 = one broadcast row
    const int* seg;       // null, or: frame g takes conditioning row seg[g]
    const uint4* W;       // flowfuse.h weight arena
    const float* B;       // flowfuse.h bias arena
    int T, xin_off, xup_off;
    bf16_t* h;            // [T][192]
    bf16_t* acts;         // [T][192]
    float* skip;          // [T][192] running skip sum
    bf16_t* outp;         // [T][192] (skip sum + biases) * mask
    int l;                // WN layer (IN, RS)
    int rpb;              // frame tiles of 32 a block walks (its weights stay in registers)
};

// 32 (+ halo) rows x NV 16-byte vectors: global [T][ld] (+ channel offset) -> LDS rows `stride` bytes apart; rows outside [0, T) are zero
template <int NV, int ROWS>
__device__ __forceinline__ void fs_stage(const bf16_t* __restrict__ src, int ld, int g_first, int T, unsigned char* __restrict__ dst, int stride) {
    constexpr int TOT = ROWS * NV, PASS = (TOT + 255) / 256;
    u32x4 v[PASS];
#pragma unroll
    for (int p = 0; p < PASS; ++p) {
        const int idx = threadIdx.x + p * 256, r = idx / NV, c = idx % NV, g = g_first + r;
        const bool ok = idx < TOT && g >= 0 && g < T;
        v[p] = *reinterpret_cast<const u32x4*>(src + (size_t)(ok ? g : 0) * ld + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[p][e] = ok ? v[p][e] : 0u;
    }
#pragma unroll
    for (int p = 0; p < PASS; ++p) {
        const int idx = threadIdx.x + p * 256;
        if (idx < TOT) *reinterpret_cast<u32x4*>(dst + (idx / NV) * stride + (idx % NV) * 16) = v[p];
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void flowstage_kernel(FlowStageArgs a) {
    constexpr int XROWS = MODE == FS_IN ? FS_ROWS + 4 : FS_ROWS;
    constexpr int STRIDE = MODE == FS_PRE ? FF_XRS : FF_HRS;
    __shared__ __attribute__((aligned(16))) unsigned char xs[XROWS * STRIDE];
    __shared__ __attribute__((aligned(16))) float red[MODE == FS_IN ? 4 * 2 * 16 * 64 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, j = lane & 31, hf = lane >> 5;
    const int ntiles = (a.T + FS_ROWS - 1) / FS_ROWS;
    const int t0 = blockIdx.x * a.rpb, t1 = min(t0 + a.rpb, ntiles);
    const uint4* Wl = a.W + lane;

    // ---- this wave's weight fragments, in flight before anything else
    constexpr int NWF = MODE == FS_PRE ? FF_KSP : (MODE == FS_IN ? 30 : FF_KSH);
    u32x4 wr[NWF];
    int mt = 0;            // PRE / POST: m-tile; RS: tile 0..11 (res 0-5, skip 6-11; layer 3: skip only)
    bool live = true;
    if constexpr (MODE == FS_PRE) {
        mt = blockIdx.y * 3 + wid;
        live = wid < 3;
#pragma unroll
        for (int ks = 0; ks < FF_KSP; ++ks) wr[ks] = __builtin_bit_cast(u32x4, Wl[(size_t)(FF_W_PRE + (live ? mt : 0) * FF_KSP + ks) * 64]);
    } else if constexpr (MODE == FS_IN) {
        const int at = blockIdx.y;      // acts channel tile: a rows = m-tile at, b rows = m-tile 6 + at
#pragma unroll
        for (int i = 0; i < 15; ++i) {
            const int s = wid * 15 + i, tap = s / FF_KSH, ks = s % FF_KSH;
            const size_t f = (size_t)FF_W_IN + (size_t)a.l * FF_W_IN_L + (size_t)tap * 12 * FF_KSH + ks;
            wr[2 * i] = __builtin_bit_cast(u32x4, Wl[(f + (size_t)at * FF_KSH) * 64]);
            wr[2 * i + 1] = __builtin_bit_cast(u32x4, Wl[(f + (size_t)(6 + at) * FF_KSH) * 64]);
        }
    } else if constexpr (MODE == FS_RS) {
        const int nt = a.l < 3 ? 12 : 6;
        mt = blockIdx.y * 4 + wid;
        live = mt < nt;
        const int m = live ? mt : 0;
        const bool is_skip = a.l == 3 || m >= 6;
        const size_t f = is_skip ? (size_t)FF_W_SKIP + (size_t)(a.l * 6 + (a.l == 3 ? m : m - 6)) * FF_KSH : (size_t)FF_W_RES + (size_t)(a.l * 6 + m) * FF_KSH;
#pragma unroll
        for (int ks = 0; ks < FF_KSH; ++ks) wr[ks] = __builtin_bit_cast(u32x4, Wl[(f + ks) * 64]);
    } else {
        mt = wid;
        live = wid < 3;
#pragma unroll
        for (int ks = 0; ks < FF_KSH; ++ks) wr[ks] = __builtin_bit_cast(u32x4, Wl[(size_t)(FF_W_POST + (live ? mt : 0) * FF_KSH + ks) * 64]);
    }

    for (int t = t0; t < t1; ++t) {
        const int g0 = t * FS_ROWS;                    // first frame of the tile
        const int g = g0 + j;                          // this lane's frame
        const bool rok = g < a.T;
        const float mk = a.mask[rok ? g : 0];
        // ---- the tile's input rows -> LDS
        if constexpr (MODE == FS_PRE) fs_stage<12, XROWS>(a.P + a.xin_off, FF_H, g0, a.T, xs, STRIDE);
        else if constexpr (MODE == FS_IN) fs_stage<24, XROWS>(a.h, FF_H, g0 - 2, a.T, xs, STRIDE);
        else if constexpr (MODE == FS_RS) fs_stage<24, XROWS>(a.acts, FF_H, g0, a.T, xs, STRIDE);
        else fs_stage<24, XROWS>(a.outp, FF_H, g0, a.T, xs, STRIDE);
        __syncthreads();
        const unsigned char* bp = xs + j * STRIDE + hf * 16;

        if constexpr (MODE == FS_PRE) {
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
            for (int ks = 0; ks < FF_KSP; ++ks) Mma<bf16_t>::run(acc, wr[ks], *reinterpret_cast<const u32x4*>(bp + ks * 32));
            if (live && rok) {
                const int ch = mt * 32 + 16 * hf;
                float v[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = (acc[q] + a.B[FF_T_PRE + ch + q]) * mk;
                u32x4 oa, ob;
                ff_pack16(v, oa, ob);
                bf16_t* hp = a.h + (size_t)g * FF_H + ch;
                *reinterpret_cast<u32x4*>(hp) = oa;
                *reinterpret_cast<u32x4*>(hp + 8) = ob;
            }
        } else if constexpr (MODE == FS_IN) {
            const int at = blockIdx.y;
            f32x16 aa, ab;
#pragma unroll
            for (int q = 0; q < 16; ++q) { aa[q] = 0.f; ab[q] = 0.f; }
#pragma unroll
            for (int i = 0; i < 15; ++i) {
                const int s = wid * 15 + i;                                   // run-time tap / k-step: address arithmetic only
                const u32x4 b = *reinterpret_cast<const u32x4*>(bp + (s / FF_KSH) * STRIDE + (s % FF_KSH) * 32);
                Mma<bf16_t>::run(aa, wr[2 * i], b);
                Mma<bf16_t>::run(ab, wr[2 * i + 1], b);
            }
            // the four K-quarters meet in LDS; wave w then finishes accumulator registers 4 w .. 4 w + 3 (4 consecutive channels)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                red[((wid * 2 + 0) * 16 + q) * 64 + lane] = aa[q];
                red[((wid * 2 + 1) * 16 + q) * 64 + lane] = ab[q];
            }
            __syncthreads();
            if (rok) {
                const int ch = at * 32 + 16 * hf + 4 * wid;
                const float* bi = a.B + FF_T_IN + a.l * 384 + ch;
                const float* gp = a.gc + (a.ldg != 0 ? (size_t)(a.seg ? a.seg[g] : g) * a.ldg : 0) + a.l * 384 + ch;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int q = 4 * wid + e;
                    float xa = 0.f, xb = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {                             // wave order: bit-reproducible
                        xa += red[((w * 2 + 0) * 16 + q) * 64 + lane];
                        xb += red[((w * 2 + 1) * 16 + q) * 64 + lane];
                    }
                    xa += bi[e] + gp[e];
                    xb += bi[192 + e] + gp[192 + e];
                    v[e] = ff_tanh(xa) * ff_sigmoid(xb);
                }
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2*>(a.acts + (size_t)g * FF_H + ch) = o;
            }
        } else if constexpr (MODE == FS_RS) {
            const bool is_skip = a.l == 3 || mt >= 6;
            const int m = a.l == 3 ? mt : (mt >= 6 ? mt - 6 : mt);
            const int ch = m * 32 + 16 * hf;
            f32x16 acc;
            if (live && is_skip && a.l > 0 && rok) {           // the running skip sum is the accumulator's starting value
                const float* sp = a.skip + (size_t)g * FF_H + ch;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 s4 = *reinterpret_cast<const f32x4*>(sp + 4 * q4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[4 * q4 + e] = s4[e];
                }
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] = 0.f;
            }
            u32x4 hraw0 = {0u, 0u, 0u, 0u}, hraw1 = {0u, 0u, 0u, 0u};
            bf16_t* hp = a.h + (size_t)(rok ? g : 0) * FF_H + ch;
            if (live && !is_skip) { hraw0 = *reinterpret_cast<const u32x4*>(hp); hraw1 = *reinterpret_cast<const u32x4*>(hp + 8); }
#pragma unroll
            for (int ks = 0; ks < FF_KSH; ++ks) Mma<bf16_t>::run(acc, wr[ks], *reinterpret_cast<const u32x4*>(bp + ks * 32));
            if (live && rok) {
                if (!is_skip) {
                    float hv[16];
                    ff_unpack16(hraw0, hraw1, hv);
                    const float* bi = a.B + FF_T_RES + a.l * 192 + ch;
#pragma unroll
                    for (int q = 0; q < 16; ++q) hv[q] = (hv[q] + acc[q] + bi[q]) * mk;
                    u32x4 oa, ob;
                    ff_pack16(hv, oa, ob);
                    *reinterpret_cast<u32x4*>(hp) = oa;
                    *reinterpret_cast<u32x4*>(hp + 8) = ob;
                } else if (a.l < 3) {
                    float* sp = a.skip + (size_t)g * FF_H + ch;
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) *reinterpret_cast<f32x4*>(sp + 4 * q4) = f32x4{acc[4 * q4], acc[4 * q4 + 1], acc[4 * q4 + 2], acc[4 * q4 + 3]};
                } else {
                    float v[16];
                    const float* bi = a.B + FF_T_SKIP + ch;
#pragma unroll
                    for (int q = 0; q < 16; ++q) v[q] = (acc[q] + bi[q]) * mk;
                    u32x4 oa, ob;
                    ff_pack16(v, oa, ob);
                    bf16_t* op = a.outp + (size_t)g * FF_H + ch;
                    *reinterpret_cast<u32x4*>(op) = oa;
                    *reinterpret_cast<u32x4*>(op + 8) = ob;
                }
            }
        } else {
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
            const int ch = mt * 32 + 16 * hf;
            bf16_t* xp = a.P + (size_t)(rok ? g : 0) * FF_H + a.xup_off + (live ? ch : 0);
            const u32x4 x0 = *reinterpret_cast<const u32x4*>(xp), x1r = *reinterpret_cast<const u32x4*>(xp + 8);
#pragma unroll
            for (int ks = 0; ks < FF_KSH; ++ks) Mma<bf16_t>::run(acc, wr[ks], *reinterpret_cast<const u32x4*>(bp + ks * 32));
            if (live && rok) {
                float x1[16];
                ff_unpack16(x0, x1r, x1);
                const float* bi = a.B + FF_T_POST + ch;
                // post is packed negated: x1 - m = x1 + (-(W out + b))
#pragma unroll
                for (int q = 0; q < 16; ++q) x1[q] = (x1[q] + (acc[q] + bi[q]) * mk) * mk;
                u32x4 oa, ob;
                ff_pack16(x1, oa, ob);
                *reinterpret_cast<u32x4*>(xp) = oa;
                *reinterpret_cast<u32x4*>(xp + 8) = ob;
            }
        }
        if (t + 1 < t1) __syncthreads();               // everyone is done with this tile's rows (and partial sums)
    }
}


// ---- merged form: FIVE launches per coupling layer ------------------------------------------------------------------------------
// Every launch above costs ~3.8 us whatever it computes (12 or 120 KB of weights per block: the time is the dependent chain kernel
// entry -> operands land -> LDS -> MFMA -> store, not the bytes), so the launches whose work is small are folded into their
// consumers by RECOMPUTATION: an in_layer block owns 28 frames and needs h of 32 (k = 5 halo); instead of reading h_l it reads
// h_{l-1} and acts_{l-1} of those 32 frames and redoes the res 1x1 conv for them (72 MFMAs per block, six times redundant across
// the channel tiles -- nothing next to a launch); channel tile `at` also owns skip tile `at` of layer l - 1, and tile 0 writes h_l
// back for the next launch.  The layer's tail -- skip conv of WN layer 3, post, the x1 update -- and the NEXT coupling layer's pre
// are one kernel per 32 frames.  Per coupling layer: in0, (rs0 + in1), (rs1 + in2), (rs2 + in3), (skip3 + post + next pre).
enum { FM_VR = 28 };

struct FlowMergeArgs {
    FlowStageArgs s;      // this coupling layer (s.l = WN layer of the in_layer being computed); s.h / s.acts are not used here:
    // h and acts PING-PONG between two buffers each -- within one launch some blocks still read h_{l-1} / acts_{l-1} of frames whose
    // h_l / acts_l other blocks already write
    const bf16_t* h_in;   // h_{l-1} (in kernel), -
    bf16_t* h_out;        // h_l, written by channel tile 0 (in kernel, l > 0); the next coupling layer's h (tail kernel)
    const bf16_t* acts_in;   // acts_{l-1} (in kernel, l > 0); acts_3 (tail kernel)
    bf16_t* acts_out;     // acts_l (in kernel)
    const uint4* Wn;      // tail kernel: the NEXT coupling layer's arenas (its pre), or null after the last one
    const float* Bn;
};

static __global__ __launch_bounds__(256) void flowmerge_in_kernel(FlowMergeArgs m) {
    const FlowStageArgs& a = m.s;
    __shared__ __attribute__((aligned(16))) unsigned char hs[36 * FF_HRS];     // h rows: tile row r (frame g0 - 2 + r) at LDS row r + 2; 2 guard rows each side
    __shared__ __attribute__((aligned(16))) unsigned char as[32 * FF_HRS];     // acts_{l-1} of the same 32 frames
    __shared__ __attribute__((aligned(16))) float red[4 * 2 * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, j = lane & 31, hf = lane >> 5;
    const int at = blockIdx.y, l = a.l;
    const int g0 = blockIdx.x * FM_VR;
    const int g = g0 - 2 + j;                          // this lane's frame (tile row j)
    const bool gin = g >= 0 && g < a.T;
    const bool valid = j >= 2 && j < 2 + FM_VR && gin; // frames this block owns
    const uint4* Wl = a.W + lane;

    // ---- weights: in_layer (15 of the 60 (tap, k-step) steps, a and b rows), res m-tiles wid and wid + 4, skip tile `at` (wave 3)
    u32x4 wi[30], wres[2][FF_KSH], wsk[FF_KSH];
#pragma unroll
    for (int i = 0; i < 15; ++i) {
        const int s = wid * 15 + i, tap = s / FF_KSH, ks = s % FF_KSH;
        const size_t f = (size_t)FF_W_IN + (size_t)l * FF_W_IN_L + (size_t)tap * 12 * FF_KSH + ks;
        wi[2 * i] = __builtin_bit_cast(u32x4, Wl[(f + (size_t)at * FF_KSH) * 64]);
        wi[2 * i + 1] = __builtin_bit_cast(u32x4, Wl[(f + (size_t)(6 + at) * FF_KSH) * 64]);
    }
    const int nres = wid < 2 ? 2 : 1;
    if (l > 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < FF_KSH; ++ks)
                wres[i][ks] = __builtin_bit_cast(u32x4, Wl[(size_t)(FF_W_RES + ((l - 1) * 6 + min(wid + 4 * i, 5)) * FF_KSH + ks) * 64]);
#pragma unroll
        for (int ks = 0; ks < FF_KSH; ++ks) wsk[ks] = __builtin_bit_cast(u32x4, Wl[(size_t)(FF_W_SKIP + ((l - 1) * 6 + at) * FF_KSH + ks) * 64]);
    }
    const float mk = a.mask[gin ? g : 0] * (gin ? 1.f : 0.f);

    // ---- stage h_{l-1} (l = 0: h from pre) and acts_{l-1}; clear the guard rows
    fs_stage<24, 32>(m.h_in, FF_H, g0 - 2, a.T, hs + 2 * FF_HRS, FF_HRS);
    if (l > 0) fs_stage<24, 32>(m.acts_in, FF_H, g0 - 2, a.T, as, FF_HRS);
    if (tid < 4 * 25) {
        const int gr = tid / 25, c = tid % 25;
        *reinterpret_cast<u32x4*>(hs + (gr < 2 ? gr : 32 + gr) * FF_HRS + c * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();

    if (l > 0) {
        // ---- h_l = (h_{l-1} + res_{l-1}(acts_{l-1}) + b) * mask for the tile's 32 frames, in place; skip tile `at` of layer l - 1
        const unsigned char* ap = as + j * FF_HRS + hf * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i < nres) {
                const int mt = wid + 4 * i;
                f32x16 acc;
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
                for (int ks = 0; ks < FF_KSH; ++ks) Mma<bf16_t>::run(acc, wres[i][ks], *reinterpret_cast<const u32x4*>(ap + ks * 32));
                const int ch = mt * 32 + 16 * hf;
                unsigned char* hp = hs + (j + 2) * FF_HRS + ch * 2;
                float hv[16];
                ff_unpack16(*reinterpret_cast<const u32x4*>(hp), *reinterpret_cast<const u32x4*>(hp + 16), hv);
                const float* bi = a.B + FF_T_RES + (l - 1) * 192 + ch;
#pragma unroll
                for (int q = 0; q < 16; ++q) hv[q] = (hv[q] + acc[q] + bi[q]) * mk;
                u32x4 oa, ob;
                ff_pack16(hv, oa, ob);
                *reinterpret_cast<u32x4*>(hp) = oa;
                *reinterpret_cast<u32x4*>(hp + 16) = ob;
                if (at == 0 && valid && l < 3) {          // h_3 has no reader
                    bf16_t* gp = m.h_out + (size_t)g * FF_H + ch;
                    *reinterpret_cast<u32x4*>(gp) = oa;
                    *reinterpret_cast<u32x4*>(gp + 8) = ob;
                }
            }
        }
        if (wid == 3) {
            const int ch = at * 32 + 16 * hf;
            float* sp = a.skip + (size_t)(gin ? g : 0) * FF_H + ch;
            f32x16 acc;
            if (l > 1 && valid) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 s4 = *reinterpret_cast<const f32x4*>(sp + 4 * q4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[4 * q4 + e] = s4[e];
                }
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] = 0.f;
            }
#pragma unroll
            for (int ks = 0; ks < FF_KSH; ++ks) Mma<bf16_t>::run(acc, wsk[ks], *reinterpret_cast<const u32x4*>(ap + ks * 32));
            if (valid) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) *reinterpret_cast<f32x4*>(sp + 4 * q4) = f32x4{acc[4 * q4], acc[4 * q4 + 1], acc[4 * q4 + 2], acc[4 * q4 + 3]};
            }
        }
        __syncthreads();
    }

    // ---- in_layer l on the tile: tile row j reads LDS rows j .. j + 4 (its frames - 2 .. + 2)
    const unsigned char* bp = hs + j * FF_HRS + hf * 16;
    f32x16 aa, ab;
#pragma unroll
    for (int q = 0; q < 16; ++q) { aa[q] = 0.f; ab[q] = 0.f; }
#pragma unroll
    for (int i = 0; i < 15; ++i) {
        const int s = wid * 15 + i;
        const u32x4 b = *reinterpret_cast<const u32x4*>(bp + (s / FF_KSH) * FF_HRS + (s % FF_KSH) * 32);
        Mma<bf16_t>::run(aa, wi[2 * i], b);
        Mma<bf16_t>::run(ab, wi[2 * i + 1], b);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        red[((wid * 2 + 0) * 16 + q) * 64 + lane] = aa[q];
        red[((wid * 2 + 1) * 16 + q) * 64 + lane] = ab[q];
    }
    __syncthreads();
    if (valid) {
        const int ch = at * 32 + 16 * hf + 4 * wid;
        const float* bi = a.B + FF_T_IN + l * 384 + ch;
        const float* gp = a.gc + (a.ldg != 0 ? (size_t)(a.seg ? a.seg[g] : g) * a.ldg : 0) + l * 384 + ch;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int q = 4 * wid + e;
            float xa = 0.f, xb = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                xa += red[((w * 2 + 0) * 16 + q) * 64 + lane];
                xb += red[((w * 2 + 1) * 16 + q) * 64 + lane];
            }
            xa += bi[e] + gp[e];
            xb += bi[192 + e] + gp[192 + e];
            v[e] = ff_tanh(xa) * ff_sigmoid(xb);
        }
        uint2 o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(m.acts_out + (size_t)g * FF_H + ch) = o;
    }
}

// the coupling layer's tail on 32 frames: skip conv of WN layer 3 onto the running sum -> out -> post -> x1 update -> the NEXT
// coupling layer's pre (its h); every GEMM's input tile is the previous one's output in LDS
static __global__ __launch_bounds__(256) void flowmerge_tail_kernel(FlowMergeArgs m) {
    const FlowStageArgs& a = m.s;
    __shared__ __attribute__((aligned(16))) unsigned char as[32 * FF_HRS];     // acts_3, then out
    __shared__ __attribute__((aligned(16))) unsigned char os[32 * FF_HRS];
    __shared__ __attribute__((aligned(16))) unsigned char xs[32 * FF_XRS];     // the updated half
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, j = lane & 31, hf = lane >> 5;
    const int g0 = blockIdx.x * FS_ROWS, g = g0 + j;
    const bool rok = g < a.T;
    const uint4* Wl = a.W + lane;
    const bool has_next = m.Wn != nullptr;
    const int n2 = wid < 2 ? 2 : 1;                    // m-tiles wid and wid + 4 of a 6-tile GEMM
    u32x4 wsk[2][FF_KSH], wpo[FF_KSH], wpr[2][FF_KSP];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < FF_KSH; ++ks) wsk[i][ks] = __builtin_bit_cast(u32x4, Wl[(size_t)(FF_W_SKIP + (3 * 6 + min(wid + 4 * i, 5)) * FF_KSH + ks) * 64]);
#pragma unroll
    for (int ks = 0; ks < FF_KSH; ++ks) wpo[ks] = __builtin_bit_cast(u32x4, Wl[(size_t)(FF_W_POST + min(wid, 2) * FF_KSH + ks) * 64]);
    if (has_next) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < FF_KSP; ++ks) wpr[i][ks] = __builtin_bit_cast(u32x4, (m.Wn + lane)[(size_t)(FF_W_PRE + min(wid + 4 * i, 5) * FF_KSP + ks) * 64]);
    }
    const float mk = a.mask[rok ? g : 0] * (rok ? 1.f : 0.f);
    fs_stage<24, 32>(m.acts_in, FF_H, g0, a.T, as, FF_HRS);
    __syncthreads();
    // ---- out = (skip sum + skip_3(acts_3) + biases) * mask
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i < n2) {
            const int ch = (wid + 4 * i) * 32 + 16 * hf;
            const float* sp = a.skip + (size_t)(rok ? g : 0) * FF_H + ch;
            f32x16 acc;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(sp + 4 * q4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[4 * q4 + e] = s4[e];
            }
            const unsigned char* ap = as + j * FF_HRS + hf * 16;
#pragma unroll
            for (int ks = 0; ks < FF_KSH; ++ks) Mma<bf16_t>::run(acc, wsk[i][ks], *reinterpret_cast<const u32x4*>(ap + ks * 32));
            float v[16];
            const float* bi = a.B + FF_T_SKIP + ch;
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = (acc[q] + bi[q]) * mk;
            u32x4 oa, ob;
            ff_pack16(v, oa, ob);
            unsigned char* op = os + j * FF_HRS + ch * 2;
            *reinterpret_cast<u32x4*>(op) = oa;
            *reinterpret_cast<u32x4*>(op + 16) = ob;
        }
    }
    __syncthreads();
    // ---- x1 <- (x1 + (post(out) + b) * mask) * mask     (post is packed negated)
    if (wid < 3) {
        const int ch = wid * 32 + 16 * hf;
        bf16_t* xp = a.P + (size_t)(rok ? g : 0) * FF_H + a.xup_off + ch;
        const u32x4 x0 = *reinterpret_cast<const u32x4*>(xp), x1r = *reinterpret_cast<const u32x4*>(xp + 8);
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        const unsigned char* op = os + j * FF_HRS + hf * 16;
#pragma unroll
        for (int ks = 0; ks < FF_KSH; ++ks) Mma<bf16_t>::run(acc, wpo[ks], *reinterpret_cast<const u32x4*>(op + ks * 32));
        float x1[16];
        ff_unpack16(x0, x1r, x1);
        const float* bi = a.B + FF_T_POST + ch;
#pragma unroll
        for (int q = 0; q < 16; ++q) x1[q] = (x1[q] + (acc[q] + bi[q]) * mk) * mk;
        u32x4 oa, ob;
        ff_pack16(x1, oa, ob);
        if (rok) {
            *reinterpret_cast<u32x4*>(xp) = oa;
            *reinterpret_cast<u32x4*>(xp + 8) = ob;
        }
        unsigned char* xq = xs + j * FF_XRS + ch * 2;
        *reinterpret_cast<u32x4*>(xq) = oa;
        *reinterpret_cast<u32x4*>(xq + 16) = ob;
    }
    if (!has_next) return;
    __syncthreads();
    // ---- the next coupling layer's h = (pre(x0') + b) * mask: its conv-input half is the half just updated
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i < n2) {
            const int ch = (wid + 4 * i) * 32 + 16 * hf;
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
            const unsigned char* xq = xs + j * FF_XRS + hf * 16;
#pragma unroll
            for (int ks = 0; ks < FF_KSP; ++ks) Mma<bf16_t>::run(acc, wpr[i][ks], *reinterpret_cast<const u32x4*>(xq + ks * 32));
            if (rok) {
                float v[16];
                const float* bi = m.Bn + FF_T_PRE + ch;
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = (acc[q] + bi[q]) * mk;
                u32x4 oa, ob;
                ff_pack16(v, oa, ob);
                bf16_t* hp = m.h_out + (size_t)g * FF_H + ch;
                *reinterpret_cast<u32x4*>(hp) = oa;
                *reinterpret_cast<u32x4*>(hp + 8) = ob;
            }
        }
    }
}

}  // namespace gsv
