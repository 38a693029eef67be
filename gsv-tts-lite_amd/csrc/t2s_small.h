// GEMMs of the batched decode step at FEW rows (17 .. kSmallMaxM sequences, bf16): the same five launches per layer as
// t2s_batch.h, with the four linears re-shaped for launch latency instead of bytes.
//
// Reference: the linears of T2SBlock.decode_next_token, gsv_tts/GPT_SoVITS/GPT/t2s_model.py:80-85 (qkv), :97 (out_proj),
// :100-103 (mlp) on B rows, the LayerNorms of :98 / :104 in the prologue of the GEMM that consumes them.
//
// What the 32 x 32 tiles of t2s_batch.h cost at 64 rows (profiles/r03_chain_launches.txt: 5.0-8.1 us per GEMM launch,
// 37 us per layer): a block read its fp32 X rows straight in B-fragment order -- every lane 16 bytes of its own row, 32-64
// cache lines per load instruction on the texture-address path -- normalised them in that layout, met its four K-split waves
// in LDS and left the epilogue to one wave.  Here:
//   * tile = 16 rows (sequences) x 16 output channels on v_mfma_f32_16x16x32_bf16, the WEIGHTS as the A operand (a lane's four
//     accumulators are four consecutive channels of one row: 16-byte stores); a block is 2 or 4 waves and every wave owns ONE
//     channel tile over the whole K = 512: 16 weight loads (1 KiB each, fragment order, packed at load) in flight at entry,
//     no K-split, no cross-wave reduction, every wave writes its own tile;
//   * the block's 16 X rows are loaded COALESCED (a wave reads whole rows: 1 KiB per instruction), a row's LayerNorm statistics
//     are wave-local (cross-lane network, no barrier), the normalised rows go to LDS as bf16 once and every wave reads its
//     B fragments from there (ds_read_b128, rows 1040 bytes apart: conflict-free): ONE barrier per block;
//   * W2 (K = 2048, X = the bf16 hidden rows): four waves split K, X fragments straight from global (64 contiguous bytes per
//     row per instruction), one LDS meeting.
// 96-256 blocks of 48-128 KB each instead of 32-128 blocks of 96-256 KB.
#pragma once
#include "t2s_batch.h"

namespace gsv {

constexpr int kSmallMaxM = 256;    // rows up to which the chain runs on these kernels (measured faster up to 256 sequences: 0.91 vs 1.05 ms
                                   // at 128, 1.25 vs 1.51 at 256; the prompt pass, thousands of rows, keeps t2s_batch.h's 32 x 32 tiles)

typedef __bf16 bf16x8_s __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mma16(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_s, a), __builtin_bit_cast(bf16x8_s, b), c, 0, 0, 0);
}

// weight fragments of the 16 x 16 x 32 MFMA: dst[ntile][kstep][lane][8] = W[ntile*16 + (lane & 15)][kstep*32 + (lane >> 4)*8 + e]
static __global__ __launch_bounds__(256) void pack16_kernel(const float* __restrict__ W, bf16_t* __restrict__ dst, int N, int K) {
    const size_t total = (size_t)N * K;
    const int ksn = K / 32;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        size_t r = idx;
        const int e = r % 8; r /= 8;
        const int lane = r % 64; r /= 64;
        const int ks = r % ksn; r /= ksn;
        const int nt = (int)r;
        dst[idx] = f32_to_bf16(W[(size_t)(nt * 16 + (lane & 15)) * K + ks * 32 + (lane >> 4) * 8 + e]);
    }
}

// fp8 (GSV_FP8 handles): e4m3 weights with one scale per output channel, activations e4m3 at unit scale (t2s_batch.h).  k-steps
// are PAIRED so that one 16-byte load feeds two MFMAs: dst[ntile][pair][lane][16] = W[n][pair*64 + (lane >> 4)*16 + e] / scale[n],
// bytes 0-7 the first MFMA of the pair, 8-15 the second (a contraction may enumerate its index in any order both operands share)
__device__ __forceinline__ f32x4 mma16_f8(uint64_t a, uint64_t b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)a, (long)b, c, 0, 0, 0);
}
static __global__ __launch_bounds__(256) void pack16_f8_kernel(const float* __restrict__ W, const float* __restrict__ scale, uint32_t* __restrict__ dst, int N, int K) {
    const size_t total = (size_t)N * K / 4;                 // dwords
    const int npair = K / 64;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        size_t r = idx;
        const int d = r % 4; r /= 4;
        const int lane = r % 64; r /= 64;
        const int p = r % npair; r /= npair;
        const int n = (int)r * 16 + (lane & 15);
        const float inv = 1.0f / scale[n];
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = clamp_e4m3(W[(size_t)n * K + p * 64 + (lane >> 4) * 16 + d * 4 + i] * inv);
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
        dst[idx] = (uint32_t)w;
    }
}

struct SGemmArgs {
    const void* X;        // sgemm: fp32 [M][512]; sgemm_k: bf16 [M][2048]
    int M;
    const float* lng;     // PRO_LN: LayerNorm weight / bias [512]
    const float* lnb;
    float* xout;          // PRO_LN: the normalised rows [M][512] fp32 (the layer's residual later), written by channel group 0; or null
    const uint4* W;       // pack16_kernel's order (bf16) / pack16_f8_kernel's (e4m3)
    const float* wscale;  // fp8: dequantisation scale per output channel
    const float* bias;    // [N]
    const float* res;     // residual rows fp32 [M][ldy] or null
    int relu;
    void* Y;              // [M][ldy] fp32, bf16 or e4m3 (saturating, unit scale)
    int ldy;
};

// lane that holds value index r after wave_sumN<N>
template <int N> __device__ __forceinline__ constexpr int sumN_lane(int r) {
    return N == 8 ? ((r >> 2) & 1) * 32 + ((r >> 1) & 1) * 16 + (r & 1) * 8 : ((r >> 1) & 1) * 32 + (r & 1) * 16;
}

template <int PRO, typename OT, int NWV, bool F8 = false>
__global__ __launch_bounds__(NWV * 64) void sgemm_kernel(SGemmArgs a) {
    static_assert(NWV == 2 || NWV == 4, "waves per block");
    constexpr int K = kD, KS = K / 32, RPW = 16 / NWV;       // k-steps; rows a wave stages
    constexpr int NWF = F8 ? KS / 2 : KS;                    // 16-byte weight loads of the wave (fp8: one per k-step pair)
    constexpr int XB = F8 ? 1 : 2;                           // bytes per staged element
    constexpr int LDX = K * XB + 16;                         // bytes per LDS row (1040 / 528): the 16 rows of a fragment read hit 64 different banks
    __shared__ __attribute__((aligned(16))) uint8_t xs[16 * LDX];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int rt = blockIdx.x, nt = blockIdx.y * NWV + wid;

    // ---- everything in flight: the rows this wave stages, the LayerNorm vectors, then the weight fragments and the epilogue operands
    const float* X = reinterpret_cast<const float*>(a.X);
    f32x4 xr[RPW][2];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = min(rt * 16 + wid * RPW + r, a.M - 1);
#pragma unroll
        for (int c = 0; c < 2; ++c) xr[r][c] = *reinterpret_cast<const f32x4*>(X + (size_t)row * K + c * 256 + lane * 4);
    }
    f32x4 lg[2], lb[2];
    if constexpr (PRO == PRO_LN) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            lg[c] = *reinterpret_cast<const f32x4*>(a.lng + c * 256 + lane * 4);
            lb[c] = *reinterpret_cast<const f32x4*>(a.lnb + c * 256 + lane * 4);
        }
    }
    u32x4 wf[NWF];
#pragma unroll
    for (int ks = 0; ks < NWF; ++ks) wf[ks] = __builtin_bit_cast(u32x4, a.W[((size_t)nt * NWF + ks) * 64 + lane]);
    const int m = lane & 15, row = rt * 16 + m, ch = nt * 16 + (lane >> 4) * 4;
    const f32x4 e_bias = *reinterpret_cast<const f32x4*>(a.bias + ch);
    f32x4 e_scale = {1.f, 1.f, 1.f, 1.f};
    if constexpr (F8) e_scale = *reinterpret_cast<const f32x4*>(a.wscale + ch);
    f32x4 e_res = {0.f, 0.f, 0.f, 0.f};
    if (a.res) e_res = *reinterpret_cast<const f32x4*>(a.res + (size_t)min(row, a.M - 1) * a.ldy + ch);
    asm volatile("" : "+v"(xr[0][0]) : : "memory");          // all loads issued, then arithmetic

    // ---- the staged rows -> (LayerNorm) -> bf16 in LDS
    if constexpr (PRO == PRO_LN) {
        float s[RPW], q[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            s[r] = 0.f; q[r] = 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) { s[r] += xr[r][c][i]; q[r] = fmaf(xr[r][c][i], xr[r][c][i], q[r]); }
        }
        const float ts = wave_sumN<RPW>(s), tq = wave_sumN<RPW>(q);      // lane sumN_lane(r) holds row r's totals
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const float rs_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ts), sumN_lane<RPW>(r)));
            const float rq_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tq), sumN_lane<RPW>(r)));
            const float mean = rs_ * (1.0f / K);
            const float var = fmaxf(rq_ * (1.0f / K) - mean * mean, 0.f);   // E[x^2] - mean^2 as the decode kernels (ln512)
            const float rstd = __builtin_amdgcn_rsqf(var + kEps);            // v_rsq_f32 (1 ulp): these kernels serve the bf16 / e4m3 modes only
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) xr[r][c][i] = (xr[r][c][i] - mean) * rstd * lg[c][i] + lb[c][i];
            const int grow = rt * 16 + wid * RPW + r;
            if (blockIdx.y == 0 && a.xout != nullptr && grow < a.M) {
#pragma unroll
                for (int c = 0; c < 2; ++c) *reinterpret_cast<f32x4*>(a.xout + (size_t)grow * K + c * 256 + lane * 4) = xr[r][c];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            uint8_t* dst = xs + (wid * RPW + r) * LDX + (c * 256 + lane * 4) * XB;
            if constexpr (F8) {
                int w = 0;
                w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(xr[r][c][0]), clamp_e4m3(xr[r][c][1]), w, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(xr[r][c][2]), clamp_e4m3(xr[r][c][3]), w, true);
                *reinterpret_cast<uint32_t*>(dst) = (uint32_t)w;
            } else {
                uint2 p;
                p.x = pack_bf16x2(xr[r][c][0], xr[r][c][1]);
                p.y = pack_bf16x2(xr[r][c][2], xr[r][c][3]);
                *reinterpret_cast<uint2*>(dst) = p;
            }
        }
    __syncthreads();

    // ---- this wave's 16 x 16 tile over K = 512: A = weights (registers), B = the rows (LDS); 16 bytes per lane per read either way
    const uint8_t* bp = xs + m * LDX + (lane >> 4) * 16;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (F8) {
#pragma unroll
        for (int p = 0; p < NWF; ++p) {
            const u32x4 bb = *reinterpret_cast<const u32x4*>(bp + p * 64);
            acc0 = mma16_f8((uint64_t)wf[p][0] | ((uint64_t)wf[p][1] << 32), (uint64_t)bb[0] | ((uint64_t)bb[1] << 32), acc0);
            acc1 = mma16_f8((uint64_t)wf[p][2] | ((uint64_t)wf[p][3] << 32), (uint64_t)bb[2] | ((uint64_t)bb[3] << 32), acc1);
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < KS; ks += 2) {
            const u32x4 b0 = *reinterpret_cast<const u32x4*>(bp + ks * 64);
            const u32x4 b1 = *reinterpret_cast<const u32x4*>(bp + ks * 64 + 64);
            acc0 = mma16(wf[ks], b0, acc0);
            acc1 = mma16(wf[ks + 1], b1, acc1);
        }
    }
    if (row >= a.M) return;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = (acc0[i] + acc1[i]) * e_scale[i] + e_bias[i];
        if (a.relu) v[i] = fmaxf(v[i], 0.f);
        v[i] += e_res[i];
    }
    if constexpr (sizeof(OT) == 4) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.Y) + (size_t)row * a.ldy + ch) = f32x4{v[0], v[1], v[2], v[3]};
    } else if constexpr (sizeof(OT) == 2) {
        uint2 p;
        p.x = pack_bf16x2(v[0], v[1]);
        p.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(a.Y) + (size_t)row * a.ldy + ch) = p;
    } else {
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(v[0]), clamp_e4m3(v[1]), w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(v[2]), clamp_e4m3(v[3]), w, true);
        *reinterpret_cast<uint32_t*>(reinterpret_cast<fp8_t*>(a.Y) + (size_t)row * a.ldy + ch) = (uint32_t)w;
    }
}

// W2: Y[M][512] = X[M][2048] (bf16 | e4m3 hidden rows) . W^T + bias + residual; one 16 x 16 tile per block, its four waves split K
template <bool F8 = false>
__global__ __launch_bounds__(256) void sgemm_k_kernel(SGemmArgs a) {
    constexpr int K = kF, KS = K / 32, KW = KS / 4;          // 64 k-steps, 16 per wave
    constexpr int NL = F8 ? KW / 2 : KW;                     // 16-byte loads per operand per wave
    constexpr int XB = F8 ? 1 : 2;
    __shared__ __attribute__((aligned(16))) float red[3 * 64 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int rt = blockIdx.x, nt = blockIdx.y;
    const int m = lane & 15, row = rt * 16 + m, ch = nt * 16 + (lane >> 4) * 4;
    // a lane's 16 bytes: bf16 8 channels of one k-step, e4m3 16 channels of one k-step pair
    const uint8_t* xp = reinterpret_cast<const uint8_t*>(a.X) + ((size_t)min(row, a.M - 1) * K + wid * (K / 4)) * XB + (lane >> 4) * 16;
    u32x4 xb[NL], wf[NL];
#pragma unroll
    for (int ks = 0; ks < NL; ++ks) xb[ks] = *reinterpret_cast<const u32x4*>(xp + ks * 64);
#pragma unroll
    for (int ks = 0; ks < NL; ++ks) wf[ks] = __builtin_bit_cast(u32x4, a.W[((size_t)nt * (NL * 4) + wid * NL + ks) * 64 + lane]);
    f32x4 e_bias = {0.f, 0.f, 0.f, 0.f}, e_res = {0.f, 0.f, 0.f, 0.f}, e_scale = {1.f, 1.f, 1.f, 1.f};
    if (wid == 0) {
        e_bias = *reinterpret_cast<const f32x4*>(a.bias + ch);
        if constexpr (F8) e_scale = *reinterpret_cast<const f32x4*>(a.wscale + ch);
        if (a.res) e_res = *reinterpret_cast<const f32x4*>(a.res + (size_t)min(row, a.M - 1) * a.ldy + ch);
    }
    asm volatile("" : "+v"(xb[0]) : : "memory");
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (F8) {
#pragma unroll
        for (int p = 0; p < NL; ++p) {
            acc0 = mma16_f8((uint64_t)wf[p][0] | ((uint64_t)wf[p][1] << 32), (uint64_t)xb[p][0] | ((uint64_t)xb[p][1] << 32), acc0);
            acc1 = mma16_f8((uint64_t)wf[p][2] | ((uint64_t)wf[p][3] << 32), (uint64_t)xb[p][2] | ((uint64_t)xb[p][3] << 32), acc1);
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < KW; ks += 2) {
            acc0 = mma16(wf[ks], xb[ks], acc0);
            acc1 = mma16(wf[ks + 1], xb[ks + 1], acc1);
        }
    }
    acc0 += acc1;
    if (wid > 0) *reinterpret_cast<f32x4*>(red + ((wid - 1) * 64 + lane) * 4) = acc0;
    __syncthreads();
    if (wid != 0 || row >= a.M) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) acc0 += *reinterpret_cast<const f32x4*>(red + (w * 64 + lane) * 4);   // wave order: bit-reproducible
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = acc0[i] * e_scale[i] + e_bias[i];
        if (a.relu) v[i] = fmaxf(v[i], 0.f);
        v[i] += e_res[i];
    }
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.Y) + (size_t)row * a.ldy + ch) = f32x4{v[0], v[1], v[2], v[3]};
}

// ---- attention of the batched step, second form -------------------------------------------------------------------------------
// t2s_batch_attn_kernel waits for kv_len before it issues any K/V load (the clamp needs it) and then computes all NIT x 64
// positions of the bucket whatever kv_len is: 10.8 us per launch at 64 sequences with ~150 live positions (20 MB of K/V).  Here
//   * the first two chunks (128 positions: a prompt alone is longer than that in most requests) are loaded BLIND at kernel entry,
//     beside kv_len and the q / k / v row; when kv_len has landed the block continues in the straight-line body for its number of
//     live chunks (2, 3, 4, 6, 8, 12 or 16 of 64 positions): the remaining loads of that body, clamped to the last live row;
//   * a chunk with no live position costs no arithmetic (block-uniform branch), a live one half of it: scores on
//     v_dot2c_f32_bf16 with q as a (hi, lo) bf16 pair (t2s_decode.h dot8), softmax in the base-2 domain on v_exp_f32.
// everything behind kv_len for a block with at most NCH live chunks of 64 positions (chunks 0 and 1 arrive loaded)
template <int NCH, bool NTKV>
__device__ __forceinline__ void battn2_rest(const BatchAttnArgs<bf16_t>& a, int h, int b, int n, int64_t n64, const raw16 (&kb)[2], const raw16 (&vb)[2],
                                            float rq, float rk, float rv, uint16_t* qh, uint16_t* ql, uint16_t* knb, float* vn, float (*pacc)[32],
                                            float* pm, float* pl) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int part = tid & 3, rsub = tid >> 2;
    bf16_t* Kp = a.kc + (((size_t)b * kH + h) * a.T) * kDh;
    bf16_t* Vp = a.vc + (((size_t)b * kH + h) * a.T) * kDh;
    const int lastrow = max(n - 1, 0);
    raw16 kr[NCH], vr[NCH];
    kr[0] = kb[0]; vr[0] = vb[0];
    if constexpr (NCH > 1) { kr[1] = kb[1]; vr[1] = vb[1]; }
#pragma unroll
    for (int it = 2; it < NCH; ++it) kr[it] = ldg16w<NTKV>(Kp + (size_t)min(it * 64 + rsub, lastrow) * kDh + part * 8);
#pragma unroll
    for (int it = 2; it < NCH; ++it) vr[it] = ldg16w<NTKV>(Vp + (size_t)min(it * 64 + rsub, lastrow) * kDh + part * 8);
    // q, k AND v pinned here: with only q pinned hipcc hoisted the bf16 rounding of k / v -- and the wait for their cold row -- above
    // the blind K/V loads, which then left a microsecond late
    asm volatile("" : "+v"(rq), "+v"(rk), "+v"(rv) : : "memory");
    stamp(a.dbg, 2);
    if (tid < 32) {
        uint16_t vh, vl;
        split_bf16(rq, vh, vl);
        qh[tid] = vh; if constexpr (kPairAct) ql[tid] = vl;
        const bf16_t kq = f32_to_bf16(rk), vq = f32_to_bf16(rv);
        knb[tid] = kq; vn[tid] = bf16_to_f32(vq);
        const int nw = n64 < 0 ? a.T - 1 : n;     // parked slot (kv_len < 0): away from the rows a staged refill writes
        Kp[(size_t)nw * kDh + tid] = kq; Vp[(size_t)nw * kDh + tid] = vq;
    }
    __syncthreads();
    stamp(a.dbg, 3);
    const XPair qp = xpair_load(qh, ql, part * 8);
    const float scale = 0.17677669529663687f * 1.4426950408889634f;   // 1/sqrt(32) x log2 e
    float sc[NCH + 1];
    float mx = -INFINITY;
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        sc[it] = -INFINITY;
        if (it * 64 < n) {                                   // block-uniform
            const float s = quad_sum(dot8(kr[it], qp));
            if (it * 64 + rsub < n) sc[it] = s * scale;      // rows [0, n): the cache; row n is the new token, below
            mx = fmaxf(mx, sc[it]);
        }
    }
    {   // the new token's own key / value ride with the first quad of wave 0 (from LDS, never from the row being written)
        const float s = quad_sum(dot8(*reinterpret_cast<const raw16*>(knb + part * 8), qp));
        sc[NCH] = tid < 4 ? s * scale : -INFINITY;
        mx = fmaxf(mx, sc[NCH]);
    }
    mx = wave_max(mx);
    stamp(a.dbg, 4);
    const float mref = mx == -INFINITY ? 0.f : mx;               // a wave without live rows
    float l = 0.f, acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        if (it * 64 < n) {
            const bool live = sc[it] != -INFINITY;
            const float p = __builtin_amdgcn_exp2f(sc[it] - mref);   // 0 for a masked row
            const raw16 vm = live ? vr[it] : raw16{0u, 0u, 0u, 0u};  // never multiply a stale row
            float vv[8];
            Unpack<bf16_t, 8>::run(vm, vv);
            l += p;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(p, vv[e], acc[e]);
        }
    }
    {
        const float p = __builtin_amdgcn_exp2f(sc[NCH] - mref);
        l += p;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(vn + part * 8), v1 = *reinterpret_cast<const f32x4*>(vn + part * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[e] = fmaf(p, v0[e], acc[e]); acc[4 + e] = fmaf(p, v1[e], acc[4 + e]); }
    }
    l = wave_sum(part == 0 ? l : 0.f);
    float r4[4], r2[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) r4[i] = halve32_sum(acc[i], acc[i + 4]);
#pragma unroll
    for (int i = 0; i < 2; ++i) r2[i] = halve16_sum(r4[i], r4[i + 2]);
    float r1 = halve8_sum(r2[0], r2[1]);
    r1 += lane_xor<4>(r1);
    if ((lane & 4) == 0) pacc[wid][part * 8 + 4 * (lane >> 5) + 2 * ((lane >> 4) & 1) + ((lane >> 3) & 1)] = r1;
    if (lane == 0) { pm[wid] = mx; pl[wid] = l; }
    stamp(a.dbg, 5);
    __syncthreads();
    if (tid < 32) {
        const float M = fmaxf(fmaxf(pm[0], pm[1]), fmaxf(pm[2], pm[3]));
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = __builtin_amdgcn_exp2f(pm[w] - M);   // exp(-inf) = 0 for an empty wave
            num = fmaf(pacc[w][tid], f, num);
            den = fmaf(pl[w], f, den);
        }
        a.out[(size_t)b * kD + h * 32 + tid] = num * __builtin_amdgcn_rcpf(den);
    }
    stamp(a.dbg, 6);
}

// NTKV: the K/V rows are loaded non-temporally (they are read once per step and, from a few dozen sequences on, are larger than
// the Infinity Cache: without the hint they evict the step's weights, which every GEMM launch then fetches from HBM)
template <int NIT, bool NTKV = false>
__global__ __launch_bounds__(256) void t2s_batch_attn2_kernel(BatchAttnArgs<bf16_t> a) {
    __shared__ __attribute__((aligned(16))) uint16_t qh[32], ql[32], knb[32];
    __shared__ __attribute__((aligned(16))) float vn[32], pacc[4][32];
    __shared__ float pm[4], pl[4];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int part = tid & 3, rsub = tid >> 2;
    const float* row = a.qkv + (size_t)b * 1536 + h * 32;
    const bf16_t* Kp = a.kc + (((size_t)b * kH + h) * a.T) * kDh;
    const bf16_t* Vp = a.vc + (((size_t)b * kH + h) * a.T) * kDh;
    stamp(a.dbg, 0);
    // kv_len FIRST (its low dword: the value fits, and hipcc re-uses an unused upper half as a temporary and waits for it): the
    // in-order load counter then lets the chunk count wait for it alone -- and that wait is pinned BEHIND the blind loads (left
    // to itself hipcc put it right behind the kv_len load: nothing was in flight while the block waited for a cold line)
    int kvl_raw = reinterpret_cast<const int*>(a.kv_len)[2 * b];
    float rq = 0.f, rk = 0.f, rv = 0.f;
    if (tid < 32) { rq = row[tid]; rk = row[512 + tid]; rv = row[1024 + tid]; }
    raw16 kb[2], vb[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) kb[it] = ldg16w<NTKV>(Kp + (size_t)min(it * 64 + rsub, a.T - 1) * kDh + part * 8);
#pragma unroll
    for (int it = 0; it < 2; ++it) vb[it] = ldg16w<NTKV>(Vp + (size_t)min(it * 64 + rsub, a.T - 1) * kDh + part * 8);
    asm volatile("" : "+v"(kvl_raw) : : "memory");
    const int64_t n64 = kvl_raw;
    const int n = (int)(n64 < 0 ? 0 : (n64 > a.T - 1 ? a.T - 1 : n64));     // position of the new token
    // One straight-line body per live-chunk count: a load instruction costs its kibibyte on the texture-address path whether its
    // lanes hit one line or sixteen (0.27 us per dead 64-position chunk per launch at 64 sequences), and loading only the live
    // chunks behind per-chunk branches made hipcc drain the load counter in every branch (0.654 -> 0.729 ms per step).
    const int nch = __builtin_amdgcn_readfirstlane((n + 63) >> 6);
    stamp(a.dbg, 1);
#define GSV_BATTN_REST(N) battn2_rest<N, NTKV>(a, h, b, n, n64, kb, vb, rq, rk, rv, qh, ql, knb, vn, pacc, pm, pl)
    if (nch <= 2) GSV_BATTN_REST(2);
    else if (nch <= 3) GSV_BATTN_REST(3);
    else if (NIT <= 4 || nch <= 4) GSV_BATTN_REST(4);
    else if (nch <= 6) GSV_BATTN_REST(6);
    else if (NIT <= 8 || nch <= 8) GSV_BATTN_REST(8);
    else if (nch <= 12) GSV_BATTN_REST(12);
    else GSV_BATTN_REST(16);
#undef GSV_BATTN_REST
}

}  // namespace gsv
