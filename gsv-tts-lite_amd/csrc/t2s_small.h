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

constexpr int kSmallMaxM = 64;     // rows up to which the chain runs on these kernels (above: t2s_batch.h's 32 x 32 tiles)

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

struct SGemmArgs {
    const void* X;        // sgemm: fp32 [M][512]; sgemm_k: bf16 [M][2048]
    int M;
    const float* lng;     // PRO_LN: LayerNorm weight / bias [512]
    const float* lnb;
    float* xout;          // PRO_LN: the normalised rows [M][512] fp32 (the layer's residual later), written by channel group 0; or null
    const uint4* W;       // pack16_kernel's order
    const float* bias;    // [N]
    const float* res;     // residual rows fp32 [M][ldy] or null
    int relu;
    void* Y;              // [M][ldy] fp32 or bf16
    int ldy;
};

// lane that holds value index r after wave_sumN<N>
template <int N> __device__ __forceinline__ constexpr int sumN_lane(int r) {
    return N == 8 ? ((r >> 2) & 1) * 32 + ((r >> 1) & 1) * 16 + (r & 1) * 8 : ((r >> 1) & 1) * 32 + (r & 1) * 16;
}

template <int PRO, typename OT, int NWV>
__global__ __launch_bounds__(NWV * 64) void sgemm_kernel(SGemmArgs a) {
    static_assert(NWV == 2 || NWV == 4, "waves per block");
    constexpr int K = kD, KS = K / 32, RPW = 16 / NWV;       // k-steps; rows a wave stages
    constexpr int LDX = K + 8;                               // bf16 per LDS row: 1040 bytes -> the 16 rows of a fragment read hit 64 different banks
    __shared__ __attribute__((aligned(16))) bf16_t xs[16 * LDX];
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
    u32x4 wf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wf[ks] = __builtin_bit_cast(u32x4, a.W[((size_t)nt * KS + ks) * 64 + lane]);
    const int m = lane & 15, row = rt * 16 + m, ch = nt * 16 + (lane >> 4) * 4;
    const f32x4 e_bias = *reinterpret_cast<const f32x4*>(a.bias + ch);
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
            const float rstd = 1.0f / sqrtf(var + kEps);
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
            uint2 p;
            p.x = pack_bf16x2(xr[r][c][0], xr[r][c][1]);
            p.y = pack_bf16x2(xr[r][c][2], xr[r][c][3]);
            *reinterpret_cast<uint2*>(xs + (wid * RPW + r) * LDX + c * 256 + lane * 4) = p;
        }
    __syncthreads();

    // ---- this wave's 16 x 16 tile over K = 512: A = weights (registers), B = the rows (LDS)
    const bf16_t* bp = xs + m * LDX + (lane >> 4) * 8;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ks += 2) {
        const u32x4 b0 = *reinterpret_cast<const u32x4*>(bp + ks * 32);
        const u32x4 b1 = *reinterpret_cast<const u32x4*>(bp + ks * 32 + 32);
        acc0 = mma16(wf[ks], b0, acc0);
        acc1 = mma16(wf[ks + 1], b1, acc1);
    }
    if (row >= a.M) return;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = acc0[i] + acc1[i] + e_bias[i];
        if (a.relu) v[i] = fmaxf(v[i], 0.f);
        v[i] += e_res[i];
    }
    if constexpr (sizeof(OT) == 4) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.Y) + (size_t)row * a.ldy + ch) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
        uint2 p;
        p.x = pack_bf16x2(v[0], v[1]);
        p.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(a.Y) + (size_t)row * a.ldy + ch) = p;
    }
}

// W2: Y[M][512] = X[M][2048] (bf16) . W^T + bias + residual; one 16 x 16 tile per block, its four waves split K
__global__ __launch_bounds__(256) void sgemm_k_kernel(SGemmArgs a) {
    constexpr int K = kF, KS = K / 32, KW = KS / 4;          // 64 k-steps, 16 per wave
    __shared__ __attribute__((aligned(16))) float red[3 * 64 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int rt = blockIdx.x, nt = blockIdx.y;
    const int m = lane & 15, row = rt * 16 + m, ch = nt * 16 + (lane >> 4) * 4;
    const bf16_t* xp = reinterpret_cast<const bf16_t*>(a.X) + (size_t)min(row, a.M - 1) * K + wid * (K / 4) + (lane >> 4) * 8;
    u32x4 xb[KW], wf[KW];
#pragma unroll
    for (int ks = 0; ks < KW; ++ks) xb[ks] = *reinterpret_cast<const u32x4*>(xp + ks * 32);
#pragma unroll
    for (int ks = 0; ks < KW; ++ks) wf[ks] = __builtin_bit_cast(u32x4, a.W[((size_t)nt * KS + wid * KW + ks) * 64 + lane]);
    f32x4 e_bias = {0.f, 0.f, 0.f, 0.f}, e_res = {0.f, 0.f, 0.f, 0.f};
    if (wid == 0) {
        e_bias = *reinterpret_cast<const f32x4*>(a.bias + ch);
        if (a.res) e_res = *reinterpret_cast<const f32x4*>(a.res + (size_t)min(row, a.M - 1) * a.ldy + ch);
    }
    asm volatile("" : "+v"(xb[0]) : : "memory");
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KW; ks += 2) {
        acc0 = mma16(wf[ks], xb[ks], acc0);
        acc1 = mma16(wf[ks + 1], xb[ks + 1], acc1);
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
        v[i] = acc0[i] + e_bias[i];
        if (a.relu) v[i] = fmaxf(v[i], 0.f);
        v[i] += e_res[i];
    }
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.Y) + (size_t)row * a.ldy + ch) = f32x4{v[0], v[1], v[2], v[3]};
}

// ---- attention of the batched step, second form -------------------------------------------------------------------------------
// t2s_batch_attn_kernel waits for kv_len before it issues any K/V load (the clamp needs it) and then computes all NIT x 64
// positions of the bucket whatever kv_len is: 10.8 us per launch at 64 sequences with ~150 live positions (20 MB of K/V).  Here
//   * the first two chunks (128 positions: a prompt alone is longer than that in most requests) are loaded BLIND at kernel entry,
//     beside kv_len and the q / k / v row; the remaining chunks follow when kv_len has landed, clamped to the last live row as
//     before (no HBM bytes for dead positions);
//   * a chunk with no live position costs no arithmetic (block-uniform branch), a live one half of it: scores on
//     v_dot2c_f32_bf16 with q as a (hi, lo) bf16 pair (t2s_decode.h dot8), softmax in the base-2 domain on v_exp_f32.
template <int NIT>
__global__ __launch_bounds__(256) void t2s_batch_attn2_kernel(BatchAttnArgs<bf16_t> a) {
    __shared__ __attribute__((aligned(16))) uint16_t qh[32], ql[32], knb[32];
    __shared__ __attribute__((aligned(16))) float vn[32], pacc[4][32];
    __shared__ float pm[4], pl[4];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int part = tid & 3, rsub = tid >> 2;
    const float* row = a.qkv + (size_t)b * 1536 + h * 32;
    bf16_t* Kp = a.kc + (((size_t)b * kH + h) * a.T) * kDh;
    bf16_t* Vp = a.vc + (((size_t)b * kH + h) * a.T) * kDh;
    constexpr int NB0 = NIT < 2 ? NIT : 2;
    // kv_len FIRST: the in-order load counter then lets the clamp wait for it alone
    const int64_t n64 = a.kv_len[b];
    float rq = 0.f, rk = 0.f, rv = 0.f;
    if (tid < 32) { rq = row[tid]; rk = row[512 + tid]; rv = row[1024 + tid]; }
    raw16 kr[NIT], vr[NIT];
#pragma unroll
    for (int it = 0; it < NB0; ++it) kr[it] = ldg16(Kp + (size_t)min(it * 64 + rsub, a.T - 1) * kDh + part * 8);
#pragma unroll
    for (int it = 0; it < NB0; ++it) vr[it] = ldg16(Vp + (size_t)min(it * 64 + rsub, a.T - 1) * kDh + part * 8);
    const int n = (int)(n64 < 0 ? 0 : (n64 > a.T - 1 ? a.T - 1 : n64));     // position of the new token
    const int lastrow = max(n - 1, 0);
    // (loading only the live chunks behind block-uniform branches was measured: hipcc drains the load counter in every branch,
    // 0.654 -> 0.729 ms per step at 64 sequences)
#pragma unroll
    for (int it = NB0; it < NIT; ++it) kr[it] = ldg16(Kp + (size_t)min(it * 64 + rsub, lastrow) * kDh + part * 8);
#pragma unroll
    for (int it = NB0; it < NIT; ++it) vr[it] = ldg16(Vp + (size_t)min(it * 64 + rsub, lastrow) * kDh + part * 8);
    asm volatile("" : "+v"(rq) : : "memory");
    if (tid < 32) {
        uint16_t vh, vl;
        split_bf16(rq, vh, vl);
        qh[tid] = vh; ql[tid] = vl;
        const bf16_t kq = f32_to_bf16(rk), vq = f32_to_bf16(rv);
        knb[tid] = kq; vn[tid] = bf16_to_f32(vq);
        const int nw = n64 < 0 ? a.T - 1 : n;     // parked slot (kv_len < 0): away from the rows a staged refill writes
        Kp[(size_t)nw * kDh + tid] = kq; Vp[(size_t)nw * kDh + tid] = vq;
    }
    __syncthreads();
    const XPair qp = xpair_load(qh, ql, part * 8);
    const float scale = 0.17677669529663687f * 1.4426950408889634f;   // 1/sqrt(32) x log2 e
    float sc[NIT + 1];
    float mx = -INFINITY;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        sc[it] = -INFINITY;
        if (it * 64 < n) {                                   // block-uniform
            const float s = quad_sum(dot8(kr[it], qp));
            if (it * 64 + rsub < n) sc[it] = s * scale;      // rows [0, n): the cache; row n is the new token, below
            mx = fmaxf(mx, sc[it]);
        }
    }
    {   // the new token's own key / value ride with the first quad of wave 0 (from LDS, never from the row being written)
        const float s = quad_sum(dot8(*reinterpret_cast<const raw16*>(knb + part * 8), qp));
        sc[NIT] = tid < 4 ? s * scale : -INFINITY;
        mx = fmaxf(mx, sc[NIT]);
    }
    mx = wave_max(mx);
    const float mref = mx == -INFINITY ? 0.f : mx;               // a wave without live rows
    float l = 0.f, acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
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
        const float p = __builtin_amdgcn_exp2f(sc[NIT] - mref);
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
        a.out[(size_t)b * kD + h * 32 + tid] = num / den;
    }
}

}  // namespace gsv
