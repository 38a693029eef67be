// GPT decode-step kernels for gfx950: one token per sequence through a post-LN block stack.
//
// Reference semantics: T2SBlock.decode_next_token, gsv_tts/GPT_SoVITS/GPT/t2s_model.py:67-105
//   qkv = x Wqkv^T + b ; K/V appended at kv_len[b] ; causal attention over [0, kv_len[b]] ;
//   x = LN1(x + attn Wo^T + bo) ; x = LN2(x + W2 relu(W1 x + b1) + b2)
//
// MI355X mapping (DESIGN.md "GPT decode"): the step is bandwidth/latency bound (weights
// stream once per token), so each layer is TWO kernels, split at the two all-to-all points,
// and every normalisation/residual lives in the prologue of the consumer:
//
//   attn kernel  grid (16 heads, B): [prev FFN partial-sum + b2 + residual + LN2] -> x ;
//                this head's 96 QKV rows (wave-per-row, 16 B/lane coalesced weight stream) ;
//                KV append ; single-query attention over the head's contiguous [T][32] panel ;
//                out_proj restricted to this head's 32 input columns -> partial y[h][512]
//   ffn kernel   grid (32 slices, B): [sum of 16 head partials + bo + residual + LN1] -> x1 ;
//                64 hidden units of W1 (+ReLU) ; their 64 columns of W2 -> partial z[j][512]
//
// Partials are summed in fixed index order by the consumer => bit-reproducible run to run
// (no atomics), which greedy-token parity against the CPU oracle depends on.
// Weight panels are pre-packed at load (t2s_pack.h) so every wave instruction reads 1 KiB
// of consecutive bytes.
#pragma once
#include "gsv_common.h"

namespace gsv {

constexpr int kD = 512;        // hidden
constexpr int kH = 16;         // heads
constexpr int kDh = 32;        // head dim
constexpr int kF = 2048;       // MLP hidden
constexpr int kNJ = 32;        // FFN slices (blocks) per sequence
constexpr int kFJ = kF / kNJ;  // hidden units per slice
constexpr int kNP = 16;        // logits slices per sequence
constexpr float kEps = 1e-5f;

struct TokPart {
    float v;
    int idx;
};

// ---- shared pieces ---------------------------------------------------------------------------
//
// Latency discipline.  The step is a chain of ~50 dependent kernels and, at batch 1, each kernel
// runs on a handful of CUs, so what matters is the length of the dependent chain inside a kernel:
//  * every global load whose ADDRESS does not depend on this kernel's activations -- partial sums,
//    weight rows, K/V rows, out-proj panel, biases -- is issued at kernel entry into registers, in
//    the order it will be consumed (loads retire in order), and an opaque asm pins "all loads
//    issued, then arithmetic" (hipcc otherwise sinks the loads next to their first use);
//  * the instruction stream per wave is what the kernel's wall time is made of once the loads
//    overlap (one wave issues ~1 VALU op per 4-5 cycles), so a block is 16 waves (1024 threads):
//    4 waves per SIMD share the rows/keys, each wave's stream is a quarter of a 256-thread block's.

constexpr int kNT = 1024;       // threads per decode block
constexpr int kNW = kNT / 64;   // 16 waves

using raw16 = u32x4;  // 16 bytes of operands, as loaded

template <typename WT, int N> struct Unpack;           // raw16 -> float[N]
template <> struct Unpack<float, 4> {
    static __device__ __forceinline__ void run(const raw16& r, float (&o)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = __uint_as_float(r[i]);
    }
};
template <> struct Unpack<bf16_t, 8> {
    static __device__ __forceinline__ void run(const raw16& r, float (&o)[8]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[2 * j] = __uint_as_float(r[j] << 16);
            o[2 * j + 1] = __uint_as_float(r[j] & 0xffff0000u);
        }
    }
};

template <typename WT> struct Geo {
    static constexpr int EPL = 16 / sizeof(WT);  // elements per 16-byte lane load
    static constexpr int CPR = kD / EPL / 64;    // 16-byte chunks per lane for one 512-wide row (1 bf16, 2 f32)
};

__device__ __forceinline__ raw16 ldg16(const void* p) { return *reinterpret_cast<const raw16*>(p); }

// dot of one lane's slice of a 512-wide weight row (CPR chunks) with the lane's activations.
// bf16: lane owns x[lane*8 .. +7]; f32: chunks c=0,1 own x[c*256 + lane*4 .. +3].
template <typename WT>
__device__ __forceinline__ float row_dot(const raw16 (&w)[Geo<WT>::CPR], const float (&xr)[8]) {
    constexpr int EPL = Geo<WT>::EPL;
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < Geo<WT>::CPR; ++c) {
        float wv[EPL];
        Unpack<WT, EPL>::run(w[c], wv);
#pragma unroll
        for (int i = 0; i < EPL; ++i) a = fmaf(wv[i], xr[c * EPL + i], a);
    }
    return a;
}
template <typename WT>
__device__ __forceinline__ void row_load(const WT* row, raw16 (&w)[Geo<WT>::CPR]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < Geo<WT>::CPR; ++c) w[c] = ldg16(row + c * (kD / Geo<WT>::CPR) + lane * Geo<WT>::EPL);
}
template <typename WT>
__device__ __forceinline__ void lane_x(const float* xs, float (&xr)[8]) {
    const int lane = threadIdx.x & 63;
    constexpr int EPL = Geo<WT>::EPL;
#pragma unroll
    for (int c = 0; c < Geo<WT>::CPR; ++c)
#pragma unroll
        for (int i = 0; i < EPL; ++i) xr[c * EPL + i] = xs[c * (kD / Geo<WT>::CPR) + lane * EPL + i];
}

// N per-lane partials -> N wave totals: halve the value count at each of the first log2(N)
// butterfly levels (N-1 shuffles), then finish the remaining levels on one value.
// Every lane returns the total of value index sumN_index<N>().
template <int N> __device__ __forceinline__ float wave_sumN(const float (&v)[N]);
template <> __device__ __forceinline__ float wave_sumN<8>(const float (&v)[8]) {
    const int lane = threadIdx.x & 63;
    float a[4], b[2], c;
    {
        const bool hi = lane & 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float send = hi ? v[i] : v[i + 4];
            const float keep = hi ? v[i + 4] : v[i];
            a[i] = keep + __shfl_xor(send, 32, 64);
        }
    }
    {
        const bool hi = lane & 16;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float send = hi ? a[i] : a[i + 2];
            const float keep = hi ? a[i + 2] : a[i];
            b[i] = keep + __shfl_xor(send, 16, 64);
        }
    }
    {
        const bool hi = lane & 8;
        const float send = hi ? b[0] : b[1];
        const float keep = hi ? b[1] : b[0];
        c = keep + __shfl_xor(send, 8, 64);
    }
    c += __shfl_xor(c, 4, 64);
    c += __shfl_xor(c, 2, 64);
    c += __shfl_xor(c, 1, 64);
    return c;
}
template <> __device__ __forceinline__ float wave_sumN<4>(const float (&v)[4]) {
    const int lane = threadIdx.x & 63;
    float b[2], c;
    {
        const bool hi = lane & 32;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float send = hi ? v[i] : v[i + 2];
            const float keep = hi ? v[i + 2] : v[i];
            b[i] = keep + __shfl_xor(send, 32, 64);
        }
    }
    {
        const bool hi = lane & 16;
        const float send = hi ? b[0] : b[1];
        const float keep = hi ? b[1] : b[0];
        c = keep + __shfl_xor(send, 16, 64);
    }
    c += __shfl_xor(c, 8, 64);
    c += __shfl_xor(c, 4, 64);
    c += __shfl_xor(c, 2, 64);
    c += __shfl_xor(c, 1, 64);
    return c;
}
template <int N> __device__ __forceinline__ int sumN_index();
template <> __device__ __forceinline__ int sumN_index<8>() {
    const int lane = threadIdx.x & 63;
    return ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
}
template <> __device__ __forceinline__ int sumN_index<4>() {
    const int lane = threadIdx.x & 63;
    return ((lane >> 5) & 1) * 2 + ((lane >> 4) & 1);
}

// LayerNorm of a 512-vector, thread t < 512 owning element t (two-pass, like torch)
__device__ __forceinline__ float ln512(float v, bool owner, float g, float bta, float* red) {
    const float mean = block_sum<kNW>(owner ? v : 0.f, red) * (1.0f / kD);
    const float d = owner ? v - mean : 0.f;
    const float var = block_sum<kNW>(d * d, red) * (1.0f / kD);
    const float rs = 1.0f / sqrtf(var + kEps);
    return d * rs * g + bta;
}

// sum of NPART partial 512-vectors (+ bias + residual), thread t < 512 owning element t; the loads
// are issued by `issue` at kernel entry, the sum (fixed index order) happens in `finish`.
template <int NPART> struct PartialSum {
    float p[NPART];
    float bias, resid, lng, lnb;
    __device__ __forceinline__ void issue(const float* __restrict__ part, const float* __restrict__ b,
                                          const float* __restrict__ r, const float* __restrict__ g,
                                          const float* __restrict__ beta) {
        const int t = threadIdx.x & (kD - 1);
#pragma unroll
        for (int j = 0; j < NPART; ++j) p[j] = part[(size_t)j * kD + t];
        bias = b[t];
        resid = r[t];
        lng = g[t];
        lnb = beta[t];
    }
    __device__ __forceinline__ float finish() {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NPART; ++j) s += p[j];
        return s + bias + resid;
    }
};

// panel GEMV out[row] = dot(P[row][0:K], v) for 512 rows, K in {32, 64}: K/EPL lanes per row.
template <typename WT, int K> struct Panel {
    static constexpr int EPL = Geo<WT>::EPL;
    static constexpr int LPR = K / EPL;
    static constexpr int RPI = kNT / LPR;
    static constexpr int NIT = kD / RPI;
    raw16 w[NIT];
    __device__ __forceinline__ void issue(const WT* __restrict__ panel) {
        const int tid = threadIdx.x;
        const int part = tid % LPR, rsub = tid / LPR;
#pragma unroll
        for (int it = 0; it < NIT; ++it) w[it] = ldg16(panel + (size_t)(rsub + it * RPI) * K + part * EPL);
    }
    __device__ __forceinline__ void finish(const float* __restrict__ vec_lds, float* __restrict__ out) {
        const int tid = threadIdx.x;
        const int part = tid % LPR, rsub = tid / LPR;
        float vr[EPL];
#pragma unroll
        for (int i = 0; i < EPL; ++i) vr[i] = vec_lds[part * EPL + i];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            float wv[EPL];
            Unpack<WT, EPL>::run(w[it], wv);
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < EPL; ++i) s = fmaf(wv[i], vr[i], s);
#pragma unroll
            for (int m = LPR / 2; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
            if (part == 0) out[rsub + it * RPI] = s;
        }
    }
};

// ---- attention kernel ----------------------------------------------------------------------

template <typename WT>
struct AttnArgs {
    // layer input: MODE 0 -> xdirect[B][512]; MODE 1 -> LN2(sum_j zpart + b2 + x1) of the previous layer
    const float* xdirect;
    const float* zpart;  // [B][kNJ][512]
    const float* b2;
    const float* x1;     // [B][512]
    const float* ln2g;
    const float* ln2b;
    float* xout;         // [B][512] layer input, written by head 0 (residual for the ffn kernel)
    const WT* wqkv;      // [16][96][512]  rows: q(32) k(32) v(32) of head h
    const float* bqkv;   // [16][96]
    const WT* wo;        // [16][512][32]  wo[h][n][d] = Wo[n][h*32+d]
    WT* kc;              // this layer: [B][16][T][32]
    WT* vc;
    const int64_t* kv_len;
    int T;
    float* ypart;        // [B][16][512]
};

constexpr int kAttnLdsFloats = kD + 96 + 32 + 16 + kNW * 32;  // + T scores

template <typename WT, int MODE>
__global__ __launch_bounds__(kNT) void t2s_attn_kernel(AttnArgs<WT> a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;            // 512
    float* qkv = xs + kD;        // 96
    float* att = qkv + 96;       // 32
    float* red = att + 32;       // 16
    float* pacc = red + 16;      // 16*32
    float* sc = pacc + kNW * 32; // T
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int EPL = Geo<WT>::EPL;
    constexpr int CPR = Geo<WT>::CPR;
    constexpr int LPR = kDh / EPL;         // lanes per K/V row
    constexpr int RPI = kNT / LPR;         // K/V rows per block iteration (256 bf16, 128 f32)
    constexpr int KCH = 2;                 // iterations held in registers per chunk (512 positions bf16, 256 f32)
    constexpr int RW = 96 / kNW;           // 6 QKV rows per wave
    const bool owner = tid < kD;

    int n = (int)a.kv_len[b];
    if (n > a.T - 1) n = a.T - 1;  // memory safety only; the host never steps a full cache
    if (n < 0) n = 0;
    const WT* Kp = a.kc + (((size_t)b * kH + h) * a.T) * kDh;
    const WT* Vp = a.vc + (((size_t)b * kH + h) * a.T) * kDh;
    const int part = tid % LPR, rsub = tid / LPR;

    // ---- issue everything whose address is known now, in consumption order
    PartialSum<MODE ? kNJ : 1> ps;
    float xd = 0.f;
    if (owner) {
        if constexpr (MODE == 0) xd = a.xdirect[(size_t)b * kD + tid];
        else ps.issue(a.zpart + (size_t)b * kNJ * kD, a.b2, a.x1 + (size_t)b * kD, a.ln2g, a.ln2b);
    }
    asm volatile("" : : : "memory");  // partials first: consumed first, and loads retire in order
    const WT* wp = a.wqkv + ((size_t)h * 96 + wid * RW) * kD;
    raw16 wq[RW][CPR];
#pragma unroll
    for (int r = 0; r < RW; ++r) row_load<WT>(wp + (size_t)r * kD, wq[r]);
    // K/V rows are loaded UNCONDITIONALLY from a clamped (always valid) row and masked at use: a
    // per-element "load or zero" select makes hipcc branch around each load and drain vmcnt(0)
    raw16 kreg[KCH], vreg[KCH];
#pragma unroll
    for (int it = 0; it < KCH; ++it) kreg[it] = ldg16(Kp + (size_t)min(rsub + it * RPI, n) * kDh + part * EPL);
#pragma unroll
    for (int it = 0; it < KCH; ++it) vreg[it] = ldg16(Vp + (size_t)min(rsub + it * RPI, n) * kDh + part * EPL);
    Panel<WT, kDh> po;
    po.issue(a.wo + (size_t)h * kD * kDh);
    const int oi = sumN_index<8>();
    const float bq = a.bqkv[h * 96 + wid * RW + min(oi, RW - 1)];
    // Pin "all loads issued, THEN arithmetic": the opaque asm redefines the head of the partial-sum
    // chain, so no add can be scheduled above it, while the memory clobber keeps every load above
    // it.  It only needs the FIRST-issued load to have landed.
    if constexpr (MODE == 0) asm volatile("" : "+v"(xd) : : "memory");
    else asm volatile("" : "+v"(ps.p[0]) : : "memory");

    // ---- layer input
    float v;
    if constexpr (MODE == 0) v = xd;
    else v = ln512(owner ? ps.finish() : 0.f, owner, ps.lng, ps.lnb, red);
    if (owner) {
        xs[tid] = v;
        if (h == 0) a.xout[(size_t)b * kD + tid] = v;
    }
    __syncthreads();

    // ---- q, k, v of this head: 96 rows, 6 per wave
    {
        float xr[8];
        lane_x<WT>(xs, xr);
        float acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = u < RW ? row_dot<WT>(wq[u < RW ? u : 0], xr) : 0.f;
        const float tot = wave_sumN<8>(acc);
        if ((lane & 7) == 0 && oi < RW) qkv[wid * RW + oi] = tot + bq;
    }
    __syncthreads();

    WT* Kw = a.kc + (((size_t)b * kH + h) * a.T) * kDh;
    WT* Vw = a.vc + (((size_t)b * kH + h) * a.T) * kDh;
    if (tid < 64) {
        // round through the cache type so this step sees exactly what later steps will read back
        WT s = from_f32<WT>(qkv[32 + tid]);
        qkv[32 + tid] = to_f32<WT>(s);
        if (tid < 32) Kw[(size_t)n * kDh + tid] = s; else Vw[(size_t)n * kDh + tid - 32] = s;
    }
    __syncthreads();

    // ---- scores over [0, n]; position n (this token) comes from LDS, the rest from registers/chunks
    const float scale = 0.17677669529663687f;  // 1/sqrt(32)
    float qr[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) qr[i] = qkv[part * EPL + i];
    for (int c0 = 0; c0 < n; c0 += KCH * RPI) {
        if (c0 > 0) {
#pragma unroll
            for (int it = 0; it < KCH; ++it)
                kreg[it] = ldg16(Kp + (size_t)min(c0 + rsub + it * RPI, n) * kDh + part * EPL);
        }
#pragma unroll
        for (int it = 0; it < KCH; ++it) {
            const int r = c0 + rsub + it * RPI;
            float kk[EPL];
            Unpack<WT, EPL>::run(kreg[it], kk);
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < EPL; ++i) s = fmaf(qr[i], kk[i], s);
#pragma unroll
            for (int m = LPR / 2; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
            if (part == 0 && r < n) sc[r] = s * scale;
        }
    }
    if (tid < LPR) {  // the new token's own key
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < EPL; ++i) s = fmaf(qr[i], qkv[32 + part * EPL + i], s);
#pragma unroll
        for (int m = LPR / 2; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        if (part == 0) sc[n] = s * scale;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int r = tid; r <= n; r += kNT) mx = fmaxf(mx, sc[r]);
    mx = block_max<kNW>(mx, red);
    float sum = 0.f;
    for (int r = tid; r <= n; r += kNT) {
        float e = expf(sc[r] - mx);
        sc[r] = e;
        sum += e;
    }
    sum = block_sum<kNW>(sum, red);  // (its barriers also publish sc[])
    {
        float acc[EPL];
#pragma unroll
        for (int i = 0; i < EPL; ++i) acc[i] = 0.f;
        for (int c0 = 0; c0 < n; c0 += KCH * RPI) {
            if (c0 > 0) {
#pragma unroll
                for (int it = 0; it < KCH; ++it)
                    vreg[it] = ldg16(Vp + (size_t)min(c0 + rsub + it * RPI, n) * kDh + part * EPL);
            }
#pragma unroll
            for (int it = 0; it < KCH; ++it) {
                const int r = c0 + rsub + it * RPI;
                float vv[EPL];
                Unpack<WT, EPL>::run(vreg[it], vv);
                const bool live = r < n;   // clamped rows hold other data: mask both factors
                const float p = live ? sc[r] / sum : 0.f;
#pragma unroll
                for (int i = 0; i < EPL; ++i) acc[i] = fmaf(p, live ? vv[i] : 0.f, acc[i]);
            }
        }
        if (tid < LPR) {
            const float p = sc[n] / sum;
#pragma unroll
            for (int i = 0; i < EPL; ++i) acc[i] = fmaf(p, qkv[64 + part * EPL + i], acc[i]);
        }
#pragma unroll
        for (int m = 32; m >= LPR; m >>= 1) {
#pragma unroll
            for (int i = 0; i < EPL; ++i) acc[i] += __shfl_xor(acc[i], m, 64);
        }
        if (lane < LPR) {
#pragma unroll
            for (int i = 0; i < EPL; ++i) pacc[wid * 32 + part * EPL + i] = acc[i];
        }
    }
    __syncthreads();
    if (tid < 32) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < kNW; ++w) s += pacc[w * 32 + tid];
        att[tid] = s;
    }
    __syncthreads();
    po.finish(att, a.ypart + ((size_t)b * kH + h) * kD);
}

// ---- ffn kernel ----------------------------------------------------------------------------

template <typename WT>
struct FfnArgs {
    const float* ypart;  // [B][16][512]
    const float* bo;
    const float* x;      // [B][512] layer input (residual)
    const float* ln1g;
    const float* ln1b;
    float* x1out;        // [B][512] LN1 output, written by slice 0
    const WT* w1;        // [2048][512] (torch layout; slice j = rows j*64..)
    const float* b1;
    const WT* w2p;       // [32][512][64]  w2p[j][n][i] = W2[n][j*64+i]
    float* zpart;        // [B][32][512]
};

template <typename WT>
__global__ __launch_bounds__(kNT) void t2s_ffn_kernel(FfnArgs<WT> a) {
    __shared__ __attribute__((aligned(16))) float smem[kD + kFJ + 16];
    float* xs = smem;
    float* hb = xs + kD;
    float* red = hb + kFJ;
    const int j = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int CPR = Geo<WT>::CPR;
    constexpr int RW = kFJ / kNW;  // 4 W1 rows per wave
    const bool owner = tid < kD;

    PartialSum<kH> ps;
    if (owner) ps.issue(a.ypart + (size_t)b * kH * kD, a.bo, a.x + (size_t)b * kD, a.ln1g, a.ln1b);
    asm volatile("" : : : "memory");
    const int row0 = j * kFJ + wid * RW;
    raw16 w1r[RW][CPR];
#pragma unroll
    for (int r = 0; r < RW; ++r) row_load<WT>(a.w1 + (size_t)(row0 + r) * kD, w1r[r]);
    Panel<WT, kFJ> p2;
    p2.issue(a.w2p + (size_t)j * kD * kFJ);
    const int oi = sumN_index<RW>();
    const float b1r = a.b1[row0 + oi];
    asm volatile("" : "+v"(ps.p[0]) : : "memory");

    const float v = ln512(owner ? ps.finish() : 0.f, owner, ps.lng, ps.lnb, red);
    if (owner) {
        xs[tid] = v;
        if (j == 0) a.x1out[(size_t)b * kD + tid] = v;
    }
    __syncthreads();
    {
        float xr[8];
        lane_x<WT>(xs, xr);
        float acc[RW];
#pragma unroll
        for (int u = 0; u < RW; ++u) acc[u] = row_dot<WT>(w1r[u], xr);
        const float tot = wave_sumN<RW>(acc);
        if ((lane & 15) == 0) hb[wid * RW + oi] = fmaxf(tot + b1r, 0.f);
    }
    __syncthreads();
    p2.finish(hb, a.zpart + ((size_t)b * kNJ + j) * kD);
}

// ---- logits kernel -------------------------------------------------------------------------

template <typename WT>
struct LogitsArgs {
    // final hidden: MODE 1 -> LN2(sum zpart + b2 + x1) of the last layer; MODE 0 -> hdirect[B][512]
    const float* hdirect;
    const float* zpart;
    const float* b2;
    const float* x1;
    const float* ln2g;
    const float* ln2b;
    const WT* wp;        // [V][512]
    int V, eos;
    int vlimit;          // logits with v >= vlimit are -inf (first sample drops the EOS column)
    int slot0;           // first state slot of row 0
    const int32_t* step;
    const int32_t* ctl;  // {use_override, suppress_steps, rep_enabled, -}
    const float* fctl;   // {rep_penalty}
    const uint8_t* seen; // [B][V]
    float* logits;       // [B][V]
    float* hidden;       // [B][512]
    TokPart* tokpart;    // [B][kNP]
    int64_t* kv_len;     // bumped by slice 0 when bump != 0
    int bump;
};

template <typename WT, int MODE>
__global__ __launch_bounds__(kNT) void t2s_logits_kernel(LogitsArgs<WT> a) {
    __shared__ __attribute__((aligned(16))) float smem[kD + 16 + 128];
    float* xs = smem;
    float* red = xs + kD;
    float* lg = red + 16;  // up to 128 rows per slice
    const int p = blockIdx.x, r_ = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = a.slot0 + r_;  // state slot; row r_ of zpart/x1/hdirect
    constexpr int CPR = Geo<WT>::CPR;
    const int rpb = (a.V + kNP - 1) / kNP;      // rows per slice (<= 128)
    const int rww = (rpb + kNW - 1) / kNW;      // rows per wave (<= 8)
    const int vbase = p * rpb;
    const int nrow = min(rpb, a.V - vbase);
    const bool owner = tid < kD;

    PartialSum<MODE ? kNJ : 1> ps;
    float xd = 0.f;
    if (owner) {
        if constexpr (MODE == 0) xd = a.hdirect[(size_t)r_ * kD + tid];
        else ps.issue(a.zpart + (size_t)r_ * kNJ * kD, a.b2, a.x1 + (size_t)r_ * kD, a.ln2g, a.ln2b);
    }
    asm volatile("" : : : "memory");
    raw16 wr[8][CPR];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int v = min(vbase + wid * rww + min(u, rww - 1), a.V - 1);
        row_load<WT>(a.wp + (size_t)v * kD, wr[u]);
    }
    const bool sup = a.step[b] < a.ctl[1];
    const bool rep = a.ctl[2] != 0;
    const float rp = a.fctl[0];
    const int oi = sumN_index<8>();
    const int myr = wid * rww + oi;             // slice row this lane will emit (if oi < rww)
    const uint8_t sn = a.seen[(size_t)b * a.V + min(vbase + myr, a.V - 1)];
    if constexpr (MODE == 0) asm volatile("" : "+v"(xd) : : "memory");
    else asm volatile("" : "+v"(ps.p[0]) : : "memory");

    float v;
    if constexpr (MODE == 0) v = xd;
    else v = ln512(owner ? ps.finish() : 0.f, owner, ps.lng, ps.lnb, red);
    if (owner) {
        xs[tid] = v;
        if (p == 0) a.hidden[(size_t)b * kD + tid] = v;
    }
    __syncthreads();
    {
        float xr[8];
        lane_x<WT>(xs, xr);
        float acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = row_dot<WT>(wr[u], xr);
        const float tot = wave_sumN<8>(acc);
        if ((lane & 7) == 0 && oi < rww && myr < nrow) {
            const int vv = vbase + myr;
            float l = tot;
            if (vv >= a.vlimit) l = -INFINITY;
            if (sup && (vv == 280 || vv == 486 || vv == a.eos)) l = -INFINITY;
            if (rep && sn) l = l < 0.f ? l * rp : l / rp;
            lg[myr] = l;
            a.logits[(size_t)b * a.V + vv] = l;
        }
    }
    __syncthreads();
    if (wid == 0) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int r = lane; r < nrow; r += 64) {
            float l = lg[r];
            if (l > bv || (l == bv && vbase + r < bi)) { bv = l; bi = vbase + r; }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            float ov = __shfl_xor(bv, m, 64);
            int oi2 = __shfl_xor(bi, m, 64);
            if (ov > bv || (ov == bv && oi2 < bi)) { bv = ov; bi = oi2; }
        }
        if (lane == 0) {
            TokPart tp; tp.v = bv; tp.idx = bi;
            a.tokpart[(size_t)b * kNP + p] = tp;
            if (p == 0 && a.bump) a.kv_len[b] += 1;
        }
    }
}

// ---- token kernel: pending token -> pre_tokens / seen / eos_at, and the next input embedding -----

struct TokenArgs {
    const TokPart* tokpart;  // [B][kNP]
    const int64_t* tok_override;
    const int32_t* ctl;
    const int64_t* kv_len;
    const int64_t* x_len;
    int64_t* pre_tokens;     // [B][T+1]
    uint8_t* seen;           // [B][V]
    int32_t* step;
    int32_t* eos_at;
    const float* emb;        // [V][512] audio embedding
    const float* pe;         // [n_pos][512] alpha_audio * pe
    float* xcur;             // [B][512]
    int T, V, eos, n_pos, advance;
};

__global__ __launch_bounds__(256) void t2s_token_kernel(TokenArgs a) {
    __shared__ int s_tok;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        int tok;
        if (a.ctl[0] != 0) {
            tok = (int)a.tok_override[b];
        } else {
            float bv = -INFINITY;
            tok = 0x7fffffff;
            for (int p = 0; p < kNP; ++p) {
                TokPart t = a.tokpart[(size_t)b * kNP + p];
                if (t.v > bv || (t.v == bv && t.idx < tok)) { bv = t.v; tok = t.idx; }
            }
        }
        if (tok < 0 || tok >= a.V) tok = 0;
        s_tok = tok;
        const int64_t n = a.kv_len[b];
        if (n >= 0 && n <= a.T) a.pre_tokens[(size_t)b * (a.T + 1) + n] = tok;
        if (a.ctl[2] != 0) a.seen[(size_t)b * a.V + tok] = 1;
        if (tok == a.eos && a.eos_at[b] < 0) a.eos_at[b] = a.step[b];
        if (a.advance) a.step[b] += 1;
    }
    __syncthreads();
    const int tok = s_tok;
    int64_t pos = a.kv_len[b] - a.x_len[b];
    if (pos < 0) pos += a.n_pos;  // torch negative indexing of the PE table (idle slots only)
    if (pos < 0) pos = 0;
    if (pos >= a.n_pos) pos = a.n_pos - 1;
    for (int c = tid; c < kD; c += 256)
        a.xcur[(size_t)b * kD + c] = a.emb[(size_t)tok * kD + c] * 1.0f + a.pe[(size_t)pos * kD + c];
}

}  // namespace gsv
