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

// ---- shared prologue pieces ---------------------------------------------------------------

// two-pass LayerNorm of a 512-vector held as (v0 = elem tid, v1 = elem tid+256)
__device__ __forceinline__ void ln512(float& v0, float& v1, const float* __restrict__ g,
                                      const float* __restrict__ bta, float* red) {
    const int tid = threadIdx.x;
    float mean = block_sum<4>(v0 + v1, red) * (1.0f / kD);
    float d0 = v0 - mean, d1 = v1 - mean;
    float var = block_sum<4>(d0 * d0 + d1 * d1, red) * (1.0f / kD);
    float rs = 1.0f / sqrtf(var + kEps);
    v0 = d0 * rs * g[tid] + bta[tid];
    v1 = d1 * rs * g[tid + 256] + bta[tid + 256];
}

// wave-per-row GEMV rows: out[r] = dot(W[row0 + r][0:512], x) for NR rows handled by this wave,
// 8 rows in flight.  xr = this lane's 8 activations (x[lane*8 .. +7]).
template <typename WT, int NR, typename Fn>
__device__ __forceinline__ void wave_rows512(const WT* __restrict__ w, const float (&xr)[8], Fn&& emit) {
    const int lane = threadIdx.x & 63;
    static_assert(NR % 8 == 0, "NR");
#pragma unroll 1
    for (int r0 = 0; r0 < NR; r0 += 8) {
        float acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float wv[8];
            Ld<WT, 8>::load(w + (size_t)(r0 + u) * kD + lane * 8, wv);
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) a = fmaf(wv[i], xr[i], a);
            acc[u] = a;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = wave_sum(acc[u]);
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) emit(r0 + u, acc[u]);
        }
    }
}

// panel GEMV: out[row] = dot(P[row][0:K], v[0:K]) for 512 rows, K in {32, 64}; LPR lanes per row.
template <typename WT, int K>
__device__ __forceinline__ void panel_rows(const WT* __restrict__ panel, const float* __restrict__ vec_lds,
                                           float* __restrict__ out) {
    constexpr int EPL = 16 / sizeof(WT);
    constexpr int LPR = K / EPL;
    constexpr int RPI = 256 / LPR;
    const int tid = threadIdx.x;
    const int part = tid % LPR, rsub = tid / LPR;
    float vr[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) vr[i] = vec_lds[part * EPL + i];
#pragma unroll 8
    for (int it = 0; it < kD / RPI; ++it) {
        const int row = rsub + it * RPI;
        float wv[EPL];
        Ld<WT, EPL>::load(panel + (size_t)row * K + part * EPL, wv);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < EPL; ++i) s = fmaf(wv[i], vr[i], s);
#pragma unroll
        for (int m = LPR / 2; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        if (part == 0) out[row] = s;
    }
}

// ---- attention kernel ----------------------------------------------------------------------

template <typename WT>
struct AttnArgs {
    // layer input: mode 0 -> xdirect[B][512]; mode 1 -> LN2(sum_j zpart + b2 + x1) of the previous layer
    int mode;
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

template <typename WT>
__global__ __launch_bounds__(256) void t2s_attn_kernel(AttnArgs<WT> a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;          // 512
    float* qkv = xs + kD;      // 96
    float* att = qkv + 96;     // 32
    float* red = att + 32;     // 16
    float* pacc = red + 16;    // 4*32
    float* sc = pacc + 128;    // T
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

    float v0, v1;
    if (a.mode == 0) {
        v0 = a.xdirect[(size_t)b * kD + tid];
        v1 = a.xdirect[(size_t)b * kD + 256 + tid];
    } else {
        const float* zp = a.zpart + (size_t)b * kNJ * kD;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
        for (int j = 0; j < kNJ; ++j) {
            s0 += zp[j * kD + tid];
            s1 += zp[j * kD + 256 + tid];
        }
        v0 = s0 + a.b2[tid] + a.x1[(size_t)b * kD + tid];
        v1 = s1 + a.b2[tid + 256] + a.x1[(size_t)b * kD + 256 + tid];
        ln512(v0, v1, a.ln2g, a.ln2b, red);
    }
    xs[tid] = v0;
    xs[tid + 256] = v1;
    if (h == 0) {
        a.xout[(size_t)b * kD + tid] = v0;
        a.xout[(size_t)b * kD + 256 + tid] = v1;
    }
    __syncthreads();

    // q, k, v of this head: 96 rows, 24 per wave
    {
        float xr[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) xr[i] = xs[lane * 8 + i];
        const WT* wp = a.wqkv + ((size_t)h * 96 + wid * 24) * kD;
        const float* bp = a.bqkv + h * 96 + wid * 24;
        float* qo = qkv + wid * 24;
        wave_rows512<WT, 24>(wp, xr, [&](int r, float v) { qo[r] = v + bp[r]; });
    }
    __syncthreads();

    int n = (int)a.kv_len[b];
    if (n > a.T - 1) n = a.T - 1;  // memory safety only; the host never steps a full cache
    if (n < 0) n = 0;
    WT* Kp = a.kc + (((size_t)b * kH + h) * a.T) * kDh;
    WT* Vp = a.vc + (((size_t)b * kH + h) * a.T) * kDh;
    if (tid < 64) {
        // round through the cache type so this step sees exactly what later steps will read back
        WT s = from_f32<WT>(qkv[32 + tid]);
        qkv[32 + tid] = to_f32<WT>(s);
        if (tid < 32) Kp[(size_t)n * kDh + tid] = s; else Vp[(size_t)n * kDh + tid - 32] = s;
    }
    __syncthreads();

    constexpr int EPL = 16 / sizeof(WT);
    constexpr int LPR = kDh / EPL;
    constexpr int RPI = 256 / LPR;
    const int part = tid % LPR, rsub = tid / LPR;
    const float scale = 0.17677669529663687f;  // 1/sqrt(32)
    {
        float qr[EPL];
#pragma unroll
        for (int i = 0; i < EPL; ++i) qr[i] = qkv[part * EPL + i];
        for (int r = rsub; r <= n; r += RPI) {
            float kk[EPL];
            if (r == n) {
#pragma unroll
                for (int i = 0; i < EPL; ++i) kk[i] = qkv[32 + part * EPL + i];
            } else {
                Ld<WT, EPL>::load(Kp + (size_t)r * kDh + part * EPL, kk);
            }
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < EPL; ++i) s = fmaf(qr[i], kk[i], s);
#pragma unroll
            for (int m = LPR / 2; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
            if (part == 0) sc[r] = s * scale;
        }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int r = tid; r <= n; r += 256) mx = fmaxf(mx, sc[r]);
    mx = block_max<4>(mx, red);
    float sum = 0.f;
    for (int r = tid; r <= n; r += 256) {
        float e = expf(sc[r] - mx);
        sc[r] = e;
        sum += e;
    }
    sum = block_sum<4>(sum, red);  // (barriers inside also publish sc[])
    {
        float acc[EPL];
#pragma unroll
        for (int i = 0; i < EPL; ++i) acc[i] = 0.f;
        for (int r = rsub; r <= n; r += RPI) {
            float vv[EPL];
            if (r == n) {
#pragma unroll
                for (int i = 0; i < EPL; ++i) vv[i] = qkv[64 + part * EPL + i];
            } else {
                Ld<WT, EPL>::load(Vp + (size_t)r * kDh + part * EPL, vv);
            }
            const float p = sc[r] / sum;
#pragma unroll
            for (int i = 0; i < EPL; ++i) acc[i] = fmaf(p, vv[i], acc[i]);
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
    if (tid < 32) att[tid] = (pacc[tid] + pacc[32 + tid]) + (pacc[64 + tid] + pacc[96 + tid]);
    __syncthreads();
    panel_rows<WT, kDh>(a.wo + (size_t)h * kD * kDh, att, a.ypart + ((size_t)b * kH + h) * kD);
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
__global__ __launch_bounds__(256) void t2s_ffn_kernel(FfnArgs<WT> a) {
    __shared__ __attribute__((aligned(16))) float smem[kD + kFJ + 16];
    float* xs = smem;
    float* hb = xs + kD;
    float* red = hb + kFJ;
    const int j = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* yp = a.ypart + (size_t)b * kH * kD;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int h = 0; h < kH; ++h) {
        s0 += yp[h * kD + tid];
        s1 += yp[h * kD + 256 + tid];
    }
    float v0 = s0 + a.bo[tid] + a.x[(size_t)b * kD + tid];
    float v1 = s1 + a.bo[tid + 256] + a.x[(size_t)b * kD + 256 + tid];
    ln512(v0, v1, a.ln1g, a.ln1b, red);
    xs[tid] = v0;
    xs[tid + 256] = v1;
    if (j == 0) {
        a.x1out[(size_t)b * kD + tid] = v0;
        a.x1out[(size_t)b * kD + 256 + tid] = v1;
    }
    __syncthreads();
    {
        float xr[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) xr[i] = xs[lane * 8 + i];
        const int row0 = j * kFJ + wid * (kFJ / 4);
        const float* bp = a.b1 + row0;
        float* ho = hb + wid * (kFJ / 4);
        wave_rows512<WT, kFJ / 4>(a.w1 + (size_t)row0 * kD, xr, [&](int r, float v) { ho[r] = fmaxf(v + bp[r], 0.f); });
    }
    __syncthreads();
    panel_rows<WT, kFJ>(a.w2p + (size_t)j * kD * kFJ, hb, a.zpart + ((size_t)b * kNJ + j) * kD);
}

// ---- logits kernel -------------------------------------------------------------------------

template <typename WT>
struct LogitsArgs {
    // final hidden: mode 1 -> LN2(sum zpart + b2 + x1) of the last layer; mode 0 -> hdirect[B][512]
    int mode;
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

template <typename WT>
__global__ __launch_bounds__(256) void t2s_logits_kernel(LogitsArgs<WT> a) {
    __shared__ __attribute__((aligned(16))) float smem[kD + 16 + 128];
    float* xs = smem;
    float* red = xs + kD;
    float* lg = red + 16;  // up to 128 rows per slice
    const int p = blockIdx.x, r_ = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = a.slot0 + r_;  // state slot; row r_ of zpart/x1/hdirect
    float v0, v1;
    if (a.mode == 0) {
        v0 = a.hdirect[(size_t)r_ * kD + tid];
        v1 = a.hdirect[(size_t)r_ * kD + 256 + tid];
    } else {
        const float* zp = a.zpart + (size_t)r_ * kNJ * kD;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
        for (int j = 0; j < kNJ; ++j) {
            s0 += zp[j * kD + tid];
            s1 += zp[j * kD + 256 + tid];
        }
        v0 = s0 + a.b2[tid] + a.x1[(size_t)r_ * kD + tid];
        v1 = s1 + a.b2[tid + 256] + a.x1[(size_t)r_ * kD + 256 + tid];
        ln512(v0, v1, a.ln2g, a.ln2b, red);
    }
    xs[tid] = v0;
    xs[tid + 256] = v1;
    if (p == 0) {
        a.hidden[(size_t)b * kD + tid] = v0;
        a.hidden[(size_t)b * kD + 256 + tid] = v1;
    }
    __syncthreads();
    const int rpb = (a.V + kNP - 1) / kNP;  // rows per slice (<= 128)
    const int vbase = p * rpb;
    const int nrow = min(rpb, a.V - vbase);
    const bool sup = a.step[b] < a.ctl[1];
    const bool rep = a.ctl[2] != 0;
    const float rp = a.fctl[0];
    {
        float xr[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) xr[i] = xs[lane * 8 + i];
        for (int r0 = wid * 8; r0 < nrow; r0 += 32) {
            float acc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int v = min(vbase + r0 + u, a.V - 1);
                float wv[8];
                Ld<WT, 8>::load(a.wp + (size_t)v * kD + lane * 8, wv);
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) s = fmaf(wv[i], xr[i], s);
                acc[u] = s;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = wave_sum(acc[u]);
            if (lane == 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = r0 + u;
                    if (r < nrow) {
                        const int v = vbase + r;
                        float l = acc[u];
                        if (v >= a.vlimit) l = -INFINITY;
                        if (sup && (v == 280 || v == 486 || v == a.eos)) l = -INFINITY;
                        if (rep && a.seen[(size_t)b * a.V + v]) l = l < 0.f ? l * rp : l / rp;
                        lg[r] = l;
                        a.logits[(size_t)b * a.V + v] = l;
                    }
                }
            }
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
            int oi = __shfl_xor(bi, m, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
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
