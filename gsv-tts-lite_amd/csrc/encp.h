// enc_p (TextEncoder.infer, reference SoVITS/models.py:196-224; attentions.py:58-220; mrte_model.py:20-38)
// on gfx950: the dense layers run on tapgemm (channels-last activations), this file holds what is left --
// the attention with windowed relative positions, the channel LayerNorm and the two gathers.  bf16
// activations / operands, fp32 softmax statistics and accumulation (production mode).  The fp32 parity mode runs the
// same layer sequence on fp32 tapgemm with the plain fp32 kernels at the end of this file.
#pragma once
#include "tapgemm.h"

namespace gsv {

// rows of a table -> channels-last bf16 rows; `rep` consecutive output rows per index (x2 nearest upsampling
// of the codebook vectors, models.py:388-392)
static __global__ void encp_gather_kernel(const int64_t* __restrict__ idx, int n_idx, int n_rows_table, const float* __restrict__ table,
                                   int C, int rep, bf16_t* __restrict__ out) {
    const int r = blockIdx.x;                                // output row
    int id = (int)idx[r / rep];
    id = min(max(id, 0), n_rows_table - 1);
    for (int c = threadIdx.x * 2; c < C; c += blockDim.x * 2) {
        const float a = table[(size_t)id * C + c], b = table[(size_t)id * C + c + 1];
        *reinterpret_cast<uint32_t*>(out + (size_t)r * C + c) = pack_bf16x2(a, b);
    }
}

// modules.LayerNorm over the channel axis (attentions.py / modules.py:14-27): y[t] = LN(x[t]) * gamma + beta, one
// wave per row, C <= 512, two-pass like F.layer_norm
static __global__ __launch_bounds__(256) void encp_ln_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, bf16_t* __restrict__ y, int rows, int C) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float v[8];
    int n = 0;
    for (int c = lane; c < C; c += 64) v[n++] = bf16_to_f32(x[(size_t)row * C + c]);
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += v[i];
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int i = 0; i < n; ++i) { v[i] -= mean; q += v[i] * v[i]; }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)C + 1e-5f);
    n = 0;
    for (int c = lane; c < C; c += 64) { y[(size_t)row * C + c] = f32_to_bf16(v[n] * rs * gamma[c] + beta[c]); ++n; }
}

// y[t] = LN(sum_s P[s][t] + bias + res[t]) * gamma + beta : consumer of rowgemm's raw (split) fp32 tiles in the encoder
// layers (x = LN(x + attn_out), x = LN(x + ffn_out), attentions.py:88-101); y may alias res row-wise
static __global__ __launch_bounds__(256) void encp_ln_sum_kernel(const float* __restrict__ P, int nsplit, size_t split_stride,
                                                          const float* __restrict__ bias, const bf16_t* res,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          bf16_t* y, int rows, int C) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float v[8];
    int n = 0;
    for (int c = lane; c < C; c += 64) {
        float s = P[(size_t)row * C + c];
        for (int k = 1; k < nsplit; ++k) s += P[(size_t)k * split_stride + (size_t)row * C + c];
        v[n++] = s + bias[c] + bf16_to_f32(res[(size_t)row * C + c]);
    }
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += v[i];
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int i = 0; i < n; ++i) { v[i] -= mean; q += v[i] * v[i]; }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)C + 1e-5f);
    n = 0;
    for (int c = lane; c < C; c += 64) { y[(size_t)row * C + c] = f32_to_bf16(v[n] * rs * gamma[c] + beta[c]); ++n; }
}

// a + b (+ per-row or broadcast fp32 row g) -> bf16 : the MRTE sum  attn_out + ssl_enc + ge  (mrte_model.py:35-36)
// g row of frame r: r >> gshift (per-TOKEN conditioning rows serve both of a token's frames: the x2 nearest upsampling of ge,
// models.py:389, as an index instead of a copy)
static __global__ void encp_add3_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, const float* __restrict__ g, int ldg, int gshift,
                                 bf16_t* __restrict__ y, int rows, int C) {
    const size_t n = (size_t)rows * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / C;
        const int c = (int)(i % C);
        y[i] = f32_to_bf16(bf16_to_f32(a[i]) + bf16_to_f32(b[i]) + g[(ldg ? (r >> gshift) * (size_t)ldg : (size_t)0) + c]);
    }
}

struct EncAttnArgs {
    const bf16_t* Q; int ldq;      // [Tq][ldq], head h at column qoff + h*D
    const bf16_t* K; int ldk;      // [Tk][ldk]
    const bf16_t* V; int ldv;
    int qoff, koff, voff;
    bf16_t* O; int ldo;            // [Tq][ldo], head h at column h*D
    int Tq, Tk, H;
    float scale;                   // 1/sqrt(D) (the reference scales q before both products)
    const float* relk;             // [2w+1][D] fp32 or null (emb_rel_k[0])
    const float* relv;
    int window;
    const int64_t* slice;          // [Tq][2] or null: key j visible iff slice[i][0] <= j < slice[i][1] or j == Tk-1
    float* P;                      // [H][Tq][Tk] softmax probabilities (cross_attention.attn) or null
};

// One block = (head, 32 queries); its 4 waves split the 32-key tiles between them (tile kt belongs to wave
// kt % 4), each staging its own tiles in a wave-private LDS patch -- no block barrier inside the loops, and the
// dependent chain is a quarter of the key tiles.  Two passes:
//   A  S^T = K Q^T on the matrix cores (+ relative-position logits, mask) -> per-wave row max / row sum,
//      merged across the waves through LDS
//   B  S^T again, p = exp(s - m) / l -> optional P output, band probabilities for the relative-value term,
//      O^T += V^T P^T with V^T staged key-permuted so the D registers feed the B operand directly;
//      the waves' partial O^T are summed through LDS.
template <int D> constexpr int encp_attn_lds_bytes() {
    return 4 * (32 * (D * 2 + 16) + D * (32 * 2 + 16)) + 4 * 32 * 9 * 4 + 32 * 9 * 4 + 2 * 4 * 32 * 4;
}

template <int D>
__global__ __launch_bounds__(256) void encp_attn_kernel(EncAttnArgs a) {
    constexpr int KST = D / 16;                              // k-steps of the score product
    constexpr int MT = D / 32;                               // 32-row tiles of O^T
    constexpr int KRS = D * 2 + 16;                          // K tile row stride (bytes)
    constexpr int VRS = 32 * 2 + 16;                         // V^T tile row stride
    constexpr int WB = 32 * KRS + D * VRS;                   // wave-private staging bytes
    extern __shared__ __attribute__((aligned(16))) unsigned char alds[];
    const int h = blockIdx.x, q0 = blockIdx.y * 32;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = lane & 31, hf = lane >> 5;
    unsigned char* Ks = alds + wid * WB;
    unsigned char* Vt = Ks + 32 * KRS;
    float* rl = reinterpret_cast<float*>(alds + 4 * WB) + wid * 32 * 9;    // [32][9] per wave: q_i . rel_k[b] * scale
    float* pb = reinterpret_cast<float*>(alds + 4 * WB) + 4 * 32 * 9;      // [32][9] shared: p[i][i + b - w]
    float* mls = pb + 32 * 9;                                               // [2][4][32] per-wave (m, l)
    const int i = q0 + j;                                    // the lane's query
    const int ic = min(i, a.Tq - 1);
    const int w = a.window;
    const float LOG2E = 1.4426950408889634f;
    u32x4 qf[KST];
#pragma unroll
    for (int ks = 0; ks < KST; ++ks)
        qf[ks] = *reinterpret_cast<const u32x4*>(a.Q + (size_t)ic * a.ldq + a.qoff + h * D + ks * 16 + hf * 8);
    for (int e = tid; e < 32 * 9; e += 256) pb[e] = 0.f;
    int s0 = 0, s1 = a.Tk;
    if (a.slice) { s0 = (int)a.slice[(size_t)ic * 2]; s1 = (int)a.slice[(size_t)ic * 2 + 1]; }
    const int nkt = (a.Tk + 31) / 32;

    // 32 rows x D bf16 into the wave's K patch (rel: fp32 [2w+1][D] -> bf16 rows, zeros after)
    auto stage_k = [&](int row0, int nrows_valid, bool is_rel) {
        constexpr int NV = 32 * (D / 8) / 64;                // vectors per lane
        u32x4 v[NV];
#pragma unroll
        for (int x = 0; x < NV; ++x) {
            const int e = lane + x * 64, r = e / (D / 8), cv = e % (D / 8);
            v[x] = u32x4{0u, 0u, 0u, 0u};
            if (r < nrows_valid) {
                if (is_rel) {
                    const float* rp = a.relk + (size_t)r * D + cv * 8;
#pragma unroll
                    for (int y = 0; y < 4; ++y) v[x][y] = pack_bf16x2(rp[2 * y], rp[2 * y + 1]);
                } else {
                    v[x] = *reinterpret_cast<const u32x4*>(a.K + (size_t)(row0 + r) * a.ldk + a.koff + h * D + cv * 8);
                }
            }
        }
#pragma unroll
        for (int x = 0; x < NV; ++x) {
            const int e = lane + x * 64;
            *reinterpret_cast<u32x4*>(Ks + (e / (D / 8)) * KRS + (e % (D / 8)) * 16) = v[x];
        }
    };
    auto stage_v = [&](int row0, int nrows_valid) {          // V^T tile, keys permuted into D-register order
        constexpr int NV = 32 * (D / 8) / 64;
        u32x4 v[NV];
#pragma unroll
        for (int x = 0; x < NV; ++x) {
            const int e = lane + x * 64, r = e / (D / 8), cv = e % (D / 8);
            v[x] = u32x4{0u, 0u, 0u, 0u};
            if (r < nrows_valid) v[x] = *reinterpret_cast<const u32x4*>(a.V + (size_t)(row0 + r) * a.ldv + a.voff + h * D + cv * 8);
        }
#pragma unroll
        for (int x = 0; x < NV; ++x) {
            const int e = lane + x * 64, r = e / (D / 8), cv = e % (D / 8);
            const int hh = (r >> 2) & 1, blk = r >> 3;
            const int pos = (blk >> 1) * 16 + 8 * hh + (r & 3) + 4 * (blk & 1);
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                *reinterpret_cast<bf16_t*>(Vt + (cv * 8 + 2 * y) * VRS + pos * 2) = (bf16_t)(v[x][y] & 0xffff);
                *reinterpret_cast<bf16_t*>(Vt + (cv * 8 + 2 * y + 1) * VRS + pos * 2) = (bf16_t)(v[x][y] >> 16);
            }
        }
    };
    auto scores = [&](f32x16& s) {
#pragma unroll
        for (int q = 0; q < 16; ++q) s[q] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) {
            const u32x4 kf = *reinterpret_cast<const u32x4*>(Ks + j * KRS + ks * 32 + hf * 16);
            Mma<bf16_t>::run(s, kf, qf[ks]);
        }
    };

    // ---- relative-position logits: one extra "key tile" holding rel_k (every wave keeps its own copy)
    if (a.relk) {
        stage_k(0, 2 * w + 1, true);
        f32x16 s;
        scores(s);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int b = (q & 3) + 8 * (q >> 2) + 4 * hf;
            if (b < 2 * w + 1) rl[j * 9 + b] = s[q] * a.scale;
        }
    }
    // score of (query i, key) with mask and relative logits, in the exp2 domain; -1e30 = not visible
    auto finish = [&](float raw, int key) -> float {
        if (key >= a.Tk || i >= a.Tq) return -1e30f;
        if (a.slice && !((key >= s0 && key < s1) || key == a.Tk - 1)) return -1e30f;
        float v = raw * a.scale;
        const int b = key - i + w;
        if (a.relk && b >= 0 && b <= 2 * w) v += rl[j * 9 + b];
        return v * LOG2E;
    };
    // ---- pass A: row max and row sum over the wave's tiles, then across the waves
    float m = -1e30f, l = 0.f;
    for (int kt = wid; kt < nkt; kt += 4) {
        stage_k(kt * 32, min(32, a.Tk - kt * 32), false);
        f32x16 s;
        scores(s);
        float sv[16], tm = -1e30f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            sv[q] = finish(s[q], kt * 32 + (q & 3) + 8 * (q >> 2) + 4 * hf);
            tm = fmaxf(tm, sv[q]);
        }
        tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
        const float mn = fmaxf(m, tm);
        float ps = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) ps += sv[q] > -1e29f ? __builtin_amdgcn_exp2f(sv[q] - mn) : 0.f;
        l = l * __builtin_amdgcn_exp2f(m - mn) + ps;
        m = mn;
    }
    l += __shfl_xor(l, 32, 64);
    if (hf == 0) { mls[wid * 32 + j] = m; mls[128 + wid * 32 + j] = l; }
    __syncthreads();
    {
        float mg = -1e30f;
#pragma unroll
        for (int x = 0; x < 4; ++x) mg = fmaxf(mg, mls[x * 32 + j]);
        float lg = 0.f;
#pragma unroll
        for (int x = 0; x < 4; ++x) lg += mls[128 + x * 32 + j] * __builtin_amdgcn_exp2f(mls[x * 32 + j] - mg);
        m = mg; l = lg;
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    // ---- pass B: probabilities, P output, band terms, O^T += V^T P^T
    f32x16 o[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) o[t][q] = 0.f;
    for (int kt = wid; kt < nkt; kt += 4) {
        const int nv = min(32, a.Tk - kt * 32);
        stage_k(kt * 32, nv, false);
        stage_v(kt * 32, nv);
        f32x16 s;
        scores(s);
        float p[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int key = kt * 32 + (q & 3) + 8 * (q >> 2) + 4 * hf;
            const float sv = finish(s[q], key);
            p[q] = sv > -1e29f ? __builtin_amdgcn_exp2f(sv - m) * inv : 0.f;
            if (a.P && i < a.Tq && key < a.Tk) a.P[((size_t)h * a.Tq + i) * a.Tk + key] = p[q];
            const int b = key - i + w;
            if (a.relv && b >= 0 && b <= 2 * w && key < a.Tk) pb[j * 9 + b] = p[q];
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 pf;
#pragma unroll
            for (int e = 0; e < 4; ++e) pf[e] = pack_bf16x2(p[8 * ks + 2 * e], p[8 * ks + 2 * e + 1]);
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const u32x4 vf = *reinterpret_cast<const u32x4*>(Vt + (t * 32 + j) * VRS + (ks * 16 + hf * 8) * 2);
                Mma<bf16_t>::run(o[t], vf, pf);
            }
        }
    }
    // ---- sum the waves' partial O^T through LDS (the staging patches are free now), add the relative-value term
    __syncthreads();
    float* red = reinterpret_cast<float*>(alds);             // [4 waves][MT][16][64]
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) red[((wid * MT + t) * 16 + q) * 64 + lane] = o[t][q];
    __syncthreads();
    if (i >= a.Tq) return;
    for (int t = wid; t < MT; t += 4) {                      // wave w finishes tile t = w
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int q = 4 * g + e;
                v[e] = (red[((0 * MT + t) * 16 + q) * 64 + lane] + red[((1 * MT + t) * 16 + q) * 64 + lane]) +
                       (red[((2 * MT + t) * 16 + q) * 64 + lane] + red[((3 * MT + t) * 16 + q) * 64 + lane]);
            }
            const int d0 = t * 32 + 8 * g + 4 * hf;
            if (a.relv) {
                for (int b = 0; b <= 2 * w; ++b) {
                    const float pw = pb[j * 9 + b];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(pw, a.relv[(size_t)b * D + d0 + e], v[e]);
                }
            }
            uint2 pk;
            pk.x = pack_bf16x2(v[0], v[1]);
            pk.y = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<uint2*>(a.O + (size_t)i * a.ldo + h * D + d0) = pk;
        }
    }
}

// ---- fp32 parity mode -------------------------------------------------------------------------------------------
static __global__ void encp_gather_f32_kernel(const int64_t* __restrict__ idx, int n_rows_table, const float* __restrict__ table, int C,
                                              int rep, float* __restrict__ out) {
    const int r = blockIdx.x;
    int id = (int)idx[r / rep];
    id = min(max(id, 0), n_rows_table - 1);
    for (int c = threadIdx.x; c < C; c += blockDim.x) out[(size_t)r * C + c] = table[(size_t)id * C + c];
}

// y[t] = LN(x[t]) * gamma + beta in place, one wave per row, two-pass like F.layer_norm
static __global__ __launch_bounds__(256) void encp_ln_f32_kernel(float* x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 int rows, int C) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float v[8];
    int n = 0;
    for (int c = lane; c < C; c += 64) v[n++] = x[(size_t)row * C + c];
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += v[i];
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int i = 0; i < n; ++i) { v[i] -= mean; q += v[i] * v[i]; }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)C + 1e-5f);
    n = 0;
    for (int c = lane; c < C; c += 64) { x[(size_t)row * C + c] = v[n] * rs * gamma[c] + beta[c]; ++n; }
}

static __global__ void encp_add3_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ g, int ldg, int gshift,
                                            float* __restrict__ y, int rows, int C) {
    const size_t n = (size_t)rows * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / C;
        const int c = (int)(i % C);
        y[i] = a[i] + b[i] + g[(ldg ? (r >> gshift) * (size_t)ldg : (size_t)0) + c];
    }
}

struct EncAttnF32Args {
    const float *Q, *K, *V;        // [T][ld], head h at column off + h*D
    int ldq, ldk, ldv, qoff, koff, voff;
    float* O; int ldo;
    int Tq, Tk, D;
    float rsqrt_d;                 // the reference divides q by sqrt(D) before both products (attentions.py:152)
    const float *relk, *relv;      // [2w+1][D] or null
    int window;
    const int64_t* slice;          // as EncAttnArgs
    float* P;                      // [H][Tq][Tk] or null
};

// one wave per (head, query): scores over the keys (a lane per key), softmax, then a lane per output channel.
// The masked score is -1e4, not -inf (attentions.py:163).  Dynamic LDS: 4 waves x (Tk + D) floats.
static __global__ __launch_bounds__(256) void encp_attn_f32_kernel(EncAttnF32Args a) {
    extern __shared__ float encp_f32_lds[];
    const int h = blockIdx.x, wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.y * 4 + wid;
    if (i >= a.Tq) return;
    float* sc = encp_f32_lds + (size_t)wid * (a.Tk + a.D);
    float* qs = sc + a.Tk;
    const float sq = sqrtf((float)a.D);
    for (int d = lane; d < a.D; d += 64) qs[d] = a.Q[(size_t)i * a.ldq + a.qoff + h * a.D + d] / sq;
    __builtin_amdgcn_wave_barrier();
    int lo = 0, hi = a.Tk;
    if (a.slice) { lo = (int)a.slice[2 * i]; hi = (int)a.slice[2 * i + 1]; }
    float mx = -INFINITY;
    for (int j = lane; j < a.Tk; j += 64) {
        const float* kr = a.K + (size_t)j * a.ldk + a.koff + h * a.D;
        float s = 0.f;
        for (int d = 0; d < a.D; ++d) s += qs[d] * kr[d];
        if (a.relk && abs(j - i) <= a.window) {
            const float* rk = a.relk + (size_t)(j - i + a.window) * a.D;
            float r = 0.f;
            for (int d = 0; d < a.D; ++d) r += qs[d] * rk[d];
            s += r;
        }
        if (a.slice && !((j >= lo && j < hi) || j == a.Tk - 1)) s = -1e4f;
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float l = 0.f;
    for (int j = lane; j < a.Tk; j += 64) { const float e = expf(sc[j] - mx); sc[j] = e; l += e; }
    l = wave_sum(l);
    const float inv = 1.0f / l;
    for (int j = lane; j < a.Tk; j += 64) {
        const float pj = sc[j] * inv;
        sc[j] = pj;
        if (a.P) a.P[((size_t)h * a.Tq + i) * a.Tk + j] = pj;
    }
    __builtin_amdgcn_wave_barrier();
    for (int d = lane; d < a.D; d += 64) {
        float o = 0.f;
        for (int j = 0; j < a.Tk; ++j) o += sc[j] * a.V[(size_t)j * a.ldv + a.voff + h * a.D + d];
        if (a.relv) {
            float r = 0.f;
            for (int j = max(0, i - a.window); j <= min(a.Tk - 1, i + a.window); ++j) r += sc[j] * a.relv[(size_t)(j - i + a.window) * a.D + d];
            o += r;
        }
        a.O[(size_t)i * a.ldo + h * a.D + d] = o;
    }
}

}  // namespace gsv
