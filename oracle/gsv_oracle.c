/*
 * TEST INFRASTRUCTURE -- CPU oracle for the GPT-SoVITS inference hot path.
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke()
 * and bench.py's `cpu_baseline` leg may load it (see DESIGN.md "Oracle").  It is a
 * plain-C, fp32 restatement of the arithmetic the reference performs through
 * PyTorch on CPU, written from the reference's model definitions:
 *
 *   GPT block (post-LN, ReLU MLP)      gsv_tts/GPT_SoVITS/GPT/t2s_model.py:31-105
 *   decode KV append + causal attention  t2s_model.py:80-92  (mask == positions [0, kv_len])
 *   prefill with explicit bool mask      t2s_model.py:42-52, 365-381
 *   WaveNet flow (reverse)               SoVITS/module/modules.py:80-104, 482-511; models.py:58-65
 *   HiFiGAN-style Generator              SoVITS/models.py:113-132; modules.py:190-203
 *
 * Parity status: pinned against the imported reference in the build container by
 * oracle/gen_golden.py -> tests/golden/*.npz (the reference ships no golden vectors of
 * its own).  Summation order differs from torch's oneDNN/MKL
 * kernels, so agreement is to fp32 round-off (tolerances live in tests/), while greedy
 * token ids agree exactly wherever the recorded top-1/top-2 margin exceeds that noise.
 *
 * Layouts: activations row-major [rows][features]; conv tensors channels-first
 * [C][T] exactly as torch; KV cache [layer][B][H][T][Dh] as t2s_model.py:269-270.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

ORC_API int orc_version(void) { return 1; }

ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

ORC_API void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------ reduced-precision modes
 * The product's production modes store weights and K/V in bf16 (and, GSV_FP8, the QKV / FFN operands in OCP
 * e4m3) with fp32 accumulation.  To pin THOSE kernels tightly -- same operands, only the summation order
 * differs -- the restatement can round at the same places (weights are rounded by the caller):
 *   ORC_R_KV      K/V rows rounded through bf16 when written to the cache (what later steps read back)
 *   ORC_R_LIN     the input rows of every linear rounded to bf16 (MFMA operand type of the prompt / batched GEMMs; since round 5
 *                 also the per-sequence decode kernels, whose dots take ONE bf16 value per activation: with it the sliced linears
 *                 of ORC_R_PART round their input rows too, and the decode step's query enters its score dots as bf16)
 *   ORC_R_ATTN    prompt attention: q and the un-normalised probabilities rounded to bf16 (MFMA flash attention)
 *   ORC_R_FP8     qkv / mlp.0 / mlp.2 inputs rounded to e4m3 at unit scale, saturating (batched fp8 step)
 *   ORC_R_VOC     flow + Generator: every stored activation rounded to bf16 where the bf16 HIP path stores bf16
 *   ORC_R_PART    decode step of the partial-sum kernels (< batched_min sequences): the out-proj is summed from its 16
 *                 per-head partial vectors and W2 from its 32 per-slice (64 hidden units) partials, each rounded to IEEE half --
 *                 the form in which they cross the kernel boundary on bf16 handles (csrc/t2s_decode.h PartOf)
 *   ORC_R_FINE    with ORC_R_PART: W2 from 64 slices of 32 hidden units (the library's choice at <= 4 sequences,
 *                 gsv_t2s_ffn_slices)
 *   ORC_R_FFN32   with ORC_R_LIN: mlp.0 / mlp.2 keep fp32 input rows (the two-sequences-per-block FFN kernel of 9..16 sequences)
 *                 (conv outputs after bias / conditioning / residual; the leaky-ReLU'd conv operands; the branch mean;
 *                 the flow's h, gate output, skip operand and updated half), fp32 accumulation inside each op
 * 0 = the fp32 reference arithmetic. */
#define ORC_R_KV 1
#define ORC_R_LIN 2
#define ORC_R_ATTN 4
#define ORC_R_FP8 8
#define ORC_R_VOC 16
#define ORC_R_PART 32
#define ORC_R_FINE 64
#define ORC_R_FFN32 128   /* with ORC_R_LIN: the FFN's two linears take their input rows in fp32 (t2s_ffn_multi_kernel, two sequences per block:
                           * 9..16 sequences; its dots are fp32 FMA chains on unpacked weights, csrc/t2s_decode_multi.h) */
static int g_round = 0;
ORC_API void orc_set_rounding(int flags) { g_round = flags; }
ORC_API int orc_get_rounding(void) { return g_round; }

static inline float bf16r(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return f;
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&f, &u, 4);
    return f;
}

/* OCP e4m3fn: 3 mantissa bits, exponent bias 7, max 448, subnormal step 2^-9; round to nearest even, saturating */
static inline float e4m3r(float f) {
    if (f != f) return f;
    float a = fabsf(f);
    if (a > 448.f) a = 448.f;
    if (a < 0.015625f) {
        a = rintf(a * 512.f) / 512.f;
    } else {
        int e;
        (void)frexpf(a, &e);
        float step = ldexpf(1.f, e - 4);
        a = rintf(a / step) * step;
        if (a > 448.f) a = 448.f;
    }
    return copysignf(a, f);
}
/* round-to-nearest-even through IEEE binary16 (normal and subnormal range; overflow saturates to +-inf as v_cvt_f16_f32 does) */
static float f16r(float f) {
    union { float f; uint32_t u; } v;
    v.f = f;
    const uint32_t sign = v.u & 0x80000000u;
    v.u &= 0x7fffffffu;
    if (v.u >= 0x7f800000u) return f;                       /* inf / nan */
    if (v.f >= 65520.0f) { v.u = sign | 0x7f800000u; return v.f; }
    if (v.f < 6.103515625e-05f) {                            /* subnormal half: multiples of 2^-24 */
        const float q = rintf(v.f * 16777216.0f);            /* rintf: round-half-even in the default mode */
        v.f = q * 5.9604644775390625e-08f;
        v.u |= sign;
        return v.f;
    }
    uint32_t u = v.u;
    u += 0xfffu + ((u >> 13) & 1u);                           /* keep 10 mantissa bits, ties to even */
    u &= ~0x1fffu;
    v.u = u | sign;
    return v.f;
}
ORC_API void orc_round_f16(float* x, long n) { for (long i = 0; i < n; ++i) x[i] = f16r(x[i]); }

ORC_API void orc_round_bf16(float* x, long n) { for (long i = 0; i < n; ++i) x[i] = bf16r(x[i]); }
static void voc_round(float* x, size_t n) {
    if (!(g_round & ORC_R_VOC)) return;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) x[i] = bf16r(x[i]);
}
ORC_API void orc_round_e4m3(float* x, long n) { for (long i = 0; i < n; ++i) x[i] = e4m3r(x[i]); }

/* ------------------------------------------------------------------ basic ops */

static inline float dotf(const float* a, const float* b, int n) {
    float acc = 0.f;
#pragma omp simd reduction(+ : acc)
    for (int i = 0; i < n; ++i) acc += a[i] * b[i];
    return acc;
}

/* y[m][n] = sum_k x[m][k] * w[n][k] + b[n]   (torch nn.Linear) ; act: 0 none, 1 relu */
ORC_API void orc_linear(const float* x, int M, int K, const float* w, const float* b, int N,
                        float* y, int act) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            float v = dotf(x + (size_t)m * K, w + (size_t)n * K, K) + (b ? b[n] : 0.f);
            if (act == 1 && v < 0.f) v = 0.f;
            y[(size_t)m * N + n] = v;
        }
}

/* torch nn.LayerNorm over the last dim, biased variance, eps inside sqrt */
ORC_API void orc_layernorm(const float* x, int M, int D, const float* g, const float* b, float eps,
                           float* y) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        const float* r = x + (size_t)m * D;
        float mean = 0.f;
        for (int i = 0; i < D; ++i) mean += r[i];
        mean /= (float)D;
        float var = 0.f;
        for (int i = 0; i < D; ++i) {
            float d = r[i] - mean;
            var += d * d;
        }
        var /= (float)D;
        float rs = 1.0f / sqrtf(var + eps);
        float* o = y + (size_t)m * D;
        for (int i = 0; i < D; ++i) o[i] = (r[i] - mean) * rs * g[i] + b[i];
    }
}

/* ------------------------------------------------------------------ GPT block
 * Per-layer parameter pack (floats, in this order), D = hidden, F = 4*D:
 *   qkv_w[3D][D] qkv_b[3D] out_w[D][D] out_b[D] ln1_g[D] ln1_b[D]
 *   w1[F][D] b1[F] w2[D][F] b2[D] ln2_g[D] ln2_b[D]
 */
typedef struct {
    const float *qkv_w, *qkv_b, *out_w, *out_b, *ln1_g, *ln1_b, *w1, *b1, *w2, *b2, *ln2_g, *ln2_b;
} layer_t;

ORC_API long orc_layer_floats(int D) {
    long F = 4L * D;
    return 3L * D * D + 3L * D + (long)D * D + D + 2L * D + F * D + F + (long)D * F + D + 2L * D;
}

static layer_t layer_at(const float* pack, int l, int D) {
    const float* p = pack + (size_t)l * orc_layer_floats(D);
    long F = 4L * D;
    layer_t L;
    L.qkv_w = p; p += 3L * D * D;
    L.qkv_b = p; p += 3L * D;
    L.out_w = p; p += (long)D * D;
    L.out_b = p; p += D;
    L.ln1_g = p; p += D;
    L.ln1_b = p; p += D;
    L.w1 = p; p += F * D;
    L.b1 = p; p += F;
    L.w2 = p; p += (long)D * F;
    L.b2 = p; p += D;
    L.ln2_g = p; p += D;
    L.ln2_b = p;
    return L;
}

/* a linear of the GPT stack whose input rows are rounded per g_round (f8ok: one of the fp8 linears) */
static void linear_r(const float* x, int M, int K, const float* w, const float* b, int N, float* y, int act, int f8ok) {
    const int f8 = f8ok && (g_round & ORC_R_FP8);
    if (!f8 && !(g_round & ORC_R_LIN)) {
        orc_linear(x, M, K, w, b, N, y, act);
        return;
    }
    float* xr = (float*)malloc(sizeof(float) * (size_t)M * K);
    for (size_t i = 0; i < (size_t)M * K; ++i) xr[i] = f8 ? e4m3r(x[i]) : bf16r(x[i]);
    orc_linear(xr, M, K, w, b, N, y, act);
    free(xr);
}

/* tail of a block shared by prefill and decode: x = LN1(x + attn@Wo^T + bo); x = LN2(x + MLP(x)) */
/* y[m][n] = sum over slices s of f16r( sum_{k in slice s} x[m][k] w[n][k] ) + b[n]: the partial-sum kernels' kernel-boundary form */
static void linear_sliced(const float* x_in, int M, int K, const float* w, const float* b, int N, int slice, float* y) {
    float* xr = NULL;
    const float* x = x_in;
    if (g_round & ORC_R_LIN) {      /* the kernels' dots take bf16 activations */
        xr = (float*)malloc(sizeof(float) * (size_t)M * K);
        for (size_t i = 0; i < (size_t)M * K; ++i) xr[i] = bf16r(x_in[i]);
        x = xr;
    }
#pragma omp parallel for collapse(2) schedule(static)
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            float acc = 0.f;
            for (int k0 = 0; k0 < K; k0 += slice) acc += f16r(dotf(x + (size_t)m * K + k0, w + (size_t)n * K + k0, slice));
            y[(size_t)m * N + n] = acc + b[n];
        }
    free(xr);
}

static void block_tail(const layer_t* L, int M, int D, int H, int part, float* x, const float* attn, float* tmp_d,
                       float* tmp_f) {
    int F = 4 * D;
    if (part) linear_sliced(attn, M, D, L->out_w, L->out_b, D, D / H, tmp_d);       /* 16 head partials */
    else linear_r(attn, M, D, L->out_w, L->out_b, D, tmp_d, 0, 0);
    for (size_t i = 0; i < (size_t)M * D; ++i) tmp_d[i] += x[i];
    orc_layernorm(tmp_d, M, D, L->ln1_g, L->ln1_b, 1e-5f, x);
    const int keep = g_round;
    if (g_round & ORC_R_FFN32) g_round &= ~ORC_R_LIN;      /* the FFN of this path multiplies fp32 activations */
    linear_r(x, M, D, L->w1, L->b1, F, tmp_f, 1, 1);
    if (part) linear_sliced(tmp_f, M, F, L->w2, L->b2, D, (g_round & ORC_R_FINE) ? 32 : 64, tmp_d);   /* 32 slices of 64 hidden units, or 64 of 32 */
    else linear_r(tmp_f, M, F, L->w2, L->b2, D, tmp_d, 0, 1);
    g_round = keep;
    for (size_t i = 0; i < (size_t)M * D; ++i) tmp_d[i] += x[i];
    orc_layernorm(tmp_d, M, D, L->ln2_g, L->ln2_b, 1e-5f, x);
}

static inline void kv_store(float* dst, const float* src, int n) {
    if (g_round & ORC_R_KV) for (int i = 0; i < n; ++i) dst[i] = bf16r(src[i]);
    else memcpy(dst, src, sizeof(float) * n);
}

/*
 * One decode step through all layers for B rows (t2s_model.py:67-105, 129-143).
 *   x      [B][D]   in: token input, out: final hidden
 *   kc, vc [n_layer][Bc][H][T][Dh] caches; row b of x uses cache row b0+b
 *   kv_len [B]      entries already in the cache; the new K/V land at kv_len[b] and the
 *                   query attends to positions [0, kv_len[b]]  (the caller bumps kv_len)
 */
ORC_API void orc_t2s_decode(const float* pack, int n_layer, int D, int H, int B, float* x, float* kc,
                            float* vc, int Bc, int T, int b0, const int64_t* kv_len) {
    int Dh = D / H;
    float scale = 1.0f / sqrtf((float)Dh);
    float* qkv = (float*)malloc(sizeof(float) * (size_t)B * 3 * D);
    float* attn = (float*)malloc(sizeof(float) * (size_t)B * D);
    float* tmp_d = (float*)malloc(sizeof(float) * (size_t)B * D);
    float* tmp_f = (float*)malloc(sizeof(float) * (size_t)B * 4 * D);
    for (int l = 0; l < n_layer; ++l) {
        layer_t L = layer_at(pack, l, D);
        linear_r(x, B, D, L.qkv_w, L.qkv_b, 3 * D, qkv, 0, 1);
#pragma omp parallel for collapse(2) schedule(static)
        for (int b = 0; b < B; ++b)
            for (int h = 0; h < H; ++h) {
                int n = (int)kv_len[b];
                size_t base = ((((size_t)l * Bc + (b0 + b)) * H + h) * (size_t)T) * Dh;
                float* K = kc + base;
                float* V = vc + base;
                float qb[256];
                const float* q = qkv + (size_t)b * 3 * D + h * Dh;
                if ((g_round & ORC_R_LIN) && Dh <= 256) {      /* the query is a bf16 operand of the score dots */
                    for (int d = 0; d < Dh; ++d) qb[d] = bf16r(q[d]);
                    q = qb;
                }
                kv_store(K + (size_t)n * Dh, qkv + (size_t)b * 3 * D + D + h * Dh, Dh);
                kv_store(V + (size_t)n * Dh, qkv + (size_t)b * 3 * D + 2 * D + h * Dh, Dh);
                int len = n + 1;
                float* s = (float*)malloc(sizeof(float) * len);
                float mx = -INFINITY;
                for (int t = 0; t < len; ++t) {
                    s[t] = dotf(q, K + (size_t)t * Dh, Dh) * scale;
                    if (s[t] > mx) mx = s[t];
                }
                float den = 0.f;
                for (int t = 0; t < len; ++t) {
                    s[t] = expf(s[t] - mx);
                    den += s[t];
                }
                float* o = attn + (size_t)b * D + h * Dh;
                for (int d = 0; d < Dh; ++d) o[d] = 0.f;
                for (int t = 0; t < len; ++t) {
                    float p = s[t] / den;
                    const float* vr = V + (size_t)t * Dh;
                    for (int d = 0; d < Dh; ++d) o[d] += p * vr[d];
                }
                free(s);
            }
        block_tail(&L, B, D, H, (g_round & ORC_R_PART) != 0, x, attn, tmp_d, tmp_f);
    }
    free(qkv); free(attn); free(tmp_d); free(tmp_f);
}

/*
 * Prefill of B packed rows of length Lq through all layers (t2s_model.py:31-65,114-127).
 *   x    [B][Lq][D] in/out
 *   mask [B][Lq][Lq] bytes, nonzero = may attend (the reference's bool mask, one head's worth;
 *        it is identical across heads, t2s_model.py:347,379).  A fully-masked query row yields
 *        zeros (torch>=2.5 SDPA behaviour).
 * Writes K/V for positions [0, Lq) of cache rows b0..b0+B-1.
 */
ORC_API void orc_t2s_prefill(const float* pack, int n_layer, int D, int H, int B, int Lq, float* x,
                             const uint8_t* mask, float* kc, float* vc, int Bc, int T, int b0) {
    int Dh = D / H;
    float scale = 1.0f / sqrtf((float)Dh);
    size_t M = (size_t)B * Lq;
    float* qkv = (float*)malloc(sizeof(float) * M * 3 * D);
    float* attn = (float*)malloc(sizeof(float) * M * D);
    float* tmp_d = (float*)malloc(sizeof(float) * M * D);
    float* tmp_f = (float*)malloc(sizeof(float) * M * 4 * D);
    for (int l = 0; l < n_layer; ++l) {
        layer_t L = layer_at(pack, l, D);
        linear_r(x, (int)M, D, L.qkv_w, L.qkv_b, 3 * D, qkv, 0, 0);
#pragma omp parallel for collapse(2) schedule(static)
        for (int b = 0; b < B; ++b)
            for (int h = 0; h < H; ++h) {
                size_t base = ((((size_t)l * Bc + (b0 + b)) * H + h) * (size_t)T) * Dh;
                for (int t = 0; t < Lq; ++t) {
                    const float* r = qkv + ((size_t)b * Lq + t) * 3 * D;
                    kv_store(kc + base + (size_t)t * Dh, r + D + h * Dh, Dh);
                    kv_store(vc + base + (size_t)t * Dh, r + 2 * D + h * Dh, Dh);
                }
                float* s = (float*)malloc(sizeof(float) * Lq);
                for (int i = 0; i < Lq; ++i) {
                    const float* q = qkv + ((size_t)b * Lq + i) * 3 * D + h * Dh;
                    float qr[256];
                    if ((g_round & ORC_R_ATTN) && Dh <= 256) {
                        for (int d = 0; d < Dh; ++d) qr[d] = bf16r(q[d]);
                        q = qr;
                    }
                    const uint8_t* mr = mask + ((size_t)b * Lq + i) * Lq;
                    float mx = -INFINITY;
                    for (int t = 0; t < Lq; ++t) {
                        if (mr[t]) {
                            s[t] = dotf(q, kc + base + (size_t)t * Dh, Dh) * scale;
                            if (s[t] > mx) mx = s[t];
                        }
                    }
                    float* o = attn + ((size_t)b * Lq + i) * D + h * Dh;
                    for (int d = 0; d < Dh; ++d) o[d] = 0.f;
                    if (mx == -INFINITY) continue;
                    float den = 0.f;
                    for (int t = 0; t < Lq; ++t)
                        if (mr[t]) {
                            s[t] = expf(s[t] - mx);
                            den += s[t];
                        }
                    for (int t = 0; t < Lq; ++t)
                        if (mr[t]) {
                            float p = ((g_round & ORC_R_ATTN) ? bf16r(s[t]) : s[t]) / den;
                            const float* vr = vc + base + (size_t)t * Dh;
                            for (int d = 0; d < Dh; ++d) o[d] += p * vr[d];
                        }
                }
                free(s);
            }
        block_tail(&L, (int)M, D, H, 0, x, attn, tmp_d, tmp_f);      /* the prompt pass is the GEMM chain: no partial rows */
    }
    free(qkv); free(attn); free(tmp_d); free(tmp_f);
}

/* ------------------------------------------------------------------ conv primitives (channels-first) */

/* torch F.conv1d, stride 1, zero padding pad on both sides, dilation dil. out length == T
 * when pad == dil*(k-1)/2.  in_slope: leaky-relu slope applied to the input first (1.0 = none). */
ORC_API void orc_conv1d(const float* x, int Cin, int T, const float* w, const float* b, int Cout, int k,
                        int dil, int pad, float in_slope, float* y) {
    float* xin = (float*)x;
    float* act = NULL;
    if (in_slope != 1.0f) {
        act = (float*)malloc(sizeof(float) * (size_t)Cin * T);
#pragma omp parallel for schedule(static)
        for (int c = 0; c < Cin; ++c)
            for (int t = 0; t < T; ++t) {
                float v = x[(size_t)c * T + t];
                v = v >= 0.f ? v : v * in_slope;
                act[(size_t)c * T + t] = (g_round & ORC_R_VOC) ? bf16r(v) : v;   /* the MFMA operand is bf16(lrelu(x)) */
            }
        xin = act;
    }
    const int TB = 1024;
    int nco = (Cout + 3) / 4, ntb = (T + TB - 1) / TB;
#pragma omp parallel for collapse(2) schedule(static)
    for (int cb = 0; cb < nco; ++cb)
        for (int tb = 0; tb < ntb; ++tb) {
            int t0 = tb * TB, t1 = t0 + TB < T ? t0 + TB : T;
            int c0 = cb * 4, nc = Cout - c0 < 4 ? Cout - c0 : 4;
            float acc[4][1024];
            for (int j = 0; j < 4; ++j) {
                float bv = (b && j < nc) ? b[c0 + j] : 0.f;
                for (int t = 0; t < t1 - t0; ++t) acc[j][t] = bv;
            }
            for (int ci = 0; ci < Cin; ++ci) {
                const float* xr = xin + (size_t)ci * T;
                for (int kk = 0; kk < k; ++kk) {
                    int off = kk * dil - pad;
                    int lo = t0 + off < 0 ? -off : t0;          /* first t with t+off >= 0 */
                    int hi = t1 + off > T ? T - off : t1;        /* last t (excl) with t+off < T */
                    if (lo >= hi) continue;
                    float w0 = w[((size_t)(c0 + 0) * Cin + ci) * k + kk];
                    float w1 = nc > 1 ? w[((size_t)(c0 + 1) * Cin + ci) * k + kk] : 0.f;
                    float w2 = nc > 2 ? w[((size_t)(c0 + 2) * Cin + ci) * k + kk] : 0.f;
                    float w3 = nc > 3 ? w[((size_t)(c0 + 3) * Cin + ci) * k + kk] : 0.f;
                    const float* xs = xr + off;
#pragma omp simd
                    for (int t = lo; t < hi; ++t) {
                        float xv = xs[t];
                        acc[0][t - t0] += w0 * xv;
                        acc[1][t - t0] += w1 * xv;
                        acc[2][t - t0] += w2 * xv;
                        acc[3][t - t0] += w3 * xv;
                    }
                }
            }
            for (int j = 0; j < nc; ++j)
                memcpy(y + (size_t)(c0 + j) * T + t0, acc[j], sizeof(float) * (t1 - t0));
        }
    free(act);
}

/* torch F.conv_transpose1d, weight [Cin][Cout][k], stride u, padding pad; Tout = (T-1)*u - 2*pad + k.
 * in_slope as above. */
ORC_API void orc_conv_transpose1d(const float* x, int Cin, int T, const float* w, const float* b,
                                  int Cout, int k, int u, int pad, float in_slope, float* y) {
    int Tout = (T - 1) * u - 2 * pad + k;
#pragma omp parallel for schedule(static)
    for (int co = 0; co < Cout; ++co) {
        float* yr = y + (size_t)co * Tout;
        float bv = b ? b[co] : 0.f;
        for (int n = 0; n < Tout; ++n) yr[n] = bv;
        for (int ci = 0; ci < Cin; ++ci) {
            const float* xr = x + (size_t)ci * T;
            const float* wr = w + ((size_t)ci * Cout + co) * k;
            for (int i = 0; i < T; ++i) {
                float xv = xr[i];
                if (in_slope != 1.0f && xv < 0.f) { xv *= in_slope; if (g_round & ORC_R_VOC) xv = bf16r(xv); }
                int n0 = i * u - pad;
                for (int kk = 0; kk < k; ++kk) {
                    int n = n0 + kk;
                    if (n >= 0 && n < Tout) yr[n] += xv * wr[kk];
                }
            }
        }
    }
}

/* ------------------------------------------------------------------ Generator (models.py:113-132)
 * Parameter pack order (floats): conv_pre.w[C0][Cz][7] conv_pre.b[C0] cond.w[C0][gin] cond.b[C0]
 *   for stage i: ups.w[Cin][Cout][k_i] ups.b[Cout]
 *                for j in 0..2 (resblock kernel rk_j): for d in 0..2: convs1[d].w[C][C][rk] convs1[d].b[C]
 *                                                      then for d in 0..2: convs2[d].w convs2[d].b
 *   conv_post.w[1][C5][7]
 * g: [gin][Tg] with Tg == 1 (broadcast) or Tg == T (per-frame cond, batched path TTS.py:740-744).
 */
ORC_API void orc_generator(const float* pack, int Cz, int C0, int gin, int n_up, const int* up_rates,
                           const int* up_kernels, int n_rk, const int* rk, const int* rdil /*[3]*/,
                           const float* z /*[Cz][T]*/, int T, const float* g, int Tg,
                           float* out /*[T*prod(up)]*/) {
    const float* p = pack;
    const float* pre_w = p; p += (size_t)C0 * Cz * 7;
    const float* pre_b = p; p += C0;
    const float* cond_w = p; p += (size_t)C0 * gin;
    const float* cond_b = p; p += C0;
    float* x = (float*)malloc(sizeof(float) * (size_t)C0 * T);
    float* zr = NULL;
    float* gr = NULL;
    if (g_round & ORC_R_VOC) {   /* the ABI converts z and ge to the activation type */
        zr = (float*)malloc(sizeof(float) * (size_t)Cz * T); memcpy(zr, z, sizeof(float) * (size_t)Cz * T); voc_round(zr, (size_t)Cz * T); z = zr;
        gr = (float*)malloc(sizeof(float) * (size_t)gin * Tg); memcpy(gr, g, sizeof(float) * (size_t)gin * Tg); voc_round(gr, (size_t)gin * Tg); g = gr;
    }
    orc_conv1d(z, Cz, T, pre_w, pre_b, C0, 7, 1, 3, 1.0f, x);
    {
        float* c = (float*)malloc(sizeof(float) * (size_t)C0 * Tg);
        orc_conv1d(g, gin, Tg, cond_w, cond_b, C0, 1, 1, 0, 1.0f, c);
        for (int co = 0; co < C0; ++co)
            for (int t = 0; t < T; ++t) x[(size_t)co * T + t] += c[(size_t)co * Tg + (Tg == 1 ? 0 : t)];
        free(c);
    }
    voc_round(x, (size_t)C0 * T);
    free(zr); free(gr);
    int C = C0, Tc = T;
    for (int i = 0; i < n_up; ++i) {
        int u = up_rates[i], k = up_kernels[i], Co = C / 2;
        const float* uw = p; p += (size_t)C * Co * k;
        const float* ub = p; p += Co;
        int Tn = Tc * u;
        float* y = (float*)malloc(sizeof(float) * (size_t)Co * Tn);
        orc_conv_transpose1d(x, C, Tc, uw, ub, Co, k, u, (k - u) / 2, 0.1f, y);
        free(x);
        C = Co; Tc = Tn;
        size_t n = (size_t)C * Tc;
        voc_round(y, n);
        float* xs = (float*)calloc(n, sizeof(float));
        float* xr = (float*)malloc(sizeof(float) * n);
        float* t1 = (float*)malloc(sizeof(float) * n);
        float* t2 = (float*)malloc(sizeof(float) * n);
        for (int j = 0; j < n_rk; ++j) {
            int kk = rk[j];
            const float* w1[3]; const float* b1[3]; const float* w2[3]; const float* b2[3];
            for (int d = 0; d < 3; ++d) { w1[d] = p; p += (size_t)C * C * kk; b1[d] = p; p += C; }
            for (int d = 0; d < 3; ++d) { w2[d] = p; p += (size_t)C * C * kk; b2[d] = p; p += C; }
            memcpy(xr, y, sizeof(float) * n);
            for (int d = 0; d < 3; ++d) {
                orc_conv1d(xr, C, Tc, w1[d], b1[d], C, kk, rdil[d], rdil[d] * (kk - 1) / 2, 0.1f, t1);
                /* bf16 path: the first conv stores lrelu(t1) (one rounding), which orc_conv1d's own lrelu + rounding of
                 * the un-rounded t1 reproduces */
                orc_conv1d(t1, C, Tc, w2[d], b2[d], C, kk, 1, (kk - 1) / 2, 0.1f, t2);
                for (size_t e = 0; e < n; ++e) xr[e] = t2[e] + xr[e];
                voc_round(xr, n);
            }
            if (g_round & ORC_R_VOC) {   /* the branch mean is ((a + b) + c) / 3 in fp32 from the stored tensors */
                if (j == 0) memcpy(xs, xr, sizeof(float) * n);
                else for (size_t e = 0; e < n; ++e) xs[e] = xs[e] + xr[e];
            } else {
                for (size_t e = 0; e < n; ++e) xs[e] += xr[e];
            }
        }
        for (size_t e = 0; e < n; ++e) xs[e] = xs[e] / (float)n_rk;
        voc_round(xs, n);
        free(xr); free(t1); free(t2); free(y);
        x = xs;
    }
    /* F.leaky_relu default slope 0.01 (models.py:128), conv_post without bias, tanh */
    float* o = (float*)malloc(sizeof(float) * (size_t)Tc);
    orc_conv1d(x, C, Tc, p, NULL, 1, 7, 1, 3, 0.01f, o);
    for (int t = 0; t < Tc; ++t) out[t] = tanhf(o[t]);
    free(o); free(x);
}

/* ------------------------------------------------------------------ Flow, reverse (appendix A.6)
 * Per coupling layer pack (weights already weight-norm folded by the caller: W = g * v/||v||):
 *   pre.w[H][half] pre.b[H] cond.w[8H][gin] cond.b[8H]
 *   for l in 0..3: in.w[2H][H][5] in.b[2H] rs.w[R][H] rs.b[R]   (R = 2H for l<3, H for l==3)
 *   post.w[half][H] post.b[half]
 * x [C][T] in/out (C = 2*half), mask [T], g [gin][Tg].  Order: Flip, RCL3, Flip, RCL2, ... RCL0.
 */
ORC_API long orc_flow_layer_floats(int half, int H, int gin) {
    long n = (long)H * half + H + 8L * H * gin + 8L * H;
    for (int l = 0; l < 4; ++l) {
        long R = l < 3 ? 2L * H : H;
        n += 2L * H * H * 5 + 2L * H + R * H + R;
    }
    n += (long)half * H + half;
    return n;
}

ORC_API void orc_flow_reverse(const float* pack, int n_flows, int half, int H, int gin, float* x, int T,
                              const float* mask, const float* g, int Tg) {
    int C = 2 * half;
    size_t HT = (size_t)H * T;
    float* h = (float*)malloc(sizeof(float) * HT);
    float* outp = (float*)malloc(sizeof(float) * HT);
    float* a = (float*)malloc(sizeof(float) * 2 * HT);
    float* acts = (float*)malloc(sizeof(float) * HT);
    float* rs = (float*)malloc(sizeof(float) * 2 * HT);
    float* gc = (float*)malloc(sizeof(float) * (size_t)8 * H * Tg);
    float* m = (float*)malloc(sizeof(float) * (size_t)half * T);
    float* tmp = (float*)malloc(sizeof(float) * (size_t)C * T);
    float* gq = NULL;
    if (g_round & ORC_R_VOC) {
        voc_round(x, (size_t)C * T);
        gq = (float*)malloc(sizeof(float) * (size_t)gin * Tg); memcpy(gq, g, sizeof(float) * (size_t)gin * Tg); voc_round(gq, (size_t)gin * Tg); g = gq;
    }
    for (int f = n_flows - 1; f >= 0; --f) {
        /* Flip: reverse channel order (modules.py:504-511) */
        for (int c = 0; c < C; ++c) memcpy(tmp + (size_t)c * T, x + (size_t)(C - 1 - c) * T, sizeof(float) * T);
        memcpy(x, tmp, sizeof(float) * (size_t)C * T);
        const float* p = pack + (size_t)f * orc_flow_layer_floats(half, H, gin);
        const float* pre_w = p; p += (size_t)H * half;
        const float* pre_b = p; p += H;
        const float* cond_w = p; p += (size_t)8 * H * gin;
        const float* cond_b = p; p += 8 * H;
        /* h = pre(x0) * mask */
        orc_conv1d(x, half, T, pre_w, pre_b, H, 1, 1, 0, 1.0f, h);
        for (int c = 0; c < H; ++c)
            for (int t = 0; t < T; ++t) h[(size_t)c * T + t] *= mask[t];
        voc_round(h, HT);
        orc_conv1d(g, gin, Tg, cond_w, cond_b, 8 * H, 1, 1, 0, 1.0f, gc);
        memset(outp, 0, sizeof(float) * HT);
        for (int l = 0; l < 4; ++l) {
            int R = l < 3 ? 2 * H : H;
            const float* in_w = p; p += (size_t)2 * H * H * 5;
            const float* in_b = p; p += 2 * H;
            const float* rs_w = p; p += (size_t)R * H;
            const float* rs_b = p; p += R;
            orc_conv1d(h, H, T, in_w, in_b, 2 * H, 5, 1, 2, 1.0f, a);
            for (int c = 0; c < H; ++c)
                for (int t = 0; t < T; ++t) {
                    int tg = Tg == 1 ? 0 : t;
                    float ta = a[(size_t)c * T + t] + gc[(size_t)(l * 2 * H + c) * Tg + tg];
                    float sa = a[(size_t)(H + c) * T + t] + gc[(size_t)(l * 2 * H + H + c) * Tg + tg];
                    acts[(size_t)c * T + t] = tanhf(ta) * (1.0f / (1.0f + expf(-sa)));
                }
            voc_round(acts, HT);
            orc_conv1d(acts, H, T, rs_w, rs_b, R, 1, 1, 0, 1.0f, rs);
            if (l < 3) {
                for (int c = 0; c < H; ++c)
                    for (int t = 0; t < T; ++t) {
                        h[(size_t)c * T + t] = (h[(size_t)c * T + t] + rs[(size_t)c * T + t]) * mask[t];
                        if (g_round & ORC_R_VOC) h[(size_t)c * T + t] = bf16r(h[(size_t)c * T + t]);
                        outp[(size_t)c * T + t] += rs[(size_t)(H + c) * T + t];
                    }
            } else {
                for (size_t e = 0; e < HT; ++e) outp[e] += rs[e];
            }
        }
        for (int c = 0; c < H; ++c)
            for (int t = 0; t < T; ++t) outp[(size_t)c * T + t] *= mask[t];
        voc_round(outp, HT);
        const float* post_w = p; p += (size_t)half * H;
        const float* post_b = p; p += half;
        orc_conv1d(outp, H, T, post_w, post_b, half, 1, 1, 0, 1.0f, m);
        /* x1 = (x1 - m*mask) * exp(-0) * mask */
        for (int c = 0; c < half; ++c)
            for (int t = 0; t < T; ++t) {
                float mm = m[(size_t)c * T + t] * mask[t];
                float* x1 = x + (size_t)(half + c) * T + t;
                *x1 = (*x1 - mm) * mask[t];
                if (g_round & ORC_R_VOC) *x1 = bf16r(*x1);
            }
    }
    free(gq);
    free(h); free(outp); free(a); free(acts); free(rs); free(gc); free(m); free(tmp);
}
