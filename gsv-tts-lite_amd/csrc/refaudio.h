// refaudio: the reference-audio path that runs once per new speaker / prompt:
//   spectrogram      TTS._get_spec, gsv_tts/TTS.py:1576-1610 (torchaudio Spectrogram: hann window, centre, reflect
//                    padding, magnitude)
//   get_ge           SynthesizerTrn.get_ge, SoVITS/models.py:371-378 -> MelStyleEncoder.forward,
//                    module/modules.py:367-444 (+ sv_emb / PReLU for v2Pro / v2ProPlus)
//   extract_latent   SynthesizerTrn.extract_latent, models.py:431-434 -> EuclideanCodebook.quantize,
//                    module/core_vq.py:124-128
//
// Everything here is a small dense contraction over a few hundred frames, so one fp32 MFMA GEMM kernel
// (v_mfma_f32_32x32x2_f32, fp32 in / fp32 accumulate: this path feeds every later stage and runs once, so it
// keeps full precision in both numerics modes) carries all of it.  The trick that removes every im2col /
// framing copy is the row stride: X rows may OVERLAP (ldx < K), so
//   * STFT framing     = rows of 2048 samples at stride hop (640) over the reflect-padded signal,
//   * a k-tap conv      = rows of k*C values at stride C over the zero-padded channels-last activations,
//   * the stride-2 conv = rows of 2*C values at stride 2*C
// are all plain  Y[m][n] = act(alpha * sum_k X[m*ldx + k] * W[n*ldw + k] + bias) + R[m][n].
#pragma once
#include "gsv_common.h"

namespace gsv {

typedef float fa16 __attribute__((ext_vector_type(16)));

struct FGemmArgs {
    const float* X; long long ldx;    // [M] rows of K values, stride ldx (may be < K)
    const float* W; long long ldw;    // [N] rows of K values
    float* Y; long long ldy;          // [M][N]
    const float* bias_n;              // [N] or null
    const float* bias_m;              // [M] or null
    const float* R; long long ldr;    // residual [M][N] or null
    int M, N, K;
    float alpha;
    int act;                          // 0 none, 1 mish (x * tanh(softplus(x)))
};

__device__ __forceinline__ float mish_f(float x) {
    // F.softplus (beta 1, threshold 20) then tanh, as module/modules.py:230-235 evaluates it
    const float sp = x > 20.f ? x : log1pf(expf(x));
    return x * tanhf(sp);
}

// 64 x 64 output tile per block, 4 waves (2 x 2) of 32 x 32, K staged through LDS in chunks of 32.
static __global__ __launch_bounds__(256) void fgemm_kernel(FGemmArgs a) {
    constexpr int KC = 32, LD = KC + 1;
    __shared__ float xs[64 * LD];
    __shared__ float ws[64 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int j = lane & 31, hf = lane >> 5;
    fa16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    const int lr = tid >> 5, lc = tid & 31;   // staging: 8 rows x 32 k per pass
    for (int k0 = 0; k0 < a.K; k0 += KC) {
        float xv[8], wv[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int r = lr + p * 8, k = k0 + lc;
            const int m = m0 + r, n = n0 + r;
            xv[p] = (m < a.M && k < a.K) ? a.X[(long long)m * a.ldx + k] : 0.f;
            wv[p] = (n < a.N && k < a.K) ? a.W[(long long)n * a.ldw + k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            xs[(lr + p * 8) * LD + lc] = xv[p];
            ws[(lr + p * 8) * LD + lc] = wv[p];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KC; kk += 2) {
            const float av = xs[(wm * 32 + j) * LD + kk + hf];
            const float bv = ws[(wn * 32 + j) * LD + kk + hf];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }
    const int n = n0 + wn * 32 + j;
    if (n >= a.N) return;
    const float bn = a.bias_n ? a.bias_n[n] : 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int m = m0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * hf;
        if (m < a.M) {
            float v = a.alpha * acc[q] + bn + (a.bias_m ? a.bias_m[m] : 0.f);
            if (a.act == 1) v = mish_f(v);
            if (a.R) v += a.R[(long long)m * a.ldr + n];
            a.Y[(long long)m * a.ldy + n] = v;
        }
    }
}

// out[c][r] = in[r][c] for r < rows, c < cols  (in row stride ldi, out row stride ldo)
static __global__ void transpose_kernel(const float* __restrict__ in, long long ldi, float* __restrict__ out, long long ldo, int rows,
                                 int cols) {
    __shared__ float t[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
    for (int i = ty; i < 32; i += 8)
        t[i][tx] = (r0 + i < rows && c0 + tx < cols) ? in[(long long)(r0 + i) * ldi + c0 + tx] : 0.f;
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) out[(long long)(c0 + i) * ldo + r0 + tx] = t[tx][i];
}

// conv weight [cout][cin][k] -> [cout][k][cin] (the row a k-tap "overlapping rows" GEMM contracts with)
static __global__ void conv_weight_kc_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int k) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)cout * cin * k) return;
    const int t = (int)(i % k), c = (int)((i / k) % cin), o = (int)(i / ((long long)k * cin));
    out[((long long)o * k + t) * cin + c] = w[i];
}

// Conv1dGLU tail, modules.py:248-254: y = x + a * sigmoid(b), conv output rows [a | b] of 2*C
static __global__ void glu_residual_kernel(const float* __restrict__ x, const float* __restrict__ conv, float* __restrict__ y, int rows,
                                    int C) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)rows * C) return;
    const int r = (int)(i / C), c = (int)(i % C);
    const float av = conv[(long long)r * 2 * C + c], bv = conv[(long long)r * 2 * C + C + c];
    y[i] = x[i] + av * (1.f / (1.f + expf(-bv)));
}

// in-place softmax of each row of S [rows][cols]; one wave per row
static __global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ S, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float* row = S + (long long)r * cols;
    float mx = -INFINITY;
    for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, row[c]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.f;
    for (int c = lane; c < cols; c += 64) {
        const float e = expf(row[c] - mx);
        row[c] = e;
        sum += e;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float inv = 1.f / sum;
    for (int c = lane; c < cols; c += 64) row[c] *= inv;
}

// temporal_avg_pool (modules.py:409-419: every element divided by the length, then summed) + the v2Pro tail
// ge = PReLU(pool + sv) (models.py:374-377).  One thread per channel; F [T][C].
static __global__ void pool_prelu_kernel(const float* __restrict__ F, int T, int C, const float* __restrict__ sv,
                                  const float* __restrict__ prelu_w, float* __restrict__ ge) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float len = (float)T;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += F[(long long)t * C + c] / len;
    if (sv) {
        s += sv[c];
        s = s >= 0.f ? s : prelu_w[c] * s;
    }
    ge[c] = s;
}

// y[n] = bias[n] + sum_k x[k] * W[n][k]: one wave per output row, 16-byte loads (the 20480 -> gin sv_emb linear is
// 84 MB of fp32 weights read once: HBM-bound, so every CU streams rows instead of 16 GEMM tiles doing it)
static __global__ __launch_bounds__(256) void gemv_rows_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ y, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* row = W + (long long)n * K;
    float s = 0.f;
    const int K4 = (K % 4 == 0 && (reinterpret_cast<size_t>(row) & 15) == 0 && (reinterpret_cast<size_t>(x) & 15) == 0) ? K / 4 : 0;
    for (int k = lane; k < K4; k += 64) {
        const float4 wv = reinterpret_cast<const float4*>(row)[k];
        const float4 xv = reinterpret_cast<const float4*>(x)[k];
        s += wv.x * xv.x + wv.y * xv.y + wv.z * xv.z + wv.w * xv.w;
    }
    for (int k = K4 * 4 + lane; k < K; k += 64) s += row[k] * x[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) y[n] = s + (bias ? bias[n] : 0.f);
}

// reflect padding of `pad` samples on both sides (torch.stft center=True, pad_mode="reflect")
static __global__ void reflect_pad_kernel(const float* __restrict__ x, int n, int pad, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n + 2 * pad) return;
    int s = i - pad;
    if (s < 0) s = -s;
    if (s >= n) s = 2 * (n - 1) - s;
    out[i] = x[s];
}

// windowed DFT rows: D[2j][k] = hann[k] cos(2 pi j k / n_fft), D[2j+1][k] = -hann[k] sin(2 pi j k / n_fft),
// hann periodic (torch.hann_window default); evaluated in fp64, stored fp32
static __global__ void dft_rows_kernel(float* __restrict__ D, int n_fft, int bins) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)bins * n_fft) return;
    const int k = (int)(i % n_fft), jbin = (int)(i / n_fft);
    const double w = 0.5 - 0.5 * cospi(2.0 * k / n_fft);
    const int ph = (int)(((long long)jbin * k) % n_fft);
    const double ang = 2.0 * ph / n_fft;
    D[((long long)2 * jbin) * n_fft + k] = (float)(w * cospi(ang));
    D[((long long)2 * jbin + 1) * n_fft + k] = (float)(-w * sinpi(ang));
}

// |re + i im| of Z [T][2*bins] -> spec [bins][T] (channels-first, what the reference's Spectrogram returns)
static __global__ void magnitude_t_kernel(const float* __restrict__ Z, int T, int bins, float* __restrict__ spec) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)T * bins) return;
    const int t = (int)(i % T), b = (int)(i / T);
    const float re = Z[(long long)t * 2 * bins + 2 * b], im = Z[(long long)t * 2 * bins + 2 * b + 1];
    spec[i] = sqrtf(re * re + im * im);
}

// row sums of squares: out[r] = sum_c x[r][c]^2
static __global__ __launch_bounds__(256) void rowsq_kernel(const float* __restrict__ x, long long ld, int rows, int cols,
                                                    float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) {
        const float v = x[(long long)r * ld + c];
        s += v * v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[r] = s;
}

// core_vq.py:124-128: dist = -(|x|^2 - 2 x.e + |e|^2), code = first arg-max; margin = best - second best
static __global__ __launch_bounds__(256) void nearest_code_kernel(const float* __restrict__ dot, const float* __restrict__ x2,
                                                           const float* __restrict__ e2, int rows, int bins,
                                                           long long* __restrict__ codes, float* __restrict__ margin) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float best = -INFINITY, second = -INFINITY;
    int bi = bins;
    for (int c = lane; c < bins; c += 64) {
        const float d = -((x2[r] - 2.f * dot[(long long)r * bins + c]) + e2[c]);
        if (d > best) { second = best; best = d; bi = c; }
        else if (d > second) second = d;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off), os = __shfl_xor(second, off);
        const int oi = __shfl_xor(bi, off);
        if (ob > best || (ob == best && oi < bi)) { second = fmaxf(best, os); best = ob; bi = oi; }
        else second = fmaxf(second, ob);
    }
    if (lane == 0) {
        codes[r] = bi;
        if (margin) margin[r] = best - second;
    }
}

}  // namespace gsv
