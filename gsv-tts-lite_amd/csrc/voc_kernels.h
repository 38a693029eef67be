// Small layout / elementwise kernels of the SoVITS flow + Generator path (gfx950).
// Everything dense goes through tapgemm.h; these only move or gate data.
#pragma once
#include "gsv_common.h"

namespace gsv {

// torch channels-first fp32 [C][T]  ->  channels-last AT [T][ld] (pad channels zeroed)
template <typename AT>
__global__ void cf_to_cl_kernel(const float* __restrict__ src, AT* __restrict__ dst, int C, int T, int ld) {
    __shared__ float tile[32][33];
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty in 0..7
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, t = t0 + tx;
        tile[i][tx] = (c < C && t < T) ? src[(size_t)c * T + t] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, c = c0 + tx;
        if (t < T && c < ld) dst[(size_t)t * ld + c] = from_f32<AT>(tile[tx][i]);
    }
}

// ---- per-frame conditioning with few distinct columns ---------------------------------------------------------------------------
// A time-concatenated batch (TTS.infer_batched, TTS.py:728-764) hands the vocoder one `ge` column PER FRAME, but the columns of one
// utterance are all the same: ten utterances are ten distinct columns among thousands.  The conditioning GEMMs (flow: 6 144 outputs,
// Generator: 512, K = gin) then run on the distinct columns only and every consumer indexes their rows through seg_id[frame].
// Detection is exact (bit compare of neighbouring columns) and stays on the device: nothing is assumed about the caller's layout.
// flag[t] |= (column t differs from column t - 1 in this block's 64-channel group); flag[0] = 1.  grid (ceil(T / 256), ceil(C / 64))
static __global__ __launch_bounds__(256) void seg_flag_kernel(const float* __restrict__ ge, int C, int T, int* __restrict__ flag) {
    const int t = blockIdx.x * 256 + threadIdx.x, c0 = blockIdx.y * 64;
    if (t >= T) return;
    if (t == 0) { if (blockIdx.y == 0) flag[0] = 1; return; }
    int diff = 0;
    for (int c = c0; c < min(c0 + 64, C); ++c) {
        const uint32_t a = __float_as_uint(ge[(size_t)c * T + t]), b = __float_as_uint(ge[(size_t)c * T + t - 1]);
        diff |= (a != b);
    }
    if (diff) atomicOr(flag + t, 1);
}
// seg_id[t] = number of flagged frames in [0, t] - 1; seg_first[s] = first frame of segment s; *nseg.  One block of 1024 threads.
static __global__ __launch_bounds__(1024) void seg_scan_kernel(const int* __restrict__ flag, int T, int* __restrict__ seg_id, int* __restrict__ seg_first,
                                                                 int* __restrict__ nseg) {
    __shared__ int part[1024];
    const int tid = threadIdx.x, per = (T + 1023) / 1024, lo = tid * per, hi = min(lo + per, T);
    int n = 0;
    for (int t = lo; t < hi; ++t) n += flag[t] != 0;
    part[tid] = n;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int id = part[tid] - n;                                  // flagged frames before this thread's range
    for (int t = lo; t < hi; ++t) {
        if (flag[t] != 0) { seg_first[id] = t; ++id; }
        seg_id[t] = id - 1;
    }
    if (tid == 1023) *nseg = part[1023];
}
// the distinct columns, channels-last: dst[s][c] = ge[c][seg_first[s]] for s < *nseg.  grid (ceil(T / 32), ceil(ld / 256)): blocks past
// the segment count leave at once
template <typename AT>
__global__ __launch_bounds__(256) void seg_gather_cl_kernel(const float* __restrict__ ge, int C, int T, const int* __restrict__ seg_first,
                                                            const int* __restrict__ nseg, AT* __restrict__ dst, int ld) {
    const int ns = *nseg, s0 = blockIdx.x * 32;
    if (s0 >= ns) return;
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= ld) return;
    for (int s = s0; s < min(s0 + 32, ns); ++s) dst[(size_t)s * ld + c] = c < C ? from_f32<AT>(ge[(size_t)c * T + seg_first[s]]) : from_f32<AT>(0.f);
}

// channels-last AT [T][ld] -> torch channels-first fp32 [C][T]
template <typename AT>
__global__ void cl_to_cf_kernel(const AT* __restrict__ src, float* __restrict__ dst, int C, int T, int ld) {
    __shared__ float tile[32][33];
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, c = c0 + tx;
        tile[i][tx] = (t < T && c < C) ? to_f32<AT>(src[(size_t)t * ld + c]) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, t = t0 + tx;
        if (c < C && t < T) dst[(size_t)c * T + t] = tile[tx][i];
    }
}

// Flip (modules.py:504-511): dst[t][c] = src[t][C-1-c]
template <typename AT>
__global__ void flip_kernel(const AT* __restrict__ src, AT* __restrict__ dst, int C, int T, int ld) {
    const size_t n = (size_t)T * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t t = i / C;
        const int c = (int)(i % C);
        dst[t * ld + c] = src[t * ld + (C - 1 - c)];
    }
}

// fused_add_tanh_sigmoid_multiply (commons.py:14-21), conditioning already added by the conv
// epilogue: acts[t][c] = tanh(a[t][c]) * sigmoid(a[t][H + c])
template <typename AT>
__global__ void gate_kernel(const AT* __restrict__ a, AT* __restrict__ acts, int H, int T) {
    const size_t n = (size_t)T * H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t t = i / H;
        const int c = (int)(i % H);
        const float ta = to_f32<AT>(a[t * 2 * H + c]);
        const float sa = to_f32<AT>(a[t * 2 * H + H + c]);
        acts[t * H + c] = from_f32<AT>(tanhf(ta) * (1.0f / (1.0f + expf(-sa))));
    }
}

// mean of the three resblock branches (models.py:121-127): y = (a + b + c) / 3, 8 elements per thread
template <typename AT>
__global__ void avg3_kernel(const AT* __restrict__ a, const AT* __restrict__ b, const AT* __restrict__ c,
                            AT* __restrict__ y, size_t n) {
    constexpr int V = 16 / sizeof(AT);
    const size_t nv = n / V;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
        float fa[V], fb[V], fc[V];
        Ld<AT, V>::load(a + i * V, fa);
        Ld<AT, V>::load(b + i * V, fb);
        Ld<AT, V>::load(c + i * V, fc);
        AT o[V];
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = from_f32<AT>(((fa[e] + fb[e]) + fc[e]) / 3.0f);
        *reinterpret_cast<u32x4*>(y + i * V) = *reinterpret_cast<const u32x4*>(o);
    }
}

// Generator tail (SoVITS/models.py:129-131): y[n] = tanh( sum_t sum_c w[c][t] * lrelu_{0.01}(x[n + t - 3][c]) ), one output
// channel, k = 7, no bias.  One sample per thread, rows staged (after the leaky-ReLU, fp32) in LDS with a
// 4-float skew; the 32x32 MFMA tile of the generic kernel would waste 31 of its 32 output rows here.
template <typename AT, int C>
__global__ __launch_bounds__(256) void conv_post_kernel(const AT* __restrict__ x, int ld, const float* __restrict__ w,
                                                        float* __restrict__ y, int n_rows) {
    constexpr int RS = C + 4;
    __shared__ float xs[(256 + 6) * RS];
    __shared__ float ws[C * 7];
    const int tid = threadIdx.x, n0 = blockIdx.x * 256;
    for (int i = tid; i < C * 7; i += 256) ws[i] = w[i];
    constexpr int VPR = C / (16 / (int)sizeof(AT));          // 16-byte vectors per row
    constexpr int EPV = 16 / (int)sizeof(AT);
    for (int e = tid; e < (256 + 6) * VPR; e += 256) {
        const int r = e / VPR, cv = e % VPR;
        const int g = n0 - 3 + r;
        float v[EPV];
#pragma unroll
        for (int i = 0; i < EPV; ++i) v[i] = 0.f;
        if (g >= 0 && g < n_rows) Ld<AT, EPV>::load(x + (size_t)g * ld + cv * EPV, v);
#pragma unroll
        for (int i = 0; i < EPV; ++i) xs[r * RS + cv * EPV + i] = fmaxf(v[i], v[i] * 0.01f);
    }
    __syncthreads();
    const int n = n0 + tid;
    if (n >= n_rows) return;
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int c = 0; c < C; ++c) acc = fmaf(ws[c * 7 + t], xs[(tid + t) * RS + c], acc);
    y[n] = tanhf(acc);
}

// W = v * (g / ||v||) per output row (torch.nn.utils.weight_norm, dim=0); one block per row
static __global__ void weight_norm_fold_kernel(const float* __restrict__ g, const float* __restrict__ v,
                                        float* __restrict__ w, int row_elems, float sign) {
    __shared__ float red[8];
    const int r = blockIdx.x;
    const float* vr = v + (size_t)r * row_elems;
    float s = 0.f;
    for (int i = threadIdx.x; i < row_elems; i += blockDim.x) s += vr[i] * vr[i];
    s = block_sum<4>(s, red);
    const float f = sign * g[r] / sqrtf(s);
    for (int i = threadIdx.x; i < row_elems; i += blockDim.x) w[(size_t)r * row_elems + i] = vr[i] * f;
}

static __global__ void scale_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n, float s) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i] * s;
}

// dst[i] = s * src[i]  or, reversed, s * src[n-1-i]
static __global__ void scale_copy_rev_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, float s, int reverse) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = s * src[reverse ? n - 1 - i : i];
}

// dst[c] = (t[c] + t[n+c]) + (t[2n+c] + t[3n+c])
static __global__ void sum4_kernel(const float* __restrict__ t, float* __restrict__ dst, int n) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) dst[c] = (t[c] + t[n + c]) + (t[2 * n + c] + t[3 * n + c]);
}

template <typename WT>
__global__ void convert_kernel(const float* __restrict__ src, WT* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = from_f32<WT>(src[i]);
}

// generic gather-pack: dst[i] = src[map(i)] for the decode weight panels (see t2s pack lambdas)
// qkv head panels: dst[h][r][c], r in [0,96): q/k/v row (r/32)*512 + h*32 + r%32
template <typename WT>
__global__ void pack_qkv_panel_kernel(const float* __restrict__ w, WT* __restrict__ dst) {
    const size_t n = (size_t)16 * 96 * 512;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = i % 512;
        const int r = (i / 512) % 96;
        const int h = i / (512 * 96);
        const int row = (r / 32) * 512 + h * 32 + (r % 32);
        dst[i] = from_f32<WT>(w[(size_t)row * 512 + c]);
    }
}
static __global__ void pack_qkv_bias_kernel(const float* __restrict__ b, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 16 * 96) {
        const int r = i % 96, h = i / 96;
        dst[i] = b[(r / 32) * 512 + h * 32 + (r % 32)];
    }
}
// column-slice panels: dst[j][n][i] = w[n][j*K + i], w is [512][J*K]
template <typename WT>
__global__ void pack_col_panel_kernel(const float* __restrict__ w, WT* __restrict__ dst, int J, int K) {
    const size_t n = (size_t)J * 512 * K;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int i = idx % K;
        const int row = (idx / K) % 512;
        const int j = idx / ((size_t)K * 512);
        dst[idx] = from_f32<WT>(w[(size_t)row * (J * K) + (size_t)j * K + i]);
    }
}

}  // namespace gsv
