// sola: the streaming splice of TTS._sola_algorithm (reference gsv_tts/TTS.py:1612-1627) on the device.
//
// A streamed utterance is vocoded chunk by chunk; consecutive chunks overlap by `overlap` samples.  The new chunk is slid over the
// previous chunk's tail by the offset (0 .. search_len) that maximises the normalised cross-correlation
//     corr[k] / sqrt(energy[k]),  corr[k] = sum_j chunk[k + j] * tail[j],  energy[k] = sum_j chunk[k + j]^2 + 1e-8
// (first maximum, torch.argmax's rule), then cross-faded over the overlap with alpha = linspace(0, 1, overlap).
//
// Two launches: (1) one block per candidate offset -- 321 blocks for the default search, every lane streams the tail and its window
// of the chunk with coalesced 4-byte loads, wave shuffles + one LDS meeting reduce the two sums; (2) a grid-stride pass in which
// every block re-derives the arg-max from the 321 scores (an L2-resident 1.3 KB read; cheaper than a third launch), block 0
// publishes it, and everyone writes out[i] = fade(i) for i < overlap, chunk[offset + i] behind it.
#pragma once
#include <hip/hip_runtime.h>

namespace gsv {

__device__ __forceinline__ float sola_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// score[k] for k = blockIdx.x.  Sums in a fixed order (lane-strided partials, xor-tree, waves in index order): reproducible.
static __global__ __launch_bounds__(256) void sola_score_kernel(const float* __restrict__ tail, const float* __restrict__ chunk, int overlap,
                                                                 float* __restrict__ score) {
    __shared__ float red[2][4];
    const int k = blockIdx.x, tid = threadIdx.x;
    float c = 0.f, e = 0.f;
    for (int j = tid; j < overlap; j += 256) {
        const float x = chunk[k + j];
        c = fmaf(x, tail[j], c);
        e = fmaf(x, x, e);
    }
    c = sola_wave_sum(c);
    e = sola_wave_sum(e);
    if ((tid & 63) == 0) { red[0][tid >> 6] = c; red[1][tid >> 6] = e; }
    __syncthreads();
    if (tid == 0) {
        const float cs = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        const float es = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]) + 1e-8f;
        score[k] = cs / sqrtf(es);
    }
}

// torch.linspace(0, 1, n)[j] as torch computes it (symmetric halves: start + step * j below the middle, end - step * (n - 1 - j) above)
__device__ __forceinline__ float sola_alpha(int j, int n) {
    if (n == 1) return 0.f;
    const float step = 1.0f / (float)(n - 1);
    return j < n / 2 ? step * (float)j : 1.0f - step * (float)(n - 1 - j);
}

static __global__ __launch_bounds__(256) void sola_splice_kernel(const float* __restrict__ tail, const float* __restrict__ chunk, int n, int overlap,
                                                                  const float* __restrict__ score, int n_off, float* __restrict__ out,
                                                                  int* __restrict__ offset_out) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    const int tid = threadIdx.x;
    // arg-max over the scores, FIRST maximum (a NaN score -- silence against silence gives 0 / sqrt(1e-8) = 0, never NaN -- would
    // lose every comparison and leave offset 0)
    float best = -INFINITY;
    int at = 0x7fffffff;
    for (int k = tid; k < n_off; k += 256) {
        const float s = score[k];
        if (s > best) { best = s; at = k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oa = __shfl_xor(at, o, 64);
        if (ob > best || (ob == best && oa < at)) { best = ob; at = oa; }
    }
    if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = at; }
    __syncthreads();
    best = bv[0]; at = bi[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (bv[w] > best || (bv[w] == best && bi[w] < at)) { best = bv[w]; at = bi[w]; }
    const int off = at == 0x7fffffff ? 0 : at;
    if (blockIdx.x == 0 && tid == 0) *offset_out = off;
    const int m = n - off;                                   // samples of the spliced chunk
    for (int i = blockIdx.x * 256 + tid; i < m; i += gridDim.x * 256) {
        float v = chunk[off + i];
        if (i < overlap) {
            const float a = sola_alpha(i, overlap);
            v = tail[i] * (1.0f - a) + v * a;
        }
        out[i] = v;
    }
}

}  // namespace gsv
