// align: monotonic Viterbi alignment of vocoder frames to phonemes for subtitles
// (reference: TTS._viterbi_monotonic, gsv_tts/TTS.py:1744-1797 -- there a Python loop of ~6 tiny tensor
// ops per frame; here two launches).
//
//   attn   fp32 [H][T][N]   MRTE cross-attention probabilities (H heads, T frames, N phonemes)
//   assign int32 [T]        phoneme index per frame, -1 for frames before the first frame whose averaged
//                           attention peaks at phoneme 0
//
// Semantics restated from the cited lines:
//   head h votes at frame t unless its arg-max is the last phoneme;  normal[t] = mean of the voting heads' rows
//   (sum in head order, divided by the vote count), or a fixed near-uniform row when no head votes
//   (1/N everywhere, 0.9/N at N-1, 1.1/N at 1, renormalised);
//   dp[0] = normal[0];  dp[t][n] = normal[t][n] + max(dp[t-1][n], dp[t-1][n-1])  (a tie stays on n);
//   path ends at the first arg-max of dp[T-1] and is traced back; frames before `first_zero` get -1.
// The renormalising sum of the fixed row is taken in fp64 (closed form) and rounded once: torch's fp32
// row sum depends on the host's vector width, so that one scalar has no device-independent reference value.
//
// Kernel 1 (T/4 blocks, one wave per frame) builds normal[] and the per-frame "peaks at 0" flag; kernel 2 is
// one block that walks the T frames: dp rows ping-pong in LDS (one barrier per frame), the normal[] rows are
// prefetched a group of frames ahead into registers, the back-pointers are one bit per (t, n) kept in LDS
// (global workspace when T*N bits exceed LDS), and one lane chases them backwards.
#pragma once
#include "gsv_common.h"

namespace gsv {

constexpr int kAlignMaxHeads = 8;

static __global__ __launch_bounds__(256) void align_normal_kernel(const float* __restrict__ attn, int H, int T, int N,
                                                            float* __restrict__ normal, int* __restrict__ rowflag) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    float mask[kAlignMaxHeads];
    int count = 0;
    for (int h = 0; h < H; ++h) {
        const float* row = attn + ((size_t)h * T + t) * N;
        float best = -INFINITY;
        int bi = N;
        for (int n = lane; n < N; n += 64) {
            const float v = row[n];
            if (v > best) { best = v; bi = n; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ob = __shfl_xor(best, off);
            const int oi = __shfl_xor(bi, off);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        const bool votes = bi != N - 1;
        mask[h] = votes ? 1.f : 0.f;
        count += votes;
    }
    // the fixed row for frames without votes
    const float f1 = (float)(1.0 / N), f09 = (float)(0.9 / N), f11 = (float)(1.1 / N);
    const double dsum = N > 2 ? (double)(N - 2) * (double)f1 + (double)f09 + (double)f11 : (double)f1 + (double)f11;
    const float fsum = (float)dsum;
    const float fc = (float)count;
    float best = -INFINITY;
    int bi = N;
    for (int n = lane; n < N; n += 64) {
        float v;
        if (count > 0) {
            float s = 0.f;
            for (int h = 0; h < H; ++h) s += attn[((size_t)h * T + t) * N + n] * mask[h];
            v = s / fc;
        } else {
            v = (n == 1 ? f11 : (n == N - 1 ? f09 : f1)) / fsum;
        }
        normal[(size_t)t * N + n] = v;
        if (v > best) { best = v; bi = n; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) rowflag[t] = bi == 0;
}

// TH threads, each owns phonemes tid + k*TH (k < NPT).  LDS: dp[2][TH*NPT] floats, then (bits_in_lds) the
// back-pointer bits [T][nw] as 64-bit words, nw = TH*NPT/64.
template <int TH, int NPT>
__global__ __launch_bounds__(TH) void align_dp_kernel(const float* __restrict__ normal, const int* __restrict__ rowflag, int T,
                                                      int N, int* __restrict__ assign, unsigned long long* bits_global,
                                                      int bits_in_lds) {
    constexpr int NP = TH * NPT;
    constexpr int NW = NP / 64;
    constexpr int G = 8;   // frames of normal[] held in registers ahead of the walk
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* dp = reinterpret_cast<float*>(lds);
    unsigned long long* bits = bits_in_lds ? reinterpret_cast<unsigned long long*>(lds + 2 * NP * sizeof(float)) : bits_global;
    __shared__ int first_zero;
    __shared__ float red_v[TH / 64];
    __shared__ int red_i[TH / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

    if (tid == 0) first_zero = T;
    __syncthreads();
    for (int t = tid; t < T; t += TH)
        if (rowflag[t]) atomicMin(&first_zero, t);

    int cn[NPT];   // clamped phoneme index for loads
#pragma unroll
    for (int k = 0; k < NPT; ++k) cn[k] = min(tid + k * TH, N - 1);
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int n = tid + k * TH;
        dp[n] = n < N ? normal[cn[k]] : -INFINITY;
    }
    __syncthreads();

    int cur = 0;
    auto step = [&](int t, const float (&v)[NPT]) {
        const float* src = dp + cur * NP;
        float* dst = dp + (cur ^ 1) * NP;
        float out[NPT];
        bool take[NPT];
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int n = tid + k * TH;
            const float p = src[n];
            const float ps = n > 0 ? src[n - 1] : -INFINITY;
            take[k] = ps > p;
            out[k] = v[k] + (take[k] ? ps : p);
        }
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int n = tid + k * TH;
            dst[n] = n < N ? out[k] : -INFINITY;
            const unsigned long long b = __ballot(take[k] && n < N);
            if (lane == 0) bits[(size_t)t * NW + (n >> 6)] = b;
        }
        __syncthreads();
        cur ^= 1;
    };

    float cu[G][NPT], nx[G][NPT];
    int t0 = 1;
    if (T - 1 >= G) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int k = 0; k < NPT; ++k) cu[g][k] = normal[(size_t)(1 + g) * N + cn[k]];
        for (; t0 + G <= T; t0 += G) {
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int k = 0; k < NPT; ++k) nx[g][k] = normal[(size_t)min(t0 + G + g, T - 1) * N + cn[k]];
#pragma unroll
            for (int g = 0; g < G; ++g) step(t0 + g, cu[g]);
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int k = 0; k < NPT; ++k) cu[g][k] = nx[g][k];
        }
    }
    for (; t0 < T; ++t0) {
        float v[NPT];
#pragma unroll
        for (int k = 0; k < NPT; ++k) v[k] = normal[(size_t)t0 * N + cn[k]];
        step(t0, v);
    }

    // first arg-max of the last row
    const float* last = dp + cur * NP;
    float best = -INFINITY;
    int bi = N;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int n = tid + k * TH;
        if (n < N && last[n] > best) { best = last[n]; bi = n; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { red_v[wid] = best; red_i[wid] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < TH / 64; ++w)
            if (red_v[w] > best || (red_v[w] == best && red_i[w] < bi)) { best = red_v[w]; bi = red_i[w]; }
        const int fz = first_zero == T ? 0 : first_zero;
        int p = bi == N ? 0 : bi;
        assign[T - 1] = T - 1 < fz ? -1 : p;
        for (int t = T - 2; t >= 0; --t) {
            const unsigned long long w = bits[(size_t)(t + 1) * NW + (p >> 6)];
            p -= (int)((w >> (p & 63)) & 1ull);
            assign[t] = t < fz ? -1 : p;
        }
    }
}

}  // namespace gsv
