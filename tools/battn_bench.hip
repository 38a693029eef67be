// Stand-alone sweep of batched-decode attention variants (gfx950); not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/battn_bench.hip -o tools/battn_bench && tools/battn_bench B T kvlo kvhi
// One launch = attention of one layer for B sequences x 16 heads; L layers of K/V buffers are cycled so that nothing
// is cache-resident.  Variants:
//   V0  thread per key, rows loaded on demand (K, then V after the softmax max): the round-1 kernel
//   V1  thread per key, every K and V row of the thread in flight at entry, blind (rows past kv_len loaded and masked)
//   V2  four lanes per row (1 KiB per wave instruction), blind                  = the product kernel with BLIND = true
//   V3  thread per key, kv_len first, then only the live rows of K and V in flight at once
//   V4  four lanes per row, kv_len first, chunks past kv_len re-read row kv_len-1  = the product kernel with BLIND = false
//   V5  as V3 with clamped (always valid, cache-hit) addresses instead of predicated loads
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../gsv-tts-lite_amd/csrc/t2s_batch.h"
using namespace gsv;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- V0: round-1 kernel ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void v0_kernel(BatchAttnArgs<bf16_t> a) {
    typedef bf16_t WT;
    __shared__ float qs[32], kn[32], vn[32], red[8], ored[4][32];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int64_t n64 = a.kv_len[b];
    const int n = (int)(n64 < 0 ? 0 : (n64 > a.T - 1 ? a.T - 1 : n64));
    const float* row = a.qkv + (size_t)b * 1536 + h * 32;
    WT* Kp = a.kc + (((size_t)b * kH + h) * a.T) * kDh;
    WT* Vp = a.vc + (((size_t)b * kH + h) * a.T) * kDh;
    if (tid < 32) {
        qs[tid] = row[tid];
        const WT kq = from_f32<WT>(row[512 + tid]), vq = from_f32<WT>(row[1024 + tid]);
        kn[tid] = to_f32<WT>(kq); vn[tid] = to_f32<WT>(vq);
        Kp[(size_t)n * kDh + tid] = kq; Vp[(size_t)n * kDh + tid] = vq;
    }
    __syncthreads();
    constexpr int MAXK = 4;
    float sc[MAXK];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXK; ++i) {
        const int t = tid + i * 256;
        sc[i] = -INFINITY;
        if (t <= n) {
            float s = 0.f;
            if (t == n) {
#pragma unroll
                for (int d = 0; d < 32; ++d) s = fmaf(qs[d], kn[d], s);
            } else {
                float kr[32];
                Ld<WT, 32>::load(Kp + (size_t)t * kDh, kr);
#pragma unroll
                for (int d = 0; d < 32; ++d) s = fmaf(qs[d], kr[d], s);
            }
            sc[i] = s * 0.17677669529663687f;
            mx = fmaxf(mx, sc[i]);
        }
    }
    mx = block_max<4>(mx, red);
    float sum = 0.f, o[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] = 0.f;
#pragma unroll
    for (int i = 0; i < MAXK; ++i) {
        const int t = tid + i * 256;
        if (t <= n) {
            const float p = expf(sc[i] - mx);
            sum += p;
            if (t == n) {
#pragma unroll
                for (int d = 0; d < 32; ++d) o[d] = fmaf(p, vn[d], o[d]);
            } else {
                float vr[32];
                Ld<WT, 32>::load(Vp + (size_t)t * kDh, vr);
#pragma unroll
                for (int d = 0; d < 32; ++d) o[d] = fmaf(p, vr[d], o[d]);
            }
        }
    }
    sum = block_sum<4>(sum, red);
#pragma unroll
    for (int d = 0; d < 32; ++d) {
        const float w = wave_sum(o[d]);
        if (lane == 0) ored[wid][d] = w;
    }
    __syncthreads();
    if (tid < 32) a.out[(size_t)b * kD + h * 32 + tid] = ((ored[0][tid] + ored[1][tid]) + (ored[2][tid] + ored[3][tid])) / sum;
}

// ---- V1 / V3: thread per key, preloaded ----------------------------------------------------------
__device__ __forceinline__ void wave_reduce32(const float (&o)[32], float (&r)[4]) {
    float a[16], b[8], c[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = halve32_sum(o[i], o[i + 16]);
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = halve16_sum(a[i], a[i + 8]);
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = halve8_sum(b[i], b[i + 4]);
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = oct_sum(c[i]);
}

template <int MAXK, int MODE>
__global__ __launch_bounds__(256) void v13_kernel(BatchAttnArgs<bf16_t> a) {
    typedef bf16_t WT;
    __shared__ __attribute__((aligned(16))) float qs[32], kn[32], vn[32], ored[4][32];
    __shared__ float red[8];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int RB = 4;
    const float* row = a.qkv + (size_t)b * 1536 + h * 32;
    WT* Kp = a.kc + (((size_t)b * kH + h) * a.T) * kDh;
    WT* Vp = a.vc + (((size_t)b * kH + h) * a.T) * kDh;
    const int64_t n64 = a.kv_len[b];
    float rq = 0.f, rk = 0.f, rv = 0.f;
    if (tid < 32) { rq = row[tid]; rk = row[512 + tid]; rv = row[1024 + tid]; }
    constexpr bool BLIND = MODE == 0;
    int nn = a.T;
    if constexpr (!BLIND) nn = (int)(n64 < 0 ? 0 : (n64 > a.T - 1 ? a.T - 1 : n64));
    const int lastrow = MODE == 2 ? max(nn - 1, 0) : a.T - 1;
    raw16 kr[MAXK][RB], vr[MAXK][RB];
#pragma unroll
    for (int i = 0; i < MAXK; ++i) {
        const int t = min(tid + i * 256, lastrow);
#pragma unroll
        for (int c = 0; c < RB; ++c) kr[i][c] = raw16{0u, 0u, 0u, 0u};
        if (MODE != 1 || tid + i * 256 < nn) {
#pragma unroll
            for (int c = 0; c < RB; ++c) kr[i][c] = ldg16(reinterpret_cast<const unsigned char*>(Kp + (size_t)t * kDh) + 16 * c);
        }
    }
#pragma unroll
    for (int i = 0; i < MAXK; ++i) {
        const int t = min(tid + i * 256, lastrow);
#pragma unroll
        for (int c = 0; c < RB; ++c) vr[i][c] = raw16{0u, 0u, 0u, 0u};
        if (MODE != 1 || tid + i * 256 < nn) {
#pragma unroll
            for (int c = 0; c < RB; ++c) vr[i][c] = ldg16(reinterpret_cast<const unsigned char*>(Vp + (size_t)t * kDh) + 16 * c);
        }
    }
    asm volatile("" : "+v"(rq) : : "memory");
    const int n = (int)(n64 < 0 ? 0 : (n64 > a.T - 1 ? a.T - 1 : n64));
    if (tid < 32) {
        qs[tid] = rq;
        const WT kq = from_f32<WT>(rk), vq = from_f32<WT>(rv);
        kn[tid] = to_f32<WT>(kq); vn[tid] = to_f32<WT>(vq);
        Kp[(size_t)n * kDh + tid] = kq; Vp[(size_t)n * kDh + tid] = vq;
    }
    __syncthreads();
    float q[32];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(qs + 4 * c);
        q[4 * c] = t[0]; q[4 * c + 1] = t[1]; q[4 * c + 2] = t[2]; q[4 * c + 3] = t[3];
    }
    float sc[MAXK];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXK; ++i) {
        const int t = tid + i * 256;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < RB; ++c) {
            float kk[8];
            Unpack<WT, 8>::run(kr[i][c], kk);
#pragma unroll
            for (int e = 0; e < 8; ++e) s = fmaf(q[c * 8 + e], kk[e], s);
        }
        if (t == n) {
            s = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) s = fmaf(q[d], kn[d], s);
        }
        sc[i] = t <= n ? s * 0.17677669529663687f : -INFINITY;
        mx = fmaxf(mx, sc[i]);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wid] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f, o[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] = 0.f;
#pragma unroll
    for (int i = 0; i < MAXK; ++i) {
        const int t = tid + i * 256;
        const float p = t <= n ? __expf(sc[i] - mx) : 0.f;
        sum += p;
        const bool own = t == n;
#pragma unroll
        for (int c = 0; c < RB; ++c) {
            float vv[8];
            Unpack<WT, 8>::run(vr[i][c], vv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = own ? vn[c * 8 + e] : (t <= n ? vv[e] : 0.f);
                o[c * 8 + e] = fmaf(p, v, o[c * 8 + e]);
            }
        }
    }
    sum = wave_sum(sum);
    float r4[4];
    wave_reduce32(o, r4);
    if ((lane & 7) == 0) {
        const int d0 = 4 * ((lane >> 3) & 1) + 8 * ((lane >> 4) & 1) + 16 * (lane >> 5);
        *reinterpret_cast<f32x4*>(&ored[wid][d0]) = f32x4{r4[0], r4[1], r4[2], r4[3]};
    }
    if (lane == 0) red[4 + wid] = sum;
    __syncthreads();
    if (tid < 32) {
        const float tot = (red[4] + red[5]) + (red[6] + red[7]);
        a.out[(size_t)b * kD + h * 32 + tid] = ((ored[0][tid] + ored[1][tid]) + (ored[2][tid] + ored[3][tid])) / tot;
    }
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, T = argc > 2 ? atoi(argv[2]) : 512;
    const int kvlo = argc > 3 ? atoi(argv[3]) : 200, kvhi = argc > 4 ? atoi(argv[4]) : 300;
    const size_t layer = (size_t)B * 16 * T * 32;
    const int L = (int)std::max<size_t>(2, std::min<size_t>(24, (600ull << 20) / (layer * 4)));
    bf16_t *kc, *vc; float *qkv, *out; int64_t* kvl;
    CK(hipMalloc(&kc, layer * L * 2)); CK(hipMalloc(&vc, layer * L * 2));
    CK(hipMemset(kc, 0x3c, layer * L * 2)); CK(hipMemset(vc, 0x3c, layer * L * 2));
    CK(hipMalloc(&qkv, sizeof(float) * B * 1536)); CK(hipMalloc(&out, sizeof(float) * B * 512)); CK(hipMalloc(&kvl, 8 * B));
    std::vector<float> hq(B * 1536); for (size_t i = 0; i < hq.size(); ++i) hq[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    std::vector<int64_t> hk(B); double kvsum = 0; for (int b = 0; b < B; ++b) { hk[b] = kvlo + (b * 7919) % (kvhi - kvlo + 1); kvsum += hk[b]; }
    CK(hipMemcpy(qkv, hq.data(), hq.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(kvl, hk.data(), 8 * B, hipMemcpyHostToDevice));
    const double bytes = kvsum * 16 * 64 * 2;   // live K + V rows per launch
    printf("B=%d T=%d kv %d..%d, %d layers of buffers, %.1f MB of live K/V per launch\n", B, T, kvlo, kvhi, L, bytes / 1e6);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ref;
    auto bench = [&](const char* name, auto kern) {
        BatchAttnArgs<bf16_t> a; a.qkv = qkv; a.kv_len = kvl; a.T = T; a.out = out;
        const int reps = 30;
        for (int w = 0; w < 2; ++w) {
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r)
                for (int l = 0; l < L; ++l) { a.kc = kc + layer * l; a.vc = vc + layer * l; hipLaunchKernelGGL(kern, dim3(16, B), dim3(256), 0, 0, a); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<float> ho(B * 512); CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
        double err = 0; if (ref.empty()) ref = ho; else for (size_t i = 0; i < ho.size(); ++i) err = std::max(err, (double)fabsf(ho[i] - ref[i]));
        const double us = ms * 1e3 / (reps * L);
        printf("%-44s %7.2f us/launch  %6.2f TB/s of live rows   max|diff vs V0| %.1e\n", name, us, bytes / us / 1e6, err);
    };
    bench("V0 thread/key, on demand", v0_kernel);
    if (T <= 512) { bench("V1 thread/key, preload blind", v13_kernel<2, 0>); bench("V3 thread/key, kv_len first, live rows", v13_kernel<2, 1>); bench("V5 thread/key, kv_len first, clamped rows", v13_kernel<2, 2>); }
    else { bench("V1 thread/key, preload blind", v13_kernel<4, 0>); bench("V3 thread/key, kv_len first, live rows", v13_kernel<4, 1>); bench("V5 thread/key, kv_len first, clamped rows", v13_kernel<4, 2>); }
    if (T <= 256) { bench("V2 4 lanes/row, blind", t2s_batch_attn_kernel<4, true>); bench("V4 4 lanes/row, kv_len first", t2s_batch_attn_kernel<4, false>); }
    else if (T <= 512) { bench("V2 4 lanes/row, blind", t2s_batch_attn_kernel<8, true>); bench("V4 4 lanes/row, kv_len first", t2s_batch_attn_kernel<8, false>); }
    else { bench("V2 4 lanes/row, blind", t2s_batch_attn_kernel<16, true>); bench("V4 4 lanes/row, kv_len first", t2s_batch_attn_kernel<16, false>); }
    return 0;
}
