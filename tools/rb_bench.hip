// Stand-alone check + timing of rbfuse (csrc/rbfuse.h): one Generator stage (three ResBlock1 branches + mean) at C = 16 / 32.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gsv-tts-lite_amd/csrc tools/rb_bench.hip -o tools/rb_bench
//   rb_bench C n_rows [creal]     small n_rows (<= 20000): every output row is compared with a CPU restatement that rounds
//                                 to bf16 where the kernel stores bf16; large n_rows: timing only
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "rbfuse.h"

using namespace gsv;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float b2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static float rb(float f) { return b2f(f2b(f)); }
static float lr(float v, float s) { return fmaxf(v, v * s); }
static uint32_t rng = 12345u;
static float urand() { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) & 0xffff) / 65536.0f - 0.5f; }

template <int C>
int run(int n_rows, int creal) {
    using S = RbShape<C>;
    const int ks[3] = {3, 7, 11}, dil[3] = {1, 3, 5};
    const float slope = 0.1f;
    std::vector<std::vector<float>> w(18), b(18);
    for (int c = 0; c < 18; ++c) {
        const int K = ks[c / 6];
        w[c].resize((size_t)creal * creal * K);
        b[c].resize(creal);
        const float sc = 1.2f / sqrtf((float)creal * K);
        for (auto& v : w[c]) v = rb(urand() * 2 * sc * 1.7f);
        for (auto& v : b[c]) v = urand() * 0.1f;
    }
    const int ld = C;
    std::vector<uint16_t> x((size_t)n_rows * ld, 0);
    for (int r = 0; r < n_rows; ++r)
        for (int c = 0; c < creal; ++c) x[(size_t)r * ld + c] = f2b(urand() * 3.f);
    // device
    RbPackArgs p;
    memset(&p, 0, sizeof(p));
    std::vector<float*> dw(18), db(18);
    int ofs = 0;
    for (int br = 0; br < 3; ++br) { p.k[br] = ks[br]; p.wofs[br] = ofs; ofs += 6 * S::steps(ks[br]) * S::HV; }
    for (int c = 0; c < 18; ++c) {
        CK(hipMalloc(&dw[c], w[c].size() * 4)); CK(hipMemcpy(dw[c], w[c].data(), w[c].size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&db[c], b[c].size() * 4)); CK(hipMemcpy(db[c], b[c].data(), b[c].size() * 4, hipMemcpyHostToDevice));
        p.w[c] = dw[c]; p.b[c] = db[c];
    }
    p.creal = creal;
    CK(hipMalloc(&p.W, (size_t)ofs * 64 * 16));
    CK(hipMalloc(&p.B, 18 * C * 4));
    hipLaunchKernelGGL(rbfuse_pack_kernel<C>, dim3(18), dim3(256), 0, 0, p);
    CK(hipDeviceSynchronize());
    RbFuseArgs a;
    memset(&a, 0, sizeof(a));
    bf16_t *dx, *dy;
    CK(hipMalloc(&dx, x.size() * 2)); CK(hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&dy, x.size() * 2)); CK(hipMemset(dy, 0xff, x.size() * 2));
    a.X = dx; a.Y = dy; a.W = p.W; a.B = p.B; a.ld = ld; a.n_rows = n_rows; a.slope = slope;
    for (int i = 0; i < 3; ++i) { a.wofs[i] = p.wofs[i]; a.dil[i] = dil[i]; }
    auto kern = rbfuse_kernel<C>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS));
    const int ntiles = (n_rows + S::BN - 1) / S::BN;
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(512), S::LDS, 0, a);
    CK(hipDeviceSynchronize());
    std::vector<uint16_t> y(x.size());
    CK(hipMemcpy(y.data(), dy, y.size() * 2, hipMemcpyDeviceToHost));
    if (n_rows <= 20000) {
        // CPU restatement: whole sequence, channels-last fp32 holding bf16-rounded values
        std::vector<float> sum((size_t)n_rows * creal, 0.f);
        for (int br = 0; br < 3; ++br) {
            const int K = ks[br], hk = (K - 1) / 2;
            std::vector<float> xc((size_t)n_rows * creal), xl((size_t)n_rows * creal), t((size_t)n_rows * creal);
            for (int r = 0; r < n_rows; ++r)
                for (int c = 0; c < creal; ++c) xc[(size_t)r * creal + c] = b2f(x[(size_t)r * ld + c]);
            for (int pr = 0; pr < 3; ++pr) {
                const float* w1 = w[br * 6 + pr * 2].data(); const float* b1 = b[br * 6 + pr * 2].data();
                const float* w2 = w[br * 6 + pr * 2 + 1].data(); const float* b2 = b[br * 6 + pr * 2 + 1].data();
                for (size_t i = 0; i < xc.size(); ++i) xl[i] = rb(lr(xc[i], slope));
                const int d = dil[pr];
                for (int r = 0; r < n_rows; ++r)
                    for (int o = 0; o < creal; ++o) {
                        float acc = 0.f;
                        for (int tp = 0; tp < K; ++tp) {
                            const int rr = r + (tp - hk) * d;
                            if (rr < 0 || rr >= n_rows) continue;
                            for (int ci = 0; ci < creal; ++ci) acc += w1[((size_t)o * creal + ci) * K + tp] * xl[(size_t)rr * creal + ci];
                        }
                        t[(size_t)r * creal + o] = rb(lr(acc + b1[o], slope));
                    }
                std::vector<float> xn(xc.size());
                for (int r = 0; r < n_rows; ++r)
                    for (int o = 0; o < creal; ++o) {
                        float acc = 0.f;
                        for (int tp = 0; tp < K; ++tp) {
                            const int rr = r + (tp - hk);
                            if (rr < 0 || rr >= n_rows) continue;
                            for (int ci = 0; ci < creal; ++ci) acc += w2[((size_t)o * creal + ci) * K + tp] * t[(size_t)rr * creal + ci];
                        }
                        xn[(size_t)r * creal + o] = rb((acc + b2[o]) + xc[(size_t)r * creal + o]);
                    }
                xc.swap(xn);
            }
            for (size_t i = 0; i < sum.size(); ++i) sum[i] = br == 0 ? xc[i] : sum[i] + xc[i];
        }
        double maxd = 0, meand = 0, maxv = 0;
        size_t nbad = 0;
        for (int r = 0; r < n_rows; ++r)
            for (int c = 0; c < C; ++c) {
                const float ref = c < creal ? rb(sum[(size_t)r * creal + c] / 3.0f) : 0.f;
                const float got = b2f(y[(size_t)r * ld + c]);
                const double dlt = fabs((double)ref - got);
                if (!(dlt <= 0.05 * (1 + fabs(ref)))) { if (nbad < 8) printf("  bad row %d ch %d: got %g ref %g\n", r, c, got, ref); ++nbad; }
                maxd = std::max(maxd, dlt); meand += dlt; maxv = std::max(maxv, (double)fabs(ref));
            }
        printf("C=%d creal=%d n_rows=%d tiles=%d: max |diff| %.3g  mean %.3g  (max |ref| %.3g)  bad %zu\n", C, creal, n_rows, ntiles, maxd,
               meand / ((double)n_rows * C), maxv, nbad);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(ntiles), dim3(512), S::LDS, 0, a);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(ntiles), dim3(512), S::LDS, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    const double flops = 2.0 * 126 * (double)C * C * n_rows;
    {
        long long* dbg; CK(hipMalloc(&dbg, 32 * 8)); CK(hipMemset(dbg, 0, 32 * 8));
        a.dbg = dbg;
        hipLaunchKernelGGL(kern, dim3(ntiles), dim3(512), S::LDS, 0, a);
        CK(hipDeviceSynchronize());
        long long h[32]; CK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
        a.dbg = nullptr;
        printf("  stamps (cycles): prologue %lld |", h[1] - h[0]);
        for (int br = 0; br < 3; ++br) {
            printf(" br%d stage %lld passes", br, h[br * 10 + 2] - h[br * 10 + 1]);
            for (int i = 2; i < 8; ++i) printf(" %lld", h[br * 10 + i + 1] - h[br * 10 + i]);
            printf(" sum %lld |", h[br * 10 + 9] - h[br * 10 + 8]);
        }
        printf(" store %lld  total %lld\n", h[31] - h[29], h[31] - h[0]);
    }
    printf("C=%d n_rows=%d tiles=%d LDS %zu: %.1f us per stage  %.1f TF/s (useful)  %.2f TB/s of in+out\n", C, n_rows, ntiles, (size_t)S::LDS, us,
           flops / us * 1e-6, 2.0 * n_rows * C * 2 / us * 1e-6);
    return 0;
}

int main(int argc, char** argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 16;
    const int n = argc > 2 ? atoi(argv[2]) : 5000;
    const int creal = argc > 3 ? atoi(argv[3]) : C;
    if (C == 16) return run<16>(n, creal);
    if (C == 32) return run<32>(n, creal);
    printf("C must be 16 or 32\n");
    return 1;
}
