// Stand-alone check + timing of cgemm (csrc/cgemm.h): a 3-branch launch (k = 11, 7, 3) of C x C convs over n_rows rows.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gsv-tts-lite_amd/csrc tools/cg_bench.hip -o tools/cg_bench
//   cg_bench C n_rows [dil] [residual 0/1]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "cgemm.h"

using namespace gsv;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float b2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static uint32_t rng = 777u;
static float urand() { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) & 0xffff) / 65536.0f - 0.5f; }

template <int C, int BN, int BM, int RT = 2>
int run(int n_rows, int dil, int use_res) {
    using S = CgShape<C, BN, BM, RT>;
    const int ks[3] = {11, 7, 3};
    const float in_slope = 0.1f, out_slope = use_res ? 1.0f : 0.1f;
    std::vector<std::vector<float>> w(3), bs(3);
    std::vector<uint16_t> x((size_t)n_rows * C), res((size_t)n_rows * C);
    for (auto& v : x) v = f2b(urand() * 2.f);
    for (auto& v : res) v = f2b(urand() * 2.f);
    CGemmArgs a;
    memset(&a, 0, sizeof(a));
    bf16_t *dx, *dr, *dy[3];
    CK(hipMalloc(&dx, x.size() * 2)); CK(hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&dr, x.size() * 2)); CK(hipMemcpy(dr, res.data(), x.size() * 2, hipMemcpyHostToDevice));
    const uint4* dw[3]; const float* db[3];
    for (int b = 0; b < 3; ++b) {
        const int K = ks[b];
        w[b].resize((size_t)C * C * K); bs[b].resize(C);
        const float sc = 1.0f / sqrtf((float)C * K);
        for (auto& v : w[b]) v = b2f(f2b(urand() * 3.4f * sc));
        for (auto& v : bs[b]) v = urand() * 0.2f;
        float* wf; CK(hipMalloc(&wf, w[b].size() * 4)); CK(hipMemcpy(wf, w[b].data(), w[b].size() * 4, hipMemcpyHostToDevice));
        bf16_t* wp; CK(hipMalloc(&wp, w[b].size() * 2));
        hipLaunchKernelGGL(cgemm_pack_kernel<C>, dim3(1024), dim3(256), 0, 0, wf, wp, K);
        float* bd; CK(hipMalloc(&bd, C * 4)); CK(hipMemcpy(bd, bs[b].data(), C * 4, hipMemcpyHostToDevice));
        dw[b] = (const uint4*)wp; db[b] = bd;
        CK(hipMalloc(&dy[b], x.size() * 2)); CK(hipMemset(dy[b], 0xff, x.size() * 2));
    }
    CK(hipDeviceSynchronize());
    const int tiles = ((n_rows + S::BM - 1) / S::BM) * S::TN;
    a.X0 = a.X1 = a.X2 = dx; a.W0 = dw[0]; a.W1 = dw[1]; a.W2 = dw[2]; a.b0 = db[0]; a.b1 = db[1]; a.b2 = db[2];
    a.R0 = a.R1 = a.R2 = use_res ? dr : nullptr; a.Y0 = dy[0]; a.Y1 = dy[1]; a.Y2 = dy[2];
    // K split per branch (CG_SPLIT="3,2,1": the 11-tap branch's chunks over three consecutive blocks, the 7-tap branch's over two)
    int ns[3] = {1, 1, 1};
    if (const char* e = getenv("CG_SPLIT")) sscanf(e, "%d,%d,%d", &ns[0], &ns[1], &ns[2]);
    for (int b = 0; b < 3; ++b) if (ns[b] < 1 || S::NCH % ns[b]) { printf("CG_SPLIT: %d does not divide %d chunks\n", ns[b], S::NCH); return 1; }
    a.k0 = 11; a.k1 = 7; a.k2 = 3; a.d0 = a.d1 = a.d2 = dil; a.nb0 = tiles * ns[0]; a.nb1 = tiles * ns[1];
    a.ns0 = ns[0]; a.ns1 = ns[1]; a.ns2 = ns[2];
    const int nblocks = tiles * (ns[0] + ns[1] + ns[2]);
    {
        const size_t tile_f = (size_t)S::BM * BN;
        CK(hipMalloc(&a.part, sizeof(float) * tile_f * 2 * 3 * tiles + 16));
        CK(hipMalloc(&a.flag, sizeof(int) * 2 * 3 * tiles + 16)); CK(hipMemset(a.flag, 0, sizeof(int) * 2 * 3 * tiles + 16));
    }
    a.ld = C; a.n_rows = n_rows; a.in_slope = in_slope; a.out_slope = out_slope;
    auto kern = cgemm_kernel<C, BN, BM, RT>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS));
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(S::NT), S::LDS, 0, a);
    CK(hipDeviceSynchronize());
    // sampled CPU check
    double maxd = 0; size_t nbad = 0;
    for (int b = 0; b < 3; ++b) {
        std::vector<uint16_t> y(x.size());
        CK(hipMemcpy(y.data(), dy[b], y.size() * 2, hipMemcpyDeviceToHost));
        const int K = ks[b], hk = (K - 1) / 2;
        for (int s = 0; s < 600; ++s) {
            int r = s < 40 ? s : (s < 80 ? n_rows - 1 - (s - 40) : (int)(((uint64_t)(rng = rng * 1664525u + 1013904223u)) % n_rows));
            if (r < 0 || r >= n_rows) continue;
            const int co = (int)((rng >> 7) % C);
            double acc = 0;
            for (int t = 0; t < K; ++t) {
                const int rr = r + (t - hk) * dil;
                if (rr < 0 || rr >= n_rows) continue;
                for (int ci = 0; ci < C; ++ci) {
                    float xv = b2f(x[(size_t)rr * C + ci]);
                    xv = b2f(f2b(fmaxf(xv, xv * in_slope)));
                    acc += (double)w[b][((size_t)co * C + ci) * K + t] * xv;
                }
            }
            float v = (float)acc + bs[b][co];
            if (out_slope != 1.0f) v = fmaxf(v, v * out_slope);
            if (use_res) v += b2f(res[(size_t)r * C + co]);
            const float got = b2f(y[(size_t)r * C + co]);
            const double dlt = fabs(got - v);
            maxd = std::max(maxd, dlt);
            if (!(dlt <= 0.02 * (1 + fabs(v)))) { if (nbad < 6) printf("  bad br %d row %d ch %d: got %g ref %g\n", b, r, co, got, v); ++nbad; }
        }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(nblocks), dim3(S::NT), S::LDS, 0, a);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(nblocks), dim3(S::NT), S::LDS, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    {
        long long* dbg; CK(hipMalloc(&dbg, 64 * 8)); CK(hipMemset(dbg, 0, 64 * 8));
        a.dbg = dbg;
        hipLaunchKernelGGL(kern, dim3(nblocks), dim3(S::NT), S::LDS, 0, a);
        CK(hipDeviceSynchronize());
        long long h[64]; CK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
        a.dbg = nullptr;
        const int nit = (C / 64) * 11;
        printf("  block 0 (k = 11) cycles: prologue %lld | iterations", h[1] - h[0]);
        for (int i = 0; i < nit && i < 48; ++i) printf(" %lld", h[3 + i] - h[2 + i]);
        printf(" | epilogue %lld | total %lld\n", h[3 + nit] - h[2 + nit], h[3 + nit] - h[0]);
    }
    const double flops = 2.0 * 21 * (double)C * C * n_rows;
    printf("split %d/%d/%d  C=%d BN=%d BM=%d RT=%d n_rows=%d dil=%d res=%d blocks=%d LDS %zu: %.1f us per launch  %.1f TF/s   max |diff| %.3g  bad %zu\n", ns[0], ns[1], ns[2], C, BN, BM, RT, n_rows, dil, use_res,
           nblocks, (size_t)S::LDS, us, flops / us * 1e-6, maxd, nbad);
    return 0;
}

int main(int argc, char** argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 256;
    const int n = argc > 2 ? atoi(argv[2]) : 5000;
    const int dil = argc > 3 ? atoi(argv[3]) : 5;
    const int res = argc > 4 ? atoi(argv[4]) : 0;
    const int bm = argc > 5 ? atoi(argv[5]) : 128;
    if (bm == 128) {
        if (C == 256) return run<256, 128, 128>(n, dil, res);
        if (C == 192) return run<192, 192, 128>(n, dil, res);
        if (C == 384) return run<384, 192, 128>(n, dil, res);
        if (C == 128) return run<128, 128, 128>(n, dil, res);
    } else if (bm == 2564) {     // 256 rows, 4 waves of 128 rows
        if (C == 256) return run<256, 128, 256, 4>(n, dil, res);
        if (C == 128) return run<128, 128, 256, 4>(n, dil, res);
    } else {
        if (C == 256) return run<256, 128, 256>(n, dil, res);
        if (C == 192) return run<192, 192, 256>(n, dil, res);
        if (C == 384) return run<384, 192, 256>(n, dil, res);
        if (C == 128) return run<128, 128, 256>(n, dil, res);
    }
    printf("C must be 128 / 192 / 256 / 384\n");
    return 1;
}
