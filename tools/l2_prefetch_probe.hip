// Does a line pulled into an XCD's L2 by one kernel survive the boundary to the next kernel of the same stream?
// The decode step's blocks pull 80-235 KB each from the Infinity Cache at the per-CU miss rate while most CUs idle; if the
// answer is yes, idle CUs of the same XCD could fetch the NEXT launch's weights ahead of it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/l2_prefetch_probe.hip -o tools/l2_prefetch_probe
// Launch l: blocks 0..NC-1 ("consumers", 1024 threads) each read KB_PER_BLOCK of layer l's region, all loads issued at entry;
// blocks NC.. ("helpers") read the consumers' regions of layer l+1.  mode 0: no helpers; mode 1: helper on the consumer's XCD
// (block id = consumer id mod 8); mode 2: helpers shifted to the next XCD (control: same traffic, wrong L2).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int NT = 1024;

template <int NL>   // 16-byte loads per thread
__global__ __launch_bounds__(NT) void probe(const u32x4* __restrict__ W, size_t layer_u4, size_t block_u4, int layer, int n_layers, int NC, int HPC,
                                            int shift, unsigned* __restrict__ sink, long long* __restrict__ cyc) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b < NC) {
        const long long t0 = clock64();
        const u32x4* p = W + (size_t)layer * layer_u4 + (size_t)b * block_u4;
        u32x4 v[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) v[i] = p[(size_t)i * NT + tid];
        unsigned s = 0;
#pragma unroll
        for (int i = 0; i < NL; ++i) s += v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
        if (s == 0x12345678u) sink[b] = s;
        if (tid == 0 && cyc) cyc[(size_t)layer * NC + b] = clock64() - t0;
        return;
    }
    // helper: hb = 8 q + x runs on XCD x (NC is a multiple of 8); it serves consumer c with c % 8 == (x - shift) % 8
    const int hb = b - NC;
    const int x = hb & 7, q = hb >> 3;                  // q in [0, HPC * NC / 8)
    const int cons_per_xcd = NC / 8;
    const int c = ((x - shift) & 7) + 8 * (q % cons_per_xcd);
    const int part = q / cons_per_xcd;                  // [0, HPC)
    const int nl = (layer + 1) % n_layers;
    const u32x4* p = W + (size_t)nl * layer_u4 + (size_t)c * block_u4;
    const size_t per = (size_t)NL * NT / HPC;           // u32x4 per helper
    unsigned s = 0;
    for (size_t i = tid; i < per; i += NT) { const u32x4 v = p[part * per + i]; s += v[0] ^ v[3]; }
    if (s == 0x12345678u) sink[b] = s;
}

template <int NL>
void run(int NC, int HPC, int n_layers, size_t layer_bytes) {
    const size_t block_u4 = (size_t)NL * NT, layer_u4 = layer_bytes / 16;
    if ((size_t)NC * block_u4 > layer_u4) { printf("layer too small\n"); return; }
    u32x4* W; CK(hipMalloc(&W, layer_bytes * n_layers)); CK(hipMemset(W, 1, layer_bytes * n_layers));
    unsigned* sink; CK(hipMalloc(&sink, 4096 * 4));
    long long* cyc; CK(hipMalloc(&cyc, sizeof(long long) * n_layers * NC));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        const int nh = mode ? NC * HPC : 0, shift = mode == 2 ? 1 : 0;
        // capture one pass over the layers as a graph (what the decode step is)
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < n_layers; ++l)
            hipLaunchKernelGGL(probe<NL>, dim3(NC + nh), dim3(NT), 0, st, W, layer_u4, block_u4, l, n_layers, NC, HPC, shift, sink, cyc);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e0, st));
        const int reps = 50;
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<long long> h((size_t)n_layers * NC);
        CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        double mean = 0; long long mx = 0;
        for (auto v : h) { mean += v; mx = v > mx ? v : mx; }
        mean /= h.size();
        printf("  mode %d (%s): %.2f us per launch; consumer block %.0f cycles mean, %lld max (%d KB per block, %d consumers, %d helpers)\n", mode,
               mode == 0 ? "no helpers" : (mode == 1 ? "helpers on the consumer's XCD" : "helpers on the next XCD"), ms * 1e3 / reps / n_layers, mean, mx,
               (int)(block_u4 * 16 / 1024), NC, nh);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    CK(hipFree(W)); CK(hipFree(sink)); CK(hipFree(cyc));
}

int main(int argc, char** argv) {
    const int NC = argc > 1 ? atoi(argv[1]) : 16;
    const int HPC = argc > 2 ? atoi(argv[2]) : 8;
    const int kb = argc > 3 ? atoi(argv[3]) : 208;
    const int n_layers = argc > 4 ? atoi(argv[4]) : 24;
    const size_t layer_bytes = (size_t)(argc > 5 ? atoi(argv[5]) : 8) << 20;   // consumers touch NC x kb of each layer's region
    printf("NC=%d HPC=%d: %d layers, %.0f MB touched per pass, buffer %.0f MB\n", NC, HPC, n_layers, n_layers * (double)NC * kb / 1024, n_layers * (double)layer_bytes / 1048576);
    if (kb == 208) run<13>(NC, HPC, n_layers, layer_bytes);
    else if (kb == 64) run<4>(NC, HPC, n_layers, layer_bytes);
    else if (kb == 128) run<8>(NC, HPC, n_layers, layer_bytes);
    else printf("kb must be 64 / 128 / 208\n");
    return 0;
}
