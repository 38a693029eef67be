// Do two chains of short dependent kernels on two HIP streams run beside each other on gfx950, or does the command
// processor take them one queue at a time?  (The slot loop's steps and a refill's prompt pass are exactly that.)
//   hipcc --offload-arch=gfx950 -O3 tools/stream_overlap.hip -o tools/stream_overlap && tools/stream_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin_kernel(long ticks, int* sink) {      // 100 MHz constant clock: 100 ticks = 1 us
    long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) sink[0] = 1;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Chain { hipStream_t s; int n, grid; long ticks; hipGraphExec_t g = nullptr; };
static void issue(const Chain& c) {
    if (c.g) { for (int r = 0; r < c.n / 100; r++) hipGraphLaunch(c.g, c.s); return; }
    for (int i = 0; i < c.n; i++) spin_kernel<<<c.grid, 256, 0, c.s>>>(c.ticks, nullptr);
}
static int capture(Chain& c) {
    hipGraph_t g;
    CK(hipStreamBeginCapture(c.s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 100; i++) spin_kernel<<<c.grid, 256, 0, c.s>>>(c.ticks, nullptr);
    CK(hipStreamEndCapture(c.s, &g));
    CK(hipGraphInstantiate(&c.g, g, nullptr, nullptr, 0));
    return 0;
}
static double run(const Chain* a, const Chain* b) {
    hipDeviceSynchronize();
    double t0 = now();
    if (a && b) {   // interleave the issue so neither queue runs dry on the host's account
        if (a->g) { issue(*a); issue(*b); }
        else for (int i = 0; i < a->n; i++) { spin_kernel<<<a->grid, 256, 0, a->s>>>(a->ticks, nullptr); if (i < b->n) spin_kernel<<<b->grid, 256, 0, b->s>>>(b->ticks, nullptr); }
    } else issue(a ? *a : *b);
    hipDeviceSynchronize();
    return (now() - t0) * 1e3;
}
int main() {
    int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("stream priority range: least %d greatest %d\n", lo, hi);
    std::vector<hipStream_t> st(10);
    for (int i = 0; i < 8; i++) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&st[8], hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithPriority(&st[9], hipStreamNonBlocking, lo));
    const int N = 2000;
    for (int w = 0; w < 2; w++) { Chain c{st[0], 200, 64, 300}; run(&c, nullptr); }
    struct Case { const char* name; int ia, ib; int ga, gb; bool graph; };
    Case cases[] = {
        {"A = stream 0, B = stream 1 (64 / 16 blocks)", 0, 1, 64, 16, false},
        {"A = stream 0, B = stream 2", 0, 2, 64, 16, false},
        {"A = stream 0, B = stream 3", 0, 3, 64, 16, false},
        {"A = stream 0, B = stream 4", 0, 4, 64, 16, false},
        {"A = stream 0, B = stream 5", 0, 5, 64, 16, false},
        {"A = null stream, B = stream 1", -1, 1, 64, 16, false},
        {"A = high priority, B = stream 1", 8, 1, 64, 16, false},
        {"A = high priority, B = low priority", 8, 9, 64, 16, false},
        {"A = stream 0 from a graph, B = stream 1", 0, 1, 64, 16, true},
        {"A 512 blocks, B 16 blocks", 0, 1, 512, 16, false},
        {"A 512 blocks, B 256 blocks", 0, 1, 512, 256, false},
        {"A = high priority 512 blocks from a graph, B = low 256 blocks", 8, 9, 512, 256, true},
    };
    for (auto& cs : cases) {
        Chain a{cs.ia < 0 ? nullptr : st[cs.ia], N, cs.ga, 400}, b{st[cs.ib], N, cs.gb, 800};   // 4 us and 8 us kernels
        if (cs.graph && capture(a)) return 1;
        double ta = run(&a, nullptr), tb = run(nullptr, &b), tab = run(&a, &b);
        printf("%-64s A alone %7.2f ms  B alone %7.2f ms  both %7.2f ms  (sum %7.2f, max %7.2f)\n", cs.name, ta, tb, tab, ta + tb, ta > tb ? ta : tb);
    }
    return 0;
}
