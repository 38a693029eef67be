// how long does a kernel wait for its kernel arguments?  entry (s_memtime) -> first use of an argument, for a small and a large
// argument block, direct launches and hipGraph replays, 1024-thread blocks on 16 CUs
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
struct Big { long long* out; const int* p[20]; int v[24]; };      // 8 + 160 + 96 = 264 bytes
struct Small { long long* out; int v; int pad; };
template <typename A>
__global__ __launch_bounds__(1024) void k(A a) {
    const char* kp = (const char*)__builtin_amdgcn_kernarg_segment_ptr();      // already in SGPRs: no load
    long long t0, t1; int x; long long* o;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)\n s_load_dwordx2 %2, %4, 0x0\n s_load_dword %3, %4, %5\n s_waitcnt lgkmcnt(0)\n s_memtime %1\n s_waitcnt lgkmcnt(0)"
                 : "=&s"(t0), "=&s"(t1), "=&s"(o), "=&s"(x) : "s"(kp), "i"((int)sizeof(A) - 4));
    if (threadIdx.x == 0) o[blockIdx.x + x * 0] = t1 - t0;
}
int main() {
    long long* out; CK(hipMalloc(&out, 64 * 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    Big b{}; b.out = out; Small s{}; s.out = out;
    for (int mode = 0; mode < 4; ++mode) {
        const bool big = mode & 1, graph = mode & 2;
        long long h[16]; double acc = 0; long long mx = 0;
        hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
        if (graph) {
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < 48; ++i) { if (big) hipLaunchKernelGGL(k<Big>, dim3(16), dim3(1024), 0, st, b); else hipLaunchKernelGGL(k<Small>, dim3(16), dim3(1024), 0, st, s); }
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        }
        for (int rep = 0; rep < 20; ++rep) {
            if (graph) CK(hipGraphLaunch(ge, st));
            else for (int i = 0; i < 48; ++i) { if (big) hipLaunchKernelGGL(k<Big>, dim3(16), dim3(1024), 0, st, b); else hipLaunchKernelGGL(k<Small>, dim3(16), dim3(1024), 0, st, s); }
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
            if (rep >= 4) for (int i = 0; i < 16; ++i) { acc += h[i]; mx = h[i] > mx ? h[i] : mx; }
        }
        printf("%s args, %s: entry -> arguments usable %.0f cycles mean, %lld max (last launch of 48)\n", big ? "264-byte" : "12-byte", graph ? "hipGraph replay" : "direct launches", acc / (16 * 16), mx);
    }
    return 0;
}
