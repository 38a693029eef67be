// Ping-pong latency of a flag hand-off between two workgroups on MI355X: same XCD vs different XCDs, agent-scope
// (sc1: through the fabric) vs workgroup-scope (sc0: L1 bypass, served by the XCD's L2) atomics.  Workgroup b is
// dispatched to XCD b % 8, so blocks (0, 8) share an XCD and blocks (0, 1) do not.  Bounded spins: no hang.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/handoff_bench.hip -o tools/handoff_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0: agent-scope atomics (loads / stores carry sc1).  MODE 1: L1-bypassing loads (sc0) + plain stores: what an
// XCD-local hand-off would use -- the L1 is write-through, so a plain store lands in the XCD's L2 and an sc0 load
// reads it there; coherent between CUs of ONE XCD only.
__device__ __forceinline__ unsigned ld_sc0(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int MODE> __device__ __forceinline__ unsigned ld_flag(const unsigned* p) {
    if constexpr (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return ld_sc0(p);
}
template <int MODE> __device__ __forceinline__ float ld_pay(const float* p) {
    if constexpr (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return __uint_as_float(ld_sc0(reinterpret_cast<const unsigned*>(p)));
}
template <int MODE> __device__ __forceinline__ void st_pay(float* p, float v) {
    if constexpr (MODE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else asm volatile("global_store_dword %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
}
template <int MODE> __device__ __forceinline__ void st_flag(unsigned* p, unsigned v) {
    if constexpr (MODE == 0) __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    else asm volatile("s_waitcnt vmcnt(0)\n\tglobal_store_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
}

template <int SCOPE>
__global__ void pingpong(unsigned* flag, float* payload, int partner, int rounds, long long* cycles, int* xcc) {
    const int b = blockIdx.x;
    if (b != 0 && b != partner) return;
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[b == 0 ? 0 : 1] = (int)(id & 0xf);
    }
    const bool first = b == 0;
    long long t0 = 0;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        const unsigned want = 2u * r + (first ? 0u : 1u);      // first waits for even, second for odd
        if (!(first && r == 0)) {
            if (threadIdx.x == 0) {
                unsigned spins = 0;
                while (ld_flag<SCOPE>(flag) < want && ++spins < (1u << 14)) { __builtin_amdgcn_s_sleep(1); }
            }
            __syncthreads();
            acc += ld_pay<SCOPE>(payload + threadIdx.x);   // fresh 1 KiB of payload
        }
        if (first && r == 1) t0 = clock64();
        st_pay<SCOPE>(payload + threadIdx.x, acc + 1.f);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) st_flag<SCOPE>(flag, want + 1u);
    }
    if (first && threadIdx.x == 0) { cycles[0] = clock64() - t0; cycles[1] = (long long)acc; }
}

template <int SCOPE>
int run(const char* name, int partner) {
    unsigned* flag; float* payload; long long* cyc; int* xcc;
    CK(hipMalloc(&flag, 4)); CK(hipMalloc(&payload, 1024)); CK(hipMalloc(&cyc, 16)); CK(hipMalloc(&xcc, 8));
    const int rounds = 400;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(flag, 0, 4)); CK(hipMemset(payload, 0, 1024));
        hipLaunchKernelGGL(pingpong<SCOPE>, dim3(16), dim3(256), 0, 0, flag, payload, partner, rounds, cyc, xcc);
        CK(hipDeviceSynchronize());
    }
    long long h[2]; int x[2];
    CK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost));
    // clock64 counts shader clocks (~2.4 GHz under load); 2 hand-offs per round; acc must equal the round count
    printf("%-34s blocks 0,%-2d on XCC %d,%d: %7.0f clk per hand-off (checksum %lld)\n", name, partner, x[0], x[1],
           h[0] / (2.0 * (rounds - 1)), h[1]);
    fflush(stdout);
    return 0;
}

int main() {
    if (run<0>("agent scope (sc1), different XCDs", 1)) return 1;
    if (run<0>("agent scope (sc1), same XCD", 8)) return 1;
    if (run<1>("sc0 loads + sc0 stores, same XCD", 8)) return 1;
    if (run<1>("sc0 loads + sc0 stores, diff XCD", 1)) return 1;   // expected to time out: not coherent across XCDs
    return 0;
}
