// The decode layer's DATA FLOW without its arithmetic: what do the two launches per layer cost when a block only pulls what the
// real block pulls (partial rows the previous launch just wrote on other XCDs + its weights from the Infinity Cache) and
// writes its own partial row?  Separates "bytes per CU" from the dependency chain of the real kernels.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/layer_flow_probe.hip -o tools/layer_flow_probe
//   layer_flow_probe [order 0|1] [NJ 32|64] [half 0|1] [store flavour 0 plain | 1 nt | 2 sc1 | 3 sc0 sc1 | 4 two-byte pieces from every wave, as the decode kernels store]   (flavours with order 0)
// A-launch ("attention"): 16 blocks x 1024 threads: NJ partial rows of Z (1 KB each as half, 2 KB as float) + 171 KB of weights,
//                         writes row b of Y.            F-launch ("ffn"): NJ blocks: 16 rows of Y + 4096/NJ KB of weights, writes
// row b of Z.   order 0: partial rows are issued first (what the kernels do); 1: weights first, partial rows last.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int NT = 1024;

// rows: partial rows to read (row bytes = 64 lanes x 16 B = 1 KB per wave-load; `wpr` wave-loads per row); NW: weight wave-loads per wave
// RED = 1 (F launch of the last-arriver form, NJ 64 half rows): the block writes its row through (sc0 sc1), drains, takes a ticket; the LAST
// arriver reads all 64 rows back (sc0 sc1 loads, 4 per wave in flight) and writes ONE reduced row (2 KB) to `red`; the A launch of that
// form reads 2 KB of `red` instead of 64 KB of partial rows.
template <int PR, int NW, int ORDER, int ST = 0, int RED = 0>   // ST: flavour of the partial-row store (0 plain, 1 nt, 2 sc1, 3 sc0 sc1)
__global__ __launch_bounds__(NT) void flow(const u32x4* __restrict__ part, int wpr, const u32x4* __restrict__ W, size_t w_block_u4,
                                           u32x4* __restrict__ out, int out_wpr, long long* __restrict__ cyc, unsigned* ticket = nullptr,
                                           u32x4* red = nullptr, int nj = 64) {
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const long long t0 = clock64();
    u32x4 p[PR], w[NW];
    const u32x4* wp = W + (size_t)b * w_block_u4;
    if (ORDER == 0) {
#pragma unroll
        for (int i = 0; i < PR; ++i) p[i] = part[(size_t)(wid * PR + i) * 64 + lane];
        __builtin_amdgcn_s_barrier();
        asm volatile("" : : : "memory");
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] = wp[(size_t)(wid * NW + i) * 64 + lane];
    if (ORDER == 1) {
        asm volatile("" : : : "memory");
#pragma unroll
        for (int i = 0; i < PR; ++i) p[i] = part[(size_t)(wid * PR + i) * 64 + lane];
    }
    (void)wpr;
    // first-issued load landed
    if (ORDER == 0) asm volatile("" : "+v"(p[0]) : : "memory"); else asm volatile("" : "+v"(w[0]) : : "memory");
    const long long t1 = clock64();
    u32x4 s = p[0];
#pragma unroll
    for (int i = 1; i < PR; ++i) s += p[i];
    asm volatile("" : "+v"(s) : : "memory");
    const long long t2 = clock64();     // partial rows landed (ORDER 1: everything landed)
#pragma unroll
    for (int i = 0; i < NW; ++i) s += w[i];
    asm volatile("" : "+v"(s) : : "memory");
    const long long t3 = clock64();     // everything landed
    // the block's own partial row(s): wave 0 .. out_wpr-1 write 1 KB each
    if (ST == 4) {
        // what the decode kernels do: every wave writes 32 two-byte results of the block's 1-KiB row, 16 per store instruction
        unsigned short* rowp = reinterpret_cast<unsigned short*>(out + (size_t)b * out_wpr * 64);
        if ((lane & 3) == 0) {
            rowp[wid * 16 + (lane >> 2)] = (unsigned short)s[0];
            rowp[256 + wid * 16 + (lane >> 2)] = (unsigned short)s[1];
        }
    } else if (wid < out_wpr) {
        u32x4* dst = out + ((size_t)b * out_wpr + wid) * 64 + lane;
        if (ST == 0) *dst = s;
        else if (ST == 1) __builtin_nontemporal_store(s, dst);
        else if (ST == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst), "v"(s) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(dst), "v"(s) : "memory");
    }
    if (RED == 1) {
        __shared__ int last;
        asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
        __syncthreads();
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = ((old + 1u) % (unsigned)nj) == 0u;
        }
        __syncthreads();
        if (last) {
            const u32x4* r0 = out + (size_t)(wid * 4) * 64 + lane;      // 64 rows of 1 KB over 16 waves: 4 per wave, all in flight
            u32x4 v0, v1, v2, v3;
            asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %5, off sc0 sc1\n\tglobal_load_dwordx4 %2, %6, off sc0 sc1\n\t"
                         "global_load_dwordx4 %3, %7, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(r0), "v"(r0 + 64), "v"(r0 + 128), "v"(r0 + 192) : "memory");
            __shared__ u32x4 acc[16][64];
            acc[wid][lane] = v0 + v1 + v2 + v3;
            __syncthreads();
            if (wid < 2) {
                u32x4 t = acc[wid][lane];
                for (int i = 2 + wid; i < 16; i += 2) t += acc[i][lane];
                red[wid * 64 + lane] = t;
            }
        }
    }
    if (cyc && b == 0 && (tid == 0 || tid == 960)) {
        long long* c = cyc + (tid ? 4 : 0);
        c[0] = t1 - t0; c[1] = t2 - t0; c[2] = t3 - t0; c[3] = clock64() - t0;
    }
}

int main(int argc, char** argv) {
    const int order = argc > 1 ? atoi(argv[1]) : 0;
    const int NJ = argc > 2 ? atoi(argv[2]) : 64;
    const int half = argc > 3 ? atoi(argv[3]) : 1;
    const int stf = argc > 4 ? atoi(argv[4]) : 0;
    const int reduce = argc > 5 ? atoi(argv[5]) : 0;    // 1: the last-arriver form (NJ 64, half rows)
    const int n_layers = 24;
    const int wpr = half ? 1 : 2;                    // wave-loads (KB) per partial row
    const size_t a_w_u4 = 176 * 64, f_w_u4 = (size_t)(4096 / NJ) * 64;   // per-block weight bytes / 16: 176 KB, 64 or 128 KB
    u32x4 *WA, *WF, *Y, *Z;
    CK(hipMalloc(&WA, n_layers * 16 * a_w_u4 * 16)); CK(hipMemset(WA, 1, n_layers * 16 * a_w_u4 * 16));
    CK(hipMalloc(&WF, n_layers * NJ * f_w_u4 * 16)); CK(hipMemset(WF, 1, n_layers * NJ * f_w_u4 * 16));
    CK(hipMalloc(&Y, 16 * wpr * 1024)); CK(hipMemset(Y, 0, 16 * wpr * 1024));
    CK(hipMalloc(&Z, NJ * wpr * 1024)); CK(hipMemset(Z, 0, NJ * wpr * 1024));
    long long* cyc; CK(hipMalloc(&cyc, 16 * 8)); CK(hipMemset(cyc, 0, 16 * 8));
    unsigned* tickets; CK(hipMalloc(&tickets, n_layers * 256)); CK(hipMemset(tickets, 0, n_layers * 256));
    u32x4* RED; CK(hipMalloc(&RED, 2048)); CK(hipMemset(RED, 0, 2048));
    if (reduce && !(NJ == 64 && half)) { printf("the last-arriver form is written for NJ 64, half rows\n"); return 1; }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // A: NJ*wpr partial wave-loads over 16 waves; 176 weight wave-loads over 16 waves = 11.  F: 16*wpr over 16 waves; 4096/NJ over 16 waves
    auto launchA = [&](int l, long long* c) {
        const u32x4* w = WA + (size_t)l * 16 * a_w_u4;
#define LA(PR) do { if (stf == 0) hipLaunchKernelGGL((flow<PR, 11, 0, 0>), dim3(16), dim3(NT), 0, st, Z, wpr, w, a_w_u4, Y, wpr, c); \
                    else if (stf == 1) hipLaunchKernelGGL((flow<PR, 11, 0, 1>), dim3(16), dim3(NT), 0, st, Z, wpr, w, a_w_u4, Y, wpr, c); \
                    else if (stf == 2) hipLaunchKernelGGL((flow<PR, 11, 0, 2>), dim3(16), dim3(NT), 0, st, Z, wpr, w, a_w_u4, Y, wpr, c); \
                    else if (stf == 3) hipLaunchKernelGGL((flow<PR, 11, 0, 3>), dim3(16), dim3(NT), 0, st, Z, wpr, w, a_w_u4, Y, wpr, c); \
                    else hipLaunchKernelGGL((flow<PR, 11, 0, 4>), dim3(16), dim3(NT), 0, st, Z, wpr, w, a_w_u4, Y, wpr, c); } while (0)
        const int pr = NJ * wpr / 16;
        if (reduce) { hipLaunchKernelGGL((flow<1, 11, 0, 0>), dim3(16), dim3(NT), 0, st, RED, wpr, w, a_w_u4, Y, wpr, c); return; }   // every wave reads 1 KB of the 2 KB row
        if (pr == 2) LA(2); else if (pr == 4) LA(4); else LA(8);
    };
    auto launchF = [&](int l, long long* c) {
        const u32x4* w = WF + (size_t)l * NJ * f_w_u4;
#define LF(PR, NW) do { if (stf == 0) hipLaunchKernelGGL((flow<PR, NW, 0, 0>), dim3(NJ), dim3(NT), 0, st, Y, wpr, w, f_w_u4, Z, wpr, c); \
                        else if (stf == 1) hipLaunchKernelGGL((flow<PR, NW, 0, 1>), dim3(NJ), dim3(NT), 0, st, Y, wpr, w, f_w_u4, Z, wpr, c); \
                        else if (stf == 2) hipLaunchKernelGGL((flow<PR, NW, 0, 2>), dim3(NJ), dim3(NT), 0, st, Y, wpr, w, f_w_u4, Z, wpr, c); \
                        else if (stf == 3) hipLaunchKernelGGL((flow<PR, NW, 0, 3>), dim3(NJ), dim3(NT), 0, st, Y, wpr, w, f_w_u4, Z, wpr, c); \
                        else hipLaunchKernelGGL((flow<PR, NW, 0, 4>), dim3(NJ), dim3(NT), 0, st, Y, wpr, w, f_w_u4, Z, wpr, c); } while (0)
        if (reduce) { hipLaunchKernelGGL((flow<1, 4, 0, 3, 1>), dim3(NJ), dim3(NT), 0, st, Y, wpr, w, f_w_u4, Z, wpr, c, tickets + l * 64, RED, NJ); return; }
        if (NJ == 64) { if (wpr == 1) LF(1, 4); else LF(2, 4); }
        else { if (wpr == 1) LF(1, 8); else LF(2, 8); }
    };
    for (int which = 0; which < 3; ++which) {   // 0: A and F alternating (the layer), 1: only A launches, 2: only F launches
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < n_layers; ++l) {
            if (which != 2) launchA(l, l == n_layers - 1 ? cyc : nullptr);
            if (which != 1) launchF(l, l == n_layers - 1 ? cyc + 8 : nullptr);
        }
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e0, st));
        const int reps = 50;
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long h[16]; CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
        printf("order %d store %d NJ %d %s rows, %s: %.2f us per layer\n", order, stf, NJ, half ? "half" : "float", which == 0 ? "A + F" : (which == 1 ? "A only (partials not fresh)" : "F only (partials not fresh)"),
               ms * 1e3 / reps / n_layers);
        if (which == 0) {
            printf("   A block 0 cycles since entry (first load landed, partial rows landed, all landed, row written): wave 0 %lld %lld %lld %lld | wave 15 %lld %lld %lld %lld\n",
                   h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
            printf("   F block 0:                                                                                    wave 0 %lld %lld %lld %lld | wave 15 %lld %lld %lld %lld\n",
                   h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15]);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
