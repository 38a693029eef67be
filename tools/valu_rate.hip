// issue rate of a few VALU instructions on gfx950 with 4 waves per SIMD (one 1024-thread block)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int OP>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters) {
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    unsigned w = 0x3f803f80u, x = 0x3f003f00u;
    float f = 0.5f, g = 0.25f;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#define REP8(S) S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
        if (OP == 0) {
#define S0(A) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(A) : "v"(f), "v"(g));
            REP8(S0) REP8(S0)
        } else if (OP == 1) {
#define S1(A) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(A) : "v"(w), "v"(x));
            REP8(S1) REP8(S1)
        } else if (OP == 2) {
#define S2(A) asm volatile("v_exp_f32 %0, %0" : "+v"(A));
            REP8(S2) REP8(S2)
        } else if (OP == 3) {
#define S3(A) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(A) : "v"(w));
            REP8(S3) REP8(S3)
        } else if (OP == 4) {
#define S4(A) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(A) : "v"(w));
            REP8(S4) REP8(S4)
        } else if (OP == 5) {
#define S5(A) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(A));
            REP8(S5) REP8(S5)
        } else if (OP == 6) {
#define S6(A) asm volatile("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(A) : "v"(w));
            REP8(S6) REP8(S6)
        } else if (OP == 7) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&a0) : "v"(*(double*)&a2), "v"(*(double*)&a4));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&a6) : "v"(*(double*)&a2), "v"(*(double*)&a4));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&a0) : "v"(*(double*)&a2), "v"(*(double*)&a4));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&a6) : "v"(*(double*)&a2), "v"(*(double*)&a4));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&a0) : "v"(*(double*)&a2), "v"(*(double*)&a4));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&a6) : "v"(*(double*)&a2), "v"(*(double*)&a4));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&a0) : "v"(*(double*)&a2), "v"(*(double*)&a4));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&a6) : "v"(*(double*)&a2), "v"(*(double*)&a4));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&a0) : "v"(*(double*)&a2), "v"(*(double*)&a4));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&a6) : "v"(*(double*)&a2), "v"(*(double*)&a4));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&a0) : "v"(*(double*)&a2), "v"(*(double*)&a4));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&a6) : "v"(*(double*)&a2), "v"(*(double*)&a4));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&a0) : "v"(*(double*)&a2), "v"(*(double*)&a4));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&a6) : "v"(*(double*)&a2), "v"(*(double*)&a4));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&a0) : "v"(*(double*)&a2), "v"(*(double*)&a4));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&a6) : "v"(*(double*)&a2), "v"(*(double*)&a4));
        } else if (OP == 8) {
#define S8(A) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(A), "+v"(f));
            REP8(S8) REP8(S8)
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * 1024 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float* out; long long* cyc; CK(hipMalloc(&out, 4096 * 4)); CK(hipMalloc(&cyc, 8));
    const char* names[] = {"v_fma_f32", "v_dot2c_f32_bf16", "v_exp_f32", "v_lshlrev_b32", "v_cvt_f32_f16", "v_add_f32_dpp", "v_fma_mix_f32", "v_pk_fma_f32", "v_permlane32_swap"};
    const int iters = 200;
    for (int op = 0; op < 9; ++op) {
        for (int r = 0; r < 2; ++r) {
            switch (op) {
#define C(N) case N: hipLaunchKernelGGL(k<N>, dim3(1), dim3(1024), 0, 0, out, cyc, iters); break;
                C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8)
            }
            CK(hipDeviceSynchronize());
        }
        long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
        printf("%-20s %.2f cycles per instruction per wave with 4 waves per SIMD (=> %.2f per SIMD issue slot)\n", names[op], (double)h / (iters * 16), (double)h / (iters * 16) / 4);
    }
    return 0;
}
