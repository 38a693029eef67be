// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the GPT-SoVITS hot path.
// Wavefront = 64 lanes everywhere in this tree; nothing here is meant to build for another arch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsv {

typedef uint16_t bf16_t;  // raw bfloat16 bits

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u16x8 = __attribute__((ext_vector_type(8))) unsigned short;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN preserved (same rounding torch uses for float -> bfloat16)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// two floats -> packed bf16 pair, round-to-nearest-even in hardware (v_cvt_pk_bf16_f32, gfx950)
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf16_to_f32(v); }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f32_to_bf16(v); }

// Load N consecutive elements (N*sizeof(T) bytes, 16-byte aligned chunks) and widen to fp32.
template <typename T, int N> struct Ld;
template <int N> struct Ld<float, N> {
    static __device__ __forceinline__ void load(const float* p, float (&o)[N]) {
        static_assert(N % 4 == 0, "N");
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * i);
            o[4 * i + 0] = v[0]; o[4 * i + 1] = v[1]; o[4 * i + 2] = v[2]; o[4 * i + 3] = v[3];
        }
    }
};
template <int N> struct Ld<bf16_t, N> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&o)[N]) {
        static_assert(N % 8 == 0, "N");
#pragma unroll
        for (int i = 0; i < N / 8; ++i) {
            u32x4 v = *reinterpret_cast<const u32x4*>(p + 8 * i);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[8 * i + 2 * j + 0] = __uint_as_float(v[j] << 16);
                o[8 * i + 2 * j + 1] = __uint_as_float(v[j] & 0xffff0000u);
            }
        }
    }
};

// ---- wave64 reductions on the DPP cross-lane network (VALU, a few cycles per step) instead of
// ds_bpermute (an LDS-crossbar round trip of >100 cycles per dependent level).
//   quad_perm(1,0,3,2), quad_perm(2,3,0,1): butterfly inside each quad
//   row_half_mirror, row_mirror: fold 8 and 16 lanes (valid because lower levels are already uniform)
//   row_bcast15 / row_bcast31: carry row totals into the next rows; lane 63 ends with the wave total
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
// every lane of an aligned 16-lane row gets the row's sum / max
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f32<0xB1>(v));
    v = fmaxf(v, dpp_f32<0x4E>(v));
    v = fmaxf(v, dpp_f32<0x141>(v));
    v = fmaxf(v, dpp_f32<0x140>(v));
    return v;
}
// sums over aligned groups of 4 / 8 lanes, result in every lane of the group
__device__ __forceinline__ float quad_sum(float v) {
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    return v;
}
__device__ __forceinline__ float oct_sum(float v) {
    v = quad_sum(v);
    return v + dpp_f32<0x141>(v);
}

__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    // rows 1 and 3 add rows 0 and 2 (row_bcast15, row_mask 0b1010); rows 2,3 add lane 31 (row_bcast31, 0b1100)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, true));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 15));
    const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 47));
    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

// Block-wide reductions for 256-thread (4-wave) blocks; `red` is >= 8 floats of LDS.
// Deterministic: fixed shuffle tree then waves summed in index order.
template <int NW> __device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = red[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) s += red[i];
    return s;
}
template <int NW> __device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = red[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) s = fmaxf(s, red[i]);
    return s;
}

}  // namespace gsv
