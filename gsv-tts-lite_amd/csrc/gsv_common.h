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

// round-to-nearest-even (the rounding torch uses for float -> bfloat16), in hardware: v_cvt_pk_bf16_f32 on gfx950 (a NaN
// comes out quiet); the bit-twiddled form cost a compare + branch per value in the decode kernels' serial phases
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    return __builtin_bit_cast(bf16_t, (__bf16)f);
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
// max of two non-NaN values as ONE instruction: fmaxf quiets a signalling NaN first (a v_max x, x per operand), v_med3 with
// +inf as the third operand does not
__device__ __forceinline__ float max_nn(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_inff()); }
__device__ __forceinline__ float row16_max(float v) {
    v = max_nn(v, dpp_f32<0xB1>(v));
    v = max_nn(v, dpp_f32<0x4E>(v));
    v = max_nn(v, dpp_f32<0x141>(v));
    v = max_nn(v, dpp_f32<0x140>(v));
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
    return max_nn(max_nn(a, b), max_nn(c, d));
}

// ---- cross-lane exchanges without the LDS crossbar.  __shfl_xor compiles to ds_bpermute_b32: an LDS instruction
// (>100 cycles per dependent level, and the 16 waves of a decode block share ONE LDS pipe).  gfx950 has
// v_permlane32_swap / v_permlane16_swap (exchange lane halves / odd-even 16-lane rows between two registers) and the
// DPP row rotations, all on the VALU.
// all lanes: v[l] + v[l ^ 32]   and   v[l] + v[l ^ 16]
__device__ __forceinline__ float xor32_sum(float v) {
    const auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}
__device__ __forceinline__ float xor16_sum(float v) {
    const auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}
// halving exchanges: lanes 0-31 (even rows) end with a[l] + a[partner], lanes 32-63 (odd rows) with b[l] + b[partner]
__device__ __forceinline__ float halve32_sum(float a, float b) {
    const auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}
__device__ __forceinline__ float halve16_sum(float a, float b) {
    const auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}
// lanes with bit 3 clear end with a[l] + a[l ^ 8], lanes with bit 3 set with b[l] + b[l ^ 8]   (row_ror:8 == xor 8 in a row)
__device__ __forceinline__ float halve8_sum(float a, float b) {
    const bool hi = (threadIdx.x & 8) != 0;
    const float send = hi ? a : b, keep = hi ? b : a;
    return keep + dpp_f32<0x128>(send);
}
// the partner's value v[l ^ M] for M = 1 .. 32 (float or 32-bit integer payload)
template <int M> __device__ __forceinline__ unsigned lane_xor_u32(unsigned v) {
    static_assert(M == 1 || M == 2 || M == 4 || M == 8 || M == 16 || M == 32, "M");
    if constexpr (M == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
    else if constexpr (M == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);
    else if constexpr (M == 4) {
        const unsigned up = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0xF, true);   // row_shl:4 -> v[l + 4]
        const unsigned dn = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4 -> v[l - 4]
        return (threadIdx.x & 4) ? dn : up;
    } else if constexpr (M == 8) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);
    else if constexpr (M == 16) {
        const auto t = __builtin_amdgcn_permlane16_swap(v, v, false, false);   // t[0] rows (0,0,2,2), t[1] rows (1,1,3,3)
        return (threadIdx.x & 16) ? t[0] : t[1];
    } else {
        const auto t = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // t[0] low half twice, t[1] high half twice
        return (threadIdx.x & 32) ? t[0] : t[1];
    }
}
template <int M> __device__ __forceinline__ float lane_xor(float v) { return __uint_as_float(lane_xor_u32<M>(__float_as_uint(v))); }
template <int M> __device__ __forceinline__ int lane_xor(int v) { return (int)lane_xor_u32<M>((unsigned)v); }

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
