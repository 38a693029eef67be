// GPT decode-step kernels for gfx950: one token per sequence through a post-LN block stack.
//
// Reference semantics: T2SBlock.decode_next_token, gsv_tts/GPT_SoVITS/GPT/t2s_model.py:67-105
//   qkv = x Wqkv^T + b ; K/V appended at kv_len[b] ; causal attention over [0, kv_len[b]] ;
//   x = LN1(x + attn Wo^T + bo) ; x = LN2(x + W2 relu(W1 x + b1) + b2)
//
// MI355X mapping (DESIGN.md "GPT decode"): the step is bandwidth/latency bound (weights
// stream once per token), so each layer is TWO kernels, split at the two all-to-all points,
// and every normalisation/residual lives in the prologue of the consumer:
//
//   attn kernel  grid (16 heads, B): [prev FFN partial-sum + b2 + residual + LN2] -> x ;
//                this head's 96 QKV rows (wave-per-row, 16 B/lane coalesced weight stream) ;
//                KV append ; single-query attention over the head's contiguous [T][32] panel ;
//                out_proj restricted to this head's 32 input columns -> partial y[h][512]
//   ffn kernel   grid (32 slices, B): [sum of 16 head partials + bo + residual + LN1] -> x1 ;
//                64 hidden units of W1 (+ReLU) ; their 64 columns of W2 -> partial z[j][512]
//
// Partials are summed in fixed index order by the consumer => bit-reproducible run to run
// (no atomics), which greedy-token parity against the CPU oracle depends on.
// Weight panels are pre-packed at load (t2s_pack.h) so every wave instruction reads 1 KiB
// of consecutive bytes.
#pragma once
#include "gsv_common.h"

namespace gsv {

constexpr int kD = 512;        // hidden
constexpr int kH = 16;         // heads
constexpr int kDh = 32;        // head dim
constexpr int kF = 2048;       // MLP hidden
constexpr int kNJ = 32;        // FFN slices (blocks) per sequence
constexpr int kFJ = kF / kNJ;  // hidden units per slice
// <= kFineMaxB sequences on a bf16 handle: 64 slices of 32 hidden units.  An FFN block then pulls 80 KB instead of 144 KB at the
// per-CU fabric rate, an attention block 32 KB more of (half) partial rows: -5 % on the step at 1-4 sequences, +3..10 % from 8 on
// (profiles/r03_ffn_64_slices.txt), hence the switch.  The slice count is part of the arithmetic (each slice partial is rounded
// to half): gsv_t2s_ffn_slices reports it and the oracle sums the same slices.
constexpr int kNJFine = 64;
constexpr int kFineMaxB = 4;
// fp32 handles (round 5): the same switch.  Their blocks pull twice the bytes (an FFN block 288 KB at 32 slices, 144 KB at 64), and a
// launch costs ~2.5 us + bytes / 60 GB/s; the fp32 sum order is not the reference's either way (it sums 2 048 products in one dot).
// GSV_F32_FINE=0 keeps 32 slices (A/B).
inline bool f32_fine() { static const bool on = !(getenv("GSV_F32_FINE") && atoi(getenv("GSV_F32_FINE")) == 0); return on; }
template <typename WT> inline int ffn_slices(int B) { return (sizeof(WT) == 2 || f32_fine()) && B <= kFineMaxB ? kNJFine : kNJ; }
constexpr int kNP = 16;        // logits slices per sequence
constexpr float kEps = 1e-5f;

struct TokPart {
    float v;
    int idx;
};

// eos_at[slot] also goes to a host-mapped mirror when the caller registered one (gsv_t2s_set_eos_mirror): the host reads the
// flag from its own memory after an event instead of enqueuing a device-to-host copy between the decode windows
__device__ __forceinline__ void eos_publish(int32_t* eos_host, int slot, int value) {
    if (eos_host != nullptr) __hip_atomic_store(eos_host + slot, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- shared pieces ---------------------------------------------------------------------------
//
// Latency discipline.  The step is a chain of ~50 dependent kernels and, at batch 1, each kernel
// runs on a handful of CUs, so what matters is the length of the dependent chain inside a kernel:
//  * every global load whose ADDRESS does not depend on this kernel's activations -- partial sums,
//    weight rows, K/V rows, out-proj panel, biases -- is issued at kernel entry into registers, in
//    the order it will be consumed (loads retire in order), and an opaque asm pins "all loads
//    issued, then arithmetic" (hipcc otherwise sinks the loads next to their first use);
//  * the instruction stream per wave is what the kernel's wall time is made of once the loads
//    overlap (one wave issues ~1 VALU op per 4-5 cycles), so a block is 16 waves (1024 threads):
//    4 waves per SIMD share the rows/keys, each wave's stream is a quarter of a 256-thread block's.

// optional phase timestamps (bring-up aid): in block (0,0) the first lane of wave 0 writes clock64() into dbg[slot], the
// first lane of the last wave into dbg[16 + slot] (32 slots)
__device__ __forceinline__ void stamp(unsigned long long* dbg, int slot) {
    if (dbg != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x == 0 || threadIdx.x == 960)) dbg[slot + (threadIdx.x ? 16 : 0)] = clock64();
}

constexpr int kNT = 1024;       // threads per decode block
constexpr int kNW = kNT / 64;   // 16 waves

using raw16 = u32x4;  // 16 bytes of operands, as loaded

template <typename WT, int N> struct Unpack;           // raw16 -> float[N]
template <> struct Unpack<float, 4> {
    static __device__ __forceinline__ void run(const raw16& r, float (&o)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = __uint_as_float(r[i]);
    }
};
template <> struct Unpack<bf16_t, 8> {
    static __device__ __forceinline__ void run(const raw16& r, float (&o)[8]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[2 * j] = __uint_as_float(r[j] << 16);
            o[2 * j + 1] = __uint_as_float(r[j] & 0xffff0000u);
        }
    }
};

// The partial 512-vectors that cross a kernel boundary (16 head partials of the out-proj, 32 slice partials of W2): fp32 on
// fp32 handles (the parity mode sums exactly what the reference's GEMV sums, in slice order); IEEE half on bf16 handles -- a
// consumer block reads 16 / 32 of them per sequence and that read is a third to a half of what it pulls.
// Half, not bf16: 11 significant bits against 8, so a rounded partial carries an eighth of a bf16 operand's rounding error;
// the values are O(1) sums behind a LayerNorm, far inside half's range.  The bf16-mode oracle rounds the same slices.
template <typename WT> struct PartOf { using T = float; };
template <> struct PartOf<bf16_t> { using T = _Float16; };
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

template <typename WT> struct Geo {
    static constexpr int EPL = 16 / sizeof(WT);  // elements per 16-byte lane load
    static constexpr int CPR = kD / EPL / 64;    // 16-byte chunks per lane for one 512-wide row (1 bf16, 2 f32)
    using PT = typename PartOf<WT>::T;
};

__device__ __forceinline__ raw16 ldg16(const void* p) { return *reinterpret_cast<const raw16*>(p); }
// streamed once per step and NOT to be kept in the Infinity Cache (see the NT note at t2s_attn_kernel)
template <bool NT> __device__ __forceinline__ raw16 ldg16w(const void* p) {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const raw16*>(p)); else return ldg16(p);
}
// Weight loads keep the DEFAULT cache policy: measured on MI355X, the non-temporal hint (global_load ... nt)
// made the bs=1 step 10 % slower (0.404 vs 0.370 ms/token) -- the 152 MB weight set lives in the 256 MiB
// Infinity Cache between tokens and nt lines are not retained there.

// dot of one lane's slice of a 512-wide weight row (CPR chunks) with the lane's activations.
// bf16: lane owns x[lane*8 .. +7]; f32: chunks c=0,1 own x[c*256 + lane*4 .. +3].
template <typename WT>
__device__ __forceinline__ float row_dot(const raw16 (&w)[Geo<WT>::CPR], const float (&xr)[8]) {
    constexpr int EPL = Geo<WT>::EPL;
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < Geo<WT>::CPR; ++c) {
        float wv[EPL];
        Unpack<WT, EPL>::run(w[c], wv);
#pragma unroll
        for (int i = 0; i < EPL; ++i) a = fmaf(wv[i], xr[c * EPL + i], a);
    }
    return a;
}
template <typename WT, bool NT = false>
__device__ __forceinline__ void row_load(const WT* row, raw16 (&w)[Geo<WT>::CPR]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < Geo<WT>::CPR; ++c) w[c] = ldg16w<NT>(row + c * (kD / Geo<WT>::CPR) + lane * Geo<WT>::EPL);
}
template <typename WT>
__device__ __forceinline__ void lane_x(const float* xs, float (&xr)[8]) {
    const int lane = threadIdx.x & 63;
    constexpr int EPL = Geo<WT>::EPL;
#pragma unroll
    for (int c = 0; c < Geo<WT>::CPR; ++c)
#pragma unroll
        for (int i = 0; i < EPL; ++i) xr[c * EPL + i] = xs[c * (kD / Geo<WT>::CPR) + lane * EPL + i];
}

// ---- bf16 handles: dots on v_dot2c_f32_bf16 ---------------------------------------------------------------------------------
// With the data in registers the attention / FFN blocks are VALU-issue bound (4 waves per SIMD; profiles/r03_decode_phase_stamps.txt:
// 11k of the attention block's 17k cycles are arithmetic after its last load has landed), and half of a bf16 row dot's
// instructions only unpack weights.  v_dot2c_f32_bf16 takes the packed weights as they were loaded and does two MACs per
// instruction; its other operand must be bf16 too, so an activation enters as the PAIR hi + lo (hi = the value truncated to bf16,
// lo = bf16(value - hi), which is exact before its rounding): 16 significant bits, a relative error <= 2^-17 per term -- far inside
// the half rounding of the partial rows -- for one instruction per MAC and no unpack.  fp32 handles keep the fp32 FMA chain.
// Round 5: the lo half is gone (kPairAct = false).  An activation enters a dot as ONE bf16 value, rounded to nearest -- what the
// reference's own bf16 path multiplies (its activations ARE bf16 tensors), half the dot instructions of every GEMV of the step
// (the kernels are vector-issue bound: profiles/r04_pmc_decode_b1_wave_cycles.txt), no lo arrays in LDS.  The bf16-mode oracle rounds
// the same operands (ORC_R_LIN in the decode step: the input rows of QKV / out-proj / W1 / W2 and the query of the score dots).
#ifndef GSV_PAIR_ACT
#define GSV_PAIR_ACT 0      // 1: the (hi, lo) pair of rounds 3-4, an A/B build only (the bf16-mode oracle describes 0)
#endif
constexpr bool kPairAct = GSV_PAIR_ACT != 0;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2c(uint32_t w, uint32_t x, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w), __builtin_bit_cast(bf16x2_t, x), acc, false);
}
// value -> (hi, lo) bf16 bit patterns
__device__ __forceinline__ void split_bf16(float v, uint16_t& hi, uint16_t& lo) {
    if constexpr (!kPairAct) { hi = f32_to_bf16(v); lo = 0; return; }
    const uint32_t b = __float_as_uint(v);
    hi = (uint16_t)(b >> 16);
    lo = f32_to_bf16(v - __uint_as_float(b & 0xffff0000u));
}
// a lane's 8 activations (elements 8 lane .. 8 lane + 7 of a vector stored as two bf16 arrays) as packed pairs
struct XPair { raw16 hi, lo; };
__device__ __forceinline__ XPair xpair_load(const uint16_t* __restrict__ vh, const uint16_t* __restrict__ vl, int first) {
    XPair x;
    x.hi = *reinterpret_cast<const raw16*>(vh + first);
    if constexpr (kPairAct) x.lo = *reinterpret_cast<const raw16*>(vl + first); else x.lo = raw16{};
    return x;
}
// 8 weights (one 16-byte load) . 8 activations
__device__ __forceinline__ float dot8(const raw16& w, const XPair& x) {
    float a = 0.f, b = 0.f;
    if constexpr (!kPairAct) {     // two chains of two: a dependent dot2c issues every other slot
        a = dot2c(w[0], x.hi[0], a); b = dot2c(w[1], x.hi[1], b);
        a = dot2c(w[2], x.hi[2], a); b = dot2c(w[3], x.hi[3], b);
        return a + b;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { a = dot2c(w[j], x.hi[j], a); b = dot2c(w[j], x.lo[j], b); }
    return a + b;
}
// e^x of the softmax: fp32 handles keep expf (the parity mode); bf16 handles work in the base-2 domain on v_exp_f32
template <bool FAST> __device__ __forceinline__ float sm_exp(float x) {
    if constexpr (FAST) return __builtin_amdgcn_exp2f(x); else return expf(x);
}

// N per-lane partials -> N wave totals: halve the value count at each of the first log2(N)
// butterfly levels (N-1 shuffles), then finish the remaining levels on one value.
// Every lane returns the total of value index sumN_index<N>().
template <int N> __device__ __forceinline__ float wave_sumN(const float (&v)[N]);
template <> __device__ __forceinline__ float wave_sumN<8>(const float (&v)[8]) {
    float a[4], b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = halve32_sum(v[i], v[i + 4]);
#pragma unroll
    for (int i = 0; i < 2; ++i) b[i] = halve16_sum(a[i], a[i + 2]);
    return oct_sum(halve8_sum(b[0], b[1]));
}
template <> __device__ __forceinline__ float wave_sumN<4>(const float (&v)[4]) {
    float b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) b[i] = halve32_sum(v[i], v[i + 2]);
    return row16_sum(halve16_sum(b[0], b[1]));
}
template <> __device__ __forceinline__ float wave_sumN<2>(const float (&v)[2]) {
    return xor16_sum(row16_sum(halve32_sum(v[0], v[1])));
}
template <int N> __device__ __forceinline__ int sumN_index();
template <> __device__ __forceinline__ int sumN_index<2>() { return (threadIdx.x >> 5) & 1; }
template <> __device__ __forceinline__ int sumN_index<8>() {
    const int lane = threadIdx.x & 63;
    return ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
}
template <> __device__ __forceinline__ int sumN_index<4>() {
    const int lane = threadIdx.x & 63;
    return ((lane >> 5) & 1) * 2 + ((lane >> 4) & 1);
}

// LayerNorm of a 512-vector, thread t < 512 owning element t.  One block reduction of (sum, sum of
// squares) = ONE barrier; var = E[x^2] - mean^2 in fp32 is within ~1e-6 of torch's two-pass form
// for these O(1) activations (tests hold it to 2e-5 against the reference's outputs).
// `red` must hold 2*kNW floats and not be in use by another reduction.
// FAST (bf16 handles): 1/sqrt on v_rsq_f32 (1 ulp) instead of the correctly rounded sqrt + IEEE division (35 dependent
// instructions in the serial chain of every decode kernel), and only the eight owner waves reduce and are summed -- the other
// eight just meet the barrier.  fp32 handles keep the exact form (the parity mode).
template <bool FAST = false>
__device__ __forceinline__ float ln512(float v, bool owner, float g, float bta, float* red) {
    constexpr int NOW = kD / 64;            // owner waves
    const int w = threadIdx.x >> 6;
    if (!FAST || w < NOW) {
        float s = owner ? v : 0.f, q = s * s;
        s = wave_sum(s);
        q = wave_sum(q);
        if ((threadIdx.x & 63) == 0) { red[2 * w] = s; red[2 * w + 1] = q; }
    }
    __syncthreads();
    if (FAST && w >= NOW) return 0.f;
    float ts = 0.f, tq = 0.f;
#pragma unroll
    for (int i = 0; i < (FAST ? NOW : kNW); ++i) { ts += red[2 * i]; tq += red[2 * i + 1]; }
    const float mean = ts * (1.0f / kD);
    const float var = fmaxf(tq * (1.0f / kD) - mean * mean, 0.f);
    const float rs = FAST ? __builtin_amdgcn_rsqf(var + kEps) : 1.0f / sqrtf(var + kEps);
    return (v - mean) * rs * g + bta;
}

// Sum of NPART partial 512-vectors (+ bias + residual) in two stages so that the global side is a
// handful of 1-KiB wave loads instead of NPART narrow ones per owner thread (the per-CU address
// path, not bandwidth, is what a 36-loads-per-thread prologue pays for):
//   stage A  wave w loads rows w*NPW .. +NPW-1 (16 B per lane), adds them lane-wise, parks the
//            result in LDS stage[w][512];        -- issue() at kernel entry, park() after the pin
//   stage B  thread t < 512 adds the 16 parked rows in index order, + bias + residual.
// Fixed order everywhere => bit-reproducible.
template <int NPART, typename PT = float> struct PartialSum;
template <int NPART> struct PartialSum<NPART, float> {
    static constexpr int NPW = NPART / kNW;     // rows per wave (2 for the 32 FFN slices, 1 for the 16 heads)
    static_assert(NPART % kNW == 0, "NPART");
    f32x4 p[NPW][2];
    float bias, resid, lng, lnb;
    __device__ __forceinline__ void issue(const float* __restrict__ part, const float* __restrict__ b,
                                          const float* __restrict__ r, const float* __restrict__ g,
                                          const float* __restrict__ beta) {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
        for (int j = 0; j < NPW; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                p[j][c] = *reinterpret_cast<const f32x4*>(part + (size_t)(wid * NPW + j) * kD + c * 256 + lane * 4);
        if (threadIdx.x < kD) {
            const int t = threadIdx.x;
            bias = b[t]; resid = r[t]; lng = g[t]; lnb = beta[t];
        }
    }
    __device__ __forceinline__ void park(float* __restrict__ stage) {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x4 s = p[0][c];
#pragma unroll
            for (int j = 1; j < NPW; ++j) s += p[j][c];
            *reinterpret_cast<f32x4*>(stage + wid * kD + c * 256 + lane * 4) = s;
        }
    }
    __device__ __forceinline__ float finish(const float* __restrict__ stage) {  // owners only, after a barrier
        const int t = threadIdx.x;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < kNW; ++w) s += stage[w * kD + t];
        return s + bias + resid;
    }
};

// half rows: ONE 1-KiB wave load per row (lane l holds elements 8 l .. 8 l + 7), summed in fp32
template <int NPART> struct PartialSum<NPART, _Float16> {
    static constexpr int NPW = NPART / kNW;
    static_assert(NPART % kNW == 0, "NPART");
    u32x4 p[NPW][1];
    float bias, resid, lng, lnb;
    __device__ __forceinline__ void issue(const _Float16* __restrict__ part, const float* __restrict__ b,
                                          const float* __restrict__ r, const float* __restrict__ g,
                                          const float* __restrict__ beta) {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
        for (int j = 0; j < NPW; ++j) p[j][0] = *reinterpret_cast<const u32x4*>(part + (size_t)(wid * NPW + j) * kD + lane * 8);
        if (threadIdx.x < kD) {
            const int t = threadIdx.x;
            bias = b[t]; resid = r[t]; lng = g[t]; lnb = beta[t];
        }
    }
    __device__ __forceinline__ void park(float* __restrict__ stage) {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        float s[8];
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const f16x8 h = __builtin_bit_cast(f16x8, p[j][0]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] = j == 0 ? (float)h[e] : s[e] + (float)h[e];
        }
        *reinterpret_cast<f32x4*>(stage + wid * kD + lane * 8) = f32x4{s[0], s[1], s[2], s[3]};
        *reinterpret_cast<f32x4*>(stage + wid * kD + lane * 8 + 4) = f32x4{s[4], s[5], s[6], s[7]};
    }
    __device__ __forceinline__ float finish(const float* __restrict__ stage) {  // owners only, after a barrier
        const int t = threadIdx.x;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < kNW; ++w) s += stage[w * kD + t];
        return s + bias + resid;
    }
};

// one butterfly level of a (value, lowest index) arg-max on the VALU cross-lane network
template <int M> __device__ __forceinline__ void argmax_step(float& v, int& idx) {
    const float ov = lane_xor<M>(v);
    const int oi = lane_xor<M>(idx);
    if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
}

template <int G> __device__ __forceinline__ float group_sum(float v) {  // aligned groups of G lanes, all lanes
    static_assert(G == 4 || G == 8 || G == 16, "group");
    if constexpr (G == 4) return quad_sum(v);
    else if constexpr (G == 8) return oct_sum(v);
    else return row16_sum(v);
}

// panel GEMV out[row] = dot(P[row][0:K], v) for 512 rows, K in {32, 64}: K/EPL lanes per row.
template <typename WT, int K> struct Panel {
    static constexpr int EPL = Geo<WT>::EPL;
    static constexpr int LPR = K / EPL;
    static constexpr int RPI = kNT / LPR;
    static constexpr int NIT = kD / RPI;
    raw16 w[NIT];
    template <bool NT = false>
    __device__ __forceinline__ void issue(const WT* __restrict__ panel) {
        const int tid = threadIdx.x;
        const int part = tid % LPR, rsub = tid / LPR;
#pragma unroll
        for (int it = 0; it < NIT; ++it) w[it] = ldg16w<NT>(panel + (size_t)(rsub + it * RPI) * K + part * EPL);
    }
    // bf16 handles: the K-vector as bf16 hi / lo arrays (xpair_load)
    template <typename OT>
    __device__ __forceinline__ void finish2(const uint16_t* __restrict__ vh, const uint16_t* __restrict__ vl, OT* __restrict__ out) {
        static_assert(EPL == 8, "bf16 panels");
        const int tid = threadIdx.x;
        const int part = tid % LPR, rsub = tid / LPR;
        const XPair x = xpair_load(vh, vl, part * 8);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const float s = group_sum<LPR>(dot8(w[it], x));
            if (part == 0) out[rsub + it * RPI] = (OT)s;
        }
    }
    template <typename OT>
    __device__ __forceinline__ void finish(const float* __restrict__ vec_lds, OT* __restrict__ out) {
        const int tid = threadIdx.x;
        const int part = tid % LPR, rsub = tid / LPR;
        float vr[EPL];
#pragma unroll
        for (int i = 0; i < EPL; ++i) vr[i] = vec_lds[part * EPL + i];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            float wv[EPL];
            Unpack<WT, EPL>::run(w[it], wv);
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < EPL; ++i) s = fmaf(wv[i], vr[i], s);
            s = group_sum<LPR>(s);
            if (part == 0) out[rsub + it * RPI] = (OT)s;
        }
    }
};

// ---- attention kernel ----------------------------------------------------------------------

// MODE 2 of the attention kernel (layer 0, greedy / host-chosen tokens): what t2s_token_kernel does with advance = 1 --
// pending token -> pre_tokens / seen / eos_at / step, next input = embedding + position row -- as the kernel's prologue, so
// the step has one launch less.  Every head block derives the token and its input row; block 0 keeps the books.
struct StepTok {
    const TokPart* tokpart;      // [B][kNP]
    const int64_t* tok_override; // ctl[0] == 1
    const int32_t* ctl;
    const int64_t* x_len;
    int64_t* pre_tokens;         // [B][T+1]
    uint8_t* seen;               // [B][V]
    int32_t* step;
    int32_t* eos_at;
    const float* emb;            // [V][512]
    const float* pe;             // [n_pos][512]
    int V, eos, n_pos;
    int32_t* eos_host;           // null, or a host-mapped mirror of eos_at (gsv_t2s_set_eos_mirror)
};
struct StepTokLoads { TokPart tp; int ctl0, ctl2; int64_t ovr, xl; };
__device__ __forceinline__ StepTokLoads steptok_issue(const StepTok& k, int b, int lane) {
    StepTokLoads L;
    L.tp = k.tokpart[(size_t)b * kNP + min(lane, kNP - 1)];
    L.ctl0 = k.ctl[0]; L.ctl2 = k.ctl[2];
    L.ovr = k.tok_override[b];
    L.xl = k.x_len[b];
    return L;
}
// -> channel `tid` of sequence b's next input row (owner threads); `books`: this thread records the token (one per sequence)
__device__ __forceinline__ float steptok_finish(const StepTok& k, const StepTokLoads& L, int b, int lane, int tid, bool owner, int64_t n64,
                                                int T, bool books) {
    // the pending token: arg-max over the logits kernel's kNP partials (lowest index on ties), or the host's choice
    float bv = lane < kNP ? L.tp.v : -INFINITY;
    int tok = lane < kNP ? L.tp.idx : 0x7fffffff;
    argmax_step<32>(bv, tok); argmax_step<16>(bv, tok); argmax_step<8>(bv, tok);
    argmax_step<4>(bv, tok); argmax_step<2>(bv, tok); argmax_step<1>(bv, tok);
    if (L.ctl0 != 0) tok = (int)L.ovr;
    if (tok < 0 || tok >= k.V) tok = 0;
    int64_t pos = n64 - L.xl;
    if (pos < 0) pos += k.n_pos;  // torch negative indexing of the PE table (idle slots only)
    if (pos < 0) pos = 0;
    if (pos >= k.n_pos) pos = k.n_pos - 1;
    const float v = owner ? k.emb[(size_t)tok * kD + tid] * 1.0f + k.pe[(size_t)pos * kD + tid] : 0.f;
    if (books) {
        if (n64 >= 0 && n64 <= T) k.pre_tokens[(size_t)b * (T + 1) + n64] = tok;
        if (L.ctl2 != 0 && n64 >= 0) k.seen[(size_t)b * k.V + tok] = 1;
        const int stp = k.step[b];
        if (tok == k.eos && k.eos_at[b] < 0) { k.eos_at[b] = stp; eos_publish(k.eos_host, b, stp); }
        k.step[b] = stp + 1;
    }
    return v;
}

template <typename WT>
struct AttnArgs {
    // layer input: MODE 0 -> xdirect[B][512]; MODE 1 -> LN2(sum_j zpart + b2 + x1) of the previous layer
    const float* xdirect;
    const typename PartOf<WT>::T* zpart;  // [B][NJ][512]
    const float* b2;
    const float* x1;     // [B][512]
    const float* ln2g;
    const float* ln2b;
    float* xout;         // [B][512] layer input, written by head 0 (residual for the ffn kernel)
    const WT* wqkv;      // [16][96][512]  rows: q(32) k(32) v(32) of head h
    const float* bqkv;   // [16][96]
    const WT* wo;        // [16][512][32]  wo[h][n][d] = Wo[n][h*32+d]
    WT* kc;              // this layer: [B][16][T][32]
    WT* vc;
    const int64_t* kv_len;
    int T;
    typename PartOf<WT>::T* ypart;        // [B][16][512]
    unsigned long long* dbg;
    StepTok tk;          // MODE 2 only
};

constexpr int kAttnLdsFloats = kD + 96 + 32 + 2 * kNW + kNW * 32 + 2 * kNW + kNW * kD;

// NT: this layer's weights are loaded non-temporally (fp32 handles, the upper layers).  A step of an fp32 handle streams 304 MB of
// weights + its K/V rows; the 256 MiB Infinity Cache holds a cyclic set of up to ~208 MB (profiles/r03_l2_prefetch_probe.txt), so
// under the default policy EVERY load of every layer misses it (per layer 20.1 us at 24 layers against 15.6 at 8, which fit).
// With the upper layers passing through without allocating, the lower ones stay resident.  bf16 handles fit as a whole: the
// hint made their step 10 % slower (round 1) and is off.
// NTKV: the K/V rows non-temporal too -- from a few sequences on a step's K/V rows + weights exceed what the Infinity Cache holds
// and the K/V stream (read once per step) evicts the weights every other launch wants back.
template <typename WT, int MODE, int NJ = kNJ, bool NT = false, bool NTKV = false>
__global__ __launch_bounds__(kNT) void t2s_attn_kernel(AttnArgs<WT> a) {
    __shared__ __attribute__((aligned(16))) float smem[kAttnLdsFloats];
    float* xs = smem;             // 512
    float* qkv = xs + kD;         // 96
    float* att = qkv + 96;        // 32
    float* red = att + 32;        // 2*16
    float* pacc = red + 2 * kNW;  // 16*32  per-wave un-normalised P.V
    float* pm = pacc + kNW * 32;  // 16     per-wave running max
    float* pl = pm + kNW;         // 16     per-wave sum of exp
    float* stage = pl + kNW;      // 16*512 parked partial sums
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int EPL = Geo<WT>::EPL;
    constexpr int CPR = Geo<WT>::CPR;
    constexpr int LPR = kDh / EPL;         // lanes per K/V row
#ifndef GSV_ATTN_WAVES
#define GSV_ATTN_WAVES 8
#endif
    // waves that run the attention phase (the others wait at its barrier).  The phase is VALU-issue bound and most of a wave's ~200 instructions
    // are per-wave overhead (running max / sum across the wave, the P.V reduction tree), not per-row work: eight waves with four K/V rows per
    // thread issue a quarter fewer instructions than sixteen with two (bf16: 0.2735 -> 0.2695 ms per step at one sequence, 0.297 -> 0.288 at 4;
    // fp32 handles keep sixteen: eight K/V vectors more per thread spill there, 0.433 -> 0.473 ms)
    constexpr int AW = sizeof(WT) == 2 ? GSV_ATTN_WAVES : kNW;
    constexpr int RPI = AW * 64 / LPR;     // K/V rows per block iteration (256 bf16, 128 f32 at 16 waves)
    constexpr int KCH = 2 * kNW / AW;      // iterations held in registers per chunk (512 positions bf16, 256 f32)
    const bool aw = __builtin_amdgcn_readfirstlane(wid) < AW;
    constexpr int RW = 96 / kNW;           // 6 QKV rows per wave
    const bool owner = tid < kD;
    constexpr bool BF = sizeof(WT) == 2;   // bf16 handle: dots on v_dot2c_f32_bf16, activations as bf16 (hi, lo) pairs in LDS
    uint16_t* xh = reinterpret_cast<uint16_t*>(xs);      // [512] hi, [512] lo over xs
    uint16_t* xl = xh + kD;
    uint16_t* qh = reinterpret_cast<uint16_t*>(qkv);     // q hi [32], q lo [32], the new key [32] over qkv[0..47]; v stays fp32 at qkv[64..]
    uint16_t* ql = qh + 32;
    uint16_t* kn = qh + 64;
    uint16_t* atth = reinterpret_cast<uint16_t*>(att);   // attention output hi [32], lo [32] over att
    uint16_t* attl = atth + 32;
    stamp(a.dbg, 0);

    // kv_len is loaded FIRST and used LAST: only the K/V row addresses need it.  (Clamping right here made hipcc wait for this --
    // cold -- load before it issued a single weight load: the whole weight stream started 1.5k cycles late.)
    // (its LOW dword only: hipcc re-used the unused upper half of a 64-bit destination as a temporary and put the wait there)
    int kvl_raw = reinterpret_cast<const int*>(a.kv_len)[2 * b];
    WT* Kp = a.kc + (((size_t)b * kH + h) * a.T) * kDh;
    WT* Vp = a.vc + (((size_t)b * kH + h) * a.T) * kDh;
    const int part = tid % LPR, rsub = tid / LPR;

    // ---- issue everything whose address is known now, in consumption order
    PartialSum<NJ, typename Geo<WT>::PT> ps;
    float xd = 0.f;
    StepTokLoads tl;
    tl.tp.v = 0.f;
    if constexpr (MODE == 0) {
        if (owner) xd = a.xdirect[(size_t)b * kD + tid];
    } else if constexpr (MODE == 2) {
        tl = steptok_issue(a.tk, b, lane);
    } else {
        ps.issue(a.zpart + (size_t)b * NJ * kD, a.b2, a.x1 + (size_t)b * kD, a.ln2g, a.ln2b);
    }
    // partials first: a CU serves its waves' loads in issue order and waves start staggered, so
    // without this rendezvous the last wave's partial rows queue behind the first waves' weights
    if constexpr (MODE == 1) __builtin_amdgcn_s_barrier();
    asm volatile("" : : : "memory");
    const WT* wp = a.wqkv + ((size_t)h * 96 + wid * RW) * kD;
    raw16 wq[RW][CPR];
#pragma unroll
    for (int r = 0; r < RW; ++r) row_load<WT, NT>(wp + (size_t)r * kD, wq[r]);
    Panel<WT, kDh> po;
    // fp32 handles: the out-proj panel (16 registers) is requested BEHIND the q / k / v rows' dots, when their 48 weight registers are
    // free -- at kernel entry the two together spilled 11 registers (and a spilled load is a wait at the top of the kernel); the panel has
    // the whole attention phase to land
    if constexpr (BF) po.template issue<NT>(a.wo + (size_t)h * kD * kDh);
    const int oi = sumN_index<8>();
    const float bq = a.bqkv[h * 96 + wid * RW + min(oi, RW - 1)];
    // ... and now kv_len (the opaque asm keeps its first use -- and with it the wait -- down here, behind the weight loads)
    asm volatile("" : "+v"(kvl_raw) : : "memory");
    int n = kvl_raw;
    // kv_len < 0 = a PARKED slot (include/gsv_tts_hip.h, staged refill): its K/V row goes to the last row of the cache,
    // which no prompt pass writes, and it attends over row 0 only -- it must not touch rows a concurrent refill fills
    const int nw = n < 0 ? a.T - 1 : (n > a.T - 1 ? a.T - 1 : n);
    if (n > a.T - 1) n = a.T - 1;  // memory safety only; the host never steps a full cache
    if (n < 0) n = 0;
    // K/V rows are loaded UNCONDITIONALLY from a clamped (always valid) row and masked at use: a
    // per-element "load or zero" select makes hipcc branch around each load and drain vmcnt(0)
    raw16 kreg[KCH], vreg[KCH];
    if (AW == kNW || aw) {
#pragma unroll
        for (int it = 0; it < KCH; ++it) kreg[it] = ldg16w<NTKV>(Kp + (size_t)min(rsub + it * RPI, n) * kDh + part * EPL);
#pragma unroll
        for (int it = 0; it < KCH; ++it) vreg[it] = ldg16w<NTKV>(Vp + (size_t)min(rsub + it * RPI, n) * kDh + part * EPL);
    }
    // Pin "all loads issued, THEN arithmetic": the opaque asm redefines the head of the partial-sum
    // chain, so no add can be scheduled above it, while the memory clobber keeps every load above
    // it.  It only needs the FIRST-issued load to have landed.
    if constexpr (MODE == 0) asm volatile("" : "+v"(xd) : : "memory");
    else if constexpr (MODE == 2) asm volatile("" : "+v"(tl.tp.v) : : "memory");
    else asm volatile("" : "+v"(ps.p[0][0]) : : "memory");
#ifdef GSV_DBG_DRAIN   // measurement only: every load of the block has landed before any arithmetic starts
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    __syncthreads();
#endif
    stamp(a.dbg, 1);

    // ---- layer input
    float v;
    if constexpr (MODE == 0) {
        v = xd;
    } else if constexpr (MODE == 2) {
        v = steptok_finish(a.tk, tl, b, lane, tid, owner, (int64_t)kvl_raw, a.T, h == 0 && tid == 0);   // (a second load of kv_len here drained the load counter)
    } else {
        ps.park(stage);
        __syncthreads();
        v = ln512<BF>(owner ? ps.finish(stage) : 0.f, owner, ps.lng, ps.lnb, red);
    }
    if (owner) {
        if constexpr (BF) { uint16_t vh, vl; split_bf16(v, vh, vl); xh[tid] = vh; if constexpr (kPairAct) xl[tid] = vl; }
        else xs[tid] = v;
        if (h == 0) a.xout[(size_t)b * kD + tid] = v;
    }
    __syncthreads();
    stamp(a.dbg, 2);

    // ---- q, k, v of this head: 96 rows, 6 per wave.  k and v are rounded through the cache type
    //      (this step must see exactly what later steps read back) and appended at position n.
    {
        float acc[8];
        if constexpr (BF) {
            const XPair x = xpair_load(xh, xl, lane * 8);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = u < RW ? dot8(wq[u < RW ? u : 0][0], x) : 0.f;
        } else {
            float xr[8];
            lane_x<WT>(xs, xr);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = u < RW ? row_dot<WT>(wq[u < RW ? u : 0], xr) : 0.f;
        }
        const float tot = wave_sumN<8>(acc);
        if ((lane & 7) == 0 && oi < RW) {
            const int row = wid * RW + oi;
            float val = tot + bq;
            WT s = from_f32<WT>(val);
            if (row >= 32) {
                val = to_f32<WT>(s);
                if (row < 64) Kp[(size_t)nw * kDh + row - 32] = s; else Vp[(size_t)nw * kDh + row - 64] = s;
            }
            if constexpr (BF) {
                if (row < 32) { uint16_t vh, vl; split_bf16(val, vh, vl); qh[row] = vh; if constexpr (kPairAct) ql[row] = vl; }
                else if (row < 64) kn[row - 32] = s;     // the new key is a bf16 value: it enters the score dot as it is
                else qkv[row] = val;
            } else {
                qkv[row] = val;
            }
        }
    }
    __syncthreads();
    stamp(a.dbg, 3);
    if constexpr (!BF) po.template issue<NT>(a.wo + (size_t)h * kD * kDh);

    // ---- single-pass attention over [0, n]: every thread owns the same rows of K and of V, so the
    //      scores never leave registers; each wave keeps a running (max, sum, P.V) and the 16 waves
    //      are merged once at the end (flash-decoding style, deterministic order).
    // 1/sqrt(32); bf16 handles keep the scores in the base-2 domain (x log2 e) for v_exp_f32
    const float scale = BF ? 0.17677669529663687f * 1.4426950408889634f : 0.17677669529663687f;
    float qr[EPL];
    XPair qp;
    if constexpr (BF) {
        qp = xpair_load(qh, ql, part * 8);
    } else {
#pragma unroll
        for (int i = 0; i < EPL; ++i) qr[i] = qkv[part * EPL + i];
    }
    float m_run = -INFINITY, l_run = 0.f, acc[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) acc[i] = 0.f;
    const bool wave0 = __builtin_amdgcn_readfirstlane(wid) == 0;
    if (AW == kNW || aw) {
    for (int c0 = 0; c0 == 0 || c0 < n; c0 += KCH * RPI) {
        if (c0 > 0) {
#pragma unroll
            for (int it = 0; it < KCH; ++it) {
                kreg[it] = ldg16w<NTKV>(Kp + (size_t)min(c0 + rsub + it * RPI, n) * kDh + part * EPL);
                vreg[it] = ldg16w<NTKV>(Vp + (size_t)min(c0 + rsub + it * RPI, n) * kDh + part * EPL);
            }
        }
        float sv[KCH + 1];
        float cmax = -INFINITY;
#pragma unroll
        for (int it = 0; it < KCH; ++it) {
            const int r = c0 + rsub + it * RPI;
            float s = 0.f;
            if constexpr (BF) {
                s = dot8(kreg[it], qp);
            } else {
                float kk[EPL];
                Unpack<WT, EPL>::run(kreg[it], kk);
#pragma unroll
                for (int i = 0; i < EPL; ++i) s = fmaf(qr[i], kk[i], s);
            }
            s = group_sum<LPR>(s);
            sv[it] = r < n ? s * scale : -INFINITY;
            cmax = max_nn(cmax, sv[it]);
        }
        // the new token's own key/value (position n) rides with wave 0's first chunk; the other waves (fifteen of sixteen, all
        // of them VALU-issue bound here) skip its score, exponential and eight FMAs, which contributed exact zeros
        const bool own = wave0 && c0 == 0;
        sv[KCH] = -INFINITY;
        if (own) {
            float s = 0.f;
            if constexpr (BF) {
                s = dot8(*reinterpret_cast<const raw16*>(kn + part * 8), qp);
            } else {
#pragma unroll
                for (int i = 0; i < EPL; ++i) s = fmaf(qr[i], qkv[32 + part * EPL + i], s);
            }
            s = group_sum<LPR>(s);
            sv[KCH] = tid < LPR ? s * scale : -INFINITY;
            cmax = max_nn(cmax, sv[KCH]);
        }
        cmax = wave_max(cmax);
        const float m_new = max_nn(m_run, cmax);
        const float mref = (m_new == -INFINITY) ? 0.f : m_new;
        const float f = sm_exp<BF>(m_run - mref);      // exp(-inf) = 0 on the first live chunk
        l_run *= f;
#pragma unroll
        for (int i = 0; i < EPL; ++i) acc[i] *= f;
#pragma unroll
        for (int it = 0; it < KCH; ++it) {
            const float p = sm_exp<BF>(sv[it] - mref);  // 0 for masked rows
            const bool live = sv[it] != -INFINITY;
            const raw16 vr = live ? vreg[it] : raw16{0u, 0u, 0u, 0u};   // a masked row may hold anything (0 x NaN)
            float vv[EPL];
            Unpack<WT, EPL>::run(vr, vv);
            if (part == 0) l_run += p;
#pragma unroll
            for (int i = 0; i < EPL; ++i) acc[i] = fmaf(p, vv[i], acc[i]);
        }
        if (own) {
            const float p = sm_exp<BF>(sv[KCH] - mref);
            if (part == 0) l_run += p;
#pragma unroll
            for (int i = 0; i < EPL; ++i) acc[i] = fmaf(p, qkv[64 + part * EPL + i], acc[i]);
        }
        m_run = m_new;
    }
    // the wave's un-normalised P.V: every thread holds EPL dims of its rows; sum over the rows by halving exchanges
    // (lane halves, 16-lane rows, 8-lane groups: one value left per lane), then the last level on the DPP network
    l_run = wave_sum(l_run);
    if constexpr (EPL == 8) {      // bf16: 4 lanes per row; lane bits 5, 4, 3 pick the dim, bit 2 is summed last
        float r4[4], r2[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) r4[i] = halve32_sum(acc[i], acc[i + 4]);
#pragma unroll
        for (int i = 0; i < 2; ++i) r2[i] = halve16_sum(r4[i], r4[i + 2]);
        float r1 = halve8_sum(r2[0], r2[1]);
        r1 += lane_xor<4>(r1);
        if ((lane & 4) == 0) pacc[wid * 32 + part * 8 + 4 * (lane >> 5) + 2 * ((lane >> 4) & 1) + ((lane >> 3) & 1)] = r1;
    } else {                       // fp32: 8 lanes per row; lane bits 5, 4 pick the dim, bit 3 is summed last
        float r2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) r2[i] = halve32_sum(acc[i], acc[i + 2]);
        float r1 = halve16_sum(r2[0], r2[1]);
        r1 += lane_xor<8>(r1);
        if ((lane & 8) == 0) pacc[wid * 32 + part * 4 + 2 * (lane >> 5) + ((lane >> 4) & 1)] = r1;
    }
    if (lane == 0) { pm[wid] = m_run; pl[wid] = l_run; }
    }   // aw
    stamp(a.dbg, 4);
    __syncthreads();
    if (wid == 0) {
        // merge the AW waves: lane l (mod 16) owns wave l's (max, sum); 2x32 lanes own the 32 dims
        const bool has = (lane & 15) < AW;
        const float mw = has ? pm[lane & 15] : -INFINITY, lw = has ? pl[lane & 15] : 0.f;
        const float M = row16_max(mw);
        const float den = row16_sum(lw * sm_exp<BF>(mw - M));      // waves with no live rows: exp(-inf) = 0
        const int hf = lane >> 5, d = lane & 31;
        float num = 0.f;
#pragma unroll
        for (int w = 0; w < AW / 2; ++w) num = fmaf(pacc[(hf * (AW / 2) + w) * 32 + d], sm_exp<BF>(pm[hf * (AW / 2) + w] - M), num);
        num = xor32_sum(num);
        if (lane < 32) {
            if constexpr (BF) { uint16_t vh, vl; split_bf16(num * __builtin_amdgcn_rcpf(den), vh, vl); atth[d] = vh; if constexpr (kPairAct) attl[d] = vl; }
            else att[d] = num / den;
        }
    }
    __syncthreads();
    stamp(a.dbg, 5);
    if constexpr (BF) po.finish2(atth, attl, a.ypart + ((size_t)b * kH + h) * kD);
    else po.finish(att, a.ypart + ((size_t)b * kH + h) * kD);
    stamp(a.dbg, 6);
}

// ---- ffn kernel ----------------------------------------------------------------------------

template <typename WT>
struct FfnArgs {
    const typename PartOf<WT>::T* ypart;  // [B][16][512]
    const float* bo;
    const float* x;      // [B][512] layer input (residual)
    const float* ln1g;
    const float* ln1b;
    float* x1out;        // [B][512] LN1 output, written by slice 0
    const WT* w1;        // [2048][512] (torch layout; slice j = rows j*64..)
    const float* b1;
    const WT* w2p;       // [NJ][512][FJ]  w2p[j][n][i] = W2[n][j*FJ+i]
    typename PartOf<WT>::T* zpart;        // [B][NJ][512]
    unsigned long long* dbg;
};

template <typename WT, int NJ = kNJ, bool NT = false>
__global__ __launch_bounds__(kNT) void t2s_ffn_kernel(FfnArgs<WT> a) {
    constexpr int FJ = kF / NJ;    // hidden units of this slice
    __shared__ __attribute__((aligned(16))) float smem[kD + FJ + 2 * kNW + kNW * kD];
    float* xs = smem;
    float* hb = xs + kD;
    float* red = hb + FJ;
    float* stage = red + 2 * kNW;
    const int j = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int CPR = Geo<WT>::CPR;
    constexpr int RW = FJ / kNW;   // 4 (2) W1 rows per wave
    const bool owner = tid < kD;
    constexpr bool BF = sizeof(WT) == 2;   // bf16 handle: dots on v_dot2c_f32_bf16 (see dot8)
    uint16_t* xh = reinterpret_cast<uint16_t*>(xs);
    uint16_t* xl = xh + kD;
    uint16_t* hbh = reinterpret_cast<uint16_t*>(hb);     // hidden units hi [FJ], lo [FJ] over hb
    uint16_t* hbl = hbh + FJ;
    stamp(a.dbg, 8);

    PartialSum<kH, typename Geo<WT>::PT> ps;
    ps.issue(a.ypart + (size_t)b * kH * kD, a.bo, a.x + (size_t)b * kD, a.ln1g, a.ln1b);
    __builtin_amdgcn_s_barrier();  // all partial loads queued before any weight load (see attn kernel)
    asm volatile("" : : : "memory");
    const int row0 = j * FJ + wid * RW;
    raw16 w1r[RW][CPR];
#pragma unroll
    for (int r = 0; r < RW; ++r) row_load<WT, NT>(a.w1 + (size_t)(row0 + r) * kD, w1r[r]);
    Panel<WT, FJ> p2;
    p2.template issue<NT>(a.w2p + (size_t)j * kD * FJ);
    const int oi = sumN_index<RW>();
    const float b1r = a.b1[row0 + oi];
    asm volatile("" : "+v"(ps.p[0][0]) : : "memory");
#ifdef GSV_DBG_DRAIN
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    __syncthreads();
#endif
    stamp(a.dbg, 9);

    ps.park(stage);
    stamp(a.dbg, 13);
    __syncthreads();
    stamp(a.dbg, 14);
    const float v = ln512<BF>(owner ? ps.finish(stage) : 0.f, owner, ps.lng, ps.lnb, red);
    stamp(a.dbg, 15);
    if (owner) {
        if constexpr (BF) { uint16_t vh, vl; split_bf16(v, vh, vl); xh[tid] = vh; if constexpr (kPairAct) xl[tid] = vl; }
        else xs[tid] = v;
        if (j == 0) a.x1out[(size_t)b * kD + tid] = v;
    }
    __syncthreads();
    stamp(a.dbg, 10);
    {
        float acc[RW];
        if constexpr (BF) {
            const XPair x = xpair_load(xh, xl, lane * 8);
#pragma unroll
            for (int u = 0; u < RW; ++u) acc[u] = dot8(w1r[u][0], x);
        } else {
            float xr[8];
            lane_x<WT>(xs, xr);
#pragma unroll
            for (int u = 0; u < RW; ++u) acc[u] = row_dot<WT>(w1r[u], xr);
        }
        const float tot = wave_sumN<RW>(acc);
        if ((lane & (64 / RW - 1)) == 0) {
            const float hv = fmaxf(tot + b1r, 0.f);
            if constexpr (BF) { uint16_t vh, vl; split_bf16(hv, vh, vl); hbh[wid * RW + oi] = vh; if constexpr (kPairAct) hbl[wid * RW + oi] = vl; }
            else hb[wid * RW + oi] = hv;
        }
    }
    __syncthreads();
    stamp(a.dbg, 11);
    if constexpr (BF) p2.finish2(hbh, hbl, a.zpart + ((size_t)b * NJ + j) * kD);
    else p2.finish(hb, a.zpart + ((size_t)b * NJ + j) * kD);
    stamp(a.dbg, 12);
}

// ---- logits kernel -------------------------------------------------------------------------

template <typename WT>
struct LogitsArgs {
    // final hidden: MODE 1 -> LN2(sum zpart + b2 + x1) of the last layer; MODE 0 -> hdirect[B][512]
    const float* hdirect;
    const typename PartOf<WT>::T* zpart;
    const float* b2;
    const float* x1;
    const float* ln2g;
    const float* ln2b;
    const WT* wp;        // [V][512]
    int V, eos;
    int vlimit;          // logits with v >= vlimit are -inf (first sample drops the EOS column)
    int slot0;           // first state slot of row 0
    const int32_t* slots;// or: state slot of every row (refills of scattered slots); null = slot0 + row
    const int32_t* step;
    const int32_t* ctl;  // {sample_mode, suppress_steps, rep_enabled, -, top_k, seed_lo, seed_hi, suppress_first}
    const float* fctl;   // {rep_penalty}
    const uint8_t* seen; // [B][V]
    float* logits;       // [B][V]
    float* hidden;       // [B][512]
    TokPart* tokpart;    // [B][kNP]
    int64_t* kv_len;     // bumped by slice 0 when bump != 0
    int bump;
};

template <typename WT, int MODE, int NJ = kNJ>
__global__ __launch_bounds__(kNT) void t2s_logits_kernel(LogitsArgs<WT> a) {
    __shared__ __attribute__((aligned(16))) float smem[kD + 2 * kNW + 128 + kNW * kD];
    float* xs = smem;
    float* red = xs + kD;
    float* lg = red + 2 * kNW;  // up to 128 rows per slice
    float* stage = lg + 128;
    const int p = blockIdx.x, r_ = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = a.slots ? a.slots[r_] : a.slot0 + r_;  // state slot; row r_ of zpart/x1/hdirect
    constexpr int CPR = Geo<WT>::CPR;
    const int rpb = (a.V + kNP - 1) / kNP;      // rows per slice (<= 128)
    const int rww = (rpb + kNW - 1) / kNW;      // rows per wave (<= 8)
    const int vbase = p * rpb;
    const int nrow = min(rpb, a.V - vbase);
    const bool owner = tid < kD;

    // the control words are loaded FIRST and used LAST (as scalars): loaded behind the weight rows, their first use drained the
    // whole load counter before the LayerNorm could start
    int c_step = a.step[b], c_sup = a.ctl[1], c_first = a.ctl[7], c_rep = a.ctl[2];
    float c_rp = a.fctl[0];
    PartialSum<NJ, typename Geo<WT>::PT> ps;
    float xd = 0.f;
    if constexpr (MODE == 0) {
        if (owner) xd = a.hdirect[(size_t)r_ * kD + tid];
    } else {
        ps.issue(a.zpart + (size_t)r_ * NJ * kD, a.b2, a.x1 + (size_t)r_ * kD, a.ln2g, a.ln2b);
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("" : : : "memory");
    raw16 wr[8][CPR];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int v = min(vbase + wid * rww + min(u, rww - 1), a.V - 1);
        row_load<WT>(a.wp + (size_t)v * kD, wr[u]);
    }
    // the first sample (the prefill's logits: vlimit < V) is suppressed unconditionally by infer / infer_stream
    // (t2s_model.py:415-416; ctl[7] set by the host), later samples while step < initial_suppression_steps (:444-445)
    const int oi = sumN_index<8>();
    const int myr = wid * rww + oi;             // slice row this lane will emit (if oi < rww)
    const uint8_t sn = a.seen[(size_t)b * a.V + min(vbase + myr, a.V - 1)];
    asm volatile("" : "+v"(c_step), "+v"(c_sup), "+v"(c_first), "+v"(c_rep), "+v"(c_rp) : : "memory");
    const bool sup = c_step < c_sup || (a.vlimit < a.V && c_first != 0);
    const bool rep = c_rep != 0;
    const float rp = c_rp;
    if constexpr (MODE == 0) asm volatile("" : "+v"(xd) : : "memory");
    else asm volatile("" : "+v"(ps.p[0][0]) : : "memory");

    float v;
    if constexpr (MODE == 0) {
        v = xd;
    } else {
        ps.park(stage);
        __syncthreads();
        v = ln512<sizeof(WT) == 2>(owner ? ps.finish(stage) : 0.f, owner, ps.lng, ps.lnb, red);
    }
    if (owner) {
        xs[tid] = v;
        if (p == 0) a.hidden[(size_t)b * kD + tid] = v;
    }
    __syncthreads();
    {
        float xr[8];
        lane_x<WT>(xs, xr);
        float acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = row_dot<WT>(wr[u], xr);
        const float tot = wave_sumN<8>(acc);
        if ((lane & 7) == 0 && oi < rww && myr < nrow) {
            const int vv = vbase + myr;
            float l = tot;
            if (vv >= a.vlimit) l = -INFINITY;
            if (sup && (vv == 280 || vv == 486 || vv == a.eos)) l = -INFINITY;
            if (rep && sn) l = l < 0.f ? l * rp : l / rp;
            lg[myr] = l;
            a.logits[(size_t)b * a.V + vv] = l;
        }
    }
    __syncthreads();
    if (wid == 0) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int r = lane; r < nrow; r += 64) {
            float l = lg[r];
            if (l > bv || (l == bv && vbase + r < bi)) { bv = l; bi = vbase + r; }
        }
        argmax_step<32>(bv, bi); argmax_step<16>(bv, bi); argmax_step<8>(bv, bi);
        argmax_step<4>(bv, bi); argmax_step<2>(bv, bi); argmax_step<1>(bv, bi);
        if (lane == 0) {
            TokPart tp; tp.v = bv; tp.idx = bi;
            a.tokpart[(size_t)b * kNP + p] = tp;
            if (p == 0 && a.bump) { const int64_t kn = a.kv_len[b]; if (kn >= 0) a.kv_len[b] = kn + 1; }   // parked slots stay parked
        }
    }
}

// ---- token kernel: pending token -> pre_tokens / seen / eos_at, and the next input embedding -----

struct TokenArgs {
    const TokPart* tokpart;  // [B][kNP]
    const int64_t* tok_override;
    const int32_t* ctl;
    const int64_t* kv_len;
    const int64_t* x_len;
    int64_t* pre_tokens;     // [B][T+1]
    uint8_t* seen;           // [B][V]
    int32_t* step;
    int32_t* eos_at;
    const float* emb;        // [V][512] audio embedding
    const float* pe;         // [n_pos][512] alpha_audio * pe
    float* xcur;             // [B][512]
    int T, V, eos, n_pos, advance;
    const float* logits;     // [B][V] penalised logits of the pending sample (device sampling, ctl[0] == 2)
    const float* fctl;       // fctl[1] = temperature
    int32_t* eos_host;       // null, or a host-mapped mirror of eos_at
};

// Counter-based uniform in (0, 1): one draw per (seed, slot, absolute position, vocabulary entry).  The
// reference draws Exp(1) noise from torch's generator (GPT/utils.py:56-59); a device sampler cannot share
// that stream, so it owns one: lowbias32 over a mix of the counters (documented in INTEGRATION.md).
__device__ __forceinline__ float t2s_uniform(uint32_t seed_lo, uint32_t seed_hi, uint32_t slot, uint32_t pos, uint32_t step,
                                             uint32_t v) {
    uint32_t h = seed_lo ^ (v * 0x9E3779B1u) ^ (pos * 0x85EBCA77u) ^ (slot * 0xC2B2AE3Du) ^ (step * 0x27D4EB2Fu) ^
                 ((seed_hi << 13) | (seed_hi >> 19));
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    h += seed_hi;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

// Device sampling (ctl[0] == 2) by ONE wave, lane l holding vocabulary entries l, l + 64, ... (V <= 64 * NPL): top-p,
// temperature, top-k with the reference's tie rule, then the exponential race of GPT/utils.py:56-59 as a Gumbel argmax
// (argmax p/q, q ~ Exp(1)  ==  argmax (x - log q)).  Every reduction is a cross-lane one: no block barrier, no LDS
// (the 256-thread version paid two barriers per top-k round and per top-p bisection step: 20 / 35 us per token).
template <int NPL>
__device__ __forceinline__ int t2s_sample_wave(const float* __restrict__ lg, int V, float temperature, float top_p, int k, uint32_t seed_lo,
                                               uint32_t seed_hi, uint32_t slot, uint32_t pos, uint32_t stp) {
    const int lane = threadIdx.x & 63;
    const float invt = 1.0f / fmaxf(temperature, 1e-5f);
    float x[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int v = lane + 64 * i;
        x[i] = v < V ? lg[min(v, V - 1)] * invt : -INFINITY;
    }
    // top-p (GPT/utils.py:29-40), on the un-tempered penalised logits: a token stays iff the probability mass of the
    // tokens at least as likely as it is <= top_p (or it is the arg-max).  That set is {p >= tau}; tau is found by
    // bisection over the float bit pattern (31 wave sums) instead of the reference's sort + cumsum.
    if (top_p > 0.f && top_p < 1.0f) {
        float lmax = -INFINITY; int li = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < NPL; ++i)
            if (x[i] > lmax) { lmax = x[i]; li = lane + 64 * i; }
        argmax_step<32>(lmax, li); argmax_step<16>(lmax, li); argmax_step<8>(lmax, li);
        argmax_step<4>(lmax, li); argmax_step<2>(lmax, li); argmax_step<1>(lmax, li);
        const float lm = lmax * (1.0f / invt);           // x holds logits * invt
        float pr[NPL], z = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            pr[i] = x[i] > -INFINITY ? expf(x[i] * (1.0f / invt) - lm) : 0.f;
            z += pr[i];
        }
        const float iz = 1.0f / wave_sum(z);
#pragma unroll
        for (int i = 0; i < NPL; ++i) pr[i] *= iz;
        unsigned lo = 0u, hi = 0x3f800001u;              // G(lo) > top_p >= G(hi)
        for (int it = 0; it < 31 && hi - lo > 1u; ++it) {
            const unsigned mid = lo + (hi - lo) / 2u;
            const float xm = __uint_as_float(mid);
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NPL; ++i) s += pr[i] >= xm ? pr[i] : 0.f;
            if (wave_sum(s) <= top_p) hi = mid; else lo = mid;
        }
        const float tau = __uint_as_float(hi);
#pragma unroll
        for (int i = 0; i < NPL; ++i)
            if (!(pr[i] >= tau) && lane + 64 * i != li) x[i] = -INFINITY;
    }
    // top-k: k rounds, each takes ONE maximum out (duplicates count, as torch.topk); the k-th one taken is the pivot and
    // everything >= it stays (GPT/utils.py:47-50 keeps ties).  Entries are taken by slot, so -inf ones count too.
    float pivot = -INFINITY, top = -INFINITY;
    uint32_t removed = 0u;
    const bool compact = k > 0 && k < V && k <= 64;      // the k entries taken are handed to lanes 0 .. k-1 as they are found
    float cx = -INFINITY; int cv = 0x7fffffff;
    if (k > 64 && k < V) {
        // many taken entries (up to V - 1): k rounds of arg-max would be O(k) dependent wave reductions.  The pivot is the k-th
        // largest entry = the largest t with #{x >= t} >= k, found by bisection over the ORDER-PRESERVING bit pattern of a float
        // (sign flipped for positives, all bits for negatives; -inf entries count as entries, as they do for torch.topk): 32 counts.
        unsigned key[NPL];
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const unsigned b = __float_as_uint(x[i]);
            key[i] = lane + 64 * i < V ? (b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u)) : 0u;   // 0: below every real entry (-inf is 0x007fffff)
        }
        unsigned lo = 0u, hi = 0xffffffffu;              // count(key >= lo) >= k always holds at lo = 0 once slots beyond V are excluded
        for (int it = 0; it < 32; ++it) {
            const unsigned mid = lo + ((hi - lo) >> 1) + ((hi - lo) & 1u);     // upper middle: the search ends on lo
            int c = 0;
#pragma unroll
            for (int i = 0; i < NPL; ++i) c += (lane + 64 * i < V && key[i] >= mid) ? 1 : 0;
            if ((int)wave_sum((float)c) >= k) lo = mid; else hi = mid - 1u;   // counts <= V: exact in fp32
            if (lo == hi) break;
        }
        const unsigned pb = lo ^ ((lo >> 31) ? 0x80000000u : 0xffffffffu);
        pivot = __uint_as_float(pb);
        float bv = -INFINITY;
#pragma unroll
        for (int i = 0; i < NPL; ++i) bv = fmaxf(bv, x[i]);
        top = wave_max(bv);
    } else if (k > 0 && k < V) {
        for (int r = 0; r < k; ++r) {
            float bv = -INFINITY; int bs = -1;
#pragma unroll
            for (int i = 0; i < NPL; ++i)
                if (!((removed >> i) & 1u) && lane + 64 * i < V && (bs < 0 || x[i] > bv)) { bv = x[i]; bs = i; }
            const float m = wave_max(bs >= 0 ? bv : -INFINITY);
            const unsigned long long cand = __builtin_amdgcn_ballot_w64(bs >= 0 && bv == m);
            if (cand == 0ull) break;                      // fewer than k entries in all (k < V excludes it)
            const int winner = __builtin_ctzll(cand);
            if (lane == winner) removed |= 1u << bs;
            const int vwin = __builtin_amdgcn_readlane(lane + 64 * bs, winner);
            if (compact && lane == r) { cx = m; cv = vwin; }
            if (r == 0) top = m;
            pivot = m;
        }
    } else {
        float bv = -INFINITY;
#pragma unroll
        for (int i = 0; i < NPL; ++i) bv = fmaxf(bv, x[i]);
        top = wave_max(bv);
    }
    // The race.  Its candidates are x >= pivot (the pivot keeps ties; logits < pivot -> -inf).  With k <= 64 the k entries
    // taken sit one per lane already; what is left are entries EQUAL to the pivot that were not taken (rare).  ONE copy of
    // the noise + score code, in a loop that runs once in the common case: a single wave runs this path cold, so its
    // instruction footprint is its latency (an unrolled per-slot version measured 14 us, mostly instruction fetch).
    float bv = -INFINITY; int bi = 0x7fffffff;
    uint32_t pend = 0u;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const bool in = lane + 64 * i < V && x[i] > -INFINITY && (compact ? (!((removed >> i) & 1u) && x[i] == pivot) : x[i] >= pivot);
        pend |= in ? 1u << i : 0u;
    }
    bool hc = compact && cx > -INFINITY;
    while (__builtin_amdgcn_ballot_w64(hc || pend != 0u) != 0ull) {
        float tx = cx; int tv = cv;
        bool act = hc;
        if (!hc && pend != 0u) {
#pragma unroll
            for (int i = NPL - 1; i >= 0; --i)
                if ((pend >> i) & 1u) { tx = x[i]; tv = lane + 64 * i; }
            pend &= pend - 1u;                           // the lowest pending slot is the one just picked
            act = true;
        }
        hc = false;
        if (act) {
            const float u = t2s_uniform(seed_lo, seed_hi, slot, pos, stp, (uint32_t)tv);
            const float sc = (tx - top) - logf(-logf(u));
            if (sc > bv || (sc == bv && tv < bi)) { bv = sc; bi = tv; }
        }
    }
    argmax_step<32>(bv, bi); argmax_step<16>(bv, bi); argmax_step<8>(bv, bi);
    argmax_step<4>(bv, bi); argmax_step<2>(bv, bi); argmax_step<1>(bv, bi);
    return bi < V ? bi : 0;
}

static __global__ __launch_bounds__(256) void t2s_token_kernel(TokenArgs a) {
    __shared__ int s_tok;
    const int b = blockIdx.x, tid = threadIdx.x;
    // ---- device sampling (ctl[0] == 2): temperature, top-k with the reference's tie rule, then the
    // exponential race of GPT/utils.py:56-59 as a Gumbel argmax: argmax p/q, q ~ Exp(1)  ==  argmax (x - log q)
    int sampled = -1;
    if (a.ctl[0] == 2 && tid < 64) {
        const float* lg = a.logits + (size_t)b * a.V;
        const uint32_t pos = (uint32_t)a.kv_len[b], stp = (uint32_t)a.step[b];
        // the noise stream of this sequence: tok_override[b] - 1 when the caller set one (> 0; continuous batching keys it
        // by REQUEST, so what a request samples does not depend on the slot, the refill order or the rank it runs on),
        // else the slot index
        const int64_t sid = a.tok_override[b];
        const uint32_t stream = sid > 0 ? (uint32_t)(sid - 1) : (uint32_t)b;
        sampled = a.V <= 64 * 17 ? t2s_sample_wave<17>(lg, a.V, a.fctl[1], a.fctl[2], a.ctl[4], (uint32_t)a.ctl[5], (uint32_t)a.ctl[6], stream, pos, stp)
                                 : t2s_sample_wave<32>(lg, a.V, a.fctl[1], a.fctl[2], a.ctl[4], (uint32_t)a.ctl[5], (uint32_t)a.ctl[6], stream, pos, stp);
    }
    if (tid == 0) {
        int tok;
        if (a.ctl[0] == 2) {
            tok = sampled;
        } else if (a.ctl[0] != 0) {
            tok = (int)a.tok_override[b];
        } else {
            float bv = -INFINITY;
            tok = 0x7fffffff;
            for (int p = 0; p < kNP; ++p) {
                TokPart t = a.tokpart[(size_t)b * kNP + p];
                if (t.v > bv || (t.v == bv && t.idx < tok)) { bv = t.v; tok = t.idx; }
            }
        }
        if (tok < 0 || tok >= a.V) tok = 0;
        s_tok = tok;
        const int64_t n = a.kv_len[b];
        if (n >= 0 && n <= a.T) a.pre_tokens[(size_t)b * (a.T + 1) + n] = tok;
        if (a.ctl[2] != 0 && n >= 0) a.seen[(size_t)b * a.V + tok] = 1;   // a parked slot's `seen` belongs to its refill
        if (tok == a.eos && a.eos_at[b] < 0) { a.eos_at[b] = a.step[b]; eos_publish(a.eos_host, b, a.step[b]); }
        if (a.advance) a.step[b] += 1;
    }
    __syncthreads();
    const int tok = s_tok;
    int64_t pos = a.kv_len[b] - a.x_len[b];
    if (pos < 0) pos += a.n_pos;  // torch negative indexing of the PE table (idle slots only)
    if (pos < 0) pos = 0;
    if (pos >= a.n_pos) pos = a.n_pos - 1;
    for (int c = tid; c < kD; c += 256)
        a.xcur[(size_t)b * kD + c] = a.emb[(size_t)tok * kD + c] * 1.0f + a.pe[(size_t)pos * kD + c];
}

}  // namespace gsv
