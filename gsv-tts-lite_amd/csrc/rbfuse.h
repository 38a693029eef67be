// rbfuse: one Generator stage's THREE ResBlock1 branches and their mean in one kernel, for stages of <= 32 channels
// (the reference: x = (RB_3(x) + RB_7(x) + RB_11(x)) / 3, SoVITS/models.py:121-127; RB_k = 3 x [lrelu -> conv(k, dil d) ->
// lrelu -> conv(k, dil 1) -> + x], module/modules.py:190-203).
//
// Why: at 16 / 32 channels these stages hold 320 000 / 160 000 rows per 10 s of audio and are bandwidth-bound -- the
// per-conv kernels (wconv.h) move every activation tensor 18 x 2.5 times (input, output, residual) plus the mean's
// 3 reads + 1 write, and pad 16 channels to the 32-row MFMA tile.  Here a block owns a ROW TILE for the whole stage:
//   * x is read from global once per branch (the 2nd and 3rd time from L2) and the mean is written once: the 18 intermediate
//     tensors of a stage never leave the CU.  The price is halo recompute: 64 rows each side of the tile (the k = 11
//     branch needs 60), 10 % / 20 % more rows at 1280 / 640 rows per tile;
//   * v_mfma_f32_16x16x32_bf16 with the weights as A (16 output channels x 32 contraction values: two taps x 16 channels,
//     or one tap x 32 channels with two output halves) and 16 ROWS as B: no channel padding at C = 16; a conv's weights
//     (<= 22 fragments) live in registers for the pass, the next conv's are fetched meanwhile;
//   * activations sit in LDS in PLANES of 8 channels ([plane][row][16 B]): the B fragment of a tap is 16 consecutive
//     16-byte rows per plane -- conflict-free ds_read_b128 with no padding bytes;
//   * the residual x of a wave's rows stays in a wave-private LDS area in accumulator layout; the three branch results
//     are summed in registers.
// Rounding points are those of the per-conv path (every stored activation is bf16: lrelu(x), lrelu(t1), x + conv2, each
// branch's result, the mean), so the bf16-mode oracle (oracle/gsv_oracle.c ORC_R_VOC) describes both; the fp32 summation
// order inside a conv differs (two taps per MFMA instead of 16-channel k-steps).
#pragma once
#include <type_traits>

#include "gsv_common.h"

namespace gsv {

struct RbFuseArgs {
    const bf16_t* X;      // [n_rows][ld] stage input (output of the stage's transposed conv, no activation applied)
    bf16_t* Y;            // [n_rows][ld] mean of the three branches
    const uint4* W;       // A fragments: [branch][conv 0..5 = pair * 2 + {c1, c2}][step][half][lane]
    const float* B;       // bias [branch][conv][C]
    int wofs[3];          // first fragment of each branch
    int dil[3];           // dilation of c1 in the three pairs (c2: 1)
    int ld, n_rows;
    float slope;          // leaky-ReLU slope inside the resblocks (0.1)
    long long* dbg;       // null, or cycle stamps of block 0 / thread 0 (tools/rb_bench)
};

template <int C>
struct RbShape {
    static constexpr int HV = C / 16;            // 16-channel output halves
    static constexpr int NPL = C / 8;            // 8-channel planes
    static constexpr int NW = 8;                 // waves per block (two per SIMD: one's epilogue runs under the other's MFMAs)
    static constexpr int RW = NW / HV;           // waves along rows; at C = 32 waves w and w + 4 take the two output halves of the same rows
    static constexpr int SR = 16 * RW;           // rows per slot (one 16-row MFMA tile per row wave)
    static constexpr int BN = C == 16 ? 1280 : 640;   // output rows per tile (250 tiles per 10 s of audio either way)
    static constexpr int HALO = 64;
    static constexpr int RC = BN + 2 * HALO;     // rows computed per tile
    static constexpr int GUARD = 32;             // zero rows either side of the computed ones (taps reach 30 rows out)
    static constexpr int ROWS = RC + 2 * GUARD;
    static constexpr int P = ROWS * 16;          // bytes per plane
    static constexpr int BUF = NPL * P;          // one planar activation buffer
    static constexpr int NS = RC / SR;           // slots per wave (slot s of row wave r = rows [SR s + 16 r, +16))
    static constexpr int NO = BN / SR;           // ... of which produce output rows
    static constexpr int XR = RC * C * 2;        // residual area, accumulator layout, wave-private
    static constexpr int BIAS = 18 * C * 4;      // every conv's bias (fp32), read from LDS at the start of its pass
    static constexpr size_t LDS = 2 * (size_t)BUF + XR + BIAS;
    static constexpr int SMAX = C == 16 ? 6 : 11;
    static constexpr int G = 1;                  // slots in flight per wave (the SIMD's other wave fills the gaps of this one's dependent MFMA chain)
    static_assert(RC % SR == 0 && BN % SR == 0 && NS % G == 0, "tile shape");
    static __host__ __device__ constexpr int steps(int K) { return C == 16 ? (K + 1) / 2 : K; }
    // first slot of row wave r that holds output rows: rows [HALO, HALO + BN)
    static __device__ __forceinline__ int first_out_slot(int r) { return (HALO - 16 * r + SR - 1) / SR; }
};

__device__ __forceinline__ f32x4 rb_mma(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// max(v, v * s) as v_med3_f32(v, v * s, +inf): the same value as fmaxf for every non-NaN input, without the v_max(v, v)
// canonicalisation IEEE fmaxf costs per operand
__device__ __forceinline__ float rb_lrelu(float v, float s) { return __builtin_amdgcn_fmed3f(v, v * s, __builtin_inff()); }
__device__ __forceinline__ void rb_unpack4(uint32_t lo, uint32_t hi, float (&f)[4]) {
    f[0] = __uint_as_float(lo << 16); f[1] = __uint_as_float(lo & 0xffff0000u);
    f[2] = __uint_as_float(hi << 16); f[3] = __uint_as_float(hi & 0xffff0000u);
}
// v / 3, correctly rounded, in three operations (q = v * RN(1/3); one exact-remainder correction)
__device__ __forceinline__ float rb_third(float v) {
    const float q = v * 0.333333343f;
    return __builtin_fmaf(__builtin_fmaf(-3.0f, q, v), 0.333333343f, q);
}

// One conv over every computed row of the tile.  MODE 1: c1 of a pair (src = lrelu(x) planes, dst = lrelu(t1) planes);
// MODE 2: c2 (src = lrelu(t1), residual from / new x to the wave's area, dst = lrelu(new x) planes); MODE 3: c2 of the last
// pair (no lrelu(new x): nobody reads it).  KN: taps of the NEXT conv, whose fragments are fetched into `wn` meanwhile.
template <int C, int K, int KN, int MODE, bool EDGE>
__device__ __forceinline__ void rb_pass(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, unsigned char* __restrict__ xr,
                                        const u32x4 (&w)[RbShape<C>::SMAX], u32x4 (&wn)[RbShape<C>::SMAX],
                                        const uint4* __restrict__ wnext, const float* __restrict__ bias, int d, int t0, int n_rows,
                                        float slope) {
    using S = RbShape<C>;
    constexpr int HV = S::HV, ST = S::steps(K), G = S::G, NG = S::NS / G;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    const int rw = wid % S::RW, h = wid / S::RW;
    if constexpr (KN > 0) {
#pragma unroll
        for (int st = 0; st < S::steps(KN); ++st) wn[st] = __builtin_bit_cast(u32x4, wnext[(st * HV + h) * 64 + lane]);
    }
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 16 * h + 4 * g);     // LDS
    constexpr int hk = (K - 1) / 2;
    // B fragment of step st, slot row i0: lane (n, g) reads 8 channels of one row -- C = 16: plane g & 1 of tap 2 st + (g >> 1);
    // C = 32: plane g of tap st
    const unsigned rbase = (C == 16 ? (unsigned)((g & 1) * S::P + (S::GUARD + n + ((g >> 1) - hk) * d) * 16)
                                    : (unsigned)(g * S::P + (S::GUARD + n - hk * d) * 16)) + (unsigned)(16 * rw) * 16;
    const unsigned sstep = (unsigned)((C == 16 ? 2 : 1) * d * 16);
    // accumulator layout: lane (n, g) holds channels 16 h + 4 g .. + 3 of row n -> plane 2 h + (g >> 1), bytes (g & 1) * 8
    const unsigned wbase = (unsigned)((2 * h + (g >> 1)) * S::P + (S::GUARD + 16 * rw + n) * 16 + (g & 1) * 8);
    unsigned char* xrw = xr + (size_t)(rw * HV + h) * 512 + lane * 8;      // + slot * RW * HV * 512
    const int gr0 = t0 - S::HALO + 16 * rw + n;

    auto ldfr = [&](int grp, u32x4 (&bf)[G][ST], uint2 (&res)[G]) {
#pragma unroll
        for (int q = 0; q < G; ++q) {
            const unsigned o = rbase + (unsigned)((grp * G + q) * S::SR) * 16;
#pragma unroll
            for (int st = 0; st < ST; ++st) bf[q][st] = *reinterpret_cast<const u32x4*>(src + o + st * sstep);
            if constexpr (MODE != 1) res[q] = *reinterpret_cast<const uint2*>(xrw + (size_t)(grp * G + q) * (S::RW * HV * 512));
        }
    };
    auto compute = [&](int grp, const u32x4 (&bf)[G][ST], const uint2 (&res)[G]) {
        f32x4 acc[G];
#pragma unroll
        for (int q = 0; q < G; ++q) acc[q] = bv;
#pragma unroll
        for (int st = 0; st < ST; ++st)
#pragma unroll
            for (int q = 0; q < G; ++q) acc[q] = rb_mma(w[st], bf[q][st], acc[q]);
#pragma unroll
        for (int q = 0; q < G; ++q) {
            const int slot = grp * G + q;
            const bool inside = !EDGE || (unsigned)(gr0 + slot * S::SR) < (unsigned)n_rows;
            unsigned char* dp = dst + wbase + (unsigned)(slot * S::SR) * 16;
            if constexpr (MODE == 1) {
                uint2 o;
                o.x = pack_bf16x2(rb_lrelu(acc[q][0], slope), rb_lrelu(acc[q][1], slope));
                o.y = pack_bf16x2(rb_lrelu(acc[q][2], slope), rb_lrelu(acc[q][3], slope));
                if (!inside) o = uint2{0u, 0u};      // "same" padding: the next conv sees zeros outside the sequence
                *reinterpret_cast<uint2*>(dp) = o;
            } else {
                float rf[4];
                rb_unpack4(res[q].x, res[q].y, rf);
                uint2 xn;
                xn.x = pack_bf16x2(acc[q][0] + rf[0], acc[q][1] + rf[1]);
                xn.y = pack_bf16x2(acc[q][2] + rf[2], acc[q][3] + rf[3]);
                if (!inside) xn = uint2{0u, 0u};
                *reinterpret_cast<uint2*>(xrw + (size_t)slot * (S::RW * HV * 512)) = xn;
                if constexpr (MODE == 2) {
                    float xf[4];
                    rb_unpack4(xn.x, xn.y, xf);
                    uint2 o;
                    o.x = pack_bf16x2(rb_lrelu(xf[0], slope), rb_lrelu(xf[1], slope));
                    o.y = pack_bf16x2(rb_lrelu(xf[2], slope), rb_lrelu(xf[3], slope));
                    *reinterpret_cast<uint2*>(dp) = o;
                }
            }
        }
    };
    u32x4 bfa[G][ST], bfb[G][ST];
    uint2 ra[G], rb2[G];
    ldfr(0, bfa, ra);
    for (int grp = 0; grp < NG; grp += 2) {
        if (grp + 1 < NG) ldfr(grp + 1, bfb, rb2);
        compute(grp, bfa, ra);
        if (grp + 2 < NG) ldfr(grp + 2, bfa, ra);
        if (grp + 1 < NG) compute(grp + 1, bfb, rb2);
    }
}

// the six convs of one branch (K taps), then the branch result into the running sum.  x0: the raw stage input of this wave's
// rows in accumulator layout (loaded once per tile): every branch starts from it -- lrelu(x0) into the planes, x0 into the
// wave's residual area -- without touching global memory again.
template <int C, int K, int KNB, bool FIRST, bool EDGE>
__device__ __forceinline__ void rb_branch(const RbFuseArgs& a, int br, int kn_branch_first_ofs, unsigned char* xl, unsigned char* tb, const float* bias_lds,
                                          u32x4 (&wa)[RbShape<C>::SMAX], u32x4 (&wb)[RbShape<C>::SMAX], const uint2 (&x0)[RbShape<C>::NS],
                                          f32x4 (&osum)[RbShape<C>::NO], int t0) {
    using S = RbShape<C>;
    constexpr int HV = S::HV, ST = S::steps(K);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, n = lane & 15, g = lane >> 4;
    const int rw = wid % S::RW, h = wid / S::RW;
    int nst = br * 10 + 1;
    auto stamp = [&]() { if (a.dbg && blockIdx.x == 0 && tid == 0) a.dbg[nst] = (long long)__builtin_readcyclecounter(); ++nst; };
    stamp();
    unsigned char* xr = xl + 2 * S::BUF;     // the evolving x of this wave's rows (residual of every c2): wave-private, accumulator layout
    {
        unsigned char* dp = xl + (unsigned)((2 * h + (g >> 1)) * S::P + (S::GUARD + 16 * rw + n) * 16 + (g & 1) * 8);
#pragma unroll
        for (int s = 0; s < S::NS; ++s) {
            *reinterpret_cast<uint2*>(xr + (size_t)((s * S::RW + rw) * HV + h) * 512 + lane * 8) = x0[s];
            float f[4];
            rb_unpack4(x0[s].x, x0[s].y, f);
            uint2 o;
            o.x = pack_bf16x2(rb_lrelu(f[0], a.slope), rb_lrelu(f[1], a.slope));
            o.y = pack_bf16x2(rb_lrelu(f[2], a.slope), rb_lrelu(f[3], a.slope));
            *reinterpret_cast<uint2*>(dp + (unsigned)(s * S::SR) * 16) = o;
        }
    }
    __syncthreads();
    stamp();
    const uint4* W = a.W + (size_t)a.wofs[br] * 64;
    const float* B = bias_lds + (size_t)br * 6 * C;
    constexpr int CS = ST * HV * 64;     // uint4 per conv of this branch
    // conv c's fragments are in wa (even c) / wb (odd c); every pass fetches the next conv's
    rb_pass<C, K, K, 1, EDGE>(xl, tb, xr, wa, wb, W + 1 * CS, B + 0 * C, a.dil[0], t0, a.n_rows, a.slope);
    __syncthreads();
    stamp();
    rb_pass<C, K, K, 2, EDGE>(tb, xl, xr, wb, wa, W + 2 * CS, B + 1 * C, 1, t0, a.n_rows, a.slope);
    __syncthreads();
    stamp();
    rb_pass<C, K, K, 1, EDGE>(xl, tb, xr, wa, wb, W + 3 * CS, B + 2 * C, a.dil[1], t0, a.n_rows, a.slope);
    __syncthreads();
    stamp();
    rb_pass<C, K, K, 2, EDGE>(tb, xl, xr, wb, wa, W + 4 * CS, B + 3 * C, 1, t0, a.n_rows, a.slope);
    __syncthreads();
    stamp();
    rb_pass<C, K, K, 1, EDGE>(xl, tb, xr, wa, wb, W + 5 * CS, B + 4 * C, a.dil[2], t0, a.n_rows, a.slope);
    __syncthreads();
    stamp();
    // the last conv fetches the NEXT branch's first conv into wa
    rb_pass<C, K, KNB, 3, EDGE>(tb, xl, xr, wb, wa, a.W + (size_t)kn_branch_first_ofs * 64, B + 5 * C, 1, t0, a.n_rows, a.slope);
    stamp();
    // ---- branch result of this wave's output rows into the running sum (wave-private: no barrier needed before it)
    // output rows start at slot 1, or at slot 0 for the row waves past the halo (C = 16: waves 4..7); wave-uniform
    auto add = [&](auto s0c) {
        constexpr int s0 = decltype(s0c)::value;
#pragma unroll
        for (int so = 0; so < S::NO; ++so) {
            const uint2 r = *reinterpret_cast<const uint2*>(xr + (size_t)(((so + s0) * S::RW + rw) * HV + h) * 512 + lane * 8);
            float f[4];
            rb_unpack4(r.x, r.y, f);
#pragma unroll
            for (int e = 0; e < 4; ++e) osum[so][e] = FIRST ? f[e] : osum[so][e] + f[e];
        }
    };
    if (S::first_out_slot(rw) == 0) add(std::integral_constant<int, 0>{});
    else add(std::integral_constant<int, 1>{});
    __syncthreads();   // everyone is done with the planes before the next branch stages into them
    stamp();
}

template <int C, bool EDGE>
__device__ __forceinline__ void rb_tile(const RbFuseArgs& a, unsigned char* lds, int t0) {
    using S = RbShape<C>;
    constexpr int HV = S::HV;
    unsigned char* xl = lds;
    unsigned char* tb = lds + S::BUF;
    const float* bl = reinterpret_cast<const float*>(lds + 2 * S::BUF + S::XR);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, n = lane & 15, g = lane >> 4;
    const int rw = wid % S::RW, h = wid / S::RW;
    u32x4 wa[S::SMAX], wb[S::SMAX];
    f32x4 osum[S::NO];
    uint2 x0[S::NS];
#pragma unroll
    for (int s = 0; s < S::NS; ++s) {
        const int gr = t0 - S::HALO + s * S::SR + 16 * rw + n;
        const bool in = !EDGE || (unsigned)gr < (unsigned)a.n_rows;
        x0[s] = in ? *reinterpret_cast<const uint2*>(a.X + (size_t)(EDGE ? (in ? gr : 0) : gr) * a.ld + 16 * h + 4 * g) : uint2{0u, 0u};
    }
    {   // the first branch's first conv
        const uint4* W = a.W + (size_t)a.wofs[0] * 64;
#pragma unroll
        for (int st = 0; st < S::steps(3); ++st) wa[st] = __builtin_bit_cast(u32x4, W[(st * HV + h) * 64 + lane]);
    }
    rb_branch<C, 3, 7, true, EDGE>(a, 0, a.wofs[1], xl, tb, bl, wa, wb, x0, osum, t0);
    rb_branch<C, 7, 11, false, EDGE>(a, 1, a.wofs[2], xl, tb, bl, wa, wb, x0, osum, t0);
    rb_branch<C, 11, 0, false, EDGE>(a, 2, 0, xl, tb, bl, wa, wb, x0, osum, t0);
    const int s0 = S::first_out_slot(rw);
#pragma unroll
    for (int so = 0; so < S::NO; ++so) {
        const int gr = t0 - S::HALO + (so + s0) * S::SR + 16 * rw + n;
        if (!EDGE || gr < a.n_rows) {
            uint2 o;
            o.x = pack_bf16x2(rb_third(osum[so][0]), rb_third(osum[so][1]));
            o.y = pack_bf16x2(rb_third(osum[so][2]), rb_third(osum[so][3]));
            *reinterpret_cast<uint2*>(a.Y + (size_t)gr * a.ld + 16 * h + 4 * g) = o;
        }
    }
}

template <int C>
__global__ __launch_bounds__(512, 2) void rbfuse_kernel(RbFuseArgs a) {
    using S = RbShape<C>;
    constexpr int NT = S::NW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * S::BN;
    if (a.dbg && blockIdx.x == 0 && tid == 0) a.dbg[0] = (long long)__builtin_readcyclecounter();
    // guard rows: zero, and never written again
    for (int q = tid; q < 2 * S::NPL * 2 * S::GUARD; q += NT) {
        const int buf = q / (S::NPL * 2 * S::GUARD), r = q % (S::NPL * 2 * S::GUARD), pl = r / (2 * S::GUARD), gi = r % (2 * S::GUARD);
        const int row = gi < S::GUARD ? gi : S::RC + gi;
        *reinterpret_cast<u32x4*>(lds + (size_t)buf * S::BUF + pl * S::P + row * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    for (int q = tid; q < 18 * C; q += NT) reinterpret_cast<float*>(lds + 2 * S::BUF + S::XR)[q] = a.B[q];
    __syncthreads();
    // a tile whose computed rows all lie inside the sequence needs no "zero outside the sequence" selects (block-uniform)
    if (t0 - S::HALO >= 0 && t0 + S::BN + S::HALO <= a.n_rows) rb_tile<C, false>(a, lds, t0);
    else rb_tile<C, true>(a, lds, t0);
    if (a.dbg && blockIdx.x == 0 && tid == 0) a.dbg[31] = (long long)__builtin_readcyclecounter();
}

// fp32 torch-layout Conv1d weights [cout][cin][K] of the 18 convs -> A fragments (see rb_pass): one thread per bf16 value
struct RbPackArgs {
    const float* w[18];   // branch-major: branch * 6 + pair * 2 + {c1, c2}
    const float* b[18];
    int k[3];
    int creal;            // channels present in the tensors (24 for the padded 32-channel stage)
    uint4* W;
    float* B;
    int wofs[3];
};
template <int C>
__global__ void rbfuse_pack_kernel(RbPackArgs p) {
    using S = RbShape<C>;
    constexpr int HV = S::HV;
    const int conv = blockIdx.x, br = conv / 6, K = p.k[br], ST = S::steps(K);
    bf16_t* out = reinterpret_cast<bf16_t*>(p.W + ((size_t)p.wofs[br] + (size_t)(conv % 6) * ST * HV) * 64);
    const float* w = p.w[conv];
    for (int idx = threadIdx.x; idx < ST * HV * 64 * 8; idx += blockDim.x) {
        const int e = idx & 7, lane = (idx >> 3) & 63, fh = idx >> 9, h = fh % HV, st = fh / HV;
        const int m = lane & 15, g = lane >> 4;
        int o, ci, tap;
        if (C == 16) { o = m; ci = 8 * (g & 1) + e; tap = 2 * st + (g >> 1); }
        else { o = 16 * h + m; ci = 8 * g + e; tap = st; }
        float v = 0.f;
        if (tap < K && o < p.creal && ci < p.creal) v = w[((size_t)o * p.creal + ci) * K + tap];
        out[idx] = f32_to_bf16(v);
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x) p.B[(size_t)conv * C + c] = (c < p.creal && p.b[conv]) ? p.b[conv][c] : 0.f;
}

}  // namespace gsv
