// cgemm: the Generator's WIDE resblock convolutions (192 / 256 / 384 channels; ResBlock1, module/modules.py:190-203) as an
// LDS-tiled MFMA GEMM for gfx950 -- both operands through LDS, weights streamed, activations staged once per channel chunk.
//
// Why not wconv.h here: wconv keeps a 32-channel slice's weights in a wave's registers for the block's lifetime, which is what
// pays at <= 128 channels and tens of thousands of rows.  At 192-384 channels a slice's weights no longer fit (the contraction
// has to be split between waves and the output slices between blocks that each re-stage the input), and the first stages have
// few rows (5 000 per 10 s of audio at 256 / 384 channels): measured 300-600 TF/s.  A conv is a GEMM whose A operand is shared
// by its taps: Y[r][co] = sum_tap sum_ci W[tap][co][ci] * X[r + (tap - hk) * d][ci].  So:
//   * block tile = BM rows x BN output channels, 4 waves (2 along rows x 2 along channels), wave tile 32 RT rows x BN/2 channels
//     (RT = 2: 128-row tiles; RT = 4: 256-row tiles, half the weight traffic per row, used at 256 channels), two blocks per CU,
//     v_mfma_f32_32x32x16_bf16 with the WEIGHTS as the A operand (M = output channel) and the ROWS as N: a lane's 16 accumulator
//     registers are 16 consecutive channels of one row (weight rows are permuted at pack time), stored as two 16-byte pieces;
//   * the contraction runs over (64-channel chunk, tap): the activation rows of a chunk (BM + halo rows) are staged ONCE and
//     serve every tap as a row-shifted window; a (tap, chunk) weight tile (BN x 64, 16-24 KB) is streamed per iteration into a
//     double buffer while the previous one is multiplied;
//   * both LDS images are PLANES of 8 contraction values ([plane][row or channel][16 B]): every MFMA fragment read is 32
//     consecutive 16-byte slots -- conflict-free ds_read_b128 without padding; weights are packed in global memory in exactly
//     this order, so staging a weight tile is a linear copy;
//   * up to three convs of one shape (the resblock branches k = 3 / 7 / 11) share a launch, heaviest first.
// Same arithmetic as the per-conv path: bf16 operands, fp32 accumulation, bias, leaky-ReLU / residual epilogue, bf16 store.
#pragma once
#include "gsv_common.h"

namespace gsv {

struct CGemmArgs {
    const bf16_t *X0, *X1, *X2;   // inputs  [n_rows][ld]
    const uint4 *W0, *W1, *W2;    // packed weights (cgemm_pack_kernel)
    const float *b0, *b1, *b2;    // bias [C]
    const bf16_t *R0, *R1, *R2;   // residual [n_rows][ld] or null
    bf16_t *Y0, *Y1, *Y2;         // outputs [n_rows][ld]
    int k0, k1, k2;               // taps
    int d0, d1, d2;               // dilations
    int nb0, nb1;                 // blocks of branch 0 / 1 (the rest: branch 2)
    int ns0, ns1, ns2;            // K split of a branch's tiles (1 = none): a tile's 64-channel chunks are dealt to ns CONSECUTIVE blocks; the last one
                                  // adds the others' fp32 partial tiles (through `part`, behind `flag`) and owns the epilogue -- few row tiles (10 s of
                                  // audio at 384 channels: 240 blocks of 66 / 42 / 18 iterations on 256 CUs) otherwise wait for the 11-tap blocks
    float* part;                  // [split tile][ns - 1][BM x BN floats] or null
    int* flag;                    // [split tile][ns - 1], zero between launches
    int ld, n_rows;
    float in_slope, out_slope;    // leaky-ReLU on the input / the output (1 = none)
    long long* dbg;               // null, or cycle stamps of block 0 / thread 0 (tools/cg_bench)
};

template <int C, int BN, int BM_ = 128, int RT_ = 2>
struct CgShape {
    static constexpr int BM = BM_, RT = RT_, KC = 64, NPL = KC / 8, NW = 2 * (BM / (32 * RT)), NT = NW * 64;
    static constexpr int XB = NW >= 8 ? 2 : 1;    // activation chunk buffers: the 4-wave shapes keep ONE (<= 80 KB of LDS: two blocks per CU,
                                                  // whose barriers and latencies overlap) and re-stage behind a barrier at a chunk boundary
    static constexpr int WN = BN / 64;            // 32-channel MFMA tiles per wave (2 waves along channels)
    static constexpr int HALO = 64;               // staged rows beyond the tile: (k - 1) * d <= 50
    static constexpr int XROWS = BM + HALO;
    static constexpr int XP = XROWS * 16;         // bytes per activation plane
    static constexpr int XBUF = NPL * XP;
    static constexpr int WP = BN * 16;            // bytes per weight plane
    static constexpr int WBUF = NPL * WP;
    static constexpr int NCH = C / KC;
    static constexpr int TN = C / BN;             // channel tiles
    static constexpr int NWB = NW >= 8 ? 3 : 2;
    static constexpr size_t LDS = XB * (size_t)XBUF + NWB * (size_t)WBUF;
    static constexpr int XV = (NPL * XROWS + NT - 1) / NT;   // 16-byte activation pieces per thread per chunk
    static constexpr int WV = NPL * BN / NT;                  // 16-byte weight pieces per thread per tile
    static_assert(C % KC == 0 && C % BN == 0 && BN % 64 == 0 && (NPL * BN) % NT == 0 && BM % 64 == 0, "shape");
};

// packed weights: [tap][chunk][plane][C channels, permuted inside groups of 32][8 input channels]; the MFMA row m of a 32-channel
// tile holds channel 16 * ((m >> 2) & 1) + (m & 3) + 4 * (m >> 3), so that accumulator register r of lane half hi is channel 16 hi + r
__host__ __device__ inline int cg_chan_of_row(int m) { return 16 * ((m >> 2) & 1) + (m & 3) + 4 * (m >> 3); }

// a pointer the compiler must treat as wave-uniform (SGPR pair): loads through it take the scalar-base form
__device__ __forceinline__ const unsigned char* cg_uniform(const void* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hh = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const unsigned char*>(((unsigned long long)hh << 32) | lo);
}

template <int C>
__global__ void cgemm_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int K) {   // w: torch Conv1d [C][C][K]
    constexpr int NCHK = C / 64;
    const size_t n = (size_t)K * NCHK * 8 * C * 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7);
        size_t q = i >> 3;
        const int row = (int)(q % C); q /= C;
        const int pl = (int)(q % 8); q /= 8;
        const int ch = (int)(q % NCHK);
        const int tap = (int)(q / NCHK);
        const int co = (row & ~31) + cg_chan_of_row(row & 31);
        const int ci = ch * 64 + pl * 8 + e;
        out[i] = f32_to_bf16(w[((size_t)co * C + ci) * K + tap]);
    }
}

template <int C, int BN, int BM = 128, int RT = 2>
__global__ __launch_bounds__((CgShape<C, BN, BM, RT>::NT), 2) void cgemm_kernel(CGemmArgs a) {
    using S = CgShape<C, BN, BM, RT>;
    constexpr int WN = S::WN;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, n = lane & 31, hi = lane >> 5;
    constexpr int WR = 32 * RT;                   // rows per wave
    const int wm = wid % (S::BM / WR), wn = wid / (S::BM / WR);        // wave position: rows [WR wm, + WR), channels [BN/2 * wn, + BN/2)
    // ---- which conv, which tile
    int b = blockIdx.x;
    const int br = b < a.nb0 ? 0 : (b < a.nb0 + a.nb1 ? 1 : 2);
    if (br == 1) b -= a.nb0; else if (br == 2) b -= a.nb0 + a.nb1;
    const bf16_t* X = br == 0 ? a.X0 : (br == 1 ? a.X1 : a.X2);
    const uint4* W = br == 0 ? a.W0 : (br == 1 ? a.W1 : a.W2);
    const float* bias = br == 0 ? a.b0 : (br == 1 ? a.b1 : a.b2);
    const bf16_t* R = br == 0 ? a.R0 : (br == 1 ? a.R1 : a.R2);
    bf16_t* Y = br == 0 ? a.Y0 : (br == 1 ? a.Y1 : a.Y2);
    const int K = br == 0 ? a.k0 : (br == 1 ? a.k1 : a.k2);
    const int d = br == 0 ? a.d0 : (br == 1 ? a.d1 : a.d2);
    const int ns = br == 0 ? a.ns0 : (br == 1 ? a.ns1 : a.ns2);
    const int sp = ns > 1 ? b % ns : 0;           // this block's share of the chunks; the LAST share (sp == ns - 1) owns the epilogue
    if (ns > 1) b /= ns;
    // split tiles are numbered through the launch: branch 0's first, then branch 1's, ... (only split branches count)
    const int tiles0 = a.nb0 / a.ns0, tiles1 = a.nb1 / a.ns1;
    const int stile = b + (br >= 1 && a.ns0 > 1 ? tiles0 : 0) + (br == 2 && a.ns1 > 1 ? tiles1 : 0);
    const int tn = b % S::TN, tm = b / S::TN;
    const int r0 = tm * S::BM, n0 = tn * BN;
    const int c_lo = ns > 1 ? sp * (S::NCH / ns) : 0, c_n = ns > 1 ? S::NCH / ns : S::NCH;
    const int hk = (K - 1) / 2;
    const int xrows = S::BM + 2 * hk * d;         // rows this conv needs staged
    unsigned char* xbuf = lds;
    unsigned char* wbuf = lds + S::XB * S::XBUF;

    // ---- staging helpers.  Every global address is (wave-uniform base, per-thread byte offset computed ONCE): the loads
    // then use the scalar-base form and write no address VGPRs per iteration -- with per-iteration 64-bit address arithmetic
    // hipcc put the addresses into the registers of the previous tile's data and waited vmcnt(0) for it before every issue
    u32x4 xreg[S::XV];
    unsigned xoff[S::XV];
#pragma unroll
    for (int v = 0; v < S::XV; ++v) {
        const int q = v * S::NT + tid, pl = q / S::XROWS, j = q % S::XROWS;
        const int gr = r0 - hk * d + j;
        const bool ok = q < S::NPL * S::XROWS && j < xrows && gr >= 0 && gr < a.n_rows;
        xoff[v] = (unsigned)(((size_t)(ok ? gr : 0) * a.ld + (q < S::NPL * S::XROWS ? pl : 0) * 8) * 2);
    }
    auto x_issue = [&](int ch) {
        const unsigned char* base = cg_uniform(X + ch * S::KC);
#pragma unroll
        for (int v = 0; v < S::XV; ++v) xreg[v] = *reinterpret_cast<const u32x4*>(base + xoff[v]);   // masked at commit
    };
    auto x_commit = [&](int buf) {
#pragma unroll
        for (int v = 0; v < S::XV; ++v) {
            const int q = v * S::NT + tid, pl = q / S::XROWS, j = q % S::XROWS;
            const int gr = r0 - hk * d + j;
            const bool ok = j < xrows && gr >= 0 && gr < a.n_rows;
            u32x4 o = ok ? xreg[v] : u32x4{0u, 0u, 0u, 0u};
            if (a.in_slope != 1.0f) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = __uint_as_float(o[e] << 16), hh = __uint_as_float(o[e] & 0xffff0000u);
                    o[e] = pack_bf16x2(fmaxf(lo, lo * a.in_slope), fmaxf(hh, hh * a.in_slope));
                }
            }
            if (q < S::NPL * S::XROWS) *reinterpret_cast<u32x4*>(xbuf + buf * S::XBUF + pl * S::XP + j * 16) = o;
        }
    };
    // weight tiles go global -> LDS directly (global_load_lds, 16 B per lane, lane-linear: the packed order IS the LDS image), TWO
    // iterations ahead of their use: no registers, no ds_write pass, and a counted vmcnt leaves the newest tile in flight across the
    // barrier (a raw s_barrier: __syncthreads would drain vmcnt(0)).  Register staging of the same tiles made hipcc compute each
    // tile's addresses into the previous tile's data registers and wait vmcnt(0) before every issue.
    unsigned woff[S::WV];
#pragma unroll
    for (int v = 0; v < S::WV; ++v) {
        const int q = v * S::NT + tid, pl = q / BN, c = q % BN;
        woff[v] = (unsigned)(((size_t)pl * C + n0 + c) * 16);
    }
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void gbl_void;
    auto w_issue = [&](int it) {        // it = chunk * K + tap  -> packed tile [tap][chunk] -> LDS buffer it % NWB
        const int ch = c_lo + it / K, tap = it % K;
        const unsigned char* base = reinterpret_cast<const unsigned char*>(W + ((size_t)(tap * S::NCH + ch) * S::NPL) * C);
        unsigned char* dst = wbuf + (it % S::NWB) * S::WBUF + wid * 64 * 16;
#pragma unroll
        for (int v = 0; v < S::WV; ++v)
            __builtin_amdgcn_global_load_lds((gbl_void*)(base + woff[v]), (lds_void*)(dst + v * S::NT * 16), 16, 0, 0);
    };

    f32x16 acc[RT][WN];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int t = 0; t < WN; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

    const int nit = c_n * K;                      // iteration it -> chunk c_lo + it / K, tap it % K
    auto stamp = [&](int i) { if (a.dbg && blockIdx.x == 0 && tid == 0 && i < 64) a.dbg[i] = (long long)__builtin_readcyclecounter(); };
    stamp(0);
    x_issue(c_lo);
    x_commit(c_lo % S::XB);
    constexpr int PF = S::NWB - 1;       // weight tiles in flight beyond the one being multiplied
    w_issue(0);
    if (PF > 1 && nit > 1) w_issue(1);
    if (PF > 1 && nit > 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(S::WV) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" : : : "memory");
    // fragment addresses: B (rows): plane 2 ks + hi, row 64 wm + 32 i + n + tap * d; A (weights): plane 2 ks + hi, channel BN/2 wn + 32 t + n
    const unsigned xlane = (unsigned)(hi * S::XP + (WR * wm + n) * 16);
    const unsigned wlane = (unsigned)(hi * S::WP + ((BN / 2) * wn + n) * 16);
    stamp(1);
    for (int it = 0; it < nit; ++it) {
        stamp(2 + it);
        const int chl = it / K, tap = it - chl * K, ch = c_lo + chl;
        // the buffer of tile it + PF was last read in iteration it - 1, which every wave has left (the barrier below)
#ifndef CG_SKIP_W
        if (it + PF < nit) w_issue(it + PF);
#endif
        const bool xi = tap == 0 && chl + 1 < c_n;
        if (xi) x_issue(ch + 1);             // lands during this chunk's K iterations
        const unsigned char* xb = xbuf + (ch % S::XB) * S::XBUF + xlane + (unsigned)(tap * d) * 16;
        const unsigned char* wb = wbuf + (it % S::NWB) * S::WBUF + wlane;
        // fragments of k-step ks + 1 are read before the MFMAs of k-step ks issue (two register sets)
        u32x4 bf[2][RT], af[2][WN];
        auto ldf = [&](int ks, u32x4 (&b_)[RT], u32x4 (&a_)[WN]) {
#pragma unroll
            for (int i = 0; i < RT; ++i) b_[i] = *reinterpret_cast<const u32x4*>(xb + 2 * ks * S::XP + i * 32 * 16);
#pragma unroll
            for (int t = 0; t < WN; ++t) a_[t] = *reinterpret_cast<const u32x4*>(wb + 2 * ks * S::WP + t * 32 * 16);
        };
        ldf(0, bf[0], af[0]);
#pragma unroll
        for (int ks = 0; ks < S::KC / 16; ++ks) {
            if (ks + 1 < S::KC / 16) ldf(ks + 1, bf[(ks + 1) & 1], af[(ks + 1) & 1]);
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int t = 0; t < WN; ++t)
#ifndef CG_SKIP_MFMA
                    acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[ks & 1][t]), __builtin_bit_cast(bf16x8, bf[ks & 1][i]), acc[i][t], 0, 0, 0);
#else
                    acc[i][t][0] += __uint_as_float(af[ks & 1][t][0] ^ bf[ks & 1][i][1]);
#endif
        }
        if (tap == K - 1 && chl + 1 < c_n) {
            if (S::XB == 1) {      // one activation buffer: everyone has to be done with this chunk before the next one lands in it
                asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" : : : "memory");
            }
            x_commit((ch + 1) % S::XB);
        }
        // tile it + 1 must have landed; the tiles issued after it (and, in the iteration that issued them last, the next chunk's
        // activation loads) may stay in flight
        if (PF > 1 && it + PF < nit) {
            if (xi) asm volatile("s_waitcnt vmcnt(%0)" : : "n"((PF - 1) * S::WV + S::XV) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" : : "n"((PF - 1) * S::WV) : "memory");
        } else if (PF == 1 && xi && it + 1 < nit) {
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(S::XV) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" : : : "memory");      // no LDS read of the next iteration above the barrier
    }
    stamp(2 + nit);
    if (ns > 1) {
        // ---- K split: partial tiles in accumulator order, thread-linear ([(i, t, q4)][thread] 16-byte pieces: coalesced, and the consumer's
        // threads own the same (row, channel) elements).  The producers are EARLIER blocks of the launch than their consumer: they have been
        // dispatched when it spins, and they wait for nobody.  Coherence is PER INSTRUCTION (sc1: the partial tiles are written through and
        // read around the XCDs' non-coherent L2 lines): an agent-scope release / acquire pair would write back and INVALIDATE a whole L2 per
        // hand-off -- measured, the launch got slower (71 -> 101 us): every other block of that XCD lost its weight tiles.
        constexpr size_t TILE_F = (size_t)RT * WN * 16 * S::NT;
        float* pt = a.part + ((size_t)stile * 2) * TILE_F;          // two slots per split tile, whatever ns is
        int* fl = a.flag + (size_t)stile * 2;
        if (sp < ns - 1) {
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int t = 0; t < WN; ++t)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const f32x4 v = {acc[i][t][4 * q4], acc[i][t][4 * q4 + 1], acc[i][t][4 * q4 + 2], acc[i][t][4 * q4 + 3]};
                        float* dstp = pt + (size_t)sp * TILE_F + ((size_t)((i * WN + t) * 4 + q4) * S::NT + tid) * 4;
                        asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dstp), "v"(v) : "memory");
                    }
            asm volatile("s_waitcnt vmcnt(0)" : : : "memory");     // written through ...
            __syncthreads();
            if (tid == 0) __hip_atomic_store(fl + sp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... before the flag says so
            return;
        }
        if (tid == 0) {
            for (int p = 0; p < ns - 1; ++p) {
                int spins = 0;                     // bounded: a launch must never hang the device (a producer that was never dispatched cannot
                while (__hip_atomic_load(fl + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1 && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(2);   // exist: blocks dispatch in order)
            }
        }
        __syncthreads();
        for (int p = 0; p < ns - 1; ++p)           // fixed order: own share + share 0 + share 1
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int t = 0; t < WN; ++t)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const float* srcp = pt + (size_t)p * TILE_F + ((size_t)((i * WN + t) * 4 + q4) * S::NT + tid) * 4;
                        f32x4 v;
                        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(srcp) : "memory");
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][t][4 * q4 + e] += v[e];
                    }
        __syncthreads();                           // everyone has read the partial tiles: the flags go back to zero for the next launch
        if (tid < ns - 1) __hip_atomic_store(fl + tid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- epilogue: lane (row n of tile i, half hi) holds channels n0 + BN/2 wn + 32 t + 16 hi + r
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int gr = r0 + WR * wm + 32 * i + n;
        if (gr >= a.n_rows) continue;
#pragma unroll
        for (int t = 0; t < WN; ++t) {
            const int c0 = n0 + (BN / 2) * wn + 32 * t + 16 * hi;
            float v[16];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + c0 + 4 * q4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * q4 + e] = acc[i][t][4 * q4 + e] + b4[e];
            }
            if (a.out_slope != 1.0f) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], v[r] * a.out_slope);
            }
            if (R) {
                const u32x4 ra = *reinterpret_cast<const u32x4*>(R + (size_t)gr * a.ld + c0), rb = *reinterpret_cast<const u32x4*>(R + (size_t)gr * a.ld + c0 + 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] += __uint_as_float(ra[e] << 16);
                    v[2 * e + 1] += __uint_as_float(ra[e] & 0xffff0000u);
                    v[8 + 2 * e] += __uint_as_float(rb[e] << 16);
                    v[8 + 2 * e + 1] += __uint_as_float(rb[e] & 0xffff0000u);
                }
            }
            u32x4 oa, ob;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                oa[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                ob[e] = pack_bf16x2(v[8 + 2 * e], v[8 + 2 * e + 1]);
            }
            *reinterpret_cast<u32x4*>(Y + (size_t)gr * a.ld + c0) = oa;
            *reinterpret_cast<u32x4*>(Y + (size_t)gr * a.ld + c0 + 8) = ob;
        }
    }
    stamp(3 + nit);
}

}  // namespace gsv
