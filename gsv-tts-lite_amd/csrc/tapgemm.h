// tapgemm: the one MFMA contraction kernel behind every dense op outside the decode step --
// Linear layers of the GPT prefill, Conv1d / 1x1 conv / ConvTranspose1d of the SoVITS flow and
// Generator -- written for gfx950 matrix cores.
//
//   Y[n*omul + r][m] = epilogue( sum_t sum_c  W_{r,t}[m][c] * pre(X[n + shift(r,t)][c]) )
//
// * activations are CHANNELS-LAST ([time][channel]), so both MFMA operands are 16-byte
//   contiguous per lane: B fragment = 8 bf16 (4 f32) consecutive channels of one time row,
//   A fragment = the same slice of one weight row.  Weights are pre-packed at load time in
//   fragment order ([phase][tap][mtile][kstep][lane][16 B]) so a wave's A load is one 1 KiB line.
// * a Conv1d tap is a row shift of X (zero rows outside [0, n_in)); a ConvTranspose1d of stride u
//   is u phases r, each a small conv with ceil(k/u) taps, writing rows n*u + r.
// * bf16 mode: v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  fp32 parity mode:
//   v_mfma_f32_32x32x2_f32, which is an exact f32 fma chain (MI355X_MICROARCH.md), 4 per 16 B.
// * epilogue fuses bias, conditioning add (broadcast or per-row), residual, mask, scale,
//   activation and accumulate; prologue fuses leaky-ReLU on the input.
// * one wave = (WM*32 channels) x (WN*32 rows) accumulators; 4 waves side by side along rows.
//
// D fragment (32x32): column j = lane&31 is the time row, register q holds MFMA row
// (q&3) + 8*(q>>2) + 4*(lane>>5).  The packer permutes weight rows inside each 32-row tile so that
// MFMA row r carries output channel 16*((r>>2)&1) + (r&3) + 4*(r>>3): register q of a lane is then
// channel 16*(lane>>5) + q -- 16 consecutive channels of one row per lane, 32-byte runs in the
// channels-last output.
#pragma once
#include "gsv_common.h"

namespace gsv {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };

struct TapGemmArgs {
    const void* X;       // [n_in][ldx]
    int ldx;             // elements per input row
    int n_in;            // valid input rows
    int cin;             // channels contracted (multiple of the k-step)
    const void* W;       // packed fragments, followed by one all-zero fragment
    int cout;            // logical output channels
    int mtiles;          // ceil(cout / 32)
    int ntaps;           // taps per phase
    int nphase;          // 1 (conv/linear) or u (transposed conv)
    // row shift of tap t in phase r: conv (tu == 0): t*tstep - tpad ; transposed conv of stride tu:
    // (r + tpad)/tu - t.  Arithmetic instead of tables: a dynamically indexed kernarg array goes to
    // scratch, and the inner loop must not divide.
    int tstep, tpad, tu;
    float in_slope;      // leaky-relu slope applied to X on load (1 = none)
    // epilogue
    const float* bias;   // [cout] or null
    const void* add;     // conditioning term, element type = OT? no: float. [rows or 1][ld_add]
    int ld_add;          // row stride of add (0 = broadcast one row)
    const int* add_index;// null, or: output row r takes add row add_index[r] (per-frame conditioning with few distinct rows)
    const void* res;     // residual, same type/layout as output (row index n*omul + r), or null
    int ld_res;
    const float* mask;   // [n_out rows] multiplies the result, or null
    float scale;         // result *= scale (after everything else)
    int act;
    int accumulate;      // Y += result instead of Y = result
    void* Y;             // [n_rows*omul][ldy]
    int ldy;
    int n_rows;          // rows n computed per phase: n in [0, n_rows)
    int omul;            // output row = n*omul + phase
    // Up to 3 independent convolutions of the same shape class in ONE launch (blockIdx.z selects
    // the branch; only for nphase == 1): the three resblocks of a Generator stage share input
    // shape and channel counts but differ in kernel size / dilation / weights / buffers.
    int nbranch;
    const void *X1, *X2, *W1, *W2, *res1, *res2;
    const float *bias1, *bias2;
    void *Y1, *Y2;
    int ntaps1, ntaps2, tstep1, tstep2, tpad1, tpad2;
    int dbg;             // bench harness only: 1 = skip staging, 2 = skip the MFMA loop, 4 = skip the epilogue
};

template <typename CT> struct MfmaK;
template <> struct MfmaK<float> { static constexpr int KS = 8; };     // channels per k-step
template <> struct MfmaK<bf16_t> { static constexpr int KS = 16; };

// leaky-ReLU for 0 < slope < 1 (slope == 1: identity)
__device__ __forceinline__ float lrelu(float v, float slope) { return fmaxf(v, v * slope); }

template <typename CT> struct Mma;
template <> struct Mma<float> {
    using AF = f32x4;
    static __device__ __forceinline__ void run(f32x16& acc, const f32x4& a, const f32x4& b) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
    }
};
template <> struct Mma<bf16_t> {
    using AF = u32x4;
    static __device__ __forceinline__ void run(f32x16& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                      acc, 0, 0, 0);
    }
};

// 4 consecutive output channels of one row: raw load (issued early), widen, narrow + store
template <typename OT> struct Out4;
template <> struct Out4<float> {
    using Raw = f32x4;
    static __device__ __forceinline__ Raw raw(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void widen(const Raw& t, float (&v)[4]) { v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
        f32x4 t = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p) = t;
    }
};
template <> struct Out4<bf16_t> {
    using Raw = uint2;
    static __device__ __forceinline__ Raw raw(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
    static __device__ __forceinline__ void widen(const Raw& t, float (&v)[4]) {
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[4]) {
        uint2 t;
        t.x = pack_bf16x2(v[0], v[1]);
        t.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(p) = t;
    }
};

// Staging: 16 bytes of CT operands per call from input type IT, split into the raw global load
// (all of a thread's loads are issued back to back) and the conversion + leaky-ReLU prologue
// (applied once per element here instead of once per tap).
template <typename IT, typename CT> struct Stage16;
template <> struct Stage16<float, float> {
    static constexpr int E = 4;
    using Raw = f32x4;
    static __device__ __forceinline__ Raw raw(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ u32x4 finish(const Raw& r, bool ok, float slope) {
        f32x4 v = r;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ok ? lrelu(v[i], slope) : 0.f;
        return __builtin_bit_cast(u32x4, v);
    }
};
template <> struct Stage16<float, bf16_t> {
    static constexpr int E = 8;
    struct Raw { f32x4 a, b; };
    static __device__ __forceinline__ Raw raw(const float* p) {
        return Raw{*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4)};
    }
    static __device__ __forceinline__ u32x4 finish(const Raw& r, bool ok, float slope) {
        u32x4 o;
        o[0] = pack_bf16x2(lrelu(r.a[0], slope), lrelu(r.a[1], slope));
        o[1] = pack_bf16x2(lrelu(r.a[2], slope), lrelu(r.a[3], slope));
        o[2] = pack_bf16x2(lrelu(r.b[0], slope), lrelu(r.b[1], slope));
        o[3] = pack_bf16x2(lrelu(r.b[2], slope), lrelu(r.b[3], slope));
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = ok ? o[i] : 0u;
        return o;
    }
};
template <> struct Stage16<bf16_t, bf16_t> {
    static constexpr int E = 8;
    using Raw = u32x4;
    static __device__ __forceinline__ Raw raw(const bf16_t* p) { return *reinterpret_cast<const u32x4*>(p); }
    static __device__ __forceinline__ u32x4 finish(const Raw& r, bool ok, float slope) {
        u32x4 v = r;
        if (slope != 1.0f) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                v[i] = pack_bf16x2(lrelu(__uint_as_float(v[i] << 16), slope), lrelu(__uint_as_float(v[i] & 0xffff0000u), slope));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ok ? v[i] : 0u;
        return v;
    }
};

// IT: input element type, CT: MFMA operand / packed-weight type, OT: output (and residual) type.
// Block = 4 waves side by side along rows; wave tile = (WM*32 channels) x (WN*32 rows).
// The input rows the block needs for ALL taps of its phase ([n0+smin, n0+BN+smax)) are staged in LDS
// in chunks of KC channels (row stride KC*sizeof(CT)+16 bytes: consecutive rows land on consecutive
// 16-byte slots, so the 32-row B-fragment read is conflict-free); B fragments come from LDS, A
// fragments (fragment-ordered weights, L2-resident, identical for the 4 waves) from global with a
// one-iteration register prefetch.
// SPLITK: the 4 waves of a block share ONE (WM x WN) output tile and split the k-steps of every
// chunk between them (partials merged through LDS in wave order, wave 0 runs the epilogue): 4x the
// parallelism and a 4x shorter dependent chain for short sequences (prefill GEMMs, flow, GEMV).
// MB: waves stacked along the channel (M) axis -- 4/MB waves side by side along rows; waves of one
// row group share B fragments through LDS, waves of one channel group share A lines through L1.
// OCC: blocks the kernel is compiled to co-reside per CU (register budget 512/OCC per lane).
template <typename IT, typename CT, typename OT, int WM, int WN, int KCB = 256, bool SPLITK = false, int MB = 1,
          int OCC = 1, int PFT = 4>
static __global__ __launch_bounds__(256, OCC) void tapgemm_kernel(TapGemmArgs a) {
    constexpr int KS = MfmaK<CT>::KS;
    constexpr int E = KS / 2;                 // elements per lane per fragment
    constexpr int KC = KCB / (int)sizeof(CT); // channels staged per chunk
    constexpr int RS = KC * (int)sizeof(CT) + 16;  // LDS row stride in bytes
    static_assert(!SPLITK || MB == 1, "split-K blocks share one tile");
    constexpr int NBW = SPLITK ? 1 : 4 / MB;  // waves along rows
    constexpr int BN = NBW * WN * 32;
    using AF = typename Mma<CT>::AF;
    using ST = Stage16<IT, CT>;
    constexpr int NV = ((BN + 64) * (KCB / 16) + 255) / 256;  // staging vectors per thread (tap span <= 64 rows)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // branch select (scalar): branches exist only for plain convs, where the phase index is 0
    const int br = a.nbranch > 1 ? (int)blockIdx.z : 0;
    const int phase = a.nbranch > 1 ? 0 : (int)blockIdx.z;
    const void* Xv = br == 0 ? a.X : (br == 1 ? a.X1 : a.X2);
    const void* Wv = br == 0 ? a.W : (br == 1 ? a.W1 : a.W2);
    const void* Rv = br == 0 ? a.res : (br == 1 ? a.res1 : a.res2);
    void* Yv = br == 0 ? a.Y : (br == 1 ? a.Y1 : a.Y2);
    const float* Bv = br == 0 ? a.bias : (br == 1 ? a.bias1 : a.bias2);
    const int ntaps = br == 0 ? a.ntaps : (br == 1 ? a.ntaps1 : a.ntaps2);
    const int tstep = br == 0 ? a.tstep : (br == 1 ? a.tstep1 : a.tstep2);
    const int tpad = br == 0 ? a.tpad : (br == 1 ? a.tpad1 : a.tpad2);
    const int nb0 = blockIdx.x * BN;          // first output row of the block
    const int wrow = SPLITK ? 0 : (wid % NBW) * (WN * 32);  // first row of the wave inside the block tile
    const int n0 = nb0 + wrow;                // first output row of the wave
    const int mt0 = blockIdx.y * (MB * WM) + (SPLITK ? 0 : wid / NBW) * WM;
    const int j = lane & 31, hf = lane >> 5;
    const int ksteps = a.cin / KS;
    const IT* X = reinterpret_cast<const IT*>(Xv);
    const uint4* Wp = reinterpret_cast<const uint4*>(Wv);

    const int sbase = a.tu > 0 ? (phase + tpad) / a.tu : -tpad;       // shift of tap 0
    const int sstep = a.tu > 0 ? -1 : tstep;                            // shift increment per tap
    const int slast = sbase + (ntaps - 1) * sstep;
    const int smin = min(sbase, slast), smax = max(sbase, slast);
    const int rows = BN + smax - smin;        // staged rows
    const int row_base = nb0 + smin;          // global row of staged row 0

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int k = 0; k < WN; ++k)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][k][q] = 0.f;

    const bool wave_live = n0 < a.n_rows && mt0 < a.mtiles;
    for (int c0 = 0; c0 < a.cin; c0 += KC) {
        const int kc = min(KC, a.cin - c0);           // channels in this chunk (multiple of KS)
        const int tpr = kc / ST::E;                    // threads (16-byte vectors) per staged row, <= 16
        const int rpp = 256 / tpr;                     // rows per pass
        const int cv = tid % tpr, r_first = tid / tpr;
        const bool tlive = tid < rpp * tpr;
        __syncthreads();                               // previous chunk's readers are done
        // every load of the thread goes out before the first is consumed: one memory latency per
        // chunk instead of one per row
        typename ST::Raw raw[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (v * rpp < rows && !(a.dbg & 1)) {
                const int grow = row_base + r_first + v * rpp;
                const bool ok = grow >= 0 && grow < a.n_in;
                raw[v] = ST::raw(X + (size_t)(ok ? grow : 0) * a.ldx + c0 + cv * ST::E);
            }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (v * rpp < rows && !(a.dbg & 1)) {
                const int r = r_first + v * rpp;
                const int grow = row_base + r;
                const bool ok = grow >= 0 && grow < a.n_in;
                if (tlive && r < rows) *reinterpret_cast<u32x4*>(lds + (size_t)r * RS + cv * 16) = ST::finish(raw[v], ok, a.in_slope);
            }
        }
        __syncthreads();
        if constexpr (!SPLITK) {
            if (wave_live && !(a.dbg & 2)) {
                // Branch-free software pipeline.  The (tap, k-step) space of the chunk is walked flat in
                // groups of PF iterations; a group's weight fragments are fetched one whole group ahead
                // (two register sets, ping-pong), and iterations past the end fetch the all-zero fragment
                // stored behind the conv's weights, so the loop body has no control flow: the compiler
                // can count its outstanding loads exactly (no vmcnt(0) drains) and hoist LDS reads over
                // the previous iteration's MFMAs.
                const int kst = kc / KS;                   // k-steps in this chunk
                const int ks0 = c0 / KS;
                const int nit = ntaps * kst;
                constexpr int PF = PFT;
                const size_t tap_stride = (size_t)a.mtiles * ksteps * 64;
                const uint4* wl = Wp + lane;
                size_t woff[WM];                           // fragment offset of (tap 0, k-step ks0) per m-tile
#pragma unroll
                for (int i = 0; i < WM; ++i)
                    woff[i] = (((size_t)phase * ntaps * a.mtiles + min(mt0 + i, a.mtiles - 1)) * ksteps + ks0) * 64;
                const size_t wzero = (size_t)a.nphase * ntaps * a.mtiles * ksteps * 64;
                const unsigned lb = (unsigned)(wrow + j) * RS + hf * 16;   // lane's row and k-half inside a staged row
                int tl = 0, ksl = 0, il = 0;              // load cursor
                int tc = 0, ksc = 0;                       // compute cursor
                uint4 wa[PF][WM], wb[PF][WM];
                auto fetch = [&](uint4 (&dst)[PF][WM]) {
#pragma unroll
                    for (int u = 0; u < PF; ++u) {
                        const bool ok = il < nit;
                        const size_t step = (size_t)tl * tap_stride + (size_t)ksl * 64;
#pragma unroll
                        for (int i = 0; i < WM; ++i) dst[u][i] = wl[ok ? woff[i] + step : wzero];
                        ++il;
                        ++ksl;
                        const bool wrap = ksl == kst;
                        ksl = wrap ? 0 : ksl;
                        tl += wrap ? 1 : 0;
                    }
                };
                auto compute = [&](const uint4 (&w)[PF][WM]) {
#pragma unroll
                    for (int u = 0; u < PF; ++u) {
                        const int tcc = min(tc, ntaps - 1);
                        const unsigned so = (unsigned)(sbase + tcc * sstep - smin) * RS + (unsigned)ksc * (KS * (int)sizeof(CT));
                        AF bf[WN];
#pragma unroll
                        for (int k = 0; k < WN; ++k)
                            bf[k] = __builtin_bit_cast(AF, *reinterpret_cast<const u32x4*>(lds + lb + so + k * 32 * RS));
#pragma unroll
                        for (int i = 0; i < WM; ++i) {
                            const AF af = __builtin_bit_cast(AF, w[u][i]);
#pragma unroll
                            for (int k = 0; k < WN; ++k) Mma<CT>::run(acc[i][k], af, bf[k]);
                        }
                        ++ksc;
                        const bool wrap = ksc == kst;
                        ksc = wrap ? 0 : ksc;
                        tc += wrap ? 1 : 0;
                    }
                };
                fetch(wa);
                for (int g = 0;;) {
                    fetch(wb);
                    compute(wa);
                    g += PF;
                    if (g >= nit) break;
                    fetch(wa);
                    compute(wb);
                    g += PF;
                    if (g >= nit) break;
                }
            }
        } else {
            if (wave_live) {
                const int kst = kc / KS;                   // k-steps in this chunk
                const int ks0 = c0 / KS;
                constexpr int STEP = SPLITK ? 4 : 1;
                // iteration space: (tap t, k-step ks) walked incrementally -- no division in the loop
                const uint4* wbase[WM];
#pragma unroll
                for (int i = 0; i < WM; ++i)
                    wbase[i] = Wp + (((size_t)phase * ntaps * a.mtiles + min(mt0 + i, a.mtiles - 1)) * ksteps + ks0) * 64 + lane;
                const size_t tap_stride = (size_t)a.mtiles * ksteps * 64;
                // Weight fragments come from L2 (~500+ cycles) and a block holds only 1-2 waves per SIMD, so
                // they are fetched a whole GROUP of PF iterations ahead: PF loads in flight cover PF
                // iterations of LDS reads + MFMAs.  Two cursors walk (tap, k-step): load and compute.
                // split-K: every k-step a wave owns in a chunk is requested at once (2 at 128 bf16 channels per chunk; 8 at the fp32 tile's 256: with two in
                // flight that chunk was four dependent L2 round trips, round 6) -- a prefetch depth, not an order: results are unchanged
                constexpr int PF = SPLITK ? ((KC / KS / 4) >= 8 ? 8 : 2) : (WM == 1 ? 8 : 4);
                int tl = 0, ksl = SPLITK ? wid : 0;          // load cursor
                while (ksl >= kst) { ksl -= kst; ++tl; }
                int tc = tl, ksc = ksl;                      // compute cursor
                uint4 wa[PF][WM], wn[PF][WM];
                auto fetch = [&](uint4 (&dst)[PF][WM]) {
#pragma unroll
                    for (int u = 0; u < PF; ++u) {
                        if (tl < ntaps) {
#pragma unroll
                            for (int i = 0; i < WM; ++i) dst[u][i] = wbase[i][tl * tap_stride + (size_t)ksl * 64];
                        }
                        ksl += STEP;
                        while (ksl >= kst) { ksl -= kst; ++tl; }
                    }
                };
                fetch(wa);
                while (tc < ntaps) {
                    fetch(wn);                               // next group's weights: PF loads in flight
#pragma unroll
                    for (int u = 0; u < PF; ++u) {
                        if (tc < ntaps) {
                            const int sh = sbase + tc * sstep - smin;
                            AF bf[WN];
#pragma unroll
                            for (int k = 0; k < WN; ++k) {
                                const int r = wrow + k * 32 + j + sh;
                                bf[k] = __builtin_bit_cast(AF, *reinterpret_cast<const u32x4*>(lds + (size_t)r * RS + (ksc * KS + hf * E) * sizeof(CT)));
                            }
#pragma unroll
                            for (int i = 0; i < WM; ++i) {
                                const AF af = __builtin_bit_cast(AF, wa[u][i]);
#pragma unroll
                                for (int k = 0; k < WN; ++k) Mma<CT>::run(acc[i][k], af, bf[k]);
                            }
                        }
                        ksc += STEP;
                        while (ksc >= kst) { ksc -= kst; ++tc; }
                    }
#pragma unroll
                    for (int u = 0; u < PF; ++u)
#pragma unroll
                        for (int i = 0; i < WM; ++i) wa[u][i] = wn[u][i];
                }
            }
        }
    }
    if constexpr (SPLITK) {
        // merge the 4 waves' partial tiles in wave order through the (now free) staging LDS
        float* red = reinterpret_cast<float*>(lds);
        __syncthreads();
        if (wid > 0) {
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int k = 0; k < WN; ++k)
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        red[(((wid - 1) * WM * WN + i * WN + k) * 16 + q) * 64 + lane] = acc[i][k][q];
        }
        __syncthreads();
        if (wid > 0) return;
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int k = 0; k < WN; ++k)
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        acc[i][k][q] += red[((w * WM * WN + i * WN + k) * 16 + q) * 64 + lane];
    }
    if (!wave_live) return;

    // epilogue.  Everything the epilogue reads (bias, conditioning, residual, previous output, mask)
    // is fetched for a whole 32-row tile -- and for the NEXT tile before this one is stored, because
    // the output may alias the residual as far as the compiler knows and it will not hoist a load
    // over a store -- so the tile pays one memory latency, not one per 4-channel group.
    if (a.dbg & 4) return;
    OT* Y = reinterpret_cast<OT*>(Yv);
    const OT* R = reinterpret_cast<const OT*>(Rv);
    const float* A = reinterpret_cast<const float*>(a.add);
    using RawO = typename Out4<OT>::Raw;
    float bv[WM][16];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m = (mt0 + i) * 32 + 16 * hf + q;
            bv[i][q] = Bv ? Bv[min(m, a.cout - 1)] : 0.f;
            bv[i][q] = m < a.cout ? bv[i][q] : 0.f;
        }
    struct TileIn {
        RawO res[WM][4];
        float mk;
    };
    auto tile_load = [&](int k, TileIn& in) {
        const int n = n0 + k * 32 + j;
        const size_t orow = (size_t)min(n, a.n_rows - 1) * a.omul + phase;
        in.mk = a.mask ? a.mask[orow] : 1.0f;
        if (R) {
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int m = (mt0 + i) * 32 + 16 * hf + 4 * g;
                    const bool full = mt0 + i < a.mtiles && m + 4 <= a.cout;
                    in.res[i][g] = Out4<OT>::raw(full ? R + orow * a.ld_res + m : R);
                }
        }
    };
    auto tile_store = [&](int k, const TileIn& in) {
        const int n = n0 + k * 32 + j;
        if (n >= a.n_rows) return;
        const size_t orow = (size_t)n * a.omul + phase;
        const float mk = in.mk;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int mt = mt0 + i;
            if (mt >= a.mtiles) continue;
            // conditioning term and previous output (flow / conv_pre only): all four groups' loads first
            f32x4 addv[4];
            RawO oldv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m = mt * 32 + 16 * hf + 4 * g;
                const bool full = m + 4 <= a.cout;
                if (A) addv[g] = *reinterpret_cast<const f32x4*>(full ? A + (a.ld_add ? (a.add_index ? (size_t)a.add_index[orow] : (size_t)orow) * (size_t)a.ld_add : (size_t)0) + m : A);
                if (a.accumulate) oldv[g] = Out4<OT>::raw(full ? Y + orow * a.ldy + m : Y);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m = mt * 32 + 16 * hf + 4 * g;
                if (m >= a.cout) continue;
                const bool full = (m + 4 <= a.cout);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][k][4 * g + e] + bv[i][4 * g + e];
                OT* yp = Y + orow * a.ldy + m;
                if (full) {
                    if (A) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += addv[g][e];
                    }
                    if (a.act == ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    } else if (a.act == ACT_TANH) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
                    }
                    if (R) {
                        float rv[4];
                        Out4<OT>::widen(in.res[i][g], rv);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += rv[e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] * mk * a.scale;
                    if (a.accumulate) {
                        float ov[4];
                        Out4<OT>::widen(oldv[g], ov);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += ov[e];
                    }
                    Out4<OT>::store(yp, v);
                } else {
                    // ragged channel tail (cout not a multiple of 4): element by element
                    const float* ap = A ? A + (a.ld_add ? (a.add_index ? (size_t)a.add_index[orow] : (size_t)orow) * (size_t)a.ld_add : (size_t)0) + m : nullptr;
                    for (int e = 0; e < 4 && m + e < a.cout; ++e) {
                        float x = v[e];
                        if (ap) x += ap[e];
                        if (a.act == ACT_RELU) x = fmaxf(x, 0.f);
                        else if (a.act == ACT_TANH) x = tanhf(x);
                        if (R) x += to_f32<OT>(R[orow * a.ld_res + m + e]);
                        x = x * mk * a.scale;
                        if (a.accumulate) x += to_f32<OT>(yp[e]);
                        yp[e] = from_f32<OT>(x);
                    }
                }
            }
        }
    };
    TileIn tin[2];
    tile_load(0, tin[0]);
#pragma unroll
    for (int k = 0; k < WN; ++k) {
        if (k + 1 < WN) tile_load(k + 1, tin[(k + 1) & 1]);
        tile_store(k, tin[k & 1]);
    }
}

// Pack torch-layout fp32 weights into fragment order.
//   src element (m, c, kk) at src[m*sm + c*sc + kk*sk];  tap index kk(phase r, tap t):
//   conv: kk = t ; transposed conv (u > 0): kk = (r + pad) % u + t*u, zero panel when kk >= k.
template <typename CT>
__global__ void tapgemm_pack_kernel(const float* __restrict__ src, CT* __restrict__ dst, int cout, int cin,
                                    int k, int64_t sm, int64_t sc, int64_t sk, int nphase, int ntaps, int u,
                                    int pad, int mtiles) {
    constexpr int KS = MfmaK<CT>::KS;
    constexpr int E = KS / 2;
    const int ksteps = cin / KS;
    const size_t total = (size_t)nphase * ntaps * mtiles * ksteps * 64 * E;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        size_t r = idx;
        const int e = r % E; r /= E;
        const int lane = r % 64; r /= 64;
        const int ks = r % ksteps; r /= ksteps;
        const int mt = r % mtiles; r /= mtiles;
        const int t = r % ntaps; r /= ntaps;
        const int ph = (int)r;
        const int mr = lane & 31;  // MFMA row; rows are permuted so that a lane's 16 D registers are 16 consecutive channels
        const int m = mt * 32 + 16 * ((mr >> 2) & 1) + (mr & 3) + 4 * (mr >> 3);
        const int c = ks * KS + (lane >> 5) * E + e;
        const int kk = (u > 0) ? ((ph + pad) % u + t * u) : t;
        float v = 0.f;
        if (m < cout && kk < k) v = src[m * sm + c * sc + kk * sk];
        dst[idx] = from_f32<CT>(v);
    }
}

}  // namespace gsv
