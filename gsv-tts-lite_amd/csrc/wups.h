// wups: the Generator's upsampling layers (ConvTranspose1d(C, C/2, k, stride u, padding (k-u)/2) on leaky-ReLU'd
// input, SoVITS/models.py:122-124 of the reference) as a persistent weights-in-registers MFMA kernel, the
// transposed sibling of wconv.h.
//
// A transposed conv of stride u is u phases r, each a conv with ceil(k/u) taps that writes output rows n*u + r
// (tapgemm.h packs the weights as [phase][tap][mtile][kstep][lane]; tap t of phase r reads input row
// n + (r + pad)/u - t).  The generic kernel runs every phase as its own grid slice, so each (row tile, phase) block
// stages its input rows and streams its weights for only ntaps * C/16 MFMA steps -- 77-116 TF/s on these shapes.
// Here a block owns a GROUP of PG phases and MS 32-channel output slices for its whole life, keeps those weights in
// registers (PG * ntaps * C/16 fragments per wave), walks row tiles, stages each input tile once for all its
// phases and writes the PG interleaved output rows.  Blocks of different phase / slice groups walk the same tiles.
#pragma once
#include "wconv.h"

namespace gsv {

struct WUpsArgs {
    const bf16_t* X;      // [n_in][ldx]
    const uint4* W;       // tapgemm fragments [phase][tap][mtile][kstep][lane]
    const float* bias;    // [cout] or null
    bf16_t* Y;            // [n_in * u][ldy]
    int ldx, ldy, n_in;
    int u, tpad;          // stride (= number of phases), (k - u) / 2
    int mtiles, cout;     // ceil(cout / 32), real output channels
    int cvalid;           // channels written per row (cout rounded up to the row's 16-channel padding)
    float in_slope;       // leaky-ReLU on the input
    int nwalk;            // row-tile walkers; grid = nwalk * (u / PG) * ceil(mtiles / MS)
    bf16_t* Ya;           // null, or lrelu(Y, act_slope) of the rounded output next to Y: what the resblocks' first convs contract (wdma.h)
    float act_slope;
};

template <int CIN, int MS, int BN, int NTAPS, int PG>
__global__ __launch_bounds__(256, 1) void wups_kernel(WUpsArgs a) {
    constexpr int KSTEPS = CIN / 16;
    constexpr int RG = 4 / MS;
    constexpr int WN = BN / 32 / RG;
    constexpr int XRS = CIN * 2 + 16;
    constexpr int HALO = NTAPS + 1;               // input rows a tile needs beyond BN, all phases of the group
    constexpr int XROWS = BN + HALO;
    constexpr int XBYTES = XROWS * XRS;
    constexpr int VPR = CIN / 8;
    constexpr int RPP = 256 / VPR > 0 ? 256 / VPR : 1;
    constexpr int NVX = (XROWS * VPR + 255) / 256;   // staging vectors per thread (flat over rows x vectors)
    constexpr int RORS = 32 * 2 + 16;
    constexpr int NVR = WN * 32 * 4 / 64;
    constexpr int ROBYTES = WN * 32 * RORS;
    constexpr int NF = PG * NTAPS * KSTEPS;       // weight fragments per wave
    static_assert(NF * 4 <= 400, "a wave's weights must fit its registers");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = lane & 31, hf = lane >> 5;
    const int ms = wid % MS, rg = wid / MS;
    const int npg = a.u / PG, nmg = (a.mtiles + MS - 1) / MS;
    const int grp = blockIdx.x % (npg * nmg), walker = blockIdx.x / (npg * nmg);
    const int pg = grp % npg, mg = grp / npg;
    const int gs = mg * MS + ms;
    const bool live = gs < a.mtiles;
    const int wrow = rg * WN * 32;
    const int ntiles = (a.n_in + BN - 1) / BN;
    if (walker >= ntiles) return;
    unsigned char* xbuf0 = lds;
    unsigned char* xbuf1 = lds + XBYTES;
    unsigned char* ro = lds + 2 * XBYTES + wid * ROBYTES;
    float* bl = reinterpret_cast<float*>(lds + 2 * XBYTES + 4 * ROBYTES);

    // shifts of the group's phases: tap t of phase r reads input row n + (r + tpad) / u - t
    int sb[PG];
    int smin = 1 << 30;
#pragma unroll
    for (int p = 0; p < PG; ++p) {
        sb[p] = (pg * PG + p + a.tpad) / a.u;
        smin = min(smin, sb[p] - (NTAPS - 1));
    }

    // the wave's weights are requested BEHIND the first tile's rows and the bias (as in wconv.h: in front of them, the bias store
    // to LDS drained every weight load before a single row was requested)
    u32x4 w[NF];
    const int gsw = live ? gs : 0;
    auto load_weights = [&]() {
#pragma unroll
        for (int p = 0; p < PG; ++p)
#pragma unroll
            for (int t = 0; t < NTAPS; ++t)
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks)
                    w[(p * NTAPS + t) * KSTEPS + ks] =
                        __builtin_bit_cast(u32x4, a.W[((((size_t)(pg * PG + p) * NTAPS + t) * a.mtiles + gsw) * KSTEPS + ks) * 64 + lane]);
    };
    float bias_raw = 0.f;                       // clamped address now, masked at its use
    if (a.bias != nullptr) bias_raw = a.bias[min(mg * MS * 32 + (tid < MS * 32 ? tid : 0), a.cout - 1)];

    u32x4 xraw[NVX];
    auto issue_x = [&](int tile) {
        const int gbase = tile * BN + smin;
#pragma unroll
        for (int v = 0; v < NVX; ++v) {
            const int idx = tid + v * 256, r = idx / VPR, cv = idx % VPR;
            const int grow = gbase + r;
            const bool ok = r < XROWS && grow >= 0 && grow < a.n_in;
            xraw[v] = *reinterpret_cast<const u32x4*>(a.X + (size_t)(ok ? grow : 0) * a.ldx + cv * 8);
        }
    };
    auto commit_x = [&](int tile, unsigned char* xb) {
        const int gbase = tile * BN + smin;
#pragma unroll
        for (int v = 0; v < NVX; ++v) {
            const int idx = tid + v * 256, r = idx / VPR, cv = idx % VPR;
            const int grow = gbase + r;
            const bool ok = grow >= 0 && grow < a.n_in;
            if (r < XROWS) *reinterpret_cast<u32x4*>(xb + (size_t)r * XRS + cv * 16) = Stage16<bf16_t, bf16_t>::finish(xraw[v], ok, a.in_slope);
        }
    };
    auto piece_ok = [&](int pc) { return gs * 32 + pc * 8 < a.cvalid; };

    issue_x(walker);
    load_weights();
    if (tid < MS * 32) bl[tid] = mg * MS * 32 + tid < a.cout ? bias_raw : 0.f;
    commit_x(walker, xbuf0);
    __syncthreads();
    int cur = 0;
    for (int tile = walker; tile < ntiles; tile += a.nwalk) {
        const int tn = tile + a.nwalk;
        const bool has_next = tn < ntiles;
        if (has_next) issue_x(tn);
        if (live) {
            const unsigned char* xb = cur ? xbuf1 : xbuf0;
            f32x16 acc[PG][WN];
#pragma unroll
            for (int p = 0; p < PG; ++p)
#pragma unroll
                for (int k = 0; k < WN; ++k)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[p][k][q] = 0.f;
            {
                constexpr int DEPTH = 3;
                u32x4 bf[DEPTH + 1][WN];
                // LDS row of (phase p, tap t) for output row j of the wave: wrow + j + sb[p] - t - smin
                auto ldb = [&](int it, u32x4 (&dst)[WN]) {
                    const int p = it / (NTAPS * KSTEPS), t = (it / KSTEPS) % NTAPS, ks = it % KSTEPS;
                    const unsigned tb = (unsigned)(wrow + j + sb[p] - t - smin) * XRS + hf * 16 + ks * 32;
#pragma unroll
                    for (int k = 0; k < WN; ++k) dst[k] = *reinterpret_cast<const u32x4*>(xb + tb + k * 32 * XRS);
                };
#pragma unroll
                for (int it = 0; it < DEPTH && it < NF; ++it) ldb(it, bf[it % (DEPTH + 1)]);
                __builtin_amdgcn_sched_group_barrier(0x100, (DEPTH < NF ? DEPTH : NF) * WN, 0);
#pragma unroll
                for (int it = 0; it < NF; ++it) {
                    if (it + DEPTH < NF) ldb(it + DEPTH, bf[(it + DEPTH) % (DEPTH + 1)]);
#pragma unroll
                    for (int k = 0; k < WN; ++k) Mma<bf16_t>::run(acc[it / (NTAPS * KSTEPS)][k], w[it], bf[it % (DEPTH + 1)][k]);
                    __builtin_amdgcn_sched_group_barrier(0x008, WN, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, WN, 0);
                }
            }
            // ---- epilogue: phase p of the group writes output rows n * u + pg*PG + p, wave-private patch
#pragma unroll
            for (int p = 0; p < PG; ++p) {
#pragma unroll
                for (int k = 0; k < WN; ++k) {
                    unsigned char* pp = ro + (k * 32 + j) * RORS + hf * 32;
                    float v[16];
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + ms * 32 + 16 * hf + 4 * q4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * q4 + e] = acc[p][k][4 * q4 + e] + b4[e];
                    }
                    u32x4 oa, ob;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        oa[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                        ob[e] = pack_bf16x2(v[8 + 2 * e], v[8 + 2 * e + 1]);
                    }
                    *reinterpret_cast<u32x4*>(pp) = oa;
                    *reinterpret_cast<u32x4*>(pp + 16) = ob;
                }
#pragma unroll
                for (int q = 0; q < NVR; ++q) {
                    const int idx = q * 64 + lane, row = idx / 4, pc = idx % 4;
                    const u32x4 o = *reinterpret_cast<const u32x4*>(ro + row * RORS + pc * 16);
                    const int n = tile * BN + wrow + row;
                    if (n < a.n_in && piece_ok(pc)) {
                        const size_t off = ((size_t)n * a.u + pg * PG + p) * a.ldy + gs * 32 + pc * 8;
                        *reinterpret_cast<u32x4*>(a.Y + off) = o;
                        if (a.Ya) {
                            u32x4 oa;
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                oa[e] = pack_bf16x2(lrelu(__uint_as_float(o[e] << 16), a.act_slope), lrelu(__uint_as_float(o[e] & 0xffff0000u), a.act_slope));
                            *reinterpret_cast<u32x4*>(a.Ya + off) = oa;
                        }
                    }
                }
            }
        }
        if (has_next) commit_x(tn, cur ? xbuf0 : xbuf1);
        __syncthreads();
        cur ^= 1;
    }
}

template <int CIN, int MS, int BN, int NTAPS, int PG>
constexpr size_t wups_lds_bytes() {
    return (size_t)2 * (BN + NTAPS + 1) * (CIN * 2 + 16) + (size_t)4 * (BN / (4 / MS)) * (32 * 2 + 16) + MS * 32 * sizeof(float);
}

}  // namespace gsv
