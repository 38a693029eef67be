// Host-side launchers of the SoVITS Generator's kernel families -- cgemm.h, wconv.h, wdma.h, wups.h -- for gsv_voc.hip ONLY (round 6: they sat
// in abi_common.h, and gsv_abi.hip, which launches none of them, carried a second copy of 47 kernels).
#pragma once
#include "abi_common.h"
#include "wconv.h"
#include "wups.h"
#include "cgemm.h"
#include "wdma.h"
#include "voc_kernels.h"

namespace {

// The weights-in-registers path for the Generator's resblock convs (wconv.h).  Returns 1 when the
// launch does not fit it (caller falls back to tapgemm): returns -1 then, 0 on success, > 0 = GSV_ERR_*.
// wide resblock convs on the LDS-tiled GEMM (cgemm.h): 384 and 192 channels always, 256 channels from 16k rows on (below that
// the weights-in-registers kernel's shorter block wins: measured 33 vs 39 us per launch at 5 000 rows, 267 vs 207 at 50 000)
// K-split launches of cgemm.h: at most this many blocks before the split (the grid underfills 256 CUs), hence at most 2 x 85 split tiles per launch
constexpr int kCgSplitMaxBlocks = 256;
constexpr int kCgSplitMaxTiles = 2 * (kCgSplitMaxBlocks / 3);
constexpr size_t kCgPartBytes = (size_t)kCgSplitMaxTiles * 2 * 128 * 192 * sizeof(float);   // the largest tile of the shapes that split: 128 rows x 192 channels

template <typename AT>
int run_cgemm(const Branch* brs, int ld, int n_rows, float in_slope, float out_slope, hipStream_t st, float* part = nullptr, size_t part_bytes = 0, int* flag = nullptr) {
    (void)brs; (void)ld; (void)n_rows; (void)in_slope; (void)out_slope; (void)st; (void)part; (void)part_bytes; (void)flag;
    return -1;
}
template <>
inline int run_cgemm<bf16_t>(const Branch* brs, int ld, int n_rows, float in_slope, float out_slope, hipStream_t st, float* part, size_t part_bytes, int* flag) {
    static const bool off = getenv("GSV_NO_CGEMM") != nullptr;
    const int C = brs[0].pc->cout;
    if (off || !(C == 384 || C == 192 || (C == 256 && n_rows >= 16384)) || ld != C) return -1;
    int order[3] = {0, 1, 2};
    for (int i = 0; i < 3; ++i) {
        const PackedConv& q = *brs[i].pc;
        if (!q.cg || q.cin != C || q.cout != C || q.u != 0 || q.k > 11 || (q.k - 1) * q.dil > 50 || q.pad != (q.k - 1) / 2 * q.dil) return -1;
    }
    if ((brs[0].res == nullptr) != (brs[1].res == nullptr) || (brs[0].res == nullptr) != (brs[2].res == nullptr)) return -1;
    std::sort(order, order + 3, [&](int x, int y) { return brs[x].pc->k > brs[y].pc->k; });  // heaviest branch dispatches first
    const Branch &b0 = brs[order[0]], &b1 = brs[order[1]], &b2 = brs[order[2]];
    CGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.X0 = (const bf16_t*)b0.X; a.X1 = (const bf16_t*)b1.X; a.X2 = (const bf16_t*)b2.X;
    a.W0 = (const uint4*)b0.pc->cg; a.W1 = (const uint4*)b1.pc->cg; a.W2 = (const uint4*)b2.pc->cg;
    a.b0 = b0.pc->bias; a.b1 = b1.pc->bias; a.b2 = b2.pc->bias;
    a.R0 = (const bf16_t*)b0.res; a.R1 = (const bf16_t*)b1.res; a.R2 = (const bf16_t*)b2.res;
    a.Y0 = (bf16_t*)b0.Y; a.Y1 = (bf16_t*)b1.Y; a.Y2 = (bf16_t*)b2.Y;
    a.k0 = b0.pc->k; a.k1 = b1.pc->k; a.k2 = b2.pc->k;
    a.d0 = b0.pc->dil; a.d1 = b1.pc->dil; a.d2 = b2.pc->dil;
    a.ld = ld; a.n_rows = n_rows; a.in_slope = in_slope; a.out_slope = out_slope;
    auto launch = [&](auto kern, size_t lds, int bm, int tn, int nt) -> int {
        const int tiles = cdiv(n_rows, bm) * tn;
        // Few row tiles (10 s of audio at 384 channels: 80 tiles per branch = 240 blocks of 66 / 42 / 18 iterations for 256 CUs; a streaming chunk:
        // 24 blocks): the heavy branches' tiles are K-split over consecutive blocks (cgemm.h `ns`): 72 -> 59 us per launch at 5 000 rows, 66 -> 43
        // at 500 (tools/cg_bench, profiles/r06_cgemm_ksplit.txt).  The scratch holds two partial tiles per split tile.
        const int nch = C / 64;
        int ns[3] = {1, 1, 1};
        static const bool nosplit = getenv("GSV_CGEMM_NO_SPLIT") != nullptr;
        if (!nosplit && part && flag && 3 * tiles <= kCgSplitMaxBlocks) {
            const int ks[3] = {a.k0, a.k1, a.k2};
            for (int i = 0; i < 3; ++i) {
                if (ks[i] >= 9) ns[i] = nch % 3 == 0 ? 3 : (nch % 2 == 0 ? 2 : 1);
                else if (ks[i] >= 5) ns[i] = nch % 2 == 0 ? 2 : 1;
            }
            const size_t need = (size_t)((ns[0] > 1 ? tiles : 0) + (ns[1] > 1 ? tiles : 0) + (ns[2] > 1 ? tiles : 0)) * 2 * bm * (C / tn) * sizeof(float);
            if (need > part_bytes) ns[0] = ns[1] = ns[2] = 1;
        }
        a.ns0 = ns[0]; a.ns1 = ns[1]; a.ns2 = ns[2];
        a.nb0 = tiles * ns[0]; a.nb1 = tiles * ns[1];
        a.part = part; a.flag = flag;
        HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(tiles * (ns[0] + ns[1] + ns[2])), dim3(nt), lds, st, a);
        HIPCHK(hipGetLastError());
        return GSV_OK;
    };
    if (C == 384) return launch(cgemm_kernel<384, 192, 128>, CgShape<384, 192, 128>::LDS, 128, CgShape<384, 192, 128>::TN, CgShape<384, 192, 128>::NT);
    // 256 channels: 256-row tiles (4 waves of 128 rows x 64 channels): half the weight-tile traffic per row of the 128-row shape
    if (C == 256) return launch(cgemm_kernel<256, 128, 256, 4>, CgShape<256, 128, 256, 4>::LDS, 256, CgShape<256, 128, 256, 4>::TN, CgShape<256, 128, 256, 4>::NT);
    return launch(cgemm_kernel<192, 192, 128>, CgShape<192, 192, 128>::LDS, 128, CgShape<192, 192, 128>::TN, CgShape<192, 192, 128>::NT);
}
// the plane-order copy of a wide resblock conv's weights (torch layout [C][C][k] fp32 in)
inline int pack_cgemm(PackedConv& pc, const float* w, int C, int k, hipStream_t st) {
    if (!(C == 384 || C == 256 || C == 192)) return GSV_OK;
    if (!pc.cg) HIPCHK(gsv_dev_malloc(&pc.cg, sizeof(bf16_t) * (size_t)C * C * k));
    if (C == 384) hipLaunchKernelGGL(cgemm_pack_kernel<384>, dim3(1024), dim3(256), 0, st, w, (bf16_t*)pc.cg, k);
    else if (C == 256) hipLaunchKernelGGL(cgemm_pack_kernel<256>, dim3(1024), dim3(256), 0, st, w, (bf16_t*)pc.cg, k);
    else hipLaunchKernelGGL(cgemm_pack_kernel<192>, dim3(1024), dim3(256), 0, st, w, (bf16_t*)pc.cg, k);
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

// 192 and 256 channels on wconv.h (K split over two waves, output slices over 3 / 4 blocks) are A/B kernels: the library ships cgemm.h for
// 192 and wdma.h / cgemm.h for 256, and wconv_kernel<256, ...> spills (144 B).  -DGSV_AB_KERNELS on gsv_voc.hip builds them in (tools/README.md).
inline bool wconv_channels(int C) {
#ifdef GSV_AB_KERNELS
    if (C == 192 || C == 256) return true;
#endif
    return C == 16 || C == 24 || C == 32 || C == 48 || C == 64 || C == 96 || C == 128;
}
template <typename AT>
int run_wconv(const Branch* brs, int ld, int n_rows, float in_slope, float out_slope, hipStream_t st) {
    (void)brs; (void)ld; (void)n_rows; (void)in_slope; (void)out_slope; (void)st;
    return -1;
}
template <>
int run_wconv<bf16_t>(const Branch* brs, int ld, int n_rows, float in_slope, float out_slope, hipStream_t st) {
    const int C = brs[0].pc->cout;
    if (!wconv_channels(C)) return -1;
    const int Ck = C == 24 ? 32 : C;   // 24 channels live in rows of 32 (zero pad channels, zero weight rows): the 32 kernel
    int order[3] = {0, 1, 2};
    for (int i = 0; i < 3; ++i) {
        const PackedConv& q = *brs[i].pc;
        if (q.cin != Ck || q.cout != C || q.u != 0 || (q.k != 3 && q.k != 7 && q.k != 11) || q.dil < 1 || q.dil > 5 ||
            q.pad != (q.k - 1) / 2 * q.dil || ld < Ck)
            return -1;
    }
    std::sort(order, order + 3, [&](int x, int y) { return brs[x].pc->k > brs[y].pc->k; });  // heaviest branch dispatches first
    // blocks are dealt in proportion to taps + a per-tile overhead (staging, epilogue) in tap units; both the
    // overhead and the block count per shape are measured (tools/tg_bench.hip)
    const int msp = C == 256 ? 4 : (C == 192 ? 3 : 1);   // blocks that share a row-tile walk (output slices split between them)
    int nblk = C >= 64 ? 256 : (C >= 32 ? 512 : (C == 24 ? 512 : 768));
    const double ovh = C >= 96 ? 8.0 : (C == 64 ? 14.0 : (C == 48 ? 30.0 : 50.0));
    double tot = 0;
    for (int i = 0; i < 3; ++i) tot += brs[i].pc->k + ovh;
    int nb[3], used = 0;
    for (int i = 0; i < 3; ++i) { nb[i] = std::max(msp, (int)(nblk * (brs[order[i]].pc->k + ovh) / tot) / msp * msp); used += nb[i]; }
    nb[0] += (nblk - used) / msp * msp;
    nblk = nb[0] + nb[1] + nb[2];
    WConvArgs a;
    memset(&a, 0, sizeof(a));
    const Branch &b0 = brs[order[0]], &b1 = brs[order[1]], &b2 = brs[order[2]];
    a.X0 = (const bf16_t*)b0.X; a.X1 = (const bf16_t*)b1.X; a.X2 = (const bf16_t*)b2.X;
    a.W0 = (const uint4*)b0.pc->w; a.W1 = (const uint4*)b1.pc->w; a.W2 = (const uint4*)b2.pc->w;
    a.b0 = b0.pc->bias; a.b1 = b1.pc->bias; a.b2 = b2.pc->bias;
    a.R0 = (const bf16_t*)b0.res; a.R1 = (const bf16_t*)b1.res; a.R2 = (const bf16_t*)b2.res;
    a.Y0 = (bf16_t*)b0.Y; a.Y1 = (bf16_t*)b1.Y; a.Y2 = (bf16_t*)b2.Y;
    a.k0 = b0.pc->k; a.k1 = b1.pc->k; a.k2 = b2.pc->k;
    a.d0 = b0.pc->dil; a.d1 = b1.pc->dil; a.d2 = b2.pc->dil;
    a.nb0 = nb[0]; a.nb1 = nb[1]; a.nb2 = nb[2];
    a.ld = ld; a.n_rows = n_rows; a.in_slope = in_slope; a.out_slope = out_slope; a.cout = C;
    if ((b0.res == nullptr) != (b1.res == nullptr) || (b0.res == nullptr) != (b2.res == nullptr)) return -1;
    auto launch = [&](auto kern, size_t lds) -> int {
        HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, st, a);
        HIPCHK(hipGetLastError());
        return GSV_OK;
    };
#ifdef GSV_AB_KERNELS
    if (C == 256) return launch(wconv_kernel<256, 2, 64, 2, 4>, wconv_lds_bytes<256, 2, 64, 2, 4>());   // K split in the block, slices over 4 blocks
    if (C == 192) return launch(wconv_kernel<192, 2, 64, 2, 3>, wconv_lds_bytes<192, 2, 64, 2, 3>());
#endif
    if (C == 128) return launch(wconv_kernel<128, 4, 64>, wconv_lds_bytes<128, 4, 64>());
    if (C == 96) return launch(wconv_kernel<96, 4, 64>, wconv_lds_bytes<96, 4, 64>());     // 3 slices + a staging-only wave
    if (C == 64) return launch(wconv_kernel<64, 2, 128>, wconv_lds_bytes<64, 2, 128>());
    if (C == 48) return launch(wconv_kernel<48, 2, 64>, wconv_lds_bytes<48, 2, 64>());
    if (Ck == 32) return launch(wconv_kernel<32, 1, 256>, wconv_lds_bytes<32, 1, 256>());
    return launch(wconv_kernel<16, 1, 256>, wconv_lds_bytes<16, 1, 256>());
}


// The resblock convs at 64 / 128 / 256 channels with rows and residual by LDS-DMA (wdma.h).  The inputs `brs[i].X` are the ACTIVATED
// copies their producers wrote; `act[i]` (null or a buffer) receives lrelu(Y_i, act_slope).  -1 = shape not covered, 0 = launched.
inline bool wdma_shape(int C, int ld, int n_rows) {
    static const bool off = getenv("GSV_NO_WDMA") != nullptr;     // A/B switch: the 64 / 128-channel convs on wconv.h (tests/test_hip_vocoder.py)
    if (ld != C) return false;
#ifdef GSV_AB_KERNELS
    if (off) return false;
#else
    if (off && C != 256) return false;                            // 256 channels have no wconv.h kernel in the shipped library
#endif
    return C == 64 || C == 128 || (C == 256 && n_rows < 16384);
}
inline int run_wdma(const Branch* brs, void* const* act, int ld, int n_rows, float out_slope, float act_slope, const void* zeros, void* sink, hipStream_t st) {
    const int C = brs[0].pc->cout;
    if (!wdma_shape(C, ld, n_rows) || !zeros || !sink) return -1;
    int order[3] = {0, 1, 2};
    for (int i = 0; i < 3; ++i) {
        const PackedConv& q = *brs[i].pc;
        if (q.cin != C || q.cout != C || q.u != 0 || (q.k != 3 && q.k != 7 && q.k != 11) || q.dil < 1 || q.dil > 5 || q.pad != (q.k - 1) / 2 * q.dil || !q.bias)
            return -1;
    }
    if ((brs[0].res == nullptr) != (brs[1].res == nullptr) || (brs[0].res == nullptr) != (brs[2].res == nullptr)) return -1;
    if ((act[0] == nullptr) != (act[1] == nullptr) || (act[0] == nullptr) != (act[2] == nullptr)) return -1;
    std::sort(order, order + 3, [&](int x, int y) { return brs[x].pc->k > brs[y].pc->k; });  // heaviest branch dispatches first
    const int msp = C == 256 ? 4 : 1;
    int nblk = 256;
    const double ovh = C == 64 ? 14.0 : 8.0;     // per-tile overhead in tap units (tools/tg_bench.hip)
    double tot = 0;
    for (int i = 0; i < 3; ++i) tot += brs[i].pc->k + ovh;
    int nb[3], used = 0;
    for (int i = 0; i < 3; ++i) { nb[i] = std::max(msp, (int)(nblk * (brs[order[i]].pc->k + ovh) / tot) / msp * msp); used += nb[i]; }
    nb[0] += (nblk - used) / msp * msp;
    nblk = nb[0] + nb[1] + nb[2];
    WDmaArgs a;
    memset(&a, 0, sizeof(a));
    const Branch &b0 = brs[order[0]], &b1 = brs[order[1]], &b2 = brs[order[2]];
    a.X0 = (const bf16_t*)b0.X; a.X1 = (const bf16_t*)b1.X; a.X2 = (const bf16_t*)b2.X;
    a.W0 = (const uint4*)b0.pc->w; a.W1 = (const uint4*)b1.pc->w; a.W2 = (const uint4*)b2.pc->w;
    a.b0 = b0.pc->bias; a.b1 = b1.pc->bias; a.b2 = b2.pc->bias;
    a.R0 = (const bf16_t*)b0.res; a.R1 = (const bf16_t*)b1.res; a.R2 = (const bf16_t*)b2.res;
    a.Y0 = (bf16_t*)b0.Y; a.Y1 = (bf16_t*)b1.Y; a.Y2 = (bf16_t*)b2.Y;
    a.A0 = (bf16_t*)act[order[0]]; a.A1 = (bf16_t*)act[order[1]]; a.A2 = (bf16_t*)act[order[2]];
    a.k0 = b0.pc->k; a.k1 = b1.pc->k; a.k2 = b2.pc->k;
    a.d0 = b0.pc->dil; a.d1 = b1.pc->dil; a.d2 = b2.pc->dil;
    a.nb0 = nb[0]; a.nb1 = nb[1]; a.nb2 = nb[2];
    a.ld = ld; a.n_rows = n_rows; a.out_slope = out_slope; a.act_slope = act_slope; a.zeros = zeros; a.sink = sink;
    auto launch = [&](auto kern, size_t lds) -> int {
        HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, st, a);
        HIPCHK(hipGetLastError());
        return GSV_OK;
    };
    if (C == 256) return launch(wdma_kernel<256, 2, 64, 2, 4>, wdma_lds_bytes<256, 2, 64, 2, 4>());
    if (C == 128) return launch(wdma_kernel<128, 4, 64>, wdma_lds_bytes<128, 4, 64>());
    return launch(wdma_kernel<64, 2, 128>, wdma_lds_bytes<64, 2, 128>());
}

// Upsampling layer (transposed conv) on the weights-in-registers kernel (wups.h); -1 = shape not covered (caller
// falls back to tapgemm), 0 = launched, > 0 = GSV_ERR_*.
template <typename AT>
int run_wups(const PackedConv& pc, const void* X, int ldx, int n_in, void* Y, int ldy, float in_slope, hipStream_t st, void* Ya = nullptr, float act_slope = 1.f) {
    (void)pc; (void)X; (void)ldx; (void)n_in; (void)Y; (void)ldy; (void)in_slope; (void)st; (void)Ya; (void)act_slope;
    return -1;
}
template <>
int run_wups<bf16_t>(const PackedConv& pc, const void* X, int ldx, int n_in, void* Y, int ldy, float in_slope, hipStream_t st, void* Ya, float act_slope) {
    if (pc.u < 1 || getenv("GSV_NO_WUPS")) return -1;
    WUpsArgs a;
    a.X = (const bf16_t*)X; a.W = (const uint4*)pc.w; a.bias = pc.bias; a.Y = (bf16_t*)Y; a.Ya = (bf16_t*)Ya; a.act_slope = act_slope;
    a.ldx = ldx; a.ldy = ldy; a.n_in = n_in; a.u = pc.u; a.tpad = pc.pad; a.mtiles = pc.mtiles; a.cout = pc.cout;
    a.cvalid = std::min(ldy, (pc.cout + 15) / 16 * 16); a.in_slope = in_slope;
    auto launch = [&](auto kern, size_t lds, int ms, int bn, int pg, int max_blocks) -> int {
        if (pc.u % pg != 0) return -1;
        static const int maxb_env = getenv("GSV_WUPS_MAXB") ? atoi(getenv("GSV_WUPS_MAXB")) : 0;   // tuning aid
        if (maxb_env > 0) max_blocks = maxb_env;
        const int groups = (pc.u / pg) * cdiv(pc.mtiles, ms);
        a.nwalk = std::max(1, std::min(cdiv(n_in, bn), max_blocks / groups));
        HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(a.nwalk * groups), dim3(256), lds, st, a);
        HIPCHK(hipGetLastError());
        return GSV_OK;
    };
#define GSV_WUPS(CIN, MS, BN, NT, PG, MAXB)                                                     \
    if (pc.cin == CIN && pc.ntaps == NT && ldx >= CIN)                                           \
        return launch(wups_kernel<CIN, MS, BN, NT, PG>, wups_lds_bytes<CIN, MS, BN, NT, PG>(), MS, BN, PG, MAXB);
    GSV_WUPS(512, 4, 32, 2, 1, 256)     // <= 256 blocks: one block per CU (512 registers), a 257th block is a second round (13.9 -> 11.5 us at 10 s)
    GSV_WUPS(256, 4, 64, 2, 2, 256)
    GSV_WUPS(128, 2, 128, 4, 2, 256)
    GSV_WUPS(64, 1, 256, 1, 2, 256)
    GSV_WUPS(32, 1, 256, 1, 2, 768)
    // 768 -> 384 channels (v2ProPlus stage 0) stays on tapgemm: 96 fragments per wave spill, and its 500 rows per 10 s of
    // audio give a block one tile to amortise a 393 KB weight load over (measured 47 vs 40 us)
    GSV_WUPS(384, 2, 64, 2, 1, 264)
    GSV_WUPS(192, 4, 64, 4, 1, 256)
    GSV_WUPS(96, 2, 128, 1, 2, 512)
    GSV_WUPS(48, 1, 256, 1, 2, 768)
#undef GSV_WUPS
    return -1;
}

}  // namespace
