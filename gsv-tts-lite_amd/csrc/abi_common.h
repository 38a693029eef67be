// Host-side helpers shared by the translation units of the C-ABI library (gsv_abi.hip: GPT; gsv_voc.hip: SoVITS):
// error plumbing, the tapgemm / wconv / wups launchers and the packed-conv record.  Everything sits in an anonymous
// namespace: each translation unit gets its own copy and instantiates only the kernels it launches.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/gsv_tts_hip.h"
#include "t2s_prefill.h"
#include "tapgemm.h"
#include "wconv.h"
#include "wups.h"
#include "cgemm.h"
#include "wdma.h"
#include "voc_kernels.h"
#include "gsv_error.h"

using namespace gsv;

namespace {

// formats into the library's thread-local error string (owned by gsv_abi.hip) and returns `code`
int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return gsv::abi_fail(code, "%s", buf);
}

}  // namespace

namespace {

// Device memory a HANDLE owns (packed weights, biases, scratch) is requested through these two calls, never through hipMalloc /
// hipFree directly: a translation unit that keeps its handles' memory in an arena (gsv_abi.hip: GSV_DEV_ALLOC_ARENA, one arena per
// GPT handle) routes them there; elsewhere they are the runtime's allocator.
#ifdef GSV_DEV_ALLOC_ARENA
template <typename T> inline hipError_t gsv_dev_malloc(T** p, size_t n) { return gsv_arena::amalloc((void**)p, n); }
inline hipError_t gsv_dev_free(void* p) { return gsv_arena::afree(p); }
#else
template <typename T> inline hipError_t gsv_dev_malloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t gsv_dev_free(void* p) { return hipFree(p); }
#endif

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(GSV_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---------------------------------------------------------------------------------------------
// tapgemm host side
// ---------------------------------------------------------------------------------------------
struct PackedConv {
    void* w = nullptr;
    void* cg = nullptr;        // the same weights in cgemm.h's plane order (wide resblock convs, bf16 handles), or null
    float* bias = nullptr;
    int cout = 0, cin = 0, cin_pad = 0, k = 1, dil = 1, pad = 0, u = 0;
    int nphase = 1, ntaps = 1, mtiles = 1;
};

template <typename CT>
int pack_conv(PackedConv& pc, const float* src, int cout, int cin, int k, int64_t sm, int64_t sc, int64_t sk,
              int dil, int pad, int u, const float* bias_src, float bias_scale, hipStream_t st) {
    constexpr int KS = MfmaK<CT>::KS;
    if (cin % KS != 0) return fail(GSV_ERR_ARG, "tapgemm: cin %d not a multiple of %d", cin, KS);
    pc.cout = cout; pc.cin = cin; pc.cin_pad = cin; pc.k = k; pc.dil = dil; pc.pad = pad; pc.u = u;
    pc.nphase = u > 0 ? u : 1;
    pc.ntaps = u > 0 ? cdiv(k, u) : k;
    pc.mtiles = cdiv(cout, 32);
    if (pc.nphase > 10 || pc.ntaps > 12) return fail(GSV_ERR_ARG, "tapgemm: too many phases/taps");
    const size_t elems = (size_t)pc.nphase * pc.ntaps * pc.mtiles * (cin / KS) * 64 * (KS / 2);
    // + one all-zero fragment: what the pipelined loop fetches for iterations past the end
    HIPCHK(gsv_dev_malloc(&pc.w, (elems + 64 * (KS / 2)) * sizeof(CT)));
    HIPCHK(hipMemsetAsync((CT*)pc.w + elems, 0, 64 * (KS / 2) * sizeof(CT), st));
    const int blocks = (int)std::min<size_t>(2048, (elems + 255) / 256);
    hipLaunchKernelGGL((tapgemm_pack_kernel<CT>), dim3(blocks), dim3(256), 0, st, src, (CT*)pc.w, cout, cin, k, sm, sc,
                       sk, pc.nphase, pc.ntaps, u, pad, pc.mtiles);
    if (bias_src) {
        HIPCHK(gsv_dev_malloc(&pc.bias, sizeof(float) * cout));
        hipLaunchKernelGGL(scale_copy_kernel, dim3(cdiv(cout, 256)), dim3(256), 0, st, bias_src, pc.bias, (size_t)cout,
                           bias_scale);
    }
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

void free_conv(PackedConv& pc) {
    if (pc.w) (void)gsv_dev_free(pc.w);
    if (pc.cg) (void)gsv_dev_free(pc.cg);
    pc.cg = nullptr;
    if (pc.bias) (void)gsv_dev_free(pc.bias);
    pc.w = nullptr; pc.bias = nullptr;
}

struct Epi {
    const float* add = nullptr; int ld_add = 0; const int* add_index = nullptr;
    const void* res = nullptr; int ld_res = 0;
    const float* mask = nullptr;
    float scale = 1.0f; int act = ACT_NONE; int accumulate = 0; float in_slope = 1.0f;
    bool use_bias = true;
    bool fixed_order = false;   // the contraction's summation order must not depend on the row count: always the split-K tile (the prompt pass:
                                // a request's rows must not change with how many prompts share its pass)
};

struct Branch {
    const PackedConv* pc;
    const void* X;
    void* Y;
    const void* res;
};

// One launch for up to 3 convolutions of the same shape class (same cin/cout/ld/rows, different
// kernel size, dilation, weights and buffers): blockIdx.z is the branch.
template <typename IT, typename CT, typename OT>
int run_conv_multi(const Branch* brs, int nbr, int ldx, int n_in, int ldy, int n_rows, const Epi& e, hipStream_t st) {
    const PackedConv& pc = *brs[0].pc;
    if (nbr < 1 || nbr > 3) return fail(GSV_ERR_ARG, "tapgemm: 1..3 branches");
    for (int i = 1; i < nbr; ++i)
        if (brs[i].pc->cout != pc.cout || brs[i].pc->cin != pc.cin || brs[i].pc->u != 0 || pc.u != 0)
            return fail(GSV_ERR_ARG, "tapgemm: branches must be plain convs of one shape");
    TapGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.X = brs[0].X; a.ldx = ldx; a.n_in = n_in; a.cin = pc.cin; a.W = pc.w; a.cout = pc.cout; a.mtiles = pc.mtiles;
    a.ntaps = pc.ntaps; a.nphase = pc.nphase;
    a.tstep = pc.dil; a.tpad = pc.pad; a.tu = pc.u;
    a.omul = pc.u > 0 ? pc.u : 1;
    a.nbranch = nbr;
    if (nbr > 1) { a.X1 = brs[1].X; a.W1 = brs[1].pc->w; a.res1 = brs[1].res; a.bias1 = e.use_bias ? brs[1].pc->bias : nullptr; a.Y1 = brs[1].Y;
                   a.ntaps1 = brs[1].pc->ntaps; a.tstep1 = brs[1].pc->dil; a.tpad1 = brs[1].pc->pad; }
    if (nbr > 2) { a.X2 = brs[2].X; a.W2 = brs[2].pc->w; a.res2 = brs[2].res; a.bias2 = e.use_bias ? brs[2].pc->bias : nullptr; a.Y2 = brs[2].Y;
                   a.ntaps2 = brs[2].pc->ntaps; a.tstep2 = brs[2].pc->dil; a.tpad2 = brs[2].pc->pad; }
    a.in_slope = e.in_slope; a.bias = e.use_bias ? pc.bias : nullptr; a.add = e.add; a.ld_add = e.ld_add; a.add_index = e.add_index;
    a.res = brs[0].res; a.ld_res = e.ld_res; a.mask = e.mask; a.scale = e.scale; a.act = e.act;
    a.accumulate = e.accumulate; a.Y = brs[0].Y; a.ldy = ldy; a.n_rows = n_rows;
    // tile choice.  Enough rows to fill the chip several times over -> wide tiles (weights reused
    // across 64 rows/channels per wave); short sequences (prefill, flow, conditioning GEMV) ->
    // one 32x32 tile per block with the 4 waves splitting K.
    int span = 0;
    for (int i = 0; i < nbr; ++i) {
        const PackedConv& q = *brs[i].pc;
        for (int r = 0; r < q.nphase; ++r) {
            int lo = 1 << 30, hi = -(1 << 30);
            for (int t = 0; t < q.ntaps; ++t) {
                const int sh = q.u > 0 ? (r + q.pad) / q.u - t : t * q.dil - q.pad;
                lo = std::min(lo, sh); hi = std::max(hi, sh);
            }
            span = std::max(span, hi - lo);
        }
    }
    const int nz = nbr > 1 ? nbr : pc.nphase;
    const long tiles11 = (long)cdiv(n_rows, 128) * pc.mtiles * nz;   // blocks at (WM,WN) = (1,1)
    const bool splitk = e.fixed_order || tiles11 < 256;
    const bool wide_m = !splitk && pc.mtiles >= 2 && tiles11 >= 1024;
    const bool wide_n = !splitk && (long)cdiv(n_rows, 256) * cdiv(pc.mtiles, wide_m ? 2 : 1) * nz >= 1024;
    // mid-size problems (the 256-channel resblock stage: 5000 rows x 8 m-tiles x 3 branches): 64-row waves at two
    // blocks per CU measured 40.9 us vs 48.8 us for the 32-row tile (tools/tg_bench.hip)
    const bool mid_n = !splitk && !wide_m && !wide_n && (long)cdiv(n_rows, 256) * pc.mtiles * nz >= 256;
    const int bn = splitk ? 32 : ((wide_n || mid_n) ? 256 : 128);
    const int kcb = (wide_n || mid_n) ? 128 : 256;    // staged bytes per row per chunk
    size_t lds = (size_t)(bn + span) * (kcb + 16);
    if (splitk) lds = std::max(lds, (size_t)3 * 16 * 64 * sizeof(float));
    if (lds > 160 * 1024) return fail(GSV_ERR_ARG, "tapgemm: tap span %d needs %zu B of LDS", span, lds);
    dim3 blk(256);
    auto launch = [&](auto kern, dim3 grid) -> int {
        if (lds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, grid, blk, lds, st, a);
        return GSV_OK;
    };
    int rc;
    if (splitk) rc = launch(tapgemm_kernel<IT, CT, OT, 1, 1, 256, true>, dim3(cdiv(n_rows, 32), pc.mtiles, nz));
    else if (wide_m && wide_n) rc = launch(tapgemm_kernel<IT, CT, OT, 2, 2, 128, false>, dim3(cdiv(n_rows, 256), cdiv(pc.mtiles, 2), nz));
    else if (wide_m) rc = launch(tapgemm_kernel<IT, CT, OT, 2, 1, 256, false>, dim3(cdiv(n_rows, 128), cdiv(pc.mtiles, 2), nz));
    else if (wide_n) rc = launch(tapgemm_kernel<IT, CT, OT, 1, 2, 128, false>, dim3(cdiv(n_rows, 256), pc.mtiles, nz));
    else if (mid_n) rc = launch(tapgemm_kernel<IT, CT, OT, 1, 2, 128, false, 1, 2, 4>, dim3(cdiv(n_rows, 256), pc.mtiles, nz));
    else rc = launch(tapgemm_kernel<IT, CT, OT, 1, 1, 256, false>, dim3(cdiv(n_rows, 128), pc.mtiles, nz));
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

template <typename IT, typename CT, typename OT>
int run_conv(const PackedConv& pc, const void* X, int ldx, int n_in, void* Y, int ldy, int n_rows, const Epi& e,
             hipStream_t st) {
    Branch b{&pc, X, Y, e.res};
    return run_conv_multi<IT, CT, OT>(&b, 1, ldx, n_in, ldy, n_rows, e, st);
}

// The weights-in-registers path for the Generator's resblock convs (wconv.h).  Returns 1 when the
// launch does not fit it (caller falls back to tapgemm): returns -1 then, 0 on success, > 0 = GSV_ERR_*.
// wide resblock convs on the LDS-tiled GEMM (cgemm.h): 384 and 192 channels always, 256 channels from 16k rows on (below that
// the weights-in-registers kernel's shorter block wins: measured 33 vs 39 us per launch at 5 000 rows, 267 vs 207 at 50 000)
template <typename AT>
int run_cgemm(const Branch* brs, int ld, int n_rows, float in_slope, float out_slope, hipStream_t st) {
    (void)brs; (void)ld; (void)n_rows; (void)in_slope; (void)out_slope; (void)st;
    return -1;
}
template <>
inline int run_cgemm<bf16_t>(const Branch* brs, int ld, int n_rows, float in_slope, float out_slope, hipStream_t st) {
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
        a.nb0 = tiles; a.nb1 = tiles;
        HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(3 * tiles), dim3(nt), lds, st, a);
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

inline bool wconv_channels(int C) {
    return C == 16 || C == 24 || C == 32 || C == 48 || C == 64 || C == 96 || C == 128 || C == 192 || C == 256;
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
    if (C == 256) return launch(wconv_kernel<256, 2, 64, 2, 4>, wconv_lds_bytes<256, 2, 64, 2, 4>());   // K split in the block, slices over 4 blocks
    if (C == 192) return launch(wconv_kernel<192, 2, 64, 2, 3>, wconv_lds_bytes<192, 2, 64, 2, 3>());
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
    static const bool off = getenv("GSV_NO_WDMA") != nullptr;
    return !off && ld == C && (C == 64 || C == 128 || (C == 256 && n_rows < 16384));
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
