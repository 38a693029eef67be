// Host-side helpers shared by the translation units of the C-ABI library (gsv_abi.hip: GPT; gsv_voc.hip: SoVITS):
// error plumbing, the tapgemm launchers and the packed-conv record (the Generator's own launchers: voc_launch.h, gsv_voc.hip only).  Everything sits in an anonymous
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
    // staged bytes per row per chunk.  The split-K tile of an fp32 conv (the parity mode's prompt pass: 97 launches per prompt) stages 256 channels
    // per chunk instead of 64: a chunk is one dependent load -> LDS -> MFMA round trip with two k-steps per wave at 64 channels, and a K = 512 /
    // 2048 contraction paid 8 / 32 of them (21-74 us per launch, 4.3 ms per prompt; round 6).  Row-count independent like every split-K tile.
    constexpr int kcb_splitk = sizeof(CT) == 4 ? 1024 : 256;
    const int kcb = splitk ? kcb_splitk : ((wide_n || mid_n) ? 128 : 256);
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
    if (splitk) rc = launch(tapgemm_kernel<IT, CT, OT, 1, 1, kcb_splitk, true>, dim3(cdiv(n_rows, 32), pc.mtiles, nz));
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

}  // namespace
