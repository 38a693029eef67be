// C ABI of the MI355X GPT-SoVITS hot path (include/gsv_tts_hip.h): handle management, weight
// repacking into library-owned arenas, kernel sequencing, hipGraph capture of the decode step.
// No allocation happens inside a step; nothing here falls back to a CPU or library path.
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
#include "t2s_decode.h"
#include "t2s_megastep.h"
#include "t2s_prefill.h"
#include "tapgemm.h"
#include "wconv.h"
#include "wups.h"
#include "flowfuse.h"
#include "encp.h"
#include "voc_kernels.h"
#include "gsv_error.h"

using namespace gsv;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

}  // namespace

int gsv::abi_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

namespace {

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
    HIPCHK(hipMalloc(&pc.w, (elems + 64 * (KS / 2)) * sizeof(CT)));
    HIPCHK(hipMemsetAsync((CT*)pc.w + elems, 0, 64 * (KS / 2) * sizeof(CT), st));
    const int blocks = (int)std::min<size_t>(2048, (elems + 255) / 256);
    hipLaunchKernelGGL((tapgemm_pack_kernel<CT>), dim3(blocks), dim3(256), 0, st, src, (CT*)pc.w, cout, cin, k, sm, sc,
                       sk, pc.nphase, pc.ntaps, u, pad, pc.mtiles);
    if (bias_src) {
        HIPCHK(hipMalloc(&pc.bias, sizeof(float) * cout));
        hipLaunchKernelGGL(scale_copy_kernel, dim3(cdiv(cout, 256)), dim3(256), 0, st, bias_src, pc.bias, (size_t)cout,
                           bias_scale);
    }
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

void free_conv(PackedConv& pc) {
    if (pc.w) (void)hipFree(pc.w);
    if (pc.bias) (void)hipFree(pc.bias);
    pc.w = nullptr; pc.bias = nullptr;
}

struct Epi {
    const float* add = nullptr; int ld_add = 0;
    const void* res = nullptr; int ld_res = 0;
    const float* mask = nullptr;
    float scale = 1.0f; int act = ACT_NONE; int accumulate = 0; float in_slope = 1.0f;
    bool use_bias = true;
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
    a.in_slope = e.in_slope; a.bias = e.use_bias ? pc.bias : nullptr; a.add = e.add; a.ld_add = e.ld_add;
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
    const bool splitk = tiles11 < 256;
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


// Upsampling layer (transposed conv) on the weights-in-registers kernel (wups.h); -1 = shape not covered (caller
// falls back to tapgemm), 0 = launched, > 0 = GSV_ERR_*.
template <typename AT>
int run_wups(const PackedConv& pc, const void* X, int ldx, int n_in, void* Y, int ldy, float in_slope, hipStream_t st) {
    (void)pc; (void)X; (void)ldx; (void)n_in; (void)Y; (void)ldy; (void)in_slope; (void)st;
    return -1;
}
template <>
int run_wups<bf16_t>(const PackedConv& pc, const void* X, int ldx, int n_in, void* Y, int ldy, float in_slope, hipStream_t st) {
    if (pc.u < 1 || getenv("GSV_NO_WUPS")) return -1;
    WUpsArgs a;
    a.X = (const bf16_t*)X; a.W = (const uint4*)pc.w; a.bias = pc.bias; a.Y = (bf16_t*)Y;
    a.ldx = ldx; a.ldy = ldy; a.n_in = n_in; a.u = pc.u; a.tpad = pc.pad; a.mtiles = pc.mtiles; a.cout = pc.cout;
    a.cvalid = std::min(ldy, (pc.cout + 15) / 16 * 16); a.in_slope = in_slope;
    auto launch = [&](auto kern, size_t lds, int ms, int bn, int pg, int max_blocks) -> int {
        if (pc.u % pg != 0) return -1;
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
    GSV_WUPS(512, 4, 32, 2, 1, 512)
    GSV_WUPS(256, 4, 64, 2, 2, 256)
    GSV_WUPS(128, 2, 128, 4, 2, 256)
    GSV_WUPS(64, 1, 256, 1, 2, 512)
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

// =============================================================================================
// GPT
// =============================================================================================
struct T2SLayer {
    void *wqkv_p = nullptr, *wo_p = nullptr, *w1 = nullptr, *w2_p = nullptr;  // decode panels (WT)
    float *bqkv_p = nullptr, *bo = nullptr, *b1 = nullptr, *b2 = nullptr, *ln1g = nullptr, *ln1b = nullptr,
          *ln2g = nullptr, *ln2b = nullptr;
    PackedConv g_qkv, g_out, g_w1, g_w2;  // prefill (tapgemm fragments)
    unsigned have = 0;
};

struct T2SBound {
    gsv_t2s_state st;
    hipGraphExec_t graph = nullptr;       // 2-kernels-per-layer step
    hipGraphExec_t graph_mega = nullptr;  // persistent (megastep) step
};

struct gsv_t2s {
    gsv_t2s_config cfg;
    std::vector<T2SLayer> layers;
    void* predict = nullptr;  // WT [V][512]
    float *emb_audio = nullptr, *emb_text = nullptr, *pe_audio = nullptr, *pe_text = nullptr;
    PackedConv g_bert;
    unsigned have_io = 0;
    bool finalized = false;
    std::map<int, T2SBound> bound;
    // scratch sized for the largest bound batch
    int scratch_b = 0;
    float *xcur = nullptr, *xbuf = nullptr, *x1buf = nullptr, *ypart = nullptr, *zpart = nullptr;
    TokPart* tokpart = nullptr;
    hipStream_t cap_stream = nullptr;
    unsigned long long* dbg = nullptr;
    void* mega_layers = nullptr;   // device MegaLayer<WT>[n_layer]
    unsigned* mega_cnt = nullptr;  // [scratch_b][2*n_layer] + 1 (err)
};

namespace {

template <typename WT>
int t2s_load_layer_tensor(gsv_t2s* h, int l, const std::string& key, const float* data, int64_t numel, hipStream_t st) {
    T2SLayer& L = h->layers[l];
    auto want = [&](int64_t n) { return numel == n ? GSV_OK : fail(GSV_ERR_ARG, "layer %d %s: numel %lld, expected %lld", l, key.c_str(), (long long)numel, (long long)n); };
    auto copy_f32 = [&](float** dst, int64_t n, unsigned bit) -> int {
        if (int rc = want(n)) return rc;
        if (!*dst) HIPCHK(hipMalloc(dst, sizeof(float) * n));
        HIPCHK(hipMemcpyAsync(*dst, data, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
        L.have |= bit;
        return GSV_OK;
    };
    if (key == "qkv.weight") {
        if (int rc = want(3LL * kD * kD)) return rc;
        if (!L.wqkv_p) HIPCHK(hipMalloc(&L.wqkv_p, sizeof(WT) * 3 * kD * kD));
        hipLaunchKernelGGL((pack_qkv_panel_kernel<WT>), dim3(1024), dim3(256), 0, st, data, (WT*)L.wqkv_p);
        float* keep = L.g_qkv.bias; L.g_qkv.bias = nullptr;
        free_conv(L.g_qkv);
        if (int rc = pack_conv<WT>(L.g_qkv, data, 3 * kD, kD, 1, kD, 1, 0, 1, 0, 0, nullptr, 1.f, st)) return rc;
        L.g_qkv.bias = keep;
        L.have |= 1u << 0;
    } else if (key == "qkv.bias") {
        if (int rc = want(3 * kD)) return rc;
        if (!L.bqkv_p) HIPCHK(hipMalloc(&L.bqkv_p, sizeof(float) * 3 * kD));
        hipLaunchKernelGGL(pack_qkv_bias_kernel, dim3(6), dim3(256), 0, st, data, L.bqkv_p);
        if (!L.g_qkv.bias) HIPCHK(hipMalloc(&L.g_qkv.bias, sizeof(float) * 3 * kD));
        HIPCHK(hipMemcpyAsync(L.g_qkv.bias, data, sizeof(float) * 3 * kD, hipMemcpyDeviceToDevice, st));
        L.have |= 1u << 1;
    } else if (key == "out_proj.weight") {
        if (int rc = want((int64_t)kD * kD)) return rc;
        if (!L.wo_p) HIPCHK(hipMalloc(&L.wo_p, sizeof(WT) * kD * kD));
        hipLaunchKernelGGL((pack_col_panel_kernel<WT>), dim3(512), dim3(256), 0, st, data, (WT*)L.wo_p, kH, kDh);
        free_conv(L.g_out);
        if (int rc = pack_conv<WT>(L.g_out, data, kD, kD, 1, kD, 1, 0, 1, 0, 0, nullptr, 1.f, st)) return rc;
        L.have |= 1u << 2;
    } else if (key == "out_proj.bias") {
        if (int rc = copy_f32(&L.bo, kD, 1u << 3)) return rc;
    } else if (key == "mlp.0.weight") {
        if (int rc = want((int64_t)kF * kD)) return rc;
        if (!L.w1) HIPCHK(hipMalloc(&L.w1, sizeof(WT) * kF * kD));
        hipLaunchKernelGGL((convert_kernel<WT>), dim3(1024), dim3(256), 0, st, data, (WT*)L.w1, (size_t)kF * kD);
        free_conv(L.g_w1);
        if (int rc = pack_conv<WT>(L.g_w1, data, kF, kD, 1, kD, 1, 0, 1, 0, 0, nullptr, 1.f, st)) return rc;
        L.have |= 1u << 4;
    } else if (key == "mlp.0.bias") {
        if (int rc = copy_f32(&L.b1, kF, 1u << 5)) return rc;
    } else if (key == "mlp.2.weight") {
        if (int rc = want((int64_t)kD * kF)) return rc;
        if (!L.w2_p) HIPCHK(hipMalloc(&L.w2_p, sizeof(WT) * kD * kF));
        hipLaunchKernelGGL((pack_col_panel_kernel<WT>), dim3(1024), dim3(256), 0, st, data, (WT*)L.w2_p, kNJ, kFJ);
        free_conv(L.g_w2);
        if (int rc = pack_conv<WT>(L.g_w2, data, kD, kF, 1, kF, 1, 0, 1, 0, 0, nullptr, 1.f, st)) return rc;
        L.have |= 1u << 6;
    } else if (key == "mlp.2.bias") {
        if (int rc = copy_f32(&L.b2, kD, 1u << 7)) return rc;
    } else if (key == "norm1.weight") {
        if (int rc = copy_f32(&L.ln1g, kD, 1u << 8)) return rc;
    } else if (key == "norm1.bias") {
        if (int rc = copy_f32(&L.ln1b, kD, 1u << 9)) return rc;
    } else if (key == "norm2.weight") {
        if (int rc = copy_f32(&L.ln2g, kD, 1u << 10)) return rc;
    } else if (key == "norm2.bias") {
        if (int rc = copy_f32(&L.ln2b, kD, 1u << 11)) return rc;
    } else {
        return fail(GSV_ERR_ARG, "unknown layer tensor '%s'", key.c_str());
    }
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

template <typename WT>
int t2s_load_io_tensor(gsv_t2s* h, const std::string& name, const float* data, int64_t numel, hipStream_t st) {
    const gsv_t2s_config& c = h->cfg;
    auto copy_f32 = [&](float** dst, int64_t n, unsigned bit) -> int {
        if (numel != n) return fail(GSV_ERR_ARG, "%s: numel %lld, expected %lld", name.c_str(), (long long)numel, (long long)n);
        if (!*dst) HIPCHK(hipMalloc(dst, sizeof(float) * n));
        HIPCHK(hipMemcpyAsync(*dst, data, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
        h->have_io |= bit;
        return GSV_OK;
    };
    if (name == "ar_predict_layer.weight") {
        if (numel != (int64_t)c.vocab * kD) return fail(GSV_ERR_ARG, "%s: bad numel", name.c_str());
        if (!h->predict) HIPCHK(hipMalloc(&h->predict, sizeof(WT) * c.vocab * kD));
        hipLaunchKernelGGL((convert_kernel<WT>), dim3(512), dim3(256), 0, st, data, (WT*)h->predict, (size_t)c.vocab * kD);
        h->have_io |= 1u << 0;
    } else if (name == "ar_audio_embedding.word_embeddings.weight") {
        return copy_f32(&h->emb_audio, (int64_t)c.vocab * kD, 1u << 1);
    } else if (name == "ar_text_embedding.word_embeddings.weight") {
        return copy_f32(&h->emb_text, (int64_t)c.n_phoneme * kD, 1u << 2);
    } else if (name == "ar_audio_position.pe_scaled") {
        return copy_f32(&h->pe_audio, (int64_t)c.n_pos * kD, 1u << 3);
    } else if (name == "ar_text_position.pe_scaled") {
        return copy_f32(&h->pe_text, (int64_t)c.n_pos * kD, 1u << 4);
    } else if (name == "bert_proj.weight") {
        if (numel != (int64_t)kD * 1024) return fail(GSV_ERR_ARG, "%s: bad numel", name.c_str());
        float* keep = h->g_bert.bias; h->g_bert.bias = nullptr;
        free_conv(h->g_bert);
        if (int rc = pack_conv<WT>(h->g_bert, data, kD, 1024, 1, 1024, 1, 0, 1, 0, 0, nullptr, 1.f, st)) return rc;
        h->g_bert.bias = keep;
        h->have_io |= 1u << 5;
    } else if (name == "bert_proj.bias") {
        return copy_f32(&h->g_bert.bias, kD, 1u << 6);
    } else if (name == "ar_text_position.alpha" || name == "ar_audio_position.alpha") {
        return GSV_OK;  // folded into the pe_scaled tables by the caller
    } else {
        return fail(GSV_ERR_ARG, "unknown tensor '%s'", name.c_str());
    }
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

int t2s_ensure_scratch(gsv_t2s* h, int B) {
    if (B <= h->scratch_b) return GSV_OK;
    for (void* p : {(void*)h->xcur, (void*)h->xbuf, (void*)h->x1buf, (void*)h->ypart, (void*)h->zpart, (void*)h->tokpart})
        if (p) (void)hipFree(p);
    HIPCHK(hipMalloc(&h->xcur, sizeof(float) * B * kD));
    HIPCHK(hipMalloc(&h->xbuf, sizeof(float) * B * kD));
    HIPCHK(hipMalloc(&h->x1buf, sizeof(float) * B * kD));
    HIPCHK(hipMalloc(&h->ypart, sizeof(float) * B * kH * kD));
    HIPCHK(hipMalloc(&h->zpart, sizeof(float) * B * kNJ * kD));
    HIPCHK(hipMalloc(&h->tokpart, sizeof(TokPart) * B * kNP));
    HIPCHK(hipMemset(h->tokpart, 0, sizeof(TokPart) * B * kNP));
    if (h->mega_cnt) (void)hipFree(h->mega_cnt);
    HIPCHK(hipMalloc(&h->mega_cnt, sizeof(unsigned) * ((size_t)B * 2 * h->cfg.n_layer + 1)));
    HIPCHK(hipMemset(h->mega_cnt, 0, sizeof(unsigned) * ((size_t)B * 2 * h->cfg.n_layer + 1)));
    h->scratch_b = B;
    // graphs captured against the old scratch pointers are stale
    for (auto& kv : h->bound)
        { if (kv.second.graph) { (void)hipGraphExecDestroy(kv.second.graph); kv.second.graph = nullptr; }
          if (kv.second.graph_mega) { (void)hipGraphExecDestroy(kv.second.graph_mega); kv.second.graph_mega = nullptr; } }
    return GSV_OK;
}

template <typename WT>
void t2s_launch_attn(gsv_t2s* h, const gsv_t2s_state& s, int l, const float* xsrc, hipStream_t st) {
    const int B = s.batch, T = s.max_kv;
    const size_t lds = 0;  // static LDS only: scores never leave registers
    const size_t layer_elems = (size_t)B * kH * T * kDh;
    T2SLayer& L = h->layers[l];
    AttnArgs<WT> a;
    a.xdirect = xsrc;
    a.zpart = h->zpart;
    a.b2 = l ? h->layers[l - 1].b2 : nullptr;
    a.x1 = h->x1buf;
    a.ln2g = l ? h->layers[l - 1].ln2g : nullptr;
    a.ln2b = l ? h->layers[l - 1].ln2b : nullptr;
    a.xout = h->xbuf;
    a.wqkv = (const WT*)L.wqkv_p; a.bqkv = L.bqkv_p; a.wo = (const WT*)L.wo_p;
    a.kc = (WT*)s.k_cache + (size_t)l * layer_elems;
    a.vc = (WT*)s.v_cache + (size_t)l * layer_elems;
    a.kv_len = s.kv_len; a.T = T; a.ypart = h->ypart; a.dbg = (l == h->cfg.n_layer - 1) ? h->dbg : nullptr;
    if (l == 0) hipLaunchKernelGGL((t2s_attn_kernel<WT, 0>), dim3(kH, B), dim3(kNT), lds, st, a);
    else hipLaunchKernelGGL((t2s_attn_kernel<WT, 1>), dim3(kH, B), dim3(kNT), lds, st, a);
}

template <typename WT>
void t2s_launch_ffn(gsv_t2s* h, const gsv_t2s_state& s, int l, hipStream_t st) {
    T2SLayer& L = h->layers[l];
    FfnArgs<WT> f;
    f.ypart = h->ypart; f.bo = L.bo; f.x = h->xbuf; f.ln1g = L.ln1g; f.ln1b = L.ln1b; f.x1out = h->x1buf;
    f.w1 = (const WT*)L.w1; f.b1 = L.b1; f.w2p = (const WT*)L.w2_p; f.zpart = h->zpart; f.dbg = (l == h->cfg.n_layer - 1) ? h->dbg : nullptr;
    hipLaunchKernelGGL((t2s_ffn_kernel<WT>), dim3(kNJ, s.batch), dim3(kNT), 0, st, f);
}

// the transformer stack for one token per slot; x from `xsrc` [B][512]
template <typename WT>
int t2s_layers(gsv_t2s* h, const gsv_t2s_state& s, const float* xsrc, hipStream_t st) {
    for (int l = 0; l < h->cfg.n_layer; ++l) {
        t2s_launch_attn<WT>(h, s, l, xsrc, st);
        t2s_launch_ffn<WT>(h, s, l, st);
    }
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

template <typename WT>
int t2s_logits(gsv_t2s* h, const gsv_t2s_state& s, int mode, const float* hdirect, int slot0, int nrows, int vlimit,
               int bump, hipStream_t st, const int32_t* slots = nullptr) {
    const T2SLayer& L = h->layers.back();
    LogitsArgs<WT> a;
    a.hdirect = hdirect; a.zpart = h->zpart; a.b2 = L.b2; a.x1 = h->x1buf; a.ln2g = L.ln2g; a.ln2b = L.ln2b;
    a.wp = (const WT*)h->predict; a.V = h->cfg.vocab; a.eos = h->cfg.eos; a.vlimit = vlimit; a.slot0 = slot0; a.slots = slots;
    a.step = s.step; a.ctl = s.ctl; a.fctl = s.fctl; a.seen = s.seen; a.logits = s.logits; a.hidden = s.hidden;
    a.tokpart = h->tokpart; a.kv_len = s.kv_len; a.bump = bump;
    if (mode == 0) hipLaunchKernelGGL((t2s_logits_kernel<WT, 0>), dim3(kNP, nrows), dim3(kNT), 0, st, a);
    else hipLaunchKernelGGL((t2s_logits_kernel<WT, 1>), dim3(kNP, nrows), dim3(kNT), 0, st, a);
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

int t2s_token(gsv_t2s* h, const gsv_t2s_state& s, int advance, hipStream_t st) {
    TokenArgs a;
    a.tokpart = h->tokpart; a.tok_override = s.tok_override; a.ctl = s.ctl; a.kv_len = s.kv_len; a.x_len = s.x_len;
    a.pre_tokens = s.pre_tokens; a.seen = s.seen; a.step = s.step; a.eos_at = s.eos_at; a.emb = h->emb_audio;
    a.pe = h->pe_audio; a.xcur = h->xcur; a.T = s.max_kv; a.V = h->cfg.vocab; a.eos = h->cfg.eos; a.n_pos = h->cfg.n_pos;
    a.advance = advance;
    a.logits = s.logits; a.fctl = s.fctl;
    a.mega_cnt = h->mega_cnt; a.mega_n = 2 * h->cfg.n_layer;
    if (a.mega_n > 256) a.mega_cnt = nullptr;
    hipLaunchKernelGGL(t2s_token_kernel, dim3(s.batch), dim3(256), 0, st, a);
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

constexpr int kMegaMaxBatch = 4;  // 48 blocks per sequence, one block per CU, all co-resident

template <typename WT>
int t2s_mega_layers(gsv_t2s* h, const gsv_t2s_state& s, const float* xsrc, hipStream_t st) {
    const int NL = h->cfg.n_layer, B = s.batch;
    if (B > kMegaMaxBatch) return fail(GSV_ERR_ARG, "megastep supports batch <= %d", kMegaMaxBatch);
    if (2 * NL > 256) return fail(GSV_ERR_ARG, "megastep supports at most 128 layers");
    // hand-off counters are zeroed by the token kernel that precedes this launch in every step
    MegaArgs<WT> a;
    a.layers = (const MegaLayer<WT>*)h->mega_layers; a.n_layer = NL; a.xin = xsrc; a.xbuf = h->xbuf; a.x1buf = h->x1buf;
    a.ypart = h->ypart; a.zpart = h->zpart; a.kc = (WT*)s.k_cache; a.vc = (WT*)s.v_cache;
    a.layer_elems = (size_t)B * kH * s.max_kv * kDh; a.T = s.max_kv; a.kv_len = s.kv_len; a.cnt = h->mega_cnt;
    a.err = h->mega_cnt + (size_t)h->scratch_b * 2 * NL;
    hipLaunchKernelGGL((t2s_megastep_kernel<WT>), dim3(kMegaRoles, B), dim3(kNT), 0, st, a);
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

// batched step (bf16, B >= kBatchedMin): the prompt GEMM chain on B rows + one attention block per (head, sequence)
constexpr int kBatchedMin = 36;   // measured: step 0.87 / 0.93 / 0.99 ms at B = 24 / 32 / 64 vs 0.82 / 0.91 / 1.65 for the per-sequence kernels

template <typename WT>
int t2s_batched_layers(gsv_t2s* h, const gsv_t2s_state& s, hipStream_t st) {
    const int B = s.batch, T = s.max_kv;
    float* qkv = h->ypart;                                   // [B][1536]
    float* attn = qkv + (size_t)B * 3 * kD;                  // [B][512]
    float* ybuf = attn + (size_t)B * kD;                     // [B][512]
    bf16_t* fb16 = (bf16_t*)(ybuf + (size_t)B * kD);         // [B][2048] bf16
    float* part = h->zpart;                                  // [4][B][512]
    float* x = h->xbuf;                                      // layer input / output
    const size_t layer_elems = (size_t)B * kH * T * kDh;
    const int rtiles = cdiv(B, 32);
    auto gemm = [&](auto kern, const void* X, int ldx, const PackedConv& pc, const float* bias, int relu, void* Y, int ldy, int nsplit,
                    size_t split_stride) {
        RowGemmArgs ra;
        ra.X = X; ra.ldx = ldx; ra.M = B; ra.W = (const uint4*)pc.w; ra.ksteps = pc.cin / 16; ra.ntaps = 1; ra.pad = 0; ra.mtiles = pc.mtiles;
        ra.bias = bias; ra.relu = relu; ra.Y = Y; ra.ldy = ldy; ra.split_stride = split_stride;
        hipLaunchKernelGGL(kern, dim3(rtiles, pc.mtiles, nsplit), dim3(256), 0, st, ra);
    };
    const float* xin = h->xcur;
    for (int l = 0; l < h->cfg.n_layer; ++l) {
        T2SLayer& L = h->layers[l];
        gemm(rowgemm_kernel<float, float>, xin, kD, L.g_qkv, L.g_qkv.bias, 0, qkv, 3 * kD, 1, 0);
        BatchAttnArgs<WT> ba;
        ba.qkv = qkv; ba.kc = (WT*)s.k_cache + (size_t)l * layer_elems; ba.vc = (WT*)s.v_cache + (size_t)l * layer_elems;
        ba.kv_len = s.kv_len; ba.T = T; ba.out = attn;
        hipLaunchKernelGGL((t2s_batch_attn_kernel<WT>), dim3(kH, B), dim3(256), 0, st, ba);
        gemm(rowgemm_kernel<float, float>, attn, kD, L.g_out, nullptr, 0, ybuf, kD, 1, 0);
        hipLaunchKernelGGL(ln_rows_sum_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, (const float*)ybuf, 1, (size_t)0, (const float*)L.bo, xin,
                           (const float*)L.ln1g, (const float*)L.ln1b, x, B);
        gemm(rowgemm_kernel<float, bf16_t>, x, kD, L.g_w1, L.b1, 1, fb16, kF, 1, 0);
        gemm(rowgemm_kernel<bf16_t, float>, fb16, kF, L.g_w2, nullptr, 0, part, kD, 4, (size_t)B * kD);
        hipLaunchKernelGGL(ln_rows_sum_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, (const float*)part, 4, (size_t)B * kD, (const float*)L.b2,
                           (const float*)x, (const float*)L.ln2g, (const float*)L.ln2b, x, B);
        xin = x;
    }
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

template <typename WT>
int t2s_step(gsv_t2s* h, const gsv_t2s_state& s, bool mega, hipStream_t st) {
    if (int rc = t2s_token(h, s, 1, st)) return rc;
    if constexpr (sizeof(WT) == 2) {
        if (s.batch >= kBatchedMin && s.max_kv <= 1024 && !getenv("GSV_NO_BATCHED_STEP")) {
            if (int rc = t2s_batched_layers<WT>(h, s, st)) return rc;
            return t2s_logits<WT>(h, s, 0, h->xbuf, 0, s.batch, h->cfg.vocab, 1, st);
        }
    }
    if (int rc = mega ? t2s_mega_layers<WT>(h, s, h->xcur, st) : t2s_layers<WT>(h, s, h->xcur, st)) return rc;
    return t2s_logits<WT>(h, s, 1, nullptr, 0, s.batch, h->cfg.vocab, 1, st);
}

template <typename WT>
int t2s_prefill_impl(gsv_t2s* h, T2SBound& bd, int slot0, int nrows, int l_max, float* xy, const int64_t* x_lens,
                     const int64_t* y_lens, void* ws, size_t ws_bytes, hipStream_t st, const int32_t* slots = nullptr) {
    const gsv_t2s_state& s = bd.st;
    const int M = nrows * l_max, T = s.max_kv;
    const size_t need = gsv_t2s_prefill_workspace(h, nrows, l_max);
    if (ws_bytes < need) return fail(GSV_ERR_ARG, "prefill workspace %zu < %zu", ws_bytes, need);
    float* qkv = (float*)ws;
    float* attn = qkv + (size_t)M * 3 * kD;
    float* ybuf = attn + (size_t)M * kD;
    float* fbuf = ybuf + (size_t)M * kD;
    float* hlast = fbuf + (size_t)M * kF;
    const size_t layer_elems = (size_t)s.batch * kH * T * kDh;
    const int qsplit = std::max(1, std::min(16, 256 / (kH * nrows)));
    const size_t lds = sizeof(float) * ((size_t)l_max * 33 + (size_t)l_max * 32 + 4 * (size_t)l_max + 128 + 8);
    if (lds > 160 * 1024) return fail(GSV_ERR_ARG, "prefill: prompt of %d positions exceeds the LDS-staged attention limit", l_max);
    HIPCHK(hipFuncSetAttribute((const void*)t2s_prefill_attn_kernel<WT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int nkt_max = cdiv(l_max, 32);
    const size_t lds_mfma = (size_t)nkt_max * 32 * 80 + 128 * 80 + (size_t)32 * (nkt_max * 64 + 16);
    if (sizeof(WT) == 2) {
        if (lds_mfma > 160 * 1024) return fail(GSV_ERR_ARG, "prefill: prompt of %d positions exceeds the LDS-staged attention limit", l_max);
        HIPCHK(hipFuncSetAttribute((const void*)t2s_prefill_attn_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mfma));
    }
    if constexpr (sizeof(WT) == 2) {
        // bf16 mode: latency-shaped GEMMs (rowgemm_kernel), flash attention on the matrix cores, and
        // bias + residual + LayerNorm in the consumer of the raw (split) GEMM tiles
        const int rtiles = cdiv(M, 32);
        bf16_t* fb16 = (bf16_t*)fbuf;                    // FFN hidden as bf16: the GEMM's operand type anyway
        bf16_t* xb16 = fb16 + (size_t)M * kF;            // bf16 copy of the LayerNorm output (second half of the fp32-sized fbuf)
        float* part = qkv;                               // W2 split partials reuse qkv + attn (dead by then): [4][M][512]
        auto gemm = [&](auto kern, const void* X, int ldx, const PackedConv& pc, const float* bias, int relu, void* Y, int ldy,
                        int nsplit, size_t split_stride) {
            RowGemmArgs ra;
            ra.X = X; ra.ldx = ldx; ra.M = M; ra.W = (const uint4*)pc.w; ra.ksteps = pc.cin / 16; ra.ntaps = 1; ra.pad = 0; ra.mtiles = pc.mtiles;
            ra.bias = bias; ra.relu = relu;
            ra.Y = Y; ra.ldy = ldy; ra.split_stride = split_stride;
            hipLaunchKernelGGL(kern, dim3(rtiles, pc.mtiles, nsplit), dim3(256), 0, st, ra);
        };
        for (int l = 0; l < h->cfg.n_layer; ++l) {
            T2SLayer& L = h->layers[l];
            if (l == 0) gemm(rowgemm_kernel<float, float>, xy, kD, L.g_qkv, L.g_qkv.bias, 0, qkv, 3 * kD, 1, 0);
            else gemm(rowgemm_kernel<bf16_t, float>, xb16, kD, L.g_qkv, L.g_qkv.bias, 0, qkv, 3 * kD, 1, 0);
            PrefillAttnMfmaArgs pm;
            pm.qkv = qkv; pm.x_lens = x_lens; pm.y_lens = y_lens;
            pm.kc = (bf16_t*)s.k_cache + (size_t)l * layer_elems; pm.vc = (bf16_t*)s.v_cache + (size_t)l * layer_elems;
            pm.T = T; pm.slot0 = slot0; pm.slots = slots; pm.l_max = l_max; pm.out = attn;
            hipLaunchKernelGGL(t2s_prefill_attn_mfma_kernel, dim3(kH, nrows, cdiv(l_max, 128)), dim3(256), lds_mfma, st, pm);
            gemm(rowgemm_kernel<float, float>, attn, kD, L.g_out, nullptr, 0, ybuf, kD, 1, 0);
            hipLaunchKernelGGL(ln_rows_sum_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, (const float*)ybuf, 1, (size_t)0, (const float*)L.bo,
                               (const float*)xy, (const float*)L.ln1g, (const float*)L.ln1b, xy, M, xb16);
            gemm(rowgemm_kernel<bf16_t, bf16_t>, xb16, kD, L.g_w1, L.b1, 1, fb16, kF, 1, 0);
            gemm(rowgemm_kernel<bf16_t, float>, fb16, kF, L.g_w2, nullptr, 0, part, kD, 4, (size_t)M * kD);
            hipLaunchKernelGGL(ln_rows_sum_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, (const float*)part, 4, (size_t)M * kD, (const float*)L.b2,
                               (const float*)xy, (const float*)L.ln2g, (const float*)L.ln2b, xy, M, xb16);
        }
        HIPCHK(hipGetLastError());
    } else {
        for (int l = 0; l < h->cfg.n_layer; ++l) {
            T2SLayer& L = h->layers[l];
            Epi e0;
            if (int rc = run_conv<float, WT, float>(L.g_qkv, xy, kD, M, qkv, 3 * kD, M, e0, st)) return rc;
            PrefillAttnArgs<WT> pa;
            pa.qkv = qkv; pa.x_lens = x_lens; pa.y_lens = y_lens;
            pa.kc = (WT*)s.k_cache + (size_t)l * layer_elems; pa.vc = (WT*)s.v_cache + (size_t)l * layer_elems;
            pa.T = T; pa.slot0 = slot0; pa.slots = slots; pa.l_max = l_max; pa.qsplit = qsplit; pa.out = attn;
            if constexpr (sizeof(WT) == 2) {   // bf16 cache: flash attention on the matrix cores
                PrefillAttnMfmaArgs pm;
                pm.qkv = qkv; pm.x_lens = x_lens; pm.y_lens = y_lens;
                pm.kc = (bf16_t*)s.k_cache + (size_t)l * layer_elems; pm.vc = (bf16_t*)s.v_cache + (size_t)l * layer_elems;
                pm.T = T; pm.slot0 = slot0; pm.slots = slots; pm.l_max = l_max; pm.out = attn;
                hipLaunchKernelGGL(t2s_prefill_attn_mfma_kernel, dim3(kH, nrows, cdiv(l_max, 128)), dim3(256), lds_mfma, st, pm);
            } else {
                hipLaunchKernelGGL((t2s_prefill_attn_kernel<WT>), dim3(kH, nrows, qsplit), dim3(256), lds, st, pa);
            }
            Epi e1; e1.res = xy; e1.ld_res = kD;
            // out_proj bias lives in the decode copy (L.bo); tapgemm bias pointer set per call
            PackedConv go = L.g_out; go.bias = L.bo;
            if (int rc = run_conv<float, WT, float>(go, attn, kD, M, ybuf, kD, M, e1, st)) return rc;
            hipLaunchKernelGGL(ln_rows_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, ybuf, L.ln1g, L.ln1b, xy, M);
            Epi e2; e2.act = ACT_RELU;
            PackedConv g1 = L.g_w1; g1.bias = L.b1;
            if (int rc = run_conv<float, WT, float>(g1, xy, kD, M, fbuf, kF, M, e2, st)) return rc;
            Epi e3; e3.res = xy; e3.ld_res = kD;
            PackedConv g2 = L.g_w2; g2.bias = L.b2;
            if (int rc = run_conv<float, WT, float>(g2, fbuf, kF, M, ybuf, kD, M, e3, st)) return rc;
            hipLaunchKernelGGL(ln_rows_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, ybuf, L.ln2g, L.ln2b, xy, M);
        }
    }
    PrefillFinishArgs fa;
    fa.hidden = xy; fa.x_lens = x_lens; fa.y_lens = y_lens; fa.hlast = hlast; fa.kv_len = s.kv_len; fa.x_len = s.x_len;
    fa.step = s.step; fa.eos_at = s.eos_at; fa.slot0 = slot0; fa.slots = slots; fa.l_max = l_max;
    hipLaunchKernelGGL(t2s_prefill_finish_kernel, dim3(nrows), dim3(128), 0, st, fa);
    HIPCHK(hipGetLastError());
    // first sample: logits[:, :-1] (t2s_model.py:417,613) -> EOS column dropped
    return t2s_logits<WT>(h, s, 0, hlast, slot0, nrows, h->cfg.vocab - 1, 0, st, slots);
}

}  // namespace

template <typename WT>
static int t2s_time_impl(gsv_t2s* h, T2SBound* b, int iters, float* out_ms, hipStream_t st) {
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    const int NL = h->cfg.n_layer;
    // save the sequence position: the sweeps below re-run real kernels on the live state
    for (int cls = 0; cls < 4; ++cls) {
        for (int rep = 0; rep < 2; ++rep) {  // rep 0 = warm-up
            HIPCHK(hipEventRecord(e0, st));
            int launches = 0;
            for (int it = 0; it < (rep ? iters : 1); ++it) {
                if (cls == 0) for (int l = 0; l < NL; ++l, ++launches) t2s_launch_attn<WT>(h, b->st, l, h->xcur, st);
                if (cls == 1) for (int l = 0; l < NL; ++l, ++launches) t2s_launch_ffn<WT>(h, b->st, l, st);
                if (cls == 2) { for (int l = 0; l < NL; ++l, ++launches) if (int rc = t2s_logits<WT>(h, b->st, 1, nullptr, 0, b->st.batch, h->cfg.vocab, 0, st)) return rc; }
                if (cls == 3) { for (int l = 0; l < NL; ++l, ++launches) if (int rc = t2s_token(h, b->st, 0, st)) return rc; }
            }
            HIPCHK(hipEventRecord(e1, st));
            HIPCHK(hipEventSynchronize(e1));
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) out_ms[cls] = ms / (float)launches;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return GSV_OK;
}

extern "C" {

int gsv_version(void) { return 1; }
const char* gsv_last_error(void) { return g_err.c_str(); }

int gsv_t2s_create(const gsv_t2s_config* cfg, gsv_t2s** out) {
    if (!cfg || !out) return fail(GSV_ERR_ARG, "null argument");
    if (cfg->hidden != kD || cfg->n_head != kH)
        return fail(GSV_ERR_ARG, "unsupported GPT shape: hidden %d heads %d (kernels are specialised for 512/16)", cfg->hidden, cfg->n_head);
    if (cfg->vocab < 2 || cfg->vocab > kNP * 128 || cfg->n_layer < 1 || cfg->n_pos < 1)
        return fail(GSV_ERR_ARG, "unsupported vocab/n_layer/n_pos");
    if (cfg->dtype != GSV_F32 && cfg->dtype != GSV_BF16) return fail(GSV_ERR_ARG, "bad dtype");
    gsv_t2s* h = new gsv_t2s();
    h->cfg = *cfg;
    h->layers.resize(cfg->n_layer);
    if (hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return fail(GSV_ERR_HIP, "hipStreamCreate failed");
    }
    *out = h;
    return GSV_OK;
}

int gsv_t2s_destroy(gsv_t2s* h) {
    if (!h) return GSV_OK;
    (void)hipDeviceSynchronize();
    for (auto& kv : h->bound) {
        if (kv.second.graph) (void)hipGraphExecDestroy(kv.second.graph);
        if (kv.second.graph_mega) (void)hipGraphExecDestroy(kv.second.graph_mega);
    }
    if (h->mega_layers) (void)hipFree(h->mega_layers);
    if (h->mega_cnt) (void)hipFree(h->mega_cnt);
    for (T2SLayer& L : h->layers) {
        for (void* p : {L.wqkv_p, L.wo_p, L.w1, L.w2_p, (void*)L.bqkv_p, (void*)L.bo, (void*)L.b1, (void*)L.b2,
                        (void*)L.ln1g, (void*)L.ln1b, (void*)L.ln2g, (void*)L.ln2b})
            if (p) (void)hipFree(p);
        free_conv(L.g_qkv); L.g_out.bias = nullptr; free_conv(L.g_out); L.g_w1.bias = nullptr; free_conv(L.g_w1);
        L.g_w2.bias = nullptr; free_conv(L.g_w2);
    }
    for (void* p : {h->predict, (void*)h->emb_audio, (void*)h->emb_text, (void*)h->pe_audio, (void*)h->pe_text,
                    (void*)h->xcur, (void*)h->xbuf, (void*)h->x1buf, (void*)h->ypart, (void*)h->zpart, (void*)h->tokpart})
        if (p) (void)hipFree(p);
    free_conv(h->g_bert);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    delete h;
    return GSV_OK;
}

int gsv_t2s_load_tensor(gsv_t2s* h, const char* name, const float* data, int64_t numel, void* stream) {
    if (!h || !name || !data) return fail(GSV_ERR_ARG, "null argument");
    std::string n(name);
    const std::string pre = "t2s_transformer.blocks.";
    h->finalized = false;
    if (n.compare(0, pre.size(), pre) == 0) {
        size_t dot = n.find('.', pre.size());
        if (dot == std::string::npos) return fail(GSV_ERR_ARG, "bad tensor name '%s'", name);
        int l = atoi(n.substr(pre.size(), dot - pre.size()).c_str());
        if (l < 0 || l >= h->cfg.n_layer) return fail(GSV_ERR_ARG, "layer index out of range in '%s'", name);
        std::string key = n.substr(dot + 1);
        return h->cfg.dtype == GSV_BF16 ? t2s_load_layer_tensor<bf16_t>(h, l, key, data, numel, S(stream))
                                        : t2s_load_layer_tensor<float>(h, l, key, data, numel, S(stream));
    }
    return h->cfg.dtype == GSV_BF16 ? t2s_load_io_tensor<bf16_t>(h, n, data, numel, S(stream))
                                    : t2s_load_io_tensor<float>(h, n, data, numel, S(stream));
}

int gsv_t2s_finalize(gsv_t2s* h, void* stream) {
    if (!h) return fail(GSV_ERR_ARG, "null handle");
    for (int l = 0; l < h->cfg.n_layer; ++l)
        if (h->layers[l].have != 0xfffu) return fail(GSV_ERR_STATE, "layer %d incomplete (mask 0x%x)", l, h->layers[l].have);
    if (h->have_io != 0x7fu) return fail(GSV_ERR_STATE, "embedding/predict tensors incomplete (mask 0x%x)", h->have_io);
    {   // device table of per-layer pointers for the persistent step
        struct Raw { const void *wqkv, *wo, *w1, *w2p; const float *bqkv, *bo, *b1, *b2, *ln1g, *ln1b, *ln2g, *ln2b; };
        static_assert(sizeof(Raw) == sizeof(MegaLayer<float>) && sizeof(Raw) == sizeof(MegaLayer<bf16_t>), "layout");
        std::vector<Raw> tab(h->cfg.n_layer);
        for (int l = 0; l < h->cfg.n_layer; ++l) {
            const T2SLayer& L = h->layers[l];
            tab[l] = Raw{L.wqkv_p, L.wo_p, L.w1, L.w2_p, L.bqkv_p, L.bo, L.b1, L.b2, L.ln1g, L.ln1b, L.ln2g, L.ln2b};
        }
        if (!h->mega_layers) HIPCHK(hipMalloc(&h->mega_layers, sizeof(Raw) * tab.size()));
        HIPCHK(hipMemcpy(h->mega_layers, tab.data(), sizeof(Raw) * tab.size(), hipMemcpyHostToDevice));
    }
    HIPCHK(hipStreamSynchronize(S(stream)));
    h->finalized = true;
    return GSV_OK;
}

int gsv_t2s_bind_state(gsv_t2s* h, const gsv_t2s_state* st) {
    if (!h || !st) return fail(GSV_ERR_ARG, "null argument");
    if (st->batch < 1 || st->max_kv < 2) return fail(GSV_ERR_ARG, "bad batch/max_kv");
    if (!st->k_cache || !st->v_cache || !st->kv_len || !st->x_len || !st->pre_tokens || !st->seen || !st->step ||
        !st->eos_at || !st->logits || !st->hidden || !st->tok_override || !st->ctl || !st->fctl)
        return fail(GSV_ERR_ARG, "state has null pointers");
    if (int rc = t2s_ensure_scratch(h, st->batch)) return rc;
    T2SBound& b = h->bound[st->batch];
    if (b.graph) { (void)hipGraphExecDestroy(b.graph); b.graph = nullptr; }
    if (b.graph_mega) { (void)hipGraphExecDestroy(b.graph_mega); b.graph_mega = nullptr; }
    b.st = *st;
    return GSV_OK;
}

static T2SBound* t2s_find(gsv_t2s* h, int batch) {
    auto it = h->bound.find(batch);
    return it == h->bound.end() ? nullptr : &it->second;
}

int gsv_t2s_embed_prompt(gsv_t2s* h, int nrows, int lx_max, int ly_max, int l_max, const int64_t* x_ids,
                         const int64_t* y_ids, const float* bert, const int64_t* x_lens, const int64_t* y_lens,
                         float* xy, float* scratch, void* stream) {
    if (!h || !h->finalized) return fail(GSV_ERR_STATE, "handle not finalized");
    if (nrows < 1 || lx_max < 1 || ly_max < 1 || l_max < 1 || l_max > h->cfg.n_pos) return fail(GSV_ERR_ARG, "bad sizes");
    const int M = nrows * lx_max;
    Epi e;
    int rc = h->cfg.dtype == GSV_BF16 ? run_conv<float, bf16_t, float>(h->g_bert, bert, 1024, M, scratch, kD, M, e, S(stream))
                                      : run_conv<float, float, float>(h->g_bert, bert, 1024, M, scratch, kD, M, e, S(stream));
    if (rc) return rc;
    EmbedArgs a;
    a.x_ids = x_ids; a.y_ids = y_ids; a.proj = scratch; a.x_lens = x_lens; a.y_lens = y_lens; a.emb_text = h->emb_text;
    a.emb_audio = h->emb_audio; a.pe_text = h->pe_text; a.pe_audio = h->pe_audio; a.xy = xy; a.lx_max = lx_max;
    a.ly_max = ly_max; a.l_max = l_max; a.n_phoneme = h->cfg.n_phoneme; a.V = h->cfg.vocab;
    hipLaunchKernelGGL(t2s_embed_kernel, dim3(l_max, nrows), dim3(128), 0, S(stream), a);
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

size_t gsv_t2s_prefill_workspace(gsv_t2s* h, int nrows, int l_max) {
    (void)h;
    const size_t M = (size_t)nrows * l_max;
    return sizeof(float) * (M * (3 * kD + kD + kD + kF) + (size_t)nrows * kD) + 256;
}

int gsv_t2s_prefill(gsv_t2s* h, int batch, int slot0, int nrows, int l_max, float* xy, const int64_t* x_lens,
                    const int64_t* y_lens, void* workspace, size_t workspace_bytes, void* stream) {
    if (!h || !h->finalized) return fail(GSV_ERR_STATE, "handle not finalized");
    T2SBound* b = t2s_find(h, batch);
    if (!b) return fail(GSV_ERR_STATE, "no state bound for batch %d", batch);
    if (slot0 < 0 || nrows < 1 || slot0 + nrows > batch) return fail(GSV_ERR_ARG, "slot range out of bounds");
    if (l_max < 1 || l_max > b->st.max_kv) return fail(GSV_ERR_ARG, "prompt of %d positions does not fit the KV cache (%d)", l_max, b->st.max_kv);
    return h->cfg.dtype == GSV_BF16
               ? t2s_prefill_impl<bf16_t>(h, *b, slot0, nrows, l_max, xy, x_lens, y_lens, workspace, workspace_bytes, S(stream))
               : t2s_prefill_impl<float>(h, *b, slot0, nrows, l_max, xy, x_lens, y_lens, workspace, workspace_bytes, S(stream));
}

int gsv_t2s_prefill_slots(gsv_t2s* h, int batch, const int32_t* slots, int nrows, int l_max, float* xy, const int64_t* x_lens,
                          const int64_t* y_lens, void* workspace, size_t workspace_bytes, void* stream) {
    if (!h || !h->finalized) return fail(GSV_ERR_STATE, "handle not finalized");
    T2SBound* b = t2s_find(h, batch);
    if (!b) return fail(GSV_ERR_STATE, "no state bound for batch %d", batch);
    if (!slots || nrows < 1 || nrows > batch) return fail(GSV_ERR_ARG, "prefill_slots: need 1..batch rows and their slot list");
    if (l_max < 1 || l_max > b->st.max_kv) return fail(GSV_ERR_ARG, "prompt of %d positions does not fit the KV cache (%d)", l_max, b->st.max_kv);
    return h->cfg.dtype == GSV_BF16
               ? t2s_prefill_impl<bf16_t>(h, *b, 0, nrows, l_max, xy, x_lens, y_lens, workspace, workspace_bytes, S(stream), slots)
               : t2s_prefill_impl<float>(h, *b, 0, nrows, l_max, xy, x_lens, y_lens, workspace, workspace_bytes, S(stream), slots);
}

int gsv_t2s_decode_hidden(gsv_t2s* h, int batch, const float* x, void* stream) {
    if (!h || !h->finalized) return fail(GSV_ERR_STATE, "handle not finalized");
    T2SBound* b = t2s_find(h, batch);
    if (!b) return fail(GSV_ERR_STATE, "no state bound for batch %d", batch);
    int rc = h->cfg.dtype == GSV_BF16 ? t2s_layers<bf16_t>(h, b->st, x, S(stream)) : t2s_layers<float>(h, b->st, x, S(stream));
    if (rc) return rc;
    return h->cfg.dtype == GSV_BF16 ? t2s_logits<bf16_t>(h, b->st, 1, nullptr, 0, batch, h->cfg.vocab, 1, S(stream))
                                    : t2s_logits<float>(h, b->st, 1, nullptr, 0, batch, h->cfg.vocab, 1, S(stream));
}

int gsv_t2s_decode(gsv_t2s* h, int batch, int n_steps, int use_graph, void* stream) {
    if (!h || !h->finalized) return fail(GSV_ERR_STATE, "handle not finalized");
    T2SBound* b = t2s_find(h, batch);
    if (!b) return fail(GSV_ERR_STATE, "no state bound for batch %d", batch);
    const bool bf = h->cfg.dtype == GSV_BF16;
    const bool graph = (use_graph & 1) != 0;
    const bool mega = (use_graph & 2) != 0 && batch <= kMegaMaxBatch && 2 * h->cfg.n_layer <= 256;
    if (!graph) {
        for (int i = 0; i < n_steps; ++i)
            if (int rc = bf ? t2s_step<bf16_t>(h, b->st, mega, S(stream)) : t2s_step<float>(h, b->st, mega, S(stream))) return rc;
        return GSV_OK;
    }
    hipGraphExec_t& exec = mega ? b->graph_mega : b->graph;
    if (!exec) {
        hipGraph_t g = nullptr;
        HIPCHK(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
        int rc = bf ? t2s_step<bf16_t>(h, b->st, mega, h->cap_stream) : t2s_step<float>(h, b->st, mega, h->cap_stream);
        hipError_t e = hipStreamEndCapture(h->cap_stream, &g);
        if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (e != hipSuccess) return fail(GSV_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
        e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) return fail(GSV_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
    }
    for (int i = 0; i < n_steps; ++i) HIPCHK(hipGraphLaunch(exec, S(stream)));
    return GSV_OK;
}

/* 1 if any persistent-step hand-off ever timed out on this handle (results are then invalid) */
int gsv_t2s_megastep_error(gsv_t2s* h) {
    if (!h || !h->mega_cnt) return 0;
    unsigned v = 0;
    if (hipMemcpy(&v, h->mega_cnt + (size_t)h->scratch_b * 2 * h->cfg.n_layer, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)v;
}

int gsv_t2s_time_kernels(gsv_t2s* h, int batch, int iters, float* out_ms, void* stream) {
    if (!h || !h->finalized || !out_ms || iters < 1) return fail(GSV_ERR_STATE, "bad call");
    T2SBound* b = t2s_find(h, batch);
    if (!b) return fail(GSV_ERR_STATE, "no state bound for batch %d", batch);
    return h->cfg.dtype == GSV_BF16 ? t2s_time_impl<bf16_t>(h, b, iters, out_ms, S(stream))
                                    : t2s_time_impl<float>(h, b, iters, out_ms, S(stream));
}

int gsv_t2s_set_debug(gsv_t2s* h, void* buf) {
    if (!h) return fail(GSV_ERR_ARG, "null handle");
    h->dbg = (unsigned long long*)buf;
    for (auto& kv : h->bound)
        { if (kv.second.graph) { (void)hipGraphExecDestroy(kv.second.graph); kv.second.graph = nullptr; }
          if (kv.second.graph_mega) { (void)hipGraphExecDestroy(kv.second.graph_mega); kv.second.graph_mega = nullptr; } }
    return GSV_OK;
}

int gsv_t2s_flush(gsv_t2s* h, int batch, void* stream) {
    if (!h || !h->finalized) return fail(GSV_ERR_STATE, "handle not finalized");
    T2SBound* b = t2s_find(h, batch);
    if (!b) return fail(GSV_ERR_STATE, "no state bound for batch %d", batch);
    return t2s_token(h, b->st, 0, S(stream));
}

}  // extern "C"

// =============================================================================================
// SoVITS flow + Generator
// =============================================================================================
struct VocFlow {
    PackedConv pre, cond, post;      // post packed NEGATED: x1 + (-(W out + b)) in the epilogue
    PackedConv in_l[4], rs_res[3], rs_skip[4];
    void* ff_w = nullptr;            // flowfuse.h weight arena (bf16 mode, hidden 192 / half 96 only)
    float* ff_b = nullptr;           // flowfuse.h bias arena
    int parity = 0;                  // 1: this layer sees the tensor channel-reversed (odd number of Flips before it)
};
struct VocResBlock {
    PackedConv c1[3], c2[3];
    int k = 3;
};
struct VocStage {
    PackedConv up;
    std::vector<VocResBlock> rb;
    int cin = 0, cout = 0, u = 1;
};

struct EncLayer {
    PackedConv qkv, o, c1, c2;
    float *relk = nullptr, *relv = nullptr, *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr;
};
struct EncP {
    bool ready = false;
    PackedConv ssl_proj, c_pre, text_pre, c_post, proj, xq, xkv, xo;
    float *text_emb = nullptr, *codebook = nullptr;
    int n_text = 0, n_code = 0;
    std::vector<EncLayer> ssl, text, enc2;
    std::vector<float*> owned;       // small fp32 tensors (norms, relative embeddings, tables)
};

struct gsv_voc {
    gsv_voc_config cfg;
    std::map<std::string, std::pair<float*, int64_t>> staged;
    bool finalized = false;
    std::vector<VocFlow> flows;
    PackedConv conv_pre, cond, conv_post;
    PackedConv cond_all;             // every flow's cond_layer stacked: one launch for the whole flow
    float* post_w = nullptr;         // conv_post weight [C][7] fp32 for conv_post_kernel
    int post_c = 0;
    bool fused_flow = false;
    EncP enc;                        // enc_p in HIP (bf16 mode, when its tensors were loaded)
    std::vector<VocStage> stages;
    int total_up = 1;
    int max_stage_elems_per_frame = 0;  // max over stages of ld(C) * time multiplier
};

namespace {

inline int ld_of(int c) { return (c + 15) / 16 * 16; }

// conditioning GEMV / small-row 1x1 convs (bf16 in, fp32 out): the latency-shaped rowgemm when the
// contraction is 512 or 1024 channels, else the generic kernel
template <typename AT>
int run_cond(const PackedConv& pc, const void* X, int ldx, int rows, float* Y, int ldy, hipStream_t st) {
    if (sizeof(AT) == 2 && pc.u == 0 && pc.k == 1 && (pc.cin == 512 || pc.cin == 1024)) {
        RowGemmArgs ra;
        ra.X = X; ra.ldx = ldx; ra.M = rows; ra.W = (const uint4*)pc.w; ra.ksteps = pc.cin / 16; ra.ntaps = 1; ra.pad = 0; ra.mtiles = pc.mtiles;
        ra.bias = pc.bias; ra.relu = 0;
        ra.Y = Y; ra.ldy = ldy; ra.split_stride = 0;
        const dim3 grid(cdiv(rows, 32), pc.mtiles, 1);
        if (pc.cin == 512) hipLaunchKernelGGL((rowgemm_kernel<bf16_t, float, 8>), grid, dim3(256), 0, st, ra);
        else hipLaunchKernelGGL((rowgemm_kernel<bf16_t, float, 16>), grid, dim3(256), 0, st, ra);
        HIPCHK(hipGetLastError());
        return GSV_OK;
    }
    Epi ec;
    return run_conv<AT, AT, float>(pc, X, ldx, rows, Y, ldy, rows, ec, st);
}

struct VocWs {
    // channels-last buffers (element type AT unless noted)
    void *zin, *zflip, *h, *outp, *a, *acts, *ge_cl;
    float *gc, *condbuf;
    void* st[11];  // stage buffers: xu, x (stage in/out), then per resblock branch {t1, xa, xb}
    size_t bytes;
};

template <typename AT>
VocWs voc_layout(const gsv_voc* v, int T, int Tg, char* base) {
    const gsv_voc_config& c = v->cfg;
    const int H = c.hidden_channels, C = c.inter_channels;
    size_t off = 0;
    auto take = [&](size_t bytes) { void* p = base ? base + off : nullptr; off += align_up(bytes, 256); return p; };
    VocWs w;
    w.zin = take(sizeof(AT) * (size_t)T * C);
    w.zflip = take(sizeof(AT) * (size_t)T * C);
    w.h = take(sizeof(AT) * (size_t)T * H);
    w.outp = take(sizeof(AT) * (size_t)T * H);
    w.a = take(sizeof(AT) * (size_t)T * 2 * H);
    w.acts = take(sizeof(AT) * (size_t)T * H);
    w.ge_cl = take(sizeof(AT) * (size_t)Tg * c.gin_channels);
    w.gc = (float*)take(sizeof(float) * (size_t)Tg * 8 * H * std::max(1, c.n_flows));
    w.condbuf = (float*)take(sizeof(float) * (size_t)Tg * c.upsample_initial_channel);
    const size_t se = (size_t)T * std::max(v->max_stage_elems_per_frame, ld_of(c.upsample_initial_channel));
    for (int i = 0; i < 11; ++i) w.st[i] = take(sizeof(AT) * se);
    w.bytes = off;
    return w;
}

template <typename AT>
int voc_flow_impl(gsv_voc* v, VocWs& w, const float* mask, int T, int Tg, hipStream_t st) {
    const gsv_voc_config& c = v->cfg;
    const int H = c.hidden_channels, C = c.inter_channels, half = C / 2;
    AT* x = (AT*)w.zin;
    AT* xf = (AT*)w.zflip;
    if (v->fused_flow && sizeof(AT) == 2) {
        // one launch for every flow's conditioning, then one fused kernel per coupling layer; no Flip passes
        const int ldg_all = 8 * H * c.n_flows;
        if (int rc = run_cond<AT>(v->cond_all, w.ge_cl, c.gin_channels, Tg, w.gc, ldg_all, st)) return rc;
        HIPCHK(hipFuncSetAttribute((const void*)flowfuse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FF_LDS_TOTAL));
        for (int f = c.n_flows - 1; f >= 0; --f) {
            VocFlow& F = v->flows[f];
            FlowFuseArgs a;
            a.P = (bf16_t*)x; a.mask = mask; a.gc = w.gc + (size_t)f * 8 * H; a.ldg = Tg == 1 ? 0 : ldg_all;
            a.W = (const uint4*)F.ff_w; a.B = F.ff_b; a.T = T;
            a.xin_off = F.parity ? half : 0; a.xup_off = F.parity ? 0 : half;
            const int nt = cdiv(T, FF_VR), nx = std::min(8, cdiv(nt, 32));
            a.per_xcd = cdiv(nt, nx);
            a.dbg = nullptr;
            static const bool ff_debug = getenv("GSV_FF_DEBUG") != nullptr;
            long long* dbg = nullptr;
            if (ff_debug) { HIPCHK(hipMalloc(&dbg, 32 * sizeof(long long))); HIPCHK(hipMemset(dbg, 0, 32 * sizeof(long long))); a.dbg = dbg; }
            hipLaunchKernelGGL(flowfuse_kernel, dim3(8 * a.per_xcd), dim3(256), FF_LDS_TOTAL, st, a);
            if (ff_debug) {
                long long h[32];
                HIPCHK(hipStreamSynchronize(st));
                HIPCHK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
                fprintf(stderr, "[flowfuse f=%d]", f);
                for (int i = 1; i < 32 && h[i]; ++i) fprintf(stderr, " %lld", h[i] - h[i - 1]);
                fprintf(stderr, "\n");
                (void)hipFree(dbg);
            }
        }
        HIPCHK(hipGetLastError());
        return GSV_OK;
    }
    const int ew = std::min(2048, cdiv(T * H, 256));
    for (int f = c.n_flows - 1; f >= 0; --f) {
        VocFlow& F = v->flows[f];
        hipLaunchKernelGGL((flip_kernel<AT>), dim3(std::min(2048, cdiv(T * C, 256))), dim3(256), 0, st, x, xf, C, T, C);
        std::swap(x, xf);  // x is now the flipped tensor
        Epi ep; ep.mask = mask;
        if (int rc = run_conv<AT, AT, AT>(F.pre, x, C, T, w.h, H, T, ep, st)) return rc;
        Epi ec;
        if (int rc = run_conv<AT, AT, float>(F.cond, w.ge_cl, c.gin_channels, Tg, w.gc, 8 * H, Tg, ec, st)) return rc;
        for (int l = 0; l < 4; ++l) {
            Epi ei; ei.add = w.gc + (size_t)l * 2 * H; ei.ld_add = Tg == 1 ? 0 : 8 * H;
            if (int rc = run_conv<AT, AT, AT>(F.in_l[l], w.h, H, T, w.a, 2 * H, T, ei, st)) return rc;
            hipLaunchKernelGGL((gate_kernel<AT>), dim3(ew), dim3(256), 0, st, (const AT*)w.a, (AT*)w.acts, H, T);
            Epi es; es.accumulate = l > 0;
            if (int rc = run_conv<AT, AT, AT>(F.rs_skip[l], w.acts, H, T, w.outp, H, T, es, st)) return rc;
            if (l < 3) {
                Epi er; er.res = w.h; er.ld_res = H; er.mask = mask;
                if (int rc = run_conv<AT, AT, AT>(F.rs_res[l], w.acts, H, T, w.h, H, T, er, st)) return rc;
            }
        }
        // x1 = (x1 - (post(out*mask)+b)*mask) * mask, binary mask; post is packed negated
        Epi eo; eo.res = x + half; eo.ld_res = C; eo.mask = mask;
        if (int rc = run_conv<AT, AT, AT>(F.post, w.outp, H, T, x + half, C, T, eo, st)) return rc;
    }
    if (x != (AT*)w.zin) HIPCHK(hipMemcpyAsync(w.zin, x, sizeof(AT) * (size_t)T * C, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

template <typename AT>
int voc_dec_impl(gsv_voc* v, VocWs& w, int T, int Tg, float* out, hipStream_t st) {
    const gsv_voc_config& c = v->cfg;
    const int C0 = c.upsample_initial_channel;
    if (int rc = run_cond<AT>(v->cond, w.ge_cl, c.gin_channels, Tg, w.condbuf, C0, st)) return rc;
    AT* x = (AT*)w.st[1];
    Epi ep; ep.add = w.condbuf; ep.ld_add = Tg == 1 ? 0 : C0;
    if (int rc = run_conv<AT, AT, AT>(v->conv_pre, w.zin, c.inter_channels, T, x, ld_of(C0), T, ep, st)) return rc;
    int Tc = T;
    AT* xu = (AT*)w.st[0];
    const int NB = (int)v->stages[0].rb.size();
    if (NB != 3) return fail(GSV_ERR_ARG, "the fused branch launch expects 3 resblock kernels per stage");
    for (size_t i = 0; i < v->stages.size(); ++i) {
        VocStage& sg = v->stages[i];
        const int ldi = ld_of(sg.cin), ldo = ld_of(sg.cout);
        const int Tn = Tc * sg.u;
        // pad channels feed zero-weight k-steps but must not hold NaN/Inf bit patterns.  The wconv path writes whole
        // rows (its pad outputs are exact zeros: zero weight rows, zero bias, zero residual), so there only the
        // transposed conv's output buffer needs clearing; the tapgemm path writes `cout` channels per row.
        const bool wc = sizeof(AT) == 2 && wconv_channels(sg.cout);
        if (ldo != sg.cout) {
            for (int q = 0; q < 11; ++q)
                if (wc ? q == 0 : q != 1) HIPCHK(hipMemsetAsync(w.st[q], 0, sizeof(AT) * (size_t)Tn * ldo, st));
        }
        Epi eu; eu.in_slope = 0.1f;
        int ru = run_wups<AT>(sg.up, x, ldi, Tc, xu, ldo, 0.1f, st);
        if (ru > 0) return ru;
        if (ru < 0)
            if (int rc = run_conv<AT, AT, AT>(sg.up, x, ldi, Tc, xu, ldo, Tc, eu, st)) return rc;
        if (ldo != sg.cout) HIPCHK(hipMemsetAsync(x, 0, sizeof(AT) * (size_t)Tn * ldo, st));
        // the three resblocks (k = 3, 7, 11) advance in lock step: one launch per conv position
        const AT* cur[3] = {xu, xu, xu};
        for (int d = 0; d < 3; ++d) {
            Branch b1[3], b2[3];
            for (int j = 0; j < 3; ++j) {
                AT* t1 = (AT*)w.st[2 + 3 * j];
                AT* dst = (AT*)w.st[2 + 3 * j + 1 + (d & 1)];
                b1[j] = Branch{&sg.rb[j].c1[d], cur[j], t1, nullptr};
                b2[j] = Branch{&sg.rb[j].c2[d], t1, dst, cur[j]};
            }
            // bf16, 16..128 channels: weights-in-registers kernel; the first conv writes lrelu(t1), which is
            // the only form its consumer reads, so the second conv stages its input without arithmetic
            int rw = run_wconv<AT>(b1, ldo, Tn, 0.1f, 0.1f, st);
            if (rw > 0) return rw;
            if (rw != 0 && wc && ldo != sg.cout) return fail(GSV_ERR_STATE, "wconv declined a padded stage whose buffers were not cleared");
            if (rw == 0) {
                rw = run_wconv<AT>(b2, ldo, Tn, 1.0f, 1.0f, st);
                if (rw > 0) return rw;
                if (rw != 0) return fail(GSV_ERR_STATE, "wconv accepted the first conv of a pair but not the second");
            } else {
                Epi e1; e1.in_slope = 0.1f;
                if (int rc = run_conv_multi<AT, AT, AT>(b1, 3, ldo, Tn, ldo, Tn, e1, st)) return rc;
                Epi e2; e2.in_slope = 0.1f; e2.ld_res = ldo;
                if (int rc = run_conv_multi<AT, AT, AT>(b2, 3, ldo, Tn, ldo, Tn, e2, st)) return rc;
            }
            for (int j = 0; j < 3; ++j) cur[j] = (const AT*)b2[j].Y;
        }
        const size_t n = (size_t)Tn * ldo;
        hipLaunchKernelGGL((avg3_kernel<AT>), dim3((unsigned)std::min<size_t>(4096, (n / 8 + 255) / 256)), dim3(256), 0, st,
                           cur[0], cur[1], cur[2], x, n);
        Tc = Tn;
    }
    if (sizeof(AT) == 2 && v->post_w && (v->post_c == 16 || v->post_c == 24)) {
        const int ldp = ld_of(v->post_c);
        if (v->post_c == 16) hipLaunchKernelGGL((conv_post_kernel<AT, 16>), dim3(cdiv(Tc, 256)), dim3(256), 0, st, (const AT*)x, ldp, (const float*)v->post_w, out, Tc);
        else hipLaunchKernelGGL((conv_post_kernel<AT, 24>), dim3(cdiv(Tc, 256)), dim3(256), 0, st, (const AT*)x, ldp, (const float*)v->post_w, out, Tc);
    } else {
        Epi eo; eo.in_slope = 0.01f; eo.act = ACT_TANH; eo.use_bias = false;
        if (int rc = run_conv<AT, AT, float>(v->conv_post, x, ld_of(v->stages.back().cout), Tc, out, 1, Tc, eo, st)) return rc;
    }
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

template <typename AT>
int voc_prepare(gsv_voc* v, VocWs& w, const float* z, const float* ge, int T, int Tg, hipStream_t st) {
    const gsv_voc_config& c = v->cfg;
    hipLaunchKernelGGL((cf_to_cl_kernel<AT>), dim3(cdiv(T, 32), cdiv(c.inter_channels, 32)), dim3(256), 0, st, z,
                       (AT*)w.zin, c.inter_channels, T, c.inter_channels);
    hipLaunchKernelGGL((cf_to_cl_kernel<AT>), dim3(cdiv(Tg, 32), cdiv(c.gin_channels, 32)), dim3(256), 0, st, ge,
                       (AT*)w.ge_cl, c.gin_channels, Tg, c.gin_channels);
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

template <typename AT>
int voc_run(gsv_voc* v, int what, const float* z, const float* mask, const float* ge, int T, int Tg, float* out,
            void* ws, size_t ws_bytes, hipStream_t st) {
    if (!v->finalized) return fail(GSV_ERR_STATE, "vocoder not finalized");
    if (T < 1 || (Tg != 1 && Tg != T)) return fail(GSV_ERR_ARG, "bad T/Tg");
    VocWs w = voc_layout<AT>(v, T, Tg, (char*)ws);
    if (ws_bytes < w.bytes) return fail(GSV_ERR_ARG, "vocoder workspace %zu < %zu", ws_bytes, w.bytes);
    if (int rc = voc_prepare<AT>(v, w, z, ge, T, Tg, st)) return rc;
    if (what & 1)
        if (int rc = voc_flow_impl<AT>(v, w, mask, T, Tg, st)) return rc;
    if (what == 1) {  // flow only: back to channels-first fp32
        hipLaunchKernelGGL((cl_to_cf_kernel<AT>), dim3(cdiv(T, 32), cdiv(v->cfg.inter_channels, 32)), dim3(256), 0, st,
                           (const AT*)w.zin, out, v->cfg.inter_channels, T, v->cfg.inter_channels);
        HIPCHK(hipGetLastError());
        return GSV_OK;
    }
    return voc_dec_impl<AT>(v, w, T, Tg, out, st);
}

// ---- enc_p (bf16): weights -------------------------------------------------------------------
int encp_finalize(gsv_voc* v, std::vector<float*>& temps, hipStream_t st) {
    EncP& E = v->enc;
    auto get = [&](const std::string& n, int64_t numel, const float** out) -> int {
        auto it = v->staged.find(n);
        if (it == v->staged.end()) return fail(GSV_ERR_STATE, "missing tensor '%s'", n.c_str());
        if (numel > 0 && it->second.second != numel) return fail(GSV_ERR_ARG, "%s: numel %lld, expected %lld", n.c_str(), (long long)it->second.second, (long long)numel);
        *out = it->second.first;
        return GSV_OK;
    };
    auto keep = [&](const std::string& n, int64_t numel, float** out) -> int {   // private fp32 copy
        const float* s;
        if (int rc = get(n, numel, &s)) return rc;
        float* d;
        HIPCHK(hipMalloc(&d, sizeof(float) * (size_t)v->staged[n].second));
        HIPCHK(hipMemcpyAsync(d, s, sizeof(float) * (size_t)v->staged[n].second, hipMemcpyDeviceToDevice, st));
        E.owned.push_back(d);
        *out = d;
        return GSV_OK;
    };
    auto conv = [&](PackedConv& pc, const std::string& base, int cout, int cin, int k) -> int {
        const float *w, *b;
        if (int rc = get(base + ".weight", (int64_t)cout * cin * k, &w)) return rc;
        if (int rc = get(base + ".bias", cout, &b)) return rc;
        return pack_conv<bf16_t>(pc, w, cout, cin, k, (int64_t)cin * k, k, 1, 1, (k - 1) / 2, 0, b, 1.f, st);
    };
    // several 1x1 convs of one input stacked along the output channels (q|k|v)
    auto stacked = [&](PackedConv& pc, const std::vector<std::string>& bases, int cout_each, int cin) -> int {
        const int n = (int)bases.size();
        float *w, *b;
        HIPCHK(hipMalloc(&w, sizeof(float) * (size_t)n * cout_each * cin));
        HIPCHK(hipMalloc(&b, sizeof(float) * (size_t)n * cout_each));
        temps.push_back(w); temps.push_back(b);
        for (int i = 0; i < n; ++i) {
            const float *ws, *bs;
            if (int rc = get(bases[i] + ".weight", (int64_t)cout_each * cin, &ws)) return rc;
            if (int rc = get(bases[i] + ".bias", cout_each, &bs)) return rc;
            HIPCHK(hipMemcpyAsync(w + (size_t)i * cout_each * cin, ws, sizeof(float) * (size_t)cout_each * cin, hipMemcpyDeviceToDevice, st));
            HIPCHK(hipMemcpyAsync(b + (size_t)i * cout_each, bs, sizeof(float) * cout_each, hipMemcpyDeviceToDevice, st));
        }
        return pack_conv<bf16_t>(pc, w, n * cout_each, cin, 1, cin, 1, 0, 1, 0, 0, b, 1.f, st);
    };
    const int Hc = v->cfg.hidden_channels;                   // 192
    const int Fc = 4 * Hc;                                   // filter channels (768)
    auto encoder = [&](std::vector<EncLayer>& Ls, const std::string& pre, int n_layers) -> int {
        Ls.resize(n_layers);
        for (int i = 0; i < n_layers; ++i) {
            EncLayer& L = Ls[i];
            const std::string a = pre + "attn_layers." + std::to_string(i) + ".";
            if (int rc = stacked(L.qkv, {a + "conv_q", a + "conv_k", a + "conv_v"}, Hc, Hc)) return rc;
            if (int rc = conv(L.o, a + "conv_o", Hc, Hc, 1)) return rc;
            if (int rc = keep(a + "emb_rel_k", 0, &L.relk)) return rc;
            if (int rc = keep(a + "emb_rel_v", 0, &L.relv)) return rc;
            if (v->staged[a + "emb_rel_k"].second != 9 * (Hc / 2)) return fail(GSV_ERR_ARG, "enc_p: expected window 4, 2 heads");
            const std::string s = std::to_string(i);
            if (int rc = keep(pre + "norm_layers_1." + s + ".gamma", Hc, &L.g1)) return rc;
            if (int rc = keep(pre + "norm_layers_1." + s + ".beta", Hc, &L.b1)) return rc;
            if (int rc = keep(pre + "norm_layers_2." + s + ".gamma", Hc, &L.g2)) return rc;
            if (int rc = keep(pre + "norm_layers_2." + s + ".beta", Hc, &L.b2)) return rc;
            auto it = v->staged.find(pre + "ffn_layers." + s + ".conv_1.weight");
            if (it == v->staged.end()) return fail(GSV_ERR_STATE, "missing enc_p ffn tensors");
            const int k = (int)(it->second.second / ((int64_t)Fc * Hc));
            if (k != 3) return fail(GSV_ERR_ARG, "enc_p: FFN kernel size %d (expected 3)", k);
            if (int rc = conv(L.c1, pre + "ffn_layers." + s + ".conv_1", Fc, Hc, k)) return rc;
            if (int rc = conv(L.c2, pre + "ffn_layers." + s + ".conv_2", Hc, Fc, k)) return rc;
        }
        return GSV_OK;
    };
    int nl = 0;
    while (v->staged.count("enc_p.encoder_text.attn_layers." + std::to_string(nl) + ".conv_q.weight")) ++nl;
    if (nl < 2 || nl % 2) return fail(GSV_ERR_ARG, "enc_p: %d text encoder layers", nl);
    if (int rc = conv(E.ssl_proj, "enc_p.ssl_proj", Hc, 768, 1)) return rc;
    if (int rc = encoder(E.ssl, "enc_p.encoder_ssl.", nl / 2)) return rc;
    if (int rc = encoder(E.text, "enc_p.encoder_text.", nl)) return rc;
    if (int rc = encoder(E.enc2, "enc_p.encoder2.", nl / 2)) return rc;
    if (int rc = keep("enc_p.text_embedding.weight", 0, &E.text_emb)) return rc;
    E.n_text = (int)(v->staged["enc_p.text_embedding.weight"].second / Hc);
    if (int rc = keep("quantizer.vq.layers.0._codebook.embed", 0, &E.codebook)) return rc;
    E.n_code = (int)(v->staged["quantizer.vq.layers.0._codebook.embed"].second / 768);
    const std::string m = "enc_p.mrte.";
    if (int rc = conv(E.c_pre, m + "c_pre", 512, Hc, 1)) return rc;
    if (int rc = conv(E.text_pre, m + "text_pre", 512, Hc, 1)) return rc;
    if (int rc = conv(E.c_post, m + "c_post", Hc, 512, 1)) return rc;
    if (int rc = conv(E.xq, m + "cross_attention.conv_q", 512, 512, 1)) return rc;
    if (int rc = stacked(E.xkv, {m + "cross_attention.conv_k", m + "cross_attention.conv_v"}, 512, 512)) return rc;
    if (int rc = conv(E.xo, m + "cross_attention.conv_o", 512, 512, 1)) return rc;
    if (int rc = conv(E.proj, "enc_p.proj", 2 * v->cfg.inter_channels, Hc, 1)) return rc;
    E.ready = true;
    return GSV_OK;
}

void encp_free(gsv_voc* v) {
    EncP& E = v->enc;
    for (PackedConv* p : {&E.ssl_proj, &E.c_pre, &E.text_pre, &E.c_post, &E.proj, &E.xq, &E.xkv, &E.xo}) free_conv(*p);
    for (auto* Ls : {&E.ssl, &E.text, &E.enc2})
        for (EncLayer& L : *Ls) { free_conv(L.qkv); free_conv(L.o); free_conv(L.c1); free_conv(L.c2); }
    for (float* p : E.owned) (void)hipFree(p);
    E.owned.clear();
    E.ready = false;
}

// ---- enc_p (bf16): run ------------------------------------------------------------------------
struct EncWs {
    bf16_t *y768, *y, *t, *qkv, *att, *tmp, *ffn, *ssl512, *text512, *xq, *xkv, *xatt, *xo, *xsum;
    float *stats, *part;
    size_t bytes;
};
EncWs encp_layout(const gsv_voc* v, int T, int P, char* base) {
    size_t off = 0;
    auto take = [&](size_t bytes) { void* p = base ? base + off : nullptr; off += align_up(bytes, 256); return p; };
    const int Hc = v->cfg.hidden_channels, R = std::max(T, P);
    EncWs w;
    w.y768 = (bf16_t*)take(2 * (size_t)T * 768);
    w.y = (bf16_t*)take(2 * (size_t)T * Hc);
    w.t = (bf16_t*)take(2 * (size_t)P * Hc);
    w.qkv = (bf16_t*)take(2 * (size_t)R * 3 * Hc);
    w.att = (bf16_t*)take(2 * (size_t)R * Hc);
    w.tmp = (bf16_t*)take(2 * (size_t)R * Hc);
    w.ffn = (bf16_t*)take(2 * (size_t)R * 4 * Hc);
    w.ssl512 = (bf16_t*)take(2 * (size_t)T * 512);
    w.text512 = (bf16_t*)take(2 * (size_t)P * 512);
    w.xq = (bf16_t*)take(2 * (size_t)T * 512);
    w.xkv = (bf16_t*)take(2 * (size_t)P * 1024);
    w.xatt = (bf16_t*)take(2 * (size_t)T * 512);
    w.xo = (bf16_t*)take(2 * (size_t)T * 512);
    w.xsum = (bf16_t*)take(2 * (size_t)T * 512);
    w.stats = (float*)take(4 * (size_t)T * 2 * v->cfg.inter_channels);
    w.part = (float*)take(4 * (size_t)3 * R * Hc);
    w.bytes = off;
    return w;
}

// dense layer of enc_p on the latency-shaped rowgemm (bf16 in; bf16 or raw fp32 split partials out)
int enc_gemm(const PackedConv& pc, const bf16_t* X, int ldx, int rows, bool with_bias, int relu, void* Y, int ldy, bool out_f32,
             int nsplit, size_t split_stride, hipStream_t st) {
    RowGemmArgs ra;
    ra.X = X; ra.ldx = ldx; ra.M = rows; ra.W = (const uint4*)pc.w; ra.ksteps = pc.cin / 16; ra.ntaps = pc.ntaps; ra.pad = pc.pad;
    ra.mtiles = pc.mtiles; ra.bias = with_bias ? pc.bias : nullptr; ra.relu = relu; ra.Y = Y; ra.ldy = ldy; ra.split_stride = split_stride;
    const int total = pc.ntaps * (pc.cin / 16);
    if (pc.u != 0 || pc.dil != 1 || total % (4 * nsplit) != 0 || pc.cout % 32 != 0) return fail(GSV_ERR_ARG, "enc_p: layer shape does not fit rowgemm");
    const int kpw = total / (4 * nsplit);
    const dim3 grid(cdiv(rows, 32), pc.mtiles, nsplit);
#define GSV_ENC_GEMM(K)                                                                                            \
    if (kpw == K) {                                                                                                  \
        if (out_f32) hipLaunchKernelGGL((rowgemm_kernel<bf16_t, float, K>), grid, dim3(256), 0, st, ra);              \
        else hipLaunchKernelGGL((rowgemm_kernel<bf16_t, bf16_t, K>), grid, dim3(256), 0, st, ra);                    \
        return GSV_OK;                                                                                               \
    }
    GSV_ENC_GEMM(3) GSV_ENC_GEMM(8) GSV_ENC_GEMM(9) GSV_ENC_GEMM(12)
#undef GSV_ENC_GEMM
    return fail(GSV_ERR_ARG, "enc_p: no rowgemm instantiation for %d k-steps per wave", kpw);
}

int encp_encoder(gsv_voc* v, std::vector<EncLayer>& Ls, bf16_t* x, int R, EncWs& w, hipStream_t st) {
    const int Hc = v->cfg.hidden_channels;
    float* part = w.part;                                    // raw fp32 tiles: [3][R][Hc]
    const size_t ps = (size_t)R * Hc;
    for (EncLayer& L : Ls) {
        if (int rc = enc_gemm(L.qkv, x, Hc, R, true, 0, w.qkv, 3 * Hc, false, 1, 0, st)) return rc;
        EncAttnArgs a;
        a.Q = w.qkv; a.ldq = 3 * Hc; a.K = w.qkv; a.ldk = 3 * Hc; a.V = w.qkv; a.ldv = 3 * Hc;
        a.qoff = 0; a.koff = Hc; a.voff = 2 * Hc; a.O = w.att; a.ldo = Hc; a.Tq = R; a.Tk = R; a.H = 2;
        a.scale = 1.0f / sqrtf((float)(Hc / 2)); a.relk = L.relk; a.relv = L.relv; a.window = 4; a.slice = nullptr; a.P = nullptr;
        hipLaunchKernelGGL(encp_attn_kernel<96>, dim3(2, cdiv(R, 32)), dim3(256), encp_attn_lds_bytes<96>(), st, a);
        if (int rc = enc_gemm(L.o, w.att, Hc, R, false, 0, part, Hc, true, 1, 0, st)) return rc;
        hipLaunchKernelGGL(encp_ln_sum_kernel, dim3(cdiv(R, 4)), dim3(256), 0, st, (const float*)part, 1, (size_t)0, (const float*)L.o.bias,
                           (const bf16_t*)x, (const float*)L.g1, (const float*)L.b1, x, R, Hc);
        if (int rc = enc_gemm(L.c1, x, Hc, R, true, 1, w.ffn, 4 * Hc, false, 1, 0, st)) return rc;
        if (int rc = enc_gemm(L.c2, w.ffn, 4 * Hc, R, false, 0, part, Hc, true, 3, ps, st)) return rc;
        hipLaunchKernelGGL(encp_ln_sum_kernel, dim3(cdiv(R, 4)), dim3(256), 0, st, (const float*)part, 3, ps, (const float*)L.c2.bias,
                           (const bf16_t*)x, (const float*)L.g2, (const float*)L.b2, x, R, Hc);
    }
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

int encp_run(gsv_voc* v, const int64_t* codes, int n_codes, const int64_t* text, int P, const float* ge512, int Tg,
             const int64_t* slice, float* m_p, float* logs_p, float* attn, void* ws, size_t ws_bytes, hipStream_t st) {
    EncP& E = v->enc;
    const int Hc = v->cfg.hidden_channels, C = v->cfg.inter_channels, T = 2 * n_codes;
    if (Hc != 192) return fail(GSV_ERR_ARG, "enc_p: hidden_channels %d (the attention kernel is built for 2 heads of 96)", Hc);
    EncWs w = encp_layout(v, T, P, (char*)ws);
    if (ws_bytes < w.bytes) return fail(GSV_ERR_ARG, "enc_p workspace %zu < %zu", ws_bytes, w.bytes);
    HIPCHK(hipFuncSetAttribute((const void*)encp_attn_kernel<96>, hipFuncAttributeMaxDynamicSharedMemorySize, encp_attn_lds_bytes<96>()));
    HIPCHK(hipFuncSetAttribute((const void*)encp_attn_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, encp_attn_lds_bytes<128>()));
    hipLaunchKernelGGL(encp_gather_kernel, dim3(T), dim3(128), 0, st, codes, n_codes, E.n_code, (const float*)E.codebook, 768, 2, w.y768);
    hipLaunchKernelGGL(encp_gather_kernel, dim3(P), dim3(96), 0, st, text, P, E.n_text, (const float*)E.text_emb, Hc, 1, w.t);
    if (int rc = enc_gemm(E.ssl_proj, w.y768, 768, T, true, 0, w.y, Hc, false, 1, 0, st)) return rc;
    if (int rc = encp_encoder(v, E.ssl, w.y, T, w, st)) return rc;
    if (int rc = encp_encoder(v, E.text, w.t, P, w, st)) return rc;
    // MRTE (mrte_model.py:20-38)
    if (int rc = enc_gemm(E.c_pre, w.y, Hc, T, true, 0, w.ssl512, 512, false, 1, 0, st)) return rc;
    if (int rc = enc_gemm(E.text_pre, w.t, Hc, P, true, 0, w.text512, 512, false, 1, 0, st)) return rc;
    if (int rc = enc_gemm(E.xq, w.ssl512, 512, T, true, 0, w.xq, 512, false, 1, 0, st)) return rc;
    if (int rc = enc_gemm(E.xkv, w.text512, 512, P, true, 0, w.xkv, 1024, false, 1, 0, st)) return rc;
    EncAttnArgs a;
    a.Q = w.xq; a.ldq = 512; a.K = w.xkv; a.ldk = 1024; a.V = w.xkv; a.ldv = 1024; a.qoff = 0; a.koff = 0; a.voff = 512;
    a.O = w.xatt; a.ldo = 512; a.Tq = T; a.Tk = P; a.H = 4; a.scale = 1.0f / sqrtf(128.0f); a.relk = nullptr; a.relv = nullptr;
    a.window = 0; a.slice = slice; a.P = attn;
    hipLaunchKernelGGL(encp_attn_kernel<128>, dim3(4, cdiv(T, 32)), dim3(256), encp_attn_lds_bytes<128>(), st, a);
    if (int rc = enc_gemm(E.xo, w.xatt, 512, T, true, 0, w.xo, 512, false, 1, 0, st)) return rc;
    hipLaunchKernelGGL(encp_add3_kernel, dim3(std::min(2048, cdiv(T * 512, 256))), dim3(256), 0, st, (const bf16_t*)w.xo, (const bf16_t*)w.ssl512, ge512,
                       Tg == 1 ? 0 : 512, w.xsum, T, 512);
    if (int rc = enc_gemm(E.c_post, w.xsum, 512, T, true, 0, w.y, Hc, false, 1, 0, st)) return rc;
    if (int rc = encp_encoder(v, E.enc2, w.y, T, w, st)) return rc;
    if (int rc = enc_gemm(E.proj, w.y, Hc, T, true, 0, w.stats, 2 * C, true, 1, 0, st)) return rc;
    hipLaunchKernelGGL((cl_to_cf_kernel<float>), dim3(cdiv(T, 32), cdiv(C, 32)), dim3(256), 0, st, (const float*)w.stats, m_p, C, T, 2 * C);
    hipLaunchKernelGGL((cl_to_cf_kernel<float>), dim3(cdiv(T, 32), cdiv(C, 32)), dim3(256), 0, st, (const float*)w.stats + C, logs_p, C, T, 2 * C);
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

template <typename CT>
int voc_finalize_impl(gsv_voc* v, hipStream_t st) {
    const gsv_voc_config& c = v->cfg;
    const int H = c.hidden_channels, C = c.inter_channels, half = C / 2, gin = c.gin_channels;
    auto get = [&](const std::string& n, int64_t numel, const float** out) -> int {
        auto it = v->staged.find(n);
        if (it == v->staged.end()) return fail(GSV_ERR_STATE, "missing tensor '%s'", n.c_str());
        if (it->second.second != numel) return fail(GSV_ERR_ARG, "%s: numel %lld, expected %lld", n.c_str(), (long long)it->second.second, (long long)numel);
        *out = it->second.first;
        return GSV_OK;
    };
    std::vector<float*> temps;
    auto folded = [&](const std::string& base, int rows, int row_elems, float sign, const float** out) -> int {
        const float *g, *vv;
        if (int rc = get(base + ".weight_g", rows, &g)) return rc;
        if (int rc = get(base + ".weight_v", (int64_t)rows * row_elems, &vv)) return rc;
        float* wbuf;
        HIPCHK(hipMalloc(&wbuf, sizeof(float) * (size_t)rows * row_elems));
        temps.push_back(wbuf);
        hipLaunchKernelGGL(weight_norm_fold_kernel, dim3(rows), dim3(256), 0, st, g, vv, wbuf, row_elems, sign);
        *out = wbuf;
        return GSV_OK;
    };
    int rc = GSV_OK;
    v->flows.resize(c.n_flows);
    // the fused coupling-layer kernel (flowfuse.h): bf16, hidden 192, 96 + 96 channels, even flow count
    const bool fuse = sizeof(CT) == 2 && H == FF_H && half == FF_HALF && c.n_flows % 2 == 0 && c.n_flows > 0;
    float* cond_w_all = nullptr;
    float* cond_b_all = nullptr;
    float* skip_b = nullptr;   // the four skip biases of the layer being packed (stream-ordered reuse)
    if (fuse) {
        HIPCHK(hipMalloc(&cond_w_all, sizeof(float) * (size_t)c.n_flows * 8 * H * gin));
        HIPCHK(hipMalloc(&cond_b_all, sizeof(float) * (size_t)c.n_flows * 8 * H));
        temps.push_back(cond_w_all); temps.push_back(cond_b_all);
        HIPCHK(hipMalloc(&skip_b, sizeof(float) * 4 * FF_H));
        temps.push_back(skip_b);
    }
    // pack one conv of a fused layer into its weight arena at fragment offset `frag`
    auto ff_pack = [&](VocFlow& F, int frag, const float* src, int cout, int cin, int k, int64_t sm, int64_t sc, int64_t sk, int pad) {
        const int mt = cdiv(cout, 32);
        const size_t elems = (size_t)k * mt * (cin / 16) * 64 * 8;
        hipLaunchKernelGGL((tapgemm_pack_kernel<bf16_t>), dim3((unsigned)std::min<size_t>(2048, (elems + 255) / 256)), dim3(256), 0, st,
                           src, (bf16_t*)F.ff_w + (size_t)frag * 512, cout, cin, k, sm, sc, sk, 1, k, 0, pad, mt);
    };
    auto ff_bias = [&](VocFlow& F, int off, const float* src, int n, float scale, bool reverse) {
        hipLaunchKernelGGL(scale_copy_rev_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, src, F.ff_b + off, n, scale, reverse ? 1 : 0);
    };
    for (int f = 0; f < c.n_flows && !rc; ++f) {
        VocFlow& F = v->flows[f];
        const std::string p = "flow.flows." + std::to_string(2 * f) + ".";
        const float *w, *b;
        F.parity = (c.n_flows - f) % 2;
        if (fuse) {
            if (!F.ff_w) HIPCHK(hipMalloc(&F.ff_w, (size_t)FF_W_TOTAL * 1024));
            if (!F.ff_b) HIPCHK(hipMalloc(&F.ff_b, sizeof(float) * FF_T_TOTAL));
        }
        if ((rc = get(p + "pre.weight", (int64_t)H * half, &w)) || (rc = get(p + "pre.bias", H, &b))) break;
        if ((rc = pack_conv<CT>(F.pre, w, H, half, 1, half, 1, 0, 1, 0, 0, b, 1.f, st))) break;
        if (fuse) {   // parity 1: the conv-input half is stored channel-reversed
            ff_pack(F, FF_W_PRE, F.parity ? w + (half - 1) : w, H, half, 1, half, F.parity ? -1 : 1, 0, 0);
            ff_bias(F, FF_T_PRE, b, H, 1.f, false);
        }
        if ((rc = folded(p + "enc.cond_layer", 8 * H, gin, 1.f, &w)) || (rc = get(p + "enc.cond_layer.bias", 8 * H, &b))) break;
        if ((rc = pack_conv<CT>(F.cond, w, 8 * H, gin, 1, gin, 1, 0, 1, 0, 0, b, 1.f, st))) break;
        if (fuse) {
            HIPCHK(hipMemcpyAsync(cond_w_all + (size_t)f * 8 * H * gin, w, sizeof(float) * (size_t)8 * H * gin, hipMemcpyDeviceToDevice, st));
            HIPCHK(hipMemcpyAsync(cond_b_all + (size_t)f * 8 * H, b, sizeof(float) * 8 * H, hipMemcpyDeviceToDevice, st));
        }
        for (int l = 0; l < 4 && !rc; ++l) {
            const std::string il = p + "enc.in_layers." + std::to_string(l), rl = p + "enc.res_skip_layers." + std::to_string(l);
            if ((rc = folded(il, 2 * H, H * 5, 1.f, &w)) || (rc = get(il + ".bias", 2 * H, &b))) break;
            if ((rc = pack_conv<CT>(F.in_l[l], w, 2 * H, H, 5, (int64_t)H * 5, 5, 1, 1, 2, 0, b, 1.f, st))) break;
            if (fuse) {
                ff_pack(F, FF_W_IN + l * FF_W_IN_L, w, 2 * H, H, 5, (int64_t)H * 5, 5, 1, 2);
                ff_bias(F, FF_T_IN + l * 384, b, 2 * H, 1.f, false);
            }
            const int R = l < 3 ? 2 * H : H;
            if ((rc = folded(rl, R, H, 1.f, &w)) || (rc = get(rl + ".bias", R, &b))) break;
            if (l < 3) {
                if ((rc = pack_conv<CT>(F.rs_res[l], w, H, H, 1, H, 1, 0, 1, 0, 0, b, 1.f, st))) break;
                if ((rc = pack_conv<CT>(F.rs_skip[l], w + (size_t)H * H, H, H, 1, H, 1, 0, 1, 0, 0, b + H, 1.f, st))) break;
            } else {
                if ((rc = pack_conv<CT>(F.rs_skip[l], w, H, H, 1, H, 1, 0, 1, 0, 0, b, 1.f, st))) break;
            }
            if (fuse) {
                if (l < 3) {
                    ff_pack(F, FF_W_RES + l * 6 * FF_KSH, w, H, H, 1, H, 1, 0, 0);
                    ff_bias(F, FF_T_RES + l * 192, b, H, 1.f, false);
                }
                ff_pack(F, FF_W_SKIP + l * 6 * FF_KSH, l < 3 ? w + (size_t)H * H : w, H, H, 1, H, 1, 0, 0);
                hipLaunchKernelGGL(scale_copy_rev_kernel, dim3(1), dim3(256), 0, st, l < 3 ? b + H : b, skip_b + l * 192, H, 1.f, 0);
                if (l == 3) hipLaunchKernelGGL(sum4_kernel, dim3(1), dim3(256), 0, st, (const float*)skip_b, F.ff_b + FF_T_SKIP, H);
            }
        }
        if (rc) break;
        if ((rc = get(p + "post.weight", (int64_t)half * H, &w)) || (rc = get(p + "post.bias", half, &b))) break;
        float* neg;
        HIPCHK(hipMalloc(&neg, sizeof(float) * half * H));
        temps.push_back(neg);
        hipLaunchKernelGGL(scale_copy_kernel, dim3(cdiv(half * H, 256)), dim3(256), 0, st, w, neg, (size_t)half * H, -1.0f);
        if ((rc = pack_conv<CT>(F.post, neg, half, H, 1, H, 1, 0, 1, 0, 0, b, -1.f, st))) break;
        if (fuse) {   // parity 1: the updated half is stored channel-reversed -> reversed output rows and bias
            ff_pack(F, FF_W_POST, F.parity ? neg + (size_t)(half - 1) * H : neg, half, H, 1, F.parity ? -(int64_t)H : (int64_t)H, 1, 0, 0);
            ff_bias(F, FF_T_POST, b, half, -1.f, F.parity != 0);
        }
    }
    if (!rc && fuse) {
        free_conv(v->cond_all);
        rc = pack_conv<CT>(v->cond_all, cond_w_all, c.n_flows * 8 * H, gin, 1, gin, 1, 0, 1, 0, 0, cond_b_all, 1.f, st);
    }
    v->fused_flow = fuse && !rc;
    const int C0 = c.upsample_initial_channel;
    const float *w = nullptr, *b = nullptr;
    if (!rc) rc = get("dec.conv_pre.weight", (int64_t)C0 * C * 7, &w);
    if (!rc) rc = get("dec.conv_pre.bias", C0, &b);
    if (!rc) rc = pack_conv<CT>(v->conv_pre, w, C0, C, 7, (int64_t)C * 7, 7, 1, 1, 3, 0, b, 1.f, st);
    if (!rc) rc = get("dec.cond.weight", (int64_t)C0 * gin, &w);
    if (!rc) rc = get("dec.cond.bias", C0, &b);
    if (!rc) rc = pack_conv<CT>(v->cond, w, C0, gin, 1, gin, 1, 0, 1, 0, 0, b, 1.f, st);
    v->stages.resize(c.n_upsample);
    int ch = C0, tm = 1;
    v->max_stage_elems_per_frame = ld_of(C0);
    constexpr int KS = MfmaK<CT>::KS;
    for (int i = 0; i < c.n_upsample && !rc; ++i) {
        VocStage& sg = v->stages[i];
        const int u = c.upsample_rates[i], k = c.upsample_kernel_sizes[i], co = ch / 2;
        sg.cin = ch; sg.cout = co; sg.u = u;
        tm *= u;
        v->max_stage_elems_per_frame = std::max(v->max_stage_elems_per_frame, ld_of(co) * tm);
        const std::string un = "dec.ups." + std::to_string(i);
        if ((rc = get(un + ".weight", (int64_t)ch * co * k, &w)) || (rc = get(un + ".bias", co, &b))) break;
        // ConvTranspose1d weight [Cin][Cout][k]: element (m=co, c=ci, kk) at ci*(Cout*k) + co*k + kk
        const int cin_pad = (ch + KS - 1) / KS * KS;
        if (cin_pad != ch) { rc = fail(GSV_ERR_ARG, "stage %d: %d input channels not a multiple of %d", i, ch, KS); break; }
        if ((rc = pack_conv<CT>(sg.up, w, co, ch, k, k, (int64_t)co * k, 1, 1, (k - u) / 2, u, b, 1.f, st))) break;
        sg.rb.resize(c.n_resblock_kernels);
        const int cpad = (co + KS - 1) / KS * KS;  // contraction over zero-padded channels when co % KS != 0
        for (int j = 0; j < c.n_resblock_kernels && !rc; ++j) {
            VocResBlock& rb = sg.rb[j];
            rb.k = c.resblock_kernel_sizes[j];
            const std::string rn = "dec.resblocks." + std::to_string(i * c.n_resblock_kernels + j);
            for (int d = 0; d < 3 && !rc; ++d) {
                for (int which = 0; which < 2 && !rc; ++which) {
                    const std::string cn = rn + (which ? ".convs2." : ".convs1.") + std::to_string(d);
                    if ((rc = get(cn + ".weight", (int64_t)co * co * rb.k, &w)) || (rc = get(cn + ".bias", co, &b))) break;
                    const float* wsrc = w;
                    if (cpad != co) {  // re-lay as [co][cpad][k] with zero channels
                        float* padded;
                        HIPCHK(hipMalloc(&padded, sizeof(float) * (size_t)co * cpad * rb.k));
                        temps.push_back(padded);
                        HIPCHK(hipMemsetAsync(padded, 0, sizeof(float) * (size_t)co * cpad * rb.k, st));
                        HIPCHK(hipMemcpy2DAsync(padded, sizeof(float) * cpad * rb.k, w, sizeof(float) * co * rb.k,
                                                sizeof(float) * co * rb.k, co, hipMemcpyDeviceToDevice, st));
                        wsrc = padded;
                    }
                    const int dil = which ? 1 : c.resblock_dilations[d];
                    PackedConv& pc = which ? rb.c2[d] : rb.c1[d];
                    rc = pack_conv<CT>(pc, wsrc, co, cpad, rb.k, (int64_t)cpad * rb.k, rb.k, 1, dil, dil * (rb.k - 1) / 2, 0, b, 1.f, st);
                }
            }
        }
        ch = co;
    }
    v->total_up = tm;
    if (!rc) rc = get("dec.conv_post.weight", (int64_t)ch * 7, &w);
    if (!rc) {
        const int cpad = (ch + KS - 1) / KS * KS;
        const float* wsrc = w;
        if (cpad != ch) {
            float* padded;
            HIPCHK(hipMalloc(&padded, sizeof(float) * (size_t)cpad * 7));
            temps.push_back(padded);
            HIPCHK(hipMemsetAsync(padded, 0, sizeof(float) * (size_t)cpad * 7, st));
            HIPCHK(hipMemcpyAsync(padded, w, sizeof(float) * (size_t)ch * 7, hipMemcpyDeviceToDevice, st));
            wsrc = padded;
        }
        rc = pack_conv<CT>(v->conv_post, wsrc, 1, cpad, 7, (int64_t)cpad * 7, 7, 1, 1, 3, 0, nullptr, 1.f, st);
        if (!rc) {   // plain fp32 copy [ch][7] for the one-output-channel tail kernel
            if (v->post_w) (void)hipFree(v->post_w);
            HIPCHK(hipMalloc(&v->post_w, sizeof(float) * (size_t)ch * 7));
            HIPCHK(hipMemcpyAsync(v->post_w, w, sizeof(float) * (size_t)ch * 7, hipMemcpyDeviceToDevice, st));
            v->post_c = ch;
        }
    }
    if (!rc && sizeof(CT) == 2 && v->staged.count("enc_p.ssl_proj.weight")) rc = encp_finalize(v, temps, st);
    (void)hipStreamSynchronize(st);
    for (float* t : temps) (void)hipFree(t);
    if (rc) return rc;
    for (auto& kv : v->staged) (void)hipFree(kv.second.first);
    v->staged.clear();
    v->finalized = true;
    return GSV_OK;
}

void voc_free(gsv_voc* v) {
    for (auto& kv : v->staged) (void)hipFree(kv.second.first);
    v->staged.clear();
    for (VocFlow& F : v->flows) {
        free_conv(F.pre); free_conv(F.cond); free_conv(F.post);
        if (F.ff_w) (void)hipFree(F.ff_w);
        if (F.ff_b) (void)hipFree(F.ff_b);
        F.ff_w = nullptr; F.ff_b = nullptr;
        for (auto& p : F.in_l) free_conv(p);
        for (auto& p : F.rs_res) free_conv(p);
        for (auto& p : F.rs_skip) free_conv(p);
    }
    free_conv(v->conv_pre); free_conv(v->cond); free_conv(v->conv_post); free_conv(v->cond_all);
    encp_free(v);
    if (v->post_w) (void)hipFree(v->post_w);
    v->post_w = nullptr;
    for (VocStage& s : v->stages) {
        free_conv(s.up);
        for (VocResBlock& r : s.rb)
            for (int d = 0; d < 3; ++d) { free_conv(r.c1[d]); free_conv(r.c2[d]); }
    }
}

}  // namespace

extern "C" {

int gsv_voc_create(const gsv_voc_config* cfg, gsv_voc** out) {
    if (!cfg || !out) return fail(GSV_ERR_ARG, "null argument");
    if (cfg->n_upsample < 1 || cfg->n_upsample > 8 || cfg->n_resblock_kernels < 1 || cfg->n_resblock_kernels > 4 ||
        cfg->n_flows < 1 || cfg->inter_channels % 32 != 0 || cfg->hidden_channels % 16 != 0 || cfg->gin_channels % 16 != 0 ||
        cfg->upsample_initial_channel % 32 != 0)
        return fail(GSV_ERR_ARG, "unsupported vocoder configuration");
    for (int i = 0; i < cfg->n_upsample; ++i)
        if (cfg->upsample_rates[i] < 1 || cfg->upsample_rates[i] > 10 || (cfg->upsample_kernel_sizes[i] - cfg->upsample_rates[i]) % 2 != 0)
            return fail(GSV_ERR_ARG, "unsupported upsample stage %d", i);
    if (cfg->dtype != GSV_F32 && cfg->dtype != GSV_BF16) return fail(GSV_ERR_ARG, "bad dtype");
    gsv_voc* v = new gsv_voc();
    v->cfg = *cfg;
    *out = v;
    return GSV_OK;
}

int gsv_voc_destroy(gsv_voc* v) {
    if (!v) return GSV_OK;
    (void)hipDeviceSynchronize();
    voc_free(v);
    delete v;
    return GSV_OK;
}

int gsv_voc_load_tensor(gsv_voc* v, const char* name, const float* data, int64_t numel, void* stream) {
    if (!v || !name || !data || numel < 1) return fail(GSV_ERR_ARG, "null argument");
    if (v->finalized) return fail(GSV_ERR_STATE, "vocoder already finalized");
    std::string n(name);
    if (n.compare(0, 4, "dec.") != 0 && n.compare(0, 5, "flow.") != 0 && n.compare(0, 6, "enc_p.") != 0 && n.compare(0, 10, "quantizer.") != 0)
        return fail(GSV_ERR_ARG, "tensor '%s' is not part of flow / dec / enc_p / quantizer", name);
    auto it = v->staged.find(n);
    if (it != v->staged.end()) { (void)hipFree(it->second.first); v->staged.erase(it); }
    float* p;
    HIPCHK(hipMalloc(&p, sizeof(float) * numel));
    HIPCHK(hipMemcpyAsync(p, data, sizeof(float) * numel, hipMemcpyDeviceToDevice, S(stream)));
    v->staged[n] = {p, numel};
    return GSV_OK;
}

int gsv_voc_finalize(gsv_voc* v, void* stream) {
    if (!v) return fail(GSV_ERR_ARG, "null handle");
    if (v->finalized) return GSV_OK;
    return v->cfg.dtype == GSV_BF16 ? voc_finalize_impl<bf16_t>(v, S(stream)) : voc_finalize_impl<float>(v, S(stream));
}

int gsv_voc_has_enc_p(gsv_voc* v) { return v && v->finalized && v->enc.ready ? 1 : 0; }

size_t gsv_voc_enc_workspace(gsv_voc* v, int n_codes, int n_text) {
    if (!v || !v->finalized || !v->enc.ready || n_codes < 1 || n_text < 1) return 0;
    return encp_layout(v, 2 * n_codes, n_text, nullptr).bytes;
}

int gsv_voc_enc_p(gsv_voc* v, const int64_t* codes, int n_codes, const int64_t* text, int n_text, const float* ge512, int Tg,
                  const int64_t* slice_indices, float* m_p, float* logs_p, float* attn, void* workspace, size_t workspace_bytes,
                  void* stream) {
    if (!v || !v->finalized) return fail(GSV_ERR_STATE, "vocoder not finalized");
    if (!v->enc.ready) return fail(GSV_ERR_STATE, "enc_p tensors were not loaded (or the handle is not bf16)");
    if (!codes || !text || !ge512 || !m_p || !logs_p || !workspace) return fail(GSV_ERR_ARG, "null argument");
    if (n_codes < 1 || n_text < 1 || (Tg != 1 && Tg != 2 * n_codes)) return fail(GSV_ERR_ARG, "enc_p: bad lengths");
    return encp_run(v, codes, n_codes, text, n_text, ge512, Tg, slice_indices, m_p, logs_p, attn, workspace, workspace_bytes, S(stream));
}

size_t gsv_voc_workspace(gsv_voc* v, int T) {
    if (!v || !v->finalized || T < 1) return 0;
    return v->cfg.dtype == GSV_BF16 ? voc_layout<bf16_t>(v, T, T, nullptr).bytes : voc_layout<float>(v, T, T, nullptr).bytes;
}

int gsv_voc_flow_dec(gsv_voc* v, const float* z_p, const float* y_mask, const float* ge, int T, int Tg, float* out,
                     void* workspace, size_t workspace_bytes, void* stream) {
    if (!v || !z_p || !y_mask || !ge || !out || !workspace) return fail(GSV_ERR_ARG, "null argument");
    return v->cfg.dtype == GSV_BF16 ? voc_run<bf16_t>(v, 3, z_p, y_mask, ge, T, Tg, out, workspace, workspace_bytes, S(stream))
                                    : voc_run<float>(v, 3, z_p, y_mask, ge, T, Tg, out, workspace, workspace_bytes, S(stream));
}

int gsv_voc_flow(gsv_voc* v, const float* z_p, const float* y_mask, const float* ge, int T, int Tg, float* z_out,
                 void* workspace, size_t workspace_bytes, void* stream) {
    if (!v || !z_p || !y_mask || !ge || !z_out || !workspace) return fail(GSV_ERR_ARG, "null argument");
    return v->cfg.dtype == GSV_BF16 ? voc_run<bf16_t>(v, 1, z_p, y_mask, ge, T, Tg, z_out, workspace, workspace_bytes, S(stream))
                                    : voc_run<float>(v, 1, z_p, y_mask, ge, T, Tg, z_out, workspace, workspace_bytes, S(stream));
}

int gsv_voc_dec(gsv_voc* v, const float* z, const float* ge, int T, int Tg, float* out, void* workspace,
                size_t workspace_bytes, void* stream) {
    if (!v || !z || !ge || !out || !workspace) return fail(GSV_ERR_ARG, "null argument");
    return v->cfg.dtype == GSV_BF16 ? voc_run<bf16_t>(v, 2, z, nullptr, ge, T, Tg, out, workspace, workspace_bytes, S(stream))
                                    : voc_run<float>(v, 2, z, nullptr, ge, T, Tg, out, workspace, workspace_bytes, S(stream));
}

}  // extern "C"
