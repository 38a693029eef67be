// C ABI of the MI355X GPT-SoVITS hot path (include/gsv_tts_hip.h), SoVITS part: flow + Generator (tapgemm / wconv /
// wups / flowfuse) and enc_p.  Workspaces come from the caller; nothing is allocated inside a pass.
#include <tuple>

#include "voc_launch.h"
#include "flowfuse.h"
#include "flowstage.h"
#include "rbfuse.h"
#include "encp.h"
#include "voc_kernels.h"

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
    // rbfuse.h (bf16 stages of <= 32 channels with the 3 / 7 / 11 resblocks): the stage's 18 convs as A fragments + biases
    void* rb_w = nullptr;
    float* rb_b = nullptr;
    int rb_c = 0;            // 16 or 32 (24 channels run padded to 32), 0 = not fused
    int rb_wofs[3] = {0, 0, 0};
};

struct EncLayer {
    PackedConv qkv, o, c1, c2;
    float *relk = nullptr, *relv = nullptr, *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr;
};
struct EncP {
    bool ready = false;
    PackedConv ssl_proj, c_pre, text_pre, c_post, proj, xq, xkv, xo;
    PackedConv ge512;                // ge_to512 (v2Pro / v2ProPlus, models.py:394): Linear gin -> 512 on the speaker embedding
    bool has_ge512 = false;
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
    void* dma_zero = nullptr;        // wdma.h: a zero page (rows outside the sequence) and a sink (stores of rows beyond it)
    void* dma_sink = nullptr;
    int* cg_flag = nullptr;          // cgemm.h K split: hand-off flags of the partial tiles (zero between launches; a handle runs one pass at a time)
    int post_c = 0;
    bool fused_flow = false;
    EncP enc;                        // enc_p in HIP (bf16 mode, when its tensors were loaded)
    std::vector<VocStage> stages;
    int total_up = 1;
    int max_stage_elems_per_frame = 0;  // max over stages of ld(C) * time multiplier
    // one captured flow + Generator pass per (static buffers, T): the reference's per-bucket CUDA graphs (models.py:322-369)
    struct GraphKey {
        const void *z, *m, *g, *o, *w; int T, Tg;
        bool operator<(const GraphKey& r) const {
            return std::tie(z, m, g, o, w, T, Tg) < std::tie(r.z, r.m, r.g, r.o, r.w, r.T, r.Tg);
        }
    };
    struct GraphEntry { hipGraphExec_t exec; unsigned long long last_use; };
    std::map<GraphKey, GraphEntry> graphs;      // at most GSV_VOC_MAX_GRAPHS; the least recently replayed one makes room
    unsigned long long graph_clock = 0;
    hipStream_t cap_stream = nullptr;
};

namespace {

inline int ld_of(int c) { return (c + 15) / 16 * 16; }

// conditioning GEMV / small-row 1x1 convs (bf16 in, fp32 out): the latency-shaped rowgemm when the
// contraction is 512 or 1024 channels, else the generic kernel
template <typename AT>
int run_cond(const PackedConv& pc, const void* X, int ldx, int rows, float* Y, int ldy, hipStream_t st, const int* rows_dev = nullptr) {
    if (sizeof(AT) == 2 && pc.u == 0 && pc.k == 1 && (pc.cin == 512 || pc.cin == 1024)) {
        RowGemmArgs ra;
        ra.X = X; ra.ldx = ldx; ra.M = rows; ra.W = (const uint4*)pc.w; ra.ksteps = pc.cin / 16; ra.ntaps = 1; ra.pad = 0; ra.mtiles = pc.mtiles;
        ra.bias = pc.bias; ra.relu = 0;
        ra.Y = Y; ra.ldy = ldy; ra.split_stride = 0; ra.m_dev = rows_dev;
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
    int *seg_flag, *seg_id, *seg_first, *nseg;   // per-frame ge with few distinct columns (voc_kernels.h); seg = null: one row per frame
    const int* seg;
    float* cg_part;   // cgemm.h K split: partial tiles (null on handles without a cgemm stage)
    void* st[15];  // stage buffers: xu, x (stage in/out), then per resblock branch {t1, xa, xb}; [11] lrelu(xu), [12..14] lrelu of a branch's state (wdma.h)
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
    w.seg_flag = (int*)take(sizeof(int) * (size_t)Tg);
    w.seg_id = (int*)take(sizeof(int) * (size_t)Tg);
    w.seg_first = (int*)take(sizeof(int) * (size_t)Tg);
    w.nseg = (int*)take(sizeof(int) * 64);
    w.seg = nullptr;
    const size_t se = (size_t)T * std::max(v->max_stage_elems_per_frame, ld_of(c.upsample_initial_channel));
    for (int i = 0; i < 15; ++i) w.st[i] = (i < 11 || sizeof(AT) == 2) ? take(sizeof(AT) * se) : nullptr;   // [11..14]: wdma's activated copies, bf16 handles only
    w.cg_part = (sizeof(AT) == 2 && v->cg_flag) ? (float*)take(kCgPartBytes) : nullptr;
    w.bytes = off;
    return w;
}

constexpr int kFlowMergedMaxT = 1024;   // frames up to which the staged flow runs in its merged (5 launches per layer) form
constexpr int kFlowStagedMaxT = 2048;   // frames up to which the flow runs as staged launches (measured: profiles/r04_flow_staged.txt)

template <typename AT>
int voc_flow_impl(gsv_voc* v, VocWs& w, const float* mask, int T, int Tg, hipStream_t st) {
    const gsv_voc_config& c = v->cfg;
    const int H = c.hidden_channels, C = c.inter_channels, half = C / 2;
    AT* x = (AT*)w.zin;
    AT* xf = (AT*)w.zflip;
    if (v->fused_flow && sizeof(AT) == 2) {
        // one launch for every flow's conditioning, then one fused kernel per coupling layer; no Flip passes
        const int ldg_all = 8 * H * c.n_flows;
        if (int rc = run_cond<AT>(v->cond_all, w.ge_cl, c.gin_channels, Tg, w.gc, ldg_all, st, w.seg ? w.nseg : nullptr)) return rc;
        // few frames: ten short many-CU launches per coupling layer (flowstage.h); many frames: one kernel per layer (flowfuse.h)
        static const int staged_max_T = getenv("GSV_FLOW_STAGED_MAX_T") ? atoi(getenv("GSV_FLOW_STAGED_MAX_T")) : kFlowStagedMaxT;
        if (T <= staged_max_T) {
            static const int rpb_env = getenv("GSV_FLOW_STAGED_RPB") ? atoi(getenv("GSV_FLOW_STAGED_RPB")) : 0;
            // merged form (5 launches per layer, each fatter): 166 vs 173 us up to ~1000 frames, slower beyond (249 vs 218 us at 2000)
            static const int merged_max_T = getenv("GSV_FLOW_MERGED_MAX_T") ? atoi(getenv("GSV_FLOW_MERGED_MAX_T")) : kFlowMergedMaxT;
            const bool merged = T <= merged_max_T;
            auto stage_args = [&](int f) {
                VocFlow& F = v->flows[f];
                FlowStageArgs a;
                a.P = (bf16_t*)x; a.mask = mask; a.gc = w.gc + (size_t)f * 8 * H; a.ldg = Tg == 1 ? 0 : ldg_all; a.seg = w.seg;
                a.W = (const uint4*)F.ff_w; a.B = F.ff_b; a.T = T;
                a.xin_off = F.parity ? half : 0; a.xup_off = F.parity ? 0 : half;
                a.h = (bf16_t*)w.h; a.acts = (bf16_t*)w.acts; a.skip = (float*)w.a; a.outp = (bf16_t*)w.outp;
                a.l = 0; a.rpb = 1;
                return a;
            };
            if (merged) {
                // five launches per coupling layer (flowstage.h, merged form): h ping-pongs between w.h and w.outp, acts between w.acts and w.zflip
                bf16_t* hb[2] = {(bf16_t*)w.h, (bf16_t*)w.outp};
                bf16_t* ab[2] = {(bf16_t*)w.acts, (bf16_t*)w.zflip};
                const int gin_ = cdiv(T, FM_VR), gt = cdiv(T, FS_ROWS);
                {
                    FlowStageArgs a = stage_args(c.n_flows - 1);
                    hipLaunchKernelGGL((flowstage_kernel<FS_PRE>), dim3(gt, 2), dim3(256), 0, st, a);
                }
                for (int f = c.n_flows - 1; f >= 0; --f) {
                    FlowMergeArgs m;
                    m.s = stage_args(f);
                    m.Wn = nullptr; m.Bn = nullptr;
                    for (int l = 0; l < 4; ++l) {
                        m.s.l = l;
                        // h_l lives in hb[l & 1] (h_0 from pre in hb[0]); acts_l in ab[l & 1]
                        m.h_in = hb[l == 0 ? 0 : (l - 1) & 1]; m.h_out = hb[l & 1];
                        m.acts_in = ab[l == 0 ? 0 : (l - 1) & 1]; m.acts_out = ab[l & 1];
                        hipLaunchKernelGGL(flowmerge_in_kernel, dim3(gin_, 6), dim3(256), 0, st, m);
                    }
                    m.acts_in = ab[1];                     // acts_3
                    m.h_out = hb[0];
                    if (f > 0) { m.Wn = (const uint4*)v->flows[f - 1].ff_w; m.Bn = v->flows[f - 1].ff_b; }
                    hipLaunchKernelGGL(flowmerge_tail_kernel, dim3(gt), dim3(256), 0, st, m);
                }
                HIPCHK(hipGetLastError());
                return GSV_OK;
            }
            const int ntiles = cdiv(T, FS_ROWS);
            for (int f = c.n_flows - 1; f >= 0; --f) {
                FlowStageArgs a = stage_args(f);
                a.rpb = rpb_env > 0 ? rpb_env : std::max(1, cdiv(ntiles, 64));      // <= 64 frame groups: (64 x 6) in_layer blocks fill the chip
                const int gx = cdiv(ntiles, a.rpb);
                hipLaunchKernelGGL((flowstage_kernel<FS_PRE>), dim3(gx, 2), dim3(256), 0, st, a);
                for (int l = 0; l < 4; ++l) {
                    a.l = l;
                    hipLaunchKernelGGL((flowstage_kernel<FS_IN>), dim3(gx, 6), dim3(256), 0, st, a);
                    hipLaunchKernelGGL((flowstage_kernel<FS_RS>), dim3(gx, l < 3 ? 3 : 2), dim3(256), 0, st, a);
                }
                hipLaunchKernelGGL((flowstage_kernel<FS_POST>), dim3(gx, 1), dim3(256), 0, st, a);
            }
            HIPCHK(hipGetLastError());
            return GSV_OK;
        }
        HIPCHK(hipFuncSetAttribute((const void*)flowfuse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FF_LDS_TOTAL));
        for (int f = c.n_flows - 1; f >= 0; --f) {
            VocFlow& F = v->flows[f];
            FlowFuseArgs a;
            a.P = (bf16_t*)x; a.mask = mask; a.gc = w.gc + (size_t)f * 8 * H; a.ldg = Tg == 1 ? 0 : ldg_all; a.seg = w.seg;
            a.W = (const uint4*)F.ff_w; a.B = F.ff_b; a.T = T;
            a.xin_off = F.parity ? half : 0; a.xup_off = F.parity ? 0 : half;
            const int nt = cdiv(T, FF_VR), nx = std::min(8, cdiv(nt, 32));
            a.per_xcd = cdiv(nt, nx);
            a.dbg = nullptr;
            static const bool ff_norot = getenv("GSV_FF_NOROT") != nullptr;
            a.rot_k = ff_norot ? 0 : 1;
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
            Epi ei; ei.add = w.gc + (size_t)l * 2 * H; ei.ld_add = Tg == 1 ? 0 : 8 * H; ei.add_index = Tg == 1 ? nullptr : w.seg;
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
    if (int rc = run_cond<AT>(v->cond, w.ge_cl, c.gin_channels, Tg, w.condbuf, C0, st, w.seg ? w.nseg : nullptr)) return rc;
    AT* x = (AT*)w.st[1];
    Epi ep; ep.add = w.condbuf; ep.ld_add = Tg == 1 ? 0 : C0; ep.add_index = Tg == 1 ? nullptr : w.seg;
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
        // 64 / 128 / 256 channels: the resblock convs stage their rows by LDS-DMA (wdma.h), so the state x of a branch travels with its
        // activated copy lrelu(x) -- written by whoever writes x (the transposed conv here, the second conv of a pair below)
        bool dma = sizeof(AT) == 2 && v->dma_zero && wdma_shape(sg.cout, ldo, Tn) && !(sg.rb_c && ldo == sg.rb_c);
        int ru = run_wups<AT>(sg.up, x, ldi, Tc, xu, ldo, 0.1f, st, dma ? w.st[11] : nullptr, 0.1f);
        if (ru > 0) return ru;
        if (ru < 0) dma = false;   // no wups shape for this transposed conv (GSV_NO_WUPS, a stride / tap count outside its table): nobody writes the
                                   // activated copy, so the stage's resblock convs take the wconv / cgemm / tapgemm path below
        if (ru < 0)
            if (int rc = run_conv<AT, AT, AT>(sg.up, x, ldi, Tc, xu, ldo, Tc, eu, st)) return rc;
        if (sizeof(AT) == 2 && sg.rb_c && ldo == sg.rb_c) {
            // <= 32 channels: the three branches and their mean in ONE kernel, intermediates never leave the CU (rbfuse.h)
            RbFuseArgs ra;
            memset(&ra, 0, sizeof(ra));
            ra.X = (const bf16_t*)xu; ra.Y = (bf16_t*)x; ra.W = (const uint4*)sg.rb_w; ra.B = sg.rb_b;
            for (int j = 0; j < 3; ++j) { ra.wofs[j] = sg.rb_wofs[j]; ra.dil[j] = c.resblock_dilations[j]; }
            ra.ld = ldo; ra.n_rows = Tn; ra.slope = 0.1f;
            if (sg.rb_c == 16) {
                HIPCHK(hipFuncSetAttribute((const void*)rbfuse_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RbShape<16>::LDS));
                hipLaunchKernelGGL(rbfuse_kernel<16>, dim3(cdiv(Tn, RbShape<16>::BN)), dim3(512), RbShape<16>::LDS, st, ra);
            } else {
                HIPCHK(hipFuncSetAttribute((const void*)rbfuse_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RbShape<32>::LDS));
                hipLaunchKernelGGL(rbfuse_kernel<32>, dim3(cdiv(Tn, RbShape<32>::BN)), dim3(512), RbShape<32>::LDS, st, ra);
            }
            Tc = Tn;
            continue;
        }
        if (ldo != sg.cout) HIPCHK(hipMemsetAsync(x, 0, sizeof(AT) * (size_t)Tn * ldo, st));
        // the three resblocks (k = 3, 7, 11) advance in lock step: one launch per conv position
        const AT* cur[3] = {xu, xu, xu};
        const void* cur_act[3] = {w.st[11], w.st[11], w.st[11]};
        for (int d = 0; d < 3; ++d) {
            Branch b1[3], b2[3];
            for (int j = 0; j < 3; ++j) {
                AT* t1 = (AT*)w.st[2 + 3 * j];
                AT* dst = (AT*)w.st[2 + 3 * j + 1 + (d & 1)];
                b1[j] = Branch{&sg.rb[j].c1[d], cur[j], t1, nullptr};
                b2[j] = Branch{&sg.rb[j].c2[d], t1, dst, cur[j]};
            }
            if (dma) {
                Branch a1[3];
                void* none[3] = {nullptr, nullptr, nullptr};
                void* act[3] = {d < 2 ? w.st[12] : nullptr, d < 2 ? w.st[13] : nullptr, d < 2 ? w.st[14] : nullptr};
                for (int j = 0; j < 3; ++j) a1[j] = Branch{b1[j].pc, cur_act[j], b1[j].Y, nullptr};
                int rd = run_wdma(a1, none, ldo, Tn, 0.1f, 1.0f, v->dma_zero, v->dma_sink, st);
                if (rd == 0) rd = run_wdma(b2, act, ldo, Tn, 1.0f, 0.1f, v->dma_zero, v->dma_sink, st);
                if (rd != 0) return rd > 0 ? rd : fail(GSV_ERR_STATE, "wdma declined a conv of a stage it accepted");
                for (int j = 0; j < 3; ++j) { cur[j] = (const AT*)b2[j].Y; cur_act[j] = act[j]; }
                continue;
            }
            // bf16, 16..128 channels: weights-in-registers kernel; the first conv writes lrelu(t1), which is
            // the only form its consumer reads, so the second conv stages its input without arithmetic
            int rw = run_cgemm<AT>(b1, ldo, Tn, 0.1f, 0.1f, st, w.cg_part, kCgPartBytes, v->cg_flag);
            if (rw > 0) return rw;
            if (rw == 0) {
                rw = run_cgemm<AT>(b2, ldo, Tn, 1.0f, 1.0f, st, w.cg_part, kCgPartBytes, v->cg_flag);
                if (rw != 0) return rw > 0 ? rw : fail(GSV_ERR_STATE, "cgemm accepted the first conv of a pair but not the second");
                for (int j = 0; j < 3; ++j) cur[j] = (const AT*)b2[j].Y;
                continue;
            }
            rw = run_wconv<AT>(b1, ldo, Tn, 0.1f, 0.1f, st);
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
    // one ge column per frame (a time-concatenated batch): the conditioning runs on the DISTINCT columns, found on the device
    // (bf16 handles; GSV_NO_GE_SEGMENTS=1 keeps one row per frame -- the A/B switch of tests/test_hip_vocoder.py)
    if (sizeof(AT) == 2 && Tg > 1 && getenv("GSV_NO_GE_SEGMENTS") == nullptr) {
        HIPCHK(hipMemsetAsync(w.seg_flag, 0, sizeof(int) * (size_t)Tg, st));
        hipLaunchKernelGGL(seg_flag_kernel, dim3(cdiv(Tg, 256), cdiv(c.gin_channels, 64)), dim3(256), 0, st, ge, c.gin_channels, Tg, w.seg_flag);
        hipLaunchKernelGGL(seg_scan_kernel, dim3(1), dim3(1024), 0, st, (const int*)w.seg_flag, Tg, w.seg_id, w.seg_first, w.nseg);
        hipLaunchKernelGGL((seg_gather_cl_kernel<AT>), dim3(cdiv(Tg, 32), cdiv(c.gin_channels, 256)), dim3(256), 0, st, ge, c.gin_channels, Tg,
                           (const int*)w.seg_first, (const int*)w.nseg, (AT*)w.ge_cl, c.gin_channels);
        w.seg = w.seg_id;
    } else {
        hipLaunchKernelGGL((cf_to_cl_kernel<AT>), dim3(cdiv(Tg, 32), cdiv(c.gin_channels, 32)), dim3(256), 0, st, ge,
                           (AT*)w.ge_cl, c.gin_channels, Tg, c.gin_channels);
    }
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

// ---- enc_p: weights (packed as bf16 fragments, or fp32 ones for the parity mode) --------------
template <typename CT>
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
        return pack_conv<CT>(pc, w, cout, cin, k, (int64_t)cin * k, k, 1, 1, (k - 1) / 2, 0, b, 1.f, st);
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
        return pack_conv<CT>(pc, w, n * cout_each, cin, 1, cin, 1, 0, 1, 0, 0, b, 1.f, st);
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
    E.has_ge512 = v->staged.count("ge_to512.weight") != 0;
    if (E.has_ge512)
        if (int rc = conv(E.ge512, "ge_to512", 512, v->cfg.gin_channels, 1)) return rc;
    E.ready = true;
    return GSV_OK;
}

void encp_free(gsv_voc* v) {
    EncP& E = v->enc;
    for (PackedConv* p : {&E.ssl_proj, &E.c_pre, &E.text_pre, &E.c_post, &E.proj, &E.xq, &E.xkv, &E.xo, &E.ge512}) free_conv(*p);
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

int encp_run(gsv_voc* v, const int64_t* codes, int n_codes, const int64_t* text, int P, const float* ge512, int Tg, int gshift,
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
                       Tg == 1 ? 0 : 512, gshift, w.xsum, T, 512);
    if (int rc = enc_gemm(E.c_post, w.xsum, 512, T, true, 0, w.y, Hc, false, 1, 0, st)) return rc;
    if (int rc = encp_encoder(v, E.enc2, w.y, T, w, st)) return rc;
    if (int rc = enc_gemm(E.proj, w.y, Hc, T, true, 0, w.stats, 2 * C, true, 1, 0, st)) return rc;
    hipLaunchKernelGGL((cl_to_cf_kernel<float>), dim3(cdiv(T, 32), cdiv(C, 32)), dim3(256), 0, st, (const float*)w.stats, m_p, C, T, 2 * C);
    hipLaunchKernelGGL((cl_to_cf_kernel<float>), dim3(cdiv(T, 32), cdiv(C, 32)), dim3(256), 0, st, (const float*)w.stats + C, logs_p, C, T, 2 * C);
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

// ---- enc_p, fp32 parity mode: the same layer sequence on fp32 tapgemm + the plain fp32 kernels of encp.h --------
struct EncWsF {
    float *y768, *y, *t, *qkv, *att, *tmp, *ffn, *ssl512, *text512, *xq, *xkv, *xatt, *xo, *xsum, *stats;
    size_t bytes;
};
EncWsF encp_layout_f32(const gsv_voc* v, int T, int P, char* base) {
    size_t off = 0;
    auto take = [&](size_t n) { float* p = base ? (float*)(base + off) : nullptr; off += align_up(4 * n, 256); return p; };
    const int Hc = v->cfg.hidden_channels, R = std::max(T, P);
    EncWsF w;
    w.y768 = take((size_t)T * 768); w.y = take((size_t)T * Hc); w.t = take((size_t)P * Hc); w.qkv = take((size_t)R * 3 * Hc);
    w.att = take((size_t)R * Hc); w.tmp = take((size_t)R * Hc); w.ffn = take((size_t)R * 4 * Hc);
    w.ssl512 = take((size_t)T * 512); w.text512 = take((size_t)P * 512); w.xq = take((size_t)T * 512); w.xkv = take((size_t)P * 1024);
    w.xatt = take((size_t)T * 512); w.xo = take((size_t)T * 512); w.xsum = take((size_t)T * 512);
    w.stats = take((size_t)T * 2 * v->cfg.inter_channels);
    w.bytes = off;
    return w;
}

int encp_dense_f32(const PackedConv& pc, const float* X, int ldx, int rows, float* Y, int ldy, int act, const float* res, hipStream_t st) {
    Epi e; e.act = act; e.res = res; e.ld_res = res ? ldy : 0;
    return run_conv<float, float, float>(pc, X, ldx, rows, Y, ldy, rows, e, st);
}

int encp_attn_f32(const float* Q, int ldq, int qoff, const float* K, const float* V, int ldkv, int koff, int voff, float* O, int ldo, int Tq, int Tk,
                  int H, int D, const float* relk, const float* relv, const int64_t* slice, float* P, hipStream_t st) {
    EncAttnF32Args a;
    a.Q = Q; a.K = K; a.V = V; a.ldq = ldq; a.ldk = ldkv; a.ldv = ldkv; a.qoff = qoff; a.koff = koff; a.voff = voff; a.O = O; a.ldo = ldo;
    a.Tq = Tq; a.Tk = Tk; a.D = D; a.rsqrt_d = 0.f; a.relk = relk; a.relv = relv; a.window = 4; a.slice = slice; a.P = P;
    const size_t lds = sizeof(float) * 4 * (size_t)(Tk + D);
    if (lds > 160 * 1024) return fail(GSV_ERR_ARG, "enc_p (fp32): %d keys exceed the attention kernel's LDS rows", Tk);
    HIPCHK(hipFuncSetAttribute((const void*)encp_attn_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(encp_attn_f32_kernel, dim3(H, cdiv(Tq, 4)), dim3(256), lds, st, a);
    return GSV_OK;
}

int encp_encoder_f32(gsv_voc* v, std::vector<EncLayer>& Ls, float* x, int R, EncWsF& w, hipStream_t st) {
    const int Hc = v->cfg.hidden_channels;
    for (EncLayer& L : Ls) {
        if (int rc = encp_dense_f32(L.qkv, x, Hc, R, w.qkv, 3 * Hc, ACT_NONE, nullptr, st)) return rc;
        if (int rc = encp_attn_f32(w.qkv, 3 * Hc, 0, w.qkv, w.qkv, 3 * Hc, Hc, 2 * Hc, w.att, Hc, R, R, 2, Hc / 2, L.relk, L.relv, nullptr, nullptr, st)) return rc;
        if (int rc = encp_dense_f32(L.o, w.att, Hc, R, w.tmp, Hc, ACT_NONE, x, st)) return rc;      // x + attn_out
        hipLaunchKernelGGL(encp_ln_f32_kernel, dim3(cdiv(R, 4)), dim3(256), 0, st, w.tmp, (const float*)L.g1, (const float*)L.b1, R, Hc);
        if (int rc = encp_dense_f32(L.c1, w.tmp, Hc, R, w.ffn, 4 * Hc, ACT_RELU, nullptr, st)) return rc;
        if (int rc = encp_dense_f32(L.c2, w.ffn, 4 * Hc, R, x, Hc, ACT_NONE, w.tmp, st)) return rc;  // x + ffn_out
        hipLaunchKernelGGL(encp_ln_f32_kernel, dim3(cdiv(R, 4)), dim3(256), 0, st, x, (const float*)L.g2, (const float*)L.b2, R, Hc);
    }
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

int encp_run_f32(gsv_voc* v, const int64_t* codes, int n_codes, const int64_t* text, int P, const float* ge512, int Tg, int gshift,
                 const int64_t* slice, float* m_p, float* logs_p, float* attn, void* ws, size_t ws_bytes, hipStream_t st) {
    EncP& E = v->enc;
    const int Hc = v->cfg.hidden_channels, C = v->cfg.inter_channels, T = 2 * n_codes;
    EncWsF w = encp_layout_f32(v, T, P, (char*)ws);
    if (ws_bytes < w.bytes) return fail(GSV_ERR_ARG, "enc_p workspace %zu < %zu", ws_bytes, w.bytes);
    hipLaunchKernelGGL(encp_gather_f32_kernel, dim3(T), dim3(256), 0, st, codes, E.n_code, (const float*)E.codebook, 768, 2, w.y768);
    hipLaunchKernelGGL(encp_gather_f32_kernel, dim3(P), dim3(64), 0, st, text, E.n_text, (const float*)E.text_emb, Hc, 1, w.t);
    if (int rc = encp_dense_f32(E.ssl_proj, w.y768, 768, T, w.y, Hc, ACT_NONE, nullptr, st)) return rc;
    if (int rc = encp_encoder_f32(v, E.ssl, w.y, T, w, st)) return rc;
    if (int rc = encp_encoder_f32(v, E.text, w.t, P, w, st)) return rc;
    if (int rc = encp_dense_f32(E.c_pre, w.y, Hc, T, w.ssl512, 512, ACT_NONE, nullptr, st)) return rc;
    if (int rc = encp_dense_f32(E.text_pre, w.t, Hc, P, w.text512, 512, ACT_NONE, nullptr, st)) return rc;
    if (int rc = encp_dense_f32(E.xq, w.ssl512, 512, T, w.xq, 512, ACT_NONE, nullptr, st)) return rc;
    if (int rc = encp_dense_f32(E.xkv, w.text512, 512, P, w.xkv, 1024, ACT_NONE, nullptr, st)) return rc;
    if (int rc = encp_attn_f32(w.xq, 512, 0, w.xkv, w.xkv, 1024, 0, 512, w.xatt, 512, T, P, 4, 128, nullptr, nullptr, slice, attn, st)) return rc;
    if (int rc = encp_dense_f32(E.xo, w.xatt, 512, T, w.xo, 512, ACT_NONE, nullptr, st)) return rc;
    hipLaunchKernelGGL(encp_add3_f32_kernel, dim3(std::min(2048, cdiv(T * 512, 256))), dim3(256), 0, st, (const float*)w.xo, (const float*)w.ssl512, ge512,
                       Tg == 1 ? 0 : 512, gshift, w.xsum, T, 512);
    if (int rc = encp_dense_f32(E.c_post, w.xsum, 512, T, w.y, Hc, ACT_NONE, nullptr, st)) return rc;
    if (int rc = encp_encoder_f32(v, E.enc2, w.y, T, w, st)) return rc;
    if (int rc = encp_dense_f32(E.proj, w.y, Hc, T, w.stats, 2 * C, ACT_NONE, nullptr, st)) return rc;
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
                    if (!rc && sizeof(CT) == 2 && cpad == co) rc = pack_cgemm(pc, w, co, rb.k, st);   // wide stages: also the cgemm.h order
                }
            }
        }
        // one fused kernel for the whole stage (three branches + mean) where the channels allow it
        if (!rc && sizeof(CT) == 2 && !getenv("GSV_NO_RBFUSE") && (co == 16 || co == 24 || co == 32) && c.n_resblock_kernels == 3 &&
            c.resblock_kernel_sizes[0] == 3 && c.resblock_kernel_sizes[1] == 7 && c.resblock_kernel_sizes[2] == 11) {
            const int RC = co == 16 ? 16 : 32;
            RbPackArgs pa;
            memset(&pa, 0, sizeof(pa));
            int ofs = 0;
            for (int j = 0; j < 3; ++j) {
                pa.k[j] = c.resblock_kernel_sizes[j];
                pa.wofs[j] = sg.rb_wofs[j] = ofs;
                ofs += 6 * (RC == 16 ? RbShape<16>::steps(pa.k[j]) * RbShape<16>::HV : RbShape<32>::steps(pa.k[j]) * RbShape<32>::HV);
            }
            for (int j = 0; j < 3 && !rc; ++j)
                for (int d = 0; d < 3 && !rc; ++d)
                    for (int which = 0; which < 2 && !rc; ++which) {
                        const std::string cn = "dec.resblocks." + std::to_string(i * 3 + j) + (which ? ".convs2." : ".convs1.") + std::to_string(d);
                        if ((rc = get(cn + ".weight", (int64_t)co * co * pa.k[j], &pa.w[j * 6 + d * 2 + which])) ||
                            (rc = get(cn + ".bias", co, &pa.b[j * 6 + d * 2 + which]))) break;
                    }
            if (!rc) {
                if (!sg.rb_w) HIPCHK(hipMalloc(&sg.rb_w, (size_t)ofs * 1024));
                if (!sg.rb_b) HIPCHK(hipMalloc(&sg.rb_b, sizeof(float) * 18 * RC));
                pa.creal = co; pa.W = (uint4*)sg.rb_w; pa.B = sg.rb_b;
                if (RC == 16) hipLaunchKernelGGL(rbfuse_pack_kernel<16>, dim3(18), dim3(256), 0, st, pa);
                else hipLaunchKernelGGL(rbfuse_pack_kernel<32>, dim3(18), dim3(256), 0, st, pa);
                sg.rb_c = RC;
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
    if (!rc && sizeof(CT) == 2 && !v->dma_zero) {   // failures here go through `rc`: the temporaries below are still freed behind the stream
        hipError_t e = hipMalloc(&v->dma_zero, 4096);
        if (e == hipSuccess) e = hipMemsetAsync(v->dma_zero, 0, 4096, st);
        if (e == hipSuccess) e = hipMalloc(&v->dma_sink, 4096);
        if (e != hipSuccess) rc = fail(GSV_ERR_HIP, "wdma zero page / sink: %s", hipGetErrorString(e));
    }
    if (!rc && sizeof(CT) == 2 && !v->cg_flag) {    // a stage of 192 / 256 / 384 channels: cgemm.h, whose small launches K-split
        bool wide = false;
        for (const VocStage& sg : v->stages) wide = wide || sg.cout == 192 || sg.cout == 256 || sg.cout == 384;
        if (wide) {
            hipError_t e = hipMalloc(&v->cg_flag, sizeof(int) * 2 * kCgSplitMaxTiles);
            if (e == hipSuccess) e = hipMemsetAsync(v->cg_flag, 0, sizeof(int) * 2 * kCgSplitMaxTiles, st);
            if (e != hipSuccess) rc = fail(GSV_ERR_HIP, "cgemm hand-off flags: %s", hipGetErrorString(e));
        }
    }
    if (!rc && v->staged.count("enc_p.ssl_proj.weight")) rc = encp_finalize<CT>(v, temps, st);
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
    if (v->dma_zero) (void)hipFree(v->dma_zero);
    if (v->dma_sink) (void)hipFree(v->dma_sink);
    v->dma_zero = v->dma_sink = nullptr;
    if (v->cg_flag) (void)hipFree(v->cg_flag);
    v->cg_flag = nullptr;
    for (VocStage& s : v->stages) {
        free_conv(s.up);
        if (s.rb_w) (void)hipFree(s.rb_w);
        if (s.rb_b) (void)hipFree(s.rb_b);
        s.rb_w = nullptr; s.rb_b = nullptr; s.rb_c = 0;
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
    for (auto& kv : v->graphs) (void)hipGraphExecDestroy(kv.second.exec);
    if (v->cap_stream) (void)hipStreamDestroy(v->cap_stream);
    voc_free(v);
    delete v;
    return GSV_OK;
}

int gsv_voc_load_tensor(gsv_voc* v, const char* name, const float* data, int64_t numel, void* stream) {
    if (!v || !name || !data || numel < 1) return fail(GSV_ERR_ARG, "null argument");
    if (v->finalized) return fail(GSV_ERR_STATE, "vocoder already finalized");
    std::string n(name);
    if (n.compare(0, 4, "dec.") != 0 && n.compare(0, 5, "flow.") != 0 && n.compare(0, 6, "enc_p.") != 0 && n.compare(0, 10, "quantizer.") != 0 &&
        n.compare(0, 9, "ge_to512.") != 0)
        return fail(GSV_ERR_ARG, "tensor '%s' is not part of flow / dec / enc_p / quantizer / ge_to512", name);
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
    return v->cfg.dtype == GSV_BF16 ? encp_layout(v, 2 * n_codes, n_text, nullptr).bytes : encp_layout_f32(v, 2 * n_codes, n_text, nullptr).bytes;
}

int gsv_voc_enc_p(gsv_voc* v, const int64_t* codes, int n_codes, const int64_t* text, int n_text, const float* ge512, int Tg,
                  const int64_t* slice_indices, float* m_p, float* logs_p, float* attn, void* workspace, size_t workspace_bytes,
                  void* stream) {
    if (!v || !v->finalized) return fail(GSV_ERR_STATE, "vocoder not finalized");
    if (!v->enc.ready) return fail(GSV_ERR_STATE, "enc_p tensors were not loaded");
    if (!codes || !text || !ge512 || !m_p || !logs_p || !workspace) return fail(GSV_ERR_ARG, "null argument");
    if (n_codes < 1 || n_text < 1 || (Tg != 1 && Tg != 2 * n_codes)) return fail(GSV_ERR_ARG, "enc_p: bad lengths");
    if (v->cfg.dtype != GSV_BF16)
        return encp_run_f32(v, codes, n_codes, text, n_text, ge512, Tg, 0, slice_indices, m_p, logs_p, attn, workspace, workspace_bytes, S(stream));
    return encp_run(v, codes, n_codes, text, n_text, ge512, Tg, 0, slice_indices, m_p, logs_p, attn, workspace, workspace_bytes, S(stream));
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

int gsv_voc_flow_dec_graph(gsv_voc* v, const float* z_p, const float* y_mask, const float* ge, int T, int Tg, float* out,
                           void* workspace, size_t workspace_bytes, void* stream) {
    if (!v || !z_p || !y_mask || !ge || !out || !workspace) return fail(GSV_ERR_ARG, "null argument");
    const gsv_voc::GraphKey key{z_p, y_mask, ge, out, workspace, T, Tg};
    auto it = v->graphs.find(key);
    if (it == v->graphs.end()) {
        if (v->graphs.size() >= GSV_VOC_MAX_GRAPHS) {      // a long-lived server sees new (workspace, T) pairs for ever: evict, never fail
            auto old = v->graphs.begin();
            for (auto jt = v->graphs.begin(); jt != v->graphs.end(); ++jt)
                if (jt->second.last_use < old->second.last_use) old = jt;
            // its last replay may have been enqueued on ANY stream (a per-request side stream of the engine), not only the caller's:
            // eviction is rare, so the whole device drains before the exec goes
            HIPCHK(hipDeviceSynchronize());
            (void)hipGraphExecDestroy(old->second.exec);
            v->graphs.erase(old);
        }
        if (!v->cap_stream && hipStreamCreateWithFlags(&v->cap_stream, hipStreamNonBlocking) != hipSuccess)
            return fail(GSV_ERR_HIP, "hipStreamCreate failed");
        // an eager pass first: it sets every kernel's dynamic-LDS attribute outside the capture and validates the arguments
        if (int rc = gsv_voc_flow_dec(v, z_p, y_mask, ge, T, Tg, out, workspace, workspace_bytes, stream)) return rc;
        HIPCHK(hipStreamSynchronize(S(stream)));
        hipGraph_t g = nullptr;
        HIPCHK(hipStreamBeginCapture(v->cap_stream, hipStreamCaptureModeThreadLocal));
        int rc = gsv_voc_flow_dec(v, z_p, y_mask, ge, T, Tg, out, workspace, workspace_bytes, v->cap_stream);
        hipError_t e = hipStreamEndCapture(v->cap_stream, &g);
        if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (e != hipSuccess) return fail(GSV_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) return fail(GSV_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
        it = v->graphs.emplace(key, gsv_voc::GraphEntry{exec, 0}).first;
    }
    it->second.last_use = ++v->graph_clock;
    HIPCHK(hipGraphLaunch(it->second.exec, S(stream)));
    return GSV_OK;
}

// y[c][j] = linear resampling of x[c][:] to T_out points, torch's F.interpolate(mode="linear", align_corners=False):
// src = (j + 0.5) * T_in / T_out - 0.5 clamped at 0, weights (1 - frac, frac), right neighbour clamped to T_in - 1
static __global__ __launch_bounds__(256) void resample_linear_kernel(const float* __restrict__ x, int C, int T_in, float* __restrict__ y, int T_out) {
    const int j = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (j >= T_out) return;
    const float scale = (float)T_in / (float)T_out;
    float src = ((float)j + 0.5f) * scale - 0.5f;
    if (src < 0.f) src = 0.f;
    const int i0 = min((int)src, T_in - 1), i1 = min(i0 + 1, T_in - 1);
    const float f = src - (float)i0;
    y[(size_t)c * T_out + j] = (1.0f - f) * x[(size_t)c * T_in + i0] + f * x[(size_t)c * T_in + i1];
}

int gsv_voc_resample_linear(const float* x, int C, int T_in, float* y, int T_out, void* stream) {
    if (!x || !y || C < 1 || T_in < 1 || T_out < 1) return fail(GSV_ERR_ARG, "resample: bad arguments");
    hipLaunchKernelGGL(resample_linear_kernel, dim3(cdiv(T_out, 256), C), dim3(256), 0, S(stream), x, C, T_in, y, T_out);
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

}  // extern "C"

// =============================================================================================
// SynthesizerTrn.decode (SoVITS/models.py:385-429) in one call
// =============================================================================================
namespace {

__device__ __forceinline__ uint32_t dec_lowbias32(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
// standard normal for element `i` of the stream `seed`: Box-Muller over two counter-based uniforms (lowbias32 of the element
// index mixed with the seed halves) -- replayable, no generator state; oracle.device_normal restates it
__device__ __forceinline__ float dec_normal(uint32_t seed_lo, uint32_t seed_hi, uint32_t i) {
    const uint32_t a = dec_lowbias32(dec_lowbias32(i * 0x9E3779B1u ^ seed_lo) + seed_hi);
    const uint32_t b = dec_lowbias32(dec_lowbias32(i * 0x85EBCA77u ^ seed_hi ^ 0x68E31DA4u) + seed_lo);
    const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(b >> 8) + 0.5f) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

// streaming (models.py:209-215, applied to the projected statistics: proj is 1x1 affine, so it commutes): drop the first
// `start` frames, cross-fade the first `ov` kept frames with the previous chunk's tail
__global__ void dec_slice_xfade_kernel(const float* __restrict__ in, int T_in, int start, const float* __restrict__ prev, int has_prev, int ov,
                                       float* __restrict__ out, int Tp, int C2) {
    const size_t n = (size_t)C2 * Tp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i / Tp), j = (int)(i % Tp);
        float v = in[(size_t)c * T_in + start + j];
        if (has_prev && j < ov) {
            const float a = ov > 1 ? (float)j / (float)(ov - 1) : 0.f;       // torch.linspace(0, 1, ov)
            v = prev[(size_t)c * ov + j] * (1.0f - a) + v * a;
        }
        out[i] = v;
    }
}
__global__ void dec_keep_tail_kernel(const float* __restrict__ x, int Tp, int ov, float* __restrict__ state, int C2) {
    const int n = C2 * ov;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int c = i / ov, j = i % ov;
        state[i] = x[(size_t)c * Tp + (Tp - ov) + j];
    }
}
// z_p = m_p + randn * exp(logs_p) * noise_scale (models.py:404); stats = [m_p | logs_p] channels-first [2C][T]; mask = ones
__global__ void dec_zp_kernel(const float* __restrict__ stats, int C, int T, float noise_scale, uint32_t seed_lo, uint32_t seed_hi,
                              float* __restrict__ z, float* __restrict__ mask) {
    const size_t n = (size_t)C * T;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = stats[i];
        if (noise_scale != 0.f) v += dec_normal(seed_lo, seed_hi, (uint32_t)i) * expf(stats[n + i]) * noise_scale;
        z[i] = v;
        if (i < (size_t)T) mask[i] = 1.0f;
    }
}
// per-token ge [gin][Tg] -> per-frame [gin][T_out]: x2 nearest (models.py:389), then F.interpolate(mode="nearest") to the
// resampled length (models.py:402): frame j reads column min(floor(j * (2 Tg / T_out)), 2 Tg - 1) / 2
__global__ void dec_ge_frames_kernel(const float* __restrict__ ge, int gin, int Tg, float* __restrict__ out, int T_out, int resized) {
    const size_t n = (size_t)gin * T_out;
    const float scale = (float)(2 * Tg) / (float)T_out;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i / T_out), j = (int)(i % T_out);
        const int f = resized ? min((int)floorf((float)j * scale), 2 * Tg - 1) : j;
        out[i] = ge[(size_t)c * Tg + (f >> 1)];
    }
}

struct DecWs {
    float *z_p, *mask, *ge_fr, *out_static;  // what the captured flow + Generator pass reads / writes: first, so their
    void* voc; size_t voc_bytes;             // addresses depend on T_out / Tg only
    void* ge_cl; float *ge512, *stats, *stats_s, *stats_r;
    void* enc; size_t enc_bytes;
    size_t bytes;
};
DecWs dec_layout(gsv_voc* v, int n_codes, int P, int Tg, int Tp, int T_out, char* base) {
    const gsv_voc_config& c = v->cfg;
    const int C = c.inter_channels, T = 2 * n_codes;
    size_t off = 0;
    auto take = [&](size_t bytes) { void* p = base ? base + off : nullptr; off += align_up(bytes, 256); return p; };
    DecWs w;
    const int Tgv = Tg == 1 ? 1 : T_out;
    w.z_p = (float*)take(sizeof(float) * (size_t)C * T_out);
    w.mask = (float*)take(sizeof(float) * (size_t)T_out);
    w.ge_fr = (float*)take(sizeof(float) * (size_t)c.gin_channels * Tgv);
    w.out_static = (float*)take(sizeof(float) * (size_t)T_out * v->total_up);
    w.voc_bytes = c.dtype == GSV_BF16 ? voc_layout<bf16_t>(v, T_out, Tgv, nullptr).bytes : voc_layout<float>(v, T_out, Tgv, nullptr).bytes;
    w.voc = take(w.voc_bytes);
    w.ge_cl = take(sizeof(float) * (size_t)Tg * c.gin_channels);
    w.ge512 = (float*)take(sizeof(float) * (size_t)Tg * 512);
    w.stats = (float*)take(sizeof(float) * (size_t)2 * C * T);
    w.stats_s = (float*)take(sizeof(float) * (size_t)2 * C * Tp);
    w.stats_r = (float*)take(sizeof(float) * (size_t)2 * C * T_out);
    w.enc_bytes = c.dtype == GSV_BF16 ? encp_layout(v, T, P, nullptr).bytes : encp_layout_f32(v, T, P, nullptr).bytes;
    w.enc = take(w.enc_bytes);
    w.bytes = off;
    return w;
}

// The frame count after the speed change, int(Tp / speed) + 1 (models.py:217), is the CALLER's: it sized `out` with it in its
// own arithmetic (Python doubles in the reference), and a second evaluation here in another precision disagrees in ~1.4 % of
// (Tp, speed) pairs -- one hop written past the buffer or left unwritten.  out_frames == Tp means no resampling.
inline int dec_lengths(int n_codes, int out_frames, int valid_start, int* Tp, int* T_out) {
    const int T = 2 * n_codes;
    *Tp = T - valid_start;
    *T_out = out_frames;
    return *Tp >= 1 && *T_out >= 1;
}

template <typename AT>
int voc_decode_impl(gsv_voc* v, const int64_t* codes, int n_codes, const int64_t* text, int P, const float* ge, int Tg, const int64_t* slice,
                    float noise_scale, unsigned long long seed, int out_frames, int valid_start, int overlap_len, float* overlap_state,
                    int has_overlap, int use_graph, float* out, float* attn, void* ws, size_t ws_bytes, hipStream_t st) {
    const gsv_voc_config& c = v->cfg;
    const int C = c.inter_channels, gin = c.gin_channels, T = 2 * n_codes;
    const bool stream = overlap_len > 0;
    int Tp, T_out;
    if (!dec_lengths(n_codes, out_frames, valid_start, &Tp, &T_out)) return fail(GSV_ERR_ARG, "decode: nothing left after valid_start %d", valid_start);
    if (stream && (Tg != 1 || !overlap_state || overlap_len > Tp)) return fail(GSV_ERR_ARG, "decode: streaming needs a broadcast ge, a state buffer and overlap_len <= frames");
    DecWs w = dec_layout(v, n_codes, P, Tg, Tp, T_out, (char*)ws);
    if (ws_bytes < w.bytes) return fail(GSV_ERR_ARG, "decode workspace %zu < %zu", ws_bytes, w.bytes);
    EncP& E = v->enc;
    // ---- conditioning of enc_p: ge_to512(ge) for v2Pro / v2ProPlus (models.py:394), ge itself (512 channels) otherwise
    const float* g512;
    hipLaunchKernelGGL((cf_to_cl_kernel<AT>), dim3(cdiv(Tg, 32), cdiv(gin, 32)), dim3(256), 0, st, ge, (AT*)w.ge_cl, gin, Tg, gin);
    if (E.has_ge512) {
        if (int rc = run_cond<AT>(E.ge512, w.ge_cl, gin, Tg, w.ge512, 512, st)) return rc;
        g512 = w.ge512;
    } else {
        if (gin != 512) return fail(GSV_ERR_STATE, "decode: %d-channel ge and no ge_to512 tensors", gin);
        hipLaunchKernelGGL((cf_to_cl_kernel<float>), dim3(cdiv(Tg, 32), cdiv(gin, 32)), dim3(256), 0, st, ge, w.ge512, gin, Tg, gin);
        g512 = w.ge512;
    }
    // ---- enc_p: quantizer lookup, x2 upsampling, encoders, MRTE -> [m_p | logs_p] [2C][T]
    int rc = sizeof(AT) == 2 ? encp_run(v, codes, n_codes, text, P, g512, Tg == 1 ? 1 : T, 1, slice, w.stats, w.stats + (size_t)C * T, attn, w.enc, w.enc_bytes, st)
                             : encp_run_f32(v, codes, n_codes, text, P, g512, Tg == 1 ? 1 : T, 1, slice, w.stats, w.stats + (size_t)C * T, attn, w.enc, w.enc_bytes, st);
    if (rc) return rc;
    const float* cur = w.stats;
    if (stream || valid_start > 0) {
        hipLaunchKernelGGL(dec_slice_xfade_kernel, dim3(std::min(1024, cdiv(2 * C * Tp, 256))), dim3(256), 0, st, cur, T, valid_start, overlap_state,
                           stream && has_overlap ? 1 : 0, overlap_len, w.stats_s, Tp, 2 * C);
        if (stream) hipLaunchKernelGGL(dec_keep_tail_kernel, dim3(cdiv(2 * C * overlap_len, 256)), dim3(256), 0, st, (const float*)w.stats_s, Tp, overlap_len, overlap_state, 2 * C);
        cur = w.stats_s;
    }
    if (T_out != Tp) {
        hipLaunchKernelGGL(resample_linear_kernel, dim3(cdiv(T_out, 256), 2 * C), dim3(256), 0, st, cur, 2 * C, Tp, w.stats_r, T_out);
        cur = w.stats_r;
    }
    hipLaunchKernelGGL(dec_zp_kernel, dim3(std::min(2048, cdiv(C * T_out, 256))), dim3(256), 0, st, cur, C, T_out, noise_scale,
                       (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), w.z_p, w.mask);
    const int Tgv = Tg == 1 ? 1 : T_out;
    if (Tg == 1) HIPCHK(hipMemcpyAsync(w.ge_fr, ge, sizeof(float) * gin, hipMemcpyDeviceToDevice, st));
    else hipLaunchKernelGGL(dec_ge_frames_kernel, dim3(std::min(2048, cdiv(gin * T_out, 256))), dim3(256), 0, st, ge, gin, Tg, w.ge_fr, T_out, T_out != T ? 1 : 0);
    HIPCHK(hipGetLastError());
    // ---- flow + Generator (models.py:380-383), replayed from the bucket's hipGraph when asked to
    if (use_graph) {
        if (int rc2 = gsv_voc_flow_dec_graph(v, w.z_p, w.mask, w.ge_fr, T_out, Tgv, w.out_static, w.voc, w.voc_bytes, st)) return rc2;
        HIPCHK(hipMemcpyAsync(out, w.out_static, sizeof(float) * (size_t)T_out * v->total_up, hipMemcpyDeviceToDevice, st));
        return GSV_OK;
    }
    return voc_run<AT>(v, 3, w.z_p, w.mask, w.ge_fr, T_out, Tgv, out, w.voc, w.voc_bytes, st);
}

}  // namespace

extern "C" {

size_t gsv_voc_decode_workspace(gsv_voc* v, int n_codes, int n_text, int Tg, int out_frames, int valid_start) {
    if (!v || !v->finalized || !v->enc.ready || n_codes < 1 || n_text < 1 || (Tg != 1 && Tg != n_codes) || out_frames < 1 || valid_start < 0) return 0;
    int Tp, T_out;
    if (!dec_lengths(n_codes, out_frames, valid_start, &Tp, &T_out)) return 0;
    return dec_layout(v, n_codes, n_text, Tg, Tp, T_out, nullptr).bytes;
}

int gsv_voc_decode(gsv_voc* v, const int64_t* codes, int n_codes, const int64_t* text, int n_text, const float* ge, int Tg,
                   const int64_t* slice_indices, float noise_scale, unsigned long long seed, int out_frames, int valid_start, int overlap_len,
                   float* overlap_state, int has_overlap, int use_graph, float* out, float* attn, void* workspace, size_t workspace_bytes,
                   void* stream) {
    if (!v || !v->finalized) return fail(GSV_ERR_STATE, "vocoder not finalized");
    if (!v->enc.ready) return fail(GSV_ERR_STATE, "decode() needs the enc_p / quantizer tensors");
    if (!codes || !text || !ge || !out || !workspace) return fail(GSV_ERR_ARG, "null argument");
    if (n_codes < 1 || n_text < 1 || (Tg != 1 && Tg != n_codes) || out_frames < 1 || valid_start < 0 || overlap_len < 0)
        return fail(GSV_ERR_ARG, "decode: bad lengths");
    return v->cfg.dtype == GSV_BF16
               ? voc_decode_impl<bf16_t>(v, codes, n_codes, text, n_text, ge, Tg, slice_indices, noise_scale, seed, out_frames, valid_start, overlap_len,
                                         overlap_state, has_overlap, use_graph, out, attn, workspace, workspace_bytes, S(stream))
               : voc_decode_impl<float>(v, codes, n_codes, text, n_text, ge, Tg, slice_indices, noise_scale, seed, out_frames, valid_start, overlap_len,
                                        overlap_state, has_overlap, use_graph, out, attn, workspace, workspace_bytes, S(stream));
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
