// C ABI of the MI355X GPT-SoVITS hot path (include/gsv_tts_hip.h), GPT part: handle management, weight
// repacking into library-owned arenas, kernel sequencing, hipGraph capture of the decode step.  (SoVITS: gsv_voc.hip.)
// No allocation happens inside a step; nothing here falls back to a CPU or library path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>
#include <utility>
#include <vector>

// One arena per GPT handle: every device buffer the handle owns (decode panels, MFMA fragments, scratch, staging) is carved out
// of a few 256 MB blocks at 64 KB alignment instead of ~450 separate hipMalloc calls.  Measured reason (10 model instances per
// setting, ms per decode step): separate allocations 0.323 - 0.343 (the small buffers land wherever the driver's sub-allocator
// has room; about one instance in four is 3 - 9 % slow); arena at 4 KB alignment 0.329 - 0.335, at 2 MB + k * 4 KB 0.323 - 0.335,
// at 2 MB + k * 68 KB 0.329 - 0.339; arena at 64 KB alignment 0.323 - 0.326 and at 2 MB + k * 256 B 0.324 - 0.326 -- every
// instance at the best time.  GSV_NO_ARENA=1 switches it off, GSV_ARENA_ALIGN / GSV_ARENA_SKEW are the experiment's knobs.
namespace gsv_arena {
struct Arena {
    std::vector<std::pair<char*, size_t>> blocks;
    size_t used = 0;
    size_t count = 0;
    std::unordered_map<void*, size_t> live;      // carved pieces by address
    std::multimap<size_t, void*> freed;          // pieces given back (a re-bound state's staging, a re-loaded tensor's fragments,
                                                 // grown scratch): handed out again by size, so a handle that is re-bound or
                                                 // re-loaded for its whole life does not grow
};
static std::mutex g_mu;
static std::vector<std::pair<std::pair<char*, size_t>, Arena*>> g_all;
static thread_local Arena* g_cur = nullptr;
static hipError_t amalloc(void** p, size_t n) {
    if (!g_cur) return hipMalloc(p, n);
    static const size_t align = getenv("GSV_ARENA_ALIGN") ? (size_t)atol(getenv("GSV_ARENA_ALIGN")) : 65536;
    static const size_t skew = getenv("GSV_ARENA_SKEW") ? (size_t)atol(getenv("GSV_ARENA_SKEW")) : 0;
    constexpr size_t kBlock = (size_t)256 << 20;
    Arena& a = *g_cur;
    bool reuse = false;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = a.freed.lower_bound(n);        // smallest given-back piece that fits, if it is not wastefully large
        if (it != a.freed.end() && it->first <= n + std::max<size_t>(n / 4, align)) {
            *p = it->second;
            a.live[*p] = it->first;
            a.freed.erase(it);
            reuse = true;
        }
    }
    if (reuse) {
        // never reached under a stream capture (hipDeviceSynchronize would invalidate it): pieces come back only from a re-bind or a
        // re-load, and neither is called from the captured step
        // kernels or graph replays on ANY stream (the side refill stream's staging, a hot-swapped conv) may still read the
        // piece's old contents: hipFree would have synchronised the device before the address could come back, so does this
        // (re-bind / re-load only: never on a step's path)
        return hipDeviceSynchronize();
    }
    const size_t sk = (a.count * skew) % align;
    size_t start = (a.used + align - 1) / align * align + sk;
    if (a.blocks.empty() || start + n > a.blocks.back().second) {
        const size_t sz = std::max(n + align, kBlock);
        char* base = nullptr;
        const hipError_t e = hipMalloc(&base, sz);
        if (e != hipSuccess) return e;
        a.blocks.emplace_back(base, sz);
        a.used = 0;
        start = sk;
        std::lock_guard<std::mutex> lk(g_mu);
        g_all.push_back({{base, sz}, &a});
    }
    *p = a.blocks.back().first + start;
    a.used = start + n;
    a.count++;
    std::lock_guard<std::mutex> lk(g_mu);
    a.live[*p] = n;
    return hipSuccess;
}
static hipError_t afree(void* p) {
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto& b : g_all)
            if ((char*)p >= b.first.first && (char*)p < b.first.first + b.first.second) {
                Arena& a = *b.second;
                auto it = a.live.find(p);
                if (it != a.live.end()) {
                    a.freed.emplace(it->second, p);
                    a.live.erase(it);
                }
                return hipSuccess;
            }
    }
    return hipFree(p);
}
static void release(Arena& a) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& b : a.blocks) {
        (void)hipFree(b.first);
        g_all.erase(std::remove_if(g_all.begin(), g_all.end(), [&](const std::pair<std::pair<char*, size_t>, Arena*>& e) { return e.first == b; }),
                    g_all.end());
    }
    a.blocks.clear();
    a.live.clear();
    a.freed.clear();
    a.used = 0;
}
struct Scope {
    Arena* prev;
    explicit Scope(Arena* a) : prev(g_cur) { g_cur = a; }
    ~Scope() { g_cur = prev; }
};
}  // namespace gsv_arena
#define GSV_DEV_ALLOC_ARENA 1   // abi_common.h: this unit's gsv_dev_malloc / gsv_dev_free are the arena's

#include "abi_common.h"
#include "t2s_decode.h"
#include "t2s_decode_multi.h"
#include "t2s_batch.h"
#include "t2s_small.h"

namespace {
thread_local std::string g_err;
}

int gsv::abi_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// =============================================================================================
// GPT
// =============================================================================================
struct T2SLayer {
    void *wqkv_p = nullptr, *wo_p = nullptr, *w1 = nullptr, *w2_p = nullptr;  // decode panels (WT)
    void* w2_p64 = nullptr;                                                    // W2 in 64 panels of 32 columns (bf16 handles, <= kFineMaxB sequences)
    float *bqkv_p = nullptr, *bo = nullptr, *b1 = nullptr, *b2 = nullptr, *ln1g = nullptr, *ln1b = nullptr,
          *ln2g = nullptr, *ln2b = nullptr;
    PackedConv g_qkv, g_out, g_w1, g_w2;  // prefill / batched step (MFMA fragments)
    void *p8_qkv = nullptr, *p8_w1 = nullptr, *p8_w2 = nullptr;   // GSV_FP8: the e4m3 form of the same (scales: s_qkv / s_w1 / s_w2)
    void *p16_qkv = nullptr, *p16_out = nullptr, *p16_w1 = nullptr, *p16_w2 = nullptr;   // bf16 handles: 16 x 16 x 32 fragments of the batched step at <= kSmallMaxM rows (t2s_small.h)
    void *f8_qkv = nullptr, *f8_w1 = nullptr, *f8_w2 = nullptr;    // GSV_FP8: e4m3 fragments of the batched step (t2s_batch.h)
    float *s_qkv = nullptr, *s_w1 = nullptr, *s_w2 = nullptr;      // ... and their per-output-channel scales
    unsigned have = 0;
};

struct T2SStateX : gsv_t2s_state {      // the caller's state + what the library keeps beside it
    int32_t* eos_host = nullptr;         // host-mapped mirror of eos_at (gsv_t2s_set_eos_mirror)
};
inline int32_t* eos_host_of(const gsv_t2s_state& s) { return static_cast<const T2SStateX&>(s).eos_host; }   // every state here is a T2SBound's

struct T2SBound {
    T2SStateX st;
    hipGraphExec_t graph = nullptr;       // the captured decode step
    hipGraphExec_t graph_ft = nullptr;    // ... with the token kernel's work in layer 0's attention kernel (GSV_STEP_FUSED_TOKEN)
    // WINDOWS of 2 .. kMaxWin steps captured as ONE graph each ([fused token][steps]; round 6): between two hipGraphLaunch'es of a one-step
    // graph the device idles ~8 us (rocprofv3 timeline, tools/step_gaps.py), inside a graph node follows node; a check window of five
    // steps replayed as one graph pays that once instead of five times
    static constexpr int kMaxWin = 8;
    hipGraphExec_t wgraph[2][kMaxWin + 1] = {};
    // staging of a refill that runs on ANOTHER stream while the decode step keeps replaying on the caller's
    // (gsv_t2s_prefill_slots_staged / gsv_t2s_commit_slots): per-slot state the step also writes must not be
    // touched by the prompt pass; it lands here and the commit, ordered on the step's stream, moves it over
    int64_t *sg_kv = nullptr, *sg_x = nullptr;
    int32_t *sg_step = nullptr, *sg_eos = nullptr;
    float *sg_logits = nullptr, *sg_hidden = nullptr;
    TokPart* sg_tok = nullptr;
};

void t2s_drop_graphs(T2SBound& b) {
    if (b.graph) { (void)hipGraphExecDestroy(b.graph); b.graph = nullptr; }
    if (b.graph_ft) { (void)hipGraphExecDestroy(b.graph_ft); b.graph_ft = nullptr; }
    for (auto& row : b.wgraph)
        for (hipGraphExec_t& g : row)
            if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
}

void t2s_free_staging(T2SBound& b) {
    for (void* p : {(void*)b.sg_kv, (void*)b.sg_x, (void*)b.sg_step, (void*)b.sg_eos, (void*)b.sg_logits, (void*)b.sg_hidden, (void*)b.sg_tok})
        if (p) (void)gsv_dev_free(p);
    b.sg_kv = b.sg_x = nullptr; b.sg_step = b.sg_eos = nullptr; b.sg_logits = b.sg_hidden = nullptr; b.sg_tok = nullptr;
}

struct gsv_t2s {
    gsv_t2s_config cfg;
    std::vector<T2SLayer> layers;
    void* predict = nullptr;  // WT [V][512]
    float *emb_audio = nullptr, *emb_text = nullptr, *pe_audio = nullptr, *pe_text = nullptr;
    PackedConv g_bert;
    unsigned have_io = 0;
    bool finalized = false;
    bool fp8 = false;              // GSV_FP8: e4m3 QKV / FFN weights in the batched step (everything else as GSV_BF16)
    int batched_min = 0;           // batch size from which the step is the batched chain
    int nt_from_layer = 0x7fffffff; // fp32 handles: layers from this one on load their weights non-temporally (t2s_attn_kernel's NT note)
    unsigned dbg_skip = 0;         // GSV_BSTEP_SKIP (tuning aid): bit i drops launch K(i+1) of the batched chain -- timing only
    std::map<int, T2SBound> bound;
    // scratch sized for the largest bound batch
    int scratch_b = 0;
    float *xcur = nullptr, *xbuf = nullptr, *x1buf = nullptr, *ypart = nullptr, *zpart = nullptr;
    TokPart* tokpart = nullptr;
    hipStream_t cap_stream = nullptr;
    unsigned long long* dbg = nullptr;
    gsv_arena::Arena arena;
    bool use_arena = true;
};
#define GSV_ARENA_SCOPE(h) gsv_arena::Scope arena_scope_((h)->use_arena ? &(h)->arena : nullptr)

namespace {

// e4m3 fragments + per-output-channel scales of one [cout][cin] linear (GSV_FP8 handles)
int t2s_pack_fp8(const float* data, int cout, int cin, void** frag, float** scale, hipStream_t st) {
    const int mtiles = cdiv(cout, 32);
    if (!*scale) HIPCHK(gsv_dev_malloc(scale, sizeof(float) * mtiles * 32));
    if (!*frag) HIPCHK(gsv_dev_malloc(frag, (size_t)mtiles * (cin / 32) * 64 * 16));
    HIPCHK(hipMemsetAsync(*scale, 0, sizeof(float) * mtiles * 32, st));
    hipLaunchKernelGGL(fp8_row_scale_kernel, dim3(cdiv(cout, 4)), dim3(256), 0, st, data, cin, *scale, cout);
    hipLaunchKernelGGL(fp8_pack_kernel, dim3(1024), dim3(256), 0, st, data, (const float*)*scale, (uint32_t*)*frag, cout, cin, mtiles);
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

// GSV_FP8 handles: the paired 16 x 16 x 32 e4m3 fragments of t2s_small.h (scales from t2s_pack_fp8, which runs first)
int t2s_pack16_f8(void** dst, const float* data, const float* scale, int N, int K, hipStream_t st) {
    if (!*dst) HIPCHK(gsv_dev_malloc(dst, (size_t)N * K));
    hipLaunchKernelGGL(pack16_f8_kernel, dim3(1024), dim3(256), 0, st, data, scale, (uint32_t*)*dst, N, K);
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

// bf16 handles: the 16 x 16 x 32 fragment order of t2s_small.h beside the 32 x 32 x 16 one
template <typename WT>
int t2s_pack16(void** dst, const float* data, int N, int K, hipStream_t st) {
    if (sizeof(WT) != 2) return GSV_OK;
    if (!*dst) HIPCHK(gsv_dev_malloc(dst, sizeof(bf16_t) * (size_t)N * K));
    hipLaunchKernelGGL(pack16_kernel, dim3(1024), dim3(256), 0, st, data, (bf16_t*)*dst, N, K);
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

template <typename WT>
int t2s_load_layer_tensor(gsv_t2s* h, int l, const std::string& key, const float* data, int64_t numel, hipStream_t st) {
    T2SLayer& L = h->layers[l];
    auto want = [&](int64_t n) { return numel == n ? GSV_OK : fail(GSV_ERR_ARG, "layer %d %s: numel %lld, expected %lld", l, key.c_str(), (long long)numel, (long long)n); };
    auto copy_f32 = [&](float** dst, int64_t n, unsigned bit) -> int {
        if (int rc = want(n)) return rc;
        if (!*dst) HIPCHK(gsv_dev_malloc(dst, sizeof(float) * n));
        HIPCHK(hipMemcpyAsync(*dst, data, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
        L.have |= bit;
        return GSV_OK;
    };
    if (key == "qkv.weight") {
        if (int rc = want(3LL * kD * kD)) return rc;
        if (!L.wqkv_p) HIPCHK(gsv_dev_malloc(&L.wqkv_p, sizeof(WT) * 3 * kD * kD));
        hipLaunchKernelGGL((pack_qkv_panel_kernel<WT>), dim3(1024), dim3(256), 0, st, data, (WT*)L.wqkv_p);
        float* keep = L.g_qkv.bias; L.g_qkv.bias = nullptr;
        free_conv(L.g_qkv);
        if (int rc = pack_conv<WT>(L.g_qkv, data, 3 * kD, kD, 1, kD, 1, 0, 1, 0, 0, nullptr, 1.f, st)) return rc;
        L.g_qkv.bias = keep;
        if (h->fp8) if (int rc = t2s_pack_fp8(data, 3 * kD, kD, &L.f8_qkv, &L.s_qkv, st)) return rc;
        if (h->fp8) if (int rc = t2s_pack16_f8(&L.p8_qkv, data, L.s_qkv, 3 * kD, kD, st)) return rc;
        if (int rc = t2s_pack16<WT>(&L.p16_qkv, data, 3 * kD, kD, st)) return rc;
        L.have |= 1u << 0;
    } else if (key == "qkv.bias") {
        if (int rc = want(3 * kD)) return rc;
        if (!L.bqkv_p) HIPCHK(gsv_dev_malloc(&L.bqkv_p, sizeof(float) * 3 * kD));
        hipLaunchKernelGGL(pack_qkv_bias_kernel, dim3(6), dim3(256), 0, st, data, L.bqkv_p);
        if (!L.g_qkv.bias) HIPCHK(gsv_dev_malloc(&L.g_qkv.bias, sizeof(float) * 3 * kD));
        HIPCHK(hipMemcpyAsync(L.g_qkv.bias, data, sizeof(float) * 3 * kD, hipMemcpyDeviceToDevice, st));
        L.have |= 1u << 1;
    } else if (key == "out_proj.weight") {
        if (int rc = want((int64_t)kD * kD)) return rc;
        if (!L.wo_p) HIPCHK(gsv_dev_malloc(&L.wo_p, sizeof(WT) * kD * kD));
        hipLaunchKernelGGL((pack_col_panel_kernel<WT>), dim3(512), dim3(256), 0, st, data, (WT*)L.wo_p, kH, kDh);
        free_conv(L.g_out);
        if (int rc = pack_conv<WT>(L.g_out, data, kD, kD, 1, kD, 1, 0, 1, 0, 0, nullptr, 1.f, st)) return rc;
        if (int rc = t2s_pack16<WT>(&L.p16_out, data, kD, kD, st)) return rc;
        L.have |= 1u << 2;
    } else if (key == "out_proj.bias") {
        if (int rc = copy_f32(&L.bo, kD, 1u << 3)) return rc;
    } else if (key == "mlp.0.weight") {
        if (int rc = want((int64_t)kF * kD)) return rc;
        if (!L.w1) HIPCHK(gsv_dev_malloc(&L.w1, sizeof(WT) * kF * kD));
        hipLaunchKernelGGL((convert_kernel<WT>), dim3(1024), dim3(256), 0, st, data, (WT*)L.w1, (size_t)kF * kD);
        free_conv(L.g_w1);
        if (int rc = pack_conv<WT>(L.g_w1, data, kF, kD, 1, kD, 1, 0, 1, 0, 0, nullptr, 1.f, st)) return rc;
        if (h->fp8) if (int rc = t2s_pack_fp8(data, kF, kD, &L.f8_w1, &L.s_w1, st)) return rc;
        if (h->fp8) if (int rc = t2s_pack16_f8(&L.p8_w1, data, L.s_w1, kF, kD, st)) return rc;
        if (int rc = t2s_pack16<WT>(&L.p16_w1, data, kF, kD, st)) return rc;
        L.have |= 1u << 4;
    } else if (key == "mlp.0.bias") {
        if (int rc = copy_f32(&L.b1, kF, 1u << 5)) return rc;
    } else if (key == "mlp.2.weight") {
        if (int rc = want((int64_t)kD * kF)) return rc;
        if (!L.w2_p) HIPCHK(gsv_dev_malloc(&L.w2_p, sizeof(WT) * kD * kF));
        hipLaunchKernelGGL((pack_col_panel_kernel<WT>), dim3(1024), dim3(256), 0, st, data, (WT*)L.w2_p, kNJ, kFJ);
        {
            if (!L.w2_p64) HIPCHK(gsv_dev_malloc(&L.w2_p64, sizeof(WT) * kD * kF));
            hipLaunchKernelGGL((pack_col_panel_kernel<WT>), dim3(1024), dim3(256), 0, st, data, (WT*)L.w2_p64, kNJFine, kF / kNJFine);
        }
        free_conv(L.g_w2);
        if (int rc = pack_conv<WT>(L.g_w2, data, kD, kF, 1, kF, 1, 0, 1, 0, 0, nullptr, 1.f, st)) return rc;
        if (h->fp8) if (int rc = t2s_pack_fp8(data, kD, kF, &L.f8_w2, &L.s_w2, st)) return rc;
        if (h->fp8) if (int rc = t2s_pack16_f8(&L.p8_w2, data, L.s_w2, kD, kF, st)) return rc;
        if (int rc = t2s_pack16<WT>(&L.p16_w2, data, kD, kF, st)) return rc;
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
        if (!*dst) HIPCHK(gsv_dev_malloc(dst, sizeof(float) * n));
        HIPCHK(hipMemcpyAsync(*dst, data, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
        h->have_io |= bit;
        return GSV_OK;
    };
    if (name == "ar_predict_layer.weight") {
        if (numel != (int64_t)c.vocab * kD) return fail(GSV_ERR_ARG, "%s: bad numel", name.c_str());
        if (!h->predict) HIPCHK(gsv_dev_malloc(&h->predict, sizeof(WT) * c.vocab * kD));
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
        if (p) (void)gsv_dev_free(p);
    HIPCHK(gsv_dev_malloc(&h->xcur, sizeof(float) * B * kD));
    HIPCHK(gsv_dev_malloc(&h->xbuf, sizeof(float) * B * kD));
    HIPCHK(gsv_dev_malloc(&h->x1buf, sizeof(float) * B * kD));
    HIPCHK(gsv_dev_malloc(&h->ypart, sizeof(float) * B * kH * kD));
    HIPCHK(gsv_dev_malloc(&h->zpart, sizeof(float) * B * kNJFine * kD));   // 64 fp32 slice partials per sequence at most
    HIPCHK(gsv_dev_malloc(&h->tokpart, sizeof(TokPart) * B * kNP));
    HIPCHK(hipMemset(h->tokpart, 0, sizeof(TokPart) * B * kNP));
    h->scratch_b = B;
    // graphs captured against the old scratch pointers are stale
    for (auto& kv : h->bound) t2s_drop_graphs(kv.second);
    return GSV_OK;
}

template <typename WT>
void t2s_launch_attn(gsv_t2s* h, const gsv_t2s_state& s, int l, const float* xsrc, hipStream_t st, bool fused_token = false) {
    const int B = s.batch, T = s.max_kv;
    const size_t lds = 0;  // static LDS only: scores never leave registers
    const size_t layer_elems = (size_t)B * kH * T * kDh;
    T2SLayer& L = h->layers[l];
    AttnArgs<WT> a;
    a.xdirect = xsrc;
    a.zpart = (const typename Geo<WT>::PT*)h->zpart;
    a.b2 = l ? h->layers[l - 1].b2 : nullptr;
    a.x1 = h->x1buf;
    a.ln2g = l ? h->layers[l - 1].ln2g : nullptr;
    a.ln2b = l ? h->layers[l - 1].ln2b : nullptr;
    a.xout = h->xbuf;
    a.wqkv = (const WT*)L.wqkv_p; a.bqkv = L.bqkv_p; a.wo = (const WT*)L.wo_p;
    a.kc = (WT*)s.k_cache + (size_t)l * layer_elems;
    a.vc = (WT*)s.v_cache + (size_t)l * layer_elems;
    a.kv_len = s.kv_len; a.T = T; a.ypart = (typename Geo<WT>::PT*)h->ypart; a.dbg = (l == h->cfg.n_layer - 1) ? h->dbg : nullptr;
    if (l == 0 && fused_token) {
        StepTok& k = a.tk;
        k.tokpart = h->tokpart; k.tok_override = s.tok_override; k.ctl = s.ctl; k.x_len = s.x_len; k.pre_tokens = s.pre_tokens;
        k.seen = s.seen; k.step = s.step; k.eos_at = s.eos_at; k.emb = h->emb_audio; k.pe = h->pe_audio; k.eos_host = eos_host_of(s);
        k.V = h->cfg.vocab; k.eos = h->cfg.eos; k.n_pos = h->cfg.n_pos;
    }
    if (B > 16) {   // two sequences per block: one round of 1024-thread blocks up to 32 sequences (t2s_decode_multi.h)
        const size_t ml = sizeof(float) * attn_multi_lds_floats<2>();
        if (l == 0 && fused_token) hipLaunchKernelGGL((t2s_attn_multi_kernel<WT, 2, 2>), dim3(kH, cdiv(B, 2)), dim3(kNT), ml, st, a, B);
        else if (l == 0) hipLaunchKernelGGL((t2s_attn_multi_kernel<WT, 0, 2>), dim3(kH, cdiv(B, 2)), dim3(kNT), ml, st, a, B);
        else hipLaunchKernelGGL((t2s_attn_multi_kernel<WT, 1, 2>), dim3(kH, cdiv(B, 2)), dim3(kNT), ml, st, a, B);
        return;
    }
    // K/V rows non-temporal from 4 sequences on: weights (152 MB) + a step's K/V rows (B x 49 KB x kv) then exceed what the Infinity
    // Cache holds (profiles/r03_kv_nontemporal.txt: at 1 sequence the hint costs 6 % at any kv -- everything fits; at 2 it pays only
    // beyond kv ~900; at 4 it is neutral at kv 250 and worth 3-5 % at kv 550-950; 0.354 -> 0.334 ms at 8, 0.411 -> 0.380 at 16).
    // GSV_SEQ_KV_NT_MIN_B moves the switch.
    static const int kvnt_b = getenv("GSV_SEQ_KV_NT_MIN_B") ? atoi(getenv("GSV_SEQ_KV_NT_MIN_B")) : 4;
    if (sizeof(WT) == 2 && B >= kvnt_b) {
        if (l == 0 && fused_token) hipLaunchKernelGGL((t2s_attn_kernel<WT, 2, kNJ, false, true>), dim3(kH, B), dim3(kNT), lds, st, a);
        else if (l == 0) hipLaunchKernelGGL((t2s_attn_kernel<WT, 0, kNJ, false, true>), dim3(kH, B), dim3(kNT), lds, st, a);
        else if (ffn_slices<WT>(B) == kNJFine) hipLaunchKernelGGL((t2s_attn_kernel<WT, 1, kNJFine, false, true>), dim3(kH, B), dim3(kNT), lds, st, a);
        else hipLaunchKernelGGL((t2s_attn_kernel<WT, 1, kNJ, false, true>), dim3(kH, B), dim3(kNT), lds, st, a);
        return;
    }
    if (l == 0 && fused_token) hipLaunchKernelGGL((t2s_attn_kernel<WT, 2>), dim3(kH, B), dim3(kNT), lds, st, a);
    else if (l == 0) hipLaunchKernelGGL((t2s_attn_kernel<WT, 0>), dim3(kH, B), dim3(kNT), lds, st, a);
    else if (ffn_slices<WT>(B) == kNJFine && sizeof(WT) == 4 && l >= h->nt_from_layer) hipLaunchKernelGGL((t2s_attn_kernel<WT, 1, kNJFine, sizeof(WT) == 4>), dim3(kH, B), dim3(kNT), lds, st, a);
    else if (ffn_slices<WT>(B) == kNJFine) hipLaunchKernelGGL((t2s_attn_kernel<WT, 1, kNJFine>), dim3(kH, B), dim3(kNT), lds, st, a);
    else if (sizeof(WT) == 4 && l >= h->nt_from_layer) hipLaunchKernelGGL((t2s_attn_kernel<WT, 1, kNJ, sizeof(WT) == 4>), dim3(kH, B), dim3(kNT), lds, st, a);
    else hipLaunchKernelGGL((t2s_attn_kernel<WT, 1>), dim3(kH, B), dim3(kNT), lds, st, a);
}

template <typename WT>
void t2s_launch_ffn(gsv_t2s* h, const gsv_t2s_state& s, int l, hipStream_t st) {
    T2SLayer& L = h->layers[l];
    FfnArgs<WT> f;
    f.ypart = (const typename Geo<WT>::PT*)h->ypart; f.bo = L.bo; f.x = h->xbuf; f.ln1g = L.ln1g; f.ln1b = L.ln1b; f.x1out = h->x1buf;
    f.w1 = (const WT*)L.w1; f.b1 = L.b1; f.w2p = (const WT*)L.w2_p; f.zpart = (typename Geo<WT>::PT*)h->zpart; f.dbg = (l == h->cfg.n_layer - 1) ? h->dbg : nullptr;
    const int B = s.batch;
    static const int ffn_single_max_b = getenv("GSV_FFN_SINGLE_MAX_B") ? atoi(getenv("GSV_FFN_SINGLE_MAX_B")) : 8;   // tuning aid
    bool four = false;
#ifdef GSV_AB_KERNELS                   // four sequences per block (bf16 handles; spills 16 registers): an A/B kernel, not in the shipped library -- above 16
    if constexpr (sizeof(WT) == 2) {     // sequences a bf16 handle runs the batched chain, and the per-sequence path behind it (caches beyond 1024 positions) R = 2
        if (B > 16) { hipLaunchKernelGGL((t2s_ffn_multi_kernel<WT, 4>), dim3(kNJ, cdiv(B, 4)), dim3(kNT), sizeof(float) * ffn_multi_lds_floats<4>(), st, f, B); four = true; }
    }
#endif
    if (four) {}
    else if (B > ffn_single_max_b) hipLaunchKernelGGL((t2s_ffn_multi_kernel<WT, 2>), dim3(kNJ, cdiv(B, 2)), dim3(kNT), sizeof(float) * ffn_multi_lds_floats<2>(), st, f, B);
    else if (ffn_slices<WT>(B) == kNJFine) {
        f.w2p = (const WT*)L.w2_p64;
        if (sizeof(WT) == 4 && l >= h->nt_from_layer) hipLaunchKernelGGL((t2s_ffn_kernel<WT, kNJFine, sizeof(WT) == 4>), dim3(kNJFine, B), dim3(kNT), 0, st, f);
        else hipLaunchKernelGGL((t2s_ffn_kernel<WT, kNJFine>), dim3(kNJFine, B), dim3(kNT), 0, st, f);
    } else if (sizeof(WT) == 4 && l >= h->nt_from_layer) hipLaunchKernelGGL((t2s_ffn_kernel<WT, kNJ, sizeof(WT) == 4>), dim3(kNJ, B), dim3(kNT), 0, st, f);
    else hipLaunchKernelGGL((t2s_ffn_kernel<WT>), dim3(kNJ, B), dim3(kNT), 0, st, f);
}

// the R-sequences-per-block kernels use more than 64 KB of dynamic LDS
template <typename WT>
int t2s_multi_lds_attr() {
    const int la = (int)(sizeof(float) * attn_multi_lds_floats<2>());
    HIPCHK(hipFuncSetAttribute((const void*)t2s_attn_multi_kernel<WT, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, la));
    HIPCHK(hipFuncSetAttribute((const void*)t2s_attn_multi_kernel<WT, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, la));
    HIPCHK(hipFuncSetAttribute((const void*)t2s_attn_multi_kernel<WT, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, la));
    HIPCHK(hipFuncSetAttribute((const void*)t2s_ffn_multi_kernel<WT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * ffn_multi_lds_floats<2>())));
#ifdef GSV_AB_KERNELS
    if constexpr (sizeof(WT) == 2)      // four sequences per block: bf16 handles only (t2s_launch_ffn); no fp32 instantiation exists
        HIPCHK(hipFuncSetAttribute((const void*)t2s_ffn_multi_kernel<WT, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * ffn_multi_lds_floats<4>())));
#endif
    return GSV_OK;
}

// the transformer stack for one token per slot; x from `xsrc` [B][512]
template <typename WT>
int t2s_layers(gsv_t2s* h, const gsv_t2s_state& s, const float* xsrc, hipStream_t st, bool fused_token = false) {
    for (int l = 0; l < h->cfg.n_layer; ++l) {
        t2s_launch_attn<WT>(h, s, l, xsrc, st, fused_token);
        t2s_launch_ffn<WT>(h, s, l, st);
    }
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

template <typename WT>
int t2s_logits(gsv_t2s* h, const gsv_t2s_state& s, int mode, const float* hdirect, int slot0, int nrows, int vlimit,
               int bump, hipStream_t st, const int32_t* slots = nullptr, const T2SBound* staged = nullptr) {
    const T2SLayer& L = h->layers.back();
    LogitsArgs<WT> a;
    a.hdirect = hdirect; a.zpart = (const typename Geo<WT>::PT*)h->zpart; a.b2 = L.b2; a.x1 = h->x1buf; a.ln2g = L.ln2g; a.ln2b = L.ln2b;
    a.wp = (const WT*)h->predict; a.V = h->cfg.vocab; a.eos = h->cfg.eos; a.vlimit = vlimit; a.slot0 = slot0; a.slots = slots;
    a.step = s.step; a.ctl = s.ctl; a.fctl = s.fctl; a.seen = s.seen; a.logits = s.logits; a.hidden = s.hidden;
    a.tokpart = h->tokpart; a.kv_len = s.kv_len; a.bump = bump;
    if (staged) { a.step = staged->sg_step; a.logits = staged->sg_logits; a.hidden = staged->sg_hidden; a.tokpart = staged->sg_tok; a.kv_len = staged->sg_kv; }
    if (mode == 0) hipLaunchKernelGGL((t2s_logits_kernel<WT, 0>), dim3(kNP, nrows), dim3(kNT), 0, st, a);
    else if (ffn_slices<WT>(s.batch) == kNJFine) hipLaunchKernelGGL((t2s_logits_kernel<WT, 1, kNJFine>), dim3(kNP, nrows), dim3(kNT), 0, st, a);
    else hipLaunchKernelGGL((t2s_logits_kernel<WT, 1>), dim3(kNP, nrows), dim3(kNT), 0, st, a);
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

int t2s_token(gsv_t2s* h, const gsv_t2s_state& s, int advance, hipStream_t st) {
    TokenArgs a;
    a.tokpart = h->tokpart; a.tok_override = s.tok_override; a.ctl = s.ctl; a.kv_len = s.kv_len; a.x_len = s.x_len;
    a.pre_tokens = s.pre_tokens; a.seen = s.seen; a.step = s.step; a.eos_at = s.eos_at; a.eos_host = eos_host_of(s); a.emb = h->emb_audio;
    a.pe = h->pe_audio; a.xcur = h->xcur; a.T = s.max_kv; a.V = h->cfg.vocab; a.eos = h->cfg.eos; a.n_pos = h->cfg.n_pos;
    a.advance = advance;
    a.logits = s.logits; a.fctl = s.fctl;
    hipLaunchKernelGGL(t2s_token_kernel, dim3(s.batch), dim3(256), 0, st, a);
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

// batched step (bf16, B >= kBatchedMin): the prompt GEMM chain on B rows + one attention block per (head, sequence)
// From this many sequences on, the bf16 / fp8 step is the batched chain (weights streamed once per step, five launches per
// layer: t2s_small.h's 16 x 16 tiles up to 64 rows, t2s_batch.h's 32 x 32 tiles above and for fp8) instead of the
// two-launches-per-layer kernels of t2s_decode.h.  Measured step time, bf16, kv 150-250 (ms; profiles/r03_chain_small.txt):
//   sequences                     8      16     17     24     32     33     48     64     128    256
//   2 launches/layer              0.36   0.41   0.59   0.62   0.64   --     --     --     --     --      (17-32: 2 / 4 sequences per block)
//   chain, 32 x 32 tiles (r02)    --     --     0.82   0.83   0.85   0.85   0.87   0.91   1.09   1.55
//   chain, 16 x 16 tiles          --     --     0.54   0.57   0.59   0.61   0.62   0.65   --     --
// GSV_BATCHED_MIN overrides it at handle creation (bench / tuning aid).
constexpr int kBatchedMinDefault = 17;
constexpr int kWideMinM = 3072;       // prompt-pass rows from which the chain's GEMMs are bgemm_wide_kernel (profiles/r04_prompt_pass_wide.txt)
constexpr size_t kPrefillLdsMax = 160 * 1024;
constexpr int kNtFromLayerF32 = 9;     // fp32 handles: 9 layers (114 MB; best of 6..14 at kv ~400, within 1 % of the best at kv ~200) + the K/V rows of a step stay in the Infinity Cache: 0.490 -> 0.440 ms per step (profiles/r03_f32_nontemporal_layers.txt)

// The 5-launches-per-layer chain of t2s_batch.h on M rows (decode: one row per sequence; prompt pass: nrows * l_max
// rows).  x0 [M][512] fp32 is the input of layer 0 and is overwritten with each layer's input (the residual of the
// out-proj); on return `y2` holds the LAST layer's pre-LayerNorm2 rows.  `attn_launch(l)` runs the attention of layer l
// from `qkv` into `attn`.
struct ChainBufs {
    float *qkv, *attn, *y1, *y2, *x1;   // [M][1536], [M][512] x 4
    void* hid;                          // [M][2048] bf16 | e4m3
};

template <typename AttnFn>
int t2s_gemm_chain(gsv_t2s* h, int M, float* x0, const ChainBufs& c, bool f8, AttnFn attn_launch, hipStream_t st, bool prompt = false) {
    const int rtiles = cdiv(M, 32);
    const unsigned skip = h->dbg_skip;
    // A prompt pass re-reads its X tile once per 32-column tile at 32 x 32 output per block (9 056 rows: 870 MB per QKV
    // launch), so there a block walks 2 / 4 / 8 column tiles with its X rows -- and their LayerNorm -- in registers.  Measured
    // prompt pass, ms, blocks of 1 / 2 / 4 / 8 column tiles: 241 rows 1.18 / 1.13 / 1.41 / 2.00; 980 rows 2.00 / 1.60 / 1.64 /
    // 2.12; 1 968 rows 3.19 / 2.43 / 2.10 / 2.39; 4 000 rows 5.91 / 4.37 / 3.51 / 3.31; 9 056 rows 12.65 / - / - / 7.10.
    // The decode step (one row per sequence) keeps one tile per block: it lives on launch latency, not on bytes.
    static const int force_cpb = getenv("GSV_CHAIN_CPB") ? atoi(getenv("GSV_CHAIN_CPB")) : 0;   // scan aid
    int cpb = 1;
    if (prompt) cpb = force_cpb > 0 ? force_cpb : (rtiles < 40 ? 2 : (rtiles < 100 ? 4 : 8));
    // per launch (scan aid): GSV_CPB_K1 / K3 / K4 / K5 = column tiles per block of the QKV / out-proj / W1 / W2 launch of a prompt pass
    static const int cpb_k[4] = {getenv("GSV_CPB_K1") ? atoi(getenv("GSV_CPB_K1")) : 0, getenv("GSV_CPB_K3") ? atoi(getenv("GSV_CPB_K3")) : 0,
                                 getenv("GSV_CPB_K4") ? atoi(getenv("GSV_CPB_K4")) : 0, getenv("GSV_CPB_K5") ? atoi(getenv("GSV_CPB_K5")) : 0};
    auto run = [&](auto kern, int nthreads, BGemmArgs ba, int which = -1) {
        // per launch (profiles/r04_prompt_pass_cpb.txt): the out-proj and W2 launches have 16 column tiles only -- at up to ~20 row tiles (one or two
        // prompts) a tile per block (128 blocks pulling half the weights each) is 0.95 -> 0.88 ms per pass; QKV and W1 at 21-39 row tiles: four
        int cb = cpb;
        if (prompt && which >= 0) {
            if (rtiles <= 20 && (which == 1 || which == 3)) cb = 1;
            if (rtiles > 20 && rtiles < 40 && (which == 0 || which == 2)) cb = 4;
            if (cpb_k[which] > 0) cb = cpb_k[which];
        }
        ba.cpb = cb;
        hipLaunchKernelGGL(kern, dim3(rtiles, cdiv(ba.mtiles, cb)), dim3(nthreads), 0, st, ba);
    };
    // few rows (the decode step at 17 .. kSmallMaxM sequences, bf16 operands): 16 x 16 tiles, one wave per channel tile (t2s_small.h)
    static const bool no_small = getenv("GSV_NO_SMALL_CHAIN") != nullptr;   // A/B switch
    static const int small_max = getenv("GSV_SMALL_MAX_M") ? atoi(getenv("GSV_SMALL_MAX_M")) : kSmallMaxM;   // tuning aid
    // NOT for a prompt pass, however few its rows (one 200-row prompt: TTFT 1.22 -> 0.98 ms on these kernels): the two tile shapes
    // sum k in different orders, and a request's K/V rows must not depend on how many prompts were packed into its pass
    // (tests/test_hip_t2s.py::test_prefill_into_scattered_slots_equals_one_by_one; the engine's ranks pack different sets)
    const bool small = !prompt && !no_small && M <= small_max && h->layers[0].p16_qkv != nullptr && (!f8 || h->layers[0].p8_qkv != nullptr);
    const int rt16 = cdiv(M, 16);
    // (the prompt pass's weight fragments non-temporal, so that they do not evict the decode step's copy: measured, no gain in the
    // cb workload, TTFT 1.11 -> 1.37 ms: not adopted)
    const bool nwv4 = M > 48;
    // a prompt pass of MANY rows (a packed batch of prompts): the same contraction, bit for bit, shaped for throughput
    // (bgemm_wide_kernel: 128 rows staged once per block, a wave per column tile over the full K, weights three groups ahead)
    static const int wide_min = getenv("GSV_WIDE_MIN_M") ? atoi(getenv("GSV_WIDE_MIN_M")) : kWideMinM;
    const bool wide = prompt && !f8 && M >= wide_min;
    auto run_wide = [&](auto kern, bool w2, BGemmArgs ba) -> int {   // its LDS attribute is set once, at gsv_t2s_finalize
        const int rgroups = cdiv(M, 32 * kWideRT), cgroups = cdiv(ba.mtiles, 4 * kWideTPW);
        // column groups per block: the block count that costs the fewest rounds of (stage the rows once + walk the groups)
        int gpb = 1;
        if (!w2) {
            double best = 1e30;
            for (int g = 1; g <= cgroups; ++g) {
                if (cgroups % g) continue;
                const double cost = (double)cdiv(rgroups * (cgroups / g), 256) * (1.0 + g);
                if (cost < best - 1e-9) { best = cost; gpb = g; }
            }
        }
        ba.cpb = gpb * 4 * kWideTPW;
        hipLaunchKernelGGL(kern, dim3(rgroups, cgroups / gpb), dim3(256), kWideLds, st, ba);
        return GSV_OK;
    };
    for (int l = 0; l < h->cfg.n_layer; ++l) {
        T2SLayer& L = h->layers[l];
        if (wide) {
            if (!(skip & 1)) {   // K1
                BGemmArgs g{};
                g.M = M; g.ldx = kD; g.W = (const uint4*)L.g_qkv.w; g.mtiles = 3 * kD / 32; g.cout = 3 * kD; g.bias = L.g_qkv.bias; g.Y = c.qkv; g.ldy = 3 * kD;
                if (l == 0) {
                    g.X = x0;
                    if (int rc = run_wide(bgemm_wide_kernel<PRO_NONE, float, bf16_t>, false, g)) return rc;
                } else {
                    g.X = c.y2; g.lng = h->layers[l - 1].ln2g; g.lnb = h->layers[l - 1].ln2b; g.xout = x0;
                    if (int rc = run_wide(bgemm_wide_kernel<PRO_LN, float, bf16_t>, false, g)) return rc;
                }
            }
            if (!(skip & 2)) attn_launch(l);
            if (!(skip & 4)) {   // K3
                BGemmArgs g{};
                g.M = M; g.X = c.attn; g.ldx = kD; g.W = (const uint4*)L.g_out.w; g.mtiles = kD / 32; g.cout = kD; g.bias = L.bo;
                g.res = x0; g.ldres = kD; g.Y = c.y1; g.ldy = kD;
                if (int rc = run_wide(bgemm_wide_kernel<PRO_NONE, bf16_t, float>, false, g)) return rc;      // the attention's bf16 rows
            }
            if (!(skip & 8)) {   // K4
                BGemmArgs g{};
                g.M = M; g.X = c.y1; g.ldx = kD; g.lng = L.ln1g; g.lnb = L.ln1b; g.xout = c.x1;
                g.W = (const uint4*)L.g_w1.w; g.mtiles = kF / 32; g.cout = kF; g.bias = L.b1; g.relu = 1; g.Y = c.hid; g.ldy = kF;
                if (int rc = run_wide(bgemm_wide_kernel<PRO_LN, float, bf16_t>, false, g)) return rc;
            }
            if (!(skip & 16)) {  // K5
                BGemmArgs g{};
                g.M = M; g.X = c.hid; g.ldx = kF; g.W = (const uint4*)L.g_w2.w; g.mtiles = kD / 32; g.cout = kD;
                g.bias = L.b2; g.res = c.x1; g.ldres = kD; g.Y = c.y2; g.ldy = kD;
                if (int rc = run_wide(bgemm_wide_kernel<PRO_NONE, bf16_t, float>, true, g)) return rc;
            }
            continue;
        }
        if (small) {
            if (!(skip & 1)) {   // K1
                SGemmArgs g{};
                g.M = M; g.W = (const uint4*)(f8 ? L.p8_qkv : L.p16_qkv); g.wscale = L.s_qkv; g.bias = L.g_qkv.bias; g.Y = c.qkv; g.ldy = 3 * kD;
                const dim3 grid(rt16, 3 * kD / 32);
                if (l == 0) {
                    g.X = x0;
                    if (f8) hipLaunchKernelGGL((sgemm_kernel<PRO_NONE, float, 2, true>), grid, dim3(128), 0, st, g);
                    else hipLaunchKernelGGL((sgemm_kernel<PRO_NONE, float, 2>), grid, dim3(128), 0, st, g);
                } else {
                    g.X = c.y2; g.lng = h->layers[l - 1].ln2g; g.lnb = h->layers[l - 1].ln2b; g.xout = x0;
                    // the LayerNorm prologue is per block: from 64 rows on, four channel tiles per block (4 rows of statistics per wave
                    // instead of 8) beat the doubled block count: 0.93 -> 0.86 ms per step at 128 sequences, equal at <= 32
                    const dim3 grid4(rt16, 3 * kD / 64);
                    if (f8 && nwv4) hipLaunchKernelGGL((sgemm_kernel<PRO_LN, float, 4, true>), grid4, dim3(256), 0, st, g);
                    else if (f8) hipLaunchKernelGGL((sgemm_kernel<PRO_LN, float, 2, true>), grid, dim3(128), 0, st, g);
                    else if (nwv4) hipLaunchKernelGGL((sgemm_kernel<PRO_LN, float, 4>), grid4, dim3(256), 0, st, g);
                    else hipLaunchKernelGGL((sgemm_kernel<PRO_LN, float, 2>), grid, dim3(128), 0, st, g);
                }
            }
            if (!(skip & 2)) attn_launch(l);
            if (!(skip & 4)) {   // K3 (bf16 on fp8 handles too)
                SGemmArgs g{};
                g.M = M; g.X = c.attn; g.W = (const uint4*)L.p16_out; g.bias = L.bo; g.res = x0; g.Y = c.y1; g.ldy = kD;
                hipLaunchKernelGGL((sgemm_kernel<PRO_NONE, float, 2>), dim3(rt16, kD / 32), dim3(128), 0, st, g);
            }
            if (!(skip & 8)) {   // K4
                SGemmArgs g{};
                g.M = M; g.X = c.y1; g.lng = L.ln1g; g.lnb = L.ln1b; g.xout = c.x1; g.W = (const uint4*)(f8 ? L.p8_w1 : L.p16_w1); g.wscale = L.s_w1;
                g.bias = L.b1; g.relu = 1; g.Y = c.hid; g.ldy = kF;
                if (f8 && nwv4) hipLaunchKernelGGL((sgemm_kernel<PRO_LN, fp8_t, 4, true>), dim3(rt16, kF / 64), dim3(256), 0, st, g);
                else if (f8) hipLaunchKernelGGL((sgemm_kernel<PRO_LN, fp8_t, 2, true>), dim3(rt16, kF / 32), dim3(128), 0, st, g);
                else if (nwv4) hipLaunchKernelGGL((sgemm_kernel<PRO_LN, bf16_t, 4>), dim3(rt16, kF / 64), dim3(256), 0, st, g);
                else hipLaunchKernelGGL((sgemm_kernel<PRO_LN, bf16_t, 2>), dim3(rt16, kF / 32), dim3(128), 0, st, g);
            }
            if (!(skip & 16)) {  // K5
                SGemmArgs g{};
                g.M = M; g.X = c.hid; g.W = (const uint4*)(f8 ? L.p8_w2 : L.p16_w2); g.wscale = L.s_w2; g.bias = L.b2; g.res = c.x1; g.Y = c.y2; g.ldy = kD;
                if (f8) hipLaunchKernelGGL(sgemm_k_kernel<true>, dim3(rt16, kD / 16), dim3(256), 0, st, g);
                else hipLaunchKernelGGL(sgemm_k_kernel<false>, dim3(rt16, kD / 16), dim3(256), 0, st, g);
            }
            continue;
        }
        if (!(skip & 1)) {   // K1: [LayerNorm2 of layer l-1] -> QKV
            BGemmArgs g{};
            g.M = M; g.ldx = kD; g.W = (const uint4*)(f8 ? L.f8_qkv : L.g_qkv.w); g.wscale = L.s_qkv; g.mtiles = 3 * kD / 32; g.cout = 3 * kD;
            g.bias = L.g_qkv.bias; g.Y = c.qkv; g.ldy = 3 * kD;
            // the prompt pass keeps qkv and the attention rows in bf16: its attention kernel rounds q / k / v to bf16 and the out-proj GEMM
            // rounds the attention rows to bf16 anyway, so the producers round instead -- the same values, half the bytes (decode step: fp32)
            if (l == 0) {
                g.X = x0;
                if (f8) run(bgemm_kernel<PRO_NONE, float, float, 4, true>, 256, g);
                else if (prompt) run(bgemm_kernel<PRO_NONE, float, bf16_t, 4, false, true>, 256, g, 0);
                else run(bgemm_kernel<PRO_NONE, float, float, 4, false, true>, 256, g);
            } else {
                g.X = c.y2; g.lng = h->layers[l - 1].ln2g; g.lnb = h->layers[l - 1].ln2b; g.xout = x0;
                if (f8) run(bgemm_kernel<PRO_LN, float, float, 4, true>, 256, g);
                else if (prompt) run(bgemm_kernel<PRO_LN, float, bf16_t, 4, false, true>, 256, g, 0);
                else run(bgemm_kernel<PRO_LN, float, float, 4, false, true>, 256, g);
            }
        }
        if (!(skip & 2)) attn_launch(l);
        if (!(skip & 4)) {   // K3: out-proj + bias + residual -> pre-LN1
            BGemmArgs g{};
            g.M = M; g.X = c.attn; g.ldx = kD; g.W = (const uint4*)L.g_out.w; g.mtiles = kD / 32; g.cout = kD; g.bias = L.bo;
            g.res = x0; g.ldres = kD; g.Y = c.y1; g.ldy = kD;
            if (prompt) run(bgemm_kernel<PRO_NONE, bf16_t, float, 4, false>, 256, g, 1);
            else run(bgemm_kernel<PRO_NONE, float, float, 4, false, true>, 256, g);
        }
        if (!(skip & 8)) {   // K4: [LayerNorm1] -> W1 + bias + ReLU
            BGemmArgs g{};
            g.M = M; g.X = c.y1; g.ldx = kD; g.lng = L.ln1g; g.lnb = L.ln1b; g.xout = c.x1;
            g.W = (const uint4*)(f8 ? L.f8_w1 : L.g_w1.w); g.wscale = L.s_w1; g.mtiles = kF / 32; g.cout = kF; g.bias = L.b1; g.relu = 1;
            g.Y = c.hid; g.ldy = kF;
            if (f8) run(bgemm_kernel<PRO_LN, float, fp8_t, 4, true>, 256, g);
            else run(bgemm_kernel<PRO_LN, float, bf16_t, 4, false, true>, 256, g, 2);
        }
        if (!(skip & 16)) {   // K5: W2 over the full K + bias + residual -> pre-LN2
            BGemmArgs g{};
            g.M = M; g.X = c.hid; g.ldx = kF; g.W = (const uint4*)(f8 ? L.f8_w2 : L.g_w2.w); g.wscale = L.s_w2; g.mtiles = kD / 32; g.cout = kD;
            g.bias = L.b2; g.res = c.x1; g.ldres = kD; g.Y = c.y2; g.ldy = kD;
            if (f8) run(bgemm_kernel<PRO_NONE, fp8_t, float, 16, true>, 1024, g);
            else run(bgemm_kernel<PRO_NONE, bf16_t, float, 16, false>, 1024, g, 3);
        }
    }
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

// batched decode step: the chain on B rows + one attention block per (head, sequence); final hidden states -> h->xbuf
template <typename WT>
int t2s_batched_layers(gsv_t2s* h, const gsv_t2s_state& s, hipStream_t st) {
    const int B = s.batch, T = s.max_kv;
    ChainBufs c;
    c.qkv = h->ypart;                                        // [B][1536]
    c.attn = c.qkv + (size_t)B * 3 * kD;                     // [B][512]
    c.y1 = c.attn + (size_t)B * kD;
    c.y2 = c.y1 + (size_t)B * kD;
    c.x1 = h->x1buf;
    c.hid = h->zpart;                                        // [B][2048] bf16 | e4m3
    const size_t layer_elems = (size_t)B * kH * T * kDh;
    auto attn = [&](int l) {
        BatchAttnArgs<WT> ba;
        ba.qkv = c.qkv; ba.kc = (WT*)s.k_cache + (size_t)l * layer_elems; ba.vc = (WT*)s.v_cache + (size_t)l * layer_elems;
        ba.kv_len = s.kv_len; ba.T = T; ba.out = c.attn; ba.dbg = (l == h->cfg.n_layer - 1) ? h->dbg : nullptr;
        // K/V rows non-temporal: measured better at every batch size the chain serves (0.567 -> 0.517 ms per step at 32 sequences,
        // 0.645 -> 0.595 at 64, 1.235 -> 1.168 at 256: profiles/r03_chain_small.txt); GSV_KV_NT_MIN_B moves the switch
        static const int kv_nt_min_b = getenv("GSV_KV_NT_MIN_B") ? atoi(getenv("GSV_KV_NT_MIN_B")) : 0;
        static const bool old_attn = getenv("GSV_OLD_BATTN") != nullptr;   // A/B switch: the first form (t2s_batch.h)
        static const int dup = getenv("GSV_BATTN_DUP") ? atoi(getenv("GSV_BATTN_DUP")) : 0;   // timing aid: launch it 1 + dup times (the repeats read warm K/V)
        for (int rep = 0; rep <= dup; ++rep)
        if (old_attn) {
            if (T <= 256) hipLaunchKernelGGL((t2s_batch_attn_kernel<4, false>), dim3(kH, B), dim3(256), 0, st, ba);
            else if (T <= 512) hipLaunchKernelGGL((t2s_batch_attn_kernel<8, false>), dim3(kH, B), dim3(256), 0, st, ba);
            else hipLaunchKernelGGL((t2s_batch_attn_kernel<16, false>), dim3(kH, B), dim3(256), 0, st, ba);
        } else if (B >= kv_nt_min_b) {
            if (T <= 256) hipLaunchKernelGGL((t2s_batch_attn2_kernel<4, true>), dim3(kH, B), dim3(256), 0, st, ba);
            else if (T <= 512) hipLaunchKernelGGL((t2s_batch_attn2_kernel<8, true>), dim3(kH, B), dim3(256), 0, st, ba);
            else hipLaunchKernelGGL((t2s_batch_attn2_kernel<16, true>), dim3(kH, B), dim3(256), 0, st, ba);
        } else if (T <= 256) hipLaunchKernelGGL((t2s_batch_attn2_kernel<4>), dim3(kH, B), dim3(256), 0, st, ba);
        else if (T <= 512) hipLaunchKernelGGL((t2s_batch_attn2_kernel<8>), dim3(kH, B), dim3(256), 0, st, ba);
        else hipLaunchKernelGGL((t2s_batch_attn2_kernel<16>), dim3(kH, B), dim3(256), 0, st, ba);
    };
    // x0 = xcur: the token kernel rewrites it at the start of every step, so the chain may use it as its residual buffer
    if (int rc = t2s_gemm_chain(h, B, h->xcur, c, h->fp8, attn, st)) return rc;
    const T2SLayer& LL = h->layers.back();
    hipLaunchKernelGGL(ln_rows_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, (const float*)c.y2, (const float*)LL.ln2g, (const float*)LL.ln2b, h->xbuf, B);
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

template <typename WT>
int t2s_step(gsv_t2s* h, const gsv_t2s_state& s, hipStream_t st, bool fused_token = false) {
    // the 2-launches-per-layer kernels with greedy / host-chosen tokens: layer 0's attention kernel does the token kernel's work
    if (fused_token && !(sizeof(WT) == 2 && s.batch >= h->batched_min && s.max_kv <= 1024)) {
        if (int rc = t2s_layers<WT>(h, s, h->xcur, st, true)) return rc;
        return t2s_logits<WT>(h, s, 1, nullptr, 0, s.batch, h->cfg.vocab, 1, st);
    }
    if (int rc = t2s_token(h, s, 1, st)) return rc;
    if constexpr (sizeof(WT) == 2) {
        if (s.batch >= h->batched_min && s.max_kv <= 1024) {
            if (int rc = t2s_batched_layers<WT>(h, s, st)) return rc;
            return t2s_logits<WT>(h, s, 0, h->xbuf, 0, s.batch, h->cfg.vocab, 1, st);
        }
    }
    if (int rc = t2s_layers<WT>(h, s, h->xcur, st)) return rc;
    return t2s_logits<WT>(h, s, 1, nullptr, 0, s.batch, h->cfg.vocab, 1, st);
}

template <typename WT>
int t2s_prefill_impl(gsv_t2s* h, T2SBound& bd, int slot0, int nrows, int l_max, float* xy, const int64_t* x_lens,
                     const int64_t* y_lens, void* ws, size_t ws_bytes, hipStream_t st, const int32_t* slots = nullptr, bool staged = false) {
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
    // LDS-staged attention: the whole prompt's K/V of one head sit in LDS.  Each numerics mode is gated by ITS kernel's
    // footprint only (fp32 parity mode: 4*(69*l_max + 136) B -> l_max <= 591; bf16: 144*ceil(l_max/32)*32 + ... -> l_max <= 1056,
    // i.e. every prompt the reference's largest default bucket (1024) accepts).  The dynamic-LDS attribute is raised once, at finalize.
    const size_t lds = sizeof(float) * ((size_t)l_max * 33 + (size_t)l_max * 32 + 4 * (size_t)l_max + 128 + 8);
    const int nkt_max = cdiv(l_max, 32);
    const size_t lds_mfma = (size_t)nkt_max * 32 * 80 + 128 * 80 + (size_t)32 * (nkt_max * 64 + 16);
    if (sizeof(WT) == 2 ? lds_mfma > kPrefillLdsMax : lds > kPrefillLdsMax)
        return fail(GSV_ERR_ARG, "prefill: prompt of %d positions exceeds the LDS-staged attention limit of the %s mode (%d positions)",
                    l_max, sizeof(WT) == 2 ? "bf16" : "fp32", sizeof(WT) == 2 ? 1056 : 591);
    if constexpr (sizeof(WT) == 2) {
        // bf16 mode: the same 5-launch chain as the batched decode step on M = nrows * l_max rows (LayerNorm in the
        // consuming GEMM's prologue, full-K W2), with flash attention on the matrix cores; bf16 operands also on GSV_FP8
        // handles (fp8 is the batched decode step's).  Workspace: qkv | attn | y1 | [y2 | x1 | hid(bf16)] in the FFN slot.
        ChainBufs c;
        c.qkv = qkv; c.attn = attn; c.y1 = ybuf; c.y2 = fbuf; c.x1 = fbuf + (size_t)M * kD; c.hid = fbuf + (size_t)2 * M * kD;
        auto attn_launch = [&](int l) {
            PrefillAttnMfmaArgs pm;
            pm.qkv = (const bf16_t*)qkv; pm.x_lens = x_lens; pm.y_lens = y_lens;
            pm.kc = (bf16_t*)s.k_cache + (size_t)l * layer_elems; pm.vc = (bf16_t*)s.v_cache + (size_t)l * layer_elems;
            pm.T = T; pm.slot0 = slot0; pm.slots = slots; pm.l_max = l_max; pm.out = (bf16_t*)attn;
            hipLaunchKernelGGL(t2s_prefill_attn_mfma_kernel, dim3(kH, nrows, cdiv(l_max, 128)), dim3(256), lds_mfma, st, pm);
        };
        if (int rc = t2s_gemm_chain(h, M, xy, c, false, attn_launch, st, true)) return rc;
        const T2SLayer& LL = h->layers.back();
        hipLaunchKernelGGL(ln_rows_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, (const float*)c.y2, (const float*)LL.ln2g, (const float*)LL.ln2b, xy, M);
        HIPCHK(hipGetLastError());
    } else {
        for (int l = 0; l < h->cfg.n_layer; ++l) {
            T2SLayer& L = h->layers[l];
            Epi e0; e0.fixed_order = true;      // fp32 prompt pass: one tile shape (= one summation order) for every row count
            if (int rc = run_conv<float, WT, float>(L.g_qkv, xy, kD, M, qkv, 3 * kD, M, e0, st)) return rc;
            PrefillAttnArgs<WT> pa;
            pa.qkv = qkv; pa.x_lens = x_lens; pa.y_lens = y_lens;
            pa.kc = (WT*)s.k_cache + (size_t)l * layer_elems; pa.vc = (WT*)s.v_cache + (size_t)l * layer_elems;
            pa.T = T; pa.slot0 = slot0; pa.slots = slots; pa.l_max = l_max; pa.qsplit = qsplit; pa.out = attn;
            {
                hipLaunchKernelGGL((t2s_prefill_attn_kernel<WT>), dim3(kH, nrows, qsplit), dim3(256), lds, st, pa);
            }
            Epi e1; e1.res = xy; e1.ld_res = kD; e1.fixed_order = true;
            // out_proj bias lives in the decode copy (L.bo); tapgemm bias pointer set per call
            PackedConv go = L.g_out; go.bias = L.bo;
            if (int rc = run_conv<float, WT, float>(go, attn, kD, M, ybuf, kD, M, e1, st)) return rc;
            hipLaunchKernelGGL(ln_rows_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, ybuf, L.ln1g, L.ln1b, xy, M);
            Epi e2; e2.act = ACT_RELU; e2.fixed_order = true;
            PackedConv g1 = L.g_w1; g1.bias = L.b1;
            if (int rc = run_conv<float, WT, float>(g1, xy, kD, M, fbuf, kF, M, e2, st)) return rc;
            Epi e3; e3.res = xy; e3.ld_res = kD; e3.fixed_order = true;
            PackedConv g2 = L.g_w2; g2.bias = L.b2;
            if (int rc = run_conv<float, WT, float>(g2, fbuf, kF, M, ybuf, kD, M, e3, st)) return rc;
            hipLaunchKernelGGL(ln_rows_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, ybuf, L.ln2g, L.ln2b, xy, M);
        }
    }
    PrefillFinishArgs fa;
    fa.hidden = xy; fa.x_lens = x_lens; fa.y_lens = y_lens; fa.hlast = hlast; fa.kv_len = s.kv_len; fa.x_len = s.x_len;
    fa.step = s.step; fa.eos_at = s.eos_at; fa.eos_host = bd.st.eos_host; fa.slot0 = slot0; fa.slots = slots; fa.l_max = l_max;
    if (staged) { fa.kv_len = bd.sg_kv; fa.x_len = bd.sg_x; fa.step = bd.sg_step; fa.eos_at = bd.sg_eos; fa.eos_host = nullptr; }
    hipLaunchKernelGGL(t2s_prefill_finish_kernel, dim3(nrows), dim3(128), 0, st, fa);
    HIPCHK(hipGetLastError());
    // first sample: logits[:, :-1] (t2s_model.py:417,613) -> EOS column dropped
    return t2s_logits<WT>(h, s, 0, hlast, slot0, nrows, h->cfg.vocab - 1, 0, st, slots, staged ? &bd : nullptr);
}

// staging -> live state of the listed slots (gsv_t2s_commit_slots)
struct CommitArgs {
    const int32_t* slots;
    const int64_t *sg_kv, *sg_x; const int32_t *sg_step, *sg_eos; const float *sg_logits, *sg_hidden; const TokPart* sg_tok;
    int64_t *kv_len, *x_len; int32_t *step, *eos_at, *eos_host; float *logits, *hidden; TokPart* tokpart;
    int V;
};
__global__ __launch_bounds__(256) void t2s_commit_kernel(CommitArgs a) {
    const int s = a.slots[blockIdx.x], tid = threadIdx.x;
    for (int v = tid; v < a.V; v += 256) a.logits[(size_t)s * a.V + v] = a.sg_logits[(size_t)s * a.V + v];
    for (int c = tid; c < kD; c += 256) a.hidden[(size_t)s * kD + c] = a.sg_hidden[(size_t)s * kD + c];
    if (tid < kNP) a.tokpart[(size_t)s * kNP + tid] = a.sg_tok[(size_t)s * kNP + tid];
    if (tid == 0) { a.kv_len[s] = a.sg_kv[s]; a.x_len[s] = a.sg_x[s]; a.step[s] = a.sg_step[s]; a.eos_at[s] = a.sg_eos[s]; eos_publish(a.eos_host, s, a.sg_eos[s]); }
}


// A prompt pass that ran AHEAD into another bound state's cache and staging (gsv_t2s_adopt_slots): its K/V rows and its staged
// state move into slots of the state the steps run on.  The slot pairs ride in the kernel arguments: no device slot list has to
// be built between two decode windows.
constexpr int kAdoptMax = 64;
struct AdoptArgs {
    short dst[kAdoptMax], src[kAdoptMax];
    long long ovr[kAdoptMax];         // tok_override of the adopting slot, or < 0: left alone
    const unsigned char *ks, *vs;     // source cache [n_layer][Bs][16][Ts][32]
    unsigned char *kd, *vd;           // destination cache [n_layer][Bd][16][Td][32]
    int Bs, Ts, Bd, Td, esz;
    const int64_t *sg_kv, *sg_x; const int32_t *sg_step, *sg_eos; const float *sg_logits, *sg_hidden; const TokPart* sg_tok;
    int64_t *kv_len, *x_len, *tok_override; int32_t *step, *eos_at, *eos_host; float *logits, *hidden; TokPart* tokpart;
    int V;
};
__global__ __launch_bounds__(256) void t2s_adopt_kv_kernel(AdoptArgs a) {
    const int lh = blockIdx.x, r = blockIdx.y, l = lh / kH, hd = lh % kH;
    const int ss = a.src[r], ds = a.dst[r];
    // rows [0, kv_len) of the panel are contiguous; a source slot no prompt pass has filled (kv_len 0 since the bind) copies nothing, and no
    // length read from the staging can run past either cache
    long long kv = a.sg_kv[ss];
    kv = kv < 0 ? 0 : (kv > a.Ts ? a.Ts : kv);
    kv = kv > a.Td ? a.Td : kv;
    const size_t n16 = (size_t)kv * kDh * a.esz / 16;
    const size_t so = ((((size_t)l * a.Bs + ss) * kH + hd) * a.Ts) * kDh * a.esz;
    const size_t d0 = ((((size_t)l * a.Bd + ds) * kH + hd) * a.Td) * kDh * a.esz;
    const uint4* sp = reinterpret_cast<const uint4*>((blockIdx.z ? a.vs : a.ks) + so);
    uint4* dp = reinterpret_cast<uint4*>((blockIdx.z ? a.vd : a.kd) + d0);
    for (size_t i = threadIdx.x; i < n16; i += 256) dp[i] = sp[i];
}
__global__ __launch_bounds__(256) void t2s_adopt_state_kernel(AdoptArgs a) {
    const int ss = a.src[blockIdx.x], s = a.dst[blockIdx.x], tid = threadIdx.x;
    for (int v = tid; v < a.V; v += 256) a.logits[(size_t)s * a.V + v] = a.sg_logits[(size_t)ss * a.V + v];
    for (int c = tid; c < kD; c += 256) a.hidden[(size_t)s * kD + c] = a.sg_hidden[(size_t)ss * kD + c];
    if (tid < kNP) a.tokpart[(size_t)s * kNP + tid] = a.sg_tok[(size_t)ss * kNP + tid];
    if (tid == 0) {
        a.kv_len[s] = a.sg_kv[ss]; a.x_len[s] = a.sg_x[ss]; a.step[s] = a.sg_step[ss]; a.eos_at[s] = a.sg_eos[ss];
        eos_publish(a.eos_host, s, a.sg_eos[ss]);
        if (a.ovr[blockIdx.x] >= 0) a.tok_override[s] = a.ovr[blockIdx.x];
    }
}

// Live slots of one stepped state move into slots of another (gsv_t2s_move_slots: the tail of a continuous-batching run continues on a
// smaller batch size): K/V rows [0, kv_len), the token history, the penalty set and everything a step reads of the previous one.
struct MoveArgs {
    short dst[kAdoptMax], src[kAdoptMax];
    const unsigned char *ks, *vs; unsigned char *kd, *vd;
    int Bs, Ts, Bd, Td, esz, V, n;
    const int64_t *s_kv, *s_x, *s_pre, *s_ovr; const int32_t *s_step, *s_eos; const float *s_logits, *s_hidden; const unsigned char* s_seen;
    int64_t *d_kv, *d_x, *d_pre, *d_ovr; int32_t *d_step, *d_eos, *d_eos_host; float *d_logits, *d_hidden; unsigned char* d_seen;
    TokPart* tokpart;                 // ONE array per handle, indexed by slot: source and destination rows may overlap
};
__global__ __launch_bounds__(256) void t2s_move_kv_kernel(MoveArgs a) {
    const int lh = blockIdx.x, r = blockIdx.y, l = lh / kH, hd = lh % kH;
    const int ss = a.src[r], ds = a.dst[r];
    long long kv = a.s_kv[ss];
    kv = kv < 0 ? 0 : (kv > a.Ts ? a.Ts : kv);
    kv = kv > a.Td ? a.Td : kv;
    const size_t n16 = (size_t)kv * kDh * a.esz / 16;
    const size_t so = ((((size_t)l * a.Bs + ss) * kH + hd) * a.Ts) * kDh * a.esz;
    const size_t d0 = ((((size_t)l * a.Bd + ds) * kH + hd) * a.Td) * kDh * a.esz;
    const uint4* sp = reinterpret_cast<const uint4*>((blockIdx.z ? a.vs : a.ks) + so);
    uint4* dp = reinterpret_cast<uint4*>((blockIdx.z ? a.vd : a.kd) + d0);
    for (size_t i = threadIdx.x; i < n16; i += 256) dp[i] = sp[i];
}
__global__ __launch_bounds__(256) void t2s_move_state_kernel(MoveArgs a) {
    const int tid = threadIdx.x;
    if (blockIdx.x == a.n) {          // the last block: the pending tokens, read whole before any is written (one array)
        __shared__ TokPart tp[kAdoptMax * kNP];
        for (int i = tid; i < a.n * kNP; i += 256) tp[i] = a.tokpart[(size_t)a.src[i / kNP] * kNP + i % kNP];
        __syncthreads();
        for (int i = tid; i < a.n * kNP; i += 256) a.tokpart[(size_t)a.dst[i / kNP] * kNP + i % kNP] = tp[i];
        return;
    }
    const int ss = a.src[blockIdx.x], s = a.dst[blockIdx.x];
    for (int v = tid; v < a.V; v += 256) { a.d_logits[(size_t)s * a.V + v] = a.s_logits[(size_t)ss * a.V + v]; a.d_seen[(size_t)s * a.V + v] = a.s_seen[(size_t)ss * a.V + v]; }
    for (int c = tid; c < kD; c += 256) a.d_hidden[(size_t)s * kD + c] = a.s_hidden[(size_t)ss * kD + c];
    const int np = (a.Ts < a.Td ? a.Ts : a.Td) + 1;
    for (int t = tid; t < np; t += 256) a.d_pre[(size_t)s * (a.Td + 1) + t] = a.s_pre[(size_t)ss * (a.Ts + 1) + t];
    if (tid == 0) {
        a.d_kv[s] = a.s_kv[ss]; a.d_x[s] = a.s_x[ss]; a.d_step[s] = a.s_step[ss]; a.d_eos[s] = a.s_eos[ss]; a.d_ovr[s] = a.s_ovr[ss];
        eos_publish(a.d_eos_host, s, a.s_eos[ss]);
    }
}

}  // namespace

// One class of decode-step kernels (all layers' launches of it) captured into a hipGraph and replayed `iters` times
// between two events on `st`: no host launch cost in the figure (an eager sweep is host-bound below ~3 us per launch).
// The result still contains the dependent-launch gap that every kernel of a real step pays too.
template <typename WT>
static int t2s_time_impl(gsv_t2s* h, T2SBound* b, int iters, float* out_ms, hipStream_t st) {
    struct Guard {      // whatever path leaves this function: no event, graph or executable graph stays behind
        hipEvent_t e0 = nullptr, e1 = nullptr;
        hipGraphExec_t exec = nullptr;
        ~Guard() {
            if (exec) (void)hipGraphExecDestroy(exec);
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } gd;
    HIPCHK(hipEventCreate(&gd.e0));
    HIPCHK(hipEventCreate(&gd.e1));
    const int NL = h->cfg.n_layer;
    for (int cls = 0; cls < 4; ++cls) {
        hipGraph_t g = nullptr;
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
        int rc = GSV_OK;                      // nothing between Begin and End returns: a stream left capturing is lost to the handle
        for (int l = 0; l < NL && !rc; ++l) {
            if (cls == 0) t2s_launch_attn<WT>(h, b->st, l, h->xcur, h->cap_stream);
            if (cls == 1) t2s_launch_ffn<WT>(h, b->st, l, h->cap_stream);
            if (cls == 2) rc = t2s_logits<WT>(h, b->st, 1, nullptr, 0, b->st.batch, h->cfg.vocab, 0, h->cap_stream);
            if (cls == 3) rc = t2s_token(h, b->st, 0, h->cap_stream);
        }
        hipError_t e = hipStreamEndCapture(h->cap_stream, &g);
        if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (e != hipSuccess) return fail(GSV_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
        e = hipGraphInstantiate(&gd.exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) { gd.exec = nullptr; return fail(GSV_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
        HIPCHK(hipGraphLaunch(gd.exec, st));                    // warm-up
        HIPCHK(hipEventRecord(gd.e0, st));
        for (int it = 0; it < iters; ++it) HIPCHK(hipGraphLaunch(gd.exec, st));
        HIPCHK(hipEventRecord(gd.e1, st));
        HIPCHK(hipEventSynchronize(gd.e1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, gd.e0, gd.e1));
        out_ms[cls] = ms / (float)(iters * NL);
        (void)hipGraphExecDestroy(gd.exec);
        gd.exec = nullptr;
    }
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
    if (cfg->dtype != GSV_F32 && cfg->dtype != GSV_BF16 && cfg->dtype != GSV_FP8) return fail(GSV_ERR_ARG, "bad dtype");
    gsv_t2s* h = new gsv_t2s();
    h->cfg = *cfg;
    if (cfg->dtype == GSV_FP8) { h->fp8 = true; h->cfg.dtype = GSV_BF16; }   // bf16 everywhere but the batched step's QKV / FFN
    h->batched_min = kBatchedMinDefault;
    if (const char* e = getenv("GSV_BATCHED_MIN")) h->batched_min = std::max(1, atoi(e));
    if (const char* e = getenv("GSV_BSTEP_SKIP")) h->dbg_skip = (unsigned)atoi(e);
    if (h->cfg.dtype == GSV_F32) h->nt_from_layer = kNtFromLayerF32;
    if (const char* e = getenv("GSV_NT_FROM_LAYER")) h->nt_from_layer = atoi(e);   // tuning aid
    if (getenv("GSV_NO_ARENA")) h->use_arena = false;
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
        t2s_drop_graphs(kv.second);
        t2s_free_staging(kv.second);
    }
    for (T2SLayer& L : h->layers) {
        for (void* p : {L.wqkv_p, L.wo_p, L.w1, L.w2_p, L.w2_p64, L.p16_qkv, L.p16_out, L.p16_w1, L.p16_w2, L.p8_qkv, L.p8_w1, L.p8_w2, (void*)L.bqkv_p, (void*)L.bo, (void*)L.b1, (void*)L.b2,
                        (void*)L.ln1g, (void*)L.ln1b, (void*)L.ln2g, (void*)L.ln2b, L.f8_qkv, L.f8_w1, L.f8_w2,
                        (void*)L.s_qkv, (void*)L.s_w1, (void*)L.s_w2})
            if (p) (void)gsv_dev_free(p);
        free_conv(L.g_qkv); L.g_out.bias = nullptr; free_conv(L.g_out); L.g_w1.bias = nullptr; free_conv(L.g_w1);
        L.g_w2.bias = nullptr; free_conv(L.g_w2);
    }
    for (void* p : {h->predict, (void*)h->emb_audio, (void*)h->emb_text, (void*)h->pe_audio, (void*)h->pe_text,
                    (void*)h->xcur, (void*)h->xbuf, (void*)h->x1buf, (void*)h->ypart, (void*)h->zpart, (void*)h->tokpart})
        if (p) (void)gsv_dev_free(p);
    free_conv(h->g_bert);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    gsv_arena::release(h->arena);
    delete h;
    return GSV_OK;
}

int gsv_t2s_load_tensor(gsv_t2s* h, const char* name, const float* data, int64_t numel, void* stream) {
    if (!h || !name || !data) return fail(GSV_ERR_ARG, "null argument");
    GSV_ARENA_SCOPE(h);
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
    GSV_ARENA_SCOPE(h);
    for (int l = 0; l < h->cfg.n_layer; ++l)
        if (h->layers[l].have != 0xfffu) return fail(GSV_ERR_STATE, "layer %d incomplete (mask 0x%x)", l, h->layers[l].have);
    if (h->have_io != 0x7fu) return fail(GSV_ERR_STATE, "embedding/predict tensors incomplete (mask 0x%x)", h->have_io);
    if (h->cfg.dtype == GSV_BF16) {
        HIPCHK(hipFuncSetAttribute((const void*)t2s_prefill_attn_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPrefillLdsMax));
        // the wide prompt-pass GEMMs (every instantiation t2s_batched_layers launches): once here, not per launch (5 x n_layer calls per pass)
        HIPCHK(hipFuncSetAttribute((const void*)bgemm_wide_kernel<PRO_NONE, float, bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWideLds));
        HIPCHK(hipFuncSetAttribute((const void*)bgemm_wide_kernel<PRO_LN, float, bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWideLds));
        HIPCHK(hipFuncSetAttribute((const void*)bgemm_wide_kernel<PRO_NONE, bf16_t, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWideLds));
        if (int rc = t2s_multi_lds_attr<bf16_t>()) return rc;
    } else {
        HIPCHK(hipFuncSetAttribute((const void*)t2s_prefill_attn_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPrefillLdsMax));
        if (int rc = t2s_multi_lds_attr<float>()) return rc;
    }
    HIPCHK(hipStreamSynchronize(S(stream)));
    h->finalized = true;
    return GSV_OK;
}

int gsv_t2s_bind_state(gsv_t2s* h, const gsv_t2s_state* st) {
    if (!h || !st) return fail(GSV_ERR_ARG, "null argument");
    GSV_ARENA_SCOPE(h);
    if (st->batch < 1 || st->max_kv < 2) return fail(GSV_ERR_ARG, "bad batch/max_kv");
    if (!st->k_cache || !st->v_cache || !st->kv_len || !st->x_len || !st->pre_tokens || !st->seen || !st->step ||
        !st->eos_at || !st->logits || !st->hidden || !st->tok_override || !st->ctl || !st->fctl)
        return fail(GSV_ERR_ARG, "state has null pointers");
    if (int rc = t2s_ensure_scratch(h, st->batch)) return rc;
    T2SBound& b = h->bound[st->batch];
    t2s_drop_graphs(b);
    static_cast<gsv_t2s_state&>(b.st) = *st;
    b.st.eos_host = nullptr;
    t2s_free_staging(b);
    const size_t B = (size_t)st->batch;
    HIPCHK(gsv_dev_malloc(&b.sg_kv, 8 * B)); HIPCHK(gsv_dev_malloc(&b.sg_x, 8 * B));
    HIPCHK(gsv_dev_malloc(&b.sg_step, 4 * B)); HIPCHK(gsv_dev_malloc(&b.sg_eos, 4 * B));
    HIPCHK(gsv_dev_malloc(&b.sg_logits, sizeof(float) * B * h->cfg.vocab)); HIPCHK(gsv_dev_malloc(&b.sg_hidden, sizeof(float) * B * kD));
    HIPCHK(gsv_dev_malloc(&b.sg_tok, sizeof(TokPart) * B * kNP));
    HIPCHK(hipMemset(b.sg_step, 0, 4 * B));
    HIPCHK(hipMemset(b.sg_kv, 0, 8 * B)); HIPCHK(hipMemset(b.sg_x, 0, 8 * B));
    HIPCHK(hipMemset(b.sg_eos, 0xff, 4 * B));      // -1: no EOS seen
    return GSV_OK;
}

int gsv_t2s_unbind_state(gsv_t2s* h, int batch) {
    if (!h) return fail(GSV_ERR_ARG, "null handle");
    GSV_ARENA_SCOPE(h);
    auto it = h->bound.find(batch);
    if (it == h->bound.end()) return GSV_OK;
    T2SBound& b = it->second;
    t2s_drop_graphs(b);
    t2s_free_staging(b);
    h->bound.erase(it);
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
    // split-K tiles whatever the row count: the wide tiles tapgemm picks from ~2 000 phoneme rows on sum K in another order, and a
    // request's embedded rows -- hence its K/V rows and first logits -- changed with the number of prompts packed beside it
    // (found in round 4: tests/test_hip_t2s.py::test_packed_prompt_pass_of_many_rows_equals_one_by_one_bf16)
    e.fixed_order = true;
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

int gsv_t2s_prefill_slots_staged(gsv_t2s* h, int batch, const int32_t* slots, int nrows, int l_max, float* xy, const int64_t* x_lens,
                                 const int64_t* y_lens, void* workspace, size_t workspace_bytes, void* stream) {
    if (!h || !h->finalized) return fail(GSV_ERR_STATE, "handle not finalized");
    T2SBound* b = t2s_find(h, batch);
    if (!b) return fail(GSV_ERR_STATE, "no state bound for batch %d", batch);
    if (!slots || nrows < 1 || nrows > batch) return fail(GSV_ERR_ARG, "prefill_slots_staged: need 1..batch rows and their slot list");
    if (l_max < 1 || l_max > b->st.max_kv - 1) return fail(GSV_ERR_ARG, "prompt of %d positions does not fit beside the parking row of the KV cache (%d)", l_max, b->st.max_kv);
    return h->cfg.dtype == GSV_BF16
               ? t2s_prefill_impl<bf16_t>(h, *b, 0, nrows, l_max, xy, x_lens, y_lens, workspace, workspace_bytes, S(stream), slots, true)
               : t2s_prefill_impl<float>(h, *b, 0, nrows, l_max, xy, x_lens, y_lens, workspace, workspace_bytes, S(stream), slots, true);
}

int gsv_t2s_commit_slots(gsv_t2s* h, int batch, const int32_t* slots, int nrows, void* stream) {
    if (!h || !h->finalized) return fail(GSV_ERR_STATE, "handle not finalized");
    T2SBound* b = t2s_find(h, batch);
    if (!b) return fail(GSV_ERR_STATE, "no state bound for batch %d", batch);
    if (!slots || nrows < 1 || nrows > batch) return fail(GSV_ERR_ARG, "commit_slots: need 1..batch slots");
    CommitArgs a;
    a.slots = slots; a.sg_kv = b->sg_kv; a.sg_x = b->sg_x; a.sg_step = b->sg_step; a.sg_eos = b->sg_eos; a.sg_logits = b->sg_logits;
    a.sg_hidden = b->sg_hidden; a.sg_tok = b->sg_tok; a.kv_len = b->st.kv_len; a.x_len = b->st.x_len; a.step = b->st.step; a.eos_at = b->st.eos_at; a.eos_host = b->st.eos_host;
    a.logits = b->st.logits; a.hidden = b->st.hidden; a.tokpart = h->tokpart; a.V = h->cfg.vocab;
    hipLaunchKernelGGL(t2s_commit_kernel, dim3(nrows), dim3(256), 0, S(stream), a);
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

int gsv_t2s_adopt_slots(gsv_t2s* h, int batch_dst, const int32_t* slots_dst, int batch_src, const int32_t* slots_src,
                        const int64_t* tok_override, int nrows, void* stream) {
    if (!h || !h->finalized) return fail(GSV_ERR_STATE, "handle not finalized");
    T2SBound* d = t2s_find(h, batch_dst);
    T2SBound* sb = t2s_find(h, batch_src);
    if (!d || !sb) return fail(GSV_ERR_STATE, "no state bound for batch %d", d ? batch_src : batch_dst);
    if (d == sb) return fail(GSV_ERR_ARG, "adopt_slots: source and destination are the same state");
    if (!slots_dst || !slots_src || nrows < 1 || nrows > batch_dst || nrows > batch_src) return fail(GSV_ERR_ARG, "adopt_slots: need 1..batch slot pairs (host arrays)");
    if (d->st.k_cache == sb->st.k_cache || d->st.v_cache == sb->st.v_cache) return fail(GSV_ERR_ARG, "adopt_slots: the two states share their KV cache");
    // a source row longer than the destination cache could not be copied whole, and the state kernel would still publish its length
    if (sb->st.max_kv > d->st.max_kv) return fail(GSV_ERR_ARG, "adopt_slots: the source cache (%d positions) is longer than the destination's (%d)", sb->st.max_kv, d->st.max_kv);
    for (int i = 0; i < nrows; ++i) {
        if (slots_dst[i] < 0 || slots_dst[i] >= batch_dst || slots_src[i] < 0 || slots_src[i] >= batch_src) return fail(GSV_ERR_ARG, "adopt_slots: slot out of range");
        for (int j = 0; j < i; ++j) if (slots_dst[j] == slots_dst[i]) return fail(GSV_ERR_ARG, "adopt_slots: destination slot %d listed twice", slots_dst[i]);
    }
    AdoptArgs a;
    a.ks = (const unsigned char*)sb->st.k_cache; a.vs = (const unsigned char*)sb->st.v_cache; a.kd = (unsigned char*)d->st.k_cache; a.vd = (unsigned char*)d->st.v_cache;
    a.Bs = batch_src; a.Ts = sb->st.max_kv; a.Bd = batch_dst; a.Td = d->st.max_kv; a.esz = h->cfg.dtype == GSV_F32 ? 4 : 2;
    a.sg_kv = sb->sg_kv; a.sg_x = sb->sg_x; a.sg_step = sb->sg_step; a.sg_eos = sb->sg_eos; a.sg_logits = sb->sg_logits; a.sg_hidden = sb->sg_hidden; a.sg_tok = sb->sg_tok;
    a.kv_len = d->st.kv_len; a.x_len = d->st.x_len; a.tok_override = d->st.tok_override; a.step = d->st.step; a.eos_at = d->st.eos_at; a.eos_host = d->st.eos_host;
    a.logits = d->st.logits; a.hidden = d->st.hidden; a.tokpart = h->tokpart; a.V = h->cfg.vocab;
    for (int r0 = 0; r0 < nrows; r0 += kAdoptMax) {
        const int n = std::min(kAdoptMax, nrows - r0);
        for (int i = 0; i < n; ++i) { a.dst[i] = (short)slots_dst[r0 + i]; a.src[i] = (short)slots_src[r0 + i]; a.ovr[i] = tok_override ? (long long)tok_override[r0 + i] : -1; }
        hipLaunchKernelGGL(t2s_adopt_kv_kernel, dim3(h->cfg.n_layer * kH, n, 2), dim3(256), 0, S(stream), a);
        hipLaunchKernelGGL(t2s_adopt_state_kernel, dim3(n), dim3(256), 0, S(stream), a);
    }
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

int gsv_t2s_move_slots(gsv_t2s* h, int batch_dst, const int32_t* slots_dst, int batch_src, const int32_t* slots_src, int nrows, void* stream) {
    if (!h || !h->finalized) return fail(GSV_ERR_STATE, "handle not finalized");
    T2SBound* d = t2s_find(h, batch_dst);
    T2SBound* sb = t2s_find(h, batch_src);
    if (!d || !sb) return fail(GSV_ERR_STATE, "no state bound for batch %d", d ? batch_src : batch_dst);
    if (d == sb) return fail(GSV_ERR_ARG, "move_slots: source and destination are the same state");
    if (!slots_dst || !slots_src || nrows < 1 || nrows > batch_dst || nrows > batch_src || nrows > kAdoptMax)
        return fail(GSV_ERR_ARG, "move_slots: need 1..min(batch, %d) slot pairs (host arrays)", kAdoptMax);
    if (d->st.k_cache == sb->st.k_cache || d->st.v_cache == sb->st.v_cache) return fail(GSV_ERR_ARG, "move_slots: the two states share their KV cache");
    if (sb->st.max_kv > d->st.max_kv) return fail(GSV_ERR_ARG, "move_slots: the source cache (%d positions) is longer than the destination's (%d)", sb->st.max_kv, d->st.max_kv);
    for (int i = 0; i < nrows; ++i) {
        if (slots_dst[i] < 0 || slots_dst[i] >= batch_dst || slots_src[i] < 0 || slots_src[i] >= batch_src) return fail(GSV_ERR_ARG, "move_slots: slot out of range");
        for (int j = 0; j < i; ++j)
            if (slots_dst[j] == slots_dst[i] || slots_src[j] == slots_src[i]) return fail(GSV_ERR_ARG, "move_slots: slot listed twice");
    }
    MoveArgs a;
    a.ks = (const unsigned char*)sb->st.k_cache; a.vs = (const unsigned char*)sb->st.v_cache; a.kd = (unsigned char*)d->st.k_cache; a.vd = (unsigned char*)d->st.v_cache;
    a.Bs = batch_src; a.Ts = sb->st.max_kv; a.Bd = batch_dst; a.Td = d->st.max_kv; a.esz = h->cfg.dtype == GSV_F32 ? 4 : 2; a.V = h->cfg.vocab; a.n = nrows;
    a.s_kv = sb->st.kv_len; a.s_x = sb->st.x_len; a.s_pre = sb->st.pre_tokens; a.s_ovr = sb->st.tok_override; a.s_step = sb->st.step; a.s_eos = sb->st.eos_at;
    a.s_logits = sb->st.logits; a.s_hidden = sb->st.hidden; a.s_seen = (const unsigned char*)sb->st.seen;
    a.d_kv = d->st.kv_len; a.d_x = d->st.x_len; a.d_pre = d->st.pre_tokens; a.d_ovr = d->st.tok_override; a.d_step = d->st.step; a.d_eos = d->st.eos_at; a.d_eos_host = d->st.eos_host;
    a.d_logits = d->st.logits; a.d_hidden = d->st.hidden; a.d_seen = (unsigned char*)d->st.seen; a.tokpart = h->tokpart;
    for (int i = 0; i < nrows; ++i) { a.dst[i] = (short)slots_dst[i]; a.src[i] = (short)slots_src[i]; }
    hipLaunchKernelGGL(t2s_move_kv_kernel, dim3(h->cfg.n_layer * kH, nrows, 2), dim3(256), 0, S(stream), a);
    hipLaunchKernelGGL(t2s_move_state_kernel, dim3(nrows + 1), dim3(256), 0, S(stream), a);
    HIPCHK(hipGetLastError());
    return GSV_OK;
}

int gsv_t2s_decode_hidden(gsv_t2s* h, int batch, const float* x, void* stream) {
    if (!h || !h->finalized) return fail(GSV_ERR_STATE, "handle not finalized");
    T2SBound* b = t2s_find(h, batch);
    if (!b) return fail(GSV_ERR_STATE, "no state bound for batch %d", batch);
    if (h->cfg.dtype == GSV_BF16 && batch >= h->batched_min && b->st.max_kv <= 1024) {   // the path gsv_t2s_decode takes
        HIPCHK(hipMemcpyAsync(h->xcur, x, sizeof(float) * (size_t)batch * kD, hipMemcpyDeviceToDevice, S(stream)));
        if (int rc = t2s_batched_layers<bf16_t>(h, b->st, S(stream))) return rc;
        return t2s_logits<bf16_t>(h, b->st, 0, h->xbuf, 0, batch, h->cfg.vocab, 1, S(stream));
    }
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
    const bool graph = (use_graph & 1) != 0, ft = (use_graph & GSV_STEP_FUSED_TOKEN) != 0;
    if (!graph) {
        for (int i = 0; i < n_steps; ++i)
            if (int rc = bf ? t2s_step<bf16_t>(h, b->st, S(stream), ft) : t2s_step<float>(h, b->st, S(stream), ft)) return rc;
        return GSV_OK;
    }
    // `steps` decode steps captured as one graph (1 = the single-step graph)
    auto graph_of = [&](int steps, hipGraphExec_t** out) -> int {
        hipGraphExec_t& exec = steps == 1 ? (ft ? b->graph_ft : b->graph) : b->wgraph[ft ? 1 : 0][steps];
        *out = &exec;
        if (exec) return GSV_OK;
        hipGraph_t g = nullptr;
        HIPCHK(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
        int rc = GSV_OK;                      // nothing between Begin and End returns: a stream left capturing is lost to the handle
        for (int i = 0; i < steps && !rc; ++i) rc = bf ? t2s_step<bf16_t>(h, b->st, h->cap_stream, ft) : t2s_step<float>(h, b->st, h->cap_stream, ft);
        hipError_t e = hipStreamEndCapture(h->cap_stream, &g);
        if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (e != hipSuccess) return fail(GSV_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
        e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) return fail(GSV_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
        return GSV_OK;
    };
    static const int win_max = getenv("GSV_STEP_WINDOW") ? std::max(1, std::min((int)T2SBound::kMaxWin, atoi(getenv("GSV_STEP_WINDOW")))) : 5;   // steps per graph (A/B: 1)
    int left = n_steps;
    while (left > 0) {
        // whole windows of `win_max` steps, then the remainder as one smaller window: a step's kernels read every position from the
        // state, so a captured window serves any kv
        const int w = std::min(left, win_max);
        hipGraphExec_t* exec = nullptr;
        if (int rc = graph_of(w, &exec)) return rc;
        HIPCHK(hipGraphLaunch(*exec, S(stream)));
        left -= w;
    }
    return GSV_OK;
}

int gsv_t2s_time_kernels(gsv_t2s* h, int batch, int iters, float* out_ms, void* stream) {
    if (!h || !h->finalized || !out_ms || iters < 1) return fail(GSV_ERR_STATE, "bad call");
    T2SBound* b = t2s_find(h, batch);
    if (!b) return fail(GSV_ERR_STATE, "no state bound for batch %d", batch);
    return h->cfg.dtype == GSV_BF16 ? t2s_time_impl<bf16_t>(h, b, iters, out_ms, S(stream))
                                    : t2s_time_impl<float>(h, b, iters, out_ms, S(stream));
}

int gsv_t2s_set_eos_mirror(gsv_t2s* h, int batch, int32_t* host_mapped) {
    if (!h || !h->finalized) return fail(GSV_ERR_STATE, "handle not finalized");
    T2SBound* b = t2s_find(h, batch);
    if (!b) return fail(GSV_ERR_STATE, "no state bound for batch %d", batch);
    b->st.eos_host = host_mapped;
    // the captured steps hold their kernel arguments by value
    t2s_drop_graphs(*b);
    return GSV_OK;
}

int gsv_t2s_batched_min(gsv_t2s* h) { return h && h->cfg.dtype == GSV_BF16 ? h->batched_min : 0x7fffffff; }
int gsv_t2s_ffn_slices(gsv_t2s* h, int batch) { return !h ? kNJ : (h->cfg.dtype == GSV_F32 ? ffn_slices<float>(batch) : ffn_slices<bf16_t>(batch)); }

size_t gsv_t2s_device_bytes(gsv_t2s* h) {
    if (!h) return 0;
    size_t n = 0;
    for (auto& b : h->arena.blocks) n += b.second;
    return n;
}

int gsv_t2s_set_debug(gsv_t2s* h, void* buf) {
    if (!h) return fail(GSV_ERR_ARG, "null handle");
    h->dbg = (unsigned long long*)buf;
    for (auto& kv : h->bound) t2s_drop_graphs(kv.second);
    return GSV_OK;
}

int gsv_t2s_flush(gsv_t2s* h, int batch, void* stream) {
    if (!h || !h->finalized) return fail(GSV_ERR_STATE, "handle not finalized");
    T2SBound* b = t2s_find(h, batch);
    if (!b) return fail(GSV_ERR_STATE, "no state bound for batch %d", batch);
    return t2s_token(h, b->st, 0, S(stream));
}

}  // extern "C"
