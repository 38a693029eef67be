// C-ABI of the reference-audio path (include/gsv_tts_hip.h, "reference audio" section); kernels in refaudio.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/gsv_tts_hip.h"
#include "gsv_error.h"
#include "refaudio.h"

using namespace gsv;

#define RCHK(expr)                                                                                        \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return abi_fail(GSV_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct gsv_ref {
    gsv_ref_config cfg;
    std::map<std::string, std::pair<float*, int64_t>> t;   // loaded tensors (device, fp32)
    bool finalized = false;
    // derived at finalize
    float *w_qk = nullptr, *b_qk = nullptr;                 // [2H][H], [2H]
    float *w_c0 = nullptr, *w_c1 = nullptr;                 // temporal convs as [2H][5*H]
    float *w_ssl = nullptr;                                 // [ssl][2*ssl]
    float *e2 = nullptr;                                    // [bins]
    float *dft = nullptr;                                   // [2*(n_fft/2+1)][n_fft]
    std::vector<void*> owned;
};

namespace {

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline size_t up(size_t v) { return (v + 63) / 64 * 64; }   // in floats: 256-byte slots

int fgemm(hipStream_t st, const float* X, long long ldx, const float* W, long long ldw, float* Y, long long ldy, int M, int N, int K,
          const float* bias_n = nullptr, int act = 0, const float* R = nullptr, long long ldr = 0, float alpha = 1.f,
          const float* bias_m = nullptr) {
    FGemmArgs a;
    a.X = X; a.ldx = ldx; a.W = W; a.ldw = ldw; a.Y = Y; a.ldy = ldy;
    a.bias_n = bias_n; a.bias_m = bias_m; a.R = R; a.ldr = ldr;
    a.M = M; a.N = N; a.K = K; a.alpha = alpha; a.act = act;
    fgemm_kernel<<<dim3((N + 63) / 64, (M + 63) / 64), 256, 0, st>>>(a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

const float* T_(gsv_ref* h, const char* name) { return h->t.at(name).first; }

struct Need { const char* name; int64_t numel; };

int dev_alloc(gsv_ref* h, float** p, size_t floats) {
    RCHK(hipMalloc(reinterpret_cast<void**>(p), floats * sizeof(float)));
    h->owned.push_back(*p);
    return GSV_OK;
}

}  // namespace

extern "C" {

int gsv_ref_create(const gsv_ref_config* cfg, gsv_ref** out) {
    if (!cfg || !out) return abi_fail(GSV_ERR_ARG, "null argument");
    if (cfg->hidden != 128 || cfg->n_head != 2 || cfg->spec_bins < 1 || cfg->spec_bins > cfg->n_fft / 2 + 1 || cfg->gin < 1 ||
        cfg->n_fft < 64 || cfg->n_fft % 2 || cfg->hop < 1 || cfg->ssl_dim < 1 || cfg->bins < 2 || cfg->kernel != 5)
        return abi_fail(GSV_ERR_ARG, "ref: unsupported config (MelStyleEncoder hidden 128 / 2 heads / kernel 5 expected)");
    gsv_ref* h = new gsv_ref();
    h->cfg = *cfg;
    *out = h;
    return GSV_OK;
}

int gsv_ref_destroy(gsv_ref* h) {
    if (!h) return GSV_OK;
    for (auto& kv : h->t) (void)hipFree(kv.second.first);
    for (void* p : h->owned) (void)hipFree(p);
    delete h;
    return GSV_OK;
}

int gsv_ref_load_tensor(gsv_ref* h, const char* name, const float* data, int64_t numel, void* stream) {
    if (!h || !name || !data || numel < 1) return abi_fail(GSV_ERR_ARG, "null argument");
    if (h->finalized) return abi_fail(GSV_ERR_STATE, "ref: load after finalize");
    const std::string n(name);
    if (n.rfind("ref_enc.", 0) != 0 && n.rfind("sv_emb.", 0) != 0 && n != "prelu.weight" && n.rfind("ssl_proj.", 0) != 0 &&
        n != "quantizer.vq.layers.0._codebook.embed")
        return abi_fail(GSV_ERR_ARG, "ref: unknown tensor %s", name);
    float* d = nullptr;
    RCHK(hipMalloc(reinterpret_cast<void**>(&d), numel * sizeof(float)));
    RCHK(hipMemcpyAsync(d, data, numel * sizeof(float), hipMemcpyDeviceToDevice, S(stream)));
    auto it = h->t.find(n);
    if (it != h->t.end()) (void)hipFree(it->second.first);
    h->t[n] = {d, numel};
    return GSV_OK;
}

int gsv_ref_finalize(gsv_ref* h, void* stream) {
    if (!h) return abi_fail(GSV_ERR_ARG, "null argument");
    if (h->finalized) return GSV_OK;
    const gsv_ref_config& c = h->cfg;
    const int64_t H = c.hidden;
    std::vector<Need> need = {
        {"ref_enc.spectral.0.fc.weight", H * c.spec_bins}, {"ref_enc.spectral.0.fc.bias", H},
        {"ref_enc.spectral.3.fc.weight", H * H}, {"ref_enc.spectral.3.fc.bias", H},
        {"ref_enc.temporal.0.conv1.conv.weight", 2 * H * H * 5}, {"ref_enc.temporal.0.conv1.conv.bias", 2 * H},
        {"ref_enc.temporal.1.conv1.conv.weight", 2 * H * H * 5}, {"ref_enc.temporal.1.conv1.conv.bias", 2 * H},
        {"ref_enc.slf_attn.w_qs.weight", H * H}, {"ref_enc.slf_attn.w_qs.bias", H},
        {"ref_enc.slf_attn.w_ks.weight", H * H}, {"ref_enc.slf_attn.w_ks.bias", H},
        {"ref_enc.slf_attn.w_vs.weight", H * H}, {"ref_enc.slf_attn.w_vs.bias", H},
        {"ref_enc.slf_attn.fc.weight", H * H}, {"ref_enc.slf_attn.fc.bias", H},
        {"ref_enc.fc.fc.weight", (int64_t)c.gin * H}, {"ref_enc.fc.fc.bias", c.gin},
        {"ssl_proj.weight", (int64_t)c.ssl_dim * c.ssl_dim * 2}, {"ssl_proj.bias", c.ssl_dim},
        {"quantizer.vq.layers.0._codebook.embed", (int64_t)c.bins * c.ssl_dim},
    };
    if (c.sv_dim > 0) {
        need.push_back({"sv_emb.weight", (int64_t)c.gin * c.sv_dim});
        need.push_back({"sv_emb.bias", c.gin});
        need.push_back({"prelu.weight", c.gin});
    }
    for (const Need& n : need) {
        auto it = h->t.find(n.name);
        if (it == h->t.end()) return abi_fail(GSV_ERR_STATE, "ref: tensor %s was not loaded", n.name);
        if (it->second.second != n.numel)
            return abi_fail(GSV_ERR_ARG, "ref: tensor %s has %lld elements, expected %lld", n.name, (long long)it->second.second, (long long)n.numel);
    }
    hipStream_t st = S(stream);
    int rc;
    if ((rc = dev_alloc(h, &h->w_qk, 2 * H * H)) || (rc = dev_alloc(h, &h->b_qk, 2 * H)) || (rc = dev_alloc(h, &h->w_c0, 2 * H * H * 5)) ||
        (rc = dev_alloc(h, &h->w_c1, 2 * H * H * 5)) || (rc = dev_alloc(h, &h->w_ssl, (size_t)c.ssl_dim * c.ssl_dim * 2)) ||
        (rc = dev_alloc(h, &h->e2, c.bins)) || (rc = dev_alloc(h, &h->dft, (size_t)(c.n_fft + 2) * c.n_fft)))
        return rc;
    RCHK(hipMemcpyAsync(h->w_qk, T_(h, "ref_enc.slf_attn.w_qs.weight"), H * H * 4, hipMemcpyDeviceToDevice, st));
    RCHK(hipMemcpyAsync(h->w_qk + H * H, T_(h, "ref_enc.slf_attn.w_ks.weight"), H * H * 4, hipMemcpyDeviceToDevice, st));
    RCHK(hipMemcpyAsync(h->b_qk, T_(h, "ref_enc.slf_attn.w_qs.bias"), H * 4, hipMemcpyDeviceToDevice, st));
    RCHK(hipMemcpyAsync(h->b_qk + H, T_(h, "ref_enc.slf_attn.w_ks.bias"), H * 4, hipMemcpyDeviceToDevice, st));
    const int nconv = (int)(2 * H * H * 5);
    conv_weight_kc_kernel<<<(nconv + 255) / 256, 256, 0, st>>>(T_(h, "ref_enc.temporal.0.conv1.conv.weight"), h->w_c0, (int)(2 * H), (int)H, 5);
    conv_weight_kc_kernel<<<(nconv + 255) / 256, 256, 0, st>>>(T_(h, "ref_enc.temporal.1.conv1.conv.weight"), h->w_c1, (int)(2 * H), (int)H, 5);
    const int nssl = c.ssl_dim * c.ssl_dim * 2;
    conv_weight_kc_kernel<<<(nssl + 255) / 256, 256, 0, st>>>(T_(h, "ssl_proj.weight"), h->w_ssl, c.ssl_dim, c.ssl_dim, 2);
    rowsq_kernel<<<(c.bins + 3) / 4, 256, 0, st>>>(T_(h, "quantizer.vq.layers.0._codebook.embed"), c.ssl_dim, c.bins, c.ssl_dim, h->e2);
    const long long ndft = (long long)(c.n_fft / 2 + 1) * c.n_fft;
    dft_rows_kernel<<<(unsigned)((ndft + 255) / 256), 256, 0, st>>>(h->dft, c.n_fft, c.n_fft / 2 + 1);
    RCHK(hipGetLastError());
    h->finalized = true;
    return GSV_OK;
}

size_t gsv_ref_workspace(gsv_ref* h, int n_samples, int n_frames, int n_ssl) {
    if (!h) return 0;
    const gsv_ref_config& c = h->cfg;
    size_t spec = 0, ge = 0, lat = 0;
    if (n_samples > 0) {
        const size_t T = 1 + n_samples / c.hop;
        spec = up((size_t)n_samples + c.n_fft) + up(T * (c.n_fft + 2));
    }
    if (n_frames > 0) {
        const size_t T = n_frames, H = c.hidden;
        ge = up(T * c.spec_bins) + 2 * up(T * H) + 2 * up((T + 4) * H) + 2 * up(T * 2 * H) + up(H * T) + up(T * T) + 2 * up(T * H) +
             up(T * c.gin) + up(c.gin);
    }
    if (n_ssl > 0) {
        const size_t Th = n_ssl, To = n_ssl / 2;
        lat = up(Th * c.ssl_dim) + up(To * c.ssl_dim) + up(To * c.bins) + up(To);
    }
    return sizeof(float) * std::max(spec, std::max(ge, lat));
}

int gsv_ref_spectrogram(gsv_ref* h, const float* audio, int n_samples, float* spec, void* workspace, size_t workspace_bytes,
                        void* stream) {
    if (!h || !audio || !spec || !workspace) return abi_fail(GSV_ERR_ARG, "null argument");
    if (!h->finalized) return abi_fail(GSV_ERR_STATE, "ref: not finalized");
    const gsv_ref_config& c = h->cfg;
    if (n_samples <= c.n_fft / 2) return abi_fail(GSV_ERR_ARG, "ref: reflect padding needs more than n_fft/2 samples (got %d)", n_samples);
    if (workspace_bytes < gsv_ref_workspace(h, n_samples, 0, 0)) return abi_fail(GSV_ERR_ARG, "ref: workspace too small");
    hipStream_t st = S(stream);
    const int T = 1 + n_samples / c.hop, bins = c.n_fft / 2 + 1;
    float* padded = static_cast<float*>(workspace);
    float* Z = padded + up((size_t)n_samples + c.n_fft);
    reflect_pad_kernel<<<(n_samples + c.n_fft + 255) / 256, 256, 0, st>>>(audio, n_samples, c.n_fft / 2, padded);
    if (fgemm(st, padded, c.hop, h->dft, c.n_fft, Z, 2 * bins, T, 2 * bins, c.n_fft)) return abi_fail(GSV_ERR_HIP, "ref: DFT launch failed");
    const long long n = (long long)T * bins;
    magnitude_t_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(Z, T, bins, spec);
    RCHK(hipGetLastError());
    return GSV_OK;
}

int gsv_ref_get_ge(gsv_ref* h, const float* spec, int n_frames, const float* sv_emb, float* ge, void* workspace,
                   size_t workspace_bytes, void* stream) {
    if (!h || !spec || !ge || !workspace) return abi_fail(GSV_ERR_ARG, "null argument");
    if (!h->finalized) return abi_fail(GSV_ERR_STATE, "ref: not finalized");
    const gsv_ref_config& c = h->cfg;
    if (n_frames < 1) return abi_fail(GSV_ERR_ARG, "ref: no frames");
    if (sv_emb && c.sv_dim <= 0) return abi_fail(GSV_ERR_ARG, "ref: this handle has no sv_emb (v2)");
    if (workspace_bytes < gsv_ref_workspace(h, 0, n_frames, 0)) return abi_fail(GSV_ERR_ARG, "ref: workspace too small");
    hipStream_t st = S(stream);
    const int T = n_frames, H = c.hidden, D = H / c.n_head;
    float* p = static_cast<float*>(workspace);
    auto take = [&](size_t n) { float* r = p; p += up(n); return r; };
    float* X0 = take((size_t)T * c.spec_bins);
    float* H1 = take((size_t)T * H);
    float* G2 = take((size_t)T * H);
    float* P0 = take((size_t)(T + 4) * H);
    float* P1 = take((size_t)(T + 4) * H);
    float* C = take((size_t)T * 2 * H);
    float* QK = take((size_t)T * 2 * H);
    float* VT = take((size_t)H * T);
    float* Sm = take((size_t)T * T);
    float* O = take((size_t)T * H);
    float* A = take((size_t)T * H);
    float* F = take((size_t)T * c.gin);
    float* sv = take(c.gin);
    int bad = 0;
    // refer[:, :704] channels-first -> frames-major
    transpose_kernel<<<dim3((T + 31) / 32, (c.spec_bins + 31) / 32), 256, 0, st>>>(spec, T, X0, c.spec_bins, c.spec_bins, T);
    // spectral: Linear + Mish, twice (modules.py:385-392); the second lands in the zero-padded buffer of the first conv
    bad |= fgemm(st, X0, c.spec_bins, T_(h, "ref_enc.spectral.0.fc.weight"), c.spec_bins, H1, H, T, H, c.spec_bins, T_(h, "ref_enc.spectral.0.fc.bias"), 1);
    RCHK(hipMemsetAsync(P0, 0, (size_t)(T + 4) * H * sizeof(float), st));
    RCHK(hipMemsetAsync(P1, 0, (size_t)(T + 4) * H * sizeof(float), st));
    bad |= fgemm(st, H1, H, T_(h, "ref_enc.spectral.3.fc.weight"), H, P0 + 2 * H, H, T, H, H, T_(h, "ref_enc.spectral.3.fc.bias"), 1);
    // temporal: two Conv1dGLU (k 5, same padding) as overlapping-row GEMMs
    const int nel = T * H;
    bad |= fgemm(st, P0, H, h->w_c0, 5 * H, C, 2 * H, T, 2 * H, 5 * H, T_(h, "ref_enc.temporal.0.conv1.conv.bias"));
    glu_residual_kernel<<<(nel + 255) / 256, 256, 0, st>>>(P0 + 2 * H, C, P1 + 2 * H, T, H);
    bad |= fgemm(st, P1, H, h->w_c1, 5 * H, C, 2 * H, T, 2 * H, 5 * H, T_(h, "ref_enc.temporal.1.conv1.conv.bias"));
    glu_residual_kernel<<<(nel + 255) / 256, 256, 0, st>>>(P1 + 2 * H, C, G2, T, H);
    // self-attention (modules.py:291-343): temperature sqrt(d_model), no mask (the reference mask is all ones)
    bad |= fgemm(st, G2, H, h->w_qk, H, QK, 2 * H, T, 2 * H, H, h->b_qk);
    bad |= fgemm(st, T_(h, "ref_enc.slf_attn.w_vs.weight"), H, G2, H, VT, T, H, T, H, nullptr, 0, nullptr, 0, 1.f, T_(h, "ref_enc.slf_attn.w_vs.bias"));
    const float inv_temp = 1.f / sqrtf((float)H);
    for (int hd = 0; hd < c.n_head; ++hd) {
        bad |= fgemm(st, QK + hd * D, 2 * H, QK + H + hd * D, 2 * H, Sm, T, T, T, D, nullptr, 0, nullptr, 0, inv_temp);
        softmax_rows_kernel<<<(T + 3) / 4, 256, 0, st>>>(Sm, T, T);
        bad |= fgemm(st, Sm, T, VT + (size_t)hd * D * T, T, O + hd * D, H, T, D, T);
    }
    bad |= fgemm(st, O, H, T_(h, "ref_enc.slf_attn.fc.weight"), H, A, H, T, H, H, T_(h, "ref_enc.slf_attn.fc.bias"), 0, G2, H);
    bad |= fgemm(st, A, H, T_(h, "ref_enc.fc.fc.weight"), H, F, c.gin, T, c.gin, H, T_(h, "ref_enc.fc.fc.bias"));
    if (sv_emb) gemv_rows_kernel<<<(c.gin + 3) / 4, 256, 0, st>>>(sv_emb, T_(h, "sv_emb.weight"), T_(h, "sv_emb.bias"), sv, c.gin, c.sv_dim);
    pool_prelu_kernel<<<(c.gin + 255) / 256, 256, 0, st>>>(F, T, c.gin, sv_emb ? sv : nullptr, sv_emb ? T_(h, "prelu.weight") : nullptr, ge);
    if (bad) return abi_fail(GSV_ERR_HIP, "ref: a get_ge launch failed");
    RCHK(hipGetLastError());
    return GSV_OK;
}

int gsv_ref_extract_latent(gsv_ref* h, const float* ssl, int n_ssl, int64_t* codes, float* margin, void* workspace,
                           size_t workspace_bytes, void* stream) {
    if (!h || !ssl || !codes || !workspace) return abi_fail(GSV_ERR_ARG, "null argument");
    if (!h->finalized) return abi_fail(GSV_ERR_STATE, "ref: not finalized");
    const gsv_ref_config& c = h->cfg;
    if (n_ssl < 2) return abi_fail(GSV_ERR_ARG, "ref: extract_latent needs at least 2 ssl frames");
    if (workspace_bytes < gsv_ref_workspace(h, 0, 0, n_ssl)) return abi_fail(GSV_ERR_ARG, "ref: workspace too small");
    hipStream_t st = S(stream);
    const int Th = n_ssl, To = n_ssl / 2, Dm = c.ssl_dim;
    float* p = static_cast<float*>(workspace);
    auto take = [&](size_t n) { float* r = p; p += up(n); return r; };
    float* Xt = take((size_t)Th * Dm);
    float* Y = take((size_t)To * Dm);
    float* dot = take((size_t)To * c.bins);
    float* x2 = take(To);
    int bad = 0;
    transpose_kernel<<<dim3((Th + 31) / 32, (Dm + 31) / 32), 256, 0, st>>>(ssl, Th, Xt, Dm, Dm, Th);
    // Conv1d(ssl, ssl, 2, stride 2): frames 2i, 2i+1 are one contiguous row of the frames-major buffer
    bad |= fgemm(st, Xt, 2 * Dm, h->w_ssl, 2 * Dm, Y, Dm, To, Dm, 2 * Dm, T_(h, "ssl_proj.bias"));
    bad |= fgemm(st, Y, Dm, T_(h, "quantizer.vq.layers.0._codebook.embed"), Dm, dot, c.bins, To, c.bins, Dm);
    rowsq_kernel<<<(To + 3) / 4, 256, 0, st>>>(Y, Dm, To, Dm, x2);
    nearest_code_kernel<<<(To + 3) / 4, 256, 0, st>>>(dot, x2, h->e2, To, c.bins, reinterpret_cast<long long*>(codes), margin);
    if (bad) return abi_fail(GSV_ERR_HIP, "ref: an extract_latent launch failed");
    RCHK(hipGetLastError());
    return GSV_OK;
}

}  // extern "C"
