// C-ABI of the subtitle alignment (include/gsv_tts_hip.h, "alignment" section); kernels in align.h.
#include <hip/hip_runtime.h>

#include "../../include/gsv_tts_hip.h"
#include "align.h"
#include "sola.h"
#include "gsv_error.h"

using namespace gsv;

namespace {

constexpr size_t kAlignLdsBudget = 144 * 1024;   // of the CU's 160 KB; the rest is dp rows + statics

struct AlignPlan {
    int threads, npt, nw;
    size_t off_flag, off_bits, bytes, lds;
    int bits_in_lds;
};

bool align_plan(int T, int N, AlignPlan* p) {
    if (T < 1 || N < 2 || N > 4096) return false;
    p->threads = N <= 256 ? 256 : 1024;
    p->npt = N <= 256 ? 1 : (N <= 1024 ? 1 : (N <= 2048 ? 2 : 4));
    const int np = p->threads * p->npt;
    p->nw = np / 64;
    const size_t normal = ((size_t)T * N * sizeof(float) + 255) / 256 * 256;
    const size_t flag = ((size_t)T * sizeof(int) + 255) / 256 * 256;
    const size_t bits = (size_t)T * p->nw * sizeof(unsigned long long);
    p->off_flag = normal;
    p->off_bits = normal + flag;
    p->bytes = normal + flag + bits;
    const size_t dp = (size_t)2 * np * sizeof(float);
    p->bits_in_lds = dp + bits <= kAlignLdsBudget;
    p->lds = dp + (p->bits_in_lds ? bits : 0);
    return true;
}

template <int TH, int NPT>
hipError_t launch_dp(const AlignPlan& p, const float* normal, const int* flag, int T, int N, int* assign,
                     unsigned long long* bits, hipStream_t st) {
    if (p.lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&align_dp_kernel<TH, NPT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
        if (e != hipSuccess) return e;
    }
    align_dp_kernel<TH, NPT><<<1, TH, p.lds, st>>>(normal, flag, T, N, assign, bits, p.bits_in_lds);
    return hipGetLastError();
}

}  // namespace

extern "C" {

size_t gsv_align_workspace(int T, int N) {
    AlignPlan p;
    return align_plan(T, N, &p) ? p.bytes : 0;
}

int gsv_align_viterbi(const float* attn, int H, int T, int N, int32_t* assign, void* workspace, size_t workspace_bytes,
                      void* stream) {
    if (!attn || !assign || !workspace) return abi_fail(GSV_ERR_ARG, "null argument");
    AlignPlan p;
    if (H < 1 || H > kAlignMaxHeads || !align_plan(T, N, &p))
        return abi_fail(GSV_ERR_ARG, "align: need 1 <= H <= %d, T >= 1, 2 <= N <= 4096 (got H=%d T=%d N=%d)", kAlignMaxHeads, H, T, N);
    if (workspace_bytes < p.bytes) return abi_fail(GSV_ERR_ARG, "align: workspace %zu < %zu bytes", workspace_bytes, p.bytes);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    float* normal = reinterpret_cast<float*>(ws);
    int* flag = reinterpret_cast<int*>(ws + p.off_flag);
    unsigned long long* bits = reinterpret_cast<unsigned long long*>(ws + p.off_bits);
    align_normal_kernel<<<(T + 3) / 4, 256, 0, st>>>(attn, H, T, N, normal, flag);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) {
        if (p.threads == 256) e = launch_dp<256, 1>(p, normal, flag, T, N, assign, bits, st);
        else if (p.npt == 1) e = launch_dp<1024, 1>(p, normal, flag, T, N, assign, bits, st);
        else if (p.npt == 2) e = launch_dp<1024, 2>(p, normal, flag, T, N, assign, bits, st);
        else e = launch_dp<1024, 4>(p, normal, flag, T, N, assign, bits, st);
    }
    if (e != hipSuccess) return abi_fail(GSV_ERR_HIP, "align launch: %s", hipGetErrorString(e));
    return GSV_OK;
}


// ---- streaming splice (sola.h)
size_t gsv_sola_workspace(int search_len) { return search_len < 0 ? 0 : ((size_t)(search_len + 1) * sizeof(float) + 255) / 256 * 256; }

int gsv_sola(const float* prev_tail, const float* chunk, int n, int overlap, int search_len, float* out, int32_t* offset,
             void* workspace, size_t workspace_bytes, void* stream) {
    if (!prev_tail || !chunk || !out || !offset || !workspace) return abi_fail(GSV_ERR_ARG, "sola: null argument");
    if (overlap < 1 || search_len < 0 || n < overlap) return abi_fail(GSV_ERR_ARG, "sola: chunk of %d samples, overlap %d, search %d", n, overlap, search_len);
    // TTS.py:1614: key = f2[:, :, :overlap + search] -- a chunk shorter than that offers fewer candidate offsets
    const int n_off = (n < overlap + search_len ? n : overlap + search_len) - overlap + 1;
    if (workspace_bytes < (size_t)n_off * sizeof(float)) return abi_fail(GSV_ERR_ARG, "sola workspace %zu < %zu", workspace_bytes, (size_t)n_off * sizeof(float));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* score = static_cast<float*>(workspace);
    hipLaunchKernelGGL(sola_score_kernel, dim3(n_off), dim3(256), 0, st, prev_tail, chunk, overlap, score);
    const int blocks = (n + 256 * 8 - 1) / (256 * 8);
    hipLaunchKernelGGL(sola_splice_kernel, dim3(blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks)), dim3(256), 0, st, prev_tail, chunk, n, overlap,
                       (const float*)score, n_off, out, offset);
    if (hipGetLastError() != hipSuccess) return abi_fail(GSV_ERR_HIP, "sola: launch failed");
    return GSV_OK;
}

}  // extern "C"
