// GPT prefill helpers for gfx950 (TTFT path): prompt embedding, prompt-masked attention with the
// KV-cache write, row LayerNorm.  The dense projections run on tapgemm (MFMA).
//
// Reference: T2SBlock.process_prompt gsv_tts/GPT_SoVITS/GPT/t2s_model.py:31-65; masks
// t2s_model.py:335-347,365-381; embeddings t2s_model.py:322-331,353-361.
#pragma once
#include "t2s_decode.h"

namespace gsv {

// rows packed [x_b | y_b | 0-pad]:  text: E_text[id] + bert_proj + alpha_t*pe[t] ; audio: E_audio[id] + alpha_a*pe[t - lx]
struct EmbedArgs {
    const int64_t* x_ids;   // [nrows][lx_max]
    const int64_t* y_ids;   // [nrows][ly_max]
    const float* proj;      // [nrows][lx_max][512] = bert @ Wb^T + bb
    const int64_t* x_lens;
    const int64_t* y_lens;
    const float* emb_text;  // [n_phoneme][512]
    const float* emb_audio; // [V][512]
    const float* pe_text;   // [n_pos][512] (alpha folded)
    const float* pe_audio;
    float* xy;              // [nrows][l_max][512]
    int lx_max, ly_max, l_max, n_phoneme, V;
};

__global__ __launch_bounds__(128) void t2s_embed_kernel(EmbedArgs a) {
    const int t = blockIdx.x, b = blockIdx.y, c = threadIdx.x * 4;
    const int lx = (int)a.x_lens[b], ly = (int)a.y_lens[b];
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (t < lx) {
        int id = (int)a.x_ids[(size_t)b * a.lx_max + t];
        id = min(max(id, 0), a.n_phoneme - 1);
        f32x4 e = *reinterpret_cast<const f32x4*>(a.emb_text + (size_t)id * kD + c);
        f32x4 p = *reinterpret_cast<const f32x4*>(a.proj + ((size_t)b * a.lx_max + t) * kD + c);
        f32x4 pe = *reinterpret_cast<const f32x4*>(a.pe_text + (size_t)t * kD + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (e[i] + p[i]) * 1.0f + pe[i];
    } else if (t < lx + ly) {
        const int ty = t - lx;
        int id = (int)a.y_ids[(size_t)b * a.ly_max + ty];
        id = min(max(id, 0), a.V - 1);
        f32x4 e = *reinterpret_cast<const f32x4*>(a.emb_audio + (size_t)id * kD + c);
        f32x4 pe = *reinterpret_cast<const f32x4*>(a.pe_audio + (size_t)ty * kD + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = e[i] * 1.0f + pe[i];
    }
    *reinterpret_cast<f32x4*>(a.xy + ((size_t)b * a.l_max + t) * kD + c) = o;
}

// y[row] = LayerNorm(x[row]) over 512, one wave per row, two-pass like torch
__global__ __launch_bounds__(256) void ln_rows_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                      const float* __restrict__ bta, float* __restrict__ y, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float v[8];
    Ld<float, 8>::load(x + (size_t)row * kD + lane * 8, v);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    const float mean = wave_sum(s) * (1.0f / kD);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] -= mean; q += v[i] * v[i]; }
    const float rs = 1.0f / sqrtf(wave_sum(q) * (1.0f / kD) + kEps);
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = v[i] * rs * g[lane * 8 + i] + bta[lane * 8 + i];
    *reinterpret_cast<f32x4*>(y + (size_t)row * kD + lane * 8) = f32x4{o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4*>(y + (size_t)row * kD + lane * 8 + 4) = f32x4{o[4], o[5], o[6], o[7]};
}

// Prompt attention for one (head, sequence, query-slice).  K/V of the head are staged in LDS
// (fp32, rows padded to 33 floats so a lane-per-key dot is bank-conflict free) and written to
// the cache in its storage type; each wave then walks its queries:
//   text query i < lx : keys [0, lx) ; audio query i >= lx : keys [0, i]   (appendix A.3)
template <typename WT>
struct PrefillAttnArgs {
    const float* qkv;    // [nrows][l_max][1536]
    const int64_t* x_lens;
    const int64_t* y_lens;
    WT* kc;              // this layer: [B][16][T][32]
    WT* vc;
    int T, slot0, l_max, qsplit;
    float* out;          // [nrows][l_max][512]
};

template <typename WT>
__global__ __launch_bounds__(256) void t2s_prefill_attn_kernel(PrefillAttnArgs<WT> a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int h = blockIdx.x, r = blockIdx.y, qs = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lx = (int)a.x_lens[r], L = lx + (int)a.y_lens[r];
    float* Ks = smem;                       // [L][33]
    float* Vs = Ks + (size_t)a.l_max * 33;  // [L][32]
    float* Sc = Vs + (size_t)a.l_max * 32;  // [4][l_max]
    float* Qs = Sc + 4 * (size_t)a.l_max;   // [4][32]
    const float* base = a.qkv + (size_t)r * a.l_max * 1536;
    WT* Kp = a.kc + (((size_t)(a.slot0 + r) * kH + h) * a.T) * kDh;
    WT* Vp = a.vc + (((size_t)(a.slot0 + r) * kH + h) * a.T) * kDh;
    for (int e = tid; e < L * 32; e += 256) {
        const int t = e >> 5, d = e & 31;
        const WT kq = from_f32<WT>(base[(size_t)t * 1536 + 512 + h * 32 + d]);
        const WT vq = from_f32<WT>(base[(size_t)t * 1536 + 1024 + h * 32 + d]);
        Ks[t * 33 + d] = to_f32<WT>(kq);
        Vs[t * 32 + d] = to_f32<WT>(vq);
        if (qs == 0 && t < a.T) { Kp[(size_t)t * kDh + d] = kq; Vp[(size_t)t * kDh + d] = vq; }
    }
    __syncthreads();
    const float scale = 0.17677669529663687f;
    float* sc = Sc + (size_t)wid * a.l_max;
    float* qv = Qs + wid * 32;
    for (int i = qs * 4 + wid; i < a.l_max; i += 4 * a.qsplit) {
        float* o = a.out + ((size_t)r * a.l_max + i) * kD + h * 32;
        if (i >= L) {  // padded query row: fully masked -> 0 (SDPA, torch >= 2.5)
            if (lane < 32) o[lane] = 0.f;
            continue;
        }
        if (lane < 32) qv[lane] = base[(size_t)i * 1536 + h * 32 + lane];
        __builtin_amdgcn_wave_barrier();
        const int nk = i < lx ? lx : i + 1;
        float mx = -INFINITY;
        for (int t = lane; t < nk; t += 64) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) s = fmaf(qv[d], Ks[t * 33 + d], s);
            s *= scale;
            sc[t] = s;
            mx = fmaxf(mx, s);
        }
        mx = wave_max(mx);
        float sum = 0.f;
        for (int t = lane; t < nk; t += 64) {
            const float e = expf(sc[t] - mx);
            sc[t] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        __builtin_amdgcn_wave_barrier();
        const int d = lane & 31, hf = lane >> 5;
        float acc = 0.f;
        for (int t = hf; t < nk; t += 2) acc = fmaf(sc[t] / sum, Vs[t * 32 + d], acc);
        acc += __shfl_xor(acc, 32, 64);
        if (lane < 32) o[lane] = acc;
        __builtin_amdgcn_wave_barrier();
    }
}

// hlast[r] = hidden[r][x_len + y_len - 1]; and per-slot state after a (re)fill
struct PrefillFinishArgs {
    const float* hidden;  // [nrows][l_max][512]
    const int64_t* x_lens;
    const int64_t* y_lens;
    float* hlast;         // [nrows][512]
    int64_t* kv_len;
    int64_t* x_len;
    int32_t* step;
    int32_t* eos_at;
    int slot0, l_max;
};

__global__ __launch_bounds__(128) void t2s_prefill_finish_kernel(PrefillFinishArgs a) {
    const int r = blockIdx.x, c = threadIdx.x * 4;
    const int lx = (int)a.x_lens[r], L = lx + (int)a.y_lens[r];
    const int last = L > 0 ? L - 1 : 0;
    *reinterpret_cast<f32x4*>(a.hlast + (size_t)r * kD + c) =
        *reinterpret_cast<const f32x4*>(a.hidden + ((size_t)r * a.l_max + last) * kD + c);
    if (threadIdx.x == 0) {
        a.kv_len[a.slot0 + r] = L;
        a.x_len[a.slot0 + r] = lx;
        a.step[a.slot0 + r] = 0;
        a.eos_at[a.slot0 + r] = -1;
    }
}

}  // namespace gsv
