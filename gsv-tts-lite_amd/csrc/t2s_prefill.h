// GPT prefill helpers for gfx950 (TTFT path): prompt embedding, prompt-masked attention with the
// KV-cache write, row LayerNorm.  The dense projections run on tapgemm (MFMA).
//
// Reference: T2SBlock.process_prompt gsv_tts/GPT_SoVITS/GPT/t2s_model.py:31-65; masks
// t2s_model.py:335-347,365-381; embeddings t2s_model.py:322-331,353-361.
#pragma once
#include "t2s_decode.h"
#include "tapgemm.h"

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

static __global__ __launch_bounds__(128) void t2s_embed_kernel(EmbedArgs a) {
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
static __global__ __launch_bounds__(256) void ln_rows_kernel(const float* __restrict__ x, const float* __restrict__ g,
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
    const int32_t* slots;  // state slot of every row, or null = slot0 + row
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
    WT* Kp = a.kc + (((size_t)(a.slots ? a.slots[r] : a.slot0 + r) * kH + h) * a.T) * kDh;
    WT* Vp = a.vc + (((size_t)(a.slots ? a.slots[r] : a.slot0 + r) * kH + h) * a.T) * kDh;
    // eight items' loads in flight per thread before the first is used, from clamped addresses and masked afterwards: one item per iteration was one
    // memory round trip per iteration (25 in a row for a 200-position prompt: most of this launch's 35 us; round 6)
    constexpr int SU = 8;
    for (int e0 = tid; e0 < L * 32; e0 += 256 * SU) {
        float kf[SU], vf[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int e = min(e0 + u * 256, L * 32 - 1), t = e >> 5, d = e & 31;
            kf[u] = base[(size_t)t * 1536 + 512 + h * 32 + d];
            vf[u] = base[(size_t)t * 1536 + 1024 + h * 32 + d];
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int e = e0 + u * 256, t = e >> 5, d = e & 31;
            if (e < L * 32) {
                const WT kq = from_f32<WT>(kf[u]), vq = from_f32<WT>(vf[u]);
                Ks[t * 33 + d] = to_f32<WT>(kq);
                Vs[t * 32 + d] = to_f32<WT>(vq);
                if (qs == 0 && t < a.T) { Kp[(size_t)t * kDh + d] = kq; Vp[(size_t)t * kDh + d] = vq; }
            }
        }
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

// The same prompt attention on the matrix cores (bf16 cache mode).  One block = (head, sequence, group of
// 4 query tiles); a wave owns one 32-query tile and runs flash attention over 32-key tiles:
//   S^T = K Q^T      A = K tile [32 keys][32 d] (2 k-steps), B = Q tile -> D: lane = query, registers = keys
//   online softmax   per lane over its 16 key registers, the two lane halves of a query combined per tile
//   O^T += V^T P^T   the probabilities go from D registers to the B operand WITHOUT a shuffle: a
//                    contraction may enumerate its index in any order, so V^T is staged in LDS with its
//                    keys permuted into the order the D registers hold them (vpos below).
// K, V and the scores' Q are rounded to bf16 (what the cache stores / what SDPA does in bf16); softmax
// statistics and accumulation are fp32.  Masking as t2s_prefill_attn_kernel (appendix A.3).
struct PrefillAttnMfmaArgs {
    const bf16_t* qkv;   // [nrows][l_max][1536] bf16: the QKV GEMM's rows, rounded where this kernel used to round them (K / V for the cache,
                         // Q for the score MFMA) -- the same values, half the bytes written and re-read
    const int64_t* x_lens;
    const int64_t* y_lens;
    bf16_t* kc;          // this layer: [B][16][T][32]
    bf16_t* vc;
    int T, slot0, l_max;
    const int32_t* slots;  // state slot of every row, or null = slot0 + row
    bf16_t* out;         // [nrows][l_max][512] bf16: what the out-proj GEMM stages (it rounded the fp32 rows the same way)
};

// position of tile-local key kk inside its 32-key group of the permuted V^T row
__device__ __forceinline__ int vpos(int kk) {
    const int hf = (kk >> 2) & 1, blk = kk >> 3;
    return (blk >> 1) * 16 + 8 * hf + (kk & 3) + 4 * (blk & 1);
}

static __global__ __launch_bounds__(256) void t2s_prefill_attn_mfma_kernel(PrefillAttnMfmaArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char plds[];
    const int h = blockIdx.x, r = blockIdx.y, qg = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = lane & 31, hf = lane >> 5;
    const int lx = (int)a.x_lens[r], L = lx + (int)a.y_lens[r];
    const int q0 = (qg * 4 + wid) * 32;                      // first query of the wave's tile
    const int qb0 = qg * 128;                                // first query of the block
    if (qb0 >= a.l_max) return;
    // keys the block can see: text queries see [0, lx), audio query i sees [0, i]
    const int qlast = min(qb0 + 128, L) - 1;
    const int nkeys = qlast < 0 ? 0 : (qlast < lx ? lx : max(lx, qlast + 1));
    const int nkt = (nkeys + 31) / 32;                       // key tiles staged
    constexpr int KRS = 32 * 2 + 16;                         // K / Q row stride (bytes)
    const int vrs = nkt * 64 + 16;                           // V^T row stride (bytes)
    unsigned char* Ks = plds;                                // [nkt*32][KRS]
    unsigned char* Qs = Ks + (size_t)nkt * 32 * KRS;         // [128][KRS]
    unsigned char* Vt = Qs + 128 * KRS;                      // [32 d][vrs]
    const bf16_t* base = a.qkv + (size_t)r * a.l_max * 1536;
    bf16_t* Kp = a.kc + (((size_t)(a.slots ? a.slots[r] : a.slot0 + r) * kH + h) * a.T) * kDh;
    bf16_t* Vp = a.vc + (((size_t)(a.slots ? a.slots[r] : a.slot0 + r) * kH + h) * a.T) * kDh;
    // ---- stage K (rows), V (transposed, permuted), Q (rows): one float4 of 4 d per item.  Four items' loads are in flight before the
    //      first is used, from clamped (always valid) addresses and masked afterwards: one item per iteration was one memory round
    //      trip per iteration (load, wait, convert, store: 7 in a row for a 200-position prompt)
    constexpr int SU = 4;
    const int n_items = nkt * 32 * 8;
    for (int e0 = tid; e0 < n_items; e0 += 256 * SU) {
        uint2 kv[SU], vv[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int e = e0 + u * 256, t = e >> 3, d4 = (e & 7) * 4;
            const int tc = min(t, max(L - 1, 0));
            kv[u] = *reinterpret_cast<const uint2*>(base + (size_t)tc * 1536 + 512 + h * 32 + d4);
            vv[u] = *reinterpret_cast<const uint2*>(base + (size_t)tc * 1536 + 1024 + h * 32 + d4);
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int e = e0 + u * 256, t = e >> 3, d4 = (e & 7) * 4;
            if (e < n_items) {
                const bool ok = t < L;
                uint2 kp, vp;
                kp.x = ok ? kv[u].x : 0u; kp.y = ok ? kv[u].y : 0u;
                vp.x = ok ? vv[u].x : 0u; vp.y = ok ? vv[u].y : 0u;
                *reinterpret_cast<uint2*>(Ks + (size_t)t * KRS + d4 * 2) = kp;
                const int pos = (t & ~31) + vpos(t & 31);
                *reinterpret_cast<bf16_t*>(Vt + (size_t)(d4 + 0) * vrs + pos * 2) = (bf16_t)(vp.x & 0xffff);
                *reinterpret_cast<bf16_t*>(Vt + (size_t)(d4 + 1) * vrs + pos * 2) = (bf16_t)(vp.x >> 16);
                *reinterpret_cast<bf16_t*>(Vt + (size_t)(d4 + 2) * vrs + pos * 2) = (bf16_t)(vp.y & 0xffff);
                *reinterpret_cast<bf16_t*>(Vt + (size_t)(d4 + 3) * vrs + pos * 2) = (bf16_t)(vp.y >> 16);
                if (t >= qb0 && t < qb0 + 128 && ok && t < a.T) {   // each block writes the cache rows of its own queries: once each
                    *reinterpret_cast<uint2*>(Kp + (size_t)t * kDh + d4) = kp;
                    *reinterpret_cast<uint2*>(Vp + (size_t)t * kDh + d4) = vp;
                }
            }
        }
    }
    {
        uint2 qv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                            // 128 x 8 items = 4 per thread, all in flight
            const int e = tid + u * 256, qi = e >> 3, d4 = (e & 7) * 4;
            const int ic = min(qb0 + qi, max(L - 1, 0));
            qv[u] = *reinterpret_cast<const uint2*>(base + (size_t)ic * 1536 + h * 32 + d4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + u * 256, qi = e >> 3, d4 = (e & 7) * 4;
            const bool ok = qb0 + qi < L;
            uint2 qp;
            qp.x = ok ? qv[u].x : 0u; qp.y = ok ? qv[u].y : 0u;
            *reinterpret_cast<uint2*>(Qs + (size_t)qi * KRS + d4 * 2) = qp;
        }
    }
    __syncthreads();
    if (q0 >= a.l_max) return;
    const int i = q0 + j;                                    // this lane's query
    const bool qvalid = i < L;
    const int klim = !qvalid ? 0 : (i < lx ? lx : i + 1);    // keys [0, klim) visible
    // the wave's key tiles: up to the last query of its tile
    const int wq_last = min(q0 + 31, L - 1);
    const int wkeys = wq_last < 0 ? 0 : (wq_last < lx ? lx : max(lx, wq_last + 1));
    const int wkt = (wkeys + 31) / 32;
    const float scale = 0.17677669529663687f * 1.4426950408889634f;   // 1/sqrt(32) in the exp2 domain
    u32x4 qf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) qf[s] = *reinterpret_cast<const u32x4*>(Qs + (size_t)(wid * 32 + j) * KRS + s * 32 + hf * 16);
    f32x16 o;
#pragma unroll
    for (int q = 0; q < 16; ++q) o[q] = 0.f;
    float m = -1e30f, l = 0.f;
    for (int kt = 0; kt < wkt; ++kt) {
        f32x16 s;
#pragma unroll
        for (int q = 0; q < 16; ++q) s[q] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const u32x4 kf = *reinterpret_cast<const u32x4*>(Ks + (size_t)(kt * 32 + j) * KRS + ks * 32 + hf * 16);
            Mma<bf16_t>::run(s, kf, qf[ks]);
        }
        // mask + tile max (register q holds key kt*32 + (q&3) + 8*(q>>2) + 4*hf)
        float tm = -1e30f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int key = kt * 32 + (q & 3) + 8 * (q >> 2) + 4 * hf;
            s[q] = key < klim ? s[q] * scale : -1e30f;
            tm = fmaxf(tm, s[q]);
        }
        tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
        const float mn = fmaxf(m, tm);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        float ps = 0.f;
        float p[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            p[q] = s[q] > -1e29f ? __builtin_amdgcn_exp2f(s[q] - mn) : 0.f;
            ps += p[q];
        }
        l = l * alpha + ps;
#pragma unroll
        for (int q = 0; q < 16; ++q) o[q] *= alpha;
        // O^T += V^T P^T : k-step ks uses registers 8ks .. 8ks+7 as the lane's 8 contraction elements
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 pf;
#pragma unroll
            for (int e = 0; e < 4; ++e) pf[e] = pack_bf16x2(p[8 * ks + 2 * e], p[8 * ks + 2 * e + 1]);
            const u32x4 vf = *reinterpret_cast<const u32x4*>(Vt + (size_t)j * vrs + (kt * 32 + ks * 16 + hf * 8) * 2);
            Mma<bf16_t>::run(o, vf, pf);
        }
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = qvalid && l > 0.f ? 1.0f / l : 0.f;
    if (i < a.l_max) {
        bf16_t* op = a.out + ((size_t)r * a.l_max + i) * kD + h * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 v;
            v.x = pack_bf16x2(o[4 * g] * inv, o[4 * g + 1] * inv);
            v.y = pack_bf16x2(o[4 * g + 2] * inv, o[4 * g + 3] * inv);
            *reinterpret_cast<uint2*>(op + 8 * g + 4 * hf) = v;
        }
    }
}

// ---- prompt GEMMs (bf16 weights): Y[M rows][N] = X[M][K] W^T for a few hundred rows ----------------
// The generic LDS-staged kernel pays one exposed memory latency per 128-channel chunk (16 chunks for the
// K = 2048 GEMM: 19 us).  Here nothing is staged: a block owns one 32x32 output tile of one K split, each
// of its 4 waves takes 8 k-steps (128 channels) and has ALL of its operands in flight at once -- weight
// fragments (packed as for tapgemm) and the X rows straight from global in B-fragment layout -- so a
// block costs one memory latency, 8 MFMAs, an LDS reduction and a store.  Splits write raw partial tiles.
// (The GPT prompt pass and batched step moved to bgemm_kernel, t2s_batch.h; this kernel serves the SoVITS side:
// conditioning GEMVs and enc_p's dense layers, gsv_voc.hip.)
struct RowGemmArgs {
    const void* X;        // [M][ldx], float or bf16
    int ldx, M;
    const uint4* W;       // fragments [tap][mtile][kpt][64 lanes]
    int ksteps;           // k-steps per tap (cin / 16)
    int ntaps, pad;       // Conv1d over the rows: tap t reads row + t - pad (zero outside [0, M)); 1 / 0 for a GEMM
    int mtiles;
    const float* bias;    // epilogue (only for nsplit == 1): + bias, ReLU
    int relu;
    void* Y;              // [nsplit][M][ldy], float or bf16
    int ldy;
    size_t split_stride;  // elements between the partial outputs of consecutive splits
    const int* m_dev = nullptr;   // or: the live row count on the device (<= M); row tiles past it leave at once (vocoder conditioning on the
                                  // distinct ge columns: the count is known on the device only)
};

template <typename XT, typename OT, int KPW = 8>       // KPW: k-steps per wave (ntaps * ksteps = 4 * KPW * nsplit)
__global__ __launch_bounds__(256) void rowgemm_kernel(RowGemmArgs a) {
    __shared__ float red[3][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = lane & 31, hf = lane >> 5;
    const int rt = blockIdx.x, mt = blockIdx.y, sp = blockIdx.z;
    if (a.m_dev != nullptr && rt * 32 >= *a.m_dev) return;      // block-uniform
    const int row = rt * 32 + j;
    const int g0 = (sp * 4 + wid) * KPW;                    // first global k-step of the wave: (tap, k-step) = (g / kpt, g % kpt)
    u32x4 wf[KPW], xf[KPW];
    bool xok[KPW];
    int xrow[KPW], xks[KPW];
#pragma unroll
    for (int s = 0; s < KPW; ++s) {
        const int g = g0 + s, tap = g / a.ksteps, kc = g - tap * a.ksteps;
        wf[s] = __builtin_bit_cast(u32x4, a.W[(((size_t)tap * a.mtiles + mt) * a.ksteps + kc) * 64 + lane]);
        const int r = row + tap - a.pad;
        xok[s] = r >= 0 && r < a.M;
        xrow[s] = min(max(r, 0), a.M - 1);
        xks[s] = kc;
    }
    if constexpr (sizeof(XT) == 2) {
        const bf16_t* xp = reinterpret_cast<const bf16_t*>(a.X) + hf * 8;
#pragma unroll
        for (int s = 0; s < KPW; ++s) xf[s] = *reinterpret_cast<const u32x4*>(xp + (size_t)xrow[s] * a.ldx + xks[s] * 16);
    } else {
        const float* xp = reinterpret_cast<const float*>(a.X) + hf * 8;
        f32x4 lo[KPW], hi[KPW];
#pragma unroll
        for (int s = 0; s < KPW; ++s) {
            lo[s] = *reinterpret_cast<const f32x4*>(xp + (size_t)xrow[s] * a.ldx + xks[s] * 16);
            hi[s] = *reinterpret_cast<const f32x4*>(xp + (size_t)xrow[s] * a.ldx + xks[s] * 16 + 4);
        }
#pragma unroll
        for (int s = 0; s < KPW; ++s) {
            xf[s][0] = pack_bf16x2(lo[s][0], lo[s][1]);
            xf[s][1] = pack_bf16x2(lo[s][2], lo[s][3]);
            xf[s][2] = pack_bf16x2(hi[s][0], hi[s][1]);
            xf[s][3] = pack_bf16x2(hi[s][2], hi[s][3]);
        }
    }
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
    for (int s = 0; s < KPW; ++s) {
#pragma unroll
        for (int e = 0; e < 4; ++e) xf[s][e] = xok[s] ? xf[s][e] : 0u;
        Mma<bf16_t>::run(acc, wf[s], xf[s]);
    }
    // the 4 waves' partial tiles are summed in wave order by wave 0
    if (wid > 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) red[wid - 1][q][lane] = acc[q];
    }
    __syncthreads();
    if (wid > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] += red[w][q][lane];
    if (row >= a.M) return;
    const int orow = row;
    const int ch = mt * 32 + 16 * hf;                      // register q = channel ch + q (packer's row permutation)
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        v[q] = acc[q] + (a.bias ? a.bias[ch + q] : 0.f);
        if (a.relu) v[q] = fmaxf(v[q], 0.f);
    }
    if constexpr (sizeof(OT) == 2) {
        bf16_t* yp = reinterpret_cast<bf16_t*>(a.Y) + (size_t)sp * a.split_stride + (size_t)orow * a.ldy + ch;
        u32x4 oa, ob;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            oa[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
            ob[e] = pack_bf16x2(v[8 + 2 * e], v[8 + 2 * e + 1]);
        }
        *reinterpret_cast<u32x4*>(yp) = oa;
        *reinterpret_cast<u32x4*>(yp + 8) = ob;
    } else {
        float* yp = reinterpret_cast<float*>(a.Y) + (size_t)sp * a.split_stride + (size_t)orow * a.ldy + ch;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(yp + 4 * g) = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
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
    int32_t* eos_host;     // null, or the host-mapped mirror of eos_at (not with staged outputs: the commit publishes)
    int slot0, l_max;
    const int32_t* slots;  // state slot of every row, or null = slot0 + row
};

static __global__ __launch_bounds__(128) void t2s_prefill_finish_kernel(PrefillFinishArgs a) {
    const int r = blockIdx.x, c = threadIdx.x * 4;
    const int lx = (int)a.x_lens[r], L = lx + (int)a.y_lens[r];
    const int last = L > 0 ? L - 1 : 0;
    *reinterpret_cast<f32x4*>(a.hlast + (size_t)r * kD + c) =
        *reinterpret_cast<const f32x4*>(a.hidden + ((size_t)r * a.l_max + last) * kD + c);
    if (threadIdx.x == 0) {
        const int slot = a.slots ? a.slots[r] : a.slot0 + r;
        a.kv_len[slot] = L;
        a.x_len[slot] = lx;
        a.step[slot] = 0;
        a.eos_at[slot] = -1;
        eos_publish(a.eos_host, slot, -1);
    }
}

}  // namespace gsv
