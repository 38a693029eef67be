// Batched decode step for gfx950: one token per sequence for B >= kBatchedMin sequences at once.
//
// Reference: T2SBlock.decode_next_token gsv_tts/GPT_SoVITS/GPT/t2s_model.py:67-105 on B rows, i.e. the
// linears of :80-85 (qkv), :97 (out_proj), :100-103 (mlp) as [B x K] x [K x N] contractions, the two
// LayerNorms of :98,104 and the K/V append + SDPA of :87-93.
//
// The step is launch/latency bound up to a few hundred sequences (24 layers x a chain of dependent kernels of a
// few microseconds each), so the design removes dependent launches, not bytes: FIVE launches per layer
//   K1  [LayerNorm2 of the previous layer] -> QKV GEMM (+bias)                 bgemm<PRO_LN, ...>
//   K2  K/V append + softmax(q K^T) V per (head, sequence)                      t2s_batch_attn_kernel
//   K3  out-proj GEMM + bias + residual  -> pre-LN1 rows                        bgemm<PRO_NONE, ...>
//   K4  [LayerNorm1] -> W1 GEMM + bias + ReLU -> hidden (bf16 | fp8)            bgemm<PRO_LN, ...>
//   K5  W2 GEMM over the full K = 2048 + bias + residual -> pre-LN2 rows        bgemm<..., NW = 16>
// (it was seven: the two LayerNorm launches are now the prologue of their consumer -- a GEMM block reads its
// 32 rows over the full K = 512 anyway, so the statistics cost one LDS exchange and no extra traffic -- and the
// 4-way split-K of W2 with its partial tensors is one 16-wave block per tile that reduces in LDS).
//
// A block owns one 32x32 output tile; each of its NW waves owns 8 MFMA k-steps (128 channels) and has ALL of its
// operands in flight at once (weight fragments packed at load + the X rows straight from global in B-fragment
// layout), so a block costs one memory latency, 8 MFMAs, an LDS reduction and a store.
//
// fp8 (BASELINE configs[4]): the QKV / W1 / W2 weights as OCP e4m3 with one fp32 scale per output channel, the
// activations as e4m3 at unit scale (post-LayerNorm rows and ReLU outputs are O(1); saturating), contraction on
// v_mfma_f32_32x32x16_fp8_fp8, fp32 accumulate, scale in the epilogue.  K/V cache and out-proj stay bf16.
#pragma once
#include "t2s_prefill.h"

namespace gsv {

constexpr int PRO_NONE = 0;   // X is the operand (fp32 / bf16 / fp8 rows)
constexpr int PRO_LN = 1;     // X = LayerNorm(pre-LN fp32 rows [M][512]) * g + b, K = 512

typedef uint8_t fp8_t;        // raw OCP e4m3 bits

struct BGemmArgs {
    const void* X;        // [M][ldx]
    int ldx, M;
    const float* lng;     // PRO_LN: LayerNorm weight / bias [512]
    const float* lnb;
    float* xout;          // PRO_LN: the normalised rows [M][512] fp32 (the block's residual later), written by column tile 0; or null
    const uint4* W;       // weight fragments: bf16 [mtile][kstep][64 lanes][16 B]; fp8 [mtile][kstep pair][64 lanes][16 B]
    const float* wscale;  // fp8: dequantisation scale per output channel
    int mtiles, cout;
    const float* bias;    // [cout] or null
    const float* res;     // residual rows fp32 [M][ldres] or null
    int ldres, relu;
    void* Y;              // [M][ldy]: fp32, bf16 or fp8 (saturating e4m3 at unit scale)
    int ldy;
    int cpb;              // column tiles per block (grid.y = ceil(mtiles / cpb)): 1 for the decode step; a prompt pass of many
                          // rows walks 8 column tiles with its X rows (and their LayerNorm) in registers instead of re-reading them
};

__device__ __forceinline__ float clamp_e4m3(float v) { return fminf(fmaxf(v, -448.f), 448.f); }

// 8 floats -> 8 e4m3 bytes (two dwords), round-to-nearest-even in hardware (v_cvt_pk_fp8_f32, OCP on gfx950)
__device__ __forceinline__ void pack_fp8x8(const float (&v)[8], uint32_t& lo, uint32_t& hi) {
    int a = 0, b = 0;
    a = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(v[0]), clamp_e4m3(v[1]), a, false);
    a = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(v[2]), clamp_e4m3(v[3]), a, true);
    b = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(v[4]), clamp_e4m3(v[5]), b, false);
    b = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(v[6]), clamp_e4m3(v[7]), b, true);
    lo = (uint32_t)a; hi = (uint32_t)b;
}

// First channel of the g-th group of 8 channels a lane (half hf) of wave `wid` contributes.  bf16: k-step g of the
// wave, the MFMA's own order (16 channels per k-step, 8 per lane half).  fp8: k-steps are PAIRED so that one 16-byte
// load feeds two MFMAs -- a lane half owns 16 consecutive channels of each 32-channel pair (a contraction may
// enumerate its index in any order as long as both operands agree; the packer uses the same map).
template <bool F8> __device__ __forceinline__ int bg_chan(int wid, int g, int hf) {
    if constexpr (F8) return wid * 128 + (g >> 1) * 32 + hf * 16 + (g & 1) * 8;
    else return wid * 128 + g * 16 + hf * 8;
}

// STG (fp32 rows, 4 waves, bf16 operands): the block's 32 X rows are loaded COALESCED -- a wave reads eight whole rows, 1 KiB per
// instruction -- their LayerNorm statistics are wave-local, and the normalised rows are staged once in LDS as bf16, from where every
// lane takes its B fragments (ds_read_b128, rows 1040 bytes apart).  Without it a lane loads 16 bytes of its own row per
// instruction (32-64 cache lines each on the texture-address path).  Same k order per output as the unstaged form.
template <int PRO, typename XT, typename OT, int NW, bool F8, bool STG = false>
__global__ __launch_bounds__(NW * 64) void bgemm_kernel(BGemmArgs a) {
    static_assert(!STG || (sizeof(XT) == 4 && NW == 4 && !F8), "staged form: fp32 rows, 4 waves, bf16 operands");
    constexpr int NRED = NW == 4 ? 3 * 16 * 64 : NW * 16 * 64;
    constexpr int LDXS = kD + 8;                             // bf16 per staged row
    __shared__ __attribute__((aligned(16))) bf16_t xstage[STG ? 32 * LDXS : 8];
    __shared__ __attribute__((aligned(16))) float red[NRED];
    __shared__ __attribute__((aligned(16))) float gsm[PRO == PRO_LN ? 2 * kD : 4];
    __shared__ float stat[PRO == PRO_LN ? NW * 32 * 2 : 2];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = lane & 31, hf = lane >> 5;
    const int rt = blockIdx.x;
    const int cpb = a.cpb > 1 ? a.cpb : 1;
    const int mt0 = blockIdx.y * cpb;
    int mt = mt0;
    const int row = rt * 32 + j;
    const int rowc = min(row, a.M - 1);
    constexpr int KS = NW * 8;                              // k-steps of the whole contraction

    // ---- everything in flight at once: weight fragments, then the X rows
    u32x4 wf[F8 ? 4 : 8];
    // epilogue operands (bias, fp8 scale, residual) of the lanes that will write the tile: addresses are known now
    constexpr int NEP = NW == 4 ? 4 : 1;                     // f32x4 per writer lane
    const bool writer = NW == 4 ? wid == 0 : tid < 256;
    const int erow = NW == 4 ? row : rt * 32 + (tid >> 3);
    f32x4 e_bias[NEP], e_scale[NEP], e_res[NEP];
    auto load_w = [&](int m) {
        if constexpr (F8) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                wf[p] = __builtin_bit_cast(u32x4, a.W[((size_t)m * (KS / 2) + wid * 4 + p) * 64 + lane]);
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s)
                wf[s] = __builtin_bit_cast(u32x4, a.W[((size_t)m * KS + wid * 8 + s) * 64 + lane]);
        }
    };
    auto load_epi = [&](int m) {
        const int ech = NW == 4 ? m * 32 + 16 * hf : m * 32 + (tid & 7) * 4;
#pragma unroll
        for (int g = 0; g < NEP; ++g) { e_bias[g] = f32x4{0.f, 0.f, 0.f, 0.f}; e_scale[g] = f32x4{1.f, 1.f, 1.f, 1.f}; e_res[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if (writer) {
            if (a.bias) {
#pragma unroll
                for (int g = 0; g < NEP; ++g) e_bias[g] = *reinterpret_cast<const f32x4*>(a.bias + ech + 4 * g);
            }
            if constexpr (F8) {
#pragma unroll
                for (int g = 0; g < NEP; ++g) e_scale[g] = *reinterpret_cast<const f32x4*>(a.wscale + ech + 4 * g);
            }
            if (a.res) {
                const float* rp = a.res + (size_t)min(erow, a.M - 1) * a.ldres + ech;
#pragma unroll
                for (int g = 0; g < NEP; ++g) e_res[g] = *reinterpret_cast<const f32x4*>(rp + 4 * g);
            }
        }
    };
    constexpr bool EPI_LATE = sizeof(XT) == 2;   // bf16 rows (W2): the epilogue operands go behind the X loads -- hipcc copies a component of
                                                 // the freshly loaded bias at once and drains the load counter for it, so with the X loads still
                                                 // to come the writer waves issued them a memory latency late
    load_w(mt);
    if constexpr (!EPI_LATE) load_epi(mt);
    uint32_t xb[8][4];   // bf16: xb[g][0..3] = 8 bf16 of group g; fp8: xb[g][0..1] = 8 e4m3 of group g
    if constexpr (STG) {
        const float* X = reinterpret_cast<const float*>(a.X);
        f32x4 xr[8][2];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int grow = min(rt * 32 + wid * 8 + r, a.M - 1);
#pragma unroll
            for (int c = 0; c < 2; ++c) xr[r][c] = *reinterpret_cast<const f32x4*>(X + (size_t)grow * a.ldx + c * 256 + lane * 4);
        }
        if constexpr (PRO == PRO_LN) {
            f32x4 lg[2], lb[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                lg[c] = *reinterpret_cast<const f32x4*>(a.lng + c * 256 + lane * 4);
                lb[c] = *reinterpret_cast<const f32x4*>(a.lnb + c * 256 + lane * 4);
            }
            float s8[8], q8[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                s8[r] = 0.f; q8[r] = 0.f;
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int i = 0; i < 4; ++i) { s8[r] += xr[r][c][i]; q8[r] = fmaf(xr[r][c][i], xr[r][c][i], q8[r]); }
            }
            const float ts = wave_sumN<8>(s8), tq = wave_sumN<8>(q8);       // lane 32 b2 + 16 b1 + 8 b0 holds row (b2 b1 b0)'s totals
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int src = ((r >> 2) & 1) * 32 + ((r >> 1) & 1) * 16 + (r & 1) * 8;
                const float rs_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ts), src));
                const float rq_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tq), src));
                const float mean = rs_ * (1.0f / kD);
                const float var = fmaxf(rq_ * (1.0f / kD) - mean * mean, 0.f);
                const float rstd = __builtin_amdgcn_rsqf(var + kEps);
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int i = 0; i < 4; ++i) xr[r][c][i] = (xr[r][c][i] - mean) * rstd * lg[c][i] + lb[c][i];
                const int grow = rt * 32 + wid * 8 + r;
                if (mt0 == 0 && a.xout != nullptr && grow < a.M) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) *reinterpret_cast<f32x4*>(a.xout + (size_t)grow * kD + c * 256 + lane * 4) = xr[r][c];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                uint2 pk;
                pk.x = pack_bf16x2(xr[r][c][0], xr[r][c][1]);
                pk.y = pack_bf16x2(xr[r][c][2], xr[r][c][3]);
                *reinterpret_cast<uint2*>(xstage + (wid * 8 + r) * LDXS + c * 256 + lane * 4) = pk;
            }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const u32x4 t = *reinterpret_cast<const u32x4*>(xstage + j * LDXS + bg_chan<false>(wid, g, hf));
            xb[g][0] = t[0]; xb[g][1] = t[1]; xb[g][2] = t[2]; xb[g][3] = t[3];
        }
    } else if constexpr (sizeof(XT) == 4) {
        const float* xp = reinterpret_cast<const float*>(a.X) + (size_t)rowc * a.ldx;
        f32x4 lo[8], hi[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int c = bg_chan<F8>(wid, g, hf);
            lo[g] = *reinterpret_cast<const f32x4*>(xp + c);
            hi[g] = *reinterpret_cast<const f32x4*>(xp + c + 4);
        }
        if constexpr (PRO == PRO_LN) {
            // LayerNorm statistics of the row: this lane holds 64 of its 512 values; the other half of the row's
            // lanes and the other waves meet in LDS (one barrier).  var = E[x^2] - mean^2 as the decode kernels.
            if (tid < 128) {
                *reinterpret_cast<f32x4*>(gsm + tid * 4) = *reinterpret_cast<const f32x4*>(a.lng + tid * 4);
                *reinterpret_cast<f32x4*>(gsm + kD + tid * 4) = *reinterpret_cast<const f32x4*>(a.lnb + tid * 4);
            }
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s += lo[g][i]; q = fmaf(lo[g][i], lo[g][i], q);
                    s += hi[g][i]; q = fmaf(hi[g][i], hi[g][i], q);
                }
            s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 32, 64);
            if (hf == 0) { stat[(wid * 32 + j) * 2] = s; stat[(wid * 32 + j) * 2 + 1] = q; }
            __syncthreads();
            float ts = 0.f, tq = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { ts += stat[(w * 32 + j) * 2]; tq += stat[(w * 32 + j) * 2 + 1]; }
            const float mean = ts * (1.0f / kD);
            const float var = fmaxf(tq * (1.0f / kD) - mean * mean, 0.f);
            const float rs = 1.0f / sqrtf(var + kEps);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int c = bg_chan<F8>(wid, g, hf);
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(gsm + c), g1 = *reinterpret_cast<const f32x4*>(gsm + c + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(gsm + kD + c), b1 = *reinterpret_cast<const f32x4*>(gsm + kD + c + 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    lo[g][i] = (lo[g][i] - mean) * rs * g0[i] + b0[i];
                    hi[g][i] = (hi[g][i] - mean) * rs * g1[i] + b1[i];
                }
                if (mt0 == 0 && a.xout != nullptr && row < a.M) {
                    *reinterpret_cast<f32x4*>(a.xout + (size_t)row * kD + c) = lo[g];
                    *reinterpret_cast<f32x4*>(a.xout + (size_t)row * kD + c + 4) = hi[g];
                }
            }
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if constexpr (F8) {
                const float v[8] = {lo[g][0], lo[g][1], lo[g][2], lo[g][3], hi[g][0], hi[g][1], hi[g][2], hi[g][3]};
                pack_fp8x8(v, xb[g][0], xb[g][1]);
            } else {
                xb[g][0] = pack_bf16x2(lo[g][0], lo[g][1]);
                xb[g][1] = pack_bf16x2(lo[g][2], lo[g][3]);
                xb[g][2] = pack_bf16x2(hi[g][0], hi[g][1]);
                xb[g][3] = pack_bf16x2(hi[g][2], hi[g][3]);
            }
        }
    } else if constexpr (sizeof(XT) == 2) {
        static_assert(sizeof(XT) != 2 || !F8, "bf16 rows feed the bf16 contraction");
        const bf16_t* xp = reinterpret_cast<const bf16_t*>(a.X) + (size_t)rowc * a.ldx;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const u32x4 t = *reinterpret_cast<const u32x4*>(xp + bg_chan<false>(wid, g, hf));
            xb[g][0] = t[0]; xb[g][1] = t[1]; xb[g][2] = t[2]; xb[g][3] = t[3];
        }
        load_epi(mt);
    } else {
        static_assert(sizeof(XT) != 1 || F8, "fp8 rows feed the fp8 contraction");
        const fp8_t* xp = reinterpret_cast<const fp8_t*>(a.X) + (size_t)rowc * a.ldx;
#pragma unroll
        for (int p = 0; p < 4; ++p) {                        // 16 consecutive channels = both k-steps of the pair
            const u32x4 t = *reinterpret_cast<const u32x4*>(xp + bg_chan<true>(wid, 2 * p, hf));
            xb[2 * p][0] = t[0]; xb[2 * p][1] = t[1]; xb[2 * p + 1][0] = t[2]; xb[2 * p + 1][1] = t[3];
        }
    }

    for (int ct = 0; ct < cpb && mt < a.mtiles; ++ct, ++mt) {
    if (ct > 0) { load_w(mt); load_epi(mt); }
    // all loads issued, THEN arithmetic: hipcc otherwise re-uses operand registers and interleaves the later loads
    // with the MFMAs (measured in the ISA: 8 of 16 loads up front), i.e. two exposed memory latencies instead of one
    asm volatile("" : "+v"(wf[0]) : : "memory");
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        if constexpr (F8) {
            const uint64_t av = (uint64_t)wf[g >> 1][(g & 1) * 2] | ((uint64_t)wf[g >> 1][(g & 1) * 2 + 1] << 32);
            const uint64_t bv = (uint64_t)xb[g][0] | ((uint64_t)xb[g][1] << 32);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8((long)av, (long)bv, acc, 0, 0, 0);
        } else {
            const u32x4 bv = {xb[g][0], xb[g][1], xb[g][2], xb[g][3]};
            Mma<bf16_t>::run(acc, wf[g], bv);
        }
    }
    const bool more = ct + 1 < cpb && mt + 1 < a.mtiles;    // block-uniform

    // ---- the NW partial tiles meet in LDS, summed in wave order (bit-reproducible)
    const int chl = 16 * hf;                                 // register q = channel mt*32 + chl + q (packer's row permutation)
    if constexpr (NW == 4) {
        if (wid > 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) red[((wid - 1) * 16 + q) * 64 + lane] = acc[q];
        }
        __syncthreads();
        if (wid == 0) {
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] += red[(w * 16 + q) * 64 + lane];
        if (row < a.M) {
        const int ch = mt * 32 + chl;
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            v[q] = acc[q];
            if constexpr (F8) v[q] *= e_scale[q >> 2][q & 3];
            v[q] += e_bias[q >> 2][q & 3];
            if (a.relu) v[q] = fmaxf(v[q], 0.f);
            v[q] += e_res[q >> 2][q & 3];
        }
        if constexpr (sizeof(OT) == 4) {
            float* yp = reinterpret_cast<float*>(a.Y) + (size_t)row * a.ldy + ch;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(yp + 4 * g) = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
        } else if constexpr (sizeof(OT) == 2) {
            bf16_t* yp = reinterpret_cast<bf16_t*>(a.Y) + (size_t)row * a.ldy + ch;
            u32x4 oa, ob;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                oa[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                ob[e] = pack_bf16x2(v[8 + 2 * e], v[8 + 2 * e + 1]);
            }
            *reinterpret_cast<u32x4*>(yp) = oa;
            *reinterpret_cast<u32x4*>(yp + 8) = ob;
        } else {
            fp8_t* yp = reinterpret_cast<fp8_t*>(a.Y) + (size_t)row * a.ldy + ch;
            uint32_t o0, o1, o2, o3;
            const float v0[8] = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
            const float v1[8] = {v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]};
            pack_fp8x8(v0, o0, o1);
            pack_fp8x8(v1, o2, o3);
            *reinterpret_cast<u32x4*>(yp) = u32x4{o0, o1, o2, o3};
        }
        }   // row < M
        }   // wave 0
        if (!more) return;
        __syncthreads();                                    // the next column tile's partials reuse `red`
    } else {
        // 16 waves: every wave parks its tile, then wave w sums register w over the 16 waves (16 LDS reads instead
        // of 240 by one wave), the tile is transposed through LDS and the first four waves write whole rows
        static_assert(NW == 16, "NW");
#pragma unroll
        for (int q = 0; q < 16; ++q) red[(q * NW + wid) * 64 + lane] = acc[q];
        __syncthreads();
        float r = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) r += red[(wid * NW + w) * 64 + lane];
        __syncthreads();
        red[j * 33 + chl + wid] = r;                        // tile [32 rows][33]: register `wid` of lane (j, hf)
        __syncthreads();
        const int orow = tid >> 3, c4 = (tid & 7) * 4;
        const int grow = rt * 32 + orow;
        if (tid < 256 && grow < a.M) {
        const int ch = mt * 32 + c4;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = red[orow * 33 + c4 + i];
            if constexpr (F8) v[i] *= e_scale[0][i];
            v[i] += e_bias[0][i];
            if (a.relu) v[i] = fmaxf(v[i], 0.f);
            v[i] += e_res[0][i];
        }
        static_assert(NW != 16 || sizeof(OT) == 4, "the 16-wave form writes fp32 rows");
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.Y) + (size_t)grow * a.ldy + ch) = f32x4{v[0], v[1], v[2], v[3]};
        }
        if (!more) return;
        __syncthreads();
    }
    }   // column tiles
}

// ---- the prompt pass at MANY rows ------------------------------------------------------------------------------------------------
// bgemm_kernel is shaped for launch latency: a 32 x 32 tile per block, its four waves split K and meet in LDS, wave 0 writes; a
// block pulls 320 KB for 8.4 MFLOP (26 FLOP per byte), and one CU sustains ~60 GB/s from L2 / the Infinity Cache (its requests in
// flight over a ~1.5 us round trip): 9-11 % MFMA utilisation at the thousands of rows of a packed prompt pass
// (profiles/r04_pmc_prompt_pass_mfma_lds.txt), 8 ms for the first batch of 32 prompts.  bgemm_wide_kernel is the same contraction
// shaped for that rate:
//   * a block owns 32 RT rows (RT = 2: 64) x `cpb` column tiles (groups of 8: 256 columns -- 44 FLOP per byte); the rows are staged ONCE in LDS as
//     bf16 (same coalesced loads, same wave-local LayerNorm, same rounding as the staged form above) and serve every column group
//     the block walks;
//   * a wave owns whole column tiles (two per group) over the full K -- no K split, no LDS meeting, every wave writes its tiles --
//     and streams their weight fragments THREE 8-k-step groups ahead (24 KB per wave, ~96 KB per CU in flight: what 60 GB/s over
//     that round trip takes); an A fragment feeds RT MFMAs.  RT = 4 (128 rows, 65 FLOP per byte) is what the model asks for: hipcc spills
//     in its unrolled walk and a rolled one measured slower (profiles/r04_prompt_pass_wide.txt);
//   * W2 (K = 2048, bf16 hidden rows) walks four 512-channel chunks of its rows through the same LDS tile.
// BIT-IDENTICAL to bgemm_kernel, which a request's K/V rows depend on (they must not change with how many prompts were packed into a
// pass: tests/test_hip_t2s.py::test_packed_prompt_pass_of_many_rows_equals_one_by_one_bf16): an output element is the sum, in
// order, of partial sums over groups of 8 k-steps (128 channels), each accumulated from zero by 8 chained MFMAs -- there a group is
// a wave and the partials meet in LDS in wave order; here `tot += acc` every 8 k-steps in the same order.
#ifndef GSV_WIDE_RT
#define GSV_WIDE_RT 2
#define GSV_WIDE_TPW 2
#define GSV_WIDE_PD 3
#endif
constexpr int kWideRT = GSV_WIDE_RT, kWideTPW = GSV_WIDE_TPW, kWidePD = GSV_WIDE_PD;
static_assert(kWideRT % 2 == 0, "bgemm_wide_kernel stages its fp32 rows two row tiles at a time: an odd GSV_WIDE_RT writes past the stage");
template <int PRO, typename XT, typename OT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void bgemm_wide_kernel(BGemmArgs a) {
    constexpr int RT = kWideRT, TPW = kWideTPW, PD = kWidePD;
    constexpr bool XBF = sizeof(XT) == 2;
    constexpr int LDXS = kD + 8, ROWS = 32 * RT;                          // bf16 per staged row
    const int NCHUNK = XBF ? a.ldx / kD : 1, KST = NCHUNK * 32;          // 512-channel chunks of the rows (W2: 4; bf16 attention rows: 1); k-steps of the contraction
    static_assert(PRO == PRO_NONE || !XBF, "LayerNorm prologue: fp32 rows");
    extern __shared__ __attribute__((aligned(16))) unsigned char wide_lds[];
    bf16_t* xstage = reinterpret_cast<bf16_t*>(wide_lds);                                   // [ROWS][LDXS]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, j = lane & 31, hf = lane >> 5;
    const int rbase = blockIdx.x * ROWS;
    const int ngroups = NCHUNK > 1 ? 1 : max(1, a.cpb / (4 * TPW));     // column groups this block walks (W2: one, its chunks re-stage the tile)
    const int mtb = blockIdx.y * (NCHUNK > 1 ? 4 * TPW : a.cpb);        // the block's first column tile
    f32x16 tot[TPW][RT];

    auto stage = [&](int c) {
        if constexpr (!XBF) {
            const float* X = reinterpret_cast<const float*>(a.X);
            f32x4 lg[2], lb[2];
            if constexpr (PRO == PRO_LN) {
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    lg[cc] = *reinterpret_cast<const f32x4*>(a.lng + cc * 256 + lane * 4);
                    lb[cc] = *reinterpret_cast<const f32x4*>(a.lnb + cc * 256 + lane * 4);
                }
            }
#pragma unroll
            for (int t2 = 0; t2 < RT; t2 += 2) {
                // two row tiles' loads in flight at once (16 rows per wave): half the memory round trips of a tile at a time
                f32x4 xall[2][8][2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const int grow = min(rbase + 32 * (t2 + u) + wid * 8 + r, a.M - 1);
#pragma unroll
                        for (int cc = 0; cc < 2; ++cc) xall[u][r][cc] = *reinterpret_cast<const f32x4*>(X + (size_t)grow * a.ldx + cc * 256 + lane * 4);
                    }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = t2 + u;
                    f32x4 (&xr)[8][2] = xall[u];
                    if constexpr (PRO == PRO_LN) {
                        float s8[8], q8[8];
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            s8[r] = 0.f; q8[r] = 0.f;
#pragma unroll
                            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                                for (int i = 0; i < 4; ++i) { s8[r] += xr[r][cc][i]; q8[r] = fmaf(xr[r][cc][i], xr[r][cc][i], q8[r]); }
                        }
                        const float ts = wave_sumN<8>(s8), tq = wave_sumN<8>(q8);
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            const int src = ((r >> 2) & 1) * 32 + ((r >> 1) & 1) * 16 + (r & 1) * 8;
                            const float rs_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ts), src));
                            const float rq_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tq), src));
                            const float mean = rs_ * (1.0f / kD);
                            const float var = fmaxf(rq_ * (1.0f / kD) - mean * mean, 0.f);
                            const float rstd = __builtin_amdgcn_rsqf(var + kEps);
#pragma unroll
                            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                                for (int i = 0; i < 4; ++i) xr[r][cc][i] = (xr[r][cc][i] - mean) * rstd * lg[cc][i] + lb[cc][i];
                            const int grow = rbase + 32 * t + wid * 8 + r;
                            if (blockIdx.y == 0 && a.xout != nullptr && grow < a.M) {
#pragma unroll
                                for (int cc = 0; cc < 2; ++cc) *reinterpret_cast<f32x4*>(a.xout + (size_t)grow * kD + cc * 256 + lane * 4) = xr[r][cc];
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 8; ++r)
#pragma unroll
                        for (int cc = 0; cc < 2; ++cc) {
                            uint2 pk;
                            pk.x = pack_bf16x2(xr[r][cc][0], xr[r][cc][1]);
                            pk.y = pack_bf16x2(xr[r][cc][2], xr[r][cc][3]);
                            *reinterpret_cast<uint2*>(xstage + (32 * t + wid * 8 + r) * LDXS + cc * 256 + lane * 4) = pk;
                        }
                }
            }
        } else {
            const bf16_t* X = reinterpret_cast<const bf16_t*>(a.X);
            u32x4 xr[RT][8];
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int grow = min(rbase + 32 * t + wid * 8 + r, a.M - 1);
                    xr[t][r] = *reinterpret_cast<const u32x4*>(X + (size_t)grow * a.ldx + c * kD + lane * 8);
                }
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int r = 0; r < 8; ++r) *reinterpret_cast<u32x4*>(xstage + (32 * t + wid * 8 + r) * LDXS + lane * 8) = xr[t][r];
        }
    };

    for (int cg = 0; cg < ngroups; ++cg) {
        const int mt0 = mtb + cg * (4 * TPW);
        if (mt0 >= a.mtiles) break;                                      // block-uniform
        for (int c = 0; c < NCHUNK; ++c) {
            if (cg == 0 || NCHUNK > 1) {
                if (c > 0) __syncthreads();                              // everyone is done with the previous chunk's rows
                stage(c);
                __syncthreads();
            }
            // ONE flat walk over (tile, 8-k-step group) with the weight fragments of the next PD groups in flight
            constexpr int NG = TPW * 4;
            u32x4 wf[PD + 1][8];
            auto wbase = [&](int n) {
                const int mt = min(mt0 + (n >> 2) * 4 + wid, a.mtiles - 1);
                return a.W + ((size_t)mt * KST + c * 32 + (n & 3) * 8) * 64 + lane;
            };
#pragma unroll
            for (int n = 0; n < PD; ++n) {
                const uint4* wp = wbase(n);
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) wf[n][s8] = __builtin_bit_cast(u32x4, wp[(size_t)s8 * 64]);
            }
#pragma unroll
            for (int n = 0; n < NG; ++n) {
                const int ti = n >> 2, g = n & 3;
                // (the fence keeps hipcc from hoisting every later group's loads up here: 64 fragments = 256 registers in flight, and spills)
                asm volatile("" : : : "memory");
                if (n + PD < NG) {
                    const uint4* wp = wbase(n + PD);
#pragma unroll
                    for (int s8 = 0; s8 < 8; ++s8) wf[(n + PD) % (PD + 1)][s8] = __builtin_bit_cast(u32x4, wp[(size_t)s8 * 64]);
                }
                asm volatile("" : : : "memory");
                f32x16 acc[RT];
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) {
#pragma unroll
                    for (int t = 0; t < RT; ++t) {
                        const u32x4 bv = *reinterpret_cast<const u32x4*>(xstage + (32 * t + j) * LDXS + (g * 8 + s8) * 16 + hf * 8);
                        Mma<bf16_t>::run(acc[t], wf[n % (PD + 1)][s8], bv);
                    }
                }
                // the partial sum of this 128-channel group joins the running total (the wave order of bgemm_kernel)
                if (c == 0 && g == 0) {
#pragma unroll
                    for (int t = 0; t < RT; ++t) tot[ti][t] = acc[t];
                } else {
#pragma unroll
                    for (int t = 0; t < RT; ++t)
#pragma unroll
                        for (int q = 0; q < 16; ++q) tot[ti][t][q] += acc[t][q];
                }
            }
        }
        // ---- epilogue: bias, ReLU, residual, store; lane (j, hf) holds channels mt * 32 + 16 hf + q of row 32 t + j
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti) {
            const int mt = mt0 + ti * 4 + wid;
            if (mt >= a.mtiles) continue;
            const int ch = mt * 32 + 16 * hf;
            f32x4 e_bias[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) e_bias[g] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + ch + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                const int row = rbase + 32 * t + j;
                if (row >= a.M) continue;
                float v[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    v[q] = tot[ti][t][q];
                    v[q] += e_bias[q >> 2][q & 3];
                    if (a.relu) v[q] = fmaxf(v[q], 0.f);
                }
                if (a.res) {
                    const float* rp = a.res + (size_t)row * a.ldres + ch;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 r4 = *reinterpret_cast<const f32x4*>(rp + 4 * g);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * g + e] += r4[e];
                    }
                }
                if constexpr (sizeof(OT) == 4) {
                    float* yp = reinterpret_cast<float*>(a.Y) + (size_t)row * a.ldy + ch;
#pragma unroll
                    for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(yp + 4 * g) = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
                } else {
                    bf16_t* yp = reinterpret_cast<bf16_t*>(a.Y) + (size_t)row * a.ldy + ch;
                    u32x4 oa, ob;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        oa[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                        ob[e] = pack_bf16x2(v[8 + 2 * e], v[8 + 2 * e + 1]);
                    }
                    *reinterpret_cast<u32x4*>(yp) = oa;
                    *reinterpret_cast<u32x4*>(yp + 8) = ob;
                }
            }
        }
    }
}
constexpr size_t kWideLds = (size_t)32 * kWideRT * (kD + 8) * sizeof(bf16_t);

// ---- attention of the batched step ------------------------------------------------------------------
// One block per (head, sequence): append the new K/V row (rounded through the cache type, t2s_model.py:87-88), then
// softmax(q K^T / sqrt(32)) V over positions [0, kv_len[b]] (the token attends to itself).
// Shape, each point measured on MI355X (B = 64, kv 200-300, 28 MB of K/V per launch):
//  * four lanes per 64-byte row, 16 bytes per lane: a wave's load instruction covers 1 KiB of CONSECUTIVE bytes.  One
//    thread per key (a lane reading its own 64-byte row) touches 64 cache lines per instruction and was bound by the
//    texture-address path at 2 TB/s (13 us per launch) whatever else the kernel did;
//  * the row addresses depend on nothing this kernel computes (rows past kv_len are loaded from clamped, valid
//    addresses and masked), so kv_len, the q / k / v row and EVERY K and V chunk of the thread are in flight at kernel
//    entry: one memory latency;
//  * per-wave softmax statistics (no block-wide max), the wave's P.V reduced by halving exchanges on the cross-lane
//    network (v_permlane32_swap / v_permlane16_swap / DPP), one LDS meeting of the four waves at the end.
template <typename WT>
struct BatchAttnArgs {
    const float* qkv;        // [B][1536]
    WT* kc;                  // this layer: [B][16][T][32]
    WT* vc;
    const int64_t* kv_len;   // [B]
    int T;
    float* out;              // [B][512]
    unsigned long long* dbg; // bring-up aid (gsv_t2s_set_debug): block (0, 0) stamps its phases into slots 0-6
};

template <int NIT, bool BLIND>   // iterations of 64 rows: T <= 64 * NIT; BLIND: load every chunk, mask after (else: kv_len first)
__global__ __launch_bounds__(256) void t2s_batch_attn_kernel(BatchAttnArgs<bf16_t> a) {
    __shared__ __attribute__((aligned(16))) float qs[32], kn[32], vn[32], pacc[4][32];
    __shared__ float pm[4], pl[4];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int part = tid & 3, rsub = tid >> 2;               // 16-byte quarter of the row; row within an iteration
    const float* row = a.qkv + (size_t)b * 1536 + h * 32;
    bf16_t* Kp = a.kc + (((size_t)b * kH + h) * a.T) * kDh;
    bf16_t* Vp = a.vc + (((size_t)b * kH + h) * a.T) * kDh;
    // ---- everything in flight: kv_len, this head's q / k / v, the thread's K and V chunks
    const int64_t n64 = a.kv_len[b];
    float rq = 0.f, rk = 0.f, rv = 0.f;
    if (tid < 32) { rq = row[tid]; rk = row[512 + tid]; rv = row[1024 + tid]; }
    // !BLIND: rows past kv_len are loaded UNCONDITIONALLY from the last live row (a cache hit, no HBM bytes) and
    // masked at use -- a per-load "if live" makes hipcc branch around each load and drain vmcnt(0) between them
    int lastrow = a.T - 1;
    if constexpr (!BLIND) lastrow = max((int)(n64 < 0 ? 0 : (n64 > a.T - 1 ? a.T - 1 : n64)) - 1, 0);
    raw16 kr[NIT], vr[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) kr[it] = ldg16(Kp + (size_t)min(it * 64 + rsub, lastrow) * kDh + part * 8);
#pragma unroll
    for (int it = 0; it < NIT; ++it) vr[it] = ldg16(Vp + (size_t)min(it * 64 + rsub, lastrow) * kDh + part * 8);
    asm volatile("" : "+v"(rq) : : "memory");
    const int n = (int)(n64 < 0 ? 0 : (n64 > a.T - 1 ? a.T - 1 : n64));     // position of the new token
    if (tid < 32) {
        qs[tid] = rq;
        const bf16_t kq = f32_to_bf16(rk), vq = f32_to_bf16(rv);
        kn[tid] = bf16_to_f32(kq); vn[tid] = bf16_to_f32(vq);
        const int nw = n64 < 0 ? a.T - 1 : n;     // parked slot (kv_len < 0): away from the rows a staged refill writes
        Kp[(size_t)nw * kDh + tid] = kq; Vp[(size_t)nw * kDh + tid] = vq;
    }
    __syncthreads();
    float q[8], knr[8], vnr[8];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(qs + part * 8 + 4 * c);
        const f32x4 u = *reinterpret_cast<const f32x4*>(kn + part * 8 + 4 * c);
        const f32x4 w = *reinterpret_cast<const f32x4*>(vn + part * 8 + 4 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { q[4 * c + e] = t[e]; knr[4 * c + e] = u[e]; vnr[4 * c + e] = w[e]; }
    }
    const float scale = 0.17677669529663687f;  // 1/sqrt(32)
    float sc[NIT + 1];
    float mx = -INFINITY;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        float kk[8];
        Unpack<bf16_t, 8>::run(kr[it], kk);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(q[e], kk[e], s);
        s = quad_sum(s);
        sc[it] = it * 64 + rsub < n ? s * scale : -INFINITY;     // rows [0, n): the cache; row n is the new token, below
        mx = fmaxf(mx, sc[it]);
    }
    {   // the new token's own key / value ride with the first quad of wave 0 (from LDS, never from the row being written)
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(q[e], knr[e], s);
        s = quad_sum(s);
        sc[NIT] = tid < 4 ? s * scale : -INFINITY;
        mx = fmaxf(mx, sc[NIT]);
    }
    mx = wave_max(mx);
    const float mref = mx == -INFINITY ? 0.f : mx;               // a wave without live rows
    float l = 0.f, acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const bool live = sc[it] != -INFINITY;
        const float p = live ? __expf(sc[it] - mref) : 0.f;
        float vv[8];
        Unpack<bf16_t, 8>::run(vr[it], vv);
        l += p;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(p, live ? vv[e] : 0.f, acc[e]);   // select: never multiply a stale row
    }
    {
        const float p = sc[NIT] != -INFINITY ? __expf(sc[NIT] - mref) : 0.f;
        l += p;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(p, vnr[e], acc[e]);
    }
    l = wave_sum(part == 0 ? l : 0.f);
    float r4[4], r2[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) r4[i] = halve32_sum(acc[i], acc[i + 4]);
#pragma unroll
    for (int i = 0; i < 2; ++i) r2[i] = halve16_sum(r4[i], r4[i + 2]);
    float r1 = halve8_sum(r2[0], r2[1]);
    r1 += lane_xor<4>(r1);
    if ((lane & 4) == 0) pacc[wid][part * 8 + 4 * (lane >> 5) + 2 * ((lane >> 4) & 1) + ((lane >> 3) & 1)] = r1;
    if (lane == 0) { pm[wid] = mx; pl[wid] = l; }
    __syncthreads();
    if (tid < 32) {
        const float M = fmaxf(fmaxf(pm[0], pm[1]), fmaxf(pm[2], pm[3]));
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = __expf(pm[w] - M);                   // exp(-inf) = 0 for an empty wave
            num = fmaf(pacc[w][tid], f, num);
            den = fmaf(pl[w], f, den);
        }
        a.out[(size_t)b * kD + h * 32 + tid] = num / den;
    }
}

// ---- fp8 weight packing --------------------------------------------------------------------------
// scale[m] = max_c |W[m][c]| / 448 (1 for an all-zero row)
static __global__ __launch_bounds__(256) void fp8_row_scale_kernel(const float* __restrict__ W, int cin, float* __restrict__ scale, int cout) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= cout) return;
    float mx = 0.f;
    for (int c = lane; c < cin; c += 64) mx = fmaxf(mx, fabsf(W[(size_t)m * cin + c]));
    mx = wave_max(mx);
    if (lane == 0) scale[m] = mx > 0.f ? mx * (1.0f / 448.f) : 1.0f;
}

// dst [mtile][kstep pair][64 lanes][16 bytes]: byte e of lane (mr, hf) = W[m(mr)][pair*32 + hf*16 + e] / scale[m],
// the same row permutation as tapgemm_pack_kernel (a lane's 16 D registers are 16 consecutive channels)
static __global__ __launch_bounds__(256) void fp8_pack_kernel(const float* __restrict__ W, const float* __restrict__ scale,
                                                       uint32_t* __restrict__ dst, int cout, int cin, int mtiles) {
    const int npair = cin / 32;
    const size_t total = (size_t)mtiles * npair * 64 * 4;    // dwords
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        size_t r = idx;
        const int d = r % 4; r /= 4;
        const int lane = r % 64; r /= 64;
        const int p = r % npair; r /= npair;
        const int mt = (int)r;
        const int mr = lane & 31, hf = lane >> 5;
        const int m = mt * 32 + 16 * ((mr >> 2) & 1) + (mr & 3) + 4 * (mr >> 3);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (m < cout) {
            const float inv = 1.0f / scale[m];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = clamp_e4m3(W[(size_t)m * cin + p * 32 + hf * 16 + d * 4 + i] * inv);
        }
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
        dst[idx] = (uint32_t)w;
    }
}

}  // namespace gsv
