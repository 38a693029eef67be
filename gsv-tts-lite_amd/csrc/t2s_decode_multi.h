// Decode-step kernels for a FEW TENS of sequences: the 2-kernels-per-layer step of t2s_decode.h with R sequences per
// block.
//
// Why: the per-sequence kernels launch 16 x B (attention) and 32 x B (FFN) blocks of 1024 threads, one block per CU:
// beyond B = 16 they run in rounds and re-stream every layer's weights once per sequence.  The batched chain
// (t2s_batch.h) streams weights once but is FIVE dependent launches per layer of ~5-8 us each, so it only wins from
// ~40 sequences on.  In between (BASELINE configs[2] is 32 slots) a block here keeps its weight rows in registers and
// applies them to R sequences: 16 x B/R and 32 x B/R blocks -- one round for B = 32 with R = 2 / 4 -- and still two
// launches per layer.  Same arithmetic per sequence as t2s_decode.h, same fixed summation orders (bit-identical
// results: the fp32 parity tests cover both).
#pragma once
#include "t2s_decode.h"

namespace gsv {

// LayerNorm of R 512-vectors at once, thread t < 512 owning element t of each: ONE barrier for all of them.
// `red` holds R * 2 * kNW floats.
template <int R>
__device__ __forceinline__ void ln512_multi(float (&v)[R], bool owner, float g, float bta, float* red) {
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float s = owner ? v[r] : 0.f, q = s * s;
        s = wave_sum(s);
        q = wave_sum(q);
        if ((threadIdx.x & 63) == 0) { red[(r * kNW + w) * 2] = s; red[(r * kNW + w) * 2 + 1] = q; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float ts = 0.f, tq = 0.f;
#pragma unroll
        for (int i = 0; i < kNW; ++i) { ts += red[(r * kNW + i) * 2]; tq += red[(r * kNW + i) * 2 + 1]; }
        const float mean = ts * (1.0f / kD);
        const float var = fmaxf(tq * (1.0f / kD) - mean * mean, 0.f);
        const float rs = 1.0f / sqrtf(var + kEps);
        v[r] = (v[r] - mean) * rs * g + bta;
    }
}

template <int R> constexpr int attn_multi_lds_floats() {
    return R * kD + R * 96 + R * 32 + R * 2 * kNW + R * kNW * 32 + R * 2 * kNW + R * kNW * kD;
}

// grid (16 heads, ceil(B / R)); sequences b0 .. b0 + R - 1 of the block (clamped to B - 1: a short last block repeats
// its last sequence and does not write the repeats)
template <typename WT, int MODE, int R>
__global__ __launch_bounds__(kNT) void t2s_attn_multi_kernel(AttnArgs<WT> a, int B) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                    // [R][512]
    float* qkv = xs + R * kD;            // [R][96]
    float* att = qkv + R * 96;           // [R][32]
    float* red = att + R * 32;           // [R][2*16]
    float* pacc = red + R * 2 * kNW;     // [R][16][32]
    float* pm = pacc + R * kNW * 32;     // [R][16]
    float* pl = pm + R * kNW;            // [R][16]
    float* stage = pl + R * kNW;         // [R][16][512]
    const int h = blockIdx.x, b0 = blockIdx.y * R, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int EPL = Geo<WT>::EPL;
    constexpr int CPR = Geo<WT>::CPR;
    constexpr int LPR = kDh / EPL;
    constexpr int RPI = kNT / LPR;
    constexpr int KCH = 2;
    constexpr int RW = 96 / kNW;
    const bool owner = tid < kD;
    const int part = tid % LPR, rsub = tid / LPR;

    int bs[R], n[R], nw[R];
    WT* Kp[R];
    WT* Vp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        bs[r] = min(b0 + r, B - 1);
        int nn = (int)a.kv_len[bs[r]];
        n[r] = min(max(nn, 0), a.T - 1);
        nw[r] = nn < 0 ? a.T - 1 : n[r];          // parked slot: see t2s_decode.h
        Kp[r] = a.kc + (((size_t)bs[r] * kH + h) * a.T) * kDh;
        Vp[r] = a.vc + (((size_t)bs[r] * kH + h) * a.T) * kDh;
    }

    // ---- issue everything whose address is known now, in consumption order (t2s_decode.h, "latency discipline")
    PartialSum<kNJ, typename Geo<WT>::PT> ps[R];
    float xd[R];
    StepTokLoads tl[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        xd[r] = 0.f;
        tl[r].tp.v = 0.f;
        if constexpr (MODE == 0) {
            if (owner) xd[r] = a.xdirect[(size_t)bs[r] * kD + tid];
        } else if constexpr (MODE == 2) {
            tl[r] = steptok_issue(a.tk, bs[r], lane);
        } else {
            ps[r].issue(a.zpart + (size_t)bs[r] * kNJ * kD, a.b2, a.x1 + (size_t)bs[r] * kD, a.ln2g, a.ln2b);
        }
    }
    if constexpr (MODE == 1) __builtin_amdgcn_s_barrier();
    asm volatile("" : : : "memory");
    const WT* wp = a.wqkv + ((size_t)h * 96 + wid * RW) * kD;
    raw16 wq[RW][CPR];
#pragma unroll
    for (int r = 0; r < RW; ++r) row_load<WT>(wp + (size_t)r * kD, wq[r]);
    raw16 kreg[KCH], vreg[KCH];          // sequence 0's first chunk; the later sequences' chunks are issued as registers free up
#pragma unroll
    for (int it = 0; it < KCH; ++it) kreg[it] = ldg16(Kp[0] + (size_t)min(rsub + it * RPI, n[0]) * kDh + part * EPL);
#pragma unroll
    for (int it = 0; it < KCH; ++it) vreg[it] = ldg16(Vp[0] + (size_t)min(rsub + it * RPI, n[0]) * kDh + part * EPL);
    Panel<WT, kDh> po;
    // fp32 handles: the out-proj panel is requested behind the q / k / v dots, when their weight registers are free (t2s_decode.h: at entry
    // the two together spill)
    if constexpr (sizeof(WT) == 2) po.issue(a.wo + (size_t)h * kD * kDh);
    const int oi = sumN_index<8>();
    const float bq = a.bqkv[h * 96 + wid * RW + min(oi, RW - 1)];
    if constexpr (MODE == 0) asm volatile("" : "+v"(xd[0]) : : "memory");
    else if constexpr (MODE == 2) asm volatile("" : "+v"(tl[0].tp.v) : : "memory");
    else asm volatile("" : "+v"(ps[0].p[0][0]) : : "memory");

    // ---- layer inputs of the R sequences
    float v[R];
    if constexpr (MODE == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = xd[r];
    } else if constexpr (MODE == 2) {   // the token kernel's work as this kernel's prologue (t2s_decode.h, StepTok)
#pragma unroll
        for (int r = 0; r < R; ++r)
            v[r] = steptok_finish(a.tk, tl[r], bs[r], lane, tid, owner, a.kv_len[bs[r]], a.T, h == 0 && tid == 0 && b0 + r < B);
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) ps[r].park(stage + (size_t)r * kNW * kD);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = owner ? ps[r].finish(stage + (size_t)r * kNW * kD) : 0.f;
        ln512_multi<R>(v, owner, ps[0].lng, ps[0].lnb, red);
    }
    if (owner) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            xs[r * kD + tid] = v[r];
            if (h == 0 && b0 + r < B) a.xout[(size_t)(b0 + r) * kD + tid] = v[r];
        }
    }
    __syncthreads();

    // ---- q, k, v of this head for every sequence from the SAME weight registers
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float xr[8];
        lane_x<WT>(xs + r * kD, xr);
        float acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = u < RW ? row_dot<WT>(wq[u < RW ? u : 0], xr) : 0.f;
        const float tot = wave_sumN<8>(acc);
        if ((lane & 7) == 0 && oi < RW) {
            const int row = wid * RW + oi;
            float val = tot + bq;
            if (row >= 32) {
                const WT s = from_f32<WT>(val);
                val = to_f32<WT>(s);
                if (b0 + r < B) {
                    if (row < 64) Kp[r][(size_t)nw[r] * kDh + row - 32] = s; else Vp[r][(size_t)nw[r] * kDh + row - 64] = s;
                }
            }
            qkv[r * 96 + row] = val;
        }
    }
    __syncthreads();
    if constexpr (sizeof(WT) == 4) po.issue(a.wo + (size_t)h * kD * kDh);

    // ---- single-pass attention per sequence; per-wave (max, sum, P.V) parked per sequence, merged once at the end
    const float scale = 0.17677669529663687f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float* qk = qkv + r * 96;
        float qr[EPL];
#pragma unroll
        for (int i = 0; i < EPL; ++i) qr[i] = qk[part * EPL + i];
        float m_run = -INFINITY, l_run = 0.f, acc[EPL];
#pragma unroll
        for (int i = 0; i < EPL; ++i) acc[i] = 0.f;
        const int nr = n[r];
        for (int c0 = 0; c0 == 0 || c0 < nr; c0 += KCH * RPI) {
            if (c0 > 0) {
#pragma unroll
                for (int it = 0; it < KCH; ++it) {
                    kreg[it] = ldg16(Kp[r] + (size_t)min(c0 + rsub + it * RPI, nr) * kDh + part * EPL);
                    vreg[it] = ldg16(Vp[r] + (size_t)min(c0 + rsub + it * RPI, nr) * kDh + part * EPL);
                }
            }
            float sv[KCH + 1];
            float cmax = -INFINITY;
            raw16 vcur[KCH];
#pragma unroll
            for (int it = 0; it < KCH; ++it) {
                const int rr = c0 + rsub + it * RPI;
                float kk[EPL];
                Unpack<WT, EPL>::run(kreg[it], kk);
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < EPL; ++i) s = fmaf(qr[i], kk[i], s);
                s = group_sum<LPR>(s);
                sv[it] = rr < nr ? s * scale : -INFINITY;
                cmax = fmaxf(cmax, sv[it]);
                vcur[it] = vreg[it];
            }
            // the next sequence's first chunk goes out as soon as this one's registers are consumed
            if (r + 1 < R && c0 + KCH * RPI >= nr) {
#pragma unroll
                for (int it = 0; it < KCH; ++it) {
                    kreg[it] = ldg16(Kp[r + 1 < R ? r + 1 : r] + (size_t)min(rsub + it * RPI, n[r + 1 < R ? r + 1 : r]) * kDh + part * EPL);
                    vreg[it] = ldg16(Vp[r + 1 < R ? r + 1 : r] + (size_t)min(rsub + it * RPI, n[r + 1 < R ? r + 1 : r]) * kDh + part * EPL);
                }
            }
            {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < EPL; ++i) s = fmaf(qr[i], qk[32 + part * EPL + i], s);
                s = group_sum<LPR>(s);
                sv[KCH] = (c0 == 0 && tid < LPR) ? s * scale : -INFINITY;
                cmax = fmaxf(cmax, sv[KCH]);
            }
            cmax = wave_max(cmax);
            const float m_new = fmaxf(m_run, cmax);
            const float mref = (m_new == -INFINITY) ? 0.f : m_new;
            const float f = expf(m_run - mref);
            l_run *= f;
#pragma unroll
            for (int i = 0; i < EPL; ++i) acc[i] *= f;
#pragma unroll
            for (int it = 0; it < KCH; ++it) {
                const float p = expf(sv[it] - mref);
                const bool live = sv[it] != -INFINITY;
                float vv[EPL];
                Unpack<WT, EPL>::run(vcur[it], vv);
                if (part == 0) l_run += p;
#pragma unroll
                for (int i = 0; i < EPL; ++i) acc[i] = fmaf(p, live ? vv[i] : 0.f, acc[i]);
            }
            {
                const float p = expf(sv[KCH] - mref);
                if (part == 0) l_run += p;
#pragma unroll
                for (int i = 0; i < EPL; ++i) acc[i] = fmaf(p, qk[64 + part * EPL + i], acc[i]);
            }
            m_run = m_new;
        }
        l_run = wave_sum(l_run);
        float* pa = pacc + (size_t)r * kNW * 32 + wid * 32;
        if constexpr (EPL == 8) {
            float r4[4], r2[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) r4[i] = halve32_sum(acc[i], acc[i + 4]);
#pragma unroll
            for (int i = 0; i < 2; ++i) r2[i] = halve16_sum(r4[i], r4[i + 2]);
            float r1 = halve8_sum(r2[0], r2[1]);
            r1 += lane_xor<4>(r1);
            if ((lane & 4) == 0) pa[part * 8 + 4 * (lane >> 5) + 2 * ((lane >> 4) & 1) + ((lane >> 3) & 1)] = r1;
        } else {
            float r2[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) r2[i] = halve32_sum(acc[i], acc[i + 2]);
            float r1 = halve16_sum(r2[0], r2[1]);
            r1 += lane_xor<8>(r1);
            if ((lane & 8) == 0) pa[part * 4 + 2 * (lane >> 5) + ((lane >> 4) & 1)] = r1;
        }
        if (lane == 0) { pm[r * kNW + wid] = m_run; pl[r * kNW + wid] = l_run; }
    }
    __syncthreads();
    if (wid < R) {   // wave r merges sequence r's 16 waves
        const int r = wid;
        const float mw = pm[r * kNW + (lane & 15)], lw = pl[r * kNW + (lane & 15)];
        const float M = row16_max(mw);
        const float den = row16_sum(lw * expf(mw - M));
        const int hf = lane >> 5, d = lane & 31;
        float num = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) num = fmaf(pacc[((size_t)r * kNW + hf * 8 + w) * 32 + d], expf(pm[r * kNW + hf * 8 + w] - M), num);
        num = xor32_sum(num);
        if (lane < 32) att[r * 32 + d] = num / den;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (b0 + r < B) po.finish(att + r * 32, a.ypart + ((size_t)(b0 + r) * kH + h) * kD);
    }
}

template <int R> constexpr int ffn_multi_lds_floats() { return R * kD + R * kFJ + R * 2 * kNW + R * kNW * kD; }

// grid (32 slices, ceil(B / R))
template <typename WT, int R>
__global__ __launch_bounds__(kNT) void t2s_ffn_multi_kernel(FfnArgs<WT> a, int B) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                    // [R][512]
    float* hb = xs + R * kD;             // [R][64]
    float* red = hb + R * kFJ;           // [R][2*16]
    float* stage = red + R * 2 * kNW;    // [R][16][512]
    const int j = blockIdx.x, b0 = blockIdx.y * R, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int CPR = Geo<WT>::CPR;
    constexpr int RW = kFJ / kNW;
    const bool owner = tid < kD;

    PartialSum<kH, typename Geo<WT>::PT> ps[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int b = min(b0 + r, B - 1);
        ps[r].issue(a.ypart + (size_t)b * kH * kD, a.bo, a.x + (size_t)b * kD, a.ln1g, a.ln1b);
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" : : : "memory");
    const int row0 = j * kFJ + wid * RW;
    raw16 w1r[RW][CPR];
#pragma unroll
    for (int r = 0; r < RW; ++r) row_load<WT>(a.w1 + (size_t)(row0 + r) * kD, w1r[r]);
    Panel<WT, kFJ> p2;
    p2.issue(a.w2p + (size_t)j * kD * kFJ);
    const int oi = sumN_index<RW>();
    const float b1r = a.b1[row0 + oi];
    asm volatile("" : "+v"(ps[0].p[0][0]) : : "memory");

#pragma unroll
    for (int r = 0; r < R; ++r) ps[r].park(stage + (size_t)r * kNW * kD);
    __syncthreads();
    float v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = owner ? ps[r].finish(stage + (size_t)r * kNW * kD) : 0.f;
    ln512_multi<R>(v, owner, ps[0].lng, ps[0].lnb, red);
    if (owner) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            xs[r * kD + tid] = v[r];
            if (j == 0 && b0 + r < B) a.x1out[(size_t)(b0 + r) * kD + tid] = v[r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float xr[8];
        lane_x<WT>(xs + r * kD, xr);
        float acc[RW];
#pragma unroll
        for (int u = 0; u < RW; ++u) acc[u] = row_dot<WT>(w1r[u], xr);
        const float tot = wave_sumN<RW>(acc);
        if ((lane & (64 / RW - 1)) == 0) hb[r * kFJ + wid * RW + oi] = fmaxf(tot + b1r, 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (b0 + r < B) p2.finish(hb + r * kFJ, a.zpart + ((size_t)(b0 + r) * kNJ + j) * kD);
    }
}

}  // namespace gsv
