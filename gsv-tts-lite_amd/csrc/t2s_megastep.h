// Persistent decode step for small batches: ALL 24 layers in ONE launch.
//
// The 2-kernels-per-layer graph (t2s_decode.h) pays, per layer, two kernel boundaries (~1.6 us
// each) and two cold starts in which the weight stream cannot begin until the kernel exists.  Here
// 48 co-resident blocks per sequence (16 "attention" roles = heads, 32 "ffn" roles = FFN slices;
// 48*B <= 256 CUs, one block per CU) walk the layers themselves.  The two all-to-all exchanges of
// a layer become flag hand-offs inside the launch, and -- the point of the exercise -- a role
// issues the NEXT phase's weight / KV / panel loads BEFORE it starts waiting, so the weight stream
// runs during the wait instead of after it.
//
// Hand-off protocol (cdna_hip_programming.md Guideline 16, write-through form):
//   producer: payload with agent-scope relaxed atomic stores (= global_store ... sc1, write-through,
//             4 B per lane is the epilogue's natural width) -> every wave `s_waitcnt vmcnt(0)` ->
//             __syncthreads() -> ONE lane: relaxed agent fetch_add on the phase counter.
//   consumer: ONE lane polls the counter relaxed (s_sleep between polls, bounded) -> ONE agent-scope
//             acquire fence (drops this CU's stale L1 lines) -> __syncthreads() -> plain loads.
//             (the payload itself is then read with sc0 sc1 loads, which cannot hit a stale line)
//   counters are zeroed by the token kernel that precedes this launch in every step (a captured
//   hipMemsetAsync node was observed to race with the first replay after eager work: consumers
//   saw the previous step's final counts, skipped the wait and read stale partials -- wrong tokens,
//   no timeout); results never depend on placement or timing; a spin that exceeds its bound raises
//   `err` and falls through (no GPU hang).
//
// MEASURED (MI355X, bs=1, bf16): 0.39 ms/token vs 0.37 for the per-layer graph.  A hand-off
// (drain write-through stores, counter, poll, acquire, fresh cross-XCD read) costs what a kernel
// boundary + cold start costs, so the persistent form buys nothing here; it stays as a tested
// alternative (Text2SemanticDecoder.use_megastep) and as the base for fewer-exchange designs.
// Buffers are single: ffn(l) reads y(l) strictly before any attn block can write y(l+1) (that
// write is behind the z(l) hand-off, which every ffn block signs only after reading y(l)), and
// symmetrically for z.
#pragma once
#include "t2s_decode.h"

namespace gsv {

template <typename WT>
struct MegaLayer {
    const WT *wqkv, *wo, *w1, *w2p;
    const float *bqkv, *bo, *b1, *b2, *ln1g, *ln1b, *ln2g, *ln2b;
};

template <typename WT>
struct MegaArgs {
    const MegaLayer<WT>* layers;  // device array [n_layer]
    int n_layer;
    const float* xin;             // [B][512] input of layer 0 (token kernel output)
    float *xbuf, *x1buf;          // [B][512]
    float *ypart, *zpart;         // [B][16][512], [B][32][512]
    WT *kc, *vc;                  // [n_layer][B][16][T][32]
    size_t layer_elems;
    int T;
    const int64_t* kv_len;
    unsigned* cnt;                // [B][2*n_layer]  y counters [0,n_layer), z counters [n_layer, 2n)
    unsigned* err;
};

constexpr int kMegaRoles = kH + kNJ;  // 48 blocks per sequence

__device__ __forceinline__ void wt_store(float* p, float v) {  // write-through (sc1) 4-byte store
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void mega_publish(unsigned* c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void mega_wait(unsigned* c, unsigned target, unsigned* err) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22)) {  // ~1 s: a lost producer must not hang the GPU
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// panel GEMV with write-through stores (the partial vector is the hand-off payload)
template <typename WT, int K>
__device__ __forceinline__ void panel_finish_wt(Panel<WT, K>& p, const float* __restrict__ vec_lds, float* out) {
    constexpr int EPL = Geo<WT>::EPL, LPR = Panel<WT, K>::LPR, RPI = Panel<WT, K>::RPI, NIT = Panel<WT, K>::NIT;
    const int tid = threadIdx.x;
    const int part = tid % LPR, rsub = tid / LPR;
    float vr[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) vr[i] = vec_lds[part * EPL + i];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        float wv[EPL];
        Unpack<WT, EPL>::run(p.w[it], wv);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < EPL; ++i) s = fmaf(wv[i], vr[i], s);
        s = group_sum<LPR>(s);
        if (part == 0) wt_store(out + rsub + it * RPI, s);
    }
}

template <typename WT>
__global__ __launch_bounds__(kNT) void t2s_megastep_kernel(MegaArgs<WT> a) {
    __shared__ __attribute__((aligned(16))) float smem[kAttnLdsFloats];
    const int role = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int EPL = Geo<WT>::EPL;
    constexpr int CPR = Geo<WT>::CPR;
    const bool owner = tid < kD;
    unsigned* cnt_y = a.cnt + (size_t)b * 2 * a.n_layer;
    unsigned* cnt_z = cnt_y + a.n_layer;

    if (role < kH) {
        // =============================== attention role: head h ===============================
        const int h = role;
        float* xs = smem;
        float* qkv = xs + kD;
        float* att = qkv + 96;
        float* red = att + 32;
        float* pacc = red + 2 * kNW;
        float* pm = pacc + kNW * 32;
        float* pl = pm + kNW;
        float* stage = pl + kNW;
        constexpr int LPR = kDh / EPL;
        constexpr int RPI = kNT / LPR;
        constexpr int KCH = 2;
        constexpr int RW = 96 / kNW;
        int n = (int)a.kv_len[b];
        if (n > a.T - 1) n = a.T - 1;
        if (n < 0) n = 0;
        const int part = tid % LPR, rsub = tid / LPR;
        const int oi = sumN_index<8>();
        for (int l = 0; l < a.n_layer; ++l) {
            const MegaLayer<WT> L = a.layers[l];
            WT* Kp = a.kc + (size_t)l * a.layer_elems + (((size_t)b * kH + h) * a.T) * kDh;
            WT* Vp = a.vc + (size_t)l * a.layer_elems + (((size_t)b * kH + h) * a.T) * kDh;
            // ---- this layer's weight / KV / panel stream starts BEFORE the wait
            const WT* wp = L.wqkv + ((size_t)h * 96 + wid * RW) * kD;
            raw16 wq[RW][CPR];
#pragma unroll
            for (int r = 0; r < RW; ++r) row_load<WT>(wp + (size_t)r * kD, wq[r]);
            raw16 kreg[KCH], vreg[KCH];
#pragma unroll
            for (int it = 0; it < KCH; ++it) kreg[it] = ldg16(Kp + (size_t)min(rsub + it * RPI, n) * kDh + part * EPL);
#pragma unroll
            for (int it = 0; it < KCH; ++it) vreg[it] = ldg16(Vp + (size_t)min(rsub + it * RPI, n) * kDh + part * EPL);
            Panel<WT, kDh> po;
            po.issue(L.wo + (size_t)h * kD * kDh);
            const float bq = L.bqkv[h * 96 + wid * RW + min(oi, RW - 1)];

            // ---- layer input
            float v;
            if (l == 0) {
                v = owner ? a.xin[(size_t)b * kD + tid] : 0.f;
            } else {
                mega_wait(cnt_z + (l - 1), kNJ, a.err);
                const MegaLayer<WT> P = a.layers[l - 1];
                PartialSum<kNJ> ps;
                ps.issue_coherent(a.zpart + (size_t)b * kNJ * kD, P.b2, a.x1buf + (size_t)b * kD, P.ln2g, P.ln2b, sizeof(float) * kNJ * kD);
                ps.park(stage);
                __syncthreads();
                v = ln512(owner ? ps.finish(stage) : 0.f, owner, ps.lng, ps.lnb, red);
            }
            if (owner) {
                xs[tid] = v;
                if (h == 0) wt_store(a.xbuf + (size_t)b * kD + tid, v);
            }
            __syncthreads();

            // ---- q, k, v
            {
                float xr[8];
                lane_x<WT>(xs, xr);
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
                        if (row < 64) Kp[(size_t)n * kDh + row - 32] = s; else Vp[(size_t)n * kDh + row - 64] = s;
                    }
                    qkv[row] = val;
                }
            }
            __syncthreads();

            // ---- single-pass attention over [0, n]
            const float scale = 0.17677669529663687f;
            float qr[EPL];
#pragma unroll
            for (int i = 0; i < EPL; ++i) qr[i] = qkv[part * EPL + i];
            float m_run = -INFINITY, l_run = 0.f, acc[EPL];
#pragma unroll
            for (int i = 0; i < EPL; ++i) acc[i] = 0.f;
            for (int c0 = 0; c0 == 0 || c0 < n; c0 += KCH * RPI) {
                if (c0 > 0) {
#pragma unroll
                    for (int it = 0; it < KCH; ++it) {
                        kreg[it] = ldg16(Kp + (size_t)min(c0 + rsub + it * RPI, n) * kDh + part * EPL);
                        vreg[it] = ldg16(Vp + (size_t)min(c0 + rsub + it * RPI, n) * kDh + part * EPL);
                    }
                }
                float sv[KCH + 1];
                float cmax = -INFINITY;
#pragma unroll
                for (int it = 0; it < KCH; ++it) {
                    const int r = c0 + rsub + it * RPI;
                    float kk[EPL];
                    Unpack<WT, EPL>::run(kreg[it], kk);
                    float s = 0.f;
#pragma unroll
                    for (int i = 0; i < EPL; ++i) s = fmaf(qr[i], kk[i], s);
                    s = group_sum<LPR>(s);
                    sv[it] = r < n ? s * scale : -INFINITY;
                    cmax = fmaxf(cmax, sv[it]);
                }
                {
                    float s = 0.f;
#pragma unroll
                    for (int i = 0; i < EPL; ++i) s = fmaf(qr[i], qkv[32 + part * EPL + i], s);
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
                    Unpack<WT, EPL>::run(vreg[it], vv);
                    if (part == 0) l_run += p;
#pragma unroll
                    for (int i = 0; i < EPL; ++i) acc[i] = fmaf(p, live ? vv[i] : 0.f, acc[i]);
                }
                {
                    const float p = expf(sv[KCH] - mref);
                    if (part == 0) l_run += p;
#pragma unroll
                    for (int i = 0; i < EPL; ++i) acc[i] = fmaf(p, qkv[64 + part * EPL + i], acc[i]);
                }
                m_run = m_new;
            }
#pragma unroll
            for (int m = 32; m >= LPR; m >>= 1) {
#pragma unroll
                for (int i = 0; i < EPL; ++i) acc[i] += __shfl_xor(acc[i], m, 64);
            }
            l_run = wave_sum(l_run);
            if (lane < LPR) {
#pragma unroll
                for (int i = 0; i < EPL; ++i) pacc[wid * 32 + part * EPL + i] = acc[i];
            }
            if (lane == 0) { pm[wid] = m_run; pl[wid] = l_run; }
            __syncthreads();
            if (wid == 0) {
                const float mw = pm[lane & 15], lw = pl[lane & 15];
                float M = mw;
#pragma unroll
                for (int m = 8; m >= 1; m >>= 1) M = fmaxf(M, __shfl_xor(M, m, 64));
                const float f = expf(mw - M);
                float den = lw * f;
#pragma unroll
                for (int m = 8; m >= 1; m >>= 1) den += __shfl_xor(den, m, 64);
                const int hf = lane >> 5, d = lane & 31;
                float num = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) num = fmaf(pacc[(hf * 8 + w) * 32 + d], __shfl(f, hf * 8 + w, 64), num);
                num += __shfl_xor(num, 32, 64);
                if (lane < 32) att[d] = num / den;
            }
            __syncthreads();
            panel_finish_wt<WT, kDh>(po, att, a.ypart + ((size_t)b * kH + h) * kD);
            mega_publish(cnt_y + l);
        }
    } else {
        // =============================== ffn role: slice j ====================================
        const int j = role - kH;
        float* xs = smem;
        float* hb = xs + kD;
        float* red = hb + kFJ;
        float* stage = red + 2 * kNW;
        constexpr int RW = kFJ / kNW;
        const int oi = sumN_index<RW>();
        for (int l = 0; l < a.n_layer; ++l) {
            const MegaLayer<WT> L = a.layers[l];
            const int row0 = j * kFJ + wid * RW;
            raw16 w1r[RW][CPR];
#pragma unroll
            for (int r = 0; r < RW; ++r) row_load<WT>(L.w1 + (size_t)(row0 + r) * kD, w1r[r]);
            Panel<WT, kFJ> p2;
            p2.issue(L.w2p + (size_t)j * kD * kFJ);
            const float b1r = L.b1[row0 + oi];

            mega_wait(cnt_y + l, kH, a.err);
            PartialSum<kH> ps;
            ps.issue_coherent(a.ypart + (size_t)b * kH * kD, L.bo, a.xbuf + (size_t)b * kD, L.ln1g, L.ln1b, sizeof(float) * kH * kD);
            ps.park(stage);
            __syncthreads();
            const float v = ln512(owner ? ps.finish(stage) : 0.f, owner, ps.lng, ps.lnb, red);
            if (owner) {
                xs[tid] = v;
                if (j == 0) wt_store(a.x1buf + (size_t)b * kD + tid, v);
            }
            __syncthreads();
            {
                float xr[8];
                lane_x<WT>(xs, xr);
                float acc[RW];
#pragma unroll
                for (int u = 0; u < RW; ++u) acc[u] = row_dot<WT>(w1r[u], xr);
                const float tot = wave_sumN<RW>(acc);
                if ((lane & 15) == 0) hb[wid * RW + oi] = fmaxf(tot + b1r, 0.f);
            }
            __syncthreads();
            panel_finish_wt<WT, kFJ>(p2, hb, a.zpart + ((size_t)b * kNJ + j) * kD);
            mega_publish(cnt_z + l);
        }
    }
}

}  // namespace gsv
