/*
 * gsv_tts_hip.h -- C ABI of the MI355X (gfx950) GPT-SoVITS inference hot path.
 *
 * The reference (chinokikiss/GSV-TTS-Lite) has no FFI: its hot path is PyTorch modules
 * driven from Python.  This header is the boundary a maintainer would bind underneath those
 * modules (ctypes stub in INTEGRATION.md).  Every entry point names the reference interface
 * it replaces (file:line in the reference tree).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types; returns GSV_OK (0) or an
 *     error code, message via gsv_last_error() (thread-local).  No exceptions cross the ABI.
 *   - All tensor pointers are DEVICE pointers unless a parameter says "host".
 *   - `stream` is a hipStream_t passed as void* (torch: torch.cuda.current_stream().cuda_stream).
 *   - Ownership mirrors the reference: the caller owns KV caches, kv_len and
 *     every I/O buffer (torch allocations made once in initialize_runtime); the library owns
 *     only its repacked weight arena and fixed scratch, created at load time.  Nothing is
 *     allocated or freed inside a step, so steps are hipGraph-capturable.
 *   - One caller per handle at a time (the reference serialises with TTS._infer_lock,
 *     gsv_tts/TTS.py:145); distinct handles (one per GPU/process) are independent.
 */
#ifndef GSV_TTS_HIP_H
#define GSV_TTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSV_OK 0
#define GSV_ERR_ARG 1     /* bad argument / unknown tensor name / shape mismatch */
#define GSV_ERR_HIP 2     /* a HIP runtime call failed */
#define GSV_ERR_STATE 3   /* call order violated (e.g. step before finalize/bind) */

#define GSV_F32 0         /* fp32 weights/KV/activations: the bit-exact-token parity mode */
#define GSV_BF16 1        /* bf16 weights + KV (+ bf16 vocoder activations), fp32 accumulate */
#define GSV_FP8 2         /* gsv_t2s only: GSV_BF16 plus OCP e4m3 QKV / FFN weights (per-output-channel scale) and e4m3
                             activations on the fp8 MFMA in the batched decode step; K/V cache, out-proj, prefill: bf16 */

int gsv_version(void);
const char* gsv_last_error(void);

/* ------------------------------------------------------------------------------------------
 * GPT semantic-token decoder  (reference: gsv_tts/GPT_SoVITS/GPT/t2s_model.py)
 * ---------------------------------------------------------------------------------------- */
typedef struct gsv_t2s gsv_t2s;

typedef struct {
    int n_layer, hidden, n_head, vocab, eos; /* config["model"], t2s_model.py:159-168 */
    int n_pos;                               /* rows of the sinusoidal tables (4000, t2s_model.py:212) */
    int n_phoneme;                           /* phoneme_vocab_size */
    int dtype;                               /* GSV_F32 | GSV_BF16 | GSV_FP8 */
} gsv_t2s_config;

/* replaces Text2SemanticDecoder.__init__ + Loader.get_gpt_weights (gsv_tts/Loader.py:111-170) */
int gsv_t2s_create(const gsv_t2s_config* cfg, gsv_t2s** out);
int gsv_t2s_destroy(gsv_t2s* h);

/* Hand one fp32 device tensor to the library under its reference state-dict name (after the
 * Loader.py:130-154 remap), e.g. "t2s_transformer.blocks.3.qkv.weight", "ar_predict_layer.weight",
 * "ar_audio_embedding.word_embeddings.weight", "bert_proj.weight".  Two synthetic names carry
 * the host-precomputed tables alpha*pe of embedding.py:58-69: "ar_text_position.pe_scaled",
 * "ar_audio_position.pe_scaled" ([n_pos][hidden]).  The data is converted/repacked into the
 * library's arena during the call (stream-ordered); the source may be freed afterwards. */
int gsv_t2s_load_tensor(gsv_t2s* h, const char* name, const float* data, int64_t numel, void* stream);
/* all tensors present?  builds decode panels; required before any step */
int gsv_t2s_finalize(gsv_t2s* h, void* stream);

/* Caller-owned runtime state for ONE batch size: the reference's Bucket family for that batch
 * size (t2s_model.py:146-156, 240-276).  All bucket lengths of a batch size alias one storage
 * with the largest T as stride, so `max_kv` is that largest T ("nested" KV cache). */
typedef struct {
    int batch;              /* B */
    int max_kv;             /* T: positions per sequence (cache stride) */
    void* k_cache;          /* [n_layer][B][n_head][T][head_dim], cfg.dtype */
    void* v_cache;          /* same */
    int64_t* kv_len;        /* [B]   Bucket.kv_cache_len */
    int64_t* x_len;         /* [B]   text length per slot (PE offset, t2s_model.py:456,728) */
    int64_t* pre_tokens;    /* [B][T+1] sampled token at kv position (t2s_model.py:605,653); column T
                               holds the sample taken when the cache is exactly full */
    uint8_t* seen;          /* [B][vocab] repetition-penalty membership (prompt + generated) */
    int32_t* step;          /* [B]   logits launches since the slot was (re)filled */
    int32_t* eos_at;        /* [B]   first step whose sample was EOS, else -1 */
    float* logits;          /* [B][vocab] last logits after suppression/penalty (host sampling) */
    float* hidden;          /* [B][hidden] last final hidden state (Bucket.graph_xy_dec) */
    int64_t* tok_override;  /* [B]   host-sampled tokens, consumed when ctl[0] == 1; with ctl[0] == 2 (device sampling) a value
                               v > 0 makes v - 1 the sequence's noise stream instead of its slot index (key it by request
                               and a request's samples do not depend on slot, refill order or rank) */
    int32_t* ctl;           /* [8]   {sample_mode, suppress_steps, rep_enabled, -, top_k, seed_lo, seed_hi,
                               suppress_first}; suppress_first != 0: the prefill's sample never takes 280 / 486 / EOS
                               (infer / infer_stream, t2s_model.py:415-416), independent of suppress_steps; sample_mode 0 = greedy argmax, 1 = tok_override (host sampling),
                               2 = device sampling: temperature fctl[1], top-k ctl[4] (<= 0: off; ties with the
                               k-th value are kept, GPT/utils.py:45-48), then argmax(softmax / Exp(1))
                               (utils.py:56-59) with a counter-based noise stream keyed by
                               (seed, slot, kv position, step, token id); top-p fctl[2] in (0,1) is applied first, on the
                               un-tempered logits, as the set {p >= tau} (ties with tau stay together) */
    float* fctl;            /* [4]   {repetition_penalty, temperature, top_p, -} */
} gsv_t2s_state;
int gsv_t2s_bind_state(gsv_t2s* h, const gsv_t2s_state* st);
/* Forgets the state bound for `batch`: its captured steps are destroyed and its staging goes back to the handle, so the caller may
 * free the tensors the state pointed at (the reference rebuilds its runtime the same way: initialize_runtime, t2s_model.py:210-298,
 * drops the old buckets).  The caller makes sure nothing of that state is still running.  Unknown batch: GSV_OK. */
int gsv_t2s_unbind_state(gsv_t2s* h, int batch);
/* Optional: `host_mapped` [batch] int32 in host memory the device can write (hipHostMalloc / a pinned torch tensor), or
 * NULL to turn it off.  Every kernel that sets state.eos_at[slot] then also publishes the value there (system-scope store),
 * so the host loop of t2s_model.py:451-453 reads the EOS flag from its own memory after an event instead of enqueuing a
 * device-to-host copy between the decode windows.  Call after gsv_t2s_bind_state (which clears it); it invalidates the
 * captured steps of this batch size. */
int gsv_t2s_set_eos_mirror(gsv_t2s* h, int batch, int32_t* host_mapped);

/* replaces process_single_data / process_batch_data (t2s_model.py:300-383): builds packed rows
 * [x_b | y_b | 0-pad] = text-emb + bert_proj + alpha_t*pe, audio-emb + alpha_a*pe.
 *   x_ids [nrows][lx_max], y_ids [nrows][ly_max], bert [nrows][lx_max][1024] (row-padded),
 *   x_lens/y_lens [nrows] -> xy [nrows][l_max][hidden] fp32.  scratch: [nrows*lx_max][hidden] f32 */
int gsv_t2s_embed_prompt(gsv_t2s* h, int nrows, int lx_max, int ly_max, int l_max, const int64_t* x_ids,
                         const int64_t* y_ids, const float* bert, const int64_t* x_lens,
                         const int64_t* y_lens, float* xy, float* scratch, void* stream);

/* replaces T2STransformer.process_prompt (t2s_model.py:31-65,114-127) + ar_predict_layer +
 * the first sample (t2s_model.py:414-420, 608-616).  xy [nrows][l_max][hidden] is consumed
 * (overwritten with the hidden states).  Attention mask is the reference's prompt mask,
 * implied by (x_len, y_len) per row: text rows see all text, audio rows see all text + causal
 * audio; rows/cols >= x_len+y_len are padding.  K/V for positions [0, x_len+y_len) go to cache
 * rows slot0..slot0+nrows-1 of the state bound for `batch`; afterwards for those slots:
 * kv_len = x_len + y_len, x_len set, step = 0, eos_at = -1, and the first token is pending
 * (sampled from logits[:, :-1], i.e. EOS impossible; suppression per ctl).
 * workspace: gsv_t2s_prefill_workspace(h, nrows, l_max) bytes. */
size_t gsv_t2s_prefill_workspace(gsv_t2s* h, int nrows, int l_max);
int gsv_t2s_prefill(gsv_t2s* h, int batch, int slot0, int nrows, int l_max, float* xy, const int64_t* x_lens,
                    const int64_t* y_lens, void* workspace, size_t workspace_bytes, void* stream);
/* The same for rows that go to scattered slots (the continuous-batching refill of every sequence that finished in a
 * check window, t2s_model.py:696-722, as ONE packed prefill): slots int32 [nrows] on the device, distinct, each
 * < batch; row r fills slot slots[r]. */
int gsv_t2s_prefill_slots(gsv_t2s* h, int batch, const int32_t* slots, int nrows, int l_max, float* xy, const int64_t* x_lens,
                          const int64_t* y_lens, void* workspace, size_t workspace_bytes, void* stream);

/* The refill as an ASYNCHRONOUS pair, for continuous batching that does not stall its decode steps on a prompt pass:
 *   gsv_t2s_prefill_slots_staged  the same prompt pass, callable on ANOTHER stream than the one the decode step
 *       replays on.  It writes the K/V rows [0, x_len + y_len) of the listed slots and puts everything else the
 *       step also writes (kv_len, x_len, step, eos_at, the first logits / hidden / pending token) into the library's
 *       staging of that batch size.  Contract: before calling, PARK each listed slot by setting kv_len[slot] = -1
 *       on the step's stream and treat it as idle: the decode step keeps a parked slot parked, sends its K/V row to the
 *       last row of the cache (which no prompt of <= max_kv - 1 positions uses), lets it attend over row 0 only and
 *       leaves its `seen` set alone.
 *   gsv_t2s_commit_slots  on the step's stream, once the staged pass has completed (event): staging -> live state of
 *       the listed slots; the next step decodes them.  Rows are independent, so tokens per request are unchanged. */
int gsv_t2s_prefill_slots_staged(gsv_t2s* h, int batch, const int32_t* slots, int nrows, int l_max, float* xy,
                                 const int64_t* x_lens, const int64_t* y_lens, void* workspace, size_t workspace_bytes,
                                 void* stream);
int gsv_t2s_commit_slots(gsv_t2s* h, int batch, const int32_t* slots, int nrows, void* stream);

/* Prompt passes AHEAD of the slots that will decode them (continuous batching whose refills cost no idle slot-steps; the
 * reference prefills a request when a slot has finished, t2s_model.py:696-722, and every slot waits for it).  Bind a SECOND
 * state of another batch size (`batch_src`, with a KV cache of its own, never stepped) and run the prompt passes of the
 * NEXT requests into it with gsv_t2s_prefill_slots_staged on a side stream, several requests per pass, while the steps of
 * `batch_dst` run.  When a slot of `batch_dst` has finished, this call -- on the step's stream, after the pass's completion
 * event -- copies K/V rows [0, kv_len) of source slot slots_src[r] into slot slots_dst[r] and moves the staged kv_len,
 * x_len, step, eos_at, first logits / hidden / pending token into the live state of slots_dst[r]; the next step decodes it.
 * slots_dst / slots_src / tok_override are HOST arrays [nrows] (they ride in the kernel arguments); tok_override (may be
 * NULL) sets state.tok_override[slots_dst[r]] (device sampling: the request's noise stream).  A prompt pass is
 * row-independent and packing-invariant, so a request's tokens do not depend on when or beside what it was prefilled. */
int gsv_t2s_adopt_slots(gsv_t2s* h, int batch_dst, const int32_t* slots_dst, int batch_src, const int32_t* slots_src,
                        const int64_t* tok_override, int nrows, void* stream);

/* Tail compaction of the slot loop (the reference lets finished slots "keep decoding garbage" once its queue is empty,
 * t2s_model.py:684-694: the last requests then pay the step of the full batch size).  Moves LIVE slots of one stepped state into
 * slots of ANOTHER bound state with a KV cache of its own (max_kv >= the source's) between two steps, on the step's stream:
 * K/V rows [0, kv_len), kv_len, x_len, the token history (pre_tokens), the repetition-penalty set (seen), step, eos_at,
 * tok_override and what the next step reads of the previous one (logits, hidden, pending token).  The next gsv_t2s_decode of
 * `batch_dst` continues every moved request where `batch_src` left it.  slots_dst / slots_src are HOST arrays [nrows <= 64];
 * a slot may be listed once per side.  Slots of `batch_dst` that are not listed keep their state (park them with kv_len = -1). */
int gsv_t2s_move_slots(gsv_t2s* h, int batch_dst, const int32_t* slots_dst, int batch_src, const int32_t* slots_src, int nrows,
                       void* stream);

/* replaces T2STransformer.decode_next_token (t2s_model.py:67-105,129-143) for an EXPLICIT input
 * x [B][hidden] (parity seam): appends K/V at kv_len[b], attends to [0, kv_len[b]], writes the
 * final hidden state to state.hidden and bumps kv_len.  No sampling.  Takes the path gsv_t2s_decode would take for
 * this batch size (per-sequence kernels, or the batched chain from gsv_t2s_batched_min sequences on). */
int gsv_t2s_decode_hidden(gsv_t2s* h, int batch, const float* x, void* stream);

/* The AR hot loop body (t2s_model.py:430-456 / 637-653, 727-728) `n_steps` times, all on device:
 * take the pending token (greedy argmax of the penalised logits, or tok_override), record it in
 * pre_tokens[b][kv_len[b]], build emb + alpha*pe[kv_len - x_len], run the layers, bump kv_len,
 * compute the next logits (suppression while step < ctl[1]; repetition penalty over `seen`).
 * `use_graph` is a bit set: GSV_STEP_GRAPH replays the step from a hipGraph captured on first use (one per batch size and
 * flag combination); GSV_STEP_FUSED_TOKEN is the caller's promise that ctl[0] is 0 or 1 (greedy or tok_override, i.e. no
 * device sampling) for these steps -- on the two-launches-per-layer path (below the batched chain's size) the first layer's
 * attention kernel then does the token kernel's work itself (one launch less per step; same tokens, same state).
 * From a tuned batch size on (bf16 / fp8 handles) the step is the batched chain of csrc/t2s_batch.h: weights
 * streamed once per step through MFMA GEMMs instead of once per sequence. */
#define GSV_STEP_GRAPH 1
#define GSV_STEP_FUSED_TOKEN 2
int gsv_t2s_decode(gsv_t2s* h, int batch, int n_steps, int use_graph, void* stream);
/* Batch size from which gsv_t2s_decode runs the batched chain (INT_MAX on fp32 handles: never).  Tests mirror the
 * choice in the oracle, whose reduced-precision modes round the operands each path rounds. */
int gsv_t2s_batched_min(gsv_t2s* h);
/* FFN slices per sequence of the two-launches-per-layer step at this batch size: 32 slices of 64 hidden units, or -- at
 * <= 4 sequences -- 64 slices of 32 (fp32 handles too: their partials stay fp32).  Each slice's partial 512-vector crosses the kernel boundary rounded to half on
 * bf16 handles, so the count is part of the arithmetic; the bf16-mode oracle sums the same slices (oracle.py). */
int gsv_t2s_ffn_slices(gsv_t2s* h, int batch);
/* Device memory the handle owns, in bytes (its arena's blocks: repacked weights, fragments, scratch, the staging of every
 * bound state).  Pieces given back inside the handle -- a state re-bound with gsv_t2s_bind_state, a tensor re-loaded with
 * gsv_t2s_load_tensor, scratch that grew -- are handed out again by size, so re-binding the same shapes or hot-swapping
 * weights of the same architecture for a handle's whole life leaves this number where it was; all of it is released by
 * gsv_t2s_destroy. */
size_t gsv_t2s_device_bytes(gsv_t2s* h);
/* Measurement aid (bench.py `roofline`): average time in ms of ONE launch of each per-sequence decode-step
 * kernel class {attn, ffn, logits, token}: the class's launches over all layers (every layer streams its own
 * weights, as in a real step) are captured into a hipGraph and replayed `iters` times between two hipEvents on
 * `stream` -- no host launch cost, but the dependent-launch gap every kernel of a real step pays is included.
 * out_ms: host float[4].  kv_len is left untouched; the sweeps rewrite the K/V row AT kv_len, the pending
 * token's pre_tokens / seen / eos_at entries and the partial-sum scratch, all of which the next real step
 * (or prefill) overwrites -- call it between utterances, not inside one. */
int gsv_t2s_time_kernels(gsv_t2s* h, int batch, int iters, float* out_ms, void* stream);
/* Bring-up aid: when `buf` (device, >= 32 x uint64) is non-null the LAST layer's attn/ffn kernels
 * write shader-clock timestamps of their phases into it (slots 0-6 attn, 8-15 ffn: the block's first wave; + 16: its last). */
int gsv_t2s_set_debug(gsv_t2s* h, void* buf);
/* materialise the pending token of every slot into pre_tokens/seen/eos_at (idempotent) */
int gsv_t2s_flush(gsv_t2s* h, int batch, void* stream);

/* ------------------------------------------------------------------------------------------
 * SoVITS flow + Generator  (reference: gsv_tts/GPT_SoVITS/SoVITS/models.py, module/modules.py)
 * ---------------------------------------------------------------------------------------- */
typedef struct gsv_voc gsv_voc;

typedef struct {
    int inter_channels, hidden_channels, gin_channels, upsample_initial_channel;
    int n_upsample;
    int upsample_rates[8], upsample_kernel_sizes[8];
    int n_resblock_kernels;
    int resblock_kernel_sizes[4];
    int resblock_dilations[4];   /* (1,3,5): same for every resblock, modules.py:116 */
    int n_flows;                 /* 4 coupling layers, models.py:31 */
    int dtype;                   /* GSV_F32 | GSV_BF16 */
} gsv_voc_config;

/* replaces SynthesizerTrn.{flow,dec} construction + Loader.get_sovits_weights (Loader.py:59-103) */
int gsv_voc_create(const gsv_voc_config* cfg, gsv_voc** out);
int gsv_voc_destroy(gsv_voc* h);
/* fp32 device tensor under its state-dict name: "dec.*" (weight-norm removed, Loader.py:95),
 * "flow.flows.{0,2,4,6}.*" (weight_g/weight_v still separate; folded by finalize) and, optionally,
 * "enc_p.*" + "quantizer.vq.layers.0._codebook.embed" (enables gsv_voc_enc_p on bf16 handles) */
int gsv_voc_load_tensor(gsv_voc* h, const char* name, const float* data, int64_t numel, void* stream);
int gsv_voc_finalize(gsv_voc* h, void* stream);

/* replaces SynthesizerTrn.flow_dec (models.py:380-383): o = dec(flow(z_p, mask, ge) * mask, g=ge).
 *   z_p [inter][T] fp32 channels-first (torch layout), y_mask [T], ge [gin][Tg] with Tg in {1, T}
 *   (Tg == T: per-frame speaker embedding of the time-concatenated batch, TTS.py:740-744)
 *   out [T * prod(upsample_rates)] fp32.  workspace from gsv_voc_workspace(h, T). */
size_t gsv_voc_workspace(gsv_voc* h, int T);
int gsv_voc_flow_dec(gsv_voc* h, const float* z_p, const float* y_mask, const float* ge, int T, int Tg,
                     float* out, void* workspace, size_t workspace_bytes, void* stream);
/* The same pass replayed from a hipGraph captured on first use for this (z_p, y_mask, ge, out, workspace, T, Tg) --
 * the reference's per-bucket CUDA graphs of SynthesizerTrn.initialize_runtime / decode (models.py:322-369, 406-423): the
 * caller owns STATIC buffers per bucket length, copies the chunk in (zero mask beyond the real length), replays, slices.
 * A 50-frame streaming chunk is ~60 launch-bound kernels; replay removes their host launch cost. */
int gsv_voc_flow_dec_graph(gsv_voc* h, const float* z_p, const float* y_mask, const float* ge, int T, int Tg,
                           float* out, void* workspace, size_t workspace_bytes, void* stream);
/* x fp32 [C][T_in] -> y fp32 [C][T_out], linear resampling as F.interpolate(mode="linear", align_corners=False): the
 * `speed != 1` resampling of TextEncoder.infer (models.py:217-219), applied to the projected statistics (proj is 1x1
 * affine, so resampling its output equals resampling its input). */
int gsv_voc_resample_linear(const float* x, int C, int T_in, float* y, int T_out, void* stream);
/* parity seams: the flow alone (ResidualCouplingBlock.forward reverse, models.py:58-65) and the
 * Generator alone (models.py:113-132); same layouts. */
int gsv_voc_flow(gsv_voc* h, const float* z_p, const float* y_mask, const float* ge, int T, int Tg,
                 float* z_out, void* workspace, size_t workspace_bytes, void* stream);
int gsv_voc_dec(gsv_voc* h, const float* z, const float* ge, int T, int Tg, float* out, void* workspace,
                size_t workspace_bytes, void* stream);

/* enc_p on device (bf16 handles that were given the "enc_p.*" and "quantizer.vq.layers.0._codebook.embed"
 * tensors): replaces quantizer.decode + the x2 nearest upsampling + TextEncoder.infer for speed == 1,
 * non-streaming calls (SoVITS/models.py:196-224, 387-400; attentions.py:58-220; mrte_model.py:20-38).
 *   codes int64 [n_codes], text int64 [n_text], ge512 fp32 channels-LAST [Tg][512] (ge_to512(ge) for v2Pro /
 *   v2ProPlus, ge itself for v2), Tg in {1, 2*n_codes}; slice_indices int64 [2*n_codes][2] or NULL
 *   (per-frame phoneme range of the time-concatenated batch, mrte_model.py:27-33)
 *   -> m_p, logs_p fp32 [inter][T] channels-first, T = 2*n_codes; attn fp32 [4][T][n_text] or NULL
 *   (enc_p.mrte.cross_attention.attn, read by TTS for subtitles).  y_mask is all ones (batch of one). */
int gsv_voc_has_enc_p(gsv_voc* h);
size_t gsv_voc_enc_workspace(gsv_voc* h, int n_codes, int n_text);
int gsv_voc_enc_p(gsv_voc* h, const int64_t* codes, int n_codes, const int64_t* text, int n_text, const float* ge512, int Tg,
                  const int64_t* slice_indices, float* m_p, float* logs_p, float* attn, void* workspace, size_t workspace_bytes,
                  void* stream);

/* SynthesizerTrn.decode (SoVITS/models.py:385-429) as ONE call -- nothing of it is left to the caller's tensor library:
 *   codes int64 [n_codes] (n_q = 1, batch 1), text int64 [n_text];
 *   ge fp32 [gin][Tg] channels-first, Tg = 1 (one speaker) or n_codes (per-TOKEN columns of a time-concatenated batch: the
 *     x2 nearest upsampling of models.py:389 and the nearest resize of :402 are index maps inside);
 *   ge_to512 (v2Pro / v2ProPlus, :394), quantizer lookup + x2 upsampling + TextEncoder.infer (enc_p, :395-400; slice_indices as
 *     for gsv_voc_enc_p), streaming slice + cross-fade (valid_start, overlap_len > 0, overlap_state fp32 [2*inter][overlap_len]
 *     in/out, has_overlap = 0 on a stream's first chunk; module/models.py:209-215), speed resampling to `out_frames` frames
 *     (:217-219: the caller evaluates int(T' / speed) + 1 ONCE, in the arithmetic it sizes `out` with -- Python doubles in the
 *     reference -- and passes the result; out_frames == T' means speed 1, no resampling), z_p = m_p + N(0,1) * exp(logs_p) * noise_scale (:404), flow + Generator (:380-383).
 *   The noise is counter-based (lowbias32 of the element index and `seed`, Box-Muller): a call is replayable from its seed; it is
 *   not torch's generator stream.  use_graph != 0 replays flow + Generator from a hipGraph captured for this workspace
 *   (keep one workspace per chunk length: the reference's per-bucket CUDA graphs, models.py:322-369); a handle keeps at most
 *   GSV_VOC_MAX_GRAPHS captured passes and evicts the least recently replayed one.
 *   -> out fp32 [out_frames * prod(upsample_rates)], T' = 2 n_codes - valid_start;
 *      attn fp32 [4][2 n_codes][n_text] or NULL.  workspace: gsv_voc_decode_workspace(...) bytes, device. */
#define GSV_VOC_MAX_GRAPHS 64
size_t gsv_voc_decode_workspace(gsv_voc* h, int n_codes, int n_text, int Tg, int out_frames, int valid_start);
int gsv_voc_decode(gsv_voc* h, const int64_t* codes, int n_codes, const int64_t* text, int n_text, const float* ge, int Tg,
                   const int64_t* slice_indices, float noise_scale, unsigned long long seed, int out_frames, int valid_start, int overlap_len,
                   float* overlap_state, int has_overlap, int use_graph, float* out, float* attn, void* workspace, size_t workspace_bytes,
                   void* stream);

/* Subtitle alignment: monotonic Viterbi path of vocoder frames over phonemes -- replaces
 * TTS._viterbi_monotonic (gsv_tts/TTS.py:1744-1797), which TTS.infer / infer_stream / infer_batched call on
 * the `attn` returned by vq_model.decode (TTS.py:250, 445, 769).
 *   attn fp32 [H][T][N] (device; H <= 8 heads, T frames, 2 <= N <= 4096 phonemes)
 *   -> assign int32 [T] (device): phoneme per frame, -1 before the first frame whose head-averaged attention
 *   peaks at phoneme 0.  workspace (device) from gsv_align_workspace(T, N); nothing is allocated. */
size_t gsv_align_workspace(int T, int N);
int gsv_align_viterbi(const float* attn, int H, int T, int N, int32_t* assign, void* workspace, size_t workspace_bytes,
                      void* stream);

/* Streaming splice: replaces TTS._sola_algorithm (gsv_tts/TTS.py:1612-1627), which TTS.infer_stream calls between the decode of a
 * chunk and its hand-out (TTS.py:429-431).
 *   prev_tail fp32 [overlap]: the last `overlap` samples of the previous (already spliced) chunk; chunk fp32 [n], n >= overlap.
 *   The chunk is slid by the offset k in [0, min(n, overlap + search_len) - overlap] that maximises
 *   sum_j chunk[k + j] prev_tail[j] / sqrt(sum_j chunk[k + j]^2 + 1e-8) (first maximum), then cross-faded with
 *   alpha = linspace(0, 1, overlap):  out[i] = prev_tail[i] (1 - alpha_i) + chunk[k + i] alpha_i for i < overlap, chunk[k + i] behind.
 *   -> out fp32 (capacity n; n - *offset samples are written), offset int32 [1] (device; read it behind the stream).
 *   workspace: gsv_sola_workspace(search_len) bytes, device.  Two launches, nothing allocated. */
size_t gsv_sola_workspace(int search_len);
int gsv_sola(const float* prev_tail, const float* chunk, int n, int overlap, int search_len, float* out, int32_t* offset,
             void* workspace, size_t workspace_bytes, void* stream);

/* Reference-audio path, once per new speaker / prompt; fp32 in both numerics modes.
 * Replaces, on the device:
 *   gsv_ref_spectrogram     the torchaudio Spectrogram inside TTS._get_spec (gsv_tts/TTS.py:1591-1604: n_fft /
 *                           win = filter_length, hop_length, periodic hann, center + reflect padding, power 1)
 *   gsv_ref_get_ge          SynthesizerTrn.get_ge (SoVITS/models.py:371-378): ref_enc = MelStyleEncoder
 *                           (module/modules.py:367-444) on refer[:, :704], + sv_emb(sv) and PReLU for v2Pro / v2ProPlus
 *   gsv_ref_extract_latent  SynthesizerTrn.extract_latent (models.py:431-434): ssl_proj (k 2, stride 2) and the
 *                           nearest-codebook search of EuclideanCodebook.quantize (module/core_vq.py:124-128)
 * Tensors ("ref_enc.*", "sv_emb.*", "prelu.weight", "ssl_proj.*", "quantizer.vq.layers.0._codebook.embed") are
 * given under their checkpoint names, device fp32, before finalize.  Audio decoding / resampling and the CN-HuBERT
 * and ERes2Net models that produce `ssl` and `sv_emb` are outside this library. */
typedef struct gsv_ref gsv_ref;
typedef struct gsv_ref_config {
    int n_fft;      /* hps.data.filter_length == win_length (2048) */
    int hop;        /* hps.data.hop_length (640) */
    int spec_bins;  /* spectrogram bins ref_enc reads (704, models.py:305,373) */
    int hidden;     /* MelStyleEncoder style_hidden (128) */
    int n_head;     /* 2 */
    int kernel;     /* Conv1dGLU kernel (5) */
    int gin;        /* gin_channels: 512 (v2) / 1024 (v2Pro, v2ProPlus) */
    int sv_dim;     /* 20480 for v2Pro / v2ProPlus, 0 for v2 (no sv_emb / prelu) */
    int ssl_dim;    /* 768 */
    int bins;       /* codebook size (1024) */
} gsv_ref_config;
int gsv_ref_create(const gsv_ref_config* cfg, gsv_ref** out);
int gsv_ref_destroy(gsv_ref* h);
int gsv_ref_load_tensor(gsv_ref* h, const char* name, const float* data, int64_t numel, void* stream);
int gsv_ref_finalize(gsv_ref* h, void* stream);
/* bytes that cover a spectrogram of n_samples, a get_ge of n_frames and an extract_latent of n_ssl (0 = not used) */
size_t gsv_ref_workspace(gsv_ref* h, int n_samples, int n_frames, int n_ssl);
/* audio fp32 [n_samples] (mono, at the model rate) -> spec fp32 [n_fft/2+1][1 + n_samples/hop], channels-first */
int gsv_ref_spectrogram(gsv_ref* h, const float* audio, int n_samples, float* spec, void* workspace, size_t workspace_bytes,
                        void* stream);
/* spec fp32 [>= spec_bins][n_frames] channels-first (row stride n_frames), sv_emb fp32 [sv_dim] or NULL -> ge fp32 [gin] */
int gsv_ref_get_ge(gsv_ref* h, const float* spec, int n_frames, const float* sv_emb, float* ge, void* workspace,
                   size_t workspace_bytes, void* stream);
/* ssl fp32 [ssl_dim][n_ssl] channels-first (CN-HuBERT last_hidden_state transposed, TTS.py:1567) -> codes int64
 * [n_ssl/2]; margin fp32 [n_ssl/2] or NULL = distance gap between the best and the second-best code */
int gsv_ref_extract_latent(gsv_ref* h, const float* ssl, int n_ssl, int64_t* codes, float* margin, void* workspace,
                           size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GSV_TTS_HIP_H */
