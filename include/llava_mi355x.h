/* llava_mi355x.h — C ABI of libllava_mi355x.so (MI355X / gfx950 native LLaVA forward path)
 *
 * The reference (LLaVA-VL/LLaVA-Plus-Codebase) has no FFI: its seam is a Python class API whose arithmetic lives in
 * third-party torch/transformers code.  This header is the boundary a maintainer binds (ctypes stub in
 * INTEGRATION.md) to run that API on hand-written HIP kernels.  Each entry point cites the reference interface it
 * replaces (paths relative to the reference repo; HF5: = transformers 5.15 sources the reference calls into).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; lmx_last_error() gives the thread-local message.
 *     Native code never aborts/exits (llava/serve/model_worker.py:194-218 turns exceptions into error JSON).
 *   - "dev" pointers are HIP device pointers (e.g. torch.Tensor.data_ptr()); buffers are caller-owned unless stated.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); work is enqueued, not
 *     synchronised, unless stated.
 *   - activations and weights share one dtype (LMX_DTYPE_*), chosen at lmx_create.
 */
#ifndef LLAVA_MI355X_H
#define LLAVA_MI355X_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LMX_ABI_VERSION 1

enum { LMX_DTYPE_F32 = 0, LMX_DTYPE_BF16 = 1, LMX_DTYPE_F16 = 2 };
enum { LMX_ACT_NONE = 0, LMX_ACT_QUICK_GELU = 1, LMX_ACT_GELU_ERF = 2, LMX_ACT_SILU_MUL = 3 };
enum { LMX_PROJ_LINEAR = 0, LMX_PROJ_MLP_GELU = 1, LMX_PROJ_IDENTITY = 2 };
enum { LMX_FEATURE_PATCH = 0, LMX_FEATURE_CLS_PATCH = 1 };

/* Model geometry.  Mirrors the HF config fields the reference reads:
 *   LlamaConfig (hidden_size, intermediate_size, num_hidden_layers, num_attention_heads, num_key_value_heads,
 *   vocab_size, rms_norm_eps, rope_theta, max_position_embeddings) — llava/model/language_model/llava_llama.py:31-33;
 *   CLIPVisionConfig (hidden_size, intermediate_size, num_hidden_layers, num_attention_heads, image_size, patch_size,
 *   layer_norm_eps) — llava/model/multimodal_encoder/clip_encoder.py:21-24;
 *   mm_vision_select_layer / mm_vision_select_feature / mm_projector_type — clip_encoder.py:14-15,29-37,
 *   llava/model/multimodal_projector/builder.py:33-51. */
typedef struct lmx_config {
    int32_t abi_version;         /* LMX_ABI_VERSION */
    int32_t dtype;               /* LMX_DTYPE_* */
    /* decoder */
    int32_t hidden_size, intermediate_size, n_layers, n_heads, n_kv_heads, head_dim, vocab_size;
    float rms_eps;
    float rope_theta;
    int32_t max_position;        /* KV-cache capacity per sequence (rounded up to 128) */
    /* vision tower (n_layers_total = depth of the checkpoint; only the layers needed for select_layer run) */
    int32_t v_hidden, v_intermediate, v_layers, v_heads, v_image_size, v_patch_size;
    float v_ln_eps;
    int32_t select_layer;        /* index into hidden_states, python semantics (-2 = output of layer v_layers-1) */
    int32_t select_feature;      /* LMX_FEATURE_* */
    int32_t projector_type;      /* LMX_PROJ_* */
    int32_t projector_depth;     /* N of mlpNx_gelu */
    /* tensor parallel (decoder only; tower/projector replicated) */
    int32_t tp_rank, tp_world;
    int32_t gemm_variant;        /* 0 = auto; tuning/debug hook */
    int32_t reserved[8];
} lmx_config;

typedef struct lmx_model lmx_model;
typedef struct lmx_seq lmx_seq;
typedef struct lmx_batch lmx_batch;

const char* lmx_last_error(void);
int lmx_abi_version(void);

/* ---- model lifetime -------------------------------------------------------------------------------------------
 * replaces: LlavaLlamaForCausalLM.__init__ / load_pretrained_model weight placement
 *           (llava/model/language_model/llava_llama.py:43-52, llava/model/builder.py:26-151) */
int lmx_create(const lmx_config* cfg, lmx_model** out);
int lmx_destroy(lmx_model* m);

/* Copy one checkpoint tensor into the engine (engine-owned, re-laid-out: fused QKV, [32 gate|32 up] interleave, K-padded
 * patch embedding, TP shard selection).  `name` uses HF checkpoint names with the llava prefixes stripped:
 *   model.embed_tokens.weight, model.layers.N.{self_attn.{q,k,v,o}_proj,mlp.{gate,up,down}_proj}.weight,
 *   model.layers.N.{input,post_attention}_layernorm.weight, model.norm.weight, lm_head.weight,
 *   mm_projector.{0,2,..}.{weight,bias} | mm_projector.{weight,bias},
 *   vision.embeddings.{class_embedding,patch_embedding.weight,position_embedding.weight}, vision.pre_layrnorm.{weight,bias},
 *   vision.encoder.layers.N.{self_attn.{q,k,v,out}_proj,mlp.{fc1,fc2},layer_norm1,layer_norm2}.{weight,bias}
 * `dev_ptr` is a device pointer to a contiguous tensor of the model dtype. */
int lmx_load_weight(lmx_model* m, const char* name, const void* dev_ptr, int32_t dtype, int32_t ndim, const int64_t* shape, void* stream);
/* Verify every tensor the configuration needs was provided; lists missing names in the error string. */
int lmx_finalize_weights(lmx_model* m);
/* RoPE cos/sin table computed by the host exactly as HF does (HF5:models/llama/modeling_llama.py:73-127):
 * host_cos_sin[p*head_dim + i] = cos(p*inv_freq[i]) for i < head_dim/2, then sin for the upper half. */
int lmx_set_rope_table(lmx_model* m, const float* host_cos_sin, int32_t n_pos);

/* ---- tensor parallel -------------------------------------------------------------------------------------------
 * new in this build (the reference only has accelerate layer placement, llava/model/builder.py:26-30). */
int lmx_tp_unique_id(void* out_128_bytes);                                   /* rank 0 creates, host broadcasts */
int lmx_tp_init(lmx_model* m, const void* unique_id_128_bytes);              /* all ranks */
int lmx_tp_comm_ranks(lmx_model* m);      /* ncclCommCount of the engine's communicator (0 = none): lets a driver verify how many ranks RCCL really joined */
/* One-shot peer-to-peer all-reduce for decode-sized messages (<= 32 rows of [hidden]): every rank writes its rows into every
 * peer's exchange buffer over xGMI (HIP IPC mapping) and sums what it received — one launch, one hop, instead of a ring.
 *   lmx_tp_p2p_local_handle: allocate this rank's exchange buffer, return its hipIpcMemHandle_t (64 bytes); the host gathers
 *   the handles of all ranks (rank order) and gives them to lmx_tp_p2p_connect on every rank.  lmx_tp_p2p_enable(0) falls back
 *   to RCCL for everything; lmx_tp_p2p_status != 0 means a wait for a peer timed out (id of that all-reduce).
 *   lmx_op_allreduce: the decoder's all-reduce on a caller buffer [count] of the model dtype (self-test / microbenchmark). */
int lmx_tp_p2p_local_handle(lmx_model* m, void* out_64_bytes);
int lmx_tp_p2p_connect(lmx_model* m, const void* handles_world_x_64_bytes);
int lmx_tp_p2p_enable(lmx_model* m, int32_t on);
int lmx_tp_p2p_status(lmx_model* m, void* stream);
int lmx_op_allreduce(lmx_model* m, void* buf_dev, uint64_t count, void* stream);
/* Test hook: route the decoder's all-reduce through `hook(buf_dev, count, dtype, stream, ctx)` instead of RCCL, so the
 * ranks of a TP group can run as threads of one process on one GPU (tests/test_tp_gpu.py). NULL restores RCCL. */
int lmx_tp_set_allreduce_hook(lmx_model* m, void (*hook)(void*, uint64_t, int32_t, void*, void*), void* ctx);

/* ---- vision path -----------------------------------------------------------------------------------------------
 * replaces: LlavaMetaForCausalLM.encode_images = mm_projector(vision_tower(images))   (llava/model/llava_arch.py:94-97,
 *           clip_encoder.py:39-51, multimodal_projector/builder.py:33-51)
 * pixels [n_images,3,S,S] -> feats [n_images * tokens_per_image, hidden_size] */
int lmx_encode_images(lmx_model* m, const void* pixels_dev, int32_t n_images, void* feats_dev, void* stream);
/* CLIPVisionTower.forward + feature_select alone (llava/model/multimodal_encoder/clip_encoder.py:29-51): the selected hidden state
 * WITHOUT the projector, [n_images, tokens_per_image, v_hidden] of the model dtype. */
int lmx_vision_tower(lmx_model* m, const void* pixels_dev, int32_t n_images, void* feats_dev, void* stream);
/* replaces: process_images + expand2square (llava/mm_utils.py:16-44) and the CLIPImageProcessor call inside them (resize shortest
 * edge to the tower's image size with PIL BICUBIC, center crop, rescale 1/255, normalise) for ONE decoded image.
 *   rgb_dev: uint8 [H][W][3] in device memory; pad_to_square != 0 = image_aspect_ratio 'pad' (canvas colour int(mean*255));
 *   pixels_out_dev: [3][S][S] of out_dtype (LMX_DTYPE_*), S = the tower's image size.  The uint8 stage is bit-exact with Pillow. */
/* host only: the resampling tables lmx_preprocess_image uploads (Pillow's precompute_coeffs + normalize_coeffs_8bpc for output
 * indices [first_out, first_out + n_out) of an in_size -> out_size bicubic resample).  Returns ksize (taps per output); bounds_out
 * [n_out][2] = (first source index, tap count), coeffs_out [n_out][ksize] 22-bit fixed point (filled when coeffs_cap is large enough). */
int lmx_preprocess_coeffs(int32_t in_size, int32_t out_size, int32_t first_out, int32_t n_out, int32_t* bounds_out, int32_t* coeffs_out, int32_t coeffs_cap);
int lmx_preprocess_image(lmx_model* m, const uint8_t* rgb_dev, int32_t H, int32_t W, int32_t out_dtype, int32_t pad_to_square,
                         const float* mean3, const float* std3, void* pixels_out_dev, void* stream);
int lmx_tokens_per_image(const lmx_model* m);
/* model.resize_token_embeddings(len(tokenizer)) (llava/model/builder.py:138): the engine's embedding / lm_head tables are allocated
 * once with `vocab_size` rows (a multiple of 8, with headroom for the <im_patch> / <im_start> / <im_end> tokens the loader may
 * add); ids >= n_real are padding: never returned by a greedy pick or a draw, and the host mirror slices them off the logits. */
int lmx_set_vocab_limit(lmx_model* m, int32_t n_real);

/* ---- multimodal splice -----------------------------------------------------------------------------------------
 * replaces: prepare_inputs_labels_for_multimodal (llava/model/llava_arch.py:99-240), integer half on the host:
 *   inputs   input_ids [B,L] int64 with IMAGE_TOKEN_INDEX=-200 markers, attention_mask [B,L] (uint8, may be NULL),
 *            labels [B,L] int64 (may be NULL), tokens_per_image P, n_image_slots = len(image_features); slot_rows
 *            (may be NULL = P each) gives the rows of each slot for list / 5-D `images` (llava_arch.py:114-119),
 *            max_len = config.tokenizer_model_max_length (<=0: none), left_pad = tokenizer_padding_side=="left"
 *   outputs  T = padded length; src [B,T] int32 gather plan (>=0 token id, -1 zero row, -2-k row k of the image-feature
 *            matrix [n_slots*P, H]); out_mask [B,T] uint8; out_pos [B,T] int64; out_labels [B,T] int64.
 * Two-call protocol: call with src==NULL to get *out_T, allocate, call again.  Bit-exact with the reference including
 * the text-only-row quirk (a row without <image> still consumes one feature slot, llava_arch.py:152-159). */
int lmx_splice_plan(const int64_t* input_ids, const uint8_t* attention_mask, const int64_t* labels,
                    int32_t B, int32_t L, int32_t tokens_per_image, const int32_t* slot_rows, int32_t n_image_slots,
                    int32_t max_len, int32_t left_pad,
                    int32_t* out_T, int32_t* src, uint8_t* out_mask, int64_t* out_pos, int64_t* out_labels);
/* device half: inputs_embeds[r] = embed_tokens[src[r]] | image_features[row] | 0   (llava_arch.py:169-181,206-223) */
int lmx_gather_embeds(lmx_model* m, const int32_t* src_dev, int32_t rows, const void* feats_dev, void* embeds_dev, void* stream);

/* ---- decoder ---------------------------------------------------------------------------------------------------
 * A sequence owns its KV cache (K [layer][kv_head][pos][d], Vᵀ [layer][kv_head][d][pos]) and the device-resident
 * decode state; replaces the past_key_values tuple of the reference (llava_arch.py:105; a14 in SURVEY §8). */
int lmx_seq_create(lmx_model* m, lmx_seq** out);
int lmx_seq_destroy(lmx_seq* s);
int lmx_seq_reset(lmx_seq* s);                       /* length := 0 (cache contents stay finite) */
int lmx_seq_length(const lmx_seq* s);                /* tokens currently in the cache (host mirror) */

/* replaces: LlamaForCausalLM.forward with inputs_embeds (llava_llama.py:88-99 -> HF5:models/llama/modeling_llama.py:
 * 367-494) for T new positions appended to the sequence, optionally in chunks (chunked prefill).
 *   embeds_dev [T, hidden]; logits_dev: [T, vocab] if logits_all else [1, vocab] (last position), may be NULL;
 *   greedy != 0: argmax of the last position is left in the sequence's device token slot for lmx_decode. */
int lmx_prefill(lmx_model* m, lmx_seq* s, const void* embeds_dev, int32_t T, int32_t chunk,
                void* logits_dev, int32_t logits_all, int32_t greedy, void* stream);

/* replaces: the same forward with `output_hidden_states=True`, which LlavaLlamaForCausalLM.forward passes through to LlamaModel
 * (llava/model/language_model/llava_llama.py:63-64, 88-99 -> HF5:models/llama/modeling_llama.py:367-418: all_hidden_states).
 *   hidden_dev [L + 1, T, hidden] in the model dtype: entry l < L = the rows entering decoder layer l (entry 0 = embeds), entry L = the output
 *   of the final RMSNorm.  logits_dev as lmx_prefill (may be NULL); one piece (no chunking), no pick; under tensor parallelism every rank
 *   receives the same (all-reduced) rows. */
int lmx_prefill_hidden(lmx_model* m, lmx_seq* s, const void* embeds_dev, int32_t T, void* logits_dev, int32_t logits_all,
                       void* hidden_dev, void* stream);

/* replaces: the same forward with `output_attentions=True` and / or `output_hidden_states=True` (llava/model/language_model/llava_llama.py:62-64, 88-99 ->
 * HF5:models/llama/modeling_llama.py:367-418; the eager attention's post-softmax weights, :191-214, returned through :281).
 *   attentions_dev [L, heads, T, len + T] in the model dtype (len = the sequence's length before the call): entry [l, h, i, j] = weight of key j for query row i of
 *   head h in layer l, 0 for the keys a row may not see (j > len + i).  Rounding points of the eager path: q k^T rounded to the model dtype, x 1/sqrt(head_dim)
 *   rounded again, softmax in fp32, cast.  The fused attention kernels that compute the layer's output never materialise this matrix; it is recomputed from
 *   the rotated q rows and the K cache for this call (analysis / debugging feature, not a serving path).  hidden_dev as lmx_prefill_hidden; either may be NULL,
 *   not both.  Single process only (a tensor-parallel rank holds a slice of the heads): LMX error otherwise. */
int lmx_prefill_outputs(lmx_model* m, lmx_seq* s, const void* embeds_dev, int32_t T, void* logits_dev, int32_t logits_all,
                        void* hidden_dev, void* attentions_dev, void* stream);

/* replaces: one iteration of GenerationMixin's loop: forward of one token with the KV cache + greedy pick
 * (llava_llama.py:101-108, llava_arch.py:103-112, model_worker.py:174-185).
 *   token >= 0: feed this id; token < 0: feed the id left on the device by the previous greedy step.
 *   n_steps > 1 (greedy only) chains steps on the device with no host round trip.
 *   logits_dev: [1, vocab] of the LAST step or NULL.  Generated ids are appended to the sequence's device token log. */
int lmx_decode(lmx_model* m, lmx_seq* s, int64_t token, int32_t n_steps, void* logits_dev, int32_t greedy, void* stream);
/* replaces: the sampling half of GenerationMixin.sample() as the worker uses it (model_worker.py:156-184: do_sample when
 * temperature > 0.001, TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper -> softmax -> multinomial).
 *   temperature <= 0: the sequence picks greedily (default).  Otherwise every "greedy" pick of lmx_prefill / lmx_decode /
 *   lmx_decode_batch for this sequence becomes a draw on the device (RNG: Philox-4x32-10 keyed by `seed`, counter = tokens produced
 *   so far), so sampled requests chain steps and join decode batches exactly like greedy ones.  top_k = 0 and top_p = 1 switch the
 *   respective filter off. */
int lmx_seq_set_sampling(lmx_seq* s, float temperature, float top_p, int32_t top_k, uint64_t seed);
/* replaces: the per-token stop test the reference runs on the HOST after every step — `eos_token_id` inside HF generate() and the id rule of
 * KeywordsStoppingCriteria, `(output_ids[0, -len(kw):] == kw).all()` (llava/mm_utils.py:94-107; model_worker.py:160-181 builds it from the request's
 * "stop" string) — one device-to-host copy per token.  Here the rule lives with the sequence on the device: the pick of lmx_prefill / lmx_decode /
 * lmx_decode_batch tests the token it just appended (up to 4 EOS ids; up to 4 keyword id sequences of 1..8 ids against the last generated ids) and,
 * once it fired, later picks leave the sequence untouched: no token is logged, the position does not move, a decode batch reports -1 for the member.
 * Steps queued ahead (a chained lmx_decode, the scheduler's two-deep pipeline) therefore produce nothing past the stop.  The text rule of the same
 * class (`keyword in decoded text`) needs the tokenizer and stays on the host.  n_eos = n_kw = 0 clears the rule; a recycled sequence starts without one.
 *   kw_flat: the keywords' ids back to back, kw_lens[k] ids each.  Call before lmx_prefill so that the prefill's own pick is tested too. */
int lmx_seq_set_stop(lmx_seq* s, const int64_t* eos_ids, int32_t n_eos, const int64_t* kw_flat, const int32_t* kw_lens, int32_t n_kw, void* stream);
/* out = 1 once the sequence's stop rule has fired (synchronises `stream`) */
int lmx_seq_stopped(lmx_seq* s, int32_t* out, void* stream);

/* ---- decode batch: continuous batching --------------------------------------------------------------
 * new in this build: the reference serves up to --limit-model-concurrency requests as independent generate() threads with no
 * batching (llava/serve/model_worker.py:174-185, :264); here the decode steps of those requests share ONE pass over the
 * weights.  A batch owns the workspaces for up to `capacity` sequences; which sequences take part is decided per call, so
 * requests join after their own (chunked) lmx_prefill and leave when they finish.
 *   seqs[n]: members of this step (each prefilled, distinct, created from `m`); tokens_host[n] or NULL: id to feed per member
 *   (< 0 / NULL: the id its previous greedy step left on the device); n_steps > 1 (greedy only) chains steps on the device;
 *   logits_dev: [n, vocab] of the LAST step or NULL; ids_out_host: [n_steps, n] greedy picks copied to the host (the call then
 *   synchronises the stream) or NULL.  Every member's position, KV cache and device token log advance exactly
 *   as with lmx_decode.  One batch is driven by one thread at a time; its members must not be used by lmx_decode meanwhile. */
int lmx_batch_create(lmx_model* m, int32_t capacity, lmx_batch** out);
int lmx_batch_destroy(lmx_batch* b);
int lmx_decode_batch(lmx_model* m, lmx_batch* b, lmx_seq* const* seqs, int32_t n, const int64_t* tokens_host, int32_t n_steps,
                     void* logits_dev, int32_t greedy, int64_t* ids_out_host, void* stream);

/* Same step(s), but the picks are copied to PINNED host memory [n_steps, n] asynchronously and the call returns without waiting:
 * the caller orders its read with an event / stream synchronisation and may enqueue the next step first (the scheduler overlaps
 * the host-side per-token callbacks of step k with the GPU work of step k+1). */
int lmx_decode_batch_async(lmx_model* m, lmx_batch* b, lmx_seq* const* seqs, int32_t n, int32_t n_steps, int64_t* ids_out_pinned_host, void* stream);

/* copy ids produced by greedy steps (prefill's pick first) to the host; synchronises the stream. */
int lmx_seq_read_tokens(lmx_seq* s, int64_t* host_out, int32_t max_n, int32_t* n_out, void* stream);

/* ---- in-situ kernel timing -------------------------------------------------------------------------------------
 * While enabled, every launch group of encode_images / prefill / decode is bracketed by a HIP event pair recorded on the
 * launch stream (no host synchronisation, kernels keep running back-to-back); lmx_profile_read synchronises the device,
 * resolves the elapsed times and returns them accumulated per group name ("prefill.gemm.qkv", "decode.gemv.gate_up", ...).
 * bench.py derives its `roofline` objects from this. */
int lmx_profile_enable(lmx_model* m, int32_t on);
int lmx_profile_read(lmx_model* m, char* names_buf, int32_t names_cap, double* ms, int64_t* counts, int32_t max_n, int32_t* n_out);
/* Options of a live model (defaults come from the environment at lmx_create: LMX_FUSE_ROPE, LMX_VIS_PACK, LMX_DECODE_SPLITQ; all default 1).  Every one selects
 * between two launch forms with bit-identical results — the switch exists for A/B timing and for the tests that prove the identity:
 *   "fuse_rope"     RoPE + KV-cache append in the prefill's q|k|v GEMM epilogue (HF5:models/llama/modeling_llama.py:130-160,243-281) vs. a rope_kv launch
 *   "vis_pack"      CLIP K / V^T pack in the tower's q|k|v GEMM epilogue (HF5:models/clip/modeling_clip.py:295-340) vs. a pack launch
 *   "decode_splitq" decode step of 16-bit models as q launch + (k|v projection with the attention workgroups) launch vs. q|k|v launch + attention launch
 * Unknown keys are an error.  Not to be flipped while requests are in flight on other threads. */
int lmx_model_set_option(lmx_model* m, const char* key, int32_t value);

/* ---- single-op entry points (unit parity tests + microbenchmarks; same kernels the engine launches) --------------- */
int lmx_op_gemm(int32_t dtype, const void* x, const void* w, void* c, const void* bias, const void* residual,
                int32_t M, int32_t N, int32_t K, int32_t ldx, int32_t ldw, int32_t ldc, int32_t ldr, int32_t act, int32_t variant, void* stream);
int lmx_op_gemv(int32_t dtype, const void* x, const void* w, void* c, const void* bias, const void* residual, const void* norm_w, float eps,
                int32_t MB, int32_t N, int32_t K, int32_t ldx, int32_t ldw, int32_t ldc, int32_t ldr, int32_t act, void* stream);
int lmx_op_rmsnorm(int32_t dtype, const void* x, const void* w, void* y, int32_t rows, int32_t H, float eps, void* stream);
int lmx_op_layernorm(int32_t dtype, const void* x, const void* w, const void* b, void* y, int32_t rows, int32_t H, float eps, void* stream);
int lmx_op_rope_kv(int32_t dtype, int32_t head_dim, void* qkv, void* kcache, void* vtcache, const float* cos_sin_dev, int32_t pos0,
                   int32_t T, int32_t n_heads, int32_t n_kv_heads, int32_t s_max, void* stream);
/* lmx_op_rope_kv_rows: the same, and the rotated k also REPLACES the k columns of qkv (same bits as the cache rows): after it qkv holds [rotated q | rotated k | v]
 * per row, which is what the attention backward of the training step reads (lmx_op_attn_bwd with k / v = column windows of qkv: no gathered copies). */
int lmx_op_rope_kv_rows(int32_t dtype, int32_t head_dim, void* qkv, void* kcache, void* vtcache, const float* cos_sin_dev, int32_t pos0,
                        int32_t T, int32_t n_heads, int32_t n_kv_heads, int32_t s_max, void* stream);
/* q|k|v projection with RoPE + KV-cache append in the GEMM's epilogue (gemm8p.hip: qkv_rope_epilogue; what Model::prefill launches where
 * the shape allows it): qkv[:, :n_heads*D] <- rotated q, kcache / vtcache rows pos0 .. pos0 + T - 1 <- rotated k / v; the k | v columns of `qkv` are
 * left untouched.  Fails when the shape does not take the fused launch (same rule as the engine: un-split ping-pong GEMM, head-aligned tiles). */
int lmx_op_gemm_qkv_rope(int32_t dtype, int32_t head_dim, const void* x, const void* w, void* qkv, void* kcache, void* vtcache, const float* cos_sin_dev,
                         int32_t pos0, int32_t T, int32_t K, int32_t n_heads, int32_t n_kv_heads, int32_t s_max, void* stream);
int lmx_op_flash_attn(int32_t dtype, int32_t head_dim, const void* q, void* o, const void* kcache, const void* vtcache,
                      int32_t q_len, int32_t kv_len, int32_t q_pos0, int32_t q_stride, int32_t o_stride,
                      int32_t n_heads, int32_t n_kv_heads, int32_t s_max, float scale, int32_t causal, void* stream);
int lmx_op_decode_attn(int32_t dtype, int32_t head_dim, const void* q, void* o, const void* kcache, const void* vtcache,
                       int32_t n_rows, int32_t pos0, int32_t kv_total, int32_t causal, int32_t q_stride, int32_t o_stride,
                       int32_t n_heads, int32_t n_kv_heads, int32_t s_max, int32_t n_split, float scale, void* ws_dev, void* stream);
/* fused single-token decode attention (RoPE + KV append + split attention + in-launch merge); ws: n_heads*ceil(s_max/128)*(D+4) floats,
 * counters: n_heads int32 zeroed once.  debug_mode != 0 is for microbenchmarks only. */
int lmx_op_decode_fused(int32_t dtype, int32_t head_dim, const void* qkv, void* kcache, void* vtcache, const float* cos_sin_dev, const int32_t* pos_dev,
                        int32_t n_heads, int32_t n_kv_heads, int32_t s_max, float scale, void* ws_dev, int32_t* counters_dev, void* out, int32_t debug_mode, void* stream);
/* the decode step's attention launch for 16-bit models (csrc/decode_attn.hip: decode_attn_step_kernel): the same contract with the position BY VALUE (the engine
 * passes its host mirror) and only the live 128-key chunks launched; ws: n_heads * ceil(s_max/128) * (D+4) floats.  Replaces the attention of
 * HF5:models/llama/modeling_llama.py:243-281 for a single cached token (llava_arch.py:103-112 is the caller's one-token branch). */
int lmx_op_decode_attn_step(int32_t dtype, int32_t head_dim, void* qkv, void* kcache, void* vtcache, const float* cos_sin_dev, int32_t pos,
                            int32_t n_heads, int32_t n_kv_heads, int32_t s_max, float scale, void* ws_dev, int32_t* counters_dev, void* out, void* stream);
/* the split-q form of the same step (decode_kv_attn_kernel): `qkv` holds the q columns (a preceding q projection wrote them); THIS launch computes the k | v rows
 * — w_kv = rows [n_heads D, (n_heads + 2 n_kv_heads) D) of the fused q|k|v weight [.., K], input row x [K] with LlamaRMSNorm(norm_w, eps) fused when norm_w is
 * non-null — next to the attention workgroups, and hands the newest key / value to them inside the launch through `granules` (2 n_kv_heads D x 8 bytes, zeroed
 * once; `tag` non-zero and different for every launch on the same granules).  Output and cache effects are bit-identical to lmx_op_gemv (all q|k|v rows) +
 * lmx_op_decode_attn_step.  timeline_dev (may be null): 10 x uint64 of in-kernel clock stamps for microbenchmarks (csrc/kernels.h: DecAttnArgs::ts). */
int lmx_op_decode_kv_attn(int32_t dtype, int32_t head_dim, void* qkv, const void* x, const void* w_kv, const void* norm_w, float eps, int32_t K, int32_t ldw,
                          void* kcache, void* vtcache, const float* cos_sin_dev, int32_t pos, int32_t n_heads, int32_t n_kv_heads, int32_t s_max, float scale,
                          void* ws_dev, int32_t* counters_dev, void* granules_dev, uint32_t tag, void* out, void* timeline_dev, void* stream);
/* the decode BATCH's attention launch (continuous batching; what Model::decode_batch issues per layer): sequence z of n_seq has its own caches
 * kcaches[z] / vtcaches[z] (this layer's [n_kv_heads][s_max][D] rows and [n_kv_heads][D][s_max] columns) and its own device position word pos_devs[z]
 * (keys already cached; the new key / value of row z of `qkv` are appended there); rows of `qkv` / `out` by element stride.  16-bit models with
 * head_dim 128 take decode_attn_wave_kernel (csrc/attention_batch.h; two heads per workgroup from n_heads * n_seq >= 512 on), everything else the
 * chunked decode_fused_kernel, which needs ws_devs[z] (n_heads * n_split * (D + 4) floats) and counters_devs[z] (n_heads zeroed int32) — both arrays
 * may be null for the wave kernel.  n_split = 128-key chunks to visit (>= longest position / 128 + 1).  The pointer arrays are HOST arrays of device
 * pointers; tab_dev = lmx_op_decode_attn_batch_tab_bytes(n_seq) bytes of device scratch.  Replaces, per sequence, the single-token branch of
 * llava/model/llava_arch.py:103-112 -> HF5:models/llama/modeling_llama.py:191-214, 243-281 (the reference has no batching: model_worker.py:174-185). */
int lmx_op_decode_attn_batch(int32_t dtype, int32_t head_dim, const void* qkv, int32_t qkv_stride, void* const* kcaches, void* const* vtcaches,
                             const int32_t* const* pos_devs, void* const* ws_devs, int32_t* const* counters_devs, int32_t n_seq, const float* cos_sin_dev,
                             int32_t n_heads, int32_t n_kv_heads, int32_t s_max, int32_t n_split, float scale, void* tab_dev, void* out, int32_t o_stride,
                             void* stream);
size_t lmx_op_decode_attn_batch_tab_bytes(int32_t n_seq);
size_t lmx_op_decode_attn_ws_bytes(int32_t n_rows, int32_t n_heads, int32_t n_split, int32_t head_dim);
int lmx_op_sample(int32_t dtype, const void* logits_dev, int32_t V, float temperature, float top_p, int32_t top_k, uint64_t seed,
                  const int32_t* offset_dev, const uint32_t* u32_override_host, int64_t* out_tok_dev, uint8_t* keep_out_dev, void* stream);
int lmx_op_argmax(int32_t dtype, const void* logits, int32_t V, int64_t* out_tok_dev, void* stream);
int lmx_op_im2col(int32_t dtype, const void* pixels, void* out, int32_t N, int32_t S, int32_t patch, int32_t kpad, void* stream);

/* ---- training-step slices (csrc/train.hip): parity-first kernels for the finetuning step of llava/train/train.py:805-1000.
 * lmx_op_ce_loss: LlamaForCausalLM's shifted cross-entropy with ignore_index (labels from llava_arch.py:181,200; IGNORE_INDEX = -100):
 *   logits [B*T][ld], labels [B][T] int64 (position t is scored against labels[t + 1]); out_loss_count[0] = mean loss, [1] = counted
 *   positions; scratch: B*(T-1) floats each; dlogits (optional, [B*T][ldd]) = grad * d(loss)/d(logits).
 * lmx_op_rmsnorm_bwd / swiglu_bwd / rope_bwd: autograd of LlamaRMSNorm, silu(gate)*up and apply_rotary_pos_emb
 *   (HF5:models/llama/modeling_llama.py:53-67, 163-176, 138-160); rmsnorm_bwd's inv_scratch holds round_up(rows, 64) + ceil(rows / 128) * H floats
 *   (inverse norms, then the weight gradient's partial rows: the caller owns every byte of scratch).  lmx_op_transpose: operand re-layout so that dgrad / wgrad run on lmx_op_gemm.
 * lmx_op_attn_bwd: causal attention backward (contract of llava/train/llama_flash_attn_monkey_patch.py:68-91): q / k / v / d_out as
 *   [T][heads][head_dim] rows of stride ldq / ldk / ldo elements; dk32 / dv32: T*kv_heads*head_dim fp32 scratch. */
int lmx_op_ce_loss(int32_t dtype, const void* logits, int32_t ld, const int64_t* labels, int32_t B, int32_t T, int32_t V, int64_t ignore_index,
                   float* lse_scratch, float* row_loss_scratch, float* out_loss_count, float grad, void* dlogits_or_null, int32_t ldd, void* stream);
int lmx_op_rmsnorm_bwd(int32_t dtype, const void* x, const void* w, const void* dy, void* dx, float* dw_or_null, float* inv_scratch, int32_t rows,
                       int32_t H, float eps, void* stream);
/* lmx_op_rmsnorm_bwd_add: the same with the residual branch's gradient joined in the same pass: dx = T(residual + T(dx_norm)) — the two roundings of
 *   lmx_op_rmsnorm_bwd followed by an elementwise add, without the add's launch (x1 = x + f(norm(x)): d/dx = d_out + norm_bwd(d_f)).  residual must not overlap dx. */
int lmx_op_rmsnorm_bwd_add(int32_t dtype, const void* x, const void* w, const void* dy, const void* residual, void* dx, float* dw_or_null, float* inv_scratch,
                           int32_t rows, int32_t H, float eps, void* stream);
int lmx_op_swiglu_bwd(int32_t dtype, const void* gate, const void* up, const void* dact, void* dgate, void* dup, int64_t n, void* stream);
int lmx_op_rope_bwd(int32_t dtype, const void* dy, void* dx, const float* cos_sin_dev, int32_t pos0, int32_t T, int32_t heads, int32_t head_dim, int32_t ld,
                    void* stream);
int lmx_op_transpose(int32_t dtype, const void* src, int32_t ld, int32_t rows, int32_t cols, void* dst, int32_t ldd, void* stream);
/* lmx_op_gemm_wgrad: the weight half of nn.Linear's backward, grad_weight = grad_output^T @ input (torch autograd under llava/train/train.py:780-1000), with BOTH
 *   operands in their forward layout: out[out_features][in_features] = sum_r dy[r][o] * x[r][i], dy [rows][lddy], x [rows][ldx]; no transposed copies
 *   (csrc/gemm8t.hip: LDS transpose reads).  16-bit dtypes; out_features and in_features multiples of 256, rows a multiple of 64, lddy / ldx multiples of 8,
 *   operands 16-byte aligned and below 4 GiB: lmx_op_gemm_wgrad_supported returns 1 for such a call, 0 otherwise (the caller then transposes and calls
 *   lmx_op_gemm; an unsupported lmx_op_gemm_wgrad call FAILS).  Bit-identical to that two-transpose path. */
int lmx_op_gemm_wgrad(int32_t dtype, const void* dy, int32_t lddy, const void* x, int32_t ldx, int32_t rows, int32_t out_features, int32_t in_features, void* out,
                      int32_t ldo, void* stream);
int lmx_op_gemm_wgrad_supported(int32_t dtype, int32_t lddy, int32_t ldx, int32_t rows, int32_t out_features, int32_t in_features, int32_t ldo);
int lmx_op_attn_bwd(int32_t dtype, int32_t head_dim, const void* q, const void* k, const void* v, const void* d_out, void* dq, float* dk32_scratch,
                    float* dv32_scratch, void* dk, void* dv, int32_t T, int32_t heads, int32_t kv_heads, int32_t ldq, int32_t ldk, int32_t ldo, float scale,
                    void* stream);
/* The same attention with the forward's softmax statistics kept for the backward, as FlashAttention-2 does (the reference's training attention:
 * llava/train/llama_flash_attn_monkey_patch.py:68-91 -> flash_attn_unpadded_qkvpacked_func):
 * lmx_op_flash_attn_lse = lmx_op_flash_attn + lse[head][row] = log2(sum_j exp2(scale * log2(e) * q_row . k_j)) over the visible keys, [n_heads][lse_stride >= q_len] floats
 *   (rows >= q_len are not written: zero the buffer when lse_stride is rounded up);
 * lmx_op_attn_bwd_lse = lmx_op_attn_bwd given that lse (lse_stride >= T rounded up to 64) and the forward's output rows out [T][ldout]: the query-side kernel skips its
 *   statistics sweep (two of its five products) and delta_i = sum_d d_out[i][d] * out[i][d].  16-bit dtypes only (no scratch arguments: the fp32 path is lmx_op_attn_bwd). */
int lmx_op_flash_attn_lse(int32_t dtype, int32_t head_dim, const void* q, void* o, const void* kcache, const void* vtcache, int32_t q_len, int32_t kv_len, int32_t q_pos0,
                          int32_t q_stride, int32_t o_stride, int32_t n_heads, int32_t n_kv_heads, int32_t s_max, float scale, int32_t causal, float* lse,
                          int32_t lse_stride, void* stream);
int lmx_op_attn_bwd_lse(int32_t dtype, int32_t head_dim, const void* q, const void* k, const void* v, const void* out, const void* d_out, const float* lse,
                        int32_t lse_stride, void* dq, void* dk, void* dv, int32_t T, int32_t heads, int32_t kv_heads, int32_t ldq, int32_t ldk, int32_t ldo,
                        int32_t ldout, float scale, void* stream);

/* ---- one optimisation step (BASELINE config 5; composed by llava_mi355x/train.py) ------------------------------------------
 * Replaces, for the trainable part of the model (LLM + mm_projector; the CLIP tower is frozen, clip_encoder.py:25), what the HF Trainer +
 * DeepSpeed run per step under llava/train/train.py:805-1000 with scripts/zero2.json:
 *   lmx_op_elementwise   op 0: silu(gate) * up with HF's rounding points (LlamaMLP, HF5:models/llama/modeling_llama.py:163-176);
 *                        op 1 / 2: nn.GELU() of the mlp2x_gelu projector and its autograd (multimodal_projector/builder.py:33-51); op 3: a + b
 *   lmx_op_gather_embed  embed_tokens + image-feature splice with an explicit (trainable) table — forward of llava_arch.py:150-225's gather
 *   lmx_op_embed_bwd     its autograd: token rows add into the fp32 table gradient, image rows go back to the projector output
 *   lmx_op_col_sum       bias gradient (fp32 column sums);   lmx_op_cast_f32: fp32 accumulator -> parameter dtype
 *   lmx_op_sumsq         acc += sum(x^2): torch.nn.utils.clip_grad_norm_ (HF Trainer max_grad_norm)
 *   lmx_op_adamw         torch.optim.AdamW's update on fp32 master weights + moments, clip factor read from the device */
int lmx_op_elementwise(int32_t dtype, int32_t op, const void* a, const void* b, void* out, int64_t n, void* stream);
int lmx_op_cast_f32(int32_t dtype, const float* src, void* dst, int64_t n, void* stream);
int lmx_op_col_sum(int32_t dtype, const void* dy, int32_t ld, int32_t rows, int32_t cols, float* out, void* stream);
int lmx_op_gather_embed(int32_t dtype, const int32_t* src_dev, const void* table, const void* feats_or_null, void* out, int32_t rows, int32_t H, void* stream);
int lmx_op_embed_bwd(int32_t dtype, const int32_t* src_dev, const void* d_embeds, float* dtable_or_null, void* dfeats_or_null, int32_t rows, int32_t H, void* stream);
int lmx_op_sumsq(int32_t dtype, const void* x, int64_t n, float* acc, void* stream);
int lmx_op_adamw(int32_t dtype, void* param, const void* grad, float* master, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int32_t step, const float* gnorm_sq_or_null, float max_grad_norm, void* stream);

/* ---- beam search (num_beams > 1 of GenerationMixin.generate, passed through by llava/eval/run_llava.py:121, model_vqa_loader.py:104) -------------
 *   lmx_op_beam_topk   device half of a beam step: per beam row, log_softmax(logits) + beam_score and the K best (score, token id) pairs in
 *                      (score desc, id asc) order; the host merges num_beams x K candidates (BeamSearchScorer bookkeeping stays on the host)
 *   lmx_seq_copy       dst := src's context (KV cache of the first len positions + length): the cache reorder of a beam step
 *                      (`_reorder_cache` / index_select over past_key_values in the reference), only for beams that were duplicated */
int lmx_seq_copy(lmx_seq* dst, const lmx_seq* src, void* stream);

/* ---- reuse across the turns of one conversation (LLaVA-Plus tool loop: llava/serve/gradio_web_server_llava_plus.py:498-637 sends the SAME image and the whole
 *      first exchange again with the tool's answer appended; llava/serve/model_worker.py:122-192 re-encodes and re-prefills all of it) -------------------------
 *   lmx_op_hash128     128-bit content hash of `items` device buffers of bytes_per_item bytes each (out_dev[2 * item + {0, 1}], 16 bytes per item): the key of
 *                      the host mirror's image-feature cache (encode_images of pixels already seen returns the stored rows)
 *   lmx_seq_truncate   keep the K / V^T rows of positions [0, n_rows) of a finished request's sequence and forget the rest of that request (position, token
 *                      log, sampling parameters, stop rule): the next request, whose spliced prompt shares those rows, prefills only what follows them */
int lmx_op_hash128(const void* base_dev, uint64_t bytes_per_item, int32_t items, uint64_t* out_dev, void* stream);
int lmx_seq_truncate(lmx_seq* s, int32_t n_rows, void* stream);
/* Packed prefill of several requests (serving; the reference prefills each request alone inside its own generate() thread, model_worker.py:174-185):
 * the rows of all sequences are processed as ONE row block in pieces of block_rows (0 = everything at once), every linear as a single GEMM over
 * the piece; RoPE / KV append / causal attention per sequence against its own cache; each sequence's last row gets the lm_head and its pick
 * (greedy != 0), exactly as lmx_prefill does.  embeds[i]: [n_tokens[i], hidden] rows of request i (output of lmx_gather_embeds). */
int lmx_prefill_batch(lmx_model* m, lmx_seq* const* seqs, int32_t n, const void* const* embeds, const int32_t* n_tokens, int32_t block_rows, int32_t greedy,
                      void* stream);
int lmx_op_beam_topk(int32_t dtype, const void* logits, int32_t ld, int32_t V, int32_t rows, const float* beam_scores_dev, int32_t K, float* out_scores, int32_t* out_ids,
                     void* stream);
/* replaces: the device half of GenerationMixin.beam_sample (transformers 4.31 generation/utils.py: num_beams > 1 with do_sample=True — what
 * llava/eval/run_llava.py:115-125 asks for when --num_beams is raised at its default temperature 0.2): per beam row
 *   s_i = (log_softmax(logits)_i + beam_score) / temperature for the ids keep_dev[row][i] != 0 allows (NULL = every id; the survivor set of the
 *   top-k / top-p warpers, lmx_op_sample's keep_out), key_i = s_i + Gumbel(Philox-4x32-10(seed, counter0 + row * V + i)),
 * and the K largest keys with their scores and ids in (key desc, id asc) order.  The num_beams * K pairs merged by key are 2 num_beams draws without
 * replacement from softmax over the num_beams x V block of warped scores (Plackett-Luce order = what torch.multinomial(replacement=False) samples). */
int lmx_op_beam_sample_topk(int32_t dtype, const void* logits, int32_t ld, int32_t V, int32_t rows, const uint8_t* keep_dev, const float* beam_scores_dev, float temperature,
                            uint64_t seed, uint32_t counter0, int32_t K, float* out_keys, float* out_scores, int32_t* out_ids, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LLAVA_MI355X_H */
