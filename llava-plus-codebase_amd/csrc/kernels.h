// Launch-side contracts of every HIP kernel on the LLaVA forward path (host view).
// All pointers are device pointers unless stated; `dtype` is lmx::DType of activations AND weights.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

namespace lmx {

// ---- linear layers (gemm.hip) -------------------------------------------------------------------------------
struct GemmArgs {
    const void* X;      // [M, K] row-major, leading dim ldx
    const void* W;      // [N, K] row-major (nn.Linear weight layout), leading dim ldw
    void* C;            // [M, N] (or [M, N/2] for kActSiluMul), leading dim ldc
    const void* bias;   // [N] or null (same dtype)
    const void* R;      // residual [M, N] added after the activation, or null; may alias C
    int M, N, K;
    int ldx, ldw, ldc, ldr;
    int act;            // lmx::Act
    const void* Wsw = nullptr;   // skinny kernel only: W re-laid in MFMA-fragment order (skinny_swizzle), else null
    // ping-pong kernel (gemm8p.hip) only: K slices per 256x256 tile (0 = let the launcher pick) and the fp32 partial-tile scratch
    // (gemm8p_splitk_ws_bytes; null = the launcher's own, one stream at a time)
    int split_k = 0;
    void* skw = nullptr;
    int* sk_cnt = nullptr;       // skinny kernel's K slices across workgroups: one arrival ticket per 64-row tile, zero before the first launch (the last arriver re-arms)
    // q|k|v projection of a prefill with RoPE and the KV-cache append in the EPILOGUE (gemm8p.hip: qkv_rope_epilogue; where gemm_fuses_qkv()
    // says so): the tile is rounded to T into LDS and leaves the workgroup as  q rows -> rotated, into C (the q columns; k | v columns of C are not
    // written)   k rows -> rotated, into the K cache [kv head][pos][D]   v rows -> the V^T cache [kv head][d][pos].  Same arithmetic and rounding points as
    // launch_gemm + launch_rope_kv (HF5:models/llama/modeling_llama.py:130-160,243-281).  qf_kc != null switches it on.
    const float* qf_rope = nullptr; void* qf_kc = nullptr; void* qf_vt = nullptr; int qf_pos0 = 0, qf_nh = 0, qf_nkv = 0, qf_smax = 0, qf_D = 0;
    // CLIP q|k|v projection with the K / V^T pack in the epilogue (round 4; HF5:models/clip/modeling_clip.py:295-340 reshapes q, k, v per head; here the attention
    // kernel wants K rows [head][pos][D] and V^T [head][d][pos]): columns [0, pk_heads * pk_D) go to C as usual, the k columns to pk_kc, the v columns
    // transposed to pk_vt (2-byte stores, 64-byte runs per register across the wave's 32 rows), same bias and the same rounding as GEMM + pack launch.  Row m
    // belongs to image m / pk_rows at position m % pk_rows; per-image cache stride pk_img_stride elements.  pk_kc != null switches it on (any 16-bit GEMM
    // kernel that ends in gemm_epilogue; not SiLU*mul).
    void* pk_kc = nullptr; void* pk_vt = nullptr; int pk_heads = 0, pk_D = 0, pk_rows = 0, pk_spad = 0; size_t pk_img_stride = 0;
};
void launch_gemm(int dtype, const GemmArgs& a, int variant, hipStream_t st);
// true when launch_gemm(variant 0) of a [M, (nh + 2 nkv) D] x K projection runs as ONE un-split ping-pong launch whose tiles are head-aligned, i.e. can take
// the qf_* fields (pos0 = first cache position of row 0)
bool gemm_fuses_qkv(int dtype, int M, int K, int D, int nh, int nkv, int pos0, int s_max, bool has_bias);
// ping-pong 256x256x64 kernel (gemm8p.hip): variant 30 of launch_gemm (33 / 34 / 35: 2 / 3 / 1 K slices forced); LMX_GEMM8P=0 keeps launch_gemm's auto choice off it
bool gemm8p_enabled();
void launch_gemm8p(int dtype, const GemmArgs& a, hipStream_t st);
size_t gemm8p_splitk_ws_bytes(int M, int N, int split_k);
int gemm8p_pick_split(int M, int N, int K);
// decode-batch linear (skinny.hip): M <= 32 rows, 16-bit, weights streamed once straight into MFMA operands; variant 20 of launch_gemm
void launch_skinny_gemm(int dtype, const GemmArgs& a, hipStream_t st);
int skinny_kslices(int M, int N, int K);          // K slices the narrow-layer form takes when a.skw / a.sk_cnt are given (1 = the 16-rows-per-workgroup form)
size_t skinny_scratch_bytes(int N);               // fp32 partial tiles of one launch (a.skw); tickets: N / 64 ints (a.sk_cnt)
// fragment-order copy of a [N, K] weight for the skinny kernel: per (16-row tile, 128-k super-step) four 1-KiB pieces, piece j =
// lane-linear 16-byte A fragments of MFMA step j (lane = 16 q + i holds row i, k = 32 q + 8 j ...), K zero-padded to 128
size_t skinny_swizzled_bytes(int N, int K, int es);
void launch_skinny_swizzle(int dtype, const void* W, int ldw, void* dst, int N, int K, hipStream_t st);

struct GemvArgs {
    const void* X;        // [MB, K]
    const void* W;        // [N, K]
    void* C;              // [MB, N] (or N/2 for SiLU·mul)
    const void* bias;     // [N] or null
    const void* R;        // residual [MB, N] or null; may alias C
    const void* norm_w;   // if non-null: x := RMSNorm(x) * norm_w, fused into the x staging
    float eps;
    int N, K;
    int ldx, ldw, ldc, ldr;
    int act;
};
void launch_gemv(int dtype, const GemvArgs& a, int MB, hipStream_t st);
// does launch_gemv(MB = 1) of this shape take the hand-counted weight stream (gemv2.h)?
bool gemv2_applies(int dtype, const GemvArgs& a);

// ---- attention (attention.hip) ------------------------------------------------------------------------------
// K cache layout  : [n_kv_heads][s_max][D]      (key rows, post-RoPE)
// Vᵀ cache layout : [n_kv_heads][D][s_max]      (value columns; keys contiguous so P·V fragments are 8-byte reads)
struct FlashArgs {
    const void* Q;      // [q_len rows][q_stride] ; head h at column h*D
    void* O;            // [q_len rows][o_stride] ; head h at column h*D
    const void* K;      // cache base for this layer
    const void* VT;
    int q_len, kv_len;  // kv_len = keys visible in the cache (past + q_len for a causal prefill chunk)
    int q_pos0;         // absolute position of q row 0 (causal: key j visible to row i iff j <= q_pos0 + i)
    int q_stride, o_stride;
    int n_heads, n_kv_heads, s_max;
    float scale;        // 1/sqrt(D)
    int causal;
    float* lse = nullptr;   // optional [n_heads][lse_stride]: log2-domain log-sum-exp of the scaled scores of every query row (the training backward's statistics)
    int lse_stride = 0;
};
void launch_flash_prefill(int dtype, int D, const FlashArgs& a, hipStream_t st);

struct DecodeAttnArgs {
    const void* Q;         // [n_rows][q_stride]; head h at h*D  (already rotated)
    void* O;               // [n_rows][o_stride]
    const void* K;
    const void* VT;
    const int* pos_ptr;    // device: absolute position of row 0 (the device-resident decode loop reads the live length); may be null
    int pos0;              // used when pos_ptr == null
    int n_rows;
    int kv_total;          // non-causal: every row sees keys [0, kv_total)
    int causal;            // causal: row i sees keys [0, pos0 + i]
    int q_stride, o_stride;
    int n_heads, n_kv_heads, s_max;
    int n_split;           // key-range splits per (row, head)
    float scale;
    float* ws;             // workspace: n_rows * n_heads * n_split * (D + 2) floats
};
void launch_decode_attn(int dtype, int D, const DecodeAttnArgs& a, hipStream_t st);

// `output_attentions=True`: the post-softmax attention weights of one decoder layer as the eager path returns them (HF5:models/llama/modeling_llama.py:191-214:
// scores = q k^T rounded to the model dtype, x 1/sqrt(d) rounded again, causal mask, softmax in fp32, cast to the model dtype).  Not on the serving path: the fused
// attention kernels never materialise the matrix; this launch recomputes it from the rotated q rows and the K cache for callers that ask for it.
struct AttnProbsArgs {
    const void* Q;         // [n_rows][q_stride]; head h at h*D (rotated)
    const void* K;         // this layer's K cache [n_kv_heads][s_max][D] (rows [0, pos0 + n_rows) valid)
    void* P;               // out [n_heads][rows_total][kv_total]; this launch fills rows [row0, row0 + n_rows); keys a row may not see get 0
    int n_rows, row0, rows_total;
    int pos0;              // absolute position of row 0 of this launch: row i sees keys [0, pos0 + i]
    int kv_total;          // row pitch of P (>= pos0 + n_rows)
    int q_stride, n_heads, n_kv_heads, s_max;
    float scale;
};
void launch_attn_probs(int dtype, int D, const AttnProbsArgs& a, hipStream_t st);

// Single-token decode step, fused: RoPE(q,k) + KV-cache append + split-K attention partials, ONE launch.
// Partials (o[D], m, l) per (head, split) go to `ws`; the last workgroup to arrive for a head merges them in the same
// launch (agent-scope counter protocol), so there is no separate combine launch.
// decode batch (continuous batching): per-sequence pointers of ONE layer; workgroup (head, split, z) serves sequence z
struct DecodeFusedSeq { void* K; void* VT; const int* pos_ptr; float* ws; int* counters; };
struct DecodeFusedArgs {
    const void* QKV;       // [qkv_n] row of this token: q heads | k heads | v heads (pre-RoPE)
    void* K;               // caches of this layer (the new key/value are appended at *pos_ptr)
    void* VT;
    const float* cos_sin;  // [max_pos][D]
    const int* pos_ptr;    // device: position of this token = keys already in the cache
    int n_heads, n_kv_heads, s_max, n_split;
    float scale;
    float* ws;             // [n_heads][n_split][D + 4] floats: o[D], m, l, pad, pad
    int* counters;         // [n_heads] arrival tickets, zero before the first launch (the merger re-arms them)
    void* O;               // [n_heads * D] merged attention output (model dtype)
    int debug_mode = 0;    // microbenchmark only: 1 = stop after the partial stores (no ticket / merge), 2 = no partial stores either
    // batch form: tab != null => n_seq sequences in one launch; K / VT / pos_ptr / ws / counters come from tab[z], QKV and O are
    // [n_seq] rows with the given element strides
    const DecodeFusedSeq* tab = nullptr;
    int n_seq = 1, qkv_stride = 0, o_stride = 0;
};
void launch_decode_fused(int dtype, int D, const DecodeFusedArgs& a, hipStream_t st);
size_t decode_fused_ws_floats(int n_heads, int n_split, int D);
size_t decode_attn_ws_floats(int n_rows, int n_heads, int n_split, int D);

// ---- decode-step attention of a single request (decode_attn.hip; 16-bit models) ---------------------------------------------------------------------
// RoPE(q, k_new) + KV append + attention over 128-key chunks + in-launch merge, position BY VALUE (the engine mirrors *d_len on the host), only the live
// chunks launched: grid = nh x n_split.  Same contract and arithmetic as launch_decode_fused (bit-identical output).
struct DecAttnArgs {
    const void* qkv;                 // this token's row q | k | v (pre-RoPE); split-q form: only the q columns are read
    void* attn;                      // [nh * D] merged attention output (model dtype)
    void* kc; void* vt;              // this layer's caches (the new key / value are appended at `pos`)
    const float* rope;               // [max_pos][D]
    float* aws; int* cnt;            // partials [nh][n_split][D + 4] floats; arrival tickets [nh], zero before the first launch (the merger re-arms them)
    int pos, n_split;                // position of this token = keys already cached; live chunks = pos / 128 + 1
    int nh, nkv, s_max;
    float scale;
    // split-q form only (launch_decode_kv_attn): k_new | v_new arrive as {bits, tag} granules [2 nkv D] published by the launch's own k|v projection
    unsigned long long* kv_gran = nullptr; unsigned tag = 0;       // tag: non-zero, unique per launch on this granule buffer
    unsigned pub_tag = 0;            // tag the projection publishes with; 0 = `tag`.  Anything else is fault injection (tests: the waiters must time out cleanly)
    unsigned* status = nullptr;      // host-mapped word raised when the (bounded) wait for the granules times out, or null
    // debug timeline (lmx_op_decode_kv_attn with ts_dev; null in the engine), s_memrealtime ticks (100 MHz): [0] earliest attention workgroup start, [1..5] head 0's
    // merger: own partial done / arrivals seen / merged / granules arrived / row stored, [6] earliest projection workgroup start, [7] latest projection
    // row published, [8] latest merger done, [9] latest non-merger arrival.  [0] and [6] must be preset to ~0ull, the others to 0.
    unsigned long long* ts = nullptr;
};
void launch_decode_attn_step(int dtype, int D, const DecAttnArgs& a, hipStream_t st);
// split-q decode step: ONE launch = the attention workgroups (q from the row a preceding launch wrote) + the k|v projection g (C unused: its rows leave as
// granules); g = rows [nh D, (nh + 2 nkv) D) of the fused q|k|v weight with the RMSNorm fused as in the plain projection.  See decode_attn.hip.
bool decode_kv_attn_applies(int dtype, int D, const GemvArgs& g);
int decode_kv_attn_resident_slots(int dtype, int D, int K);      // occupancy x CUs of the split-q launch on the current device
void launch_decode_kv_attn(int dtype, int D, const DecAttnArgs& a, const GemvArgs& g, hipStream_t st);

// ---- kernel-only timing (in-situ profile) -----------------------------------------------------------------------------------------------------
// A profiling scope that brackets exactly ONE instrumented launch arms this thread-local slot; the launcher then uses hipExtLaunchKernelGGL with the
// scope's start / stop events, which are stamped at the kernel's own begin and end (what rocprofv3 reports), instead of stream markers around the
// launch that add their own packets' latency to every figure.
struct KernelTimer { hipEvent_t e0 = nullptr, e1 = nullptr; bool used = false; };
extern thread_local KernelTimer* g_kernel_timer;
#define LMX_LAUNCH(kern, grid, block, smem, st, ...)                                                                              \
    do {                                                                                                                          \
        ::lmx::KernelTimer* kt_ = ::lmx::g_kernel_timer;                                                                          \
        if (kt_ && !kt_->used) { kt_->used = true; hipExtLaunchKernelGGL(kern, grid, block, smem, st, kt_->e0, kt_->e1, 0, __VA_ARGS__); } \
        else hipLaunchKernelGGL(kern, grid, block, smem, st, __VA_ARGS__);                                                        \
    } while (0)

// ---- row / elementwise kernels (elementwise.hip) --------------------------------------------------------------
void launch_rmsnorm(int dtype, const void* x, const void* w, void* y, int rows, int H, int ldx, int ldy, float eps, hipStream_t st);
void launch_layernorm(int dtype, const void* x, const void* w, const void* b, void* y, int rows, int H, int ldx, int ldy, float eps, hipStream_t st);

struct RopeKvArgs {
    void* QKV;            // [T][qkv_stride]: q heads | k heads | v heads. q is rotated in place.
    void* K;              // K cache (this layer)
    void* VT;             // Vᵀ cache (this layer)
    const float* cos_sin; // [max_pos][D] fp32: cos[0..D/2) then sin[0..D/2) per position; null => no rotation (CLIP)
    const int* pos_ptr;   // device pointer to position of token 0, or null
    int pos0;
    int T;
    int qkv_stride;
    int n_heads, n_kv_heads, s_max;
    int k_inplace = 0;    // the rotated k ALSO replaces the k columns of QKV (the training step's backward reads k from there: no gathered copy)
};
void launch_rope_kv(int dtype, int D, const RopeKvArgs& a, hipStream_t st);

// token-embedding gather + image-feature splice: out[r] = src[r] >= 0 ? table[src[r]] : src[r] == -1 ? 0 : feats[-2 - src[r]]
void launch_gather_embed(int dtype, const int* src, const void* table, const void* feats, void* out, int rows, int H, hipStream_t st);
// same, but the single source index is read from *tok_ptr (int64, decode loop)
void launch_gather_token(int dtype, const int64_t* tok_ptr, const void* table, void* out, int H, int vocab, hipStream_t st);

// CLIP patch extraction: pixels [N,3,S,S] (dtype) -> patches [N*P, kpad], k = c*ps*ps + py*ps + px, zero padded
void launch_im2col(int dtype, const void* pix, void* out, int N, int S, int ps, int kpad, hipStream_t st);
// CLIP embeddings + pre-LN: y[n][t] = LN( round(t==0 ? cls : patch[n][t-1]) + pos[t] )
void launch_clip_embed_ln(int dtype, const void* patch, const void* cls, const void* pos, const void* w, const void* b,
                          void* y, int N, int P, int Dm, float eps, hipStream_t st);
// copy rows [N][tok0 .. tok0+P) of [N][Ttot][Dm] -> [N*P][Dm]  (feature_select 'patch' drops CLS)
void launch_copy_rows(int dtype, const void* src, void* dst, int N, int Ttot, int tok0, int P, int Dm, hipStream_t st);

// greedy sampling: token = argmax(logits[0..V)), first index wins ties (torch.argmax semantics)
void launch_argmax(int dtype, const void* logits, int V, int64_t* out_tok, hipStream_t st);
// decode-loop bookkeeping: *len += 1 ; tokens_out[*n_out++] = *tok
void launch_advance(int* len_ptr, const int64_t* tok_ptr, int64_t* out_tokens, int* n_out_ptr, int max_out, hipStream_t st);

// ---- decode batch (continuous batching): per-sequence device state, one table row per member -------------------------
// how a sequence picks its next token on the device: temperature <= 0 -> greedy argmax, else temperature / top-k / top-p draw
struct SampleParams { float temperature = 0.f; float top_p = 1.f; int top_k = 0; uint32_t seed_lo = 0, seed_hi = 0; };
// device-side stop rule of a sequence (lmx_seq_set_stop): the reference checks every new token on the host — `eos_token_id` inside HF generate and
// KeywordsStoppingCriteria's id test `(output_ids[0, -len(kw):] == kw).all()` (llava/mm_utils.py:94-107), one D2H copy per token.  Here the pick kernel
// applies the same two id rules to the token it just appended and sets `done`; from then on the sequence's picks do not advance it (no token is logged,
// the position stays), so steps that were queued ahead — a read-ahead window, the second stage of the scheduler's pipeline, other members of a decode
// batch still running — produce nothing past the stop and the host needs no per-token round trip to know where it was.
constexpr int STOP_MAX_EOS = 4, STOP_MAX_KW = 4, STOP_MAX_KW_LEN = 8;
struct StopSpec { int done; int n_eos; int n_kw; int pad; int64_t eos[STOP_MAX_EOS]; int kw_len[STOP_MAX_KW]; int64_t kw[STOP_MAX_KW][STOP_MAX_KW_LEN]; };
struct SeqStateRef { int* len; int* n_out; int64_t* tok; int64_t* log; int log_cap; int pad; SampleParams sample; StopSpec* stop; };
// the two id rules on a sequence whose n_after generated ids (the last one = t) sit in log[0 .. n_after): t is an EOS id | the last len(kw) ids equal kw
__device__ __forceinline__ bool stop_rule_fires(const StopSpec* sp, int64_t t, const int64_t* log, int n_after, int log_cap) {
    bool hit = false;
    for (int e = 0; e < sp->n_eos; ++e) hit = hit || (t == sp->eos[e]);
    for (int k = 0; k < sp->n_kw && !hit; ++k) {
        const int L = sp->kw_len[k];
        if (L <= 0 || n_after < L || !log || n_after > log_cap) continue;
        bool m = true;
        for (int j = 0; j < L; ++j) m = m && (log[n_after - L + j] == sp->kw[k][j]);
        hit = m;
    }
    return hit;
}
// out[i] = table[*tab[i].tok]
void launch_gather_tokens_batch(int dtype, const SeqStateRef* tab, int n, const void* table, void* out, int H, int vocab, hipStream_t st);
// per member i: *tok = argmax(logits[i]) (first index wins) or a draw (tab[i].sample), *len += 1, log[n_out++] = tok; ids_out[i] = tok; then the stop
// rule (tab[i].stop, may be null).  A member whose rule already fired is left untouched and reports ids_out[i] = -1
// V = ids considered (the real vocabulary), ld = row pitch of `logits` (the padded vocabulary)
// then (embed != null) h_out[i] = embed[tok]: the next step's input row.  State comes from the device table `tab` (n members)
// or from `single` (host pointer, passed by value; n must be 1).
void launch_argmax_advance_batch(int dtype, const void* logits, int V, int ld, const SeqStateRef* tab, const SeqStateRef* single, int n, int64_t* ids_out,
                                 const void* embed, void* h_out, int H, hipStream_t st);

// ---- one-shot peer-to-peer all-reduce for decode-sized messages (p2p.hip) ---------------------------------------------
constexpr int P2P_MAX_ROWS = 32;     // rows ([H] each) per launch = largest decode batch that goes through the skinny path
constexpr int P2P_MAX_WORLD = 8;
struct P2PLaunch {
    void* buf;                       // [rows][H], in place
    int H, world, rank, rows;
    uint32_t seq;                    // 1, 2, 3, ... identical on every rank for the same all-reduce
    void* peer[P2P_MAX_WORLD];       // exchange buffers as mapped in this process
    int last_len = 0;                // elements of the last row when the message is not a whole number of [H] rows (0 = H)
    // LlamaRMSNorm of the summed rows inside the same launch (round 6; whole rows only): x_out [rows][H] <- rmsnorm(sum) * norm_w — the workgroup that owns a row has it
    // complete, so the decode batch's next linear needs no rmsnorm launch (65 of a tensor-parallel batched step's 227 launches).  rmsnorm_kernel's arithmetic, bit for bit
    const void* norm_w = nullptr; void* x_out = nullptr; float eps = 0.f;
};
void launch_p2p_allreduce(int dtype, const P2PLaunch& l, hipStream_t st);
size_t p2p_buffer_bytes(int world, int H, int es);
size_t p2p_flags_offset(int world, int H, int es);

// ---- two-shot peer-to-peer all-reduce for PREFILL-sized messages (p2p.hip, round 4): reduce-scatter + all-gather over all links at once ------------------
// xGMI is point-to-point: a ring keeps one link per direction busy (2 (W - 1) / W x message over ONE 153 GB/s link: ~114 us for the 8.9 MB of a 1087-row
// 7B prefill at W = 8).  Here rank r owns chunk r of the message: every rank pushes chunk p of its partial sums straight into rank p's exchange buffer (W - 1
// links busy at once), the owner adds the W versions in rank order and pushes the finished chunk to every peer.  One launch, message / W per link and
// phase (~2 x 7.3 us of wire time at W = 8), deterministic and bit-identical on every rank (each chunk is summed once, by its owner).
// The region lives behind the one-shot region of the same exchange buffer (p2p_big_offset); geometry is a function of (world, max_count, es).
constexpr int P2P_BIG_SLICE = 4096;   // elements per slice: one workgroup walks slice s of every chunk; flags are per (phase, source rank, slice)
struct P2PBigGeom {
    size_t chunk_max;                 // elements per rank chunk at max_count (multiple of P2P_BIG_SLICE)
    int n_slices_max;
    size_t rs_off, ag_off, flags_off; // byte offsets of parity 0 inside the exchange buffer
    size_t rs_par, ag_par, flags_par; // byte strides between the two parities
    size_t end;                       // first byte behind the region
};
P2PBigGeom p2p_big_geometry(int world, size_t max_count, int es, size_t base_off);
struct P2PBigLaunch {
    void* buf; size_t count;          // elements, in place
    int world, rank;
    uint32_t seq;                     // as P2PLaunch::seq (own counter)
    size_t status_off;                // the one-shot region's status word (shared)
    void* peer[P2P_MAX_WORLD];
    P2PBigGeom g;
};
void launch_p2p_allreduce_big(int dtype, const P2PBigLaunch& l, hipStream_t st);

// temperature -> top-k -> top-p -> multinomial draw of one row (sampling.hip); RNG = Philox(seed, *offset_ptr) unless u32_override
// (host pointer, tests) is given; keep_out (device, [V], debug) receives the survivor mask
void launch_beam_topk(int dtype, const void* logits, int ld, int V, int rows, const float* beam_scores, int K, float* out_scores, int* out_ids, hipStream_t st);
// beam-sample step (sampling.hip): per row the K largest (score / T + Gumbel) keys among the ids `keep` allows (null = all), with their scores and ids
void launch_beam_gumbel_topk(int dtype, const void* logits, int ld, int V, int rows, const uint8_t* keep, const float* beam_scores, float temperature, uint64_t seed,
                             uint32_t counter0, int K, float* out_keys, float* out_scores, int* out_ids, hipStream_t st);
void launch_sample(int dtype, const void* logits, int V, const SampleParams& p, const int* offset_ptr, int64_t* out_tok,
                   const uint32_t* u32_override, uint8_t* keep_out, hipStream_t st);

// CLIP image preprocessing on the device (preprocess.hip): uint8 RGB [H][W][3] -> [3][size][size] of the model dtype.
// Returns the scratch bytes needed; does nothing else when `scratch` is null or too small.
int preprocess_coeffs(int in_size, int out_size, int o0, int on, int* bounds_out, int* kk_out, int kk_cap);
size_t launch_preprocess(int dtype, const uint8_t* rgb, int H, int W, int size, int pad_to_square, const float* mean, const float* std,
                         void* out, void* scratch, size_t scratch_bytes, hipStream_t st);

// ---- training-step slices (train.hip): parity-first kernels, see the file header for the reference lines ------------------
// shifted cross-entropy over logits [B*T][ld] vs labels [B][T] (label of position t = labels[t + 1]); lse / row_loss: [B*(T-1)] floats;
// out2[0] = mean loss over the counted positions, out2[1] = their number
void launch_ce_loss_fwd(int dtype, const void* logits, int ld, const int64_t* labels, int B, int Tlen, int V, int64_t ignore, float* lse,
                        float* row_loss, float* out2, hipStream_t st);
void launch_ce_loss_bwd(int dtype, const void* logits, int ld, const int64_t* labels, int B, int Tlen, int V, int64_t ignore, const float* lse,
                        const float* out2, float grad, void* dlogits, int ldd, hipStream_t st);
// dx [rows][H]; dw [H] fp32 (or null); inv_scratch: rows floats
void launch_rmsnorm_bwd(int dtype, const void* x, const void* w, const void* dy, const void* residual_or_null, void* dx, float* dw, float* inv_scratch, int rows, int H,
                        float eps, hipStream_t st);
void launch_swiglu_bwd(int dtype, const void* g, const void* u, const void* dact, void* dg, void* du, size_t n, hipStream_t st);
void launch_rope_bwd(int dtype, const void* dy, void* dx, const float* cos_sin, int pos0, int Tn, int heads, int D, int ld, hipStream_t st);
void launch_transpose(int dtype, const void* src, int ld, int rows, int cols, void* dst, int ldd, hipStream_t st);
// gemm8t.hip: out[M][N] = sum_r dy[r][m] * x[r][n] with both operands in their forward layout (no transposes); gemm_wgrad_direct_ok = the shapes it takes
bool gemm_wgrad_direct_ok(int dtype, int M, int N, int rows, int lddy, int ldx, int ldo);
void launch_gemm_wgrad(int dtype, const void* dy, int lddy, const void* x, int ldx, int rows, int M, int N, void* out, int ldo, hipStream_t st);
// matrix-core form for 16-bit models (attn_bwd.hip; LMX_ATTN_BWD_MFMA=0 keeps the VALU kernels): launch_attn_bwd takes it when attn_bwd_mfma_wanted()
bool attn_bwd_mfma_wanted(int dtype, int D);
void launch_attn_bwd_mfma(int dtype, int D, const void* q, const void* k, const void* v, const void* dO, void* dq, void* dk, void* dv, int Tn, int heads,
                          int kv_heads, int ldq, int ldk, int ldo, float scale, hipStream_t st, const void* out = nullptr, int ldout = 0, const float* lse_in = nullptr,
                          int lse_stride = 0);
void launch_attn_bwd(int dtype, int D, const void* q, const void* k, const void* v, const void* dO, void* dq, float* dk32, float* dv32, void* dk, void* dv,
                     int Tn, int heads, int kv_heads, int ldq, int ldk, int ldo, float scale, hipStream_t st);
void launch_elementwise(int dtype, int op, const void* a, const void* b, void* out, size_t n, hipStream_t st);       // 0 swiglu, 1 gelu, 2 gelu_bwd, 3 add
void launch_cast_f32(int dtype, const float* src, void* dst, size_t n, hipStream_t st);
void launch_col_sum(int dtype, const void* dy, int ld, int rows, int cols, float* out, hipStream_t st);
void launch_embed_bwd(int dtype, const int* src, const void* d, float* dtable, void* dfeats, int rows, int H, hipStream_t st);
void launch_sumsq(int dtype, const void* x, size_t n, float* acc, hipStream_t st);
void launch_adamw(int dtype, void* param, const void* grad, float* master, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, float wd,
                  int step, const float* gnorm_sq, float max_norm, hipStream_t st);

// weight re-layout helpers (launch_interleave_half lives in engine.h)
void launch_cast(int src_dtype, int dst_dtype, const void* src, void* dst, size_t n, hipStream_t st);

}  // namespace lmx
