// Host-side engine: owns weights (kernel-optimal layouts), KV caches, workspaces, the device-resident decode state and the RCCL
// communicator.  One engine per process/GPU (tensor-parallel rank).  See include/llava_mi355x.h for the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <map>
#include <atomic>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/llava_mi355x.h"
#include "common.h"
#include "kernels.h"

namespace lmx {

// Synchronous zero fill on a private non-blocking stream: never touches the legacy stream (other threads may be
// stream-capturing or launching there) and returns only when the bytes are zero.
void zero_fill(void* p, size_t n);

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
    void ensure(size_t n, bool zero = false) {
        if (n <= bytes) return;
        release();
        LMX_CHECK_HIP(hipMalloc(&p, n));
        bytes = n;
        if (zero) zero_fill(p, n);
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct DecLayerW {
    void *wqkv = nullptr, *wo = nullptr, *wgu = nullptr, *wd = nullptr, *ln1 = nullptr, *ln2 = nullptr;
    void *sw_qkv = nullptr, *sw_o = nullptr, *sw_gu = nullptr, *sw_d = nullptr;   // fragment-order copies for the decode-batch kernel
};
struct VisLayerW {
    void *wqkv = nullptr, *bqkv = nullptr, *wo = nullptr, *bo = nullptr, *fc1 = nullptr, *b1 = nullptr, *fc2 = nullptr, *b2 = nullptr;
    void *ln1w = nullptr, *ln1b = nullptr, *ln2w = nullptr, *ln2b = nullptr;
};

struct Seq;

struct Model {
    lmx_config cfg{};
    int es = 2;                 // element size
    // decoder geometry (TP-local)
    int H = 0, I_l = 0, I_sh = 0, nh_l = 0, nkv_l = 0, D = 0, V = 0, L = 0, qkv_n = 0, s_max = 0;
    int V_l = 0, v_off = 0;     // lm_head rows held by this rank (vocabulary-parallel head under TP) and the id of its first row
    int Vr = 0;                 // ids a pick may return (real vocabulary, lmx_set_vocab_limit); V is the padded row pitch of logits / embeddings
    // vision geometry
    int Dv = 0, Fv = 0, v_run = 0, P = 0, Tv = 0, kpad = 0, spad = 0, vD = 0, out_tokens = 0;

    std::vector<DevBuf> pool;   // owns every weight allocation
    void* embed = nullptr; void* final_norm = nullptr; void* lm_head = nullptr; void* sw_lm_head = nullptr;
    bool batch_weights_ready = false;
    void ensure_batch_weights(hipStream_t st);      // second, fragment-order copy of the decoder weights (first Batch pays for it)
    std::vector<DecLayerW> dec;
    void *v_cls = nullptr, *v_patch_w = nullptr, *v_pos = nullptr, *v_pre_w = nullptr, *v_pre_b = nullptr;
    std::vector<VisLayerW> vis;
    std::vector<void*> proj_w, proj_b;
    float* rope = nullptr; int rope_npos = 0;
    std::set<std::string> loaded;

    std::mutex mu;              // guards lazy allocations + the shared vision workspace
    // Sequence pool: a finished request's KV caches / workspaces (≈ 1 GB at 7B, s_max 2048) are handed to the next request instead of
    // hipFree + hipMalloc — hipFree synchronises the whole device, which stalls every other request of a busy worker.
    std::mutex pool_mu; std::vector<struct lmx_seq*> seq_pool; size_t seq_pool_max = 64;
    size_t seq_pool_bytes = 0, seq_pool_bytes_max = (size_t)96 << 30;
    DevBuf vws;                 // vision workspace
    int vws_images = 0;
    DevBuf vkc, vvt;            // CLIP K / Vᵀ scratch (zero padded)

    ncclComm_t comm = nullptr;
    // one-shot P2P all-reduce for decode-sized messages (p2p.hip): exchange buffer of this rank + the peers' mapped through HIP IPC
    void* p2p_local = nullptr; void* p2p_peer[P2P_MAX_WORLD] = {}; bool p2p_on = false, p2p_all = false; uint32_t p2p_seq = 0;
    // two-shot (reduce-scatter + all-gather) region of the same exchange buffer for prefill-sized messages (p2p.hip: p2p_allreduce_big_kernel)
    P2PBigGeom p2p_big{}; size_t p2p_big_max_count = 0; uint32_t p2p_big_seq = 0; bool p2p_big_on = true;
    // two-shot kernel for prefill-sized messages: from 4 ranks on (all W - 1 links busy at once; with two ranks a ring IS the direct exchange and the kernel's
    // ~50 us of protocol only costs), or whenever there is no RCCL communicator to fall back to (ranks sharing one GPU: tests, dry runs)
    bool p2p_big_usable(size_t count) const {
        return p2p_on && p2p_big_on && (cfg.tp_world >= 4 || comm == nullptr) && p2p_big_max_count > 0 && count <= p2p_big_max_count && count % 8 == 0;
    }
    void p2p_local_handle(void* out64);
    void p2p_connect(const void* handles);
    int p2p_status(hipStream_t st);
    hipStream_t comm_stream = nullptr;      // prefill all-reduces run here, overlapped with the other row half's compute
    bool tp_overlap = true, tp_overlap_force = false;   // LMX_TP_OVERLAP=0 serialises them on the launch stream, =2 pipelines every chunk >= 256 rows
    void ensure_comm_stream();
    // test hook: replaces ncclAllReduce (lets two ranks of a TP group live in one process / on one GPU in tests)
    typedef void (*AllReduceHook)(void* buf, uint64_t count, int dtype, void* stream, void* ctx);
    AllReduceHook ar_hook = nullptr; void* ar_ctx = nullptr;

    // ---- in-situ kernel timing (HIP events on the launch stream; forces eager decode while enabled) --------------
    // Scopes only RECORD an event pair on the launch stream (no host sync, so kernels keep running back-to-back);
    // elapsed times are resolved when the profile is read.
    bool prof_on = false;
    struct ProfAcc { double ms = 0; long count = 0; };
    struct ProfRec { const char* name; hipEvent_t e0, e1; };
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;          // recycled events
    std::map<std::string, ProfAcc> prof_resolve();
    hipEvent_t prof_event();
    hipEvent_t vws_done = nullptr; hipStream_t vws_stream = nullptr;

    explicit Model(const lmx_config& c);
    ~Model();
    void* alloc_weight(size_t bytes, bool zero = false);
    void load_weight(const std::string& name, const void* src, int dtype, int ndim, const int64_t* shape, hipStream_t st);
    void finalize();
    void set_rope(const float* host, int n_pos);
    void allreduce(void* buf, size_t count, hipStream_t st);
    // all-reduce of [rows][H] partial sums AND LlamaRMSNorm(norm_w) of the sums into x_out, in one launch where the one-shot peer-to-peer kernel serves the message
    // (returns true); otherwise the plain all-reduce (returns false: the caller still has to normalise)
    bool allreduce_norm(void* buf, int rows, const void* norm_w, void* x_out, hipStream_t st);
    // logits [rows][V] of the vocabulary-parallel head: every rank filled columns [v_off, v_off + V_l) of a zeroed buffer; the sum over
    // ranks IS the all-gather (exact in any dtype), so the decode-sized case rides the one-shot P2P all-reduce
    void gather_logits(void* logits, int rows, hipStream_t st);

    void encode_images(const void* pixels, int n, void* feats, hipStream_t st, bool tower_only = false);
    void gather_embeds(const int32_t* src, int rows, const void* feats, void* out, hipStream_t st);
    // hidden != null: also writes `output_hidden_states`' tuple, [L + 1][T][H] in the model dtype: the input rows of every decoder layer, then the final norm's output
    // (LlamaModel's all_hidden_states, HF5:models/llama/modeling_llama.py:367-418); the rows then run without the two-half tensor-parallel pipeline
    // attn != null: also writes `output_attentions`' tuple, [L][heads][T][len + T] in the model dtype (eager rounding points, kernels.h: AttnProbsArgs); single process only
    void prefill(Seq* s, const void* embeds, int T, int chunk, void* logits, bool logits_all, bool greedy, hipStream_t st, void* hidden = nullptr, void* attn = nullptr);
    void decode(Seq* s, int64_t token, int n_steps, void* logits, bool greedy, hipStream_t st);
    void prefill_multi(Seq* const* seqs, const void* const* embeds, const int* Ts, int n, int block_rows, bool greedy, hipStream_t st);
    void decode_step_launch(Seq* s, hipStream_t st, int64_t* id_out = nullptr);      // id_out (device): this step's picked id, -1 after a device-side stop
    void seq_copy(Seq* dst, const Seq* src, hipStream_t st);      // dst := src's context (KV of the first src->len positions, length): beam reordering
    void decode_batch(struct Batch* b, Seq* const* seqs, int n, const int64_t* tokens, int n_steps, void* logits, bool greedy, int64_t* ids_out_host, hipStream_t st,
                      bool sync_ids = true);
    // ---- options a caller may flip on a live model (lmx_model_set_option; defaults from the environment at creation) ----------------------------------
    // fuse_rope: RoPE + KV append in the prefill's q|k|v GEMM epilogue (gemm8p.hip qkv_rope_epilogue) instead of a rope_kv launch — bit-identical, faster
    // vis_pack:  CLIP K / V^T pack in the tower's q|k|v GEMM epilogue instead of a pack launch — bit-identical, launch count
    // splitq:    the decode step's q|k|v projection as q-launch + (k|v projection ∥ attention) launch (decode_attn.hip) — bit-identical
    bool opt_fuse_rope = true, opt_vis_pack = true, opt_splitq = true;
    // Sequences in existence (lmx_seq_create .. lmx_seq_destroy).  The split-q launch parks one waiting workgroup per head until its own launch's projection has
    // run; launches of DIFFERENT sequences (request threads on their own streams) may be resident together, so the form is only taken while all sequences'
    // waiters together (live_seqs x heads) fill at most half of the chip's ~1024 workgroup slots — the projection's workgroups then always find a slot
    std::atomic<int> live_seqs{0};
    int splitq_slots = 0;              // workgroups of the split-q launch the device holds at once (occupancy x CUs, queried on first use); -1: the kernel does not apply
    bool splitq_allowed() {
        if (!opt_splitq) return false;
        if (splitq_slots == 0) {
            const bool ok = (cfg.dtype == kBF16 || cfg.dtype == kF16) && (D == 64 || D == 128) && H <= 8192;
            splitq_slots = ok ? decode_kv_attn_resident_slots(cfg.dtype, D, H) : -1;
        }
        return splitq_slots > 0 && (long)live_seqs.load(std::memory_order_relaxed) * nh_l * 2 <= splitq_slots;
    }
    // fault injection (lmx_model_set_option "debug_splitq_timeout" = n): the next n split-q launches publish their k | v granules under a wrong tag, so their
    // waiters run into the bounded wait's timeout — what a preempted / starved projection would look like (tests/test_decode_splitq_gpu.py)
    std::atomic<int> debug_splitq_timeout{0};
};

struct Seq {
    Model* m = nullptr;
    uint64_t uid = 0;           // never reused (a Batch caches per-member device pointers keyed by this)
    hipStream_t last_stream = nullptr; bool used = false;   // stream of the most recent work on this sequence (pool reuse waits for it)
    hipEvent_t ev_idle = nullptr;
    SampleParams samp;          // temperature <= 0: greedy; set by lmx_seq_set_sampling (bumps uid so a Batch re-reads it)
    DevBuf kc, vt;              // [L][nkv_l][s_max][D] and [L][nkv_l][D][s_max]
    size_t layer_stride = 0;    // bytes per layer in each cache
    int len = 0;                // host mirror of *d_len.  A device-side stop freezes *d_len while steps queued ahead still count here: resync_len()
    void resync_len(int dev_len, int dev_done) { if (dev_done != 0 && dev_len >= 0 && dev_len < len) len = dev_len; }   // called wherever the host has just read both
    DevBuf state;               // device: [0] int len, [1] int n_out, then int64 tok at byte 8, token log from byte 16
    int* d_len = nullptr; int* d_nout = nullptr; int64_t* d_tok = nullptr; int64_t* d_log = nullptr;
    DevBuf stopbuf; StopSpec* d_stop = nullptr;          // device-side stop rule (lmx_seq_set_stop); all zero = no rule
    int log_cap = 0;
    DevBuf pws;  int pws_tokens = 0;   // prefill workspace
    DevBuf skw;                        // split-K partial tiles of the ping-pong GEMM (o_proj, down_proj of a prefill)
    DevBuf dws;                        // decode workspace
    void *d_h = nullptr, *d_qkv = nullptr, *d_attn = nullptr, *d_act = nullptr, *d_logits = nullptr; float* d_aws = nullptr; int* d_cnt = nullptr;
    int n_split = 8;
    DevBuf kv_gran;                    // split-q decode step: {bits, tag} granules [2 nkv_l D] of the newest key / value (zeroed at creation: tag 0 is never used)
    unsigned attn_tag = 1;             // tag of the next split-q launch on kv_gran
    // the decode attention's bounded in-launch waits (decode_attn.hip: a merger waiting for the other chunks' arrivals, or for the k | v granules of its own
    // launch): a host-mapped word PER SEQUENCE, raised by the kernel when a wait times out.  check_wait_status() reports it once — for this sequence only —,
    // clears it together with the arrival tickets and switches the model to the three-launch form
    unsigned* h_status = nullptr; unsigned* d_status = nullptr;
    void check_wait_status(hipStream_t st);
    hipEvent_t ev_c[2] = {nullptr, nullptr}, ev_r[2] = {nullptr, nullptr};   // TP prefill pipeline: compute-done / reduce-done per row half
    void ensure_events();
    explicit Seq(Model* mm);
    ~Seq();
};

// Decode batch (continuous batching): workspaces for up to `cap` sequences stepping together + the device tables that
// point the batched kernels at each member's own KV cache / position / token state.  One Batch is driven by one thread.
struct Batch {
    Model* m = nullptr;
    int cap = 0;
    DevBuf ws;                         // h | x (normed) | qkv | attn | act | logits, each [cap][...]
    void *h = nullptr, *x = nullptr, *qkv = nullptr, *attn = nullptr, *act = nullptr, *logits = nullptr;
    DevBuf tab;                        // device: DecodeFusedSeq[L][cap] then SeqStateRef[cap]
    DecodeFusedSeq* d_attn_tab = nullptr; SeqStateRef* d_state_tab = nullptr;
    std::vector<uint64_t> members;     // Seq::uid of what the device tables currently describe
    std::vector<char> host_tab;
    DevBuf ids; int ids_steps = 0;     // [ids_steps][cap] int64: greedy picks of the chained steps of the last call
    DevBuf sk_scratch, sk_cnt;         // skinny kernel's K slices across workgroups (narrow layers): fp32 partial tiles of one launch, one ticket per 64-row tile (zeroed once)
    Batch(Model* mm, int capacity);
    void bind(Seq* const* seqs, int n, hipStream_t st);
};

// RAII timing scope: records an event pair around the launches issued inside the scope and accumulates the elapsed
// time under `name`.  No-op unless Model::prof_on.
struct ProfScope {
    Model* m; const char* name; hipStream_t st; hipEvent_t e0 = nullptr;
    ProfScope(Model* mm, const char* n, hipStream_t s) : m(mm), name(n), st(s) {
        if (m->prof_on && name) { e0 = m->prof_event(); (void)hipEventRecord(e0, st); }
    }
    ~ProfScope() {
        if (!e0) return;
        hipEvent_t e1 = m->prof_event();
        (void)hipEventRecord(e1, st);
        m->prof_recs.push_back({name, e0, e1});
    }
};
// Scope around exactly ONE GEMM / GEMV launch: the launcher stamps the events at the kernel's own start and stop (kernels.h LMX_LAUNCH), so the figure is
// the kernel's duration as rocprofv3 reports it.  If no instrumented launch happens inside, nothing is recorded.
struct ProfKernelScope {
    Model* m; const char* name; KernelTimer kt; KernelTimer* prev = nullptr;
    ProfKernelScope(Model* mm, const char* n) : m(mm), name(n) {
        if (m->prof_on) { kt.e0 = m->prof_event(); kt.e1 = m->prof_event(); prev = g_kernel_timer; g_kernel_timer = &kt; }
    }
    ~ProfKernelScope() {
        if (!kt.e0) return;
        g_kernel_timer = prev;
        if (kt.used) m->prof_recs.push_back({name, kt.e0, kt.e1});
        else { m->prof_pool.push_back(kt.e0); m->prof_pool.push_back(kt.e1); }
    }
};
#define LMX_PROF(name) ::lmx::ProfScope _prof_scope_##__LINE__(this, name, st)
// a scope around a collective: recorded only when the call does something (no tensor parallelism = no launch = an empty event pair that would show up as
// ~4.6 us of phantom time per call in the kernel breakdown)
#define LMX_PROF_AR(name) ::lmx::ProfScope _prof_scope_##__LINE__(this, (cfg.tp_world > 1 || comm) ? name : nullptr, st)
#define LMX_PROF_K(name) ::lmx::ProfKernelScope _prof_kscope_##__LINE__(this, name)

uint64_t next_seq_uid();

// splice.cpp
int splice_plan(const int64_t* input_ids, const uint8_t* attention_mask, const int64_t* labels, int B, int L,
                int tokens_per_image, const int32_t* slot_rows, int n_image_slots, int max_len, int left_pad,
                int32_t* out_T, int32_t* src, uint8_t* out_mask, int64_t* out_pos, int64_t* out_labels);

// elementwise.hip (state helpers)
void launch_set_state(int* len_ptr, int len, int64_t* tok_ptr, int64_t tok, int set_tok, int* nout_ptr, int set_nout, hipStream_t st);
void launch_set_stop(StopSpec* dst, const StopSpec& v, hipStream_t st);
void launch_hash128(const void* base, size_t bytes_per_item, int items, uint64_t* out_dev, hipStream_t st);      // out_dev[2 * item + {0, 1}]      // *dst = v on the stream (v travels as a kernel argument; re-arms done = v.done)
void launch_interleave_half(int dtype, const void* src, void* dst, int I, int K, int half, hipStream_t st);
void launch_log_token(const int64_t* tok_ptr, int64_t* log, int* n_out_ptr, int max_out, StopSpec* stop, hipStream_t st);

}  // namespace lmx
