// extern "C" surface of libllava_mi355x.so — see include/llava_mi355x.h for the contract of every entry point.
// Exceptions never cross the boundary: they become a non-zero status + thread-local message.
#include <cstring>
#include <map>
#include <tuple>
#include <mutex>
#include <exception>

#include "engine.h"

using namespace lmx;

struct lmx_model { Model impl; explicit lmx_model(const lmx_config& c) : impl(c) {} };
struct lmx_seq { Seq impl; explicit lmx_seq(Model* m) : impl(m) {} };
struct lmx_batch { Batch impl; lmx_batch(Model* m, int cap) : impl(m, cap) {} };

#define LMX_API_BEGIN try {
#define LMX_API_END                                                              \
    return 0;                                                                    \
    } catch (const lmx::Error& e) { lmx::set_last_error(e.msg); return 1; }      \
    catch (const std::exception& e) { lmx::set_last_error(e.what()); return 2; } \
    catch (...) { lmx::set_last_error("unknown native error"); return 3; }

static hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

extern "C" {

const char* lmx_last_error(void) { return lmx::get_last_error(); }
int lmx_abi_version(void) { return LMX_ABI_VERSION; }

int lmx_create(const lmx_config* cfg, lmx_model** out) {
    LMX_API_BEGIN
    LMX_REQUIRE(cfg && out, "lmx_create: null argument");
    int n = 0;
    LMX_CHECK_HIP(hipGetDeviceCount(&n));
    LMX_REQUIRE(n > 0, "no HIP device visible");
    *out = new lmx_model(*cfg);
    LMX_API_END
}
int lmx_destroy(lmx_model* m) {
    LMX_API_BEGIN
    if (m) {
        (void)hipDeviceSynchronize();
        for (lmx_seq* s : m->impl.seq_pool) delete s;
        m->impl.seq_pool.clear();
    }
    delete m;
    LMX_API_END
}
int lmx_load_weight(lmx_model* m, const char* name, const void* dev_ptr, int32_t dtype, int32_t ndim, const int64_t* shape, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(m && name && shape, "lmx_load_weight: null argument");
    m->impl.load_weight(name, dev_ptr, dtype, ndim, shape, S(stream));
    LMX_API_END
}
int lmx_finalize_weights(lmx_model* m) {
    LMX_API_BEGIN
    LMX_REQUIRE(m, "null model");
    m->impl.finalize();
    LMX_API_END
}
int lmx_set_rope_table(lmx_model* m, const float* host_cos_sin, int32_t n_pos) {
    LMX_API_BEGIN
    LMX_REQUIRE(m, "null model");
    m->impl.set_rope(host_cos_sin, n_pos);
    LMX_API_END
}

int lmx_tp_unique_id(void* out_128_bytes) {
    LMX_API_BEGIN
    LMX_REQUIRE(out_128_bytes, "null output");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) throw Error{std::string("ncclGetUniqueId failed: ") + ncclGetErrorString(r)};
    memcpy(out_128_bytes, &id, sizeof(id));
    LMX_API_END
}
int lmx_tp_init(lmx_model* m, const void* unique_id_128_bytes) {
    LMX_API_BEGIN
    LMX_REQUIRE(m && unique_id_128_bytes, "null argument");
    if (m->impl.comm) { (void)ncclCommDestroy(m->impl.comm); m->impl.comm = nullptr; }
    ncclUniqueId id;
    memcpy(&id, unique_id_128_bytes, sizeof(id));
    const ncclResult_t r = ncclCommInitRank(&m->impl.comm, m->impl.cfg.tp_world, id, m->impl.cfg.tp_rank);
    if (r != ncclSuccess) throw Error{std::string("ncclCommInitRank failed: ") + ncclGetErrorString(r)};
    LMX_API_END
}

int lmx_tp_comm_ranks(lmx_model* m) {
    if (!m || !m->impl.comm) return 0;
    int n = 0;
    return ncclCommCount(m->impl.comm, &n) == ncclSuccess ? n : -1;
}

int lmx_tp_p2p_local_handle(lmx_model* m, void* out_64_bytes) {
    LMX_API_BEGIN
    LMX_REQUIRE(m && out_64_bytes, "null argument");
    m->impl.p2p_local_handle(out_64_bytes);
    LMX_API_END
}
int lmx_tp_p2p_connect(lmx_model* m, const void* handles_world_x_64_bytes) {
    LMX_API_BEGIN
    LMX_REQUIRE(m && handles_world_x_64_bytes, "null argument");
    m->impl.p2p_connect(handles_world_x_64_bytes);
    LMX_API_END
}
int lmx_tp_p2p_enable(lmx_model* m, int32_t on) {
    LMX_API_BEGIN
    LMX_REQUIRE(m, "null model");
    LMX_REQUIRE(!on || m->impl.p2p_peer[m->impl.cfg.tp_rank] != nullptr, "p2p all-reduce is not connected");
    m->impl.p2p_on = on != 0;
    LMX_API_END
}
int lmx_tp_p2p_status(lmx_model* m, void* stream) {
    try { return m ? m->impl.p2p_status(S(stream)) : -1; } catch (...) { return -1; }
}
int lmx_op_allreduce(lmx_model* m, void* buf_dev, uint64_t count, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(m && buf_dev, "null argument");
    m->impl.allreduce(buf_dev, (size_t)count, S(stream));
    LMX_API_END
}

int lmx_tp_set_allreduce_hook(lmx_model* m, void (*hook)(void*, uint64_t, int32_t, void*, void*), void* ctx) {
    LMX_API_BEGIN
    LMX_REQUIRE(m, "null model");
    m->impl.ar_hook = reinterpret_cast<Model::AllReduceHook>(hook);
    m->impl.ar_ctx = ctx;
    LMX_API_END
}

int lmx_encode_images(lmx_model* m, const void* pixels_dev, int32_t n_images, void* feats_dev, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(m, "null model");
    m->impl.encode_images(pixels_dev, n_images, feats_dev, S(stream));
    LMX_API_END
}
int lmx_vision_tower(lmx_model* m, const void* pixels_dev, int32_t n_images, void* feats_dev, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(m, "null model");
    m->impl.encode_images(pixels_dev, n_images, feats_dev, S(stream), /*tower_only=*/true);
    LMX_API_END
}
int lmx_preprocess_coeffs(int32_t in_size, int32_t out_size, int32_t first_out, int32_t n_out, int32_t* bounds_out, int32_t* coeffs_out, int32_t coeffs_cap) {
    try { return preprocess_coeffs(in_size, out_size, first_out, n_out, bounds_out, coeffs_out, coeffs_cap); } catch (...) { return -1; }
}
int lmx_preprocess_image(lmx_model* m, const uint8_t* rgb_dev, int32_t H, int32_t W, int32_t out_dtype, int32_t pad_to_square,
                         const float* mean3, const float* std3, void* pixels_out_dev, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(m && rgb_dev && mean3 && std3 && pixels_out_dev, "null argument");
    LMX_REQUIRE(m->impl.cfg.v_layers > 0, "no vision tower configured");
    const int size = m->impl.cfg.v_image_size;
    const size_t need = launch_preprocess(out_dtype, rgb_dev, H, W, size, pad_to_square, mean3, std3, pixels_out_dev, nullptr, 0, S(stream));
    DevBuf scratch;                                   // per call: request threads preprocess concurrently
    scratch.ensure(need);
    launch_preprocess(out_dtype, rgb_dev, H, W, size, pad_to_square, mean3, std3, pixels_out_dev, scratch.p, need, S(stream));
    LMX_API_END
}
int lmx_tokens_per_image(const lmx_model* m) { return m ? m->impl.out_tokens : -1; }
int lmx_set_vocab_limit(lmx_model* m, int32_t n_real) {
    LMX_API_BEGIN
    LMX_REQUIRE(m, "null model");
    LMX_REQUIRE(n_real >= 1 && n_real <= m->impl.V, "vocab limit must be in [1, vocab_size]");
    m->impl.Vr = n_real;
    LMX_API_END
}

int lmx_splice_plan(const int64_t* input_ids, const uint8_t* attention_mask, const int64_t* labels,
                    int32_t B, int32_t L, int32_t tokens_per_image, const int32_t* slot_rows, int32_t n_image_slots,
                    int32_t max_len, int32_t left_pad,
                    int32_t* out_T, int32_t* src, uint8_t* out_mask, int64_t* out_pos, int64_t* out_labels) {
    LMX_API_BEGIN
    splice_plan(input_ids, attention_mask, labels, B, L, tokens_per_image, slot_rows, n_image_slots, max_len, left_pad,
                out_T, src, out_mask, out_pos, out_labels);
    LMX_API_END
}
int lmx_gather_embeds(lmx_model* m, const int32_t* src_dev, int32_t rows, const void* feats_dev, void* embeds_dev, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(m && src_dev && embeds_dev, "null argument");
    m->impl.gather_embeds(src_dev, rows, feats_dev, embeds_dev, S(stream));
    LMX_API_END
}

int lmx_seq_create(lmx_model* m, lmx_seq** out) {
    LMX_API_BEGIN
    LMX_REQUIRE(m && out, "null argument");
    lmx_seq* s = nullptr;
    {
        std::lock_guard<std::mutex> lk(m->impl.pool_mu);
        if (!m->impl.seq_pool.empty()) {
            s = m->impl.seq_pool.back(); m->impl.seq_pool.pop_back();
            m->impl.seq_pool_bytes -= s->impl.kc.bytes + s->impl.vt.bytes + s->impl.dws.bytes + s->impl.pws.bytes;
        }
    }
    if (s) {
        // recycled sequence: wait for the last work that touched it (event recorded when it was released), then start clean.
        // Stale K / V rows stay (finite values; every kernel masks by length), only the state words are cleared.
        if (s->impl.ev_idle) LMX_CHECK_HIP(hipEventSynchronize(s->impl.ev_idle));
        s->impl.len = 0;
        s->impl.samp = SampleParams{};
        s->impl.uid = next_seq_uid();
        zero_fill(s->impl.state.p, 16);
        zero_fill(s->impl.stopbuf.p, sizeof(StopSpec));
    } else {
        s = new lmx_seq(&m->impl);
    }
    m->impl.live_seqs.fetch_add(1, std::memory_order_relaxed);
    *out = s;
    LMX_API_END
}
int lmx_seq_destroy(lmx_seq* s) {
    LMX_API_BEGIN
    if (!s) return 0;
    Model* m = s->impl.m;
    if (m) m->live_seqs.fetch_sub(1, std::memory_order_relaxed);
    bool pooled = false;
    bool idle_known = true;
    if (m && s->impl.used) {
        // the stream of the last work may be gone by now (a scheduler's stream after disable_batching): then drain the device and
        // free the sequence instead of pooling it — never leak ~1-2 GB of KV cache behind a thrown HIP error
        if (!s->impl.ev_idle && hipEventCreateWithFlags(&s->impl.ev_idle, hipEventDisableTiming) != hipSuccess) idle_known = false;
        if (idle_known && hipEventRecord(s->impl.ev_idle, s->impl.last_stream) != hipSuccess) idle_known = false;
        if (!idle_known) { (void)hipGetLastError(); (void)hipDeviceSynchronize(); }
    }
    // A packed prefill grows the FIRST sequence's workspace to the whole packed block (8 x 1087 rows at 7B: ~0.6 GB).  Such a block does not go to the
    // pool with the sequence: above the size of a ~2.5k-row prefill it is freed here (hipFree waits for the device), so pooled sequences stay KV-sized.
    constexpr size_t kPwsKeepMax = (size_t)192 << 20;
    if (m && idle_known && s->impl.pws.bytes > kPwsKeepMax) { s->impl.pws.release(); s->impl.pws_tokens = 0; }
    if (m && idle_known) {
        std::lock_guard<std::mutex> lk(m->pool_mu);
        const size_t bytes = s->impl.kc.bytes + s->impl.vt.bytes + s->impl.dws.bytes + s->impl.pws.bytes;
        if (m->seq_pool.size() < m->seq_pool_max && m->seq_pool_bytes + bytes <= m->seq_pool_bytes_max) {
            m->seq_pool.push_back(s); m->seq_pool_bytes += bytes; pooled = true;
        }
    }
    if (!pooled) delete s;
    LMX_API_END
}
int lmx_seq_set_stop(lmx_seq* s, const int64_t* eos_ids, int32_t n_eos, const int64_t* kw_flat, const int32_t* kw_lens, int32_t n_kw, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(s, "null sequence");
    LMX_REQUIRE(n_eos >= 0 && n_eos <= STOP_MAX_EOS && (n_eos == 0 || eos_ids), "set_stop: at most 4 end-of-sequence ids");
    LMX_REQUIRE(n_kw >= 0 && n_kw <= STOP_MAX_KW && (n_kw == 0 || (kw_flat && kw_lens)), "set_stop: at most 4 keyword id sequences");
    StopSpec h{};
    h.n_eos = n_eos; h.n_kw = n_kw;
    for (int i = 0; i < n_eos; ++i) h.eos[i] = eos_ids[i];
    size_t off = 0;
    for (int k = 0; k < n_kw; ++k) {
        LMX_REQUIRE(kw_lens[k] >= 1 && kw_lens[k] <= STOP_MAX_KW_LEN, "set_stop: a keyword is 1..8 ids");
        h.kw_len[k] = kw_lens[k];
        for (int j = 0; j < kw_lens[k]; ++j) h.kw[k][j] = kw_flat[off + (size_t)j];
        off += (size_t)kw_lens[k];
    }
    // ordered with the sequence's work on `stream`; the rule is a kernel argument, so the caller's stream is not drained (a scheduler admitting a
    // request between two decode steps keeps its pipeline) and done is re-armed (= 0) by the same store
    launch_set_stop(s->impl.d_stop, h, S(stream));
    LMX_API_END
}
int lmx_seq_stopped(lmx_seq* s, int32_t* out, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(s && out, "null argument");
    int done = 0, st2[2] = {0, 0};
    LMX_CHECK_HIP(hipMemcpyAsync(&done, &s->impl.d_stop->done, sizeof(int), hipMemcpyDeviceToHost, S(stream)));
    LMX_CHECK_HIP(hipMemcpyAsync(st2, s->impl.d_len, sizeof(int), hipMemcpyDeviceToHost, S(stream)));
    LMX_CHECK_HIP(hipStreamSynchronize(S(stream)));
    s->impl.resync_len(st2[0], done);           // steps queued past the stop advanced the host mirror only: the device position is the sequence's position
    *out = done != 0;
    LMX_API_END
}
int lmx_seq_set_sampling(lmx_seq* s, float temperature, float top_p, int32_t top_k, uint64_t seed) {
    LMX_API_BEGIN
    LMX_REQUIRE(s, "null sequence");
    LMX_REQUIRE(temperature <= 0.f || (top_p > 0.f && top_p <= 1.f), "top_p must be in (0, 1]");
    LMX_REQUIRE(top_k >= 0, "top_k must be >= 0 (0 = off)");
    s->impl.samp = SampleParams{temperature > 0.f ? temperature : 0.f, top_p, top_k, (uint32_t)seed, (uint32_t)(seed >> 32)};
    s->impl.uid = next_seq_uid();          // a decode batch holding this sequence rebuilds its table entry
    LMX_API_END
}
int lmx_op_sample(int32_t dtype, const void* logits_dev, int32_t V, float temperature, float top_p, int32_t top_k, uint64_t seed,
                  const int32_t* offset_dev, const uint32_t* u32_override_host, int64_t* out_tok_dev, uint8_t* keep_out_dev, void* stream) {
    LMX_API_BEGIN
    launch_sample(dtype, logits_dev, V, SampleParams{temperature, top_p, top_k, (uint32_t)seed, (uint32_t)(seed >> 32)}, offset_dev, out_tok_dev,
                  u32_override_host, keep_out_dev, S(stream));
    LMX_API_END
}
int lmx_seq_reset(lmx_seq* s) {
    LMX_API_BEGIN
    LMX_REQUIRE(s, "null sequence");
    s->impl.len = 0;
    zero_fill(s->impl.state.p, 16);       // caller guarantees no work on this sequence is in flight
    zero_fill(&s->impl.d_stop->done, sizeof(int));      // a sequence that stopped by its id rule is live again (the rule itself stays until the next set_stop)
    LMX_API_END
}
int lmx_seq_length(const lmx_seq* s) { return s ? s->impl.len : -1; }

int lmx_prefill(lmx_model* m, lmx_seq* s, const void* embeds_dev, int32_t T, int32_t chunk,
                void* logits_dev, int32_t logits_all, int32_t greedy, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(m && s, "null argument");
    m->impl.prefill(&s->impl, embeds_dev, T, chunk, logits_dev, logits_all != 0, greedy != 0, S(stream));
    LMX_API_END
}
int lmx_prefill_hidden(lmx_model* m, lmx_seq* s, const void* embeds_dev, int32_t T, void* logits_dev, int32_t logits_all, void* hidden_dev, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(m && s && hidden_dev, "null argument");
    m->impl.prefill(&s->impl, embeds_dev, T, 0, logits_dev, logits_all != 0, false, S(stream), hidden_dev);
    LMX_API_END
}
int lmx_prefill_outputs(lmx_model* m, lmx_seq* s, const void* embeds_dev, int32_t T, void* logits_dev, int32_t logits_all, void* hidden_dev, void* attentions_dev,
                        void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(m && s && (hidden_dev || attentions_dev), "null argument");
    m->impl.prefill(&s->impl, embeds_dev, T, 0, logits_dev, logits_all != 0, false, S(stream), hidden_dev, attentions_dev);
    LMX_API_END
}
int lmx_decode(lmx_model* m, lmx_seq* s, int64_t token, int32_t n_steps, void* logits_dev, int32_t greedy, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(m && s, "null argument");
    m->impl.decode(&s->impl, token, n_steps, logits_dev, greedy != 0, S(stream));
    LMX_API_END
}
int lmx_batch_create(lmx_model* m, int32_t capacity, lmx_batch** out) {
    LMX_API_BEGIN
    LMX_REQUIRE(m && out, "null argument");
    *out = new lmx_batch(&m->impl, capacity);
    if (capacity > 1) m->impl.ensure_batch_weights(nullptr);     // one-time fragment-order weight copy: pay for it here, not in the first step
    LMX_API_END
}
int lmx_batch_destroy(lmx_batch* b) {
    LMX_API_BEGIN
    if (b) { (void)hipDeviceSynchronize(); delete b; }
    LMX_API_END
}
int lmx_decode_batch(lmx_model* m, lmx_batch* b, lmx_seq* const* seqs, int32_t n, const int64_t* tokens_host, int32_t n_steps,
                     void* logits_dev, int32_t greedy, int64_t* ids_out_host, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(m && b && seqs, "null argument");
    LMX_REQUIRE(n >= 1 && n <= b->impl.cap, "decode_batch: number of sequences exceeds the batch capacity");
    std::vector<Seq*> ss((size_t)n);
    for (int i = 0; i < n; ++i) { LMX_REQUIRE(seqs[i] != nullptr, "null sequence"); ss[(size_t)i] = &seqs[i]->impl; }
    m->impl.decode_batch(&b->impl, ss.data(), n, tokens_host, n_steps, logits_dev, greedy != 0, ids_out_host, S(stream));
    LMX_API_END
}
int lmx_decode_batch_async(lmx_model* m, lmx_batch* b, lmx_seq* const* seqs, int32_t n, int32_t n_steps, int64_t* ids_out_pinned_host, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(m && b && seqs && ids_out_pinned_host, "null argument");
    LMX_REQUIRE(n >= 1 && n <= b->impl.cap, "decode_batch: number of sequences exceeds the batch capacity");
    std::vector<Seq*> ss((size_t)n);
    for (int i = 0; i < n; ++i) { LMX_REQUIRE(seqs[i] != nullptr, "null sequence"); ss[(size_t)i] = &seqs[i]->impl; }
    m->impl.decode_batch(&b->impl, ss.data(), n, nullptr, n_steps, nullptr, true, ids_out_pinned_host, S(stream), /*sync_ids=*/false);
    LMX_API_END
}
int lmx_seq_read_tokens(lmx_seq* s, int64_t* host_out, int32_t max_n, int32_t* n_out, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(s && host_out && n_out, "null argument");
    int n = 0, dev_len = -1, done = 0;
    LMX_CHECK_HIP(hipMemcpyAsync(&n, s->impl.d_nout, sizeof(int), hipMemcpyDeviceToHost, S(stream)));
    LMX_CHECK_HIP(hipMemcpyAsync(&dev_len, s->impl.d_len, sizeof(int), hipMemcpyDeviceToHost, S(stream)));
    LMX_CHECK_HIP(hipMemcpyAsync(&done, &s->impl.d_stop->done, sizeof(int), hipMemcpyDeviceToHost, S(stream)));
    LMX_CHECK_HIP(hipStreamSynchronize(S(stream)));
    s->impl.resync_len(dev_len, done);
    s->impl.check_wait_status(S(stream));        // a bounded in-launch wait of this sequence's decode attention timed out: say so (once, for this sequence)
    if (n > s->impl.log_cap) n = s->impl.log_cap;
    if (n > max_n) n = max_n;
    if (n > 0) {
        LMX_CHECK_HIP(hipMemcpyAsync(host_out, s->impl.d_log, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, S(stream)));
        LMX_CHECK_HIP(hipStreamSynchronize(S(stream)));
    }
    *n_out = n;
    LMX_API_END
}

int lmx_profile_enable(lmx_model* m, int32_t on) {
    LMX_API_BEGIN
    LMX_REQUIRE(m, "null model");
    LMX_CHECK_HIP(hipDeviceSynchronize());
    m->impl.prof_on = on != 0;
    if (on) (void)m->impl.prof_resolve();      // drop stale records
    LMX_API_END
}
int lmx_model_set_option(lmx_model* m, const char* key, int32_t value) {
    LMX_API_BEGIN
    LMX_REQUIRE(m && key, "null argument");
    const std::string k(key);
    if (k == "fuse_rope") m->impl.opt_fuse_rope = value != 0;
    else if (k == "vis_pack") m->impl.opt_vis_pack = value != 0;
    else if (k == "decode_splitq") m->impl.opt_splitq = value != 0;
    else if (k == "debug_splitq_timeout") m->impl.debug_splitq_timeout = value;
    else throw Error{"lmx_model_set_option: unknown option '" + k + "'"};
    LMX_API_END
}
int lmx_profile_read(lmx_model* m, char* names_buf, int32_t names_cap, double* ms, int64_t* counts, int32_t max_n, int32_t* n_out) {
    LMX_API_BEGIN
    LMX_REQUIRE(m && names_buf && ms && counts && n_out, "null argument");
    int n = 0; size_t off = 0;
    const auto prof = m->impl.prof_resolve();
    for (const auto& kv : prof) {
        if (n >= max_n || off + kv.first.size() + 1 >= (size_t)names_cap) break;
        memcpy(names_buf + off, kv.first.c_str(), kv.first.size()); off += kv.first.size(); names_buf[off++] = '\n';
        ms[n] = kv.second.ms; counts[n] = kv.second.count; ++n;
    }
    names_buf[off] = 0;
    *n_out = n;
    LMX_API_END
}

// ---- single-op entry points ---------------------------------------------------------------------------------------
// decode-batch kernel on the fragment-order copy of w (tests: variant 21 = copy made on every call; microbenchmarks: 22 = copy cached per
// (pointer, N, K) — only valid while that weight tensor is alive and unchanged); 20 = the [N][K] layout as it is
static void skinny_op(int32_t dtype, GemmArgs g, int32_t variant, hipStream_t st) {
    // 23 / 24 = 22 / 21 with the K-slices-across-workgroups form's scratch (narrow layers, more than 8 rows: skinny_kslices)
    if (variant == 23 || variant == 24) {
        static std::mutex mu2;
        static DevBuf sk_scratch, sk_cnt;
        std::lock_guard<std::mutex> lk(mu2);
        const size_t need = skinny_scratch_bytes(g.N), ncnt = (size_t)(g.N / 64 + 1) * sizeof(int);
        if (sk_scratch.bytes < need || sk_cnt.bytes < ncnt) { LMX_CHECK_HIP(hipDeviceSynchronize()); sk_scratch.ensure(need); sk_cnt.ensure(ncnt, true); }
        g.skw = sk_scratch.p; g.sk_cnt = sk_cnt.as<int>();
        variant = variant == 23 ? 22 : 21;
    }
    if (variant == 21 || variant == 22) {
        static std::mutex mu;
        static std::map<std::tuple<const void*, int, int>, DevBuf> cache;
        static DevBuf scratch;
        std::lock_guard<std::mutex> lk(mu);
        const size_t bytes = skinny_swizzled_bytes(g.N, g.K, (int)dtype_size(dtype));
        if (variant == 21) {
            if (scratch.bytes < bytes) { LMX_CHECK_HIP(hipDeviceSynchronize()); scratch.ensure(bytes); }
            launch_skinny_swizzle(dtype, g.W, g.ldw, scratch.p, g.N, g.K, st);
            g.Wsw = scratch.p;
        } else {
            DevBuf& d = cache[std::make_tuple(g.W, (int)g.N, (int)g.K)];
            if (!d.p) {
                d.ensure(bytes);
                launch_skinny_swizzle(dtype, g.W, g.ldw, d.p, g.N, g.K, st);
                LMX_CHECK_HIP(hipStreamSynchronize(st));
            }
            g.Wsw = d.p;
        }
        launch_skinny_gemm(dtype, g, st);
        return;
    }
    launch_skinny_gemm(dtype, g, st);
}
int lmx_op_gemm(int32_t dtype, const void* x, const void* w, void* c, const void* bias, const void* residual,
                int32_t M, int32_t N, int32_t K, int32_t ldx, int32_t ldw, int32_t ldc, int32_t ldr, int32_t act, int32_t variant, void* stream) {
    LMX_API_BEGIN
    if (variant >= 21 && variant <= 24) {
        skinny_op(dtype, GemmArgs{x, w, c, bias, residual, M, N, K, ldx, ldw, ldc, ldr, act}, variant, S(stream));
        return 0;
    }
    launch_gemm(dtype, GemmArgs{x, w, c, bias, residual, M, N, K, ldx, ldw, ldc, ldr, act}, variant, S(stream));
    LMX_API_END
}
int lmx_op_gemv(int32_t dtype, const void* x, const void* w, void* c, const void* bias, const void* residual, const void* norm_w, float eps,
                int32_t MB, int32_t N, int32_t K, int32_t ldx, int32_t ldw, int32_t ldc, int32_t ldr, int32_t act, void* stream) {
    LMX_API_BEGIN
    launch_gemv(dtype, GemvArgs{x, w, c, bias, residual, norm_w, eps, N, K, ldx, ldw, ldc, ldr, act}, MB, S(stream));
    LMX_API_END
}
int lmx_op_rmsnorm(int32_t dtype, const void* x, const void* w, void* y, int32_t rows, int32_t H, float eps, void* stream) {
    LMX_API_BEGIN
    launch_rmsnorm(dtype, x, w, y, rows, H, H, H, eps, S(stream));
    LMX_API_END
}
int lmx_op_layernorm(int32_t dtype, const void* x, const void* w, const void* b, void* y, int32_t rows, int32_t H, float eps, void* stream) {
    LMX_API_BEGIN
    launch_layernorm(dtype, x, w, b, y, rows, H, H, H, eps, S(stream));
    LMX_API_END
}
int lmx_op_rope_kv(int32_t dtype, int32_t head_dim, void* qkv, void* kcache, void* vtcache, const float* cos_sin_dev, int32_t pos0,
                   int32_t T, int32_t n_heads, int32_t n_kv_heads, int32_t s_max, void* stream) {
    LMX_API_BEGIN
    launch_rope_kv(dtype, head_dim, RopeKvArgs{qkv, kcache, vtcache, cos_sin_dev, nullptr, pos0, T, (n_heads + 2 * n_kv_heads) * head_dim, n_heads, n_kv_heads, s_max}, S(stream));
    LMX_API_END
}
int lmx_op_rope_kv_rows(int32_t dtype, int32_t head_dim, void* qkv, void* kcache, void* vtcache, const float* cos_sin_dev, int32_t pos0, int32_t T, int32_t n_heads,
                        int32_t n_kv_heads, int32_t s_max, void* stream) {
    LMX_API_BEGIN
    RopeKvArgs a{qkv, kcache, vtcache, cos_sin_dev, nullptr, pos0, T, (n_heads + 2 * n_kv_heads) * head_dim, n_heads, n_kv_heads, s_max};
    a.k_inplace = 1;
    launch_rope_kv(dtype, head_dim, a, S(stream));
    LMX_API_END
}
int lmx_op_gemm_qkv_rope(int32_t dtype, int32_t head_dim, const void* x, const void* w, void* qkv, void* kcache, void* vtcache, const float* cos_sin_dev,
                         int32_t pos0, int32_t T, int32_t K, int32_t n_heads, int32_t n_kv_heads, int32_t s_max, void* stream) {
    LMX_API_BEGIN
    const int N = (n_heads + 2 * n_kv_heads) * head_dim;
    LMX_REQUIRE(gemm_fuses_qkv(dtype, T, K, head_dim, n_heads, n_kv_heads, pos0, s_max, false), "gemm_qkv_rope: this shape does not take the fused launch");
    GemmArgs g{x, w, qkv, nullptr, nullptr, T, N, K, K, K, N, 0, kActNone};
    g.qf_rope = cos_sin_dev; g.qf_kc = kcache; g.qf_vt = vtcache; g.qf_pos0 = pos0; g.qf_nh = n_heads; g.qf_nkv = n_kv_heads; g.qf_smax = s_max; g.qf_D = head_dim;
    launch_gemm(dtype, g, 0, S(stream));
    LMX_API_END
}
int lmx_op_flash_attn(int32_t dtype, int32_t head_dim, const void* q, void* o, const void* kcache, const void* vtcache,
                      int32_t q_len, int32_t kv_len, int32_t q_pos0, int32_t q_stride, int32_t o_stride,
                      int32_t n_heads, int32_t n_kv_heads, int32_t s_max, float scale, int32_t causal, void* stream) {
    LMX_API_BEGIN
    launch_flash_prefill(dtype, head_dim, FlashArgs{q, o, kcache, vtcache, q_len, kv_len, q_pos0, q_stride, o_stride, n_heads, n_kv_heads, s_max, scale, causal}, S(stream));
    LMX_API_END
}
int lmx_op_flash_attn_lse(int32_t dtype, int32_t head_dim, const void* q, void* o, const void* kcache, const void* vtcache, int32_t q_len, int32_t kv_len, int32_t q_pos0,
                          int32_t q_stride, int32_t o_stride, int32_t n_heads, int32_t n_kv_heads, int32_t s_max, float scale, int32_t causal, float* lse,
                          int32_t lse_stride, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(lse && lse_stride >= q_len, "flash_attn_lse: lse must hold n_heads rows of at least q_len floats");
    FlashArgs a{q, o, kcache, vtcache, q_len, kv_len, q_pos0, q_stride, o_stride, n_heads, n_kv_heads, s_max, scale, causal};
    a.lse = lse; a.lse_stride = lse_stride;
    launch_flash_prefill(dtype, head_dim, a, S(stream));
    LMX_API_END
}
int lmx_op_attn_bwd_lse(int32_t dtype, int32_t head_dim, const void* q, const void* k, const void* v, const void* out, const void* d_out, const float* lse,
                        int32_t lse_stride, void* dq, void* dk, void* dv, int32_t T, int32_t heads, int32_t kv_heads, int32_t ldq, int32_t ldk, int32_t ldo,
                        int32_t ldout, float scale, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(out && lse, "attn_bwd_lse: the forward's output and log-sum-exp are inputs");
    LMX_REQUIRE(attn_bwd_mfma_wanted(dtype, head_dim) && ldq % 8 == 0 && ldk % 8 == 0 && ldo % 8 == 0, "attn_bwd_lse: 16-bit dtypes, head_dim 64 / 128, 16-byte aligned rows");
    launch_attn_bwd_mfma(dtype, head_dim, q, k, v, d_out, dq, dk, dv, T, heads, kv_heads, ldq, ldk, ldo, scale, S(stream), out, ldout, lse, lse_stride);
    LMX_API_END
}
int lmx_op_decode_attn(int32_t dtype, int32_t head_dim, const void* q, void* o, const void* kcache, const void* vtcache,
                       int32_t n_rows, int32_t pos0, int32_t kv_total, int32_t causal, int32_t q_stride, int32_t o_stride,
                       int32_t n_heads, int32_t n_kv_heads, int32_t s_max, int32_t n_split, float scale, void* ws_dev, void* stream) {
    LMX_API_BEGIN
    launch_decode_attn(dtype, head_dim, DecodeAttnArgs{q, o, kcache, vtcache, nullptr, pos0, n_rows, kv_total, causal, q_stride, o_stride,
                                                       n_heads, n_kv_heads, s_max, n_split, scale, static_cast<float*>(ws_dev)}, S(stream));
    LMX_API_END
}
int lmx_op_decode_fused(int32_t dtype, int32_t head_dim, const void* qkv, void* kcache, void* vtcache, const float* cos_sin_dev, const int32_t* pos_dev,
                        int32_t n_heads, int32_t n_kv_heads, int32_t s_max, float scale, void* ws_dev, int32_t* counters_dev, void* out, int32_t debug_mode, void* stream) {
    LMX_API_BEGIN
    DecodeFusedArgs a{qkv, kcache, vtcache, cos_sin_dev, pos_dev, n_heads, n_kv_heads, s_max, (s_max + 127) / 128, scale, static_cast<float*>(ws_dev), counters_dev, out};
    a.debug_mode = debug_mode;
    launch_decode_fused(dtype, head_dim, a, S(stream));
    LMX_API_END
}
int lmx_op_decode_attn_step(int32_t dtype, int32_t head_dim, void* qkv, void* kcache, void* vtcache, const float* cos_sin_dev, int32_t pos,
                            int32_t n_heads, int32_t n_kv_heads, int32_t s_max, float scale, void* ws_dev, int32_t* counters_dev, void* out, void* stream) {
    LMX_API_BEGIN
    DecAttnArgs a{};
    a.qkv = qkv; a.attn = out; a.kc = kcache; a.vt = vtcache; a.rope = cos_sin_dev; a.aws = static_cast<float*>(ws_dev); a.cnt = counters_dev;
    a.pos = pos; a.n_split = pos / 128 + 1; a.nh = n_heads; a.nkv = n_kv_heads; a.s_max = s_max; a.scale = scale;
    launch_decode_attn_step(dtype, head_dim, a, S(stream));
    LMX_API_END
}
int lmx_op_decode_kv_attn(int32_t dtype, int32_t head_dim, void* qkv, const void* x, const void* w_kv, const void* norm_w, float eps, int32_t K, int32_t ldw,
                          void* kcache, void* vtcache, const float* cos_sin_dev, int32_t pos, int32_t n_heads, int32_t n_kv_heads, int32_t s_max, float scale,
                          void* ws_dev, int32_t* counters_dev, void* granules_dev, uint32_t tag, void* out, void* timeline_dev, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(qkv && x && w_kv && granules_dev && out, "null argument");
    DecAttnArgs a{};
    a.qkv = qkv; a.attn = out; a.kc = kcache; a.vt = vtcache; a.rope = cos_sin_dev; a.aws = static_cast<float*>(ws_dev); a.cnt = counters_dev;
    a.pos = pos; a.n_split = pos / 128 + 1; a.nh = n_heads; a.nkv = n_kv_heads; a.s_max = s_max; a.scale = scale;
    a.kv_gran = static_cast<unsigned long long*>(granules_dev); a.tag = tag; a.ts = static_cast<unsigned long long*>(timeline_dev);
    const int kv_n = 2 * n_kv_heads * head_dim;
    const GemvArgs g{x, w_kv, nullptr, nullptr, nullptr, norm_w, eps, kv_n, K, K, ldw, kv_n, 0, kActNone};
    launch_decode_kv_attn(dtype, head_dim, a, g, S(stream));
    LMX_API_END
}
int lmx_op_decode_attn_batch(int32_t dtype, int32_t head_dim, const void* qkv, int32_t qkv_stride, void* const* kcaches, void* const* vtcaches,
                             const int32_t* const* pos_devs, void* const* ws_devs, int32_t* const* counters_devs, int32_t n_seq, const float* cos_sin_dev,
                             int32_t n_heads, int32_t n_kv_heads, int32_t s_max, int32_t n_split, float scale, void* tab_dev, void* out, int32_t o_stride,
                             void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(n_seq >= 1 && kcaches && vtcaches && pos_devs && tab_dev, "decode_attn_batch: bad arguments");
    std::vector<DecodeFusedSeq> tab((size_t)n_seq);
    for (int i = 0; i < n_seq; ++i)
        tab[(size_t)i] = DecodeFusedSeq{kcaches[i], vtcaches[i], pos_devs[i], ws_devs ? static_cast<float*>(ws_devs[i]) : nullptr, counters_devs ? counters_devs[i] : nullptr};
    LMX_CHECK_HIP(hipMemcpyAsync(tab_dev, tab.data(), sizeof(DecodeFusedSeq) * (size_t)n_seq, hipMemcpyHostToDevice, S(stream)));
    LMX_CHECK_HIP(hipStreamSynchronize(S(stream)));              // `tab` is pageable host memory: done with it before it goes out of scope
    DecodeFusedArgs a{qkv, nullptr, nullptr, cos_sin_dev, nullptr, n_heads, n_kv_heads, s_max, n_split, scale, nullptr, nullptr, out};
    a.tab = static_cast<const DecodeFusedSeq*>(tab_dev); a.n_seq = n_seq; a.qkv_stride = qkv_stride; a.o_stride = o_stride;
    launch_decode_fused(dtype, head_dim, a, S(stream));
    LMX_API_END
}
size_t lmx_op_decode_attn_batch_tab_bytes(int32_t n_seq) { return sizeof(DecodeFusedSeq) * (size_t)(n_seq > 0 ? n_seq : 0); }
size_t lmx_op_decode_attn_ws_bytes(int32_t n_rows, int32_t n_heads, int32_t n_split, int32_t head_dim) {
    return decode_attn_ws_floats(n_rows, n_heads, n_split, head_dim) * sizeof(float);
}
int lmx_op_argmax(int32_t dtype, const void* logits, int32_t V, int64_t* out_tok_dev, void* stream) {
    LMX_API_BEGIN
    launch_argmax(dtype, logits, V, out_tok_dev, S(stream));
    LMX_API_END
}
int lmx_op_im2col(int32_t dtype, const void* pixels, void* out, int32_t N, int32_t S_, int32_t patch, int32_t kpad, void* stream) {
    LMX_API_BEGIN
    launch_im2col(dtype, pixels, out, N, S_, patch, kpad, S(stream));
    LMX_API_END
}

// ---- training-step slices (train.hip) ---------------------------------------------------------------------------------------------
int lmx_op_ce_loss(int32_t dtype, const void* logits, int32_t ld, const int64_t* labels, int32_t B, int32_t T, int32_t V, int64_t ignore_index,
                   float* lse_scratch, float* row_loss_scratch, float* out_loss_count, float grad, void* dlogits_or_null, int32_t ldd, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(logits && labels && lse_scratch && row_loss_scratch && out_loss_count, "null argument");
    launch_ce_loss_fwd(dtype, logits, ld, labels, B, T, V, ignore_index, lse_scratch, row_loss_scratch, out_loss_count, S(stream));
    if (dlogits_or_null)
        launch_ce_loss_bwd(dtype, logits, ld, labels, B, T, V, ignore_index, lse_scratch, out_loss_count, grad, dlogits_or_null, ldd, S(stream));
    LMX_API_END
}
int lmx_op_rmsnorm_bwd(int32_t dtype, const void* x, const void* w, const void* dy, void* dx, float* dw_or_null, float* inv_scratch, int32_t rows,
                       int32_t H, float eps, void* stream) {
    LMX_API_BEGIN
    launch_rmsnorm_bwd(dtype, x, w, dy, nullptr, dx, dw_or_null, inv_scratch, rows, H, eps, S(stream));
    LMX_API_END
}
int lmx_op_rmsnorm_bwd_add(int32_t dtype, const void* x, const void* w, const void* dy, const void* residual, void* dx, float* dw_or_null, float* inv_scratch,
                           int32_t rows, int32_t H, float eps, void* stream) {
    LMX_API_BEGIN
    launch_rmsnorm_bwd(dtype, x, w, dy, residual, dx, dw_or_null, inv_scratch, rows, H, eps, S(stream));
    LMX_API_END
}
int lmx_op_swiglu_bwd(int32_t dtype, const void* gate, const void* up, const void* dact, void* dgate, void* dup, int64_t n, void* stream) {
    LMX_API_BEGIN
    launch_swiglu_bwd(dtype, gate, up, dact, dgate, dup, (size_t)n, S(stream));
    LMX_API_END
}
int lmx_op_rope_bwd(int32_t dtype, const void* dy, void* dx, const float* cos_sin_dev, int32_t pos0, int32_t T, int32_t heads, int32_t head_dim, int32_t ld,
                    void* stream) {
    LMX_API_BEGIN
    launch_rope_bwd(dtype, dy, dx, cos_sin_dev, pos0, T, heads, head_dim, ld, S(stream));
    LMX_API_END
}
int lmx_op_transpose(int32_t dtype, const void* src, int32_t ld, int32_t rows, int32_t cols, void* dst, int32_t ldd, void* stream) {
    LMX_API_BEGIN
    launch_transpose(dtype, src, ld, rows, cols, dst, ldd, S(stream));
    LMX_API_END
}
int lmx_op_gemm_wgrad(int32_t dtype, const void* dy, int32_t lddy, const void* x, int32_t ldx, int32_t rows, int32_t out_features, int32_t in_features, void* out,
                      int32_t ldo, void* stream) {
    LMX_API_BEGIN
    launch_gemm_wgrad(dtype, dy, lddy, x, ldx, rows, out_features, in_features, out, ldo, S(stream));
    LMX_API_END
}
int lmx_op_gemm_wgrad_supported(int32_t dtype, int32_t lddy, int32_t ldx, int32_t rows, int32_t out_features, int32_t in_features, int32_t ldo) {
    return gemm_wgrad_direct_ok(dtype, out_features, in_features, rows, lddy, ldx, ldo) ? 1 : 0;
}
int lmx_op_attn_bwd(int32_t dtype, int32_t head_dim, const void* q, const void* k, const void* v, const void* d_out, void* dq, float* dk32_scratch,
                    float* dv32_scratch, void* dk, void* dv, int32_t T, int32_t heads, int32_t kv_heads, int32_t ldq, int32_t ldk, int32_t ldo, float scale,
                    void* stream) {
    LMX_API_BEGIN
    launch_attn_bwd(dtype, head_dim, q, k, v, d_out, dq, dk32_scratch, dv32_scratch, dk, dv, T, heads, kv_heads, ldq, ldk, ldo, scale, S(stream));
    LMX_API_END
}
int lmx_op_elementwise(int32_t dtype, int32_t op, const void* a, const void* b, void* out, int64_t n, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(a && out && (b || op == 1), "null argument");
    launch_elementwise(dtype, op, a, b, out, (size_t)n, S(stream));
    LMX_API_END
}
int lmx_op_cast_f32(int32_t dtype, const float* src, void* dst, int64_t n, void* stream) {
    LMX_API_BEGIN
    launch_cast_f32(dtype, src, dst, (size_t)n, S(stream));
    LMX_API_END
}
int lmx_op_col_sum(int32_t dtype, const void* dy, int32_t ld, int32_t rows, int32_t cols, float* out, void* stream) {
    LMX_API_BEGIN
    launch_col_sum(dtype, dy, ld, rows, cols, out, S(stream));
    LMX_API_END
}
int lmx_op_gather_embed(int32_t dtype, const int32_t* src_dev, const void* table, const void* feats_or_null, void* out, int32_t rows, int32_t H, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(src_dev && table && out, "null argument");
    launch_gather_embed(dtype, src_dev, table, feats_or_null, out, rows, H, S(stream));
    LMX_API_END
}
int lmx_op_embed_bwd(int32_t dtype, const int32_t* src_dev, const void* d_embeds, float* dtable_or_null, void* dfeats_or_null, int32_t rows, int32_t H, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(src_dev && d_embeds, "null argument");
    launch_embed_bwd(dtype, src_dev, d_embeds, dtable_or_null, dfeats_or_null, rows, H, S(stream));
    LMX_API_END
}
int lmx_op_sumsq(int32_t dtype, const void* x, int64_t n, float* acc, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(acc, "null argument");
    launch_sumsq(dtype, x, (size_t)n, acc, S(stream));
    LMX_API_END
}
int lmx_op_adamw(int32_t dtype, void* param, const void* grad, float* master, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int32_t step, const float* gnorm_sq_or_null, float max_grad_norm, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(param && grad && master && exp_avg && exp_avg_sq, "null argument");
    launch_adamw(dtype, param, grad, master, exp_avg, exp_avg_sq, (size_t)n, lr, beta1, beta2, eps, weight_decay, step, gnorm_sq_or_null, max_grad_norm, S(stream));
    LMX_API_END
}
int lmx_prefill_batch(lmx_model* m, lmx_seq* const* seqs, int32_t n, const void* const* embeds, const int32_t* n_tokens, int32_t block_rows, int32_t greedy,
                      void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(m && seqs && embeds && n_tokens && n >= 1, "null argument");
    std::vector<Seq*> ss((size_t)n);
    for (int i = 0; i < n; ++i) { LMX_REQUIRE(seqs[i] != nullptr, "null sequence"); ss[(size_t)i] = &seqs[i]->impl; }
    m->impl.prefill_multi(ss.data(), embeds, n_tokens, n, block_rows, greedy != 0, S(stream));
    LMX_API_END
}
int lmx_seq_copy(lmx_seq* dst, const lmx_seq* src, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(dst && src, "null argument");
    dst->impl.m->seq_copy(&dst->impl, &src->impl, S(stream));
    LMX_API_END
}
int lmx_seq_truncate(lmx_seq* s, int32_t n_rows, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(s, "null sequence");
    LMX_REQUIRE(n_rows >= 0 && n_rows <= s->impl.len, "seq_truncate: 0 <= n_rows <= current length");
    // a finished request's sequence starts another request that shares its first n_rows positions (prefix reuse): keep K / V^T rows [0, n_rows), forget the
    // rest of the request — position, token log, sampling parameters, stop rule (ordered on `stream` behind whatever still runs on the sequence)
    s->impl.len = n_rows;
    s->impl.samp = SampleParams{};
    s->impl.uid = next_seq_uid();
    s->impl.last_stream = S(stream); s->impl.used = true;
    launch_set_state(s->impl.d_len, n_rows, s->impl.d_tok, 0, 1, s->impl.d_nout, 0, S(stream));
    launch_set_stop(s->impl.d_stop, StopSpec{}, S(stream));
    LMX_API_END
}
int lmx_op_hash128(const void* base_dev, uint64_t bytes_per_item, int32_t items, uint64_t* out_dev, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(base_dev && out_dev, "null argument");
    launch_hash128(base_dev, (size_t)bytes_per_item, items, out_dev, S(stream));
    LMX_API_END
}
int lmx_op_beam_topk(int32_t dtype, const void* logits, int32_t ld, int32_t V, int32_t rows, const float* beam_scores_dev, int32_t K, float* out_scores, int32_t* out_ids,
                     void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(logits && out_scores && out_ids, "null argument");
    launch_beam_topk(dtype, logits, ld, V, rows, beam_scores_dev, K, out_scores, out_ids, S(stream));
    LMX_API_END
}
int lmx_op_beam_sample_topk(int32_t dtype, const void* logits, int32_t ld, int32_t V, int32_t rows, const uint8_t* keep_dev, const float* beam_scores_dev, float temperature,
                            uint64_t seed, uint32_t counter0, int32_t K, float* out_keys, float* out_scores, int32_t* out_ids, void* stream) {
    LMX_API_BEGIN
    LMX_REQUIRE(logits && out_keys && out_scores && out_ids, "null argument");
    launch_beam_gumbel_topk(dtype, logits, ld, V, rows, keep_dev, beam_scores_dev, temperature, seed, counter0, K, out_keys, out_scores, out_ids, S(stream));
    LMX_API_END
}

}  // extern "C"
