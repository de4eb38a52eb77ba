// Engine: weights in kernel-optimal layouts, per-sequence KV caches (K rows + Vᵀ columns), prefill / decode
// orchestration, device-resident decode loop, RCCL all-reduce for the tensor-parallel decoder.
//
// Reference call stacks this replaces: llava/model/llava_arch.py:94-97 (encode_images),
// llava/model/language_model/llava_llama.py:88-99 -> HF5:models/llama/modeling_llama.py:367-494 (decoder forward),
// HF5:models/clip/modeling_clip.py:594-657 (CLIP vision transformer).
#include "engine.h"

#include <atomic>
#include <cstdlib>
#include <cstring>

namespace lmx {

thread_local KernelTimer* g_kernel_timer = nullptr;


static thread_local std::string g_err;
void set_last_error(const std::string& s) { g_err = s; }
const char* get_last_error() { return g_err.c_str(); }

#define LMX_CHECK_NCCL(expr)                                                                                 \
    do {                                                                                                     \
        ncclResult_t _r = (expr);                                                                            \
        if (_r != ncclSuccess) throw Error{std::string(#expr) + " failed: " + ncclGetErrorString(_r)};      \
    } while (0)

static int round_up(int a, int b) { return (a + b - 1) / b * b; }

void zero_fill(void* p, size_t n) {
    static std::mutex mu;
    static hipStream_t util = nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!util) LMX_CHECK_HIP(hipStreamCreateWithFlags(&util, hipStreamNonBlocking));
    LMX_CHECK_HIP(hipMemsetAsync(p, 0, n, util));
    LMX_CHECK_HIP(hipStreamSynchronize(util));
}

// ---------------------------------------------------------------------------------------------------------------
Model::Model(const lmx_config& c) : cfg(c) {
    LMX_REQUIRE(c.abi_version == LMX_ABI_VERSION, "lmx_config.abi_version mismatch");
    LMX_REQUIRE(c.dtype == kF32 || c.dtype == kBF16 || c.dtype == kF16, "bad dtype");
    LMX_REQUIRE(c.tp_world >= 1 && c.tp_rank >= 0 && c.tp_rank < c.tp_world, "bad tensor-parallel rank/world");
    es = (int)dtype_size(c.dtype);
    // defaults of the options lmx_model_set_option can flip later (A/B runs and the bit-identity tests of the fused forms)
    { const char* e = getenv("LMX_FUSE_ROPE"); if (e) opt_fuse_rope = atoi(e) != 0; }
    { const char* e = getenv("LMX_VIS_PACK"); if (e) opt_vis_pack = atoi(e) != 0; }
    { const char* e = getenv("LMX_DECODE_SPLITQ"); if (e) opt_splitq = atoi(e) != 0; }
    { const char* e = getenv("LMX_TP_OVERLAP"); if (e) { tp_overlap = atoi(e) != 0; tp_overlap_force = atoi(e) == 2; } }
    H = c.hidden_size; D = c.head_dim; V = c.vocab_size; Vr = V; L = c.n_layers;
    LMX_REQUIRE(D == 64 || D == 128, "head_dim must be 64 or 128");
    LMX_REQUIRE(c.n_heads % c.tp_world == 0 && c.n_kv_heads % c.tp_world == 0, "heads must divide by tp_world");
    LMX_REQUIRE(c.n_heads % c.n_kv_heads == 0, "n_heads must be a multiple of n_kv_heads");
    LMX_REQUIRE(c.intermediate_size % (32 * c.tp_world) == 0, "intermediate_size must divide by 32*tp_world");
    LMX_REQUIRE(H % 64 == 0, "hidden size must be a multiple of 64");
    LMX_REQUIRE(V % 8 == 0, "vocab_size must be a multiple of 8");
    nh_l = c.n_heads / c.tp_world; nkv_l = c.n_kv_heads / c.tp_world;
    // local MLP width: the rank's I/world columns, zero-padded up to a multiple of 64 so the down-projection's K keeps the GEMM's
    // k-slab granularity (7B at TP=8: 1376 -> 1408).  Padded gate|up rows are zero => silu(0)*0 = 0 => no contribution.
    I_sh = c.intermediate_size / c.tp_world; I_l = round_up(I_sh, 64);
    qkv_n = (nh_l + 2 * nkv_l) * D;
    // vocabulary-parallel lm_head: each rank streams V / world rows per token instead of all of them (262 MB at 7B: 8 % of
    // a TP=1 token, 40 % of a TP=8 token if replicated); replicated when the vocabulary does not split into 8-row-aligned shards
    if (c.tp_world > 1 && V % (8 * c.tp_world) == 0) { V_l = V / c.tp_world; v_off = c.tp_rank * V_l; } else { V_l = V; v_off = 0; }
    s_max = round_up(c.max_position > 0 ? c.max_position : 2048, 128);   // decode attention works in 128-key chunks
    dec.resize(L);

    if (c.v_layers > 0) {
        Dv = c.v_hidden; Fv = c.v_intermediate;
        LMX_REQUIRE(c.v_image_size % c.v_patch_size == 0, "image size must be a multiple of the patch size");
        LMX_REQUIRE(Dv % 64 == 0 && Fv % 64 == 0 && Dv % c.v_heads == 0, "vision widths must be multiples of 64");
        vD = Dv / c.v_heads;
        LMX_REQUIRE(vD == 64 || vD == 128, "vision head_dim must be 64 or 128");
        const int g = c.v_image_size / c.v_patch_size;
        P = g * g; Tv = P + 1;
        kpad = round_up(3 * c.v_patch_size * c.v_patch_size, 64);
        spad = round_up(Tv, 64);
        const int n_hs = c.v_layers + 1;
        int idx = c.select_layer < 0 ? n_hs + c.select_layer : c.select_layer;
        LMX_REQUIRE(idx >= 0 && idx < n_hs, "select_layer out of range");
        v_run = idx;
        vis.resize(v_run);
        out_tokens = c.select_feature == LMX_FEATURE_PATCH ? P : Tv;
        const int nlin = c.projector_type == LMX_PROJ_LINEAR ? 1 : c.projector_type == LMX_PROJ_MLP_GELU ? c.projector_depth : 0;
        LMX_REQUIRE(c.projector_type != LMX_PROJ_MLP_GELU || c.projector_depth >= 1, "mlpNx_gelu needs depth >= 1");
        if (c.projector_type == LMX_PROJ_IDENTITY) LMX_REQUIRE(Dv == H, "identity projector needs mm_hidden_size == hidden_size");
        proj_w.assign(nlin, nullptr); proj_b.assign(nlin, nullptr);
    }
    LMX_CHECK_HIP(hipEventCreateWithFlags(&vws_done, hipEventDisableTiming));
}

Model::~Model() {
    if (comm) (void)ncclCommDestroy(comm);
    for (int r = 0; r < P2P_MAX_WORLD; ++r) if (p2p_peer[r] && p2p_peer[r] != p2p_local) (void)hipIpcCloseMemHandle(p2p_peer[r]);
    if (p2p_local) (void)hipFree(p2p_local);
    if (comm_stream) (void)hipStreamDestroy(comm_stream);
    for (auto& r : prof_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto e : prof_pool) (void)hipEventDestroy(e);
    if (vws_done) (void)hipEventDestroy(vws_done);
    if (rope) (void)hipFree(rope);
}

hipEvent_t Model::prof_event() {
    if (!prof_pool.empty()) { hipEvent_t e = prof_pool.back(); prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    LMX_CHECK_HIP(hipEventCreate(&e));
    return e;
}

std::map<std::string, Model::ProfAcc> Model::prof_resolve() {
    LMX_CHECK_HIP(hipDeviceSynchronize());
    std::map<std::string, ProfAcc> out;
    for (auto& r : prof_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { auto& a = out[r.name]; a.ms += ms; a.count += 1; }
        prof_pool.push_back(r.e0); prof_pool.push_back(r.e1);
    }
    prof_recs.clear();
    return out;
}

void* Model::alloc_weight(size_t bytes, bool zero) {
    pool.emplace_back();
    pool.back().ensure(bytes, zero);
    return pool.back().p;
}

static bool starts_with(const std::string& s, const char* p) { return s.rfind(p, 0) == 0; }

// copy a [rows x cols] sub-block (row0.., col0..) of a row-major [R x C] matrix into dst (row-major, pitch dst_cols)
static void copy_block(void* dst, int dst_cols, int dst_row0, const void* src, int C, int row0, int col0, int rows, int cols, int es, hipStream_t st) {
    LMX_CHECK_HIP(hipMemcpy2DAsync(static_cast<char*>(dst) + (size_t)dst_row0 * dst_cols * es, (size_t)dst_cols * es,
                                   static_cast<const char*>(src) + ((size_t)row0 * C + col0) * es, (size_t)C * es,
                                   (size_t)cols * es, rows, hipMemcpyDeviceToDevice, st));
}

void Model::load_weight(const std::string& name, const void* src, int dtype, int ndim, const int64_t* shape, hipStream_t st) {
    LMX_REQUIRE(dtype == cfg.dtype, "weight dtype must equal the model dtype (cast on the host side): " + name);
    LMX_REQUIRE(src != nullptr && ndim >= 1 && ndim <= 4, "bad tensor: " + name);
    int64_t numel = 1;
    for (int i = 0; i < ndim; ++i) numel *= shape[i];
    auto expect = [&](std::initializer_list<int64_t> want) {
        bool ok = (int)want.size() == ndim;
        int i = 0;
        for (auto w : want) { if (ok && shape[i] != w) ok = false; ++i; }
        if (!ok) {
            std::string got = "[", exp = "[";
            for (int k = 0; k < ndim; ++k) got += std::to_string(shape[k]) + (k + 1 < ndim ? "," : "");
            for (auto w : want) exp += std::to_string(w) + ",";
            throw Error{"shape mismatch for " + name + ": got " + got + "] expected " + exp + "]"};
        }
    };
    auto plain = [&](void*& dst) {
        if (!dst) dst = alloc_weight((size_t)numel * es);
        LMX_CHECK_HIP(hipMemcpyAsync(dst, src, (size_t)numel * es, hipMemcpyDeviceToDevice, st));
    };
    const int r = cfg.tp_rank;
    const int nh = cfg.n_heads, nkv = cfg.n_kv_heads, I = cfg.intermediate_size;

    if (name == "model.embed_tokens.weight") { expect({V, H}); plain(embed); }
    else if (name == "model.norm.weight") { expect({H}); plain(final_norm); }
    else if (name == "lm_head.weight") {
        expect({V, H});
        if (V_l == V) plain(lm_head);
        else {
            if (!lm_head) lm_head = alloc_weight((size_t)V_l * H * es);
            LMX_CHECK_HIP(hipMemcpyAsync(lm_head, static_cast<const char*>(src) + (size_t)v_off * H * es, (size_t)V_l * H * es, hipMemcpyDeviceToDevice, st));
        }
    }
    else if (starts_with(name, "model.layers.")) {
        const size_t p0 = strlen("model.layers.");
        const size_t dot = name.find('.', p0);
        LMX_REQUIRE(dot != std::string::npos, "bad layer name " + name);
        const int li = atoi(name.substr(p0, dot - p0).c_str());
        LMX_REQUIRE(li >= 0 && li < L, "layer index out of range: " + name);
        const std::string sub = name.substr(dot + 1);
        DecLayerW& w = dec[li];
        if (!w.wqkv) {
            w.wqkv = alloc_weight((size_t)qkv_n * H * es);
            w.wo = alloc_weight((size_t)H * nh_l * D * es);
            w.wgu = alloc_weight((size_t)2 * I_l * H * es, I_l != I_sh);
            w.wd = alloc_weight((size_t)H * I_l * es, I_l != I_sh);
        }
        if (sub == "self_attn.q_proj.weight") { expect({(int64_t)nh * D, H}); copy_block(w.wqkv, H, 0, src, H, r * nh_l * D, 0, nh_l * D, H, es, st); }
        else if (sub == "self_attn.k_proj.weight") { expect({(int64_t)nkv * D, H}); copy_block(w.wqkv, H, nh_l * D, src, H, r * nkv_l * D, 0, nkv_l * D, H, es, st); }
        else if (sub == "self_attn.v_proj.weight") { expect({(int64_t)nkv * D, H}); copy_block(w.wqkv, H, (nh_l + nkv_l) * D, src, H, r * nkv_l * D, 0, nkv_l * D, H, es, st); }
        else if (sub == "self_attn.o_proj.weight") { expect({H, (int64_t)nh * D}); copy_block(w.wo, nh_l * D, 0, src, nh * D, 0, r * nh_l * D, H, nh_l * D, es, st); }
        else if (sub == "mlp.gate_proj.weight") { expect({I, H}); launch_interleave_half(cfg.dtype, static_cast<const char*>(src) + (size_t)r * I_sh * H * es, w.wgu, I_sh, H, 0, st); }
        else if (sub == "mlp.up_proj.weight") { expect({I, H}); launch_interleave_half(cfg.dtype, static_cast<const char*>(src) + (size_t)r * I_sh * H * es, w.wgu, I_sh, H, 1, st); }
        else if (sub == "mlp.down_proj.weight") { expect({H, I}); copy_block(w.wd, I_l, 0, src, I, 0, r * I_sh, H, I_sh, es, st); }
        else if (sub == "input_layernorm.weight") { expect({H}); plain(w.ln1); }
        else if (sub == "post_attention_layernorm.weight") { expect({H}); plain(w.ln2); }
        else if (sub == "self_attn.rotary_emb.inv_freq") { return; }   // buffer in old checkpoints; recomputed
        else throw Error{"unknown decoder tensor: " + name};
    }
    else if (starts_with(name, "mm_projector.")) {
        LMX_REQUIRE(!proj_w.empty(), "projector tensor given but projector is identity/absent: " + name);
        int li = 0; std::string leaf;
        const std::string rest = name.substr(strlen("mm_projector."));
        if (rest == "weight" || rest == "bias") { li = 0; leaf = rest; }
        else {
            const size_t dot = rest.find('.');
            LMX_REQUIRE(dot != std::string::npos, "bad projector name " + name);
            const int seq_idx = atoi(rest.substr(0, dot).c_str());
            LMX_REQUIRE(seq_idx % 2 == 0, "projector index must address a Linear (even index): " + name);
            li = seq_idx / 2; leaf = rest.substr(dot + 1);
        }
        LMX_REQUIRE(li >= 0 && li < (int)proj_w.size(), "projector layer out of range: " + name);
        const int in = li == 0 ? Dv : H;
        if (leaf == "weight") { expect({H, in}); plain(proj_w[li]); }
        else if (leaf == "bias") { expect({H}); plain(proj_b[li]); }
        else throw Error{"unknown projector tensor: " + name};
    }
    else if (starts_with(name, "vision.")) {
        LMX_REQUIRE(cfg.v_layers > 0, "vision tensor given but no vision tower configured: " + name);
        const std::string sub = name.substr(strlen("vision."));
        const int ps = cfg.v_patch_size;
        if (sub == "embeddings.class_embedding") { expect({Dv}); plain(v_cls); }
        else if (sub == "embeddings.patch_embedding.weight") {
            expect({Dv, 3, ps, ps});
            if (!v_patch_w) v_patch_w = alloc_weight((size_t)Dv * kpad * es, true);
            copy_block(v_patch_w, kpad, 0, src, 3 * ps * ps, 0, 0, Dv, 3 * ps * ps, es, st);
        }
        else if (sub == "embeddings.position_embedding.weight") { expect({Tv, Dv}); plain(v_pos); }
        else if (sub == "embeddings.position_ids") { return; }
        else if (sub == "pre_layrnorm.weight") { expect({Dv}); plain(v_pre_w); }
        else if (sub == "pre_layrnorm.bias") { expect({Dv}); plain(v_pre_b); }
        else if (starts_with(sub, "post_layernorm.")) { return; }      // only feeds the unused pooled output
        else if (starts_with(sub, "encoder.layers.")) {
            const size_t p0 = strlen("encoder.layers.");
            const size_t dot = sub.find('.', p0);
            LMX_REQUIRE(dot != std::string::npos, "bad vision layer name " + name);
            const int li = atoi(sub.substr(p0, dot - p0).c_str());
            LMX_REQUIRE(li >= 0 && li < cfg.v_layers, "vision layer index out of range: " + name);
            if (li >= v_run) return;                                    // layers past select_layer are never executed
            const std::string leaf = sub.substr(dot + 1);
            VisLayerW& w = vis[li];
            if (!w.wqkv) { w.wqkv = alloc_weight((size_t)3 * Dv * Dv * es); w.bqkv = alloc_weight((size_t)3 * Dv * es); }
            auto qkv_w = [&](int slot) { expect({Dv, Dv}); copy_block(w.wqkv, Dv, slot * Dv, src, Dv, 0, 0, Dv, Dv, es, st); };
            auto qkv_b = [&](int slot) { expect({Dv}); LMX_CHECK_HIP(hipMemcpyAsync(static_cast<char*>(w.bqkv) + (size_t)slot * Dv * es, src, (size_t)Dv * es, hipMemcpyDeviceToDevice, st)); };
            if (leaf == "self_attn.q_proj.weight") qkv_w(0);
            else if (leaf == "self_attn.k_proj.weight") qkv_w(1);
            else if (leaf == "self_attn.v_proj.weight") qkv_w(2);
            else if (leaf == "self_attn.q_proj.bias") qkv_b(0);
            else if (leaf == "self_attn.k_proj.bias") qkv_b(1);
            else if (leaf == "self_attn.v_proj.bias") qkv_b(2);
            else if (leaf == "self_attn.out_proj.weight") { expect({Dv, Dv}); plain(w.wo); }
            else if (leaf == "self_attn.out_proj.bias") { expect({Dv}); plain(w.bo); }
            else if (leaf == "mlp.fc1.weight") { expect({Fv, Dv}); plain(w.fc1); }
            else if (leaf == "mlp.fc1.bias") { expect({Fv}); plain(w.b1); }
            else if (leaf == "mlp.fc2.weight") { expect({Dv, Fv}); plain(w.fc2); }
            else if (leaf == "mlp.fc2.bias") { expect({Dv}); plain(w.b2); }
            else if (leaf == "layer_norm1.weight") { expect({Dv}); plain(w.ln1w); }
            else if (leaf == "layer_norm1.bias") { expect({Dv}); plain(w.ln1b); }
            else if (leaf == "layer_norm2.weight") { expect({Dv}); plain(w.ln2w); }
            else if (leaf == "layer_norm2.bias") { expect({Dv}); plain(w.ln2b); }
            else throw Error{"unknown vision tensor: " + name};
        }
        else throw Error{"unknown vision tensor: " + name};
    }
    else throw Error{"unknown tensor name: " + name};
    loaded.insert(name);
}

void Model::finalize() {
    std::string missing;
    auto need = [&](const std::string& n) { if (!loaded.count(n)) missing += n + " "; };
    need("model.embed_tokens.weight"); need("model.norm.weight"); need("lm_head.weight");
    for (int i = 0; i < L; ++i) {
        const std::string p = "model.layers." + std::to_string(i) + ".";
        for (const char* s : {"self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight", "self_attn.o_proj.weight",
                              "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight", "input_layernorm.weight",
                              "post_attention_layernorm.weight"})
            need(p + s);
    }
    if (cfg.v_layers > 0) {
        for (const char* s : {"embeddings.class_embedding", "embeddings.patch_embedding.weight", "embeddings.position_embedding.weight",
                              "pre_layrnorm.weight", "pre_layrnorm.bias"})
            need(std::string("vision.") + s);
        for (int i = 0; i < v_run; ++i) {
            const std::string p = "vision.encoder.layers." + std::to_string(i) + ".";
            for (const char* s : {"self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.out_proj", "mlp.fc1", "mlp.fc2",
                                  "layer_norm1", "layer_norm2"}) {
                need(p + s + ".weight"); need(p + s + ".bias");
            }
        }
        for (size_t i = 0; i < proj_w.size(); ++i) {
            const std::string p = cfg.projector_type == LMX_PROJ_LINEAR ? "mm_projector." : "mm_projector." + std::to_string(2 * i) + ".";
            need(p + "weight"); need(p + "bias");
        }
    }
    if (!rope) missing += "<rope table: call lmx_set_rope_table> ";
    if (!missing.empty()) throw Error{"missing tensors: " + missing};
}

void Model::set_rope(const float* host, int n_pos) {
    LMX_REQUIRE(host != nullptr && n_pos >= s_max, "rope table must cover max_position (rounded up to 64)");
    if (rope) { (void)hipFree(rope); rope = nullptr; }
    LMX_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&rope), (size_t)n_pos * D * sizeof(float)));
    LMX_CHECK_HIP(hipMemcpy(rope, host, (size_t)n_pos * D * sizeof(float), hipMemcpyHostToDevice));
    rope_npos = n_pos;
}

void Model::ensure_comm_stream() {
    std::lock_guard<std::mutex> lk(mu);
    if (!comm_stream) LMX_CHECK_HIP(hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking));
}

void Seq::ensure_events() {
    if (ev_c[0]) return;
    for (int i = 0; i < 2; ++i) {
        LMX_CHECK_HIP(hipEventCreateWithFlags(&ev_c[i], hipEventDisableTiming));
        LMX_CHECK_HIP(hipEventCreateWithFlags(&ev_r[i], hipEventDisableTiming));
    }
}

// ---- one-shot P2P all-reduce plumbing ----------------------------------------------------------------------------------
void Model::p2p_local_handle(void* out64) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t size");
    if (!p2p_local) {
        // one allocation: the one-shot region (decode-sized rows) and, behind it, the two-shot region for messages of up to max(s_max, 4096) rows
        p2p_big_max_count = (size_t)std::max(s_max, 4096) * H;
        p2p_big = p2p_big_geometry(cfg.tp_world, p2p_big_max_count, es, p2p_buffer_bytes(cfg.tp_world, H, es));
        const size_t bytes = p2p_big.end;
        // uncached: peers' stores and this rank's flag polls must not sit in a cache while a kernel runs (fine-grained coherent memory where the
        // uncached kind is not available; measured equal, profiles/r04_p2p_mem_ab.txt)
        if (hipExtMallocWithFlags(&p2p_local, bytes, hipDeviceMallocUncached) != hipSuccess) {
            (void)hipGetLastError();
            LMX_CHECK_HIP(hipExtMallocWithFlags(&p2p_local, bytes, hipDeviceMallocFinegrained));
        }
        LMX_CHECK_HIP(hipMemset(p2p_local, 0, bytes));
        LMX_CHECK_HIP(hipDeviceSynchronize());
    }
    hipIpcMemHandle_t hd;
    LMX_CHECK_HIP(hipIpcGetMemHandle(&hd, p2p_local));
    memcpy(out64, &hd, sizeof(hd));
}

void Model::p2p_connect(const void* handles) {
    LMX_REQUIRE(p2p_local != nullptr, "lmx_tp_p2p_connect before lmx_tp_p2p_local_handle");
    LMX_REQUIRE(cfg.tp_world <= P2P_MAX_WORLD, "p2p all-reduce supports up to 8 ranks");
    for (int r = 0; r < cfg.tp_world; ++r) {
        if (r == cfg.tp_rank) { p2p_peer[r] = p2p_local; continue; }
        hipIpcMemHandle_t hd;
        memcpy(&hd, static_cast<const char*>(handles) + (size_t)r * 64, sizeof(hd));
        LMX_CHECK_HIP(hipIpcOpenMemHandle(&p2p_peer[r], hd, hipIpcMemLazyEnablePeerAccess));
    }
    { const char* e = getenv("LMX_TP_P2P_ALL"); p2p_all = e && atoi(e) != 0; }
    { const char* e = getenv("LMX_TP_P2P_BIG"); p2p_big_on = !(e && atoi(e) == 0); }       // 0: prefill-sized messages stay on RCCL (or, with LMX_TP_P2P_ALL, on 32-row one-shot launches)
    p2p_seq = 0; p2p_big_seq = 0;
    p2p_on = true;
}

// 0 = every wait so far saw its flags; otherwise the sequence number of an all-reduce whose wait timed out
int Model::p2p_status(hipStream_t st) {
    if (!p2p_local) return 0;
    uint32_t v = 0;
    const size_t off = p2p_flags_offset(cfg.tp_world, H, es) + (size_t)2 * cfg.tp_world * P2P_MAX_ROWS * 4;
    LMX_CHECK_HIP(hipMemcpyAsync(&v, static_cast<char*>(p2p_local) + off, 4, hipMemcpyDeviceToHost, st));
    LMX_CHECK_HIP(hipStreamSynchronize(st));
    if (v != 0) {
        // report a timeout once, then re-arm: the caller turns it into an error for THIS request; the word would otherwise fail
        // every later generate() on the model although the peers have recovered
        LMX_CHECK_HIP(hipMemsetAsync(static_cast<char*>(p2p_local) + off, 0, 4, st));
        LMX_CHECK_HIP(hipStreamSynchronize(st));
    }
    return (int)v;
}

void Model::allreduce(void* buf, size_t count, hipStream_t st) {
    if (cfg.tp_world == 1 && !comm) return;      // a 1-rank communicator (tests) still goes through RCCL
    if (ar_hook) { ar_hook(buf, (uint64_t)count, cfg.dtype, st, ar_ctx); return; }
    if ((count + H - 1) / H > (size_t)P2P_MAX_ROWS && p2p_big_usable(count)) {
        // prefill-sized message: reduce-scatter + all-gather over all xGMI links in one launch (p2p.hip)
        P2PBigLaunch l{buf, count, cfg.tp_world, cfg.tp_rank, ++p2p_big_seq, p2p_flags_offset(cfg.tp_world, H, es) + (size_t)2 * cfg.tp_world * P2P_MAX_ROWS * 4, {}, p2p_big};
        for (int p = 0; p < cfg.tp_world; ++p) l.peer[p] = p2p_peer[p];
        launch_p2p_allreduce_big(cfg.dtype, l, st);
        return;
    }
    if (p2p_on && count % 8 == 0 && (p2p_all || (count + H - 1) / H <= (size_t)P2P_MAX_ROWS)) {
        // decode-sized message: one launch, one xGMI hop (p2p.hip).  Larger ones only when forced (LMX_TP_P2P_ALL, tests).
        // The message is cut into [H]-element rows (one workgroup each); the last row may be partial (logits: V is not a multiple of H).
        const size_t rows = (count + H - 1) / H;
        for (size_t r0 = 0; r0 < rows; r0 += P2P_MAX_ROWS) {
            const int n = (int)(rows - r0 < (size_t)P2P_MAX_ROWS ? rows - r0 : P2P_MAX_ROWS);
            P2PLaunch l{static_cast<char*>(buf) + r0 * H * es, H, cfg.tp_world, cfg.tp_rank, n, ++p2p_seq, {}};
            for (int p = 0; p < cfg.tp_world; ++p) l.peer[p] = p2p_peer[p];
            if (r0 + n == rows) l.last_len = (int)(count - (rows - 1) * H);
            launch_p2p_allreduce(cfg.dtype, l, st);
        }
        return;
    }
    LMX_REQUIRE(comm != nullptr, "tensor-parallel model used before lmx_tp_init");
    const ncclDataType_t dt = cfg.dtype == kF32 ? ncclFloat32 : cfg.dtype == kBF16 ? ncclBfloat16 : ncclFloat16;
    LMX_CHECK_NCCL(ncclAllReduce(buf, buf, count, dt, ncclSum, comm, st));
}

bool Model::allreduce_norm(void* buf, int rows, const void* norm_w, void* x_out, hipStream_t st) {
    const bool tp = cfg.tp_world > 1 || comm;
    if (tp && !ar_hook && p2p_on && norm_w && x_out && rows >= 1 && rows <= P2P_MAX_ROWS && H % 8 == 0 &&
        !(rows > P2P_MAX_ROWS && p2p_big_usable((size_t)rows * H))) {
        P2PLaunch l{buf, H, cfg.tp_world, cfg.tp_rank, rows, ++p2p_seq, {}};
        for (int p = 0; p < cfg.tp_world; ++p) l.peer[p] = p2p_peer[p];
        l.norm_w = norm_w; l.x_out = x_out; l.eps = cfg.rms_eps;
        launch_p2p_allreduce(cfg.dtype, l, st);
        return true;
    }
    allreduce(buf, (size_t)rows * H, st);
    return false;
}

void Model::gather_logits(void* logits, int rows, hipStream_t st) {
    if (V_l == V) return;
    allreduce(logits, (size_t)rows * V, st);
}

// ---------------------------------------------------------------------------------------------------------------
// vision tower + projector
// ---------------------------------------------------------------------------------------------------------------
void Model::encode_images(const void* pixels, int n, void* feats, hipStream_t st, bool tower_only) {
    LMX_REQUIRE(cfg.v_layers > 0, "no vision tower configured");
    LMX_REQUIRE(n > 0 && pixels && feats, "encode_images: bad arguments");
    std::lock_guard<std::mutex> lk(mu);     // one shared vision workspace
    if (vws_stream && vws_stream != st) LMX_CHECK_HIP(hipStreamWaitEvent(st, vws_done, 0));   // previous user on another stream
    const int dt = cfg.dtype;
    const int rows = n * Tv;                // token rows incl. CLS
    const int prow = n * P;
    const int srows = n * out_tokens;
    // workspace carve (element counts)
    size_t off = 0;
    auto carve = [&](size_t elems) { size_t o = off; off += (elems * es + 255) / 256 * 256; return o; };
    const size_t o_patch = carve((size_t)prow * kpad), o_pe = carve((size_t)prow * Dv), o_h = carve((size_t)rows * Dv),
                 o_x = carve((size_t)rows * Dv), o_qkv = carve((size_t)rows * 3 * Dv), o_attn = carve((size_t)rows * Dv),
                 o_mlp = carve((size_t)rows * Fv), o_sel = carve((size_t)srows * Dv), o_p1 = carve((size_t)srows * H),
                 o_p2 = carve((size_t)srows * H);
    const int vh = cfg.v_heads;
    const size_t aws_floats = dt == kF32 ? decode_attn_ws_floats(Tv, vh, 1, vD) : 0;
    const size_t o_aws = off; off += aws_floats * 4;
    if (off > vws.bytes) { LMX_CHECK_HIP(hipStreamSynchronize(st)); vws.ensure(off); }
    const size_t kv_bytes = (size_t)vh * spad * vD * es;
    if (vkc.bytes < kv_bytes * n) {
        LMX_CHECK_HIP(hipStreamSynchronize(st));
        vkc.ensure(kv_bytes * n, true); vvt.ensure(kv_bytes * n, true);
    }
    char* W = vws.as<char>();
    void *patch = W + o_patch, *pe = W + o_pe, *h = W + o_h, *x = W + o_x, *qkv = W + o_qkv, *attn = W + o_attn, *mlp = W + o_mlp,
         *sel = W + o_sel, *p1 = W + o_p1, *p2 = W + o_p2;
    const int gv = cfg.gemm_variant;

    { LMX_PROF("vis.im2col"); launch_im2col(dt, pixels, patch, n, cfg.v_image_size, cfg.v_patch_size, kpad, st); }
    { LMX_PROF("vis.gemm.patch"); launch_gemm(dt, GemmArgs{patch, v_patch_w, pe, nullptr, nullptr, prow, Dv, kpad, kpad, kpad, Dv, 0, kActNone}, gv, st); }
    { LMX_PROF("vis.embed_ln"); launch_clip_embed_ln(dt, pe, v_cls, v_pos, v_pre_w, v_pre_b, h, n, P, Dv, cfg.v_ln_eps, st); }
    const float scale = 1.f / sqrtf((float)vD);
    for (int l = 0; l < v_run; ++l) {
        const VisLayerW& w = vis[l];
        { LMX_PROF("vis.layernorm"); launch_layernorm(dt, h, w.ln1w, w.ln1b, x, rows, Dv, Dv, Dv, cfg.v_ln_eps, st); }
        // K rows / V^T columns leave the q|k|v GEMM's epilogue (16-bit models; option vis_pack = 0 keeps the separate pack launch: 23 launches of ~5 us per image)
        const bool pack_fused = dt != kF32 && opt_vis_pack && vD % 4 == 0;
        {
            LMX_PROF("vis.gemm.qkv");
            GemmArgs g{x, w.wqkv, qkv, w.bqkv, nullptr, rows, 3 * Dv, Dv, Dv, Dv, 3 * Dv, 0, kActNone};
            if (pack_fused) { g.pk_kc = vkc.p; g.pk_vt = vvt.p; g.pk_heads = vh; g.pk_D = vD; g.pk_rows = Tv; g.pk_spad = spad; g.pk_img_stride = kv_bytes / es; }
            launch_gemm(dt, g, gv, st);
        }
        for (int i = 0; i < n; ++i) {
            char* qkv_i = static_cast<char*>(qkv) + (size_t)i * Tv * 3 * Dv * es;
            char* attn_i = static_cast<char*>(attn) + (size_t)i * Tv * Dv * es;
            void* kc = vkc.as<char>() + (size_t)i * kv_bytes;
            void* vt = vvt.as<char>() + (size_t)i * kv_bytes;
            RopeKvArgs ra{qkv_i, kc, vt, nullptr, nullptr, 0, Tv, 3 * Dv, vh, vh, spad};
            if (!pack_fused) { LMX_PROF("vis.kv_pack"); launch_rope_kv(dt, vD, ra, st); }
            if (dt == kF32) {
                DecodeAttnArgs da{qkv_i, attn_i, kc, vt, nullptr, 0, Tv, Tv, 0, 3 * Dv, Dv, vh, vh, spad, 1, scale,
                                  reinterpret_cast<float*>(W + o_aws)};
                { LMX_PROF("vis.attn"); launch_decode_attn(dt, vD, da, st); }
            } else {
                FlashArgs fa{qkv_i, attn_i, kc, vt, Tv, Tv, 0, 3 * Dv, Dv, vh, vh, spad, scale, 0};
                { LMX_PROF("vis.attn"); launch_flash_prefill(dt, vD, fa, st); }
            }
        }
        { LMX_PROF("vis.gemm.out"); launch_gemm(dt, GemmArgs{attn, w.wo, h, w.bo, h, rows, Dv, Dv, Dv, Dv, Dv, Dv, kActNone}, gv, st); }
        { LMX_PROF("vis.layernorm"); launch_layernorm(dt, h, w.ln2w, w.ln2b, x, rows, Dv, Dv, Dv, cfg.v_ln_eps, st); }
        { LMX_PROF("vis.gemm.fc1"); launch_gemm(dt, GemmArgs{x, w.fc1, mlp, w.b1, nullptr, rows, Fv, Dv, Dv, Dv, Fv, 0, kActQuickGelu}, gv, st); }
        { LMX_PROF("vis.gemm.fc2"); launch_gemm(dt, GemmArgs{mlp, w.fc2, h, w.b2, h, rows, Dv, Fv, Fv, Fv, Dv, Dv, kActNone}, gv, st); }
    }
    // feature_select (clip_encoder.py:29-37)
    const void* selp = h;
    if (cfg.select_feature == LMX_FEATURE_PATCH) { launch_copy_rows(dt, h, sel, n, Tv, 1, P, Dv, st); selp = sel; }
    // projector (multimodal_projector/builder.py:33-51); tower_only: CLIPVisionTower.forward's own result (clip_encoder.py:39-51)
    if (proj_w.empty() || tower_only) {
        LMX_CHECK_HIP(hipMemcpyAsync(feats, selp, (size_t)srows * Dv * es, hipMemcpyDeviceToDevice, st));
    } else {
        const void* in = selp; int in_dim = Dv;
        const int nl = (int)proj_w.size();
        for (int i = 0; i < nl; ++i) {
            void* out = i == nl - 1 ? feats : (i % 2 == 0 ? p1 : p2);
            const int act = i == nl - 1 ? kActNone : kActGeluErf;
            { LMX_PROF("proj.gemm"); launch_gemm(dt, GemmArgs{in, proj_w[i], out, proj_b[i], nullptr, srows, H, in_dim, in_dim, in_dim, H, 0, act}, gv, st); }
            in = out; in_dim = H;
        }
    }
    LMX_CHECK_HIP(hipEventRecord(vws_done, st));
    vws_stream = st;
}

void Model::gather_embeds(const int32_t* src, int rows, const void* feats, void* out, hipStream_t st) {
    LMX_REQUIRE(embed != nullptr, "embed_tokens not loaded");
    launch_gather_embed(cfg.dtype, src, embed, feats, out, rows, H, st);
}

// ---------------------------------------------------------------------------------------------------------------
// sequences
// ---------------------------------------------------------------------------------------------------------------
static std::atomic<uint64_t> g_seq_uid{1};

uint64_t next_seq_uid() { return g_seq_uid.fetch_add(1); }

Seq::Seq(Model* mm) : m(mm) {
    uid = g_seq_uid.fetch_add(1);
    layer_stride = (size_t)m->nkv_l * m->s_max * m->D * m->es;
    kc.ensure(layer_stride * m->L, true);
    vt.ensure(layer_stride * m->L, true);
    log_cap = m->s_max + 8;
    state.ensure(16 + (size_t)log_cap * 8, true);
    d_len = state.as<int>(); d_nout = d_len + 1;
    d_tok = reinterpret_cast<int64_t*>(state.as<char>() + 8);
    d_log = d_tok + 1;
    stopbuf.ensure(sizeof(StopSpec), true);
    d_stop = stopbuf.as<StopSpec>();
    // decode workspace
    const int es = m->es;
    n_split = (m->s_max + 127) / 128;          // fixed 128-key chunks (attention.hip: DF_CHUNK)
    const size_t aws = decode_fused_ws_floats(m->nh_l, n_split, m->D);
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t o_h = carve((size_t)m->H * es), o_qkv = carve((size_t)m->qkv_n * es), o_attn = carve((size_t)m->nh_l * m->D * es),
                 o_act = carve((size_t)m->I_l * es), o_log = carve((size_t)m->V * es), o_aws = carve(aws * 4), o_cnt = carve((size_t)m->nh_l * 4);
    dws.ensure(off, true);
    char* W = dws.as<char>();
    d_h = W + o_h; d_qkv = W + o_qkv; d_attn = W + o_attn; d_act = W + o_act; d_logits = W + o_log;
    d_aws = reinterpret_cast<float*>(W + o_aws);
    d_cnt = reinterpret_cast<int*>(W + o_cnt);
    kv_gran.ensure((size_t)2 * m->nkv_l * m->D * 8, true);
    LMX_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_status), sizeof(unsigned), hipHostMallocMapped));
    *h_status = 0;
    LMX_CHECK_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_status), h_status, 0));
}

Seq::~Seq() {
    if (h_status) (void)hipHostFree(h_status);
    if (ev_idle) (void)hipEventDestroy(ev_idle);
    for (int i = 0; i < 2; ++i) { if (ev_c[i]) (void)hipEventDestroy(ev_c[i]); if (ev_r[i]) (void)hipEventDestroy(ev_r[i]); }
}

// ---------------------------------------------------------------------------------------------------------------
// prefill
// ---------------------------------------------------------------------------------------------------------------
void Model::prefill(Seq* s, const void* embeds, int T, int chunk, void* logits, bool logits_all, bool greedy, hipStream_t st, void* hidden, void* attn_probs) {
    LMX_REQUIRE(T > 0 && embeds, "prefill: empty input");
    LMX_REQUIRE(!attn_probs || (cfg.tp_world == 1 && comm == nullptr), "output_attentions: single process only (a tensor-parallel rank holds a slice of the heads)");
    const int kv_after = s->len + T;                        // key range of the attention maps' rows: [0, len + T)
    LMX_REQUIRE(s->len + T <= s_max, "prefill: sequence would exceed the KV-cache capacity (max_position)");
    LMX_REQUIRE(rope != nullptr, "rope table not set");
    s->last_stream = st; s->used = true;
    if (chunk <= 0 || chunk > T) chunk = T;
    const int dt = cfg.dtype;
    // workspace
    {
        size_t off = 0;
        auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
        const size_t aws = dt == kF32 ? decode_attn_ws_floats(chunk, nh_l, 1, D) * 4 : 0;
        carve((size_t)chunk * H * es); carve((size_t)chunk * H * es); carve((size_t)chunk * qkv_n * es);
        carve((size_t)chunk * nh_l * D * es); carve((size_t)chunk * I_l * es); carve(aws); carve((size_t)V * es);
        if (off > s->pws.bytes) { LMX_CHECK_HIP(hipStreamSynchronize(st)); s->pws.ensure(off); }
    }
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    char* W = s->pws.as<char>();
    void* h = W + carve((size_t)chunk * H * es);
    void* x = W + carve((size_t)chunk * H * es);
    void* qkv = W + carve((size_t)chunk * qkv_n * es);
    void* attn = W + carve((size_t)chunk * nh_l * D * es);
    void* act = W + carve((size_t)chunk * I_l * es);
    float* aws = reinterpret_cast<float*>(W + carve(dt == kF32 ? decode_attn_ws_floats(chunk, nh_l, 1, D) * 4 : 0));
    void* last_logits = W + carve((size_t)V * es);
    const float scale = 1.f / sqrtf((float)D);
    const int gv = cfg.gemm_variant;
    const bool lead = cfg.tp_rank == 0;
    // N = hidden linears (o_proj, down_proj) give too few 256x256 tiles at prompt lengths around 1k: the ping-pong GEMM slices K
    // over up to three workgroups per tile and reduces fp32 partial tiles in the launch (gemm8p.hip).  The scratch is per sequence
    // (request threads prefill concurrently); <= 256 partial tiles = 64 MiB by construction of gemm8p_pick_split.
    if (dt != kF32 && !s->skw.p) {
        LMX_CHECK_HIP(hipStreamSynchronize(st));
        s->skw.ensure((size_t)256 * 256 * 256 * sizeof(float));
    }
    auto with_scratch = [&](GemmArgs g) { g.skw = s->skw.p; return g; };

    for (int c0 = 0; c0 < T; c0 += chunk) {
        const int tc = (T - c0) < chunk ? (T - c0) : chunk;
        const int pos0 = s->len + c0;
        LMX_CHECK_HIP(hipMemcpyAsync(h, static_cast<const char*>(embeds) + (size_t)c0 * H * es, (size_t)tc * H * es, hipMemcpyDeviceToDevice, st));
        // rows [r0, r0+n) of this chunk through the attention block / the MLP block of layer l (partial sums land in h)
        auto rows = [&](void* base, int r0, size_t width) { return static_cast<char*>(base) + (size_t)r0 * width * es; };
        const bool tp_active = cfg.tp_world > 1 || comm != nullptr;
        auto attn_block = [&](int l, int r0, int n) {
            const DecLayerW& w = dec[l];
            void* kc = s->kc.as<char>() + (size_t)l * s->layer_stride;
            void* vt = s->vt.as<char>() + (size_t)l * s->layer_stride;
            void *hr = rows(h, r0, H), *xr = rows(x, r0, H), *qr = rows(qkv, r0, qkv_n), *ar = rows(attn, r0, (size_t)nh_l * D);
            { LMX_PROF("prefill.rmsnorm"); launch_rmsnorm(dt, hr, w.ln1, xr, n, H, H, H, cfg.rms_eps, st); }
            // RoPE + KV-cache append ride in the q|k|v GEMM's epilogue where that launch is the un-split ping-pong kernel over head-aligned tiles
            const bool qf = gv == 0 && opt_fuse_rope && gemm_fuses_qkv(dt, n, H, D, nh_l, nkv_l, pos0 + r0, s_max, false);
            {
                LMX_PROF_K("prefill.gemm.qkv");
                GemmArgs g = with_scratch(GemmArgs{xr, w.wqkv, qr, nullptr, nullptr, n, qkv_n, H, H, H, qkv_n, 0, kActNone});
                if (qf) { g.qf_rope = rope; g.qf_kc = kc; g.qf_vt = vt; g.qf_pos0 = pos0 + r0; g.qf_nh = nh_l; g.qf_nkv = nkv_l; g.qf_smax = s_max; g.qf_D = D; }
                launch_gemm(dt, g, gv, st);
            }
            if (!qf) { LMX_PROF("prefill.rope_kv"); launch_rope_kv(dt, D, RopeKvArgs{qr, kc, vt, rope, nullptr, pos0 + r0, n, qkv_n, nh_l, nkv_l, s_max}, st); }
            if (attn_probs) {
                void* pl = static_cast<char*>(attn_probs) + (size_t)l * nh_l * T * kv_after * es;
                launch_attn_probs(dt, D, AttnProbsArgs{qr, kc, pl, n, c0 + r0, T, pos0 + r0, kv_after, qkv_n, nh_l, nkv_l, s_max, scale}, st);
            }
            if (dt == kF32) {
                { LMX_PROF("prefill.attn"); launch_decode_attn(dt, D, DecodeAttnArgs{qr, ar, kc, vt, nullptr, pos0 + r0, n, 0, 1, qkv_n, nh_l * D, nh_l, nkv_l, s_max, 1, scale, aws}, st); }
            } else {
                { LMX_PROF("prefill.attn"); launch_flash_prefill(dt, D, FlashArgs{qr, ar, kc, vt, n, pos0 + r0 + n, pos0 + r0, qkv_n, nh_l * D, nh_l, nkv_l, s_max, scale, 1}, st); }
            }
            { LMX_PROF_K("prefill.gemm.o"); launch_gemm(dt, with_scratch(GemmArgs{ar, w.wo, hr, nullptr, lead ? hr : nullptr, n, H, nh_l * D, nh_l * D, nh_l * D, H, H, kActNone}), gv, st); }
        };
        auto mlp_block = [&](int l, int r0, int n) {
            const DecLayerW& w = dec[l];
            void *hr = rows(h, r0, H), *xr = rows(x, r0, H), *cr = rows(act, r0, I_l);
            { LMX_PROF("prefill.rmsnorm"); launch_rmsnorm(dt, hr, w.ln2, xr, n, H, H, H, cfg.rms_eps, st); }
            { LMX_PROF_K("prefill.gemm.gate_up"); launch_gemm(dt, with_scratch(GemmArgs{xr, w.wgu, cr, nullptr, nullptr, n, 2 * I_l, H, H, H, I_l, 0, kActSiluMul}), gv, st); }
            { LMX_PROF_K("prefill.gemm.down"); launch_gemm(dt, with_scratch(GemmArgs{cr, w.wd, hr, nullptr, lead ? hr : nullptr, n, H, I_l, I_l, I_l, H, H, kActNone}), gv, st); }
        };
        // The two-half pipeline hides 35-55 % of the all-reduce time but costs GEMM efficiency (each half alone cannot fill the chip): measured with a
        // timed stand-in all-reduce (tools/mb_tp_overlap.py, 7B shards) it wins from rows x ranks >= 4096 on — TP=2 at 2048 rows, TP=8 at 1087 —
        // and loses below (TP=2 at 1087 rows: 17.7 vs 15.0 ms serialised).  LMX_TP_OVERLAP=2 forces it (tests), =0 switches it off.
        // output_hidden_states: rows [c0, c0 + tc) of entry e of the [L + 1][T][H] tuple
        auto hidden_rows = [&](int e) { return static_cast<char*>(hidden) + ((size_t)e * T + c0) * H * es; };
        // With the two-shot peer-to-peer all-reduce (p2p.hip: ~25 us per 8.9 MB message instead of ~114 us on a ring) the sums are 1.6 ms of a TP = 8 prefill:
        // splitting the chunk to hide them costs more GEMM efficiency than it can win, so the pipeline is only for the RCCL path (or forced).
        const bool big_p2p = p2p_big_usable((size_t)tc * H) && tc > P2P_MAX_ROWS && !ar_hook;
        if (tp_active && tp_overlap && tc >= 256 && !hidden && !attn_probs && (tp_overlap_force || (!big_p2p && (long)tc * std::max(cfg.tp_world, 1) >= 4096))) {
            // Tensor parallel: the chunk runs as two row halves so that the all-reduce of one half (comm stream) overlaps the
            // GEMMs / attention of the other (launch stream).  Half 1's causal attention sees half 0's keys: same-stream order.
            ensure_comm_stream();
            s->ensure_events();
            const int half0 = round_up((tc + 1) / 2, 128) < tc ? round_up((tc + 1) / 2, 128) : tc / 2;
            const int r0[2] = {0, half0}, rn[2] = {half0, tc - half0};
            auto reduce_async = [&](int i) {
                LMX_CHECK_HIP(hipEventRecord(s->ev_c[i], st));
                LMX_CHECK_HIP(hipStreamWaitEvent(comm_stream, s->ev_c[i], 0));
                { ProfScope ps(this, "prefill.allreduce", comm_stream); allreduce(rows(h, r0[i], H), (size_t)rn[i] * H, comm_stream); }
                LMX_CHECK_HIP(hipEventRecord(s->ev_r[i], comm_stream));
            };
            for (int l = 0; l < L; ++l) {
                for (int i = 0; i < 2; ++i) {
                    if (l > 0) LMX_CHECK_HIP(hipStreamWaitEvent(st, s->ev_r[i], 0));      // previous layer's MLP sum of this half
                    attn_block(l, r0[i], rn[i]);
                    reduce_async(i);
                }
                for (int i = 0; i < 2; ++i) {
                    LMX_CHECK_HIP(hipStreamWaitEvent(st, s->ev_r[i], 0));
                    mlp_block(l, r0[i], rn[i]);
                    reduce_async(i);
                }
            }
            for (int i = 0; i < 2; ++i) LMX_CHECK_HIP(hipStreamWaitEvent(st, s->ev_r[i], 0));
        } else {
            for (int l = 0; l < L; ++l) {
                if (hidden) LMX_CHECK_HIP(hipMemcpyAsync(hidden_rows(l), h, (size_t)tc * H * es, hipMemcpyDeviceToDevice, st));      // the layer's input rows
                attn_block(l, 0, tc);
                allreduce(h, (size_t)tc * H, st);
                mlp_block(l, 0, tc);
                { LMX_PROF_AR("prefill.allreduce"); allreduce(h, (size_t)tc * H, st); }
            }
            if (hidden) launch_rmsnorm(dt, h, final_norm, hidden_rows(L), tc, H, H, H, cfg.rms_eps, st);       // the tuple's last entry is normalised
        }
        const bool last_chunk = c0 + tc == T;
        // vocabulary-parallel head: this rank computes columns [v_off, v_off + V_l) of a zeroed row, the sum over ranks completes it
        const bool vsplit = V_l != V;
        if (logits_all && logits) {
            char* lc = static_cast<char*>(logits) + (size_t)c0 * V * es;
            launch_rmsnorm(dt, h, final_norm, x, tc, H, H, H, cfg.rms_eps, st);
            if (vsplit) LMX_CHECK_HIP(hipMemsetAsync(lc, 0, (size_t)tc * V * es, st));
            launch_gemm(dt, GemmArgs{x, lm_head, lc + (size_t)v_off * es, nullptr, nullptr, tc, V_l, H, H, H, V, 0, kActNone}, gv, st);
            gather_logits(lc, tc, st);
        }
        if (last_chunk && (greedy || (logits && !logits_all))) {
            const void* hl = static_cast<const char*>(h) + (size_t)(tc - 1) * H * es;
            void* dst = (logits && !logits_all) ? logits : last_logits;
            if (vsplit) LMX_CHECK_HIP(hipMemsetAsync(dst, 0, (size_t)V * es, st));
            { LMX_PROF_K("prefill.gemv.lm_head"); launch_gemv(dt, GemvArgs{hl, lm_head, static_cast<char*>(dst) + (size_t)v_off * es, nullptr, nullptr, final_norm, cfg.rms_eps, V_l, H, H, H, V, 0, kActNone}, 1, st); }
            gather_logits(dst, 1, st);
            if (greedy) {
                if (s->samp.temperature > 0.f) launch_sample(dt, dst, Vr, s->samp, s->d_nout, s->d_tok, nullptr, nullptr, st);
                else launch_argmax(dt, dst, Vr, s->d_tok, st);
                launch_log_token(s->d_tok, s->d_log, s->d_nout, s->log_cap, s->d_stop, st);
            }
        }
    }
    s->len += T;
    launch_set_state(s->d_len, s->len, s->d_tok, 0, 0, s->d_nout, -1, st);
}

// ---------------------------------------------------------------------------------------------------------------
// packed prefill of several sequences (serving: BASELINE configs 3 / 4 prefill many requests at once)
// ---------------------------------------------------------------------------------------------------------------
// The rows of all sequences are walked as ONE packed row block in pieces of `block_rows`: every linear of a piece is a single GEMM over the piece's
// rows (a 512-row chunk of one request reaches ~0.19 of the MFMA peak, 4096 packed rows ~0.4), RoPE / KV append / causal attention run per
// (sequence, row range) segment against that sequence's own cache, as in the chunked prefill of one sequence.  The row arithmetic is unchanged, so a
// sequence's result does not depend on who shares the piece with it (up to the split-K choice of the N = hidden GEMMs, which changes the summation
// order of their fp32 partial sums).  Each sequence's last row goes through the lm_head and its pick, exactly as Model::prefill does.
void Model::prefill_multi(Seq* const* seqs, const void* const* embeds, const int* Ts, int n, int block_rows, bool greedy, hipStream_t st) {
    LMX_REQUIRE(n >= 1 && seqs && embeds && Ts, "prefill_multi: bad arguments");
    LMX_REQUIRE(rope != nullptr, "rope table not set");
    long total = 0;
    for (int i = 0; i < n; ++i) {
        LMX_REQUIRE(seqs[i] && seqs[i]->m == this && embeds[i] && Ts[i] > 0, "prefill_multi: bad sequence / input");
        LMX_REQUIRE(seqs[i]->len + Ts[i] <= s_max, "prefill_multi: a sequence would exceed the KV-cache capacity (max_position)");
        for (int j = 0; j < i; ++j) LMX_REQUIRE(seqs[j] != seqs[i], "prefill_multi: the same sequence appears twice");
        seqs[i]->last_stream = st; seqs[i]->used = true;
        total += Ts[i];
    }
    if (block_rows <= 0 || block_rows > total) block_rows = (int)total;
    const int dt = cfg.dtype;
    Seq* s0 = seqs[0];                                   // owner of the piece workspace and the split-K scratch of this call
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t o_h = carve((size_t)block_rows * H * es), o_x = carve((size_t)block_rows * H * es), o_qkv = carve((size_t)block_rows * qkv_n * es),
                 o_attn = carve((size_t)block_rows * nh_l * D * es), o_act = carve((size_t)block_rows * I_l * es),
                 o_aws = carve(dt == kF32 ? decode_attn_ws_floats(block_rows, nh_l, 1, D) * 4 : 0), o_log = carve((size_t)V * es);
    if (off > s0->pws.bytes) { LMX_CHECK_HIP(hipStreamSynchronize(st)); s0->pws.ensure(off); }
    char* W = s0->pws.as<char>();
    void *h = W + o_h, *x = W + o_x, *qkv = W + o_qkv, *attn = W + o_attn, *act = W + o_act, *last_logits = W + o_log;
    float* aws = reinterpret_cast<float*>(W + o_aws);
    const float scale = 1.f / sqrtf((float)D);
    const int gv = cfg.gemm_variant;
    const bool lead = cfg.tp_rank == 0;
    if (dt != kF32 && !s0->skw.p) {
        LMX_CHECK_HIP(hipStreamSynchronize(st));
        s0->skw.ensure((size_t)256 * 256 * 256 * sizeof(float));
    }
    auto with_scratch = [&](GemmArgs g) { g.skw = s0->skw.p; return g; };
    auto rows = [&](void* base, long r0, size_t width) { return static_cast<char*>(base) + (size_t)r0 * width * es; };
    struct Segment { int seq, src0, dst0, n, pos0; };    // rows [src0, src0 + n) of sequence `seq` sit at rows [dst0, ...) of the piece; first position pos0
    int cur = 0, cur_done = 0;                           // next sequence / rows of it already consumed
    const bool vsplit = V_l != V;
    while (cur < n) {
        std::vector<Segment> seg;
        int fill = 0;
        while (cur < n && fill < block_rows) {
            const int take = std::min(block_rows - fill, Ts[cur] - cur_done);
            seg.push_back(Segment{cur, cur_done, fill, take, seqs[cur]->len + cur_done});
            fill += take; cur_done += take;
            if (cur_done == Ts[cur]) { ++cur; cur_done = 0; }
        }
        for (const Segment& g : seg)
            LMX_CHECK_HIP(hipMemcpyAsync(rows(h, g.dst0, H), static_cast<const char*>(embeds[g.seq]) + (size_t)g.src0 * H * es, (size_t)g.n * H * es, hipMemcpyDeviceToDevice, st));
        for (int l = 0; l < L; ++l) {
            const DecLayerW& w = dec[l];
            { LMX_PROF("prefill.rmsnorm"); launch_rmsnorm(dt, h, w.ln1, x, fill, H, H, H, cfg.rms_eps, st); }
            { LMX_PROF_K("prefill.gemm.qkv"); launch_gemm(dt, with_scratch(GemmArgs{x, w.wqkv, qkv, nullptr, nullptr, fill, qkv_n, H, H, H, qkv_n, 0, kActNone}), gv, st); }
            for (const Segment& g : seg) {
                Seq* s = seqs[g.seq];
                void* kc = s->kc.as<char>() + (size_t)l * s->layer_stride;
                void* vt = s->vt.as<char>() + (size_t)l * s->layer_stride;
                void *qr = rows(qkv, g.dst0, qkv_n), *ar = rows(attn, g.dst0, (size_t)nh_l * D);
                { LMX_PROF("prefill.rope_kv"); launch_rope_kv(dt, D, RopeKvArgs{qr, kc, vt, rope, nullptr, g.pos0, g.n, qkv_n, nh_l, nkv_l, s_max}, st); }
                LMX_PROF("prefill.attn");
                if (dt == kF32) launch_decode_attn(dt, D, DecodeAttnArgs{qr, ar, kc, vt, nullptr, g.pos0, g.n, 0, 1, qkv_n, nh_l * D, nh_l, nkv_l, s_max, 1, scale, aws}, st);
                else launch_flash_prefill(dt, D, FlashArgs{qr, ar, kc, vt, g.n, g.pos0 + g.n, g.pos0, qkv_n, nh_l * D, nh_l, nkv_l, s_max, scale, 1}, st);
            }
            { LMX_PROF_K("prefill.gemm.o"); launch_gemm(dt, with_scratch(GemmArgs{attn, w.wo, h, nullptr, lead ? h : nullptr, fill, H, nh_l * D, nh_l * D, nh_l * D, H, H, kActNone}), gv, st); }
            allreduce(h, (size_t)fill * H, st);
            { LMX_PROF("prefill.rmsnorm"); launch_rmsnorm(dt, h, w.ln2, x, fill, H, H, H, cfg.rms_eps, st); }
            { LMX_PROF_K("prefill.gemm.gate_up"); launch_gemm(dt, with_scratch(GemmArgs{x, w.wgu, act, nullptr, nullptr, fill, 2 * I_l, H, H, H, I_l, 0, kActSiluMul}), gv, st); }
            { LMX_PROF_K("prefill.gemm.down"); launch_gemm(dt, with_scratch(GemmArgs{act, w.wd, h, nullptr, lead ? h : nullptr, fill, H, I_l, I_l, I_l, H, H, kActNone}), gv, st); }
            { LMX_PROF_AR("prefill.allreduce"); allreduce(h, (size_t)fill * H, st); }
        }
        // sequences that END in this piece: lm_head on their last row + the pick
        for (const Segment& g : seg) {
            if (g.src0 + g.n != Ts[g.seq] || !greedy) continue;
            Seq* s = seqs[g.seq];
            const void* hl = rows(h, g.dst0 + g.n - 1, H);
            if (vsplit) LMX_CHECK_HIP(hipMemsetAsync(last_logits, 0, (size_t)V * es, st));
            { LMX_PROF_K("prefill.gemv.lm_head"); launch_gemv(dt, GemvArgs{hl, lm_head, static_cast<char*>(last_logits) + (size_t)v_off * es, nullptr, nullptr, final_norm, cfg.rms_eps, V_l, H, H, H, V, 0, kActNone}, 1, st); }
            gather_logits(last_logits, 1, st);
            if (s->samp.temperature > 0.f) launch_sample(dt, last_logits, Vr, s->samp, s->d_nout, s->d_tok, nullptr, nullptr, st);
            else launch_argmax(dt, last_logits, Vr, s->d_tok, st);
            launch_log_token(s->d_tok, s->d_log, s->d_nout, s->log_cap, s->d_stop, st);
        }
    }
    for (int i = 0; i < n; ++i) {
        seqs[i]->len += Ts[i];
        launch_set_state(seqs[i]->d_len, seqs[i]->len, seqs[i]->d_tok, 0, 0, seqs[i]->d_nout, -1, st);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// decode
// ---------------------------------------------------------------------------------------------------------------
// A bounded in-launch wait of THIS sequence's decode attention timed out (contention from other streams / processes, preemption, a profiler): the step that
// raised the word appended a zero k / v row and merged incomplete partials, so the sequence's output from there on is not valid — the caller's request fails,
// nobody else's.  Reported once: the word and the arrival tickets (which the kernel leaves alone after a timeout) are cleared, and the model stops taking the
// split-q form, whose waiters depend on co-residency with their own launch's projection.
void Seq::check_wait_status(hipStream_t st) {
    if (!h_status || __atomic_load_n(h_status, __ATOMIC_RELAXED) == 0) return;
    __atomic_store_n(h_status, 0u, __ATOMIC_RELAXED);
    m->opt_splitq = false;
    (void)hipMemsetAsync(d_cnt, 0, (size_t)m->nh_l * sizeof(int), st);
    throw Error{"decode step: a bounded in-launch wait of this sequence's attention timed out (decode_attn.hip); its output from that step on is not valid. "
                "The model now uses the separate attention launch (decode_splitq = 0); other sequences are unaffected"};
}

// One decode step of one sequence = per layer {q|k|v projection (+RMSNorm), RoPE + KV append + attention, o_proj (+residual), gate|up (+RMSNorm, SiLU*mul),
// down (+residual)} + lm_head (+final norm) + the pick, as plain stream launches (HF5:models/llama/modeling_llama.py:367-418 for one new token through
// llava_arch.py:103-112).  16-bit models take the split-q form (decode_attn.hip): q alone, then ONE launch holding the attention workgroups and the k|v
// projection, so that the attention's latency chain runs under the k|v weight stream — bit-identical to the three-launch form.
void Model::decode_step_launch(Seq* s, hipStream_t st, int64_t* id_out) {
    const int dt = cfg.dtype;
    const bool lead = cfg.tp_rank == 0;
    const float scale = 1.f / sqrtf((float)D);
    // s->d_h holds the embedding of the token to feed: put there by decode() / decode_batch() before the first step and by the
    // fused pick kernel at the end of every step
    const bool attn16 = (dt == kBF16 || dt == kF16) && s_max % 128 == 0 && s_max / 128 <= 32;      // the chunked launch with the position by value
    const int q_n = nh_l * D, kv_n = 2 * nkv_l * D;
    for (int l = 0; l < L; ++l) {
        const DecLayerW& w = dec[l];
        void* kc = s->kc.as<char>() + (size_t)l * s->layer_stride;
        void* vt = s->vt.as<char>() + (size_t)l * s->layer_stride;
        DecAttnArgs a{};
        a.qkv = s->d_qkv; a.attn = s->d_attn; a.kc = kc; a.vt = vt; a.rope = rope; a.aws = s->d_aws; a.cnt = s->d_cnt;
        // only the 128-key chunks that exist are launched: the host mirrors the position (s->len == *d_len while this step is queued)
        a.pos = s->len; a.n_split = s->len / 128 + 1; a.nh = nh_l; a.nkv = nkv_l; a.s_max = s_max; a.scale = scale;
        a.status = s->d_status;                      // both forms: the merger's wait for the other chunks is bounded in either
        // rows [q_n, q_n + kv_n) of the fused q|k|v weight: the k | v projection of the split-q form
        const GemvArgs gkv{s->d_h, static_cast<const char*>(w.wqkv) + (size_t)q_n * H * es, nullptr, nullptr, nullptr, w.ln1, cfg.rms_eps, kv_n, H, H, H, kv_n, 0, kActNone};
        if (attn16 && splitq_allowed() && decode_kv_attn_applies(dt, D, gkv)) {
            { LMX_PROF_K("decode.gemv.q"); launch_gemv(dt, GemvArgs{s->d_h, w.wqkv, s->d_qkv, nullptr, nullptr, w.ln1, cfg.rms_eps, q_n, H, H, H, q_n, 0, kActNone}, 1, st); }
            a.kv_gran = s->kv_gran.as<unsigned long long>(); a.tag = s->attn_tag;
            if (debug_splitq_timeout.load(std::memory_order_relaxed) > 0 && debug_splitq_timeout.fetch_sub(1) > 0) a.pub_tag = a.tag ^ 0x40000000u;
            s->attn_tag = s->attn_tag >= 0xfffffff0u ? 1u : s->attn_tag + 1;
            { LMX_PROF_K("decode.kv_attn"); launch_decode_kv_attn(dt, D, a, gkv, st); }
        } else {
            { LMX_PROF_K("decode.gemv.qkv"); launch_gemv(dt, GemvArgs{s->d_h, w.wqkv, s->d_qkv, nullptr, nullptr, w.ln1, cfg.rms_eps, qkv_n, H, H, H, qkv_n, 0, kActNone}, 1, st); }
            LMX_PROF("decode.attn");
            if (attn16) launch_decode_attn_step(dt, D, a, st);
            else {
                DecodeFusedArgs fa{s->d_qkv, kc, vt, rope, s->d_len, nh_l, nkv_l, s_max, s->len / 128 + 1, scale, s->d_aws, s->d_cnt, s->d_attn};
                launch_decode_fused(dt, D, fa, st);
            }
        }
        { LMX_PROF_K("decode.gemv.o"); launch_gemv(dt, GemvArgs{s->d_attn, w.wo, s->d_h, nullptr, lead ? s->d_h : nullptr, nullptr, 0.f, H, nh_l * D, nh_l * D, nh_l * D, H, H, kActNone}, 1, st); }
        { LMX_PROF_AR("decode.allreduce"); allreduce(s->d_h, (size_t)H, st); }
        { LMX_PROF_K("decode.gemv.gate_up"); launch_gemv(dt, GemvArgs{s->d_h, w.wgu, s->d_act, nullptr, nullptr, w.ln2, cfg.rms_eps, 2 * I_l, H, H, H, I_l, 0, kActSiluMul}, 1, st); }
        { LMX_PROF_K("decode.gemv.down"); launch_gemv(dt, GemvArgs{s->d_act, w.wd, s->d_h, nullptr, lead ? s->d_h : nullptr, nullptr, 0.f, H, I_l, I_l, I_l, H, H, kActNone}, 1, st); }
        allreduce(s->d_h, (size_t)H, st);
    }
    if (V_l != V) LMX_CHECK_HIP(hipMemsetAsync(s->d_logits, 0, (size_t)V * es, st));
    { LMX_PROF_K("decode.gemv.lm_head"); launch_gemv(dt, GemvArgs{s->d_h, lm_head, static_cast<char*>(s->d_logits) + (size_t)v_off * es, nullptr, nullptr, final_norm, cfg.rms_eps, V_l, H, H, H, V, 0, kActNone}, 1, st); }
    { ProfScope ps_ag(this, V_l != V ? "decode.allgather.logits" : nullptr, st); gather_logits(s->d_logits, 1, st); }
    {
        LMX_PROF("decode.argmax");      // pick (argmax | draw) + *len += 1 + token log + next token's embedding row -> d_h, one launch
        const SeqStateRef r{s->d_len, s->d_nout, s->d_tok, s->d_log, s->log_cap, 0, s->samp, s->d_stop};
        launch_argmax_advance_batch(dt, s->d_logits, Vr, V, nullptr, &r, 1, id_out, embed, s->d_h, H, st);      // id_out: the picked id, or -1 once the stop rule has fired
    }
}

void Model::seq_copy(Seq* dst, const Seq* src, hipStream_t st) {
    LMX_REQUIRE(dst && src && dst->m == this && src->m == this && dst != src, "seq_copy: bad sequences");
    LMX_REQUIRE(src->len > 0, "seq_copy: the source has no context");
    const size_t n = (size_t)src->len;
    // K rows [L][kv_head][s_max][D]: n * D contiguous elements per (layer, head);  V^T [L][kv_head][D][s_max]: n contiguous elements per (layer, head, d)
    LMX_CHECK_HIP(hipMemcpy2DAsync(dst->kc.p, (size_t)s_max * D * es, src->kc.p, (size_t)s_max * D * es, n * D * es, (size_t)L * nkv_l, hipMemcpyDeviceToDevice, st));
    LMX_CHECK_HIP(hipMemcpy2DAsync(dst->vt.p, (size_t)s_max * es, src->vt.p, (size_t)s_max * es, n * es, (size_t)L * nkv_l * D, hipMemcpyDeviceToDevice, st));
    dst->len = src->len;
    launch_set_state(dst->d_len, dst->len, dst->d_tok, 0, 0, dst->d_nout, 0, st);
    dst->last_stream = st; dst->used = true;
}

void Model::decode(Seq* s, int64_t token, int n_steps, void* logits, bool greedy, hipStream_t st) {
    LMX_REQUIRE(n_steps >= 1, "decode: n_steps must be >= 1");
    LMX_REQUIRE(greedy || n_steps == 1, "decode: chained steps need greedy sampling on the device");
    LMX_REQUIRE(s->len + n_steps <= s_max, "decode: sequence would exceed the KV-cache capacity (max_position)");
    LMX_REQUIRE(s->len > 0, "decode before prefill");
    LMX_REQUIRE(token < V, "token id out of range");
    s->last_stream = st; s->used = true;
    if (token >= 0) launch_set_state(s->d_len, -1, s->d_tok, token, 1, s->d_nout, -1, st);
    // Plain stream launches: the ~160 kernels of a step average >20 us each against ~3.5 us of host launch cost, so the
    // host runs far ahead of the GPU; a captured hipGraph measured no faster (3.57 vs 3.56 ms/token, profiles/EXPERIMENTS.md)
    // and stream capture is not safe next to other threads using the legacy stream (model_worker runs 5 request threads).
    { LMX_PROF("decode.embed"); launch_gather_token(cfg.dtype, s->d_tok, embed, s->d_h, H, V, st); }
    for (int i = 0; i < n_steps; ++i) {
        decode_step_launch(s, st);
        s->len += 1;
    }
    if (logits) LMX_CHECK_HIP(hipMemcpyAsync(logits, s->d_logits, (size_t)V * es, hipMemcpyDeviceToDevice, st));
}

// ---------------------------------------------------------------------------------------------------------------
// decode batch (continuous batching): several sequences, each with its own KV cache and position, advance
// one token per step and share ONE pass over the weights.
// ---------------------------------------------------------------------------------------------------------------
// Decode-batch weights: the skinny MFMA kernel loads its A fragments per lane, so from the [N][K] layout every 16-lane group of a load
// touches 16 different rows (uncoalesced: 3.3-4.4 TB/s).  With 288 GB of HBM the decoder weights are simply kept a second time in
// fragment order (13 GB at 7B), where each load instruction of a wave reads 1 KiB of contiguous memory.  LMX_BATCH_SWIZZLE=0 disables.
void Model::ensure_batch_weights(hipStream_t st) {
    std::lock_guard<std::mutex> lk(mu);
    if (batch_weights_ready) return;
    batch_weights_ready = true;
    const char* e = getenv("LMX_BATCH_SWIZZLE");
    if (cfg.dtype == kF32 || (e && atoi(e) == 0)) return;
    auto mk = [&](const void* W, int N, int K) -> void* {
        if (!W) return nullptr;
        void* d = alloc_weight(skinny_swizzled_bytes(N, K, es));
        launch_skinny_swizzle(cfg.dtype, W, K, d, N, K, st);
        return d;
    };
    for (auto& w : dec) {
        w.sw_qkv = mk(w.wqkv, qkv_n, H);
        w.sw_o = mk(w.wo, H, nh_l * D);
        w.sw_gu = mk(w.wgu, 2 * I_l, H);
        w.sw_d = mk(w.wd, H, I_l);
    }
    sw_lm_head = mk(lm_head, V_l, H);
    LMX_CHECK_HIP(hipStreamSynchronize(st));
}

Batch::Batch(Model* mm, int capacity) : m(mm), cap(capacity) {
    LMX_REQUIRE(capacity >= 1 && capacity <= 256, "batch capacity must be 1..256");
    const int es = m->es;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t o_h = carve((size_t)cap * m->H * es), o_x = carve((size_t)cap * m->H * es), o_qkv = carve((size_t)cap * m->qkv_n * es),
                 o_attn = carve((size_t)cap * m->nh_l * m->D * es), o_act = carve((size_t)cap * m->I_l * es), o_log = carve((size_t)cap * m->V * es);
    ws.ensure(off, true);
    char* W = ws.as<char>();
    h = W + o_h; x = W + o_x; qkv = W + o_qkv; attn = W + o_attn; act = W + o_act; logits = W + o_log;
    const size_t attn_bytes = sizeof(DecodeFusedSeq) * (size_t)m->L * cap, state_bytes = sizeof(SeqStateRef) * (size_t)cap;
    tab.ensure(attn_bytes + state_bytes, true);
    if (m->cfg.dtype != kF32) {                     // K-sliced linears of a decode batch: o_proj / down_proj (N = H) and q|k|v
        const int nmax = std::max(m->H, m->qkv_n);
        sk_scratch.ensure(skinny_scratch_bytes(nmax)); sk_cnt.ensure((size_t)(nmax / 64 + 1) * sizeof(int), true);
    }
    d_attn_tab = tab.as<DecodeFusedSeq>();
    d_state_tab = reinterpret_cast<SeqStateRef*>(tab.as<char>() + attn_bytes);
    host_tab.resize(attn_bytes + state_bytes);
}

// (re)build the device tables when the membership changed since the last step
void Batch::bind(Seq* const* seqs, int n, hipStream_t st) {
    bool same = (int)members.size() == n;
    for (int i = 0; same && i < n; ++i) same = members[(size_t)i] == seqs[i]->uid;
    if (same) return;
    const int L = m->L;
    DecodeFusedSeq* at = reinterpret_cast<DecodeFusedSeq*>(host_tab.data());
    SeqStateRef* stt = reinterpret_cast<SeqStateRef*>(host_tab.data() + sizeof(DecodeFusedSeq) * (size_t)L * cap);
    for (int i = 0; i < n; ++i) {
        Seq* s = seqs[i];
        for (int l = 0; l < L; ++l)
            at[(size_t)l * cap + i] = DecodeFusedSeq{s->kc.as<char>() + (size_t)l * s->layer_stride, s->vt.as<char>() + (size_t)l * s->layer_stride,
                                                     s->d_len, s->d_aws, s->d_cnt};
        stt[i] = SeqStateRef{s->d_len, s->d_nout, s->d_tok, s->d_log, s->log_cap, 0, s->samp, s->d_stop};
    }
    // the previous step's kernels may still be reading the old tables on this stream, and the host image is reused
    LMX_CHECK_HIP(hipStreamSynchronize(st));
    LMX_CHECK_HIP(hipMemcpyAsync(tab.p, host_tab.data(), host_tab.size(), hipMemcpyHostToDevice, st));
    LMX_CHECK_HIP(hipStreamSynchronize(st));
    members.resize((size_t)n);
    for (int i = 0; i < n; ++i) members[(size_t)i] = seqs[i]->uid;
}

void Model::decode_batch(Batch* b, Seq* const* seqs, int n, const int64_t* tokens, int n_steps, void* logits, bool greedy, int64_t* ids_out_host, hipStream_t st,
                         bool sync_ids) {
    LMX_REQUIRE(b && b->m == this, "decode_batch: batch belongs to another model");
    LMX_REQUIRE(n >= 1 && n <= b->cap, "decode_batch: number of sequences exceeds the batch capacity");
    LMX_REQUIRE(n_steps >= 1, "decode_batch: n_steps must be >= 1");
    LMX_REQUIRE(greedy || n_steps == 1, "decode_batch: chained steps need greedy sampling on the device");
    for (int i = 0; i < n; ++i) {
        Seq* s = seqs[i];
        LMX_REQUIRE(s && s->m == this, "decode_batch: sequence belongs to another model");
        LMX_REQUIRE(s->len > 0, "decode before prefill");
        LMX_REQUIRE(s->len + n_steps <= s_max, "decode_batch: a sequence would exceed the KV-cache capacity (max_position)");
        for (int j = 0; j < i; ++j) LMX_REQUIRE(seqs[j] != s, "decode_batch: the same sequence appears twice");
        if (tokens) LMX_REQUIRE(tokens[i] < V, "token id out of range");
        s->last_stream = st; s->used = true;
    }
    if (n > 1) ensure_batch_weights(st);
    b->bind(seqs, n, st);
    if (n_steps > b->ids_steps) { LMX_CHECK_HIP(hipStreamSynchronize(st)); b->ids.ensure((size_t)n_steps * b->cap * 8); b->ids_steps = n_steps; }
    int64_t* d_ids = b->ids.as<int64_t>();
    if (tokens)
        for (int i = 0; i < n; ++i)
            if (tokens[i] >= 0) launch_set_state(seqs[i]->d_len, -1, seqs[i]->d_tok, tokens[i], 1, seqs[i]->d_nout, -1, st);

    const int dt = cfg.dtype;
    const bool lead = cfg.tp_rank == 0;
    const float scale = 1.f / sqrtf((float)D);
    // A batch row-block linear y = act(norm(x) W^T) (+ residual): rmsnorm launch + the skinny MFMA kernel on the fragment-order weight
    // copy for up to 32 rows (7B, per step: 2 rows ~3.3 ms, 8 rows 4.2 ms, 32 rows 8.8 ms; the multi-row GEMV chain measured 3.75 ms at
    // 2 rows and 5.05 ms at 4, so it is not used); fp32 verification engine / larger batches: prefill GEMM family.
    auto linear = [&](const void* x_in, const void* norm_w, void* x_normed, GemmArgs g, const void* wsw) {
        if (norm_w) {
            LMX_PROF("decode_batch.rmsnorm");
            launch_rmsnorm(dt, x_in, norm_w, x_normed, g.M, g.K, g.ldx, g.K, cfg.rms_eps, st);
            g.X = x_normed; g.ldx = g.K;
        }
        LMX_PROF("decode_batch.linear");
        if (dt != kF32 && g.M <= 32) { g.Wsw = wsw; g.skw = b->sk_scratch.p; g.sk_cnt = b->sk_cnt.as<int>(); launch_skinny_gemm(dt, g, st); } else launch_gemm(dt, g, cfg.gemm_variant, st);
    };
    if (n == 1) {
        // a lone member takes the single-sequence step (GEMV with fused RMSNorm: fewer launches, full-rate weight stream)
        Seq* s = seqs[0];
        launch_gather_token(dt, s->d_tok, embed, s->d_h, H, V, st);
        for (int step = 0; step < n_steps; ++step) {
            decode_step_launch(s, st, d_ids + (size_t)step * b->cap);      // a lone member reports -1 after a device-side stop exactly like the members of a larger batch
            s->len += 1;
        }
        if (logits) LMX_CHECK_HIP(hipMemcpyAsync(logits, s->d_logits, (size_t)V * es, hipMemcpyDeviceToDevice, st));
    } else {
    { LMX_PROF("decode_batch.embed"); launch_gather_tokens_batch(dt, b->d_state_tab, n, embed, b->h, H, V, st); }
    for (int step = 0; step < n_steps; ++step) {
        bool x_ready = false;          // b->x already holds the normalised rows of the NEXT linear (written by the tensor-parallel all-reduce's launch: allreduce_norm)
        for (int l = 0; l < L; ++l) {
            const DecLayerW& w = dec[l];
            linear(x_ready ? b->x : b->h, x_ready ? nullptr : w.ln1, b->x, GemmArgs{x_ready ? b->x : b->h, w.wqkv, b->qkv, nullptr, nullptr, n, qkv_n, H, H, H, qkv_n, 0, kActNone}, w.sw_qkv);
            {
                LMX_PROF("decode_batch.attn");
                int max_len = 0;
                for (int i = 0; i < n; ++i) max_len = std::max(max_len, seqs[i]->len);
                // live 128-key chunks of the longest member (the host mirrors every position); shorter members' surplus workgroups only take their ticket
                DecodeFusedArgs a{b->qkv, nullptr, nullptr, rope, nullptr, nh_l, nkv_l, s_max, max_len / 128 + 1, scale, nullptr, nullptr, b->attn};
                a.tab = b->d_attn_tab + (size_t)l * b->cap; a.n_seq = n; a.qkv_stride = qkv_n; a.o_stride = nh_l * D;
                launch_decode_fused(dt, D, a, st);
            }
            linear(b->attn, nullptr, nullptr, GemmArgs{b->attn, w.wo, b->h, nullptr, lead ? b->h : nullptr, n, H, nh_l * D, nh_l * D, nh_l * D, H, H, kActNone}, w.sw_o);
            { LMX_PROF_AR("decode_batch.allreduce"); x_ready = allreduce_norm(b->h, n, w.ln2, b->x, st); }
            linear(x_ready ? b->x : b->h, x_ready ? nullptr : w.ln2, b->x, GemmArgs{x_ready ? b->x : b->h, w.wgu, b->act, nullptr, nullptr, n, 2 * I_l, H, H, H, I_l, 0, kActSiluMul}, w.sw_gu);
            linear(b->act, nullptr, nullptr, GemmArgs{b->act, w.wd, b->h, nullptr, lead ? b->h : nullptr, n, H, I_l, I_l, I_l, H, H, kActNone}, w.sw_d);
            x_ready = allreduce_norm(b->h, n, l + 1 < L ? dec[l + 1].ln1 : final_norm, b->x, st);       // the next layer's input norm (or the final one) rides in the same launch
        }
        if (V_l != V) LMX_CHECK_HIP(hipMemsetAsync(b->logits, 0, (size_t)n * V * es, st));
        linear(x_ready ? b->x : b->h, x_ready ? nullptr : final_norm, b->x, GemmArgs{x_ready ? b->x : b->h, lm_head, static_cast<char*>(b->logits) + (size_t)v_off * es, nullptr, nullptr, n, V_l, H, H, H, V, 0, kActNone}, sw_lm_head);
        gather_logits(b->logits, n, st);
        // pick + advance + the picked tokens' embedding rows -> b->h (input of the next step), one launch
        { LMX_PROF("decode_batch.argmax"); launch_argmax_advance_batch(dt, b->logits, Vr, V, b->d_state_tab, nullptr, n, d_ids + (size_t)step * b->cap, embed, b->h, H, st); }
        for (int i = 0; i < n; ++i) seqs[i]->len += 1;
    }
    }
    if (logits && n > 1) LMX_CHECK_HIP(hipMemcpyAsync(logits, b->logits, (size_t)n * V * es, hipMemcpyDeviceToDevice, st));
    if (ids_out_host) {       // [n_steps][n], after the stream drained
        LMX_CHECK_HIP(hipMemcpy2DAsync(ids_out_host, (size_t)n * 8, d_ids, (size_t)b->cap * 8, (size_t)n * 8, (size_t)n_steps, hipMemcpyDeviceToHost, st));
        if (sync_ids) {
            LMX_CHECK_HIP(hipStreamSynchronize(st));
            // a member whose id rule fired reports -1 from then on and its device position stands still: bring the host mirror back to it
            for (int i = 0; i < n; ++i) {
                int live = 0;
                for (int step = 0; step < n_steps; ++step) live += ids_out_host[(size_t)step * n + i] >= 0;
                seqs[i]->len -= n_steps - live;
            }
        }
    }
}

}  // namespace lmx
