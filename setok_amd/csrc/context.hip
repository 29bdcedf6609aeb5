// context.hip — the host-language-neutral entry of the path: setok_create / setok_load_weight / setok_weights_ready / setok_encode.
//
// `SetokTokenizer.forward` (src/model/setok/tokenizer.py:157-182) — tower (clip_encoder.py:50-62: HF CLIP ViT hidden_states[select_layer],
// feature_select) -> + PositionalEncoding2D (:164-168) -> cluster_dpc_knn (:174) -> group_encoding (:177-178) -> inter_encoder (:179) ->
// out (:180) — as ONE C call on caller-allocated buffers, so that a host in any language binds the path without re-writing the ~400 lines
// of orchestration that setok_amd/clip_encoder.py and setok_amd/tokenizer.py used to hold.  Everything below is a sequence of this
// library's own entry points (setok_linear, setok_attention, ...) on the caller's stream; the context owns only the weights (device
// copies in compute layout: fused q|k|v, folded LayerNorms, padded patch matrix) and one pinned host word array for the token counts.
//
// The one host synchronisation of the path lives here, exactly where the Python host had it: after the clustering the per-image token
// counts L_i are read back (B ints) because they size the ragged stages (sum L_i rows through inter_encoder / out).
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

namespace {

struct DevTensor {
    void* p = nullptr;
    int dtype = SETOK_F32;
    std::vector<int64_t> shape;
    int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};

struct Folded { void* wg = nullptr; float* colsum = nullptr; float* bias = nullptr; float* frag = nullptr; };

struct TowerLayer {
    void* wqkv = nullptr; float* bqkv = nullptr;
    const void *wo, *w1, *w2;
    const float *bo, *b1, *b2, *ln1w, *ln1b, *ln2w, *ln2b;
    Folded qkv_ln, fc1_ln;
};

struct BlockW {
    const float *n1w, *n1b, *n2w, *n2b;
    struct Attn { const void* wqkv; const float* bqkv; const void* wproj; const float* bproj; };
    std::vector<Attn> attn;
    const void *w1, *w2;
    const float *b1, *b2;
};

template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ in, TO* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (TO)(float)in[i];
}

int elem_size(int dtype) { return dtype == SETOK_BF16 ? 2 : 4; }

}  // namespace

struct setok_ctx {
    setok_config cfg;
    int N, T, kpad, Dh, layers_run;
    bool ready = false, fold = false;
    std::map<std::string, DevTensor> w;                  // by the reference's state-dict name
    std::vector<void*> owned;                            // every device allocation of the context
    std::vector<TowerLayer> tower;
    void* wp = nullptr;                                  // patch conv as a (C, kpad) GEMM operand
    BlockW inner, inter;
    int32_t* counts_pinned = nullptr;
    int counts_cap = 0;
    hipEvent_t counts_ready = nullptr;            // recorded behind the counts' device-to-host copy (setok_encode's convenience form)
    char err[512];
};

namespace {

int ctx_fail(setok_ctx* c, int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof(c->err), fmt, ap);
    va_end(ap);
    return setok_fail(code, "%s", c->err);
}

void* ctx_alloc(setok_ctx* c, size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
    c->owned.push_back(p);
    return p;
}

// the stored name: strip the wrappers the reference's checkpoints put in front of the tower keys
std::string canonical(const char* name) {
    std::string s(name);
    for (const char* pre : {"image_feature_encoder.vision_tower.", "vision_tower.", "vision_model."}) {
        const size_t n = strlen(pre);
        if (s.compare(0, n, pre) == 0) s = "tower." + s.substr(n);
    }
    if (s.compare(0, 19, "tower.vision_model.") == 0) s = "tower." + s.substr(19);          // transformers 4.x spelling under the tower prefix
    return s;
}

const DevTensor* find(setok_ctx* c, const std::string& name) {
    auto it = c->w.find(name);
    return it == c->w.end() ? nullptr : &it->second;
}

}  // namespace

extern "C" int setok_create(const setok_config* cfg, setok_ctx** out) {
    SETOK_CHECK_ARG(cfg && out, "setok_create: null argument");
    SETOK_CHECK_ARG(cfg->dtype == SETOK_BF16 || cfg->dtype == SETOK_F32, "setok_create: bad dtype %d", cfg->dtype);
    SETOK_CHECK_ARG(cfg->image_size > 0 && cfg->patch_size > 0 && cfg->image_size % cfg->patch_size == 0, "setok_create: image_size %d / patch_size %d",
                    cfg->image_size, cfg->patch_size);
    SETOK_CHECK_ARG(cfg->hidden_size > 0 && cfg->num_attention_heads > 0 && cfg->hidden_size % cfg->num_attention_heads == 0 &&
                    cfg->nheads > 0 && cfg->hidden_size % cfg->nheads == 0, "setok_create: hidden_size %d not divisible by the head counts", cfg->hidden_size);
    const int L = cfg->num_hidden_layers;
    const int idx = cfg->select_layer >= 0 ? cfg->select_layer : L + 1 + cfg->select_layer;      // index into hidden_states (clip_encoder.py:41)
    SETOK_CHECK_ARG(idx >= 0 && idx <= L, "setok_create: select_layer %d out of range for %d layers (IndexError in the reference)", cfg->select_layer, L);
    setok_ctx* c = new setok_ctx();
    c->cfg = *cfg;
    const int g = cfg->image_size / cfg->patch_size;
    c->N = g * g; c->T = c->N + 1;
    const int kalign = cfg->dtype == SETOK_BF16 ? 64 : 16;
    c->kpad = (3 * cfg->patch_size * cfg->patch_size + kalign - 1) / kalign * kalign;
    c->Dh = cfg->hidden_size / cfg->num_attention_heads;
    c->layers_run = idx;                                                                          // select_layer = -2 skips the last encoder layer
    c->fold = cfg->dtype == SETOK_BF16 && cfg->fold_layernorm && cfg->hidden_size % 64 == 0 && cfg->intermediate_size % 64 == 0;
    c->err[0] = 0;
    *out = c;
    return SETOK_OK;
}

extern "C" void setok_destroy(setok_ctx* c) {
    if (!c) return;
    for (void* p : c->owned) (void)hipFree(p);
    if (c->counts_pinned) (void)hipHostFree(c->counts_pinned);
    if (c->counts_ready) (void)hipEventDestroy(c->counts_ready);
    delete c;
}

extern "C" const char* setok_ctx_error(const setok_ctx* c) { return c ? c->err : ""; }

// Matrices are kept in the compute dtype, vectors (biases, LayerNorm affine, class embedding excepted) in fp32.
extern "C" int setok_load_weight(setok_ctx* c, void* stream, const char* name, const void* ptr, int dtype, const int64_t* shape, int ndim) {
    SETOK_CHECK_ARG(c && name && ptr && shape && ndim >= 1 && ndim <= 4, "setok_load_weight: bad argument");
    SETOK_CHECK_ARG(dtype == SETOK_BF16 || dtype == SETOK_F32, "setok_load_weight(%s): bad dtype %d", name, dtype);
    const std::string key = canonical(name);
    DevTensor t;
    t.shape.assign(shape, shape + ndim);
    const int64_t n = t.numel();
    SETOK_CHECK_ARG(n > 0, "setok_load_weight(%s): empty tensor", name);
    const bool vec = ndim == 1 && key != "tower.embeddings.class_embedding";
    t.dtype = vec ? SETOK_F32 : c->cfg.dtype;
    t.p = ctx_alloc(c, (size_t)n * elem_size(t.dtype));
    if (!t.p) return ctx_fail(c, SETOK_ELAUNCH, "setok_load_weight(%s): out of device memory", name);
    hipStream_t s = (hipStream_t)stream;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (dtype == t.dtype) {
        if (hipMemcpyAsync(t.p, ptr, (size_t)n * elem_size(dtype), hipMemcpyDeviceToDevice, s) != hipSuccess)
            return ctx_fail(c, SETOK_ELAUNCH, "setok_load_weight(%s): copy failed", name);
    } else if (dtype == SETOK_BF16) cast_kernel<bf16, float><<<grid, 256, 0, s>>>((const bf16*)ptr, (float*)t.p, n);
    else cast_kernel<float, bf16><<<grid, 256, 0, s>>>((const float*)ptr, (bf16*)t.p, n);
    SETOK_CHECK_LAUNCH("setok_load_weight");
    c->w[key] = t;
    c->ready = false;
    return SETOK_OK;
}

namespace {

#define NEED(var, name, ...)                                                                       \
    const DevTensor* var = find(c, name);                                                          \
    if (!var) return ctx_fail(c, SETOK_EINVAL, "setok_weights_ready: weight `%s` was not loaded", std::string(name).c_str());

int get(setok_ctx* c, const std::string& name, int64_t numel, const DevTensor** out) {
    const DevTensor* t = find(c, name);
    if (!t) return ctx_fail(c, SETOK_EINVAL, "setok_weights_ready: weight `%s` was not loaded", name.c_str());
    if (t->numel() != numel) return ctx_fail(c, SETOK_EINVAL, "setok_weights_ready: `%s` has %lld elements, expected %lld", name.c_str(), (long long)t->numel(), (long long)numel);
    *out = t;
    return SETOK_OK;
}

int pack_block(setok_ctx* c, const char* prefix, int depth, BlockW& b) {
    const int C = c->cfg.hidden_size, F = c->cfg.dim_feedforward;
    const DevTensor* t;
    const std::string p(prefix);
    int rc;
#define G(field, nm, n) if ((rc = get(c, p + nm, n, &t)) != SETOK_OK) return rc; field = (decltype(field))t->p;
    G(b.n1w, "norm1.weight", C) G(b.n1b, "norm1.bias", C) G(b.n2w, "norm2.weight", C) G(b.n2b, "norm2.bias", C)
    b.attn.resize(depth);
    for (int i = 0; i < depth; ++i) {
        const std::string a = "layers." + std::to_string(i) + ".1.";
        G(b.attn[i].wqkv, a + "qkv.weight", 3LL * C * C) G(b.attn[i].bqkv, a + "qkv.bias", 3LL * C)
        G(b.attn[i].wproj, a + "proj.weight", (int64_t)C * C) G(b.attn[i].bproj, a + "proj.bias", C)
    }
    G(b.w1, "mlp.fc1.weight", (int64_t)F * C) G(b.b1, "mlp.fc1.bias", F) G(b.w2, "mlp.fc2.weight", (int64_t)C * F) G(b.b2, "mlp.fc2.bias", C)
#undef G
    return SETOK_OK;
}

}  // namespace

extern "C" int setok_ln_fold(void*, const void*, const float*, const float*, const float*, void*, float*, float*, float*, int, int);

extern "C" int setok_weights_ready(setok_ctx* c, void* stream) {
    SETOK_CHECK_ARG(c, "setok_weights_ready: null context");
    const setok_config& f = c->cfg;
    const int C = f.hidden_size, I = f.intermediate_size, p = f.patch_size, es = elem_size(f.dtype);
    hipStream_t s = (hipStream_t)stream;
    const DevTensor* t;
    int rc;
    // patch conv (C, 3, p, p) -> (C, kpad), zero padded
    if ((rc = get(c, "tower.embeddings.patch_embedding.weight", (int64_t)C * 3 * p * p, &t)) != SETOK_OK) return rc;
    c->wp = ctx_alloc(c, (size_t)C * c->kpad * es);
    if (!c->wp) return ctx_fail(c, SETOK_ELAUNCH, "setok_weights_ready: out of device memory");
    if (hipMemsetAsync(c->wp, 0, (size_t)C * c->kpad * es, s) != hipSuccess ||
        hipMemcpy2DAsync(c->wp, (size_t)c->kpad * es, t->p, (size_t)3 * p * p * es, (size_t)3 * p * p * es, C, hipMemcpyDeviceToDevice, s) != hipSuccess)
        return ctx_fail(c, SETOK_ELAUNCH, "setok_weights_ready: packing the patch matrix failed");
    if ((rc = get(c, "tower.embeddings.class_embedding", C, &t)) != SETOK_OK) return rc;
    if ((rc = get(c, "tower.embeddings.position_embedding.weight", (int64_t)c->T * C, &t)) != SETOK_OK) return rc;
    if ((rc = get(c, "tower.pre_layrnorm.weight", C, &t)) != SETOK_OK) return rc;
    if ((rc = get(c, "tower.pre_layrnorm.bias", C, &t)) != SETOK_OK) return rc;
    if ((rc = get(c, "position_embedding.table", (int64_t)c->N * C, &t)) != SETOK_OK) return rc;
    c->tower.assign(c->layers_run, TowerLayer());
    for (int i = 0; i < c->layers_run; ++i) {
        TowerLayer& L = c->tower[i];
        const std::string q = "tower.encoder.layers." + std::to_string(i) + ".";
        const DevTensor *wq, *wk, *wv, *bq, *bk, *bv;
#define G(var, nm, n) if ((rc = get(c, q + nm, n, &var)) != SETOK_OK) return rc;
        G(wq, "self_attn.q_proj.weight", (int64_t)C * C) G(wk, "self_attn.k_proj.weight", (int64_t)C * C) G(wv, "self_attn.v_proj.weight", (int64_t)C * C)
        G(bq, "self_attn.q_proj.bias", C) G(bk, "self_attn.k_proj.bias", C) G(bv, "self_attn.v_proj.bias", C)
        L.wqkv = ctx_alloc(c, (size_t)3 * C * C * es);
        L.bqkv = (float*)ctx_alloc(c, (size_t)3 * C * 4);
        if (!L.wqkv || !L.bqkv) return ctx_fail(c, SETOK_ELAUNCH, "setok_weights_ready: out of device memory");
        const DevTensor* ws[3] = {wq, wk, wv}; const DevTensor* bs[3] = {bq, bk, bv};
        for (int j = 0; j < 3; ++j) {
            if (hipMemcpyAsync((char*)L.wqkv + (size_t)j * C * C * es, ws[j]->p, (size_t)C * C * es, hipMemcpyDeviceToDevice, s) != hipSuccess ||
                hipMemcpyAsync(L.bqkv + (size_t)j * C, bs[j]->p, (size_t)C * 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
                return ctx_fail(c, SETOK_ELAUNCH, "setok_weights_ready: fusing q|k|v failed");
        }
        G(t, "self_attn.out_proj.weight", (int64_t)C * C) L.wo = t->p;   G(t, "self_attn.out_proj.bias", C) L.bo = (const float*)t->p;
        G(t, "mlp.fc1.weight", (int64_t)I * C) L.w1 = t->p;              G(t, "mlp.fc1.bias", I) L.b1 = (const float*)t->p;
        G(t, "mlp.fc2.weight", (int64_t)C * I) L.w2 = t->p;              G(t, "mlp.fc2.bias", C) L.b2 = (const float*)t->p;
        G(t, "layer_norm1.weight", C) L.ln1w = (const float*)t->p;       G(t, "layer_norm1.bias", C) L.ln1b = (const float*)t->p;
        G(t, "layer_norm2.weight", C) L.ln2w = (const float*)t->p;       G(t, "layer_norm2.bias", C) L.ln2b = (const float*)t->p;
#undef G
        if (c->fold) {
            Folded* fo[2] = {&L.qkv_ln, &L.fc1_ln};
            const int Ns[2] = {3 * C, I};
            for (int j = 0; j < 2; ++j) {
                fo[j]->wg = ctx_alloc(c, (size_t)Ns[j] * C * 2);
                fo[j]->colsum = (float*)ctx_alloc(c, (size_t)Ns[j] * 4);
                fo[j]->bias = (float*)ctx_alloc(c, (size_t)Ns[j] * 4);
                fo[j]->frag = (float*)ctx_alloc(c, (size_t)Ns[j] * 16);
                if (!fo[j]->wg || !fo[j]->colsum || !fo[j]->bias || !fo[j]->frag) return ctx_fail(c, SETOK_ELAUNCH, "setok_weights_ready: out of device memory");
            }
            if ((rc = setok_ln_fold(stream, L.wqkv, L.ln1w, L.ln1b, L.bqkv, L.qkv_ln.wg, L.qkv_ln.colsum, L.qkv_ln.bias, L.qkv_ln.frag, 3 * C, C)) != SETOK_OK) return rc;
            if ((rc = setok_ln_fold(stream, L.w1, L.ln2w, L.ln2b, L.b1, L.fc1_ln.wg, L.fc1_ln.colsum, L.fc1_ln.bias, L.fc1_ln.frag, I, C)) != SETOK_OK) return rc;
        }
    }
    if ((rc = pack_block(c, "inner_encoder.", f.inner_cluster_layers, c->inner)) != SETOK_OK) return rc;
    if ((rc = pack_block(c, "inter_encoder.", f.intra_cluster_layers, c->inter)) != SETOK_OK) return rc;
    if ((rc = get(c, "out.weight", (int64_t)f.token_feat_dim * C, &t)) != SETOK_OK) return rc;
    if ((rc = get(c, "out.bias", f.token_feat_dim, &t)) != SETOK_OK) return rc;
    c->ready = true;
    return SETOK_OK;
}

// (C++ linkage: the forms with a device-side row count, gemm.hip / norm_attn.hip)
int setok_linear_dev(void*, int, int, const void*, int64_t, const void*, const float*, const void*, void*, int64_t, int, int, int, int, int, int64_t, int64_t, int64_t, const int32_t*);
int setok_layernorm_dev(void*, int, const void*, const float*, const float*, void*, int, int, float, const int32_t*);

// ---- workspace ------------------------------------------------------------------------------------------------------------------------
namespace {

struct Bump {
    char* base; int64_t off = 0, cap;
    void* take(int64_t bytes) { const int64_t at = (off + 255) & ~255LL; off = at + bytes; return base ? base + at : nullptr; }
};

struct Plan {
    void *h, *x, *idx_tables;                 // persistent over the call
    int32_t *perm, *seg_offsets, *img_offsets;
    // tower scratch
    void *patches, *pe, *y, *qkv, *a, *u; float* stats;
    // head scratch (overlaps the tower's)
    void *hs, *y2, *qkv2, *o2, *u2, *group; float *dist_ws, *vec_ws;
    int64_t bytes;
};

extern "C" int setok_cluster_workspace(int, int, int, int, int64_t*, int64_t*);

Plan plan(const setok_ctx* c, int B, char* base) {
    const setok_config& f = c->cfg;
    const int64_t es = elem_size(f.dtype), C = f.hidden_size, I = f.intermediate_size, F = f.dim_feedforward;
    const int64_t rT = (int64_t)B * c->T, rN = (int64_t)B * c->N;
    Plan p;
    Bump b{base, 0, 0};
    p.h = b.take(rT * C * es);
    p.x = b.take(rN * C * es);
    p.perm = (int32_t*)b.take(rN * 4);
    p.seg_offsets = (int32_t*)b.take((rN + 1) * 4);
    p.img_offsets = (int32_t*)b.take((B + 1) * 4);
    const int64_t mark = b.off;
    p.patches = b.take(rN * c->kpad * es);
    p.pe = b.take(rN * C * es);
    p.y = b.take(rT * C * es);
    p.qkv = b.take(rT * 3 * C * es);
    p.a = b.take(rT * C * es);
    p.u = b.take(rT * I * es);
    p.stats = (float*)b.take(rT * 32);
    const int64_t tower_end = b.off;
    b.off = mark;
    int64_t nd = 0, nv = 0;
    setok_cluster_workspace(f.dtype, B, c->N, (int)C, &nd, &nv);
    p.dist_ws = nd ? (float*)b.take(nd * 4) : nullptr;
    p.vec_ws = nv ? (float*)b.take(nv * 4) : nullptr;
    p.hs = b.take(rN * C * es);
    p.y2 = b.take(rN * C * es);
    p.qkv2 = b.take(rN * 3 * C * es);
    p.o2 = b.take(rN * C * es);
    p.u2 = b.take(rN * F * es);
    p.group = b.take(rN * C * es);
    p.bytes = (tower_end > b.off ? tower_end : b.off) + 256;
    return p;
}

extern "C" {
int setok_linear(void*, int, int, const void*, int64_t, const void*, const float*, const void*, void*, int64_t, int, int, int, int, int, int64_t, int64_t, int64_t);

int setok_linear_ln(void*, const void*, int64_t, const void*, const float*, const float*, void*, int64_t, int, int, int, int);
int setok_row_stats(void*, int, const void*, float*, int, int, float);
int setok_layernorm(void*, int, const void*, const float*, const float*, void*, int, int, float);
int setok_attention(void*, int, const void*, const int32_t*, int, int, void*, int, int, int, float);
int setok_patchify(void*, int, const void*, void*, int, int, int, int, int);
int setok_vit_assemble(void*, int, const void*, const void*, const void*, void*, int, int, int);
int setok_select_add_pos(void*, int, const void*, const void*, void*, int, int, int, int);
int setok_cluster_dpc_knn(void*, int, const void*, int, int, int, int, float, int, const float*, const float*, int64_t*, float*, int64_t*, int32_t*, float*, float*);
int setok_cluster_sort(void*, const int64_t*, const int32_t*, int, int, int32_t*, int32_t*, int32_t*);
int setok_gather_rows(void*, int, const void*, const int32_t*, void*, int, int);
int setok_segment_mean(void*, int, const void*, const int32_t*, const int32_t*, int, void*, int);
}

#define RUN(call) do { const int rc__ = (call); if (rc__ != SETOK_OK) return rc__; } while (0)

int lin(void* st, int dt, const void* a, int K, const void* w, const float* b, const void* res, void* out, int M, int N, int act, const int32_t* m_dev = nullptr) {
    return setok_linear_dev(st, dt, dt, a, K, w, b, res, out, N, M, N, K, act, 1, 0, 0, 0, m_dev);
}

// Block.forward (module.py:95-100) on packed rows, each row attending within its own segment; h is updated in place.  `rows` sizes the launches;
// `rows_dev` (optional, device) is the number of rows that exist: the ragged stages run without the host knowing it.
int block_rows(const setok_ctx* c, void* st, const BlockW& b, void* h, int rows, const int32_t* seg_offsets, int n_segs, int seg_bound,
               void* y, void* qkv, void* o, void* u, const int32_t* rows_dev = nullptr) {
    const setok_config& f = c->cfg;
    const int C = f.hidden_size, H = f.nheads, Dh = C / H, dt = f.dtype;
    const float eps = 1e-5f, scale = 1.0f / sqrtf((float)Dh);                       // nn.LayerNorm default eps; qk_scale None -> head_dim ** -0.5 (module.py:54)
    for (const auto& a : b.attn) {
        RUN(setok_layernorm_dev(st, dt, h, b.n1w, b.n1b, y, rows, C, eps, rows_dev));
        RUN(lin(st, dt, y, C, a.wqkv, a.bqkv, nullptr, qkv, rows, 3 * C, SETOK_ACT_NONE, rows_dev));
        RUN(setok_attention(st, dt, qkv, seg_offsets, n_segs, seg_bound, o, rows, H, Dh, scale));
        RUN(lin(st, dt, o, C, a.wproj, a.bproj, h, h, rows, C, SETOK_ACT_NONE, rows_dev));
    }
    RUN(setok_layernorm_dev(st, dt, h, b.n2w, b.n2b, y, rows, C, eps, rows_dev));
    RUN(lin(st, dt, y, C, b.w1, b.b1, nullptr, u, rows, f.dim_feedforward, SETOK_ACT_GELU_ERF, rows_dev));
    RUN(lin(st, dt, u, f.dim_feedforward, b.w2, b.b2, h, h, rows, C, SETOK_ACT_NONE, rows_dev));
    return SETOK_OK;
}

}  // namespace

extern "C" int64_t setok_encode_workspace_bytes(const setok_ctx* c, int B) {
    if (!c || B <= 0) return -1;
    return plan(c, B, nullptr).bytes;
}

extern "C" int setok_encode(setok_ctx* c, void* stream, const void* images, int B, int k, float threshold, const float* noise,
                            const float* token_mask, void* workspace, int64_t workspace_bytes, void* tokens, int32_t* counts,
                            int64_t* idx_cluster, float* score, int64_t* index_down, int32_t* counts_host, int64_t* total_tokens_host,
                            void** stage_x, void** stage_group, void** stage_inter) {
    SETOK_CHECK_ARG(c && images && workspace && tokens && counts && idx_cluster && score && index_down, "setok_encode: null argument");
    SETOK_CHECK_ARG(B > 0, "setok_encode: B=%d", B);
    if (!c->ready) return ctx_fail(c, SETOK_EINVAL, "setok_encode: call setok_weights_ready after loading the weights");
    const setok_config& f = c->cfg;
    const int dt = f.dtype, C = f.hidden_size, I = f.intermediate_size, H = f.num_attention_heads, N = c->N, T = c->T;
    const Plan p = plan(c, B, (char*)workspace);
    SETOK_CHECK_ARG(workspace_bytes >= p.bytes, "setok_encode: workspace of %lld bytes, %lld needed (setok_encode_workspace_bytes)", (long long)workspace_bytes,
                    (long long)p.bytes);
    SETOK_CHECK_ARG(((size_t)workspace & 255) == 0, "setok_encode: the workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    void* st = stream;
    const float eps = f.layer_norm_eps;
    auto W = [&](const char* n) { return find(c, n)->p; };

    // ---- tower: CLIPVisionTower.forward -> hidden_states[select_layer] (clip_encoder.py:50-62) ------------------------------------------
    RUN(setok_patchify(st, dt, images, p.patches, B, f.image_size, f.image_size, f.patch_size, c->kpad));
    RUN(lin(st, dt, p.patches, c->kpad, c->wp, nullptr, nullptr, p.pe, B * N, C, SETOK_ACT_NONE));
    RUN(setok_vit_assemble(st, dt, p.pe, W("tower.embeddings.class_embedding"), W("tower.embeddings.position_embedding.weight"), p.h, B, N, C));
    RUN(setok_layernorm(st, dt, p.h, (const float*)W("tower.pre_layrnorm.weight"), (const float*)W("tower.pre_layrnorm.bias"), p.h, B * T, C, eps));
    const float scale = 1.0f / sqrtf((float)c->Dh);
    for (const TowerLayer& L : c->tower) {
        if (c->fold) {
            RUN(setok_row_stats(st, dt, p.h, p.stats, B * T, C, eps));
            RUN(setok_linear_ln(st, p.h, C, L.qkv_ln.wg, L.qkv_ln.frag, p.stats, p.qkv, 3 * C, B * T, 3 * C, C, SETOK_ACT_NONE));
        } else {
            RUN(setok_layernorm(st, dt, p.h, L.ln1w, L.ln1b, p.y, B * T, C, eps));
            RUN(lin(st, dt, p.y, C, L.wqkv, L.bqkv, nullptr, p.qkv, B * T, 3 * C, SETOK_ACT_NONE));
        }
        RUN(setok_attention(st, dt, p.qkv, nullptr, 0, T, p.a, B * T, H, c->Dh, scale));
        RUN(lin(st, dt, p.a, C, L.wo, L.bo, p.h, p.h, B * T, C, SETOK_ACT_NONE));
        if (c->fold) {
            RUN(setok_row_stats(st, dt, p.h, p.stats, B * T, C, eps));
            RUN(setok_linear_ln(st, p.h, C, L.fc1_ln.wg, L.fc1_ln.frag, p.stats, p.u, I, B * T, I, C, SETOK_ACT_QUICK_GELU));
        } else {
            RUN(setok_layernorm(st, dt, p.h, L.ln2w, L.ln2b, p.y, B * T, C, eps));
            RUN(lin(st, dt, p.y, C, L.w1, L.b1, nullptr, p.u, B * T, I, SETOK_ACT_QUICK_GELU));
        }
        RUN(lin(st, dt, p.u, I, L.w2, L.b2, p.h, p.h, B * T, C, SETOK_ACT_NONE));
    }

    // ---- head: tokenizer.py:162-180 -------------------------------------------------------------------------------------------------------
    const int skip = f.select_cls_patch ? 0 : 1;
    const int Nh = N + 1 - skip;                                                      // 'cls_patch' keeps the class token: N + 1 tokens (not a square grid: the reference fails too)
    SETOK_CHECK_ARG(Nh == N, "setok_encode: select_feature 'cls_patch' gives %d tokens, not a square grid (einops rearrange fails in the reference, tokenizer.py:165)", Nh);
    RUN(setok_select_add_pos(st, dt, p.h, W("position_embedding.table"), p.x, B, N, C, skip));
    const float thr = threshold != 0.0f ? threshold : f.threshold;                    // `threshold if threshold else self.threshold` (:171)
    const int kk = k != 0 ? k : f.min_cluster_num;                                    // `k if k else self.min_cluster_num` (:172)
    RUN(setok_cluster_dpc_knn(st, dt, p.x, B, N, C, kk, thr, f.min_cluster_num, noise, token_mask, idx_cluster, score, index_down, counts, p.dist_ws, p.vec_ws));
    RUN(setok_cluster_sort(st, idx_cluster, counts, B, N, p.perm, p.seg_offsets, p.img_offsets));
    if (counts_host) {
        // The token counts are final HERE.  A host that wants the shapes (counts_host != NULL) gets them copied now and waits for THIS copy at the end
        // of the call, not for the stream: the head's launches (~1.2 ms of device time at batch 256) are then still queued or running, and whatever
        // the host enqueues next — the projector — lands behind them with the device never idle.  Round 5: waiting for the whole stream left the
        // device idle for 220-300 us per call (wake-up of the blocked host thread + the way back up into the caller + its next launch).
        if (c->counts_cap < B) {
            if (c->counts_pinned) (void)hipHostFree(c->counts_pinned);
            c->counts_pinned = nullptr; c->counts_cap = 0;
            if (hipHostMalloc((void**)&c->counts_pinned, (size_t)B * 4, hipHostMallocDefault) != hipSuccess) return ctx_fail(c, SETOK_ELAUNCH, "setok_encode: pinned allocation failed");
            c->counts_cap = B;
        }
        if (!c->counts_ready && hipEventCreateWithFlags(&c->counts_ready, hipEventDisableTiming) != hipSuccess) {
            c->counts_ready = nullptr;
            return ctx_fail(c, SETOK_ELAUNCH, "setok_encode: event creation failed");
        }
        if (hipMemcpyAsync(c->counts_pinned, counts, (size_t)B * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipEventRecord(c->counts_ready, s) != hipSuccess)
            return ctx_fail(c, SETOK_ELAUNCH, "setok_encode: reading the token counts failed: %s", hipGetErrorString(hipGetLastError()));
    }
    // No host synchronisation between the stages: the ragged stages are launched at their worst-case size — B * N cluster
    // segments, B * N cluster-token rows — and take the actual numbers from the device: the segment offsets past the last cluster are empty
    // segments (setok_cluster_sort), the GEMMs / LayerNorms of the inter encoder and `out` read the row count sum(L_i) = img_offsets[B] and skip
    // everything beyond it.  Rows [sum L_i, B * N) of `tokens` are never written.
    const int rN = B * N;
    const int32_t* total_dev = p.img_offsets + B;
    RUN(setok_gather_rows(st, dt, p.x, p.perm, p.hs, rN, C));                                                                // x[m] for every cluster (:150)
    RUN(block_rows(c, st, c->inner, p.hs, rN, p.seg_offsets, rN, N, p.y2, p.qkv2, p.o2, p.u2));                              // inner_encoder (:150)
    RUN(setok_segment_mean(st, dt, p.hs, p.seg_offsets, total_dev, rN, p.group, C));                                         // mean over members (:151-153)
    if (stage_x) *stage_x = p.x;
    void* inter = p.group;
    if (stage_group) {                                                                                                       // keep the group stage: run inter_encoder on a copy
        if (hipMemcpyAsync(p.o2, p.group, (size_t)rN * C * elem_size(dt), hipMemcpyDeviceToDevice, s) != hipSuccess) return ctx_fail(c, SETOK_ELAUNCH, "setok_encode: copy failed");
        *stage_group = p.group;
        inter = p.hs;                                                                                                        // hs is dead now
        if (hipMemcpyAsync(inter, p.o2, (size_t)rN * C * elem_size(dt), hipMemcpyDeviceToDevice, s) != hipSuccess) return ctx_fail(c, SETOK_ELAUNCH, "setok_encode: copy failed");
    }
    RUN(block_rows(c, st, c->inter, inter, rN, p.img_offsets, B, N, p.y2, p.qkv2, p.o2, p.u2, total_dev));                    // inter_encoder (:179, +D2)
    if (stage_inter) *stage_inter = inter;
    RUN(lin(st, dt, inter, C, W("out.weight"), (const float*)W("out.bias"), nullptr, tokens, rN, f.token_feat_dim, SETOK_ACT_NONE, total_dev));   // :180
    if (!counts_host) return SETOK_OK;                                                // fully asynchronous: the caller reads `counts` when it wants shapes
    // convenience for hosts that want the shapes right away: ONE wait, at the END of the call (every launch is already queued), for the copy of the
    // counts issued behind the clustering — NOT for the launches behind it: `tokens` and the other outputs are complete in stream order
    static const bool wait_stream = [] { const char* e = getenv("SETOK_ENCODE_WAIT"); return e && e[0] == 's'; }();   // SETOK_ENCODE_WAIT=stream: rounds 3-4 (A/B runs)
    if ((wait_stream ? hipStreamSynchronize(s) : hipEventSynchronize(c->counts_ready)) != hipSuccess)
        return ctx_fail(c, SETOK_ELAUNCH, "setok_encode: reading the token counts failed: %s", hipGetErrorString(hipGetLastError()));
    int64_t total = 0;
    for (int b = 0; b < B; ++b) { counts_host[b] = c->counts_pinned[b]; total += counts_host[b]; }
    if (total_tokens_host) *total_tokens_host = total;
    return SETOK_OK;
}
