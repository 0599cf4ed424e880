// Model handles: weight repacking and the forward passes of OrigUNet (+ConvLSTM), the Mix-Transformer
// velocity models (LSTMNetVIT / ViT) and their composite, as HIP launch sequences on the caller's stream.
//
// Replaces learner/learner_models.py:339-636, learner/ConvLSTM_pytorch/convlstm.py:38-176,
// learner/vitfly_models.py:18-31,111-186, learner/ViTsubmodules.py:15-148 (per-entry citations in
// include/evfly_hip.h). Activations are fp32 NHWC in one device arena owned by the handle; streams are
// processed in chunks so the arena stays bounded.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "bf16.h"
#include "common.h"
#include "igemm.h"
#include "ops.h"

using namespace evfly;

namespace {

struct HostTensor {
    std::vector<float> v;
    std::vector<int64_t> shape;
    int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};

struct Tap {
    float *ptr;
    int64_t shape[4];
    bool bf16;      // the buffer holds bf16 elements (bf16 pipeline)
};

struct ProfRec {
    std::string name;
    double flops, bytes, exec, useful;
    int launches;               // kernel launches behind the site (2: a Winograd split plan)
    hipEvent_t e0, e1;
};
struct ProfAgg {
    std::string name;
    // exec: matrix-core flops actually issued (Winograd: 16/36 of flops plus tile padding); useful: the part of exec that
    // lands on real output tiles (Winograd: 16/36 of the direct-conv count, no padding; equal to flops for direct GEMMs)
    double ms = 0, flops = 0, bytes = 0, exec = 0, useful = 0;
    int launches = 0;
};

}  // namespace

struct evfly_model {
    evfly_model_config cfg{};
    std::map<std::string, HostTensor> host;
    bool finalized = false;
    int device = 0;
    // packed weights: one device allocation, name -> offset
    float *wdev = nullptr;
    std::map<std::string, size_t> woff;
    std::map<std::string, int> wld;       // padded k stride of GEMM weights
    std::vector<float> wstage;
    // activation arena
    char *arena = nullptr;
    size_t arena_cap = 0, arena_off = 0;
    int arena_generation = 0;      // bumped whenever the arena is reallocated (pointers captured earlier are stale)
    bool planning = false;
    size_t plan_peak = 0;
    std::map<std::string, Tap> taps;
    // profiling
    bool profiling = false;
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    std::vector<ProfAgg> agg;
    hipStream_t st = nullptr;
    int vp_hidden = 0;   // lstm_velpred hidden size = flattened conv features (set at finalize)
    // full-resolution encoder maps: kept complete when EVFLY_FULL_ENCODER_OUTPUTS is set at evfly_model_create time (debug
    // taps "e1".."e4"); otherwise the fused-skip Winograd launches store only the block-border pixels (`skip_bands`)
    // bf16 pipeline (compute_dtype BF16): activations are bf16 NHWC in the arena, GEMM weights bf16 (rounded at pack time).
    // EVFLY_BF16_LEGACY at create time keeps the round-1 path (fp32 activations rounded while they are staged) for A/B runs.
    bool act16 = false;
    size_t esz() const { return act16 ? 2 : 4; }           // bytes per activation element
    bool full_encoder_outputs = false;
    bool bands_used = false;   // the last forward left "e1".."e4" partial
    bool dot_used = false;     // the last forward fused unet_out into d42: "d4" was not written
    // ConvLSTM state rows (streams x 104) of the forward's FIRST chunk: the bf16 pipeline picks its recurrence kernel (one launch per chunk
    // or per-step launches) from it ONCE per forward, so that a stream's depth and state do not depend on which chunk of the batch it
    // falls in (a smaller tail chunk used to take the other path: a few bf16 ulps apart)
    int64_t clstm_rows_first = 0;
    struct { const float *w = nullptr, *b = nullptr; float *y = nullptr; bool done = false; } dot;     // request for the next conv()

    ~evfly_model() {
        if (wdev) (void)hipFree(wdev);
        if (arena) (void)hipFree(arena);
        for (auto e : ev_pool) (void)hipEventDestroy(e);
    }

    // ------------------------------------------------------------------ host tensors
    const HostTensor *find(const std::string &key, const char *prefix) const {
        auto it = host.find(std::string(prefix) + key);
        if (it != host.end()) return &it->second;
        it = host.find(key);
        return it == host.end() ? nullptr : &it->second;
    }

    // ------------------------------------------------------------------ packed weights
    float *stage(const std::string &name, size_t n) {
        size_t off = (wstage.size() + 63) / 64 * 64;   // 256-B aligned
        wstage.resize(off + n, 0.f);
        woff[name] = off;
        return wstage.data() + off;
    }
    // bf16 weights share the staging vector (two elements per float slot, 256-B aligned like the others)
    bf16_t *stage16(const std::string &name, size_t n) { return reinterpret_cast<bf16_t *>(stage(name, (n + 1) / 2)); }
    const float *W(const std::string &name) const {
        auto it = woff.find(name);
        return it == woff.end() ? nullptr : wdev + it->second;
    }
    bool has(const std::string &name) const { return woff.count(name) != 0; }

    // ------------------------------------------------------------------ arena
    float *alloc(int64_t n_floats) {
        size_t bytes = ((size_t)n_floats * 4 + 255) / 256 * 256;
        size_t off = arena_off;
        arena_off += bytes;
        if (planning) { plan_peak = std::max(plan_peak, arena_off); return reinterpret_cast<float *>(16); }
        return reinterpret_cast<float *>(arena + off);
    }
    // activation buffers: n ELEMENTS of the pipeline's activation type (fp32, or bf16 in the bf16 pipeline)
    float *alloc_act(int64_t n_elems) { return act16 ? alloc((n_elems + 1) / 2) : alloc(n_elems); }
    // p + n elements of the activation type
    float *eoff(float *p, int64_t n_elems) const { return reinterpret_cast<float *>(reinterpret_cast<char *>(p) + n_elems * (int64_t)esz()); }
    const float *eoff(const float *p, int64_t n_elems) const { return reinterpret_cast<const float *>(reinterpret_cast<const char *>(p) + n_elems * (int64_t)esz()); }
    void tap(const char *name, float *p, int64_t a, int64_t b, int64_t c, int64_t d, bool is16 = false) {
        if (!planning) taps[name] = Tap{p, {a, b, c, d}, is16};
    }

    // ------------------------------------------------------------------ profiling
    std::string prof_filter;   // non-empty: only launch sites whose name starts with it are bracketed by events
    bool prof_skipped = false;
    // fused first-conv producer for the next conv() call (consumed and cleared there; Winograd path only)
    struct { const float *frames = nullptr, *w = nullptr, *b = nullptr; int cin = 0, form_bev = 0, apply_form = 0; float cutoff = 0.f; } pre;
    double next_exec = 0, next_useful = 0;   // set by conv() before RUN when the kernel issues fewer flops than the algorithmic count
    int next_launches = 1;
    int prof_begin(const char *name, double flops, double bytes) {
        const double ex = next_exec > 0 ? next_exec : flops;
        const double us = next_useful > 0 ? next_useful : flops;
        const int nl = next_launches;
        next_exec = 0; next_useful = 0; next_launches = 1;
        prof_skipped = false;
        if (!profiling || planning) return 0;
        if (!prof_filter.empty() && std::strncmp(name, prof_filter.c_str(), prof_filter.size()) != 0) { prof_skipped = true; return 0; }
        while (ev_pool.size() < ev_used + 2) {
            hipEvent_t e;
            EVFLY_HIP(hipEventCreate(&e));
            ev_pool.push_back(e);
        }
        ProfRec r{name, flops, bytes, ex, us, nl, ev_pool[ev_used], ev_pool[ev_used + 1]};
        ev_used += 2;
        EVFLY_HIP(hipEventRecord(r.e0, st));
        prof.push_back(r);
        return 0;
    }
    int prof_end() {
        if (!profiling || planning || prof_skipped) return 0;
        EVFLY_HIP(hipEventRecord(prof.back().e1, st));
        return 0;
    }
    int prof_collect() {
        if (prof.empty()) return 0;
        EVFLY_HIP(hipEventSynchronize(prof.back().e1));
        for (auto &r : prof) {
            float ms = 0;
            EVFLY_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
            ProfAgg *a = nullptr;
            for (auto &x : agg) if (x.name == r.name) a = &x;
            if (!a) { agg.push_back(ProfAgg{r.name}); a = &agg.back(); }
            a->ms += ms; a->flops += r.flops; a->bytes += r.bytes; a->exec += r.exec; a->useful += r.useful; a->launches += r.launches;
        }
        prof.clear();
        ev_used = 0;
        return 0;
    }
};

namespace {

#define RUN(m, name, flops, bytes, call)                         \
    do {                                                         \
        if (!(m)->planning) {                                    \
            if (int _rc = (m)->prof_begin(name, flops, bytes)) return _rc; \
            if (int _rc = (call)) return _rc;                    \
            if (int _rc = (m)->prof_end()) return _rc;           \
        }                                                        \
    } while (0)

// ============================================================================ packing (host)
// Conv2d weight (O, I, kh, kw) -> [O][kh][kw][I], k padded with zeros to a multiple of 32
// pad_cin: lay the weights out for an input whose channel count is padded to a multiple of 32 with zero channels (the
// vectorised / DMA K walk of the GEMM kernel needs C % 32 == 0; the generic gather is an order of magnitude slower)
int pack_conv(evfly_model *m, const char *prefix, const std::string &key, const std::string &name, bool need = true, bool pad_cin = false,
              bool w16 = false) {
    const HostTensor *t = m->find(key + ".weight", prefix);
    if (!t) { if (need) return fail(-4, "missing tensor %s%s.weight", prefix, key.c_str()); return 1; }
    EVFLY_REQUIRE(t->shape.size() == 4, "%s.weight: expected 4 dims", key.c_str());
    const int O = (int)t->shape[0], I = (int)t->shape[1], kh = (int)t->shape[2], kw = (int)t->shape[3];
    const int Ip = pad_cin ? round_up(I, 32) : I;
    w16 = w16 && Ip % 32 == 0;                                  // bf16 weights go with the bf16 GEMM kernel (bf16 input, C % 32 == 0)
    const int K = kh * kw * Ip, ld = round_up(K, w16 ? 64 : 32);
    if (w16) {
        bf16_t *dst = m->stage16(name + ".w", (size_t)O * ld);     // (staged zero-filled)
        for (int o = 0; o < O; ++o)
            for (int i = 0; i < I; ++i)
                for (int y = 0; y < kh; ++y)
                    for (int x = 0; x < kw; ++x)
                        dst[(size_t)o * ld + conv_k_index(y * kw + x, i, Ip, kh * kw)] = host_f2bf(t->v[(((size_t)o * I + i) * kh + y) * kw + x]);
    } else {
        float *dst = m->stage(name + ".w", (size_t)O * ld);        // (staged zero-filled)
        for (int o = 0; o < O; ++o)
            for (int i = 0; i < I; ++i)
                for (int y = 0; y < kh; ++y)
                    for (int x = 0; x < kw; ++x)
                        dst[(size_t)o * ld + conv_k_index(y * kw + x, i, Ip, kh * kw)] = t->v[(((size_t)o * I + i) * kh + y) * kw + x];
    }
    m->wld[name] = ld;
    const HostTensor *b = m->find(key + ".bias", prefix);
    if (b) std::memcpy(m->stage(name + ".b", b->v.size()), b->v.data(), b->v.size() * 4);
    return 0;
}

// Linear weight (O, I) (optionally spectral-normalised, optionally with permuted input columns)
int pack_linear(evfly_model *m, const char *prefix, const std::string &key, const std::string &name, bool need = true,
                const std::vector<int> *col_perm = nullptr, bool w16 = false) {
    const HostTensor *t = m->find(key + ".weight", prefix);
    std::vector<float> folded;
    if (!t) {   // old-style torch.nn.utils.spectral_norm: W = weight_orig / (u . (W v)), no power iteration in eval
        const HostTensor *wo = m->find(key + ".weight_orig", prefix), *u = m->find(key + ".weight_u", prefix),
                         *v = m->find(key + ".weight_v", prefix);
        if (!wo || !u || !v) { if (need) return fail(-4, "missing tensor %s%s.weight[_orig/_u/_v]", prefix, key.c_str()); return 1; }
        const int O = (int)wo->shape[0], I = (int)wo->shape[1];
        float sigma = 0.f;   // fp32 like torch.dot(u, torch.mv(W, v))
        for (int o = 0; o < O; ++o) {
            float acc = 0.f;
            for (int i = 0; i < I; ++i) acc += wo->v[(size_t)o * I + i] * v->v[i];
            sigma += u->v[o] * acc;
        }
        folded.resize(wo->v.size());
        for (size_t i = 0; i < folded.size(); ++i) folded[i] = wo->v[i] / sigma;
        t = wo;
    }
    const std::vector<float> &src = folded.empty() ? t->v : folded;
    const int O = (int)t->shape[0], I = (int)t->shape[1], ld = round_up(I, w16 ? 64 : 32);
    if (w16) {
        bf16_t *dst = m->stage16(name + ".w", (size_t)O * ld);
        for (int o = 0; o < O; ++o)
            for (int i = 0; i < I; ++i) dst[(size_t)o * ld + (col_perm ? (*col_perm)[i] : i)] = host_f2bf(src[(size_t)o * I + i]);
    } else {
        float *dst = m->stage(name + ".w", (size_t)O * ld);
        for (int o = 0; o < O; ++o)
            for (int i = 0; i < I; ++i) dst[(size_t)o * ld + (col_perm ? (*col_perm)[i] : i)] = src[(size_t)o * I + i];
    }
    m->wld[name] = ld;
    const HostTensor *b = m->find(key + ".bias", prefix);
    if (b) std::memcpy(m->stage(name + ".b", b->v.size()), b->v.data(), b->v.size() * 4);
    return 0;
}

int pack_vec(evfly_model *m, const char *prefix, const std::string &key, const std::string &name) {
    const HostTensor *t = m->find(key, prefix);
    if (!t) return fail(-4, "missing tensor %s%s", prefix, key.c_str());
    std::memcpy(m->stage(name, t->v.size()), t->v.data(), t->v.size() * 4);
    return 0;
}

const char *kUnetP = "origunet.";
const char *kVitP = "vitfly_vitlstm.";

int pack_unet(evfly_model *m) {
    const auto &c = m->cfg;
    {   // e11: (32, cin, 3, 3) -> [tap*cin + ci][32]
        const HostTensor *t = m->find("unet_e11.weight", kUnetP);
        if (!t) return fail(-4, "missing tensor unet_e11.weight");
        const int cin = (int)t->shape[1];
        const int want = (c.form_bev == 1 || c.form_bev == 2) ? 1 : c.num_in_channels;
        EVFLY_REQUIRE(cin == want && t->shape[0] == 32, "unet_e11.weight: expected (32,%d,3,3)", want);
        float *dst = m->stage("e11.w", (size_t)9 * cin * 32);
        for (int o = 0; o < 32; ++o)
            for (int i = 0; i < cin; ++i)
                for (int k = 0; k < 9; ++k) dst[(k * cin + i) * 32 + o] = t->v[((size_t)o * cin + i) * 9 + k];
        if (int rc = pack_vec(m, kUnetP, "unet_e11.bias", "e11.b")) return rc;
    }
    const char *convs[] = {"e12", "e21", "e22", "e31", "e32", "e41", "e42", "e51", "e52",
                           "d11", "d12", "d21", "d22", "d31", "d32", "d41", "d42"};
    const bool w16 = m->act16;
    for (const char *n : convs) {
        if (int rc = pack_conv(m, kUnetP, std::string("unet_") + n, n, true, false, w16)) return rc;
        if (w16) {   // bf16 pipeline, shallow layers: the direct-convolution kernel's weight stream (conv16.hip)
            const HostTensor *tw = m->find(std::string("unet_") + n + ".weight", kUnetP);
            if (tw && tw->shape[1] >= 128 && tw->shape[1] % 64 == 0 && tw->shape[0] % 64 == 0) {      // deep layers: k_conv16p's stream (conv16w.hip)
                const int O = (int)tw->shape[0], I = (int)tw->shape[1];
                m->stage16(std::string(n) + ".wp", conv16p_weight_elems(O, I));      // (staging may reallocate: pointers afterwards)
                conv16p_pack_host(m->wstage.data() + m->woff[std::string(n) + ".w"], O, I, m->wld[n], m->wstage.data() + m->woff[std::string(n) + ".wp"]);
            }
            if (tw && (tw->shape[1] == 32 || tw->shape[1] == 64) && tw->shape[0] % 32 == 0) {
                const int O = (int)tw->shape[0], I = (int)tw->shape[1];
                conv16_pack_host(tw->v.data(), O, I, m->stage16(std::string(n) + ".wd", conv16_weight_elems(O, I, conv16_ntb(O))));
            }
        }
        // Winograd F(2x2,3x3) weights U = G g G^T in the streamed layout of wino.hip (exact-fp32 path only)
        const HostTensor *t = m->find(std::string("unet_") + n + ".weight", kUnetP);
        if (c.compute_dtype == EVFLY_DTYPE_F32 && t && t->shape[1] % 32 == 0) {
            const int O = (int)t->shape[0], I = (int)t->shape[1];
            wino_pack_host(t->v.data(), O, I, m->stage(std::string(n) + ".u", wino_u_floats(O, I)));
        }
    }
    for (int l = 1; l <= 4; ++l) {   // ConvTranspose2d (Cin, Cout, 2, 2) -> [(dy*2+dx)*Cout + co][ci]
        const std::string key = "unet_upconv" + std::to_string(l);
        const HostTensor *t = m->find(key + ".weight", kUnetP);
        if (!t) return fail(-4, "missing tensor %s.weight", key.c_str());
        const int I = (int)t->shape[0], O = (int)t->shape[1], ld = round_up(I, w16 ? 64 : 32);
        if (w16) {
            bf16_t *dst = m->stage16("up" + std::to_string(l) + ".w", (size_t)4 * O * ld);
            for (int i = 0; i < I; ++i)
                for (int o = 0; o < O; ++o)
                    for (int q = 0; q < 4; ++q) dst[((size_t)q * O + o) * ld + i] = host_f2bf(t->v[((size_t)i * O + o) * 4 + q]);
        } else {
            float *dst = m->stage("up" + std::to_string(l) + ".w", (size_t)4 * O * ld);
            for (int i = 0; i < I; ++i)
                for (int o = 0; o < O; ++o)
                    for (int q = 0; q < 4; ++q) dst[((size_t)q * O + o) * ld + i] = t->v[((size_t)i * O + o) * 4 + q];
        }
        m->wld["up" + std::to_string(l)] = ld;
        if (int rc = pack_vec(m, kUnetP, key + ".bias", "up" + std::to_string(l) + ".b")) return rc;
    }
    {   // unet_out (1, 32, 1, 1)
        if (int rc = pack_vec(m, kUnetP, "unet_out.weight", "out.w")) return rc;
        if (int rc = pack_vec(m, kUnetP, "unet_out.bias", "out.b")) return rc;
    }
    if (c.num_recurrent_unet > 0) {   // (4*hid, 2*hid, 1, 1): columns [x | h] (convlstm.py:41)
        const HostTensor *t = m->find("lstm.cell_list.0.conv.weight", kUnetP);
        if (!t) return fail(-4, "missing tensor lstm.cell_list.0.conv.weight");
        const int O = (int)t->shape[0], I = (int)t->shape[1], hid = O / 4;
        EVFLY_REQUIRE(I == 2 * hid && hid == 512, "ConvLSTM weight: expected (2048,1024,1,1)");
        if (w16) {
            m->stage16("clstm.wx", (size_t)O * hid);
            m->stage16("clstm.wh", (size_t)O * hid);   // staging may reallocate: take the pointers afterwards
            bf16_t *wx = reinterpret_cast<bf16_t *>(m->wstage.data() + m->woff["clstm.wx"]);
            bf16_t *wh = reinterpret_cast<bf16_t *>(m->wstage.data() + m->woff["clstm.wh"]);
            for (int o = 0; o < O; ++o)
                for (int i = 0; i < hid; ++i) {
                    wx[(size_t)o * hid + i] = host_f2bf(t->v[(size_t)o * I + i]);
                    wh[(size_t)o * hid + i] = host_f2bf(t->v[(size_t)o * I + hid + i]);
                }
            // the same two matrices with gate-interleaved rows (row 4 cell + gate) for the one-launch recurrence (clstm16.hip)
            std::vector<bf16_t> wxs(wx, wx + (size_t)O * hid), whs(wh, wh + (size_t)O * hid);
            m->stage16("clstm.wx_i", (size_t)O * hid);
            m->stage16("clstm.wh_i", (size_t)O * hid);
            clstm16_interleave_host(wxs.data(), hid, hid, reinterpret_cast<bf16_t *>(m->wstage.data() + m->woff["clstm.wx_i"]));
            m->stage16("clstm.wh_ir", (size_t)O * hid);      // (row-major: the GEMM kernel's operand for the gate-fused per-step path)
            clstm16_interleave_host(whs.data(), hid, hid, reinterpret_cast<bf16_t *>(m->wstage.data() + m->woff["clstm.wh_ir"]));
            {   // the hidden side also in MFMA fragment order: the kernel's weight requests are then 1 KiB of consecutive bytes
                std::vector<bf16_t> whi((size_t)O * hid);
                clstm16_interleave_host(whs.data(), hid, hid, whi.data());
                clstm16_fragment_host(whi.data(), hid, reinterpret_cast<bf16_t *>(m->wstage.data() + m->woff["clstm.wh_i"]));
            }
        } else {
        m->stage("clstm.wx", (size_t)O * hid);
        m->stage("clstm.wh", (size_t)O * hid);   // staging may reallocate: take the pointers afterwards
        float *wx = m->wstage.data() + m->woff["clstm.wx"], *wh = m->wstage.data() + m->woff["clstm.wh"];
        for (int o = 0; o < O; ++o)
            for (int i = 0; i < hid; ++i) {
                wx[(size_t)o * hid + i] = t->v[(size_t)o * I + i];
                wh[(size_t)o * hid + i] = t->v[(size_t)o * I + hid + i];
            }
        }
    }
    return 0;
}

// velpred geometry: input of the DynamicConvNet and the (H, W, C) after every conv+pool pair
struct VpGeom {
    int H0, W0, C0;
    int ch[4], cw[4];      // after the conv
    int ph[4], pw[4];      // after the pool (== conv dims when pool_type is none)
    int C[4];
};

int velpred_geometry(const evfly_model_config &c, VpGeom &g) {
    if (c.velpred == 1) { g.H0 = c.input_h; g.W0 = c.input_w; g.C0 = 1; }
    else if (c.velpred == 11) { g.H0 = 68; g.W0 = 148; g.C0 = 1; }
    else { g.H0 = 8; g.W0 = 13; g.C0 = 512; }
    int H = g.H0, W = g.W0;
    for (int i = 0; i < c.enc_num_layers; ++i) {
        const int k = c.enc_kernel[i], s = c.enc_stride[i];
        EVFLY_REQUIRE(k >= 1 && s >= 1 && H >= k && W >= k, "velpred conv %d: kernel %d does not fit a %dx%d input", i, k, H, W);
        H = (H - k) / s + 1; W = (W - k) / s + 1;
        g.ch[i] = H; g.cw[i] = W; g.C[i] = c.enc_out_channels[i];
        if (c.enc_pool_type != EVFLY_POOL_NONE) {
            const int pk = c.enc_pool_kernel[i], ps = c.enc_pool_stride[i];
            EVFLY_REQUIRE(pk >= 1 && ps >= 1 && H >= pk && W >= pk, "velpred pool %d: kernel %d does not fit %dx%d", i, pk, H, W);
            H = (H - pk) / ps + 1; W = (W - pk) / ps + 1;
        }
        g.ph[i] = H; g.pw[i] = W;
    }
    return 0;
}

// DynamicConvNet / DynamicFCNet weights (learner_models.py:18-145). BatchNorm2d in eval mode is folded into the
// bias-free conv: w' = w * gamma / sqrt(var + eps), b' = beta - mean * gamma / sqrt(var + eps)  (eps = 1e-5).
int pack_velpred(evfly_model *m) {
    const auto &c = m->cfg;
    VpGeom g;
    if (int rc = velpred_geometry(c, g)) return rc;
    int cin = g.C0;
    for (int i = 0; i < c.enc_num_layers; ++i) {
        const std::string si = std::to_string(i), L = "convnet_velpred.layers.";
        const HostTensor *w = m->find(L + "conv2d_" + si + ".weight", kUnetP);
        const HostTensor *ga = m->find(L + "batchnorm_" + si + ".weight", kUnetP), *be = m->find(L + "batchnorm_" + si + ".bias", kUnetP);
        const HostTensor *mu = m->find(L + "batchnorm_" + si + ".running_mean", kUnetP), *var = m->find(L + "batchnorm_" + si + ".running_var", kUnetP);
        if (!w || !ga || !be || !mu || !var) return fail(-4, "missing DynamicConvNet tensors of layer %d (%sconv2d_%d / batchnorm_%d)", i, L.c_str(), i, i);
        const int O = c.enc_out_channels[i], k = c.enc_kernel[i];
        EVFLY_REQUIRE(w->shape.size() == 4 && w->shape[0] == O && w->shape[1] == cin && w->shape[2] == k && w->shape[3] == k,
                      "convnet_velpred conv2d_%d.weight: expected (%d,%d,%d,%d)", i, O, cin, k, k);
        EVFLY_REQUIRE((int)ga->v.size() == O && (int)be->v.size() == O && (int)mu->v.size() == O && (int)var->v.size() == O,
                      "convnet_velpred batchnorm_%d: expected %d channels", i, O);
        HostTensor fw = *w, fb;
        fb.shape = {O}; fb.v.resize(O);
        const size_t per = fw.v.size() / O;
        for (int o = 0; o < O; ++o) {
            const float sc = ga->v[o] / std::sqrt(var->v[o] + 1e-5f);
            for (size_t j = 0; j < per; ++j) fw.v[o * per + j] *= sc;
            fb.v[o] = be->v[o] - mu->v[o] * sc;
        }
        m->host["__vpconv" + si + ".weight"] = std::move(fw);
        m->host["__vpconv" + si + ".bias"] = std::move(fb);
        if (int rc = pack_conv(m, "", "__vpconv" + si, "vp.conv" + si)) return rc;
        cin = O;
    }
    // the consumer of torch.flatten(x, 1) of the (C, H, W) conv features (:605) -- lstm_velpred layer 0 when present
    // (:607-609), else fc_0 -- gets its input columns permuted to our HWC order
    const int L = c.enc_num_layers;
    const int Cf = L ? g.C[L - 1] : g.C0, Hf = L ? g.ph[L - 1] : g.H0, Wf = L ? g.pw[L - 1] : g.W0;
    std::vector<int> perm((size_t)Cf * Hf * Wf);
    for (int ch = 0; ch < Cf; ++ch)
        for (int p = 0; p < Hf * Wf; ++p) perm[(size_t)ch * Hf * Wf + p] = p * Cf + ch;
    int fin = Cf * Hf * Wf;
    m->vp_hidden = fin;
    for (int k = 0; k < c.velpred_lstm_layers; ++k) {
        // nn.LSTM(fin, fin): gate rows (i, f, g, o) are reordered to (i, f, o, g), the order of the shared cell kernel;
        // b_ih + b_hh ride on the input-side GEMM
        const std::string sk = std::to_string(k);
        const HostTensor *wi = m->find("lstm_velpred.weight_ih_l" + sk, kUnetP), *wh = m->find("lstm_velpred.weight_hh_l" + sk, kUnetP);
        const HostTensor *bi = m->find("lstm_velpred.bias_ih_l" + sk, kUnetP), *bh = m->find("lstm_velpred.bias_hh_l" + sk, kUnetP);
        if (!wi || !wh || !bi || !bh) return fail(-4, "missing lstm_velpred tensors of layer %d", k);
        const int H = fin;
        EVFLY_REQUIRE(wi->shape.size() == 2 && wi->shape[0] == 4 * H && wi->shape[1] == H && wh->shape[0] == 4 * H && wh->shape[1] == H,
                      "lstm_velpred layer %d: expected (%d,%d) weights", k, 4 * H, H);
        auto src_row = [&](int r) { const int gate = r / H, u = r % H; return (gate == 2 ? 3 : gate == 3 ? 2 : gate) * H + u; };
        HostTensor pi, ph, pb;
        pi.shape = {4 * H, H}; pi.v.resize((size_t)4 * H * H); ph = pi; pb.shape = {4 * H}; pb.v.resize(4 * H);
        for (int r = 0; r < 4 * H; ++r) {
            const int sr = src_row(r);
            for (int j = 0; j < H; ++j) {
                const int dj = (k == 0) ? perm[j] : j;
                pi.v[(size_t)r * H + dj] = wi->v[(size_t)sr * H + j];
                ph.v[(size_t)r * H + j] = wh->v[(size_t)sr * H + j];
            }
            pb.v[r] = bi->v[sr] + bh->v[sr];
        }
        m->host["__vplstm_ih" + sk + ".weight"] = std::move(pi);
        m->host["__vplstm_ih" + sk + ".bias"] = std::move(pb);
        m->host["__vplstm_hh" + sk + ".weight"] = std::move(ph);
        if (int rc = pack_linear(m, "", "__vplstm_ih" + sk, "vp.lstm.ih" + sk)) return rc;
        if (int rc = pack_linear(m, "", "__vplstm_hh" + sk, "vp.lstm.hh" + sk)) return rc;
    }
    for (int i = 0; i < c.fc_num_layers; ++i) {
        const std::string key = "velpred_head.fcnet.layers.fc_" + std::to_string(i);
        const HostTensor *w = m->find(key + ".weight", kUnetP);
        if (!w) return fail(-4, "missing tensor %s.weight", key.c_str());
        EVFLY_REQUIRE(w->shape.size() == 2 && w->shape[0] == c.fc_size[i] && w->shape[1] == fin,
                      "%s.weight: expected (%d,%d), got (%lld,%lld)", key.c_str(), c.fc_size[i], fin,
                      (long long)w->shape[0], (long long)(w->shape.size() > 1 ? w->shape[1] : 0));
        if (int rc = pack_linear(m, kUnetP, key, "vp.fc" + std::to_string(i), true,
                                 (i == 0 && c.velpred_lstm_layers == 0) ? &perm : nullptr)) return rc;
        fin = c.fc_size[i];
    }
    return 0;
}

int pack_vit(evfly_model *m) {
    const auto &c = m->cfg;
    const bool w16 = m->act16;      // bf16 pipeline: every GEMM whose input is a bf16 activation (C % 32 == 0) gets bf16 weights
    for (int s = 0; s < 2; ++s) {
        if (c.vit_layers[s] == 0 && s == 1) continue;   // single-stage handle
        const std::string P = "encoder_blocks." + std::to_string(s) + ".", N = "s" + std::to_string(s) + ".";
        if (int rc = pack_conv(m, kVitP, P + "patchMerge.cn1", N + "pm", true, false, w16)) return rc;
        if (int rc = pack_vec(m, kVitP, P + "patchMerge.layerNorm.weight", N + "pm.g")) return rc;
        if (int rc = pack_vec(m, kVitP, P + "patchMerge.layerNorm.bias", N + "pm.beta")) return rc;
        for (int l = 0; l < c.vit_layers[s]; ++l) {
            const std::string A = P + "_attn." + std::to_string(l) + ".", F = P + "_ffn." + std::to_string(l) + ".";
            const std::string NL = N + std::to_string(l) + ".";
            if (int rc = pack_conv(m, kVitP, A + "cn1", NL + "red", true, false, w16)) return rc;
            if (int rc = pack_vec(m, kVitP, A + "ln1.weight", NL + "ln1.g")) return rc;
            if (int rc = pack_vec(m, kVitP, A + "ln1.bias", NL + "ln1.beta")) return rc;
            if (int rc = pack_linear(m, kVitP, A + "keyValueExtractor", NL + "kv", true, nullptr, w16)) return rc;
            if (int rc = pack_linear(m, kVitP, A + "query", NL + "q", true, nullptr, w16)) return rc;
            if (int rc = pack_linear(m, kVitP, A + "finalLayer", NL + "fin", true, nullptr, w16)) return rc;
            if (int rc = pack_linear(m, kVitP, F + "mlp1", NL + "mlp1", true, nullptr, w16)) return rc;
            if (int rc = pack_vec(m, kVitP, F + "depthwise.weight", NL + "dw.w")) return rc;
            if (int rc = pack_vec(m, kVitP, F + "depthwise.bias", NL + "dw.b")) return rc;
            {   // the same weights laid out per group for the scalar-operand kernel (gconv.hip)
                const HostTensor *dwt = m->find(F + "depthwise.weight", kVitP);
                EVFLY_REQUIRE(dwt && dwt->shape.size() == 4 && dwt->shape[1] == 8 && dwt->shape[2] == 3 && dwt->shape[3] == 3 && dwt->shape[0] % 8 == 0,
                              "%sdepthwise.weight: expected (Ce, 8, 3, 3)", F.c_str());
                gconv_pack_host(dwt->v.data(), (int)dwt->shape[0], m->stage(NL + "dw.wp", dwt->v.size()));
            }
            if (w16) {   // fused MixFFN of the bf16 pipeline (mixffn16.hip): grouped-conv weights in lane order, two-term bf16 bias of mlp1
                const HostTensor *dwt = m->find(F + "depthwise.weight", kVitP), *dwb = m->find(F + "depthwise.bias", kVitP),
                                 *b1 = m->find(F + "mlp1.bias", kVitP);
                EVFLY_REQUIRE(dwb && b1 && dwb->v.size() == (size_t)dwt->shape[0] && b1->v.size() == (size_t)dwt->shape[0] && dwt->shape[0] % 32 == 0,
                              "%s: MixFFN biases do not match the hidden width", F.c_str());
                const int Ce = (int)dwt->shape[0];
                std::vector<float> wp(dwt->v.size());
                gconv_pack_host(dwt->v.data(), Ce, wp.data());
                std::vector<unsigned char> rec(mixffn16_rec_bytes(Ce), 0);
                std::vector<bf16_t> b1p((size_t)Ce * 8, 0);
                mixffn16_pack_host(wp.data(), dwb->v.data(), b1->v.data(), Ce, rec.data(), b1p.data());
                std::memcpy(m->stage(NL + "ffn.rec", (rec.size() + 3) / 4), rec.data(), rec.size());
                std::memcpy(m->stage16(NL + "mlp1.b16", b1p.size()), b1p.data(), b1p.size() * 2);
            }
            if (int rc = pack_linear(m, kVitP, F + "mlp2", NL + "mlp2", true, nullptr, w16)) return rc;
            if (int rc = pack_vec(m, kVitP, P + "_lNorm." + std::to_string(l) + ".weight", NL + "ln.g")) return rc;
            if (int rc = pack_vec(m, kVitP, P + "_lNorm." + std::to_string(l) + ".bias", NL + "ln.beta")) return rc;
        }
    }
    if (c.head == EVFLY_HEAD_NONE) return 0;
    if (int rc = pack_conv(m, kVitP, "down_sample", "ds", true, true, w16)) return rc;
    // decoder consumes out.flatten(1) of a (12,16,24) CHW tensor (vitfly_models.py:143); ours is HWC
    std::vector<int> perm(4608);
    for (int ch = 0; ch < 12; ++ch)
        for (int p = 0; p < 384; ++p) perm[ch * 384 + p] = p * 12 + ch;
    if (int rc = pack_linear(m, kVitP, "decoder", "dec", true, &perm, w16)) return rc;
    if (c.head == EVFLY_HEAD_LSTMNETVIT) {
        // layer 0 input side as a GEMM over all frames; biases b_ih + b_hh folded into its bias
        const HostTensor *wi0 = m->find("lstm.weight_ih_l0", kVitP);
        if (!wi0) return fail(-4, "missing tensor lstm.weight_ih_l0");
        const int G = (int)wi0->shape[0], I = (int)wi0->shape[1], ld = round_up(I, w16 ? 64 : 32);
        EVFLY_REQUIRE(G == 512 && I == 517, "lstm.weight_ih_l0: expected (512,517)");
        if (w16) {
            bf16_t *d0 = m->stage16("lstm.ih0.w", (size_t)G * ld);
            for (int g = 0; g < G; ++g)
                for (int i = 0; i < I; ++i) d0[(size_t)g * ld + i] = host_f2bf(wi0->v[(size_t)g * I + i]);
        } else {
            float *d0 = m->stage("lstm.ih0.w", (size_t)G * ld);
            for (int g = 0; g < G; ++g)
                for (int i = 0; i < I; ++i) d0[(size_t)g * ld + i] = wi0->v[(size_t)g * I + i];
        }
        m->wld["lstm.ih0"] = ld;
        for (int l = 0; l < 3; ++l) {
            const std::string sl = std::to_string(l);
            const HostTensor *bi = m->find("lstm.bias_ih_l" + sl, kVitP), *bh = m->find("lstm.bias_hh_l" + sl, kVitP);
            const HostTensor *wh = m->find("lstm.weight_hh_l" + sl, kVitP);
            if (!bi || !bh || !wh) return fail(-4, "missing nn.LSTM tensors of layer %d", l);
            float *b = m->stage("lstm.b" + sl, 512);
            for (int g = 0; g < 512; ++g) b[g] = bi->v[g] + bh->v[g];
            float *wt = m->stage("lstm.hh" + sl, 128 * 512);    // [k][gate row]
            for (int g = 0; g < 512; ++g)
                for (int k = 0; k < 128; ++k) wt[k * 512 + g] = wh->v[g * 128 + k];
            if (l > 0) {
                const HostTensor *wi = m->find("lstm.weight_ih_l" + sl, kVitP);
                if (!wi) return fail(-4, "missing lstm.weight_ih_l%d", l);
                float *it = m->stage("lstm.ih" + sl, 128 * 512);
                for (int g = 0; g < 512; ++g)
                    for (int k = 0; k < 128; ++k) it[k * 512 + g] = wi->v[g * 128 + k];
            }
        }
        if (int rc = pack_linear(m, kVitP, "nn_fc2", "fc2")) return rc;
    } else {
        if (int rc = pack_linear(m, kVitP, "nn_fc1", "fc1", true, nullptr, w16)) return rc;
        if (int rc = pack_linear(m, kVitP, "nn_fc2", "fc2", true, nullptr, w16)) return rc;
    }
    return 0;
}

// ============================================================================ forward helpers
struct Ctx {
    evfly_model *m;
    hipStream_t st;
};

// element types of a GEMM's tensors in the bf16 pipeline (fp32 pipelines pass 0)
enum : int { IN16 = 1, OUT16 = 2, RES16 = 4, IO16 = IN16 | OUT16 };

// y[N,OH,OW,Cout] = act(conv(x) + b (+res))
int conv(evfly_model *m, const char *pname, const std::string &wname, const float *x, int n, int H, int W, int C,
         int64_t ldx, int cout, int kh, int kw, int stride, int pad, int act, const float *res, int64_t ldres, float *y,
         int64_t ldy, float *y_pool = nullptr, bool *pool_fused = nullptr, float *skip_y = nullptr, int skip_h = 0, int skip_w = 0,
         int64_t skip_ld = 0, SkipGrid *skip_region = nullptr, int f16 = 0) {
    ConvDesc d;
    d.x = x; d.ldx = ldx; d.NI = n; d.H = H; d.W = W; d.C = C;
    d.in_bf16 = (f16 & IN16) != 0; d.out_bf16 = (f16 & OUT16) != 0; d.res_bf16 = (f16 & RES16) != 0;
    d.w = m->W(wname + ".w"); d.ldw = m->planning ? round_up(kh * kw * C, d.in_bf16 ? 64 : 32) : m->wld[wname];
    d.bias = m->W(wname + ".b");
    d.KH = kh; d.KW = kw; d.stride = stride; d.pad = pad;
    conv_finish(d);
    d.Nc = cout; d.res = res; d.ldres = ldres; d.act = act; d.y = y; d.ldy = ldy; d.dtype = m->cfg.compute_dtype;
    if (!m->planning && !d.w) return fail(-4, "weights '%s' were not loaded", wname.c_str());
    const double bytes = (d.in_bf16 ? 2.0 : 4.0) * ((double)n * H * W * C + (double)cout * d.K) + (d.out_bf16 ? 2.0 : 4.0) * (double)d.M * cout;
    const std::string pn = std::string(pname) + "/" + wname;   // family/layer: bench.py groups by family
    double extra_flops = 0;
    if (m->pre.frames) {                                       // first U-Net conv computed on the fly by the consumer
        d.pre_frames = m->pre.frames; d.pre_w = m->pre.w; d.pre_b = m->pre.b; d.pre_cin = m->pre.cin;
        d.pre_form_bev = m->pre.form_bev; d.pre_apply_form = m->pre.apply_form; d.pre_cutoff = m->pre.cutoff;
        extra_flops = 2.0 * n * H * W * C * 9 * m->pre.cin;
        m->pre.frames = nullptr;
    }
    const bool want_dot = m->dot.y != nullptr;
    float *const dot_y = m->dot.y;
    m->dot.y = nullptr; m->dot.done = false;
    if (!f16 && wino_applicable(d) && m->has(wname + ".u")) {  // Winograd F(2x2,3x3): 2.25x fewer MFMA flops
        if (want_dot && cout == 32 && ldy == 32 && !y_pool && !skip_y) {      // 1x1 consumer in the epilogue instead of the map
            d.dot_w = m->dot.w; d.dot_b = m->dot.b; d.dot_y = dot_y;
            m->dot.done = true;
            if (!m->planning) m->dot_used = true;
        }
        d.y_pool = y_pool;                                     // nn.MaxPool2d(2,2): one window per Winograd tile
        if (pool_fused) *pool_fused = y_pool != nullptr;
        if (skip_y && skip_region) {                           // 'interp' skip: resampled from the tile in LDS where the taps allow
            d.skip_y = skip_y; d.skip_h = skip_h; d.skip_w = skip_w; d.skip_ld = skip_ld;
            wino_skip_grid(d, skip_region);
            if (skip_region->rh0 == 0) d.skip_y = nullptr;
            // the map's other readers: the 2x2 pool (fused above) and the debug taps "e1".."e4"
            d.skip_bands = d.skip_y && y_pool && !m->full_encoder_outputs;
            if (d.skip_bands && !m->planning) m->bands_used = true;
        }
        if (!m->planning && m->profiling) {                                    // only a profiled launch consumes them
            m->next_exec = wino_exec_flops(d);
            m->next_useful = igemm_flops(d) * (16.0 / 36.0);
            m->next_launches = wino_launch_count(d);
        }
        RUN(m, pn.c_str(), igemm_flops(d) + extra_flops, bytes, wino_launch(d, m->W(wname + ".u"), m->st));
        return 0;
    }
    if (f16 == IO16 && m->has(wname + ".wd") && (m->planning || conv16_applicable(d))) {   // bf16 pipeline, C_in <= 64: direct conv from an LDS patch
        if (pool_fused) *pool_fused = y_pool != nullptr;
        if (want_dot && !y_pool && !skip_y && !m->planning) {      // 1x1 consumer in the epilogue instead of the map
            d.dot_w = m->dot.w; d.dot_b = m->dot.b; d.dot_y = dot_y;
            if (conv16_dot_fusable(d)) { m->dot.done = true; m->dot_used = true; }
            else d.dot_w = d.dot_b = nullptr, d.dot_y = nullptr;
        }
        // the map's readers are the fused pool and the decoder's 'interp' resize: only that resize's tap rows / columns are stored (e12: half
        // of the pixels, e22: 0.6; the debug taps "e1" / "e2" are then partial like the fp32 pipeline's band stores)
        static const bool no_tap_mask = getenv("EVFLY_NO_SKIP_TAP_STORES") != nullptr;
        if (y_pool && skip_y && skip_h > 0 && !m->full_encoder_outputs && !no_tap_mask && !d.dot_y && skip_h <= d.OH && skip_w <= d.OW) {
            d.tap_h = skip_h; d.tap_w = skip_w;
            if (!m->planning) m->bands_used = true;
        }
        RUN(m, pn.c_str(), igemm_flops(d) + extra_flops, bytes + (y_pool ? 0.5 * d.M * cout : 0.0), conv16_launch(d, m->W(wname + ".wd"), y_pool, m->st));
        return 0;
    }
    EVFLY_REQUIRE(!d.pre_frames, "the fused first conv needs the Winograd / direct-convolution kernels");
    if (pool_fused) *pool_fused = false;
    if (f16 == IO16 && !m->planning && conv16w_applicable(d)) {    // bf16 pipeline, deep 3x3 layers: 256-pixel tiles, 8 waves
        if (m->has(wname + ".wp")) d.w_patch = m->W(wname + ".wp");
        RUN(m, pn.c_str(), igemm_flops(d), bytes, conv16w_launch(d, m->st));
        return 0;
    }
    RUN(m, pn.c_str(), igemm_flops(d), bytes, igemm_launch(d, m->st));
    return 0;
}

int linear(evfly_model *m, const char *pname, const std::string &wname, const float *x, int64_t rows, int K, int64_t ldx,
           int cout, int act, const float *res, int64_t ldres, float *y, int64_t ldy, int f16 = 0) {
    // rows can exceed int range only for absurd batches; NI is an int
    return conv(m, pname, wname, x, (int)rows, 1, 1, K, ldx, cout, 1, 1, 1, 0, act, res, ldres, y, ldy, nullptr, nullptr, nullptr, 0, 0, 0,
                nullptr, f16);
}

}  // namespace

// ============================================================================ U-Net forward (one chunk)
static int velpred_chunk(evfly_model *m, const float *x, int S, int T, float *vp_h, float *vp_c, float *yvel);

static int unet_chunk(evfly_model *m, const float *frames, int S, int T, float *h_state, float *c_state,
                      float *depth_out, float *upconv_out, float **depth_dev, float *yvel_out = nullptr, float *vp_h = nullptr,
                      float *vp_c = nullptr) {
    const auto &c = m->cfg;
    const int F = S * T;
    if (!m->planning) { m->bands_used = false; m->dot_used = false; }
    const int cin = (c.form_bev == 1 || c.form_bev == 2) ? 1 : c.num_in_channels;
    const int apply_form = (c.num_in_channels == 2 || c.form_bev > 0) ? 1 : 0;   // learner_models.py:523
    hipStream_t st = m->st;
    const bool a16 = m->act16;                 // bf16 pipeline: every activation below is bf16 unless it says fp32
    const int io = a16 ? IO16 : 0;

    // ---- encoder (valid 3x3 convs; sizes of learner_models.py:373-390)
    // exact-fp32 mode: e11 (1 or 2 -> 32 channels, HBM-write-bound: 11.4 MB per frame) is never materialised; the
    // Winograd kernel of e12 computes its input patch from the raw frame while staging it
    static const bool no_fuse = getenv("EVFLY_NO_E11_FUSION") != nullptr;
    // (bf16 pipeline: the direct-convolution kernel of e12 has the same producer, for one frame channel)
    const bool fuse_e11 = !no_fuse && ((c.compute_dtype == EVFLY_DTYPE_F32 && m->has("e12.u")) ||
                                       (m->act16 && m->has("e12.wd") && ((c.form_bev == 1 || c.form_bev == 2) ? 1 : c.num_in_channels) == 1 &&
                                        getenv("EVFLY_NO_CONV16") == nullptr));
    float *e11 = fuse_e11 ? nullptr : m->alloc_act((int64_t)F * 258 * 344 * 32);
    if (!fuse_e11 && a16)
        RUN(m, "e11_direct", 2.0 * F * 258 * 344 * 32 * 9 * cin, F * (4.0 * 260 * 346 + 2.0 * 258 * 344 * 32),
            launch16_e11(frames, F, 260, 346, cin, c.form_bev, apply_form, c.evs_min_cutoff, m->W("e11.w"), m->W("e11.b"), e11, st));
    else if (!fuse_e11)
    RUN(m, "e11_direct", 2.0 * F * 258 * 344 * 32 * 9 * cin, 4.0 * F * (260.0 * 346 + 258.0 * 344 * 32),
        launch_e11(frames, F, 260, 346, cin, c.form_bev, apply_form, c.evs_min_cutoff, m->W("e11.w"), m->W("e11.b"), e11, st));
    struct Lvl { int H, W, C; float *y; } lv[5];
    // 'interp' skips: the concat buffers of the decoder are allocated up front so that the encoder's Winograd kernels can
    // write the resampled skip straight into them (every pixel whose taps lie inside one block's tile; the resize
    // kernel in the decoder loop below writes the rest). EVFLY_NO_SKIP_FUSION: the resize kernel writes everything.
    static const int small[4][2] = {{16, 26}, {24, 44}, {40, 80}, {72, 152}};
    static const bool no_skip_fuse = getenv("EVFLY_NO_SKIP_FUSION") != nullptr;
    static const bool no_dot_fuse = getenv("EVFLY_NO_OUT_FUSION") != nullptr;      // A/B switch: unet_out as its own kernel
    float *up_fused = nullptr;                                                      // unet_out's map when d42's kernel wrote it
    const bool run_decoder = !(c.is_deployment && !(c.velpred == 1 || c.velpred == 11));
    float *cats[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};       // by decoder level 1..4
    bool skip_done[5] = {false, false, false, false, false};              // the level's skip is already in its concat buffer (bf16: pool + skip kernel)
    static const bool no_pool_skip = getenv("EVFLY_NO_POOL_SKIP_FUSION") != nullptr;      // A/B switch
    SkipGrid skip_region[5];                                               // block regions of the fused producer (rh0 == 0: not fused)
    if (run_decoder && c.skip_type == EVFLY_SKIP_INTERP && !no_skip_fuse)
        for (int l = 1; l <= 4; ++l) cats[l] = m->alloc_act((int64_t)F * small[l - 1][0] * small[l - 1][1] * 2 * (512 >> l));
    const float *cur = e11;
    int H = 258, W = 344, C = 32;
    const char *names[5][2] = {{nullptr, "e12"}, {"e21", "e22"}, {"e31", "e32"}, {"e41", "e42"}, {"e51", "e52"}};
    const int chans[5] = {32, 64, 128, 256, 512};
    float *pooled = nullptr;         // pool of the previous level's output (already filled when the conv fused it)
    bool pool_done = false;
    for (int l = 0; l < 5; ++l) {
        if (l > 0) {
            float *p = pooled;
            // bf16 pipeline, levels 3 / 4 (their conv kernels fuse neither the pool nor the 'interp' skip): both in one pass over the map, the
            // skip written into the decoder's concat buffer now instead of a ConvLSTM later
            const int sdl = 4 - (l - 1);
            if (!pool_done && a16 && !no_pool_skip && cats[sdl]) {
                RUN(m, "maxpool", 0, 2.0 * F * (H * W * C * 1.25 + small[sdl - 1][0] * small[sdl - 1][1] * C),
                    launch16_pool_bilinear(cur, F, H, W, C, p, cats[sdl], small[sdl - 1][0], small[sdl - 1][1], 2 * C, st));
                skip_done[sdl] = true;
            } else
            if (!pool_done && a16) RUN(m, "maxpool", 0, 2.0 * F * H * W * C * 1.25, launch16_maxpool2x2(cur, F, H, W, C, p, st));
            else if (!pool_done) RUN(m, "maxpool", 0, 4.0 * F * H * W * C * 1.25, launch_maxpool2x2(cur, F, H, W, C, p, st));
            cur = p; H /= 2; W /= 2;
            float *a = m->alloc_act((int64_t)F * (H - 2) * (W - 2) * chans[l]);
            if (int rc = conv(m, "conv3x3", names[l][0], cur, F, H, W, C, C, chans[l], 3, 3, 1, 0, ACT_RELU, nullptr, 0, a, chans[l],
                              nullptr, nullptr, nullptr, 0, 0, 0, nullptr, io)) return rc;
            cur = a; H -= 2; W -= 2; C = chans[l];
        }
        float *b = m->alloc_act((int64_t)F * (H - 2) * (W - 2) * chans[l]);
        pooled = l < 4 ? m->alloc_act((int64_t)F * ((H - 2) / 2) * ((W - 2) / 2) * chans[l]) : nullptr;
        pool_done = false;
        if (l == 0 && fuse_e11) {
            m->pre.frames = frames; m->pre.w = m->W("e11.w"); m->pre.b = m->W("e11.b"); m->pre.cin = cin;
            m->pre.form_bev = c.form_bev; m->pre.apply_form = apply_form; m->pre.cutoff = c.evs_min_cutoff;
        }
        const int dl = 4 - l;       // the decoder level this output is the skip of (levels 0..3)
        float *skip_dst = l < 4 ? cats[dl] : nullptr;
        if (int rc = conv(m, "conv3x3", names[l][1], cur, F, H, W, C, C, chans[l], 3, 3, 1, 0, ACT_RELU, nullptr, 0, b, chans[l],
                          pooled, &pool_done, skip_dst, skip_dst ? small[dl - 1][0] : 0, skip_dst ? small[dl - 1][1] : 0, 2 * chans[l],
                          skip_dst ? &skip_region[dl] : nullptr, io)) return rc;
        cur = b; H -= 2; W -= 2; C = chans[l];
        lv[l] = Lvl{H, W, C, b};
        static const char *tn[5] = {"e1", "e2", "e3", "e4", "e5"};
        m->tap(tn[l], b, F, H, W, C, a16);
    }
    // ---- ConvLSTM bottleneck (batch-as-time: the T frames of a stream are its time steps)
    float *y5 = lv[4].y;   // (F, 8, 13, 512)
    if (c.num_recurrent_unet > 0) {
        const int rpi = 8 * 13, hid = 512;
        float *zx = m->alloc((int64_t)F * rpi * 4 * hid);          // fp32 pre-activations in every pipeline
        // bf16 pipeline: the T steps of a chunk in ONE launch (clstm16.hip: a 1x1 ConvLSTM is an independent LSTM per position) on
        // gate-interleaved pre-activations; otherwise a GEMM + a gate launch per step
        // (chosen from the forward's first chunk, see clstm_rows_first; the kernel's 32-bit byte offsets bound the chunk: zx < 4 GB)
        const int64_t rows_first = m->clstm_rows_first > 0 ? m->clstm_rows_first : (int64_t)S * rpi;
        // up to 7 168 state rows per chunk (C5: 2 080, C3: 6 656): the cooperative form -- weights resident in the registers of groups of 16 CUs,
        // h handed over through the h sequence once per step (k_clstm16_coop); more rows: a workgroup per 64 rows, weights streamed
        // (one path per forward: a tail chunk of fewer rows than the kernel's lower bound stays on it -- groups without rows leave at once --, so a
        // stream's result does not depend on which chunk it falls into beyond the three kernels' tested bit-equality)
        const bool coop = a16 && m->has("clstm.wh_i") && clstm16_coop_available(rows_first, T) && (int64_t)S * rpi <= rows_first &&
                          (int64_t)F * rpi * 4 * hid * 4 < ((int64_t)1 << 32);
        const bool seq = !coop && a16 && m->has("clstm.wh_i") && clstm16_seq_available(rows_first) && (int64_t)F * rpi * 4 * hid * 4 < ((int64_t)1 << 32);
        // small chunks: per-step launches, the cell update in the hidden-side GEMM's epilogue (igemm16 OUT_LSTM) on the same interleaved
        // layout -- the step's fp32 pre-activations no longer go through HBM twice and the gate launch is gone
        static const bool no_gate_fusion = getenv("EVFLY_NO_CLSTM16_GATE_FUSION") != nullptr;
        const bool gfuse = a16 && !seq && !coop && m->has("clstm.wh_ir") && !no_gate_fusion;
        const bool il = coop || seq || gfuse;                              // interleaved pre-activations
        {   // input-side 1x1 conv for every frame at once
            ConvDesc d; d.x = y5; d.ldx = hid; d.NI = F * rpi; d.C = hid; d.w = m->W(il ? "clstm.wx_i" : "clstm.wx"); d.ldw = hid;
            conv_finish(d); d.Nc = 4 * hid; d.y = zx; d.ldy = 4 * hid; d.dtype = c.compute_dtype; d.in_bf16 = a16;
            RUN(m, "convlstm_x_gemm", igemm_flops(d), d.M * (a16 ? 2.0 : 4.0) * hid + d.M * 16.0 * hid + (a16 ? 2.0 : 4.0) * 4.0 * hid * hid,
                (a16 && !m->planning && conv16w_gemm_applicable(d)) ? conv16w_gemm_launch(d, st) : igemm_launch(d, st));
        }
        float *z = m->alloc((int64_t)S * rpi * 4 * hid);
        float *hseq = m->alloc_act((int64_t)F * rpi * hid);
        float *hs = h_state, *cs = c_state;
        if (!hs) {
            hs = m->alloc((int64_t)S * rpi * hid); cs = m->alloc((int64_t)S * rpi * hid);
            if (!m->planning) {
                EVFLY_HIP(hipMemsetAsync(hs, 0, (size_t)S * rpi * hid * 4, st));
                EVFLY_HIP(hipMemsetAsync(cs, 0, (size_t)S * rpi * hid * 4, st));
            }
        }
        // bf16 pipeline: the hidden-side GEMM reads a bf16 copy of h that the gate kernel keeps beside the fp32 state
        // The per-step path keeps TWO copies and alternates: with the cell update in the GEMM's epilogue (OUT_LSTM) the launch that
        // reads h(t - 1) as its A operand also writes h(t), and a block that starts late -- a grid of more blocks than the chip holds, or
        // a chip shared with another stream's kernels -- must not find rows its neighbours have already replaced
        float *h16 = a16 ? m->alloc_act((int64_t)S * rpi * hid) : nullptr;
        float *h16_alt = a16 && !seq && !coop ? m->alloc_act((int64_t)S * rpi * hid) : nullptr;
        if (a16 && !m->planning) {
            if (int rc = launch_f32_to_bf16(hs, (int64_t)S * rpi * hid, h16, st)) return rc;
        }
        if (coop) {
            float *ws = m->alloc((int64_t)clstm16_coop_scratch_words(T));
            float *save = h_state ? m->alloc((int64_t)2 * S * rpi * hid) : nullptr;      // the stand-by's copy of the incoming state
            RUN(m, "convlstm_seq", 2.0 * S * rpi * 4.0 * hid * hid * (h_state ? T : T - 1), (double)F * rpi * hid * (16.0 + 2.0 + 2.0) + 2.0 * 4.0 * hid * hid,
                launch_clstm16_coop(zx, m->W("clstm.wh_i"), S, T, rpi, hs, cs, h16, hseq, !h_state, ws, save, st));
        } else if (seq)
            RUN(m, "convlstm_seq", 2.0 * S * rpi * 4.0 * hid * hid * (h_state ? T : T - 1), (double)F * rpi * hid * (16.0 + 2.0) + 2.0 * 4.0 * hid * hid,
                launch_clstm16_seq(zx, m->W("clstm.wh_i"), S, T, rpi, hs, cs, h16, hseq, !h_state, st));
        else
        for (int t = 0; t < T; ++t) {
            // fresh streams (no incoming state): h0 = 0, so conv(h0) contributes exactly nothing to step 0 -- its GEMM is skipped
            // and the cell reads zx[:, 0] in place
            const bool skip0 = t == 0 && !h_state;
            const float *zt = skip0 ? zx : z;
            const int64_t zrows = skip0 ? (int64_t)T * rpi : 0;
            float *h16_in = h16, *h16_out = a16 ? h16_alt : nullptr;      // step t reads h16_in, writes h16_out
#ifdef EVFLY_CLSTM_INPLACE_H16      // (variant build only: the single-copy form, to show that the test below it catches the hazard)
            h16_out = h16;
#else
            std::swap(h16, h16_alt);
#endif
            ConvDesc d; d.x = a16 ? h16_in : hs; d.ldx = hid; d.NI = S * rpi; d.C = hid; d.w = m->W(gfuse ? "clstm.wh_ir" : "clstm.wh"); d.ldw = hid;
            conv_finish(d); d.Nc = 4 * hid; d.y = z; d.ldy = 4 * hid; d.dtype = c.compute_dtype; d.in_bf16 = a16;
            d.res = zx + (int64_t)t * rpi * 4 * hid; d.ldres = 4 * hid; d.res_rpi = rpi; d.res_img_rows = (int64_t)T * rpi;
            if (gfuse && !skip0) {
                d.out_mode = OUT_LSTM; d.lstm_c = cs; d.lstm_h = t == T - 1 ? hs : nullptr /* nobody reads the fp32 h between steps */; d.lstm_h16 = h16_out; d.lstm_hseq = m->eoff(hseq, (int64_t)t * rpi * hid);
                d.lstm_seq_img_rows = (int64_t)T * rpi;
                RUN(m, "convlstm_h_gemm", igemm_flops(d), d.M * 2.0 * hid + d.M * 16.0 * hid + d.M * (16.0 + 4.0) * hid + 2.0 * 4.0 * hid * hid, igemm_launch(d, st));
                continue;
            }
            if (!skip0)
            RUN(m, "convlstm_h_gemm", igemm_flops(d), d.M * (a16 ? 2.0 : 4.0) * hid + d.M * 32.0 * hid + (a16 ? 2.0 : 4.0) * 4.0 * hid * hid,
                igemm_launch(d, st));
            if (a16)
                RUN(m, "convlstm_gates", 0, 4.0 * S * rpi * hid * 8,
                    launch16_convlstm_gates(zt, (int64_t)S * rpi, hid, cs, hs, h16_out, m->eoff(hseq, (int64_t)t * rpi * hid), rpi, (int64_t)T * rpi, st, zrows, gfuse));
            else
            RUN(m, "convlstm_gates", 0, 4.0 * S * rpi * hid * 8,
                launch_convlstm_gates(zt, (int64_t)S * rpi, hid, cs, hs, hseq + (int64_t)t * rpi * hid, rpi, (int64_t)T * rpi, st, zrows));
        }
        y5 = hseq;
        m->tap("e5_lstm", hseq, F, 8, 13, 512, a16);
    }
    // ---- decoder (learner_models.py:553-583); is_deployment skips it unless a velpred head reads its output
    // the velpred head runs on fp32 activations in every pipeline (its inputs are fp32 images except y_e5)
    auto y5_f32 = [&](const float **out) -> int {
        *out = y5;
        if (!a16) return 0;
        float *t = m->alloc((int64_t)F * 8 * 13 * 512);
        if (!m->planning) { if (int rc = launch_bf16_to_f32(y5, (int64_t)F * 8 * 13 * 512, t, st)) return rc; }
        *out = t;
        return 0;
    };
    if (!run_decoder) {
        if (depth_dev) *depth_dev = nullptr;
        if (c.velpred == 2 && yvel_out) {
            const float *y5f = nullptr;
            if (int rc = y5_f32(&y5f)) return rc;
            return velpred_chunk(m, y5f, S, T, vp_h, vp_c, yvel_out);
        }
        return 0;
    }
    const float *dcur = y5;
    int dh = 8, dw = 13, dc = 512;
    for (int l = 1; l <= 4; ++l) {
        const Lvl &enc = lv[4 - l];
        const int co = dc / 2, uh = 2 * dh, uw = 2 * dw;
        EVFLY_REQUIRE(uh == small[l - 1][0] && uw == small[l - 1][1], "decoder geometry");
        const int ccat = c.skip_type == EVFLY_SKIP_NONE ? co : 2 * co;
        float *cat = cats[l] ? cats[l] : m->alloc_act((int64_t)F * uh * uw * ccat);
        float *up_dst = m->eoff(cat, ccat - co);
        if (c.skip_type == EVFLY_SKIP_INTERP && a16 && skip_done[l]) {
        } else if (c.skip_type == EVFLY_SKIP_INTERP && a16)
            RUN(m, "skip_bilinear", 0, 2.0 * F * uh * uw * co * 5, launch16_bilinear(enc.y, F, enc.H, enc.W, enc.C, enc.C, cat, uh, uw, ccat, 0, st));
        else if (c.skip_type == EVFLY_SKIP_CROP && a16)
            RUN(m, "skip_crop", 0, 2.0 * F * uh * uw * co * 2,
                launch16_crop(enc.y, F, enc.H, enc.W, enc.C, enc.H / 2 - uh / 2, enc.W / 2 - uw / 2, cat, uh, uw, ccat, st));
        else if (c.skip_type == EVFLY_SKIP_INTERP)  // F.interpolate(y, size=small, bilinear, align_corners=False) (:514)
        {
            // share of the skip pixels left to the resize kernel (taps in two producer blocks): rows / columns whose second tap
            // starts a block region, counted with the resize's own index arithmetic (for the bytes figure of the profile only)
            auto straddling = [](int in, int out, int region) {
                if (region <= 0) return out;
                const float sc = (float)in / (float)out;
                int cnt = 0;
                for (int o = 0; o < out; ++o) {
                    const float real = std::max(sc * ((float)o + 0.5f) - 0.5f, 0.f);
                    const int i0 = std::min((int)std::floor(real), in - 1), i1 = i0 + (i0 < in - 1 ? 1 : 0);
                    cnt += i1 != i0 && i1 % region == 0;
                }
                return cnt;
            };
            const int rs = straddling(enc.H, uh, skip_region[l].rh0), cs = straddling(enc.W, uw, skip_region[l].rw0);
            const double rest = (double)uh * uw - (double)(uh - rs) * (uw - cs);
            RUN(m, "skip_bilinear", 0, 4.0 * F * rest * co * 5,
                launch_bilinear(enc.y, F, enc.H, enc.W, enc.C, enc.C, cat, uh, uw, ccat, 0, 0, st, skip_region[l]));
        }
        else if (c.skip_type == EVFLY_SKIP_CROP)    // centre crop (:512)
            RUN(m, "skip_crop", 0, 4.0 * F * uh * uw * co * 2,
                launch_crop(enc.y, F, enc.H, enc.W, enc.C, enc.H / 2 - uh / 2, enc.W / 2 - uw / 2, cat, uh, uw, ccat, st));
        {   // ConvTranspose2d(k=2, s=2) as one GEMM with a 2x2 scatter epilogue into the concat buffer
            const std::string un = "up" + std::to_string(l);
            ConvDesc d; d.x = dcur; d.ldx = dc; d.NI = F; d.H = dh; d.W = dw; d.C = dc; d.w = m->W(un + ".w");
            d.ldw = m->planning ? dc : m->wld[un]; d.bias = m->W(un + ".b");
            conv_finish(d); d.Nc = 4 * co; d.y = up_dst; d.ldy = ccat; d.out_mode = OUT_UPCONV2X2; d.up_cout = co;
            d.dtype = c.compute_dtype; d.in_bf16 = d.out_bf16 = a16;
            RUN(m, ("upconv2x2/" + un).c_str(), igemm_flops(d), (a16 ? 2.0 : 4.0) * F * dh * dw * (dc + 4.0 * co),
                (a16 && !m->planning && conv16w_up_applicable(d)) ? conv16w_up_launch(d, st) : igemm_launch(d, st));
        }
        float *a = m->alloc_act((int64_t)F * (uh - 2) * (uw - 2) * co);
        const std::string n1 = "d" + std::to_string(l) + "1", n2 = "d" + std::to_string(l) + "2";
        if (int rc = conv(m, "conv3x3", n1, cat, F, uh, uw, ccat, ccat, co, 3, 3, 1, 0, ACT_RELU, nullptr, 0, a, co, nullptr, nullptr, nullptr,
                          0, 0, 0, nullptr, io)) return rc;
        float *b = m->alloc_act((int64_t)F * (uh - 4) * (uw - 4) * co);
        if (l == 4 && !m->full_encoder_outputs && !no_dot_fuse) {      // unet_out in d42's epilogue (fp32: Winograd kernel; bf16: conv16.hip)
            up_fused = upconv_out ? upconv_out : m->alloc((int64_t)F * 68 * 148);
            m->dot.w = m->W("out.w"); m->dot.b = m->W("out.b"); m->dot.y = up_fused;
        }
        if (int rc = conv(m, "conv3x3", n2, a, F, uh - 2, uw - 2, co, co, co, 3, 3, 1, 0, ACT_RELU, nullptr, 0, b, co, nullptr, nullptr, nullptr,
                          0, 0, 0, nullptr, io)) return rc;
        if (l == 4 && !m->dot.done) up_fused = nullptr;
        dcur = b; dh = uh - 4; dw = uw - 4; dc = co;
        static const char *tn[4] = {"d1", "d2", "d3", "d4"};
        m->tap(tn[l - 1], b, F, dh, dw, dc, a16);
    }
    // ---- unet_out (1x1, 32 -> 1) and form_output (:496-508)
    float *up = up_fused ? up_fused : upconv_out ? upconv_out : m->alloc((int64_t)F * 68 * 148);
    if (up_fused) {}
    else if (a16) RUN(m, "unet_out", 2.0 * F * 68 * 148 * 32, F * 68.0 * 148 * (2.0 * 32 + 4), launch16_dot_out(dcur, (int64_t)F * 68 * 148, 32, m->W("out.w"), m->W("out.b"), up, st));
    else
    RUN(m, "unet_out", 2.0 * F * 68 * 148 * 32, 4.0 * F * 68 * 148 * 33, launch_dot_out(dcur, (int64_t)F * 68 * 148, 32, m->W("out.w"), m->W("out.b"), up, st));
    float *dp = depth_out ? depth_out : m->alloc((int64_t)F * c.input_h * c.input_w);
    RUN(m, "depth_bilinear", 0, 4.0 * F * c.input_h * c.input_w * 2, launch_bilinear(up, F, 68, 148, 1, 1, dp, c.input_h, c.input_w, 1, 0, 0, st));
    if (depth_dev) *depth_dev = dp;
    if (c.velpred > 0 && yvel_out) {
        const float *y5f = nullptr;
        if (c.velpred == 2) { if (int rc = y5_f32(&y5f)) return rc; }
        return velpred_chunk(m, c.velpred == 1 ? dp : c.velpred == 11 ? up : y5f, S, T, vp_h, vp_c, yvel_out);
    }
    return 0;
}

// ============================================================================ velpred head (OrigUNet, velpred > 0)
// learner_models.py:593-614: convnet_velpred(y_interp | y_upconv | y_e5) -> flatten -> velpred_head
static int velpred_chunk(evfly_model *m, const float *x, int S, int T, float *vp_h, float *vp_c, float *yvel) {
    const auto &c = m->cfg;
    const int F = S * T;
    VpGeom g;
    if (int rc = velpred_geometry(c, g)) return rc;
    hipStream_t st = m->st;
    const float *cur = x;
    int H = g.H0, W = g.W0, C = g.C0;
    for (int i = 0; i < c.enc_num_layers; ++i) {
        const std::string wn = "vp.conv" + std::to_string(i);
        float *a = m->alloc((int64_t)F * g.ch[i] * g.cw[i] * g.C[i]);
        if (int rc = conv(m, "velpred_conv", wn, cur, F, H, W, C, C, g.C[i], c.enc_kernel[i], c.enc_kernel[i], c.enc_stride[i], 0,
                          c.enc_act[i], nullptr, 0, a, g.C[i])) return rc;
        cur = a; H = g.ch[i]; W = g.cw[i]; C = g.C[i];
        // one InvertLayer survives, in front of the pool (both are registered as 'invert_i', :77-92)
        const bool pool = c.enc_pool_type != EVFLY_POOL_NONE;
        if (pool || c.enc_invert_pool_inputs) {
            float *b = m->alloc((int64_t)F * g.ph[i] * g.pw[i] * C);
            RUN(m, "velpred_pool", 0, 4.0 * F * H * W * C * 1.25,
                launch_pool2d(cur, F, H, W, C, pool ? c.enc_pool_kernel[i] : 1, pool ? c.enc_pool_stride[i] : 1,
                              pool ? c.enc_pool_type : EVFLY_POOL_MAX, c.enc_invert_pool_inputs, b, st));
            cur = b; H = g.ph[i]; W = g.pw[i];
        }
    }
    m->tap("velpred_enc", const_cast<float *>(cur), F, H, W, C);
    int fin = H * W * C;
    // ---- lstm_velpred (:607-609): nn.LSTM(fin, fin, num_layers) over the T frames of each stream (batch-as-time)
    for (int k = 0; k < c.velpred_lstm_layers; ++k) {
        const int Hd = fin;
        const std::string sk = std::to_string(k);
        float *xg = m->alloc((int64_t)F * 4 * Hd);
        if (int rc = linear(m, "velpred_lstm_x", "vp.lstm.ih" + sk, cur, F, Hd, Hd, 4 * Hd, ACT_NONE, nullptr, 0, xg, 4 * Hd)) return rc;
        float *z = m->alloc((int64_t)S * 4 * Hd), *hseq = m->alloc((int64_t)F * Hd);
        float *hs = m->alloc((int64_t)S * Hd), *cs = m->alloc((int64_t)S * Hd);     // working state, (S, Hd) contiguous
        if (!m->planning) {
            if (vp_h) {   // caller state is (S, layers, Hd)
                EVFLY_HIP(hipMemcpy2DAsync(hs, (size_t)Hd * 4, vp_h + (int64_t)k * Hd, (size_t)c.velpred_lstm_layers * Hd * 4, (size_t)Hd * 4, S, hipMemcpyDeviceToDevice, st));
                EVFLY_HIP(hipMemcpy2DAsync(cs, (size_t)Hd * 4, vp_c + (int64_t)k * Hd, (size_t)c.velpred_lstm_layers * Hd * 4, (size_t)Hd * 4, S, hipMemcpyDeviceToDevice, st));
            } else {
                EVFLY_HIP(hipMemsetAsync(hs, 0, (size_t)S * Hd * 4, st));
                EVFLY_HIP(hipMemsetAsync(cs, 0, (size_t)S * Hd * 4, st));
            }
        }
        for (int t = 0; t < T; ++t) {
            ConvDesc d; d.x = hs; d.ldx = Hd; d.NI = S; d.C = Hd; d.w = m->W("vp.lstm.hh" + sk + ".w");
            d.ldw = m->planning ? round_up(Hd, 32) : m->wld["vp.lstm.hh" + sk];
            conv_finish(d); d.Nc = 4 * Hd; d.y = z; d.ldy = 4 * Hd; d.dtype = c.compute_dtype;
            d.res = xg + (int64_t)t * 4 * Hd; d.ldres = 4 * Hd; d.res_rpi = 1; d.res_img_rows = T;
            RUN(m, "velpred_lstm_h", igemm_flops(d), 4.0 * (4.0 * Hd * Hd + S * 9.0 * Hd), igemm_launch(d, st));
            RUN(m, "velpred_lstm_cell", 0, 4.0 * S * Hd * 8,
                launch_convlstm_gates(z, S, Hd, cs, hs, hseq + (int64_t)t * Hd, 1, T, st));
        }
        if (vp_h && !m->planning) {
            EVFLY_HIP(hipMemcpy2DAsync(vp_h + (int64_t)k * Hd, (size_t)c.velpred_lstm_layers * Hd * 4, hs, (size_t)Hd * 4, (size_t)Hd * 4, S, hipMemcpyDeviceToDevice, st));
            EVFLY_HIP(hipMemcpy2DAsync(vp_c + (int64_t)k * Hd, (size_t)c.velpred_lstm_layers * Hd * 4, cs, (size_t)Hd * 4, (size_t)Hd * 4, S, hipMemcpyDeviceToDevice, st));
        }
        cur = hseq;
    }
    for (int i = 0; i < c.fc_num_layers; ++i) {
        float *y = m->alloc((int64_t)F * c.fc_size[i]);
        if (int rc = linear(m, "velpred_fc", "vp.fc" + std::to_string(i), cur, F, fin, fin, c.fc_size[i], c.fc_act[i], nullptr, 0, y,
                            c.fc_size[i])) return rc;
        cur = y; fin = c.fc_size[i];
    }
    RUN(m, "velpred_vec", 0, 16.0 * F, launch_velpred_vec(cur, F, fin, yvel, st));
    return 0;
}

// ============================================================================ Mix-Transformer stage
// x16: the stage input is a bf16 activation (bf16 pipeline, stage 2 / a stand-alone stage with C_in % 32 == 0); a stage fed
// by the fp32 depth image (stage 1, C_in = 1) reads fp32 and writes bf16.
static int vit_stage(evfly_model *m, int s, const float *x, int n, int H, int W, int Cin, float *y_out, float **y_ret,
                     int *Ho, int *Wo, bool x16 = false) {
    const auto &c = m->cfg;
    hipStream_t st = m->st;
    const bool a16 = m->act16;
    const int io = a16 ? IO16 : 0;
    const double eb = a16 ? 2.0 : 4.0;
    const int C = c.vit_width[s], heads = c.vit_heads[s], R = c.vit_reduction[s], E = C * c.vit_expansion;
    const int k = c.vit_patch[s], sd = c.vit_stride[s], pd = c.vit_pad[s];
    const int h = (H + 2 * pd - k) / sd + 1, w = (W + 2 * pd - k) / sd + 1;
    const int64_t rows = (int64_t)n * h * w;
    const std::string N = "s" + std::to_string(s) + ".";
    auto layernorm = [&](const float *src, int64_t nrows, const std::string &g, const std::string &bta, float *dst) -> int {
        if (a16) RUN(m, "vit_layernorm", 0, 2 * eb * nrows * C, launch16_layernorm(src, nrows, C, m->W(g), m->W(bta), dst, st));
        else RUN(m, "vit_layernorm", 0, 8.0 * nrows * C, launch_layernorm(src, nullptr, nrows, C, m->W(g), m->W(bta), dst, st));
        return 0;
    };
    // OverlapPatchMerging: conv + LayerNorm (ViTsubmodules.py:30-33)
    float *t0 = m->alloc_act(rows * C);
    if (int rc = conv(m, "vit_patch_conv", N + "pm", x, n, H, W, Cin, Cin, C, k, k, sd, pd, ACT_NONE, nullptr, 0, t0, C, nullptr, nullptr, nullptr,
                      0, 0, 0, nullptr, a16 ? (OUT16 | (x16 ? IN16 : 0)) : 0)) return rc;
    float *xcur = m->alloc_act(rows * C);
    if (int rc = layernorm(t0, rows, N + "pm.g", N + "pm.beta", xcur)) return rc;
    const int rh = (h - R) / R + 1, rw = (w - R) / R + 1, nkv = rh * rw;
    for (int l = 0; l < c.vit_layers[s]; ++l) {
        const std::string NL = N + std::to_string(l) + ".";
        // --- EfficientSelfAttention (:54-83)
        float *red = m->alloc_act((int64_t)n * nkv * C);
        if (int rc = conv(m, "vit_kv_reduce_conv", NL + "red", xcur, n, h, w, C, C, C, R, R, R, 0, ACT_NONE, nullptr, 0, red, C, nullptr, nullptr,
                          nullptr, 0, 0, 0, nullptr, io)) return rc;
        float *redn = m->alloc_act((int64_t)n * nkv * C);
        if (int rc = layernorm(red, (int64_t)n * nkv, NL + "ln1.g", NL + "ln1.beta", redn)) return rc;
        float *kv = m->alloc_act((int64_t)n * nkv * 2 * C);
        if (int rc = linear(m, "vit_linear", NL + "kv", redn, (int64_t)n * nkv, C, C, 2 * C, ACT_NONE, nullptr, 0, kv, 2 * C, io)) return rc;
        float *att = m->alloc_act(rows * C);
        // bf16 pipeline, ViT-base widths: the attention in the query projection's epilogue (igemm16 OUT_ATTN: a thread per (token, head) of the tile, the
        // attention kernel's arithmetic on the rounded q) -- q is neither written nor read back, one launch less; same bits
        static const bool no_attn_fuse = getenv("EVFLY_NO_ATTN_FUSION") != nullptr;      // A/B switch
        if (a16 && !no_attn_fuse && C % 128 == 0 && C / heads == 32 && nkv <= 16) {
            ConvDesc d; d.x = xcur; d.ldx = C; d.NI = (int)rows; d.H = 1; d.W = 1; d.C = C; d.in_bf16 = d.out_bf16 = 1;
            d.w = m->W(NL + "q.w"); d.ldw = m->planning ? round_up(C, 64) : m->wld[NL + "q"]; d.bias = m->W(NL + "q.b");
            conv_finish(d); d.Nc = C; d.y = att; d.ldy = C; d.dtype = c.compute_dtype;
            d.out_mode = OUT_ATTN; d.attn_kv = kv; d.attn_nkv = nkv; d.attn_n = h * w;
            if (!m->planning && !d.w) return fail(-4, "weights '%s' were not loaded", (NL + "q").c_str());
            RUN(m, ("vit_linear/" + NL + "q").c_str(), igemm_flops(d) + 4.0 * rows * C * nkv, 2.0 * (2.0 * rows * C + (double)C * C), igemm_launch(d, st));
        } else {
        float *q = m->alloc_act(rows * C);
        if (int rc = linear(m, "vit_linear", NL + "q", xcur, rows, C, C, C, ACT_NONE, nullptr, 0, q, C, io)) return rc;
        if (a16) RUN(m, "vit_attention", 4.0 * rows * C * nkv, 2 * eb * rows * C, launch16_attention(q, kv, n, h * w, nkv, C, heads, att, st));
        else RUN(m, "vit_attention", 4.0 * rows * C * nkv, 8.0 * rows * C, launch_attention(q, kv, n, h * w, nkv, C, heads, att, st));
        }
        float *x1 = m->alloc_act(rows * C);   // x = x + attn(x)   (:144)
        if (int rc = linear(m, "vit_linear", NL + "fin", att, rows, C, C, C, ACT_NONE, xcur, C, x1, C, a16 ? (IO16 | RES16) : 0)) return rc;
        // --- MixFFN (:98-120)
        const bool last = l == c.vit_layers[s] - 1;
        float *xn = (last && y_out) ? y_out : m->alloc_act(rows * C);
        if (a16 && m->has(NL + "ffn.rec") && m->wld[NL + "mlp1"] == C && m->wld[NL + "mlp2"] == E && mixffn16_fits(h, w, C, E) &&
            rows * C * 2 < ((int64_t)1 << 32)) {
            // one launch, the hidden tensor never in HBM (mixffn16.hip): mlp1 -> grouped conv + GELU -> mlp2 -> + x1 -> LayerNorm
            RUN(m, "vit_mixffn_fused", 4.0 * rows * C * E + 2.0 * rows * E * 72, 2 * eb * rows * C,
                launch_mixffn16(x1, n, h, w, C, E, m->W(NL + "mlp1.w"), m->W(NL + "mlp1.b16"), m->W(NL + "ffn.rec"), m->W(NL + "mlp2.w"),
                                m->W(NL + "mlp2.b"), m->W(NL + "ln.g"), m->W(NL + "ln.beta"), xn, st));
            xcur = xn;
            continue;
        }
        float *h1 = m->alloc_act(rows * E);
        if (int rc = linear(m, "vit_linear", NL + "mlp1", x1, rows, C, C, E, ACT_NONE, nullptr, 0, h1, E, io)) return rc;
        float *h2 = m->alloc_act(rows * E);
        static const bool old_gconv = getenv("EVFLY_OLD_GCONV") != nullptr;      // A/B switch: the round-2 kernels
        if (gconv_fits(h, w, E) && !old_gconv)
            RUN(m, "vit_grouped_conv_gelu", 2.0 * rows * E * 72, 2 * eb * rows * E,
                launch_gconv_gelu(h1, n, h, w, E, m->W(NL + "dw.wp"), m->W(NL + "dw.b"), h2, a16, st));
        else if (a16) RUN(m, "vit_grouped_conv_gelu", 2.0 * rows * E * 72, 2 * eb * rows * E,
                     launch16_grouped_conv_gelu(h1, n, h, w, E, m->W(NL + "dw.w"), m->W(NL + "dw.b"), h2, st));
        else
        RUN(m, "vit_grouped_conv_gelu", 2.0 * rows * E * 72, 8.0 * rows * E,
            launch_grouped_conv_gelu(h1, n, h, w, E, m->W(NL + "dw.w"), m->W(NL + "dw.b"), h2, st));
        float *x2 = m->alloc_act(rows * C);   // x = x + ffn(x)    (:145)
        if (int rc = linear(m, "vit_linear", NL + "mlp2", h2, rows, E, E, C, ACT_NONE, x1, C, x2, C, a16 ? (IO16 | RES16) : 0)) return rc;
        if (int rc = layernorm(x2, rows, NL + "ln.g", NL + "ln.beta", xn)) return rc;
        xcur = xn;
    }
    if (c.vit_layers[s] == 0 && y_out && !m->planning)
        EVFLY_HIP(hipMemcpyAsync(y_out, xcur, (size_t)rows * C * m->esz(), hipMemcpyDeviceToDevice, st));
    if (y_ret) *y_ret = xcur;
    *Ho = h; *Wo = w;
    return 0;
}

// ============================================================================ velocity model forward (one chunk)
static int vit_chunk(evfly_model *m, const float *img, int ih, int iw, int clip2x, const float *desvel, const float *quat,
                     int S, int T, float *lstm_h, float *lstm_c, float *vel) {
    const auto &c = m->cfg;
    hipStream_t st = m->st;
    const int F = S * T;
    const bool a16 = m->act16;
    const int io = a16 ? IO16 : 0;
    const double eb = a16 ? 2.0 : 4.0;
    // refine_inputs (vitfly_models.py:27-29) + the depth hand-off clip (learner_models.py:634); fp32 in every pipeline
    const float *vin = img;
    if (ih != 60 || iw != 90 || clip2x) {
        float *t = m->alloc((int64_t)F * 60 * 90);
        RUN(m, "vit_input_resize", 0, 4.0 * F * (60.0 * 90 * 5), launch_bilinear(img, F, ih, iw, 1, 1, t, 60, 90, 1, 0, clip2x ? 1 : 0, st));
        vin = t;
    }
    m->tap("vit_in", const_cast<float *>(vin), F, 60, 90, 1);
    float *s1 = nullptr, *s2 = nullptr;
    int h1, w1, h2, w2;
    if (int rc = vit_stage(m, 0, vin, F, 60, 90, c.vit_in_channels, nullptr, &s1, &h1, &w1, false)) return rc;
    m->tap("s1", s1, F, h1, w1, c.vit_width[0], a16);
    if (int rc = vit_stage(m, 1, s1, F, h1, w1, c.vit_width[0], nullptr, &s2, &h2, &w2, a16)) return rc;
    m->tap("s2", s2, F, h2, w2, c.vit_width[1], a16);
    EVFLY_REQUIRE(2 * h2 == 16 && 2 * w2 == 24, "ViT head expects a 16x24 map (got %dx%d)", 2 * h2, 2 * w2);
    // cat[PixelShuffle(2)(s2), Upsample(s1 -> 16x24, align_corners=True)]   (vitfly_models.py:141)
    // (pixel pitch padded to a multiple of 32 channels, the padding zero: the head conv then takes the GEMM kernel's
    // vectorised K walk -- with 48 channels it fell back to the per-element gather, 0.14 ms for 2.5 GFLOP)
    const int c_ps = c.vit_width[1] / 4, ccat_real = c_ps + c.vit_width[0], ccat = round_up(ccat_real, 32);
    float *cat = m->alloc_act((int64_t)F * 16 * 24 * ccat);
    if (ccat != ccat_real && !m->planning) EVFLY_HIP(hipMemsetAsync(cat, 0, (size_t)F * 16 * 24 * ccat * m->esz(), st));
    if (a16) {
        EVFLY_REQUIRE(c_ps % 8 == 0 && c.vit_width[0] % 8 == 0, "bf16 pipeline: ViT head widths must be multiples of 8 channels");
        RUN(m, "vit_pixel_shuffle", 0, 2 * eb * F * 384 * c_ps, launch16_pixel_shuffle2(s2, F, h2, w2, c.vit_width[1], cat, ccat, st));
        RUN(m, "vit_upsample", 0, 2 * eb * F * 384 * c.vit_width[0],
            launch16_bilinear(s1, F, h1, w1, c.vit_width[0], c.vit_width[0], m->eoff(cat, c_ps), 16, 24, ccat, 1, st));
    } else {
    RUN(m, "vit_pixel_shuffle", 0, 8.0 * F * 384 * c_ps, launch_pixel_shuffle2(s2, F, h2, w2, c.vit_width[1], cat, ccat, st));
    RUN(m, "vit_upsample", 0, 8.0 * F * 384 * c.vit_width[0],
        launch_bilinear(s1, F, h1, w1, c.vit_width[0], c.vit_width[0], cat + c_ps, 16, 24, ccat, 1, 0, st));
    }
    float *flat = m->alloc_act((int64_t)F * 4608);
    if (int rc = conv(m, "vit_head_conv", "ds", cat, F, 16, 24, ccat, ccat, 12, 3, 3, 1, 1, ACT_NONE, nullptr, 0, flat, 12, nullptr, nullptr, nullptr,
                      0, 0, 0, nullptr, io)) return rc;
    m->tap("flat", flat, F, 16, 24, 12, a16);
    const int LD = 544;   // 517 padded to a multiple of 32
    float *x517 = m->alloc_act((int64_t)F * LD);
    if (int rc = linear(m, "vit_decoder_linear", "dec", flat, F, 4608, 4608, 512, ACT_NONE, nullptr, 0, x517, LD, io)) return rc;
    if (a16) RUN(m, "vit_meta_fill", 0, eb * F * 32, launch16_meta_fill(x517, F, LD, desvel, quat, st));
    else RUN(m, "vit_meta_fill", 0, 4.0 * F * 32, launch_meta_fill(x517, F, LD, desvel, quat, st));
    m->tap("x517", x517, F, LD, 1, 1, a16);
    if (c.head == EVFLY_HEAD_LSTMNETVIT) {
        float *xg0 = m->alloc((int64_t)F * 512);        // fp32: the recurrence kernel's input
        {
            ConvDesc d; d.x = x517; d.ldx = LD; d.NI = F; d.C = LD; d.w = m->W("lstm.ih0.w"); d.ldw = a16 ? round_up(LD, 64) : LD; d.bias = m->W("lstm.b0");
            conv_finish(d); d.Nc = 512; d.y = xg0; d.ldy = 512; d.dtype = c.compute_dtype; d.in_bf16 = a16;
            RUN(m, "lstm_x_gemm", igemm_flops(d), eb * (F * (double)LD + 512.0 * LD) + 4.0 * F * 512.0, igemm_launch(d, st));
        }
        LstmWeights w{};
        for (int l = 0; l < 3; ++l) {
            w.whh_t[l] = m->W("lstm.hh" + std::to_string(l));
            w.wih_t[l] = l ? m->W("lstm.ih" + std::to_string(l)) : nullptr;
            w.bias[l] = m->W("lstm.b" + std::to_string(l));
        }
        w.fc_w = m->W("fc2.w"); w.fc_b = m->W("fc2.b");
        RUN(m, "lstm_recurrence", 2.0 * F * 5 * 128 * 512, 4.0 * F * 5 * 128 * 512, launch_lstm(xg0, S, T, w, lstm_h, lstm_c, vel, st));
    } else {
        float *f1 = m->alloc_act((int64_t)F * 256);
        if (int rc = linear(m, "vit_fc", "fc1", x517, F, LD, LD, 256, ACT_LEAKY, nullptr, 0, f1, 256, io)) return rc;
        if (int rc = linear(m, "vit_fc", "fc2", f1, F, 256, 256, 3, ACT_NONE, nullptr, 0, vel, 3, a16 ? IN16 : 0)) return rc;
    }
    return 0;
}

// ============================================================================ chunked drivers
namespace {

template <typename Fn>
int with_arena(evfly_model *m, Fn &&body) {
    // pass 1: plan the arena, pass 2: run
    m->planning = true; m->arena_off = 0; m->plan_peak = 0;
    if (int rc = body()) { m->planning = false; return rc; }
    m->planning = false;
    if (m->plan_peak > m->arena_cap) {
        if (m->arena) EVFLY_HIP(hipFree(m->arena));
        m->arena = nullptr; m->arena_cap = 0;
        EVFLY_HIP(hipMalloc(reinterpret_cast<void **>(&m->arena), m->plan_peak));
        m->arena_cap = m->plan_peak;
        ++m->arena_generation;
    }
    m->arena_off = 0;
    return body();
}

int check_model(evfly_model *m, void *stream) {
    EVFLY_REQUIRE(m && m->finalized, "model handle is null or not finalized");
    int dev = 0;
    EVFLY_HIP(hipGetDevice(&dev));
    EVFLY_REQUIRE(dev == m->device, "model handle lives on device %d, current device is %d", m->device, dev);
    m->st = as_stream(stream);
    return 0;
}

// frames per chunk of the depth model: 640 (the arena is ~32 GB in fp32, of 288). Larger launches lose less to partly empty last
// rounds: C4 shard 112.4 -> 111.4 ms, C3 75.8 -> 74.1 ms against 320; 1280 gains another 0.7 % at C4 but pushes the bf16 pipeline's
// full-resolution maps past the 4 GB its kernels address with 32-bit offsets. EVFLY_CHUNK_FRAMES overrides (A/B runs).
static const int kChunkFrames = getenv("EVFLY_CHUNK_FRAMES") ? std::min(640, std::max(16, atoi(getenv("EVFLY_CHUNK_FRAMES")))) : 640;

}  // namespace

extern "C" int evfly_unet_forward(evfly_model *m, const float *frames, int n_streams, int T, float *h_state, float *c_state,
                                  float *depth_out, float *upconv_out, float *yvel_out, float *velpred_h, float *velpred_c,
                                  void *stream) {
    if (int rc = check_model(m, stream)) return rc;
    EVFLY_REQUIRE(m->cfg.has_unet, "handle has no U-Net");
    EVFLY_REQUIRE(frames && n_streams > 0 && T > 0, "unet_forward: empty batch");
    EVFLY_REQUIRE((h_state == nullptr) == (c_state == nullptr), "unet_forward: h_state and c_state go together");
    EVFLY_REQUIRE(m->cfg.velpred == 0 || yvel_out, "unet_forward: the handle has a velpred head, yvel_out is required");
    EVFLY_REQUIRE((velpred_h == nullptr) == (velpred_c == nullptr), "unet_forward: velpred_h and velpred_c go together");
    const int per = std::max(1, kChunkFrames / T);
    const int64_t fr = (int64_t)m->cfg.input_h * m->cfg.input_w;
    m->clstm_rows_first = (int64_t)std::min(per, n_streams) * 104;
    for (int s0 = 0; s0 < n_streams; s0 += per) {
        const int S = std::min(per, n_streams - s0);
        auto body = [&]() {
            return unet_chunk(m, frames + (int64_t)s0 * T * fr, S, T, h_state ? h_state + (int64_t)s0 * 104 * 512 : nullptr,
                              c_state ? c_state + (int64_t)s0 * 104 * 512 : nullptr,
                              depth_out ? depth_out + (int64_t)s0 * T * fr : nullptr,
                              upconv_out ? upconv_out + (int64_t)s0 * T * 68 * 148 : nullptr, nullptr,
                              m->cfg.velpred > 0 ? yvel_out + (int64_t)s0 * T * 3 : nullptr,
                              velpred_h ? velpred_h + (int64_t)s0 * m->cfg.velpred_lstm_layers * m->vp_hidden : nullptr,
                              velpred_c ? velpred_c + (int64_t)s0 * m->cfg.velpred_lstm_layers * m->vp_hidden : nullptr);
        };
        if (int rc = with_arena(m, body)) return rc;
    }
    return 0;
}

extern "C" int evfly_vit_forward(evfly_model *m, const float *img, int img_h, int img_w, int clip2x, const float *desvel,
                                 const float *quat, int n_streams, int T, float *lstm_h, float *lstm_c, float *vel_out,
                                 void *stream) {
    if (int rc = check_model(m, stream)) return rc;
    EVFLY_REQUIRE(m->cfg.head != EVFLY_HEAD_NONE, "handle has no velocity head");
    EVFLY_REQUIRE(img && desvel && vel_out && n_streams > 0 && T > 0, "vit_forward: null argument");
    const int per = std::max(1, 4 * kChunkFrames / T);
    for (int s0 = 0; s0 < n_streams; s0 += per) {
        const int S = std::min(per, n_streams - s0);
        const int64_t f0 = (int64_t)s0 * T;
        auto body = [&]() {
            return vit_chunk(m, img + f0 * img_h * img_w, img_h, img_w, clip2x, desvel + f0, quat ? quat + f0 * 4 : nullptr, S, T,
                             lstm_h ? lstm_h + (int64_t)s0 * 384 : nullptr, lstm_c ? lstm_c + (int64_t)s0 * 384 : nullptr,
                             vel_out + f0 * 3);
        };
        if (int rc = with_arena(m, body)) return rc;
    }
    return 0;
}

extern "C" int evfly_vit_stage_forward(evfly_model *m, int stage, const float *x, int n, int h, int w, float *y, void *stream) {
    if (int rc = check_model(m, stream)) return rc;
    EVFLY_REQUIRE(stage == 0 || stage == 1, "stage must be 0 or 1");
    EVFLY_REQUIRE(x && y && n > 0, "vit_stage_forward: null argument");
    const int cin = stage == 0 ? m->cfg.vit_in_channels : m->cfg.vit_width[0];
    int ho, wo;
    if (!m->act16) {
        auto body = [&]() { return vit_stage(m, stage, x, n, h, w, cin, y, nullptr, &ho, &wo); };
        return with_arena(m, body);
    }
    // bf16 pipeline behind the fp32 ABI: the input is rounded to bf16 when the stage's first conv takes a bf16 operand
    // (C_in % 32 == 0), the stage output is widened back to fp32
    auto body = [&]() {
        const bool x16 = cin % 32 == 0;
        const float *xin = x;
        if (x16) {
            float *t = m->alloc_act((int64_t)n * h * w * cin);
            if (!m->planning) { if (int rc = launch_f32_to_bf16(x, (int64_t)n * h * w * cin, t, m->st)) return rc; }
            xin = t;
        }
        float *yr = nullptr;
        if (int rc = vit_stage(m, stage, xin, n, h, w, cin, nullptr, &yr, &ho, &wo, x16)) return rc;
        if (!m->planning) return launch_bf16_to_f32(yr, (int64_t)n * ho * wo * m->cfg.vit_width[stage], y, m->st);
        return 0;
    };
    return with_arena(m, body);
}

// ============================================================================ one half of a Mix-Transformer block on its own
// EfficientSelfAttention.forward (ViTsubmodules.py:54-83: reduction conv + LayerNorm, key / value and query projections, softmax attention,
// finalLayer -- WITHOUT the block's residual) or MixFFN.forward (:98-120: mlp1, grouped 3x3 conv, erf-GELU, mlp2 -- without residual and
// LayerNorm) of layer `l` of stage `s`, on tokens (n, h * w, C). Exact-fp32 kernels, the launches vit_stage issues for the same step.
static int vit_block_part(evfly_model *m, int s, int l, int part, const float *x, int n, int h, int w, float *y) {
    const auto &c = m->cfg;
    hipStream_t st = m->st;
    const int C = c.vit_width[s], heads = c.vit_heads[s], R = c.vit_reduction[s], E = C * c.vit_expansion;
    const int64_t rows = (int64_t)n * h * w;
    const std::string NL = "s" + std::to_string(s) + "." + std::to_string(l) + ".";
    if (part == EVFLY_VIT_PART_ATTENTION) {
        const int rh = (h - R) / R + 1, rw = (w - R) / R + 1, nkv = rh * rw;
        float *red = m->alloc_act((int64_t)n * nkv * C);
        if (int rc = conv(m, "vit_kv_reduce_conv", NL + "red", x, n, h, w, C, C, C, R, R, R, 0, ACT_NONE, nullptr, 0, red, C, nullptr, nullptr,
                          nullptr, 0, 0, 0, nullptr, 0)) return rc;
        float *redn = m->alloc_act((int64_t)n * nkv * C);
        RUN(m, "vit_layernorm", 0, 8.0 * n * nkv * C, launch_layernorm(red, nullptr, (int64_t)n * nkv, C, m->W(NL + "ln1.g"), m->W(NL + "ln1.beta"), redn, st));
        float *kv = m->alloc_act((int64_t)n * nkv * 2 * C);
        if (int rc = linear(m, "vit_linear", NL + "kv", redn, (int64_t)n * nkv, C, C, 2 * C, ACT_NONE, nullptr, 0, kv, 2 * C, 0)) return rc;
        float *q = m->alloc_act(rows * C), *att = m->alloc_act(rows * C);
        if (int rc = linear(m, "vit_linear", NL + "q", x, rows, C, C, C, ACT_NONE, nullptr, 0, q, C, 0)) return rc;
        RUN(m, "vit_attention", 4.0 * rows * C * nkv, 8.0 * rows * C, launch_attention(q, kv, n, h * w, nkv, C, heads, att, st));
        return linear(m, "vit_linear", NL + "fin", att, rows, C, C, C, ACT_NONE, nullptr, 0, y, C, 0);
    }
    float *h1 = m->alloc_act(rows * E), *h2 = m->alloc_act(rows * E);
    if (int rc = linear(m, "vit_linear", NL + "mlp1", x, rows, C, C, E, ACT_NONE, nullptr, 0, h1, E, 0)) return rc;
    if (gconv_fits(h, w, E))
        RUN(m, "vit_grouped_conv_gelu", 2.0 * rows * E * 72, 8.0 * rows * E, launch_gconv_gelu(h1, n, h, w, E, m->W(NL + "dw.wp"), m->W(NL + "dw.b"), h2, false, st));
    else
        RUN(m, "vit_grouped_conv_gelu", 2.0 * rows * E * 72, 8.0 * rows * E, launch_grouped_conv_gelu(h1, n, h, w, E, m->W(NL + "dw.w"), m->W(NL + "dw.b"), h2, st));
    return linear(m, "vit_linear", NL + "mlp2", h2, rows, E, E, C, ACT_NONE, nullptr, 0, y, C, 0);
}

extern "C" int evfly_vit_block_forward(evfly_model *m, int stage, int layer, int part, const float *x, int n, int h, int w, float *y, void *stream) {
    if (int rc = check_model(m, stream)) return rc;
    EVFLY_REQUIRE(stage == 0 || stage == 1, "vit_block_forward: stage must be 0 or 1");
    EVFLY_REQUIRE(layer >= 0 && layer < m->cfg.vit_layers[stage], "vit_block_forward: layer %d of %d", layer, m->cfg.vit_layers[stage]);
    EVFLY_REQUIRE(part == EVFLY_VIT_PART_ATTENTION || part == EVFLY_VIT_PART_MIXFFN, "vit_block_forward: part must be EVFLY_VIT_PART_ATTENTION or _MIXFFN");
    EVFLY_REQUIRE(x && y && n > 0 && h > 0 && w > 0, "vit_block_forward: null or empty argument");
    EVFLY_REQUIRE(!m->act16, "vit_block_forward: the stand-alone block halves run in the exact-fp32 pipeline (compute_dtype f32)");
    EVFLY_REQUIRE(part != EVFLY_VIT_PART_ATTENTION || (h >= m->cfg.vit_reduction[stage] && w >= m->cfg.vit_reduction[stage]),
                  "vit_block_forward: a %d x %d token grid is smaller than the reduction ratio %d", h, w, m->cfg.vit_reduction[stage]);
    auto body = [&]() { return vit_block_part(m, stage, layer, part, x, n, h, w, y); };
    return with_arena(m, body);
}

extern "C" int evfly_e2v_forward(evfly_model *m, const float *frames, const float *desvel, int n_streams, int T,
                                 float *h_state, float *c_state, float *lstm_h, float *lstm_c, float *depth_out,
                                 float *upconv_out, float *vel_out, void *stream) {
    if (int rc = check_model(m, stream)) return rc;
    EVFLY_REQUIRE(m->cfg.has_unet && m->cfg.head != EVFLY_HEAD_NONE, "e2v_forward needs a composite handle");
    EVFLY_REQUIRE(!(m->cfg.is_deployment && !(m->cfg.velpred == 1 || m->cfg.velpred == 11)),
                  "e2v_forward: is_deployment=True skips the decoder, there is no depth for the velocity model "
                  "(the reference fails on None * 2 at learner_models.py:634)");
    EVFLY_REQUIRE(frames && desvel && vel_out && n_streams > 0 && T > 0, "e2v_forward: null argument");
    EVFLY_REQUIRE((h_state == nullptr) == (c_state == nullptr), "e2v_forward: h_state and c_state go together");
    const int per = std::max(1, kChunkFrames / T);
    const int H = m->cfg.input_h, Wd = m->cfg.input_w;
    const int64_t fr = (int64_t)H * Wd;
    m->clstm_rows_first = (int64_t)std::min(per, n_streams) * 104;
    if (depth_out && n_streams > per) {
        // More than one chunk and the caller takes the depth maps: the depth model runs over all its 320-frame chunks first (depth
        // lands in the caller's buffer), then the velocity model reads it back in chunks four times as large, like evfly_vit_forward
        // does -- its launches are small, at 320 frames most of them leave the chip half empty (C4 shard, 1280 frames: 115.5 ->
        // 113 ms; the per-frame arithmetic is the same)
        for (int s0 = 0; s0 < n_streams; s0 += per) {
            const int S = std::min(per, n_streams - s0);
            const int64_t f0 = (int64_t)s0 * T;
            auto body = [&]() {
                float *depth = nullptr;
                return unet_chunk(m, frames + f0 * fr, S, T, h_state ? h_state + (int64_t)s0 * 104 * 512 : nullptr,
                                  c_state ? c_state + (int64_t)s0 * 104 * 512 : nullptr, depth_out + f0 * fr,
                                  upconv_out ? upconv_out + f0 * 68 * 148 : nullptr, &depth);
            };
            if (int rc = with_arena(m, body)) return rc;
        }
        const int perv = std::max(1, 4 * kChunkFrames / T);
        for (int s0 = 0; s0 < n_streams; s0 += perv) {
            const int S = std::min(perv, n_streams - s0);
            const int64_t f0 = (int64_t)s0 * T;
            auto body = [&]() {
                return vit_chunk(m, depth_out + f0 * fr, H, Wd, 1, desvel + f0, nullptr, S, T, lstm_h ? lstm_h + (int64_t)s0 * 384 : nullptr,
                                 lstm_c ? lstm_c + (int64_t)s0 * 384 : nullptr, vel_out + f0 * 3);
            };
            if (int rc = with_arena(m, body)) return rc;
        }
        return 0;
    }
    for (int s0 = 0; s0 < n_streams; s0 += per) {
        const int S = std::min(per, n_streams - s0);
        const int64_t f0 = (int64_t)s0 * T;
        auto body = [&]() {
            float *depth = nullptr;
            if (int rc = unet_chunk(m, frames + f0 * fr, S, T, h_state ? h_state + (int64_t)s0 * 104 * 512 : nullptr,
                                    c_state ? c_state + (int64_t)s0 * 104 * 512 : nullptr,
                                    depth_out ? depth_out + f0 * fr : nullptr, upconv_out ? upconv_out + f0 * 68 * 148 : nullptr,
                                    &depth))
                return rc;
            // x_depth_input = clip(x_depth * 2, 0, 1) (learner_models.py:634), fused into the 60x90 resize
            return vit_chunk(m, depth, H, Wd, 1, desvel + f0, nullptr, S, T, lstm_h ? lstm_h + (int64_t)s0 * 384 : nullptr,
                             lstm_c ? lstm_c + (int64_t)s0 * 384 : nullptr, vel_out + f0 * 3);
        };
        if (int rc = with_arena(m, body)) return rc;
    }
    return 0;
}

// ============================================================================ lifecycle
extern "C" int evfly_model_create(const evfly_model_config *cfg, evfly_model **out) {
    EVFLY_REQUIRE(cfg && out, "model_create: null argument");
    if (cfg->has_unet) {
        EVFLY_REQUIRE(cfg->input_h == 260 && cfg->input_w == 346,
                      "OrigUNet geometry is hard-wired for 260x346 (learner_models.py:373-419)");
        EVFLY_REQUIRE(cfg->form_bev >= 0 && cfg->form_bev <= 2, "form_BEV should be 0/1/2, but is %d", cfg->form_bev);
        EVFLY_REQUIRE(cfg->skip_type >= 0 && cfg->skip_type <= 2, "skip_type should be crop/interp/none");
        EVFLY_REQUIRE(cfg->num_out_channels == 1, "num_out_channels != 1 is not built");
        EVFLY_REQUIRE(cfg->num_recurrent_unet == 0 || cfg->num_recurrent_unet == 1, "only 0 or 1 ConvLSTM layers are built");
    }
    if (cfg->velpred != 0) {
        EVFLY_REQUIRE(cfg->has_unet, "velpred needs the U-Net");
        EVFLY_REQUIRE(cfg->velpred == 1 || cfg->velpred == 11 || cfg->velpred == 2, "velpred should be 0/1/11/2, but is %d", cfg->velpred);
        EVFLY_REQUIRE(cfg->enc_num_layers >= 0 && cfg->enc_num_layers <= EVFLY_MAX_ENC_LAYERS, "enc_num_layers out of range");
        EVFLY_REQUIRE(cfg->fc_num_layers >= 1 && cfg->fc_num_layers <= EVFLY_MAX_FC_LAYERS, "fc_num_layers out of range");
        EVFLY_REQUIRE(cfg->velpred_lstm_layers >= 0 && cfg->velpred_lstm_layers <= 4, "velpred_lstm_layers out of range");
        EVFLY_REQUIRE(cfg->fc_size[cfg->fc_num_layers - 1] == 1, "velpred_head is built with num_out=1 (learner_models.py:462): "
                      "the last fc layer size must be 1");
        EVFLY_REQUIRE(cfg->enc_pool_type >= EVFLY_POOL_NONE && cfg->enc_pool_type <= EVFLY_POOL_AVG, "bad enc_pool_type");
        for (int i = 0; i < cfg->enc_num_layers; ++i)
            EVFLY_REQUIRE(cfg->enc_act[i] >= EVFLY_ACT_NONE && cfg->enc_act[i] <= EVFLY_ACT_SIGMOID && cfg->enc_out_channels[i] > 0,
                          "bad velpred encoder layer %d", i);
        for (int i = 0; i < cfg->fc_num_layers; ++i)
            EVFLY_REQUIRE(cfg->fc_act[i] >= EVFLY_ACT_RELU && cfg->fc_act[i] <= EVFLY_ACT_SIGMOID && cfg->fc_size[i] > 0,
                          "bad velpred fc layer %d", i);
        VpGeom g;
        if (int rc = velpred_geometry(*cfg, g)) return rc;
    }
    EVFLY_REQUIRE(cfg->compute_dtype >= EVFLY_DTYPE_F32 && cfg->compute_dtype <= EVFLY_DTYPE_BF16X3, "bad compute_dtype");
    auto *m = new evfly_model();
    m->cfg = *cfg;
    m->full_encoder_outputs = getenv("EVFLY_FULL_ENCODER_OUTPUTS") != nullptr;
    m->act16 = cfg->compute_dtype == EVFLY_DTYPE_BF16 && getenv("EVFLY_BF16_LEGACY") == nullptr;
    if (hipGetDevice(&m->device) != hipSuccess) { delete m; return fail(-2, "hipGetDevice failed"); }
    *out = m;
    return 0;
}

extern "C" int evfly_model_load_tensor(evfly_model *m, const char *key, const float *data_host, const int64_t *shape, int ndim) {
    EVFLY_REQUIRE(m && key && data_host && ndim >= 0 && ndim <= 8, "load_tensor: bad argument");
    EVFLY_REQUIRE(!m->finalized, "load_tensor after finalize");
    HostTensor t;
    t.shape.assign(shape, shape + ndim);
    t.v.assign(data_host, data_host + t.numel());
    m->host[key] = std::move(t);
    return 0;
}

extern "C" int evfly_model_finalize(evfly_model *m) {
    EVFLY_REQUIRE(m && !m->finalized, "finalize: bad handle");
    if (m->cfg.has_unet)
        if (int rc = pack_unet(m)) return rc;
    if (m->cfg.velpred > 0)
        if (int rc = pack_velpred(m)) return rc;
    bool any_vit = false;
    for (auto &kv : m->host) any_vit |= kv.first.find("encoder_blocks.") != std::string::npos;
    if (m->cfg.head != EVFLY_HEAD_NONE || any_vit)
        if (int rc = pack_vit(m)) return rc;
    EVFLY_REQUIRE(!m->wstage.empty(), "finalize: no tensors were loaded");
    EVFLY_HIP(hipMalloc(reinterpret_cast<void **>(&m->wdev), m->wstage.size() * 4));
    EVFLY_HIP(hipMemcpy(m->wdev, m->wstage.data(), m->wstage.size() * 4, hipMemcpyHostToDevice));
    m->wstage.clear(); m->wstage.shrink_to_fit();
    m->host.clear();
    m->finalized = true;
    return 0;
}

extern "C" void evfly_model_destroy(evfly_model *m) { delete m; }
extern "C" int evfly_model_arena_generation(evfly_model *m) { return m ? m->arena_generation : -1; }

extern "C" int64_t evfly_model_tap(evfly_model *m, const char *name, float *dst_host, int64_t max_elems, int64_t *shape_out,
                                   void *stream) {
    EVFLY_REQUIRE(m && name && dst_host, "tap: null argument");
    auto it = m->taps.find(name);
    EVFLY_REQUIRE(it != m->taps.end(), "tap '%s' was not produced by the last forward", name);
    EVFLY_REQUIRE(!(m->dot_used && name[0] == 'd' && name[1] == '4' && name[2] == 0),
                  "tap 'd4': partial -- the last forward fused unet_out into d42's kernel and did not write the 32-channel map "
                  "(EVFLY_FULL_ENCODER_OUTPUTS=1 or EVFLY_NO_OUT_FUSION=1 keep it)");
    EVFLY_REQUIRE(!(m->bands_used && name[0] == 'e' && name[1] >= '1' && name[1] <= '4' && name[2] == 0),
                  "tap '%s' is partial: the last forward stored only the block-border pixels of the full-resolution encoder maps "
                  "(fused 'interp' skip). Set EVFLY_FULL_ENCODER_OUTPUTS=1 before the handle is created to keep them complete", name);
    int64_t n = 1;
    for (int i = 0; i < 4; ++i) { n *= it->second.shape[i]; if (shape_out) shape_out[i] = it->second.shape[i]; }
    EVFLY_REQUIRE(n <= max_elems, "tap '%s' has %lld elements, buffer holds %lld", name, (long long)n, (long long)max_elems);
    EVFLY_HIP(hipStreamSynchronize(as_stream(stream)));
    if (it->second.bf16) {      // bf16 pipeline: widen on the host
        std::vector<bf16_t> tmp((size_t)n);
        EVFLY_HIP(hipMemcpy(tmp.data(), it->second.ptr, (size_t)n * 2, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; ++i) dst_host[i] = host_bf2f(tmp[(size_t)i]);
        return n;
    }
    EVFLY_HIP(hipMemcpy(dst_host, it->second.ptr, (size_t)n * 4, hipMemcpyDeviceToHost));
    return n;
}

extern "C" int evfly_model_set_profiling(evfly_model *m, int enable) {
    EVFLY_REQUIRE(m, "null handle");
    m->profiling = enable != 0;
    return 0;
}
extern "C" int evfly_model_set_profile_filter(evfly_model *m, const char *prefix) {
    EVFLY_REQUIRE(m, "null handle");
    m->prof_filter = prefix ? prefix : "";
    return 0;
}
extern "C" int evfly_model_profile_count(evfly_model *m) {
    if (!m) return 0;
    if (m->prof_collect()) return 0;
    return (int)m->agg.size();
}
extern "C" int evfly_model_profile_get(evfly_model *m, int i, char *name_out, int name_cap, double *ms_out, double *flops_out,
                                       double *bytes_out, int *launches_out) {
    EVFLY_REQUIRE(m && i >= 0 && i < (int)m->agg.size(), "profile_get: index out of range");
    const ProfAgg &a = m->agg[i];
    if (name_out && name_cap > 0) { std::strncpy(name_out, a.name.c_str(), name_cap - 1); name_out[name_cap - 1] = 0; }
    if (ms_out) *ms_out = a.ms;
    if (flops_out) *flops_out = a.flops;
    if (bytes_out) *bytes_out = a.bytes;
    if (launches_out) *launches_out = a.launches;
    return 0;
}
extern "C" int evfly_model_profile_exec_flops(evfly_model *m, int i, double *exec_flops_out) {
    EVFLY_REQUIRE(m && exec_flops_out && i >= 0 && i < (int)m->agg.size(), "profile_exec_flops: index out of range");
    *exec_flops_out = m->agg[i].exec;
    return 0;
}

extern "C" int evfly_model_profile_useful_flops(evfly_model *m, int i, double *useful_flops_out) {
    EVFLY_REQUIRE(m && useful_flops_out && i >= 0 && i < (int)m->agg.size(), "profile_useful_flops: index out of range");
    *useful_flops_out = m->agg[i].useful;
    return 0;
}

extern "C" int evfly_model_profile_reset(evfly_model *m) {
    EVFLY_REQUIRE(m, "null handle");
    if (int rc = m->prof_collect()) return rc;
    m->agg.clear();
    return 0;
}

// ============================================================================ single-operator entry point
extern "C" int evfly_op_conv2d_nhwc(const float *x, int n, int h, int w, int cin, const float *w_packed, const float *bias,
                                    int cout, int kh, int kw, int stride, int pad, int act, const float *res, float *y,
                                    int dtype, void *stream) {
    EVFLY_REQUIRE(x && w_packed && y, "op_conv2d: null argument");
    ConvDesc d;
    d.x = x; d.ldx = cin; d.NI = n; d.H = h; d.W = w; d.C = cin; d.w = w_packed; d.bias = bias;
    d.KH = kh; d.KW = kw; d.stride = stride; d.pad = pad;
    conv_finish(d);
    // the kernel walks K in the order of igemm.h conv_k_index, zero padded to a multiple of 32: lay a copy out like that
    const int ld = round_up(d.K, 32);
    {
        void *scr = nullptr;
        if (int rc = scratch_get((size_t)cout * ld * 4, &scr, as_stream(stream))) return rc;
        if (int rc = launch_repack_w(w_packed, cout, kh * kw, cin, ld, static_cast<float *>(scr), as_stream(stream))) return rc;
        d.w = static_cast<const float *>(scr);
    }
    d.ldw = ld; d.Nc = cout; d.res = res; d.ldres = cout; d.act = act; d.y = y; d.ldy = cout; d.dtype = dtype;
#ifdef EVFLY_WITH_WINO4      // developer library only (tools/scripts/build_w4.sh): the F(4x4,3x3) prototype of tools/proto/wino4.hip, 0.51x the shipped kernel
    static const bool use_w4 = getenv("EVFLY_WINO4") && atoi(getenv("EVFLY_WINO4")) == 1;
    if (use_w4 && wino4_applicable(d)) {
        void *u = nullptr;
        if (int rc = scratch_get(wino4_u_floats(cout, cin) * 4, &u, as_stream(stream), 2)) return rc;
        if (int rc = wino4_pack_device(w_packed, cout, cin, (int64_t)9 * cin, 1, cin, static_cast<float *>(u), as_stream(stream))) return rc;
        return wino4_launch(d, static_cast<const float *>(u), as_stream(stream));
    }
#endif
    if (wino_applicable(d) && !(getenv("EVFLY_WINO_OP") && atoi(getenv("EVFLY_WINO_OP")) == 0)) {
        void *u = nullptr;
        if (int rc = scratch_get(wino_u_floats(cout, cin) * 4, &u, as_stream(stream), 2)) return rc;
        if (int rc = wino_pack_device(w_packed, cout, cin, (int64_t)9 * cin, 1, cin, static_cast<float *>(u), as_stream(stream))) return rc;
        return wino_launch(d, static_cast<const float *>(u), as_stream(stream));
    }
    return igemm_launch(d, as_stream(stream));
}

// bf16-pipeline twin: x, res, y are bf16 NHWC (raw bits), cin % 32 == 0; the fp32 weights are rounded to bf16 and laid out
// for the kernel on the device (scratch slot 2), exactly what evfly_model_finalize does on the host.
extern "C" int evfly_op_conv2d_nhwc_bf16(const uint16_t *x, int n, int h, int w, int cin, const float *w_packed, const float *bias,
                                         int cout, int kh, int kw, int stride, int pad, int act, const uint16_t *res, uint16_t *y,
                                         void *stream) {
    EVFLY_REQUIRE(x && w_packed && y, "op_conv2d_bf16: null argument");
    EVFLY_REQUIRE(cin % 32 == 0, "op_conv2d_bf16: cin must be a multiple of 32");
    ConvDesc d;
    d.x = reinterpret_cast<const float *>(x); d.ldx = cin; d.NI = n; d.H = h; d.W = w; d.C = cin; d.bias = bias;
    d.KH = kh; d.KW = kw; d.stride = stride; d.pad = pad;
    conv_finish(d);
    const int ld = round_up(d.K, 64);
    void *scr = nullptr;
    if (int rc = scratch_get((size_t)cout * ld * 2, &scr, as_stream(stream), 2)) return rc;
    if (int rc = launch16_repack_w(w_packed, cout, kh * kw, cin, ld, scr, as_stream(stream))) return rc;
    d.w = static_cast<const float *>(scr); d.ldw = ld;
    d.Nc = cout; d.res = reinterpret_cast<const float *>(res); d.ldres = cout; d.act = act;
    d.y = reinterpret_cast<float *>(y); d.ldy = cout; d.dtype = EVFLY_DTYPE_BF16;
    d.in_bf16 = d.out_bf16 = 1; d.res_bf16 = res != nullptr;
    if (conv16_applicable(d)) {     // the shallow 3x3 layers of the pipeline run the direct-convolution kernel
        void *wdp = nullptr;
        if (int rc = scratch_get(conv16_weight_elems(cout, cin, conv16_ntb(cout)) * 2, &wdp, as_stream(stream), 1)) return rc;
        if (int rc = conv16_pack_device(w_packed, cout, cin, wdp, as_stream(stream))) return rc;
        return conv16_launch(d, wdp, nullptr, as_stream(stream));
    }
    if (conv16w_applicable(d)) {      // the deep 3x3 layers
        void *wpp = nullptr;
        if (int rc = scratch_get(conv16p_weight_elems(cout, cin) * 2, &wpp, as_stream(stream), 1)) return rc;
        if (int rc = conv16p_pack_device(scr, cout, cin, ld, wpp, as_stream(stream))) return rc;
        d.w_patch = wpp;
        return conv16w_launch(d, as_stream(stream));
    }
    return igemm_launch(d, as_stream(stream));
}

// ---------------------------------------------------------------------------------------- small stateless operators
// (the kernels the model handles launch, exposed for the stand-alone forwards of the reference's helper modules)
extern "C" int evfly_op_pool2d_nhwc(const float *x, int n, int h, int w, int c, int k, int stride, int type, int negate, float *y, void *stream) {
    EVFLY_REQUIRE(x && y && n > 0 && c > 0, "op_pool2d: null or empty argument");
    EVFLY_REQUIRE(type == EVFLY_POOL_MAX || type == EVFLY_POOL_AVG, "op_pool2d: type must be EVFLY_POOL_MAX or EVFLY_POOL_AVG");
    return launch_pool2d(x, n, h, w, c, k, stride, type, negate, y, as_stream(stream));
}

// MixFFN middle (ViTsubmodules.py:92-116: nn.Conv2d(E, E, 3, padding=1, groups=E/8) + nn.GELU()) as a stateless operator: the weight is
// repacked per group on the device into the scratch buffer, then the same kernels the model handles launch
extern "C" int evfly_op_grouped_conv_gelu(const void *x, int n, int h, int w, int ce, const float *weight, const float *bias, void *y,
                                          int bf16, void *stream) {
    EVFLY_REQUIRE(x && weight && bias && y && n > 0, "op_grouped_conv_gelu: null or empty argument");
    EVFLY_REQUIRE(ce % 8 == 0, "op_grouped_conv_gelu: channels must be a multiple of 8 (groups of 8)");
    if (!gconv_fits(h, w, ce)) {
        if (bf16) return launch16_grouped_conv_gelu(x, n, h, w, ce, weight, bias, y, as_stream(stream));
        return launch_grouped_conv_gelu(static_cast<const float *>(x), n, h, w, ce, weight, bias, static_cast<float *>(y), as_stream(stream));
    }
    void *wp = nullptr;
    if (int rc = scratch_get((size_t)ce * 72 * 4, &wp, as_stream(stream), 1)) return rc;
    if (int rc = gconv_pack_device(weight, ce, static_cast<float *>(wp), as_stream(stream))) return rc;
    return launch_gconv_gelu(x, n, h, w, ce, static_cast<const float *>(wp), bias, y, bf16 != 0, as_stream(stream));
}

// Tail of a Mix-Transformer block (ViTsubmodules.py:85-120,145-146) in the bf16 pipeline as a stateless operator:
// y = LayerNorm(x + MixFFN(x)), one launch (mixffn16.hip). The fp32 state-dict tensors are rounded and packed per call
// (host round trip, synchronous: a test / tooling entry -- model handles pack once at finalize).
extern "C" int evfly_op_mixffn_block_bf16(const void *x, int n, int h, int w, int c, int e, const float *w1, const float *b1, const float *dw_w,
                                          const float *dw_b, const float *w2, const float *b2, const float *ln_g, const float *ln_b, void *y,
                                          void *stream) {
    EVFLY_REQUIRE(x && w1 && b1 && dw_w && dw_b && w2 && b2 && ln_g && ln_b && y && n > 0, "op_mixffn_block_bf16: null or empty argument");
    EVFLY_REQUIRE(mixffn16_fits(h, w, c, e), "op_mixffn_block_bf16: %dx%d tokens x %d channels (hidden %d) have no fused kernel", h, w, c, e);
    hipStream_t st = as_stream(stream);
    EVFLY_HIP(hipStreamSynchronize(st));
    std::vector<float> hw1((size_t)e * c), hb1(e), hdw((size_t)e * 72), hdb(e), hw2((size_t)c * e), wp((size_t)e * 72);
    EVFLY_HIP(hipMemcpy(hw1.data(), w1, hw1.size() * 4, hipMemcpyDeviceToHost));
    EVFLY_HIP(hipMemcpy(hb1.data(), b1, hb1.size() * 4, hipMemcpyDeviceToHost));
    EVFLY_HIP(hipMemcpy(hdw.data(), dw_w, hdw.size() * 4, hipMemcpyDeviceToHost));
    EVFLY_HIP(hipMemcpy(hdb.data(), dw_b, hdb.size() * 4, hipMemcpyDeviceToHost));
    EVFLY_HIP(hipMemcpy(hw2.data(), w2, hw2.size() * 4, hipMemcpyDeviceToHost));
    gconv_pack_host(hdw.data(), e, wp.data());
    const size_t nrec = mixffn16_rec_bytes(e), o1 = round_up((int64_t)nrec, 256), o2 = o1 + (size_t)e * 16, o3 = o2 + (size_t)e * c * 2,
                 total = o3 + (size_t)c * e * 2;
    std::vector<unsigned char> host(total, 0);
    mixffn16_pack_host(wp.data(), hdb.data(), hb1.data(), e, host.data(), reinterpret_cast<bf16_t *>(host.data() + o1));
    bf16_t *pw1 = reinterpret_cast<bf16_t *>(host.data() + o2), *pw2 = reinterpret_cast<bf16_t *>(host.data() + o3);
    for (size_t i = 0; i < hw1.size(); ++i) pw1[i] = host_f2bf(hw1[i]);
    for (size_t i = 0; i < hw2.size(); ++i) pw2[i] = host_f2bf(hw2[i]);
    unsigned char *dev = nullptr;
    EVFLY_HIP(hipMalloc(reinterpret_cast<void **>(&dev), total));
    int rc = 0;
    if (hipMemcpy(dev, host.data(), total, hipMemcpyHostToDevice) != hipSuccess) rc = fail(-2, "op_mixffn_block_bf16: upload failed");
    if (!rc) rc = launch_mixffn16(x, n, h, w, c, e, dev + o2, dev + o1, dev, dev + o3, b2, ln_g, ln_b, y, st);
    if (const char *rep = getenv("EVFLY_MIXFFN_TIME")) {      // developer switch (tools/mixffn_check.py): time <n> more launches
        const int reps = std::max(1, atoi(rep));
        hipEvent_t e0, e1;
        if (!rc && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
            (void)hipEventRecord(e0, st);
            for (int i = 0; i < reps && !rc; ++i) rc = launch_mixffn16(x, n, h, w, c, e, dev + o2, dev + o1, dev, dev + o3, b2, ln_g, ln_b, y, st);
            (void)hipEventRecord(e1, st);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            fprintf(stderr, "mixffn16: %d frames, %.4f ms per launch\n", n, ms / reps);
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
    }
    if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = fail(-2, "op_mixffn_block_bf16: kernel failed");
    (void)hipFree(dev);
    return rc;
}

extern "C" int evfly_op_velpred_vec(const float *y, int64_t rows, int num_out, float *vel, void *stream) {
    EVFLY_REQUIRE(y && vel && rows > 0, "op_velpred_vec: null or empty argument");
    EVFLY_REQUIRE(num_out == 1 || num_out == 2, "op_velpred_vec: num_out must be 1 or 2 (3 is the identity)");
    return num_out == 1 ? launch_velpred_vec(y, rows, 1, vel, as_stream(stream)) : launch_velpred_vec2(y, rows, 2, vel, as_stream(stream));
}

// ---------------------------------------------------------------------------------------- stand-alone ConvLSTM layer
// workspace: [wx 4hid x ldx][wh 4hid x ldh][zx b*t*h*w x 4hid][z b*h*w x 4hid], each rounded up to 256 B
namespace {
struct ClstmWs { int ldx, ldh; int64_t o_wx, o_wh, o_zx, o_z, bytes; };
ClstmWs clstm_ws(int b, int t, int h, int w, int cin, int hid, int kh, int kw) {
    ClstmWs s;
    s.ldx = round_up(kh * kw * cin, 32); s.ldh = round_up(kh * kw * hid, 32);
    auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
    const int64_t co = 4 * (int64_t)hid, px = (int64_t)h * w;
    s.o_wx = 0;
    s.o_wh = s.o_wx + up(co * s.ldx * 4);
    s.o_zx = s.o_wh + up(co * s.ldh * 4);
    s.o_z = s.o_zx + up((int64_t)b * t * px * co * 4);
    s.bytes = s.o_z + up((int64_t)b * px * co * 4);
    return s;
}
}  // namespace

extern "C" int64_t evfly_convlstm_workspace_bytes(int b, int t, int h, int w, int cin, int hid, int kh, int kw) {
    if (b <= 0 || t <= 0 || h <= 0 || w <= 0 || cin <= 0 || hid <= 0 || kh <= 0 || kw <= 0) return 0;
    return clstm_ws(b, t, h, w, cin, hid, kh, kw).bytes;
}

extern "C" int evfly_convlstm_forward(const float *x, int b, int t, int h, int w, int cin, const float *weight, const float *bias,
                                      int hid, int kh, int kw, float *h_state, float *c_state, float *out, void *workspace,
                                      int64_t workspace_bytes, void *stream) {
    EVFLY_REQUIRE(x && weight && h_state && c_state && out && workspace, "convlstm_forward: null argument");
    EVFLY_REQUIRE(b > 0 && t > 0 && h > 0 && w > 0 && cin > 0 && hid > 0, "convlstm_forward: empty argument");
    EVFLY_REQUIRE(kh == kw && (kh & 1), "convlstm_forward: the kernels pad symmetrically (odd square kernel_size)");
    const ClstmWs s = clstm_ws(b, t, h, w, cin, hid, kh, kw);
    EVFLY_REQUIRE(workspace_bytes >= s.bytes, "convlstm_forward: workspace smaller than evfly_convlstm_workspace_bytes");
    hipStream_t st = as_stream(stream);
    char *ws = static_cast<char *>(workspace);
    float *wx = reinterpret_cast<float *>(ws + s.o_wx), *wh = reinterpret_cast<float *>(ws + s.o_wh);
    float *zx = reinterpret_cast<float *>(ws + s.o_zx), *z = reinterpret_cast<float *>(ws + s.o_z);
    const int co = 4 * hid, taps = kh * kw;
    if (int rc = launch_split_w(weight, co, cin + hid, 0, cin, taps, s.ldx, wx, st)) return rc;
    if (int rc = launch_split_w(weight, co, cin + hid, cin, hid, taps, s.ldh, wh, st)) return rc;
    const int64_t px = (int64_t)h * w;
    // input half of conv(cat[x, h]) + bias for every time step at once (convlstm.py:41 split along the input channels)
    ConvDesc d;
    d.x = x; d.ldx = cin; d.NI = b * t; d.H = h; d.W = w; d.C = cin; d.w = wx; d.ldw = s.ldx; d.bias = bias;
    d.KH = kh; d.KW = kw; d.stride = 1; d.pad = kh / 2;
    conv_finish(d);
    d.Nc = co; d.res = nullptr; d.ldres = co; d.act = EVFLY_ACT_NONE; d.y = zx; d.ldy = co; d.dtype = EVFLY_DTYPE_F32;
    if (int rc = igemm_launch(d, st)) return rc;
    // hidden half per step with the input half as the addend, then the cell (convlstm.py:161-164, :44-51)
    ConvDesc dh;
    dh.x = h_state; dh.ldx = hid; dh.NI = b; dh.H = h; dh.W = w; dh.C = hid; dh.w = wh; dh.ldw = s.ldh; dh.bias = nullptr;
    dh.KH = kh; dh.KW = kw; dh.stride = 1; dh.pad = kh / 2;
    conv_finish(dh);
    dh.Nc = co; dh.ldres = co; dh.act = EVFLY_ACT_NONE; dh.y = z; dh.ldy = co; dh.dtype = EVFLY_DTYPE_F32;
    for (int k = 0; k < t; ++k) {
        // output row (image i, pixel p) of step k takes row (i * t + k) * px + p of zx as its addend
        dh.res = zx + (int64_t)k * px * co; dh.res_rpi = (int)px; dh.res_img_rows = (int64_t)t * px;
        if (int rc = igemm_launch(dh, st)) return rc;
        if (int rc = launch_convlstm_gates(z, (int64_t)b * px, hid, c_state, h_state, out + (int64_t)k * px * hid, (int)px, (int64_t)t * px, st))
            return rc;
    }
    return 0;
}

extern "C" int evfly_op_convlstm_gates(const float *z, int64_t rows, int hid, float *c, float *h, void *stream) {
    EVFLY_REQUIRE(z && c && h && rows > 0 && hid > 0, "op_convlstm_gates: null or empty argument");
    return launch_convlstm_gates(z, rows, hid, c, h, nullptr, 1, 0, as_stream(stream));
}
