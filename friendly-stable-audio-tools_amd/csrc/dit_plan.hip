// DiT plan: owns the re-packed weights and drives one DiffusionTransformer forward
// (models/dit.py:135-226 + models/transformer.py:764-809, 656-702 of the reference) as a
// fixed sequence of HIP kernel launches on the caller's stream.  Host-side C++ only; the
// arithmetic lives in gemm_bf16.hip / attention.hip / layernorm.hip / dit_glue.hip.
#include <math.h>
#include <stdarg.h>
#include <stddef.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "dit_glue.h"
#include "sat_common.h"

// ------------------------------------------------------------------------------ errors
static thread_local std::string g_last_error;
void sat_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}
extern "C" const char* sat_last_error(void) { return g_last_error.c_str(); }

int sat_ensure_dynamic_lds(const void* kernel, int bytes) {
    // Launch-path cost: one thread-local table probe (no lock, no allocation) once a (kernel, device) pair has been seen by this
    // thread; the mutex-protected set is only consulted on a thread's first launch of a kernel on a device.
    struct Seen { const void* k; int dev; };
    static thread_local Seen seen[64];
    static thread_local int n_seen = 0;
    int dev = 0;
    SAT_HIP(hipGetDevice(&dev));
    for (int i = 0; i < n_seen; ++i)
        if (seen[i].k == kernel && seen[i].dev == dev) return 0;
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!done.count({kernel, dev})) {
            SAT_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
            done.insert({kernel, dev});
        }
    }
    if (n_seen < 64) seen[n_seen++] = Seen{kernel, dev};
    return 0;
}
// compute units of the current device: one attribute query per device and process (no allocation, no synchronisation); 0 on failure
int sat_device_cus() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        sat_set_error("hipGetDevice failed");
        return 0;
    }
    int c = (dev >= 0 && dev < 64) ? cache[dev].load(std::memory_order_relaxed) : 0;
    if (!c) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) {
            sat_set_error("hipDeviceGetAttribute(MultiprocessorCount) failed");
            return 0;
        }
        if (dev >= 0 && dev < 64) cache[dev].store(c, std::memory_order_relaxed);
    }
    return c;
}
extern "C" int sat_version(void) { return 6; }

// ------------------------------------------------------------------------------ plan
namespace {

struct Arena {
    char* base = nullptr;
    size_t off = 0;
    bool dry = true;
    void* take(size_t bytes) {
        size_t o = off;
        off += (size_t)round_up((int64_t)bytes, 256);
        return dry ? nullptr : base + o;
    }
};

struct LayerW {
    float *pre_g, *pre_b, *cross_g, *cross_b, *ff_g, *ff_b;
    op_t *w_qkv, *w_o, *w_cq, *w_ckv, *w_co, *w_ff1, *w_ff2;
    float *b_ff1, *b_ff2;
    float *s_qkv, *s_cq, *s_ff1, *s_ff2, *s_o, *s_co;      // gemm_dtype: per-output-channel scales (the weights above then hold e4m3 bytes)
    // LayerNorm folded into the GEMM behind it (plan->ln_fold): the weights above are bf16(gamma (.) W); GemmArgs::ln_c1 / ln_c2
    float *c1_qkv, *c2_qkv, *c1_cq, *c2_cq, *c1_ff1, *c2_ff1;
    bool fold_qkv;      // false for layer 0: its pre_norm reads rows written by the input projection, not by a GEMM epilogue
};

}  // namespace

struct sat_dit_plan {
    sat_dit_cfg cfg;
    std::map<std::string, std::pair<const float*, int64_t>> tensors;
    bool finalized = false;
    int inner = 0;          // FF inner dim
    int kvh_cross = 0;
    char* arena = nullptr;
    std::vector<LayerW> layers;
    float *ts_w, *te0_w, *te0_b, *te2_w, *te2_b;
    float *ce0_w, *ce2_w, *ge0_w, *ge2_w;
    float *win_eff, *wout_eff;
    float *rope_cos, *rope_sin, *inv_freq;
    int f16 = 0;                    // cfg.gemm_dtype == 3: every 16-bit operand buffer holds IEEE fp16 and the fp16 build of the kernels runs
    bool cross_fusion = true;       // cfg.cross_attention == 0: to_q + cross-attention core in one launch where it applies
    int tile_bits = 0;              // cfg.tile_policy as GemmArgs::variant bits (sat_tile_policy_bits)
    bool ln_fold = false;           // cfg.ln_fold, bf16 / fp16 operands, "prepend" conditioning: LayerNorms run inside the GEMM epilogues
    int fp8_mode = 2;               // 2: v_mfma_scale_f32_32x32x64_f8f6f4 (unit scales, 2x rate); 1: v_mfma_f32_32x32x16_fp8_fp8
    // gemm_dtype == 1: which GEMM families take e4m3 operands (sat_dit_cfg.fp8_families; SAT_FP8_* bits); 0 in every other mode
    bool f8_qkv = false, f8_cq = false, f8_ff1 = false, f8_ff2 = false, f8_o = false;
    float* ssg_w = nullptr;         // adaLN: [depth * 6D, D] stacked to_scale_shift_gate weights
    // per-generation context (sat_dit_prepare_context)
    char* ctx_buf = nullptr;
    size_t ctx_cap = 0;
    int ctx_bf = 0, ctx_lc = 0, ctx_lcpad = 0;
    bool has_global = false;
    int ctx_null_from = -1;         // sequences >= this index have an all-zero context (sat_dit_set_null_context_from)
    float* ge = nullptr;            // [bf, D] projected global embedding
    op_t* kc = nullptr;           // [depth][bf, kvh, lcpad, 64]
    op_t* vct = nullptr;          // [depth][bf, kvh, 64, lcpad]
    float* kc32 = nullptr;          // fp32 verification mode: [depth][bf, kvh, lc, 64]
    float* vc32 = nullptr;
    // optional HIP-event timing of the FFN-in (SwiGLU) GEMM of one layer per forward (sat_dit_profile)
    bool prof_on = false;
    int prof_n = 0;
    std::vector<hipEvent_t> prof_ev;   // pairs
    long long prof_m = 0, prof_nn = 0, prof_k = 0;
    // optional diagnostics of the residual stream (sat_dit_debug): [depth][3 updates][4] floats, overwritten by every forward while enabled
    float* dbg = nullptr;
};
static const int kProfMaxPairs = 4096;

namespace {

int get_tensor(sat_dit_plan* p, const std::string& name, int64_t numel, const float** out) {
    auto it = p->tensors.find(name);
    SAT_CHECK_ARG(it != p->tensors.end(), SAT_E_MISSING, "dit plan: tensor '%s' was never set", name.c_str());
    SAT_CHECK_ARG(numel < 0 || it->second.second == numel, SAT_E_INVALID, "dit plan: tensor '%s' has %lld elements, expected %lld",
                  name.c_str(), (long long)it->second.second, (long long)numel);
    *out = it->second.first;
    return 0;
}

int copy_f32(sat_dit_plan* p, Arena& ar, const std::string& name, int64_t numel, float** dst, hipStream_t s) {
    *dst = (float*)ar.take(numel * 4);
    if (ar.dry) return 0;
    const float* src;
    SAT_TRY(get_tensor(p, name, numel, &src));
    SAT_HIP(hipMemcpyAsync(*dst, src, numel * 4, hipMemcpyDeviceToDevice, s));
    return 0;
}

int pack_w(sat_dit_plan* p, Arena& ar, const std::string& name, int n, int k, int interleave, op_t** dst, hipStream_t s) {
    *dst = (op_t*)ar.take((size_t)n * k * 2);
    if (ar.dry) return 0;
    const float* src;
    SAT_TRY(get_tensor(p, name, (int64_t)n * k, &src));
    return sat_launch_pack_rows_bf16(src, *dst, n, k, interleave, s, p->f16);
}

// LayerNorm fold: bf16(gamma (.) W) + the two correction vectors of GemmArgs::ln_c1 / ln_c2 (bias folded into c2)
int pack_w_ln(sat_dit_plan* p, Arena& ar, const std::string& name, const float* gamma, const float* beta, const std::string& bias_name, int n,
              int k, int interleave, op_t** dst, float** c1, float** c2, hipStream_t s) {
    *dst = (op_t*)ar.take((size_t)n * k * 2);
    *c1 = (float*)ar.take((size_t)n * 4);
    *c2 = (float*)ar.take((size_t)n * 4);
    if (ar.dry) return 0;
    const float *src, *bias = nullptr;
    SAT_TRY(get_tensor(p, name, (int64_t)n * k, &src));
    if (!bias_name.empty()) SAT_TRY(get_tensor(p, bias_name, n, &bias));
    return sat_launch_pack_rows_ln(src, gamma, beta, bias, *dst, *c1, *c2, n, k, interleave, s, p->f16);
}

// gemm_dtype: e4m3 bytes + one scale per output channel instead of bf16
int pack_w8(sat_dit_plan* p, Arena& ar, const std::string& name, int n, int k, int interleave, op_t** dst, float** scale,
            hipStream_t s) {
    *dst = (op_t*)ar.take((size_t)n * k);
    *scale = (float*)ar.take((size_t)n * 4);
    if (ar.dry) return 0;
    const float* src;
    SAT_TRY(get_tensor(p, name, (int64_t)n * k, &src));
    return sat_launch_quant_rows_fp8(src, *dst, *scale, n, k, interleave, s);
}

int build(sat_dit_plan* p, Arena& ar, hipStream_t s) {
    const sat_dit_cfg& c = p->cfg;
    const int D = c.embed_dim, C = c.io_channels, Dc = c.cond_embed_dim, Dct = c.cond_token_dim, Dg = c.global_cond_dim;
    const int inner = p->inner;
    SAT_TRY(copy_f32(p, ar, "timestep_features.weight", 128, &p->ts_w, s));
    SAT_TRY(copy_f32(p, ar, "to_timestep_embed.0.weight", (int64_t)D * 256, &p->te0_w, s));
    SAT_TRY(copy_f32(p, ar, "to_timestep_embed.0.bias", D, &p->te0_b, s));
    SAT_TRY(copy_f32(p, ar, "to_timestep_embed.2.weight", (int64_t)D * D, &p->te2_w, s));
    SAT_TRY(copy_f32(p, ar, "to_timestep_embed.2.bias", D, &p->te2_b, s));
    if (Dct > 0) {
        SAT_TRY(copy_f32(p, ar, "to_cond_embed.0.weight", (int64_t)Dc * Dct, &p->ce0_w, s));
        SAT_TRY(copy_f32(p, ar, "to_cond_embed.2.weight", (int64_t)Dc * Dc, &p->ce2_w, s));
    }
    if (Dg > 0) {
        SAT_TRY(copy_f32(p, ar, "to_global_embed.0.weight", (int64_t)D * Dg, &p->ge0_w, s));
        SAT_TRY(copy_f32(p, ar, "to_global_embed.2.weight", (int64_t)D * D, &p->ge2_w, s));
    }
    SAT_TRY(copy_f32(p, ar, "transformer.rotary_pos_emb.inv_freq", 16, &p->inv_freq, s));
    p->win_eff = (float*)ar.take((size_t)D * C * 4);
    p->wout_eff = (float*)ar.take((size_t)D * C * 4);
    const int smax = c.max_seq_len + 1;
    p->rope_cos = (float*)ar.take((size_t)smax * 16 * 4);
    p->rope_sin = (float*)ar.take((size_t)smax * 16 * 4);
    if (!ar.dry) {
        const float *win, *wpre, *wout, *wpost;
        SAT_TRY(get_tensor(p, "transformer.project_in.weight", (int64_t)D * C, &win));
        SAT_TRY(get_tensor(p, "preprocess_conv.weight", (int64_t)C * C, &wpre));
        SAT_TRY(get_tensor(p, "transformer.project_out.weight", (int64_t)D * C, &wout));
        SAT_TRY(get_tensor(p, "postprocess_conv.weight", (int64_t)C * C, &wpost));
        SAT_TRY(glue_fold_in(win, wpre, p->win_eff, D, C, s));
        SAT_TRY(glue_fold_out(wout, wpost, p->wout_eff, D, C, s));
        SAT_TRY(sat_launch_rope_table(p->inv_freq, p->rope_cos, p->rope_sin, smax, s));
    }
    if (c.adaln) p->ssg_w = (float*)ar.take((size_t)c.depth * 6 * D * D * 4);
    p->layers.resize(c.depth);
    for (int l = 0; l < c.depth; ++l) {
        LayerW& L = p->layers[l];
        const std::string pf = "transformer.layers." + std::to_string(l) + ".";
        if (c.adaln && !ar.dry) {   // transformer.py:651-655: Sequential(SiLU, Linear(D, 6D, bias=False)) -> key "...1.weight"
            const float* wsrc;
            SAT_TRY(get_tensor(p, pf + "to_scale_shift_gate.1.weight", (int64_t)6 * D * D, &wsrc));
            SAT_HIP(hipMemcpyAsync(p->ssg_w + (size_t)l * 6 * D * D, wsrc, (size_t)6 * D * D * 4, hipMemcpyDeviceToDevice, s));
        }
        SAT_TRY(copy_f32(p, ar, pf + "pre_norm.gamma", D, &L.pre_g, s));
        SAT_TRY(copy_f32(p, ar, pf + "pre_norm.beta", D, &L.pre_b, s));
        SAT_TRY(copy_f32(p, ar, pf + "ff_norm.gamma", D, &L.ff_g, s));
        SAT_TRY(copy_f32(p, ar, pf + "ff_norm.beta", D, &L.ff_b, s));
        if (c.gemm_dtype == 2) {      // fp32 verification mode: the reference's own fp32 weights, no re-packing (f32_ref.hip)
            auto w32 = [&](const std::string& name, int64_t numel, op_t** dst) { return copy_f32(p, ar, pf + name, numel, (float**)dst, s); };
            SAT_TRY(w32("self_attn.to_qkv.weight", (int64_t)3 * D * D, &L.w_qkv));
            SAT_TRY(w32("self_attn.to_out.weight", (int64_t)D * D, &L.w_o));
            if (Dct > 0) {
                SAT_TRY(copy_f32(p, ar, pf + "cross_attend_norm.gamma", D, &L.cross_g, s));
                SAT_TRY(copy_f32(p, ar, pf + "cross_attend_norm.beta", D, &L.cross_b, s));
                SAT_TRY(w32("cross_attn.to_q.weight", (int64_t)D * D, &L.w_cq));
                SAT_TRY(w32("cross_attn.to_kv.weight", (int64_t)2 * Dc * Dc, &L.w_ckv));
                SAT_TRY(w32("cross_attn.to_out.weight", (int64_t)D * D, &L.w_co));
            }
            SAT_TRY(w32("ff.ff.0.proj.weight", (int64_t)2 * inner * D, &L.w_ff1));
            SAT_TRY(copy_f32(p, ar, pf + "ff.ff.0.proj.bias", 2 * inner, &L.b_ff1, s));
            SAT_TRY(w32("ff.ff.2.weight", (int64_t)D * inner, &L.w_ff2));
            SAT_TRY(copy_f32(p, ar, pf + "ff.ff.2.bias", D, &L.b_ff2, s));
            continue;
        }
        const bool lf = p->ln_fold;
        L.fold_qkv = lf && l > 0;
        if (p->f8_qkv) SAT_TRY(pack_w8(p, ar, pf + "self_attn.to_qkv.weight", 3 * D, D, 0, &L.w_qkv, &L.s_qkv, s));
        else if (L.fold_qkv) SAT_TRY(pack_w_ln(p, ar, pf + "self_attn.to_qkv.weight", L.pre_g, L.pre_b, "", 3 * D, D, 0, &L.w_qkv, &L.c1_qkv, &L.c2_qkv, s));
        else SAT_TRY(pack_w(p, ar, pf + "self_attn.to_qkv.weight", 3 * D, D, 0, &L.w_qkv, s));
        if (p->f8_o) SAT_TRY(pack_w8(p, ar, pf + "self_attn.to_out.weight", D, D, 0, &L.w_o, &L.s_o, s));
        else SAT_TRY(pack_w(p, ar, pf + "self_attn.to_out.weight", D, D, 0, &L.w_o, s));
        if (Dct > 0) {
            SAT_TRY(copy_f32(p, ar, pf + "cross_attend_norm.gamma", D, &L.cross_g, s));
            SAT_TRY(copy_f32(p, ar, pf + "cross_attend_norm.beta", D, &L.cross_b, s));
            if (p->f8_cq) SAT_TRY(pack_w8(p, ar, pf + "cross_attn.to_q.weight", D, D, 0, &L.w_cq, &L.s_cq, s));
            else if (lf) SAT_TRY(pack_w_ln(p, ar, pf + "cross_attn.to_q.weight", L.cross_g, L.cross_b, "", D, D, 0, &L.w_cq, &L.c1_cq, &L.c2_cq, s));
            else SAT_TRY(pack_w(p, ar, pf + "cross_attn.to_q.weight", D, D, 0, &L.w_cq, s));
            SAT_TRY(pack_w(p, ar, pf + "cross_attn.to_kv.weight", 2 * Dc, Dc, 0, &L.w_ckv, s));
            if (p->f8_o) SAT_TRY(pack_w8(p, ar, pf + "cross_attn.to_out.weight", D, D, 0, &L.w_co, &L.s_co, s));
            else SAT_TRY(pack_w(p, ar, pf + "cross_attn.to_out.weight", D, D, 0, &L.w_co, s));
        }
        if (p->f8_ff1) SAT_TRY(pack_w8(p, ar, pf + "ff.ff.0.proj.weight", 2 * inner, D, 1, &L.w_ff1, &L.s_ff1, s));
        else if (lf) SAT_TRY(pack_w_ln(p, ar, pf + "ff.ff.0.proj.weight", L.ff_g, L.ff_b, pf + "ff.ff.0.proj.bias", 2 * inner, D, 1, &L.w_ff1, &L.c1_ff1,
                                       &L.c2_ff1, s));
        else SAT_TRY(pack_w(p, ar, pf + "ff.ff.0.proj.weight", 2 * inner, D, 1, &L.w_ff1, s));
        L.b_ff1 = (float*)ar.take((size_t)2 * inner * 4);
        if (!ar.dry) {
            const float* b1;
            SAT_TRY(get_tensor(p, pf + "ff.ff.0.proj.bias", 2 * inner, &b1));
            SAT_TRY(sat_launch_pack_bias(b1, L.b_ff1, 2 * inner, 1, s));
        }
        if (p->f8_ff2) SAT_TRY(pack_w8(p, ar, pf + "ff.ff.2.weight", D, inner, 0, &L.w_ff2, &L.s_ff2, s));
        else SAT_TRY(pack_w(p, ar, pf + "ff.ff.2.weight", D, inner, 0, &L.w_ff2, s));
        SAT_TRY(copy_f32(p, ar, pf + "ff.ff.2.bias", D, &L.b_ff2, s));
    }
    return 0;
}

struct Workspace {
    float* X;
    op_t *A, *AO, *Q, *K, *Vt, *Hh;
    float *ff, *h1, *mo;
    float* As;              // gemm_dtype: per-row scale of the quantised LayerNorm output in A
    unsigned char* Hs;      // gemm_dtype: E8M0 block scales of the MXFP8 hidden activation in Hh, [M][inner / 32]
    unsigned char* AOs;     // gemm_dtype: E8M0 block scales of the MXFP8 attention output in AO, [M][D / 32]
    float *gsum, *ssg;      // adaLN: silu(global + timestep embed) [bf, D]; per-layer modulation [bf, depth, 6, D]
    float* ln_part = nullptr;    // ln_fold: [M][D / 64][2] partial (sum, sum of squares) of the bf16 image of X kept in A
    float* f32_wide = nullptr;   // fp32 verification mode: [M, max(3D, 2 inner)] GEMM output before the head split / SwiGLU
    float* slab = nullptr;       // K-split scratch of the 8-phase FF-out GEMM (GemmArgs::slab), present where that schedule splits
    size_t slab_bytes = 0;
    size_t qkv_bytes;
    size_t total;
};

Workspace carve(const sat_dit_plan* p, int bf, int T, char* base) {
    const sat_dit_cfg& c = p->cfg;
    const int D = c.embed_dim, H = c.num_heads;
    const int S = T + (c.adaln ? 0 : 1);
    const size_t M = (size_t)bf * S;
    const int Spad = (int)round_up(S + 3, 128);
    Workspace w;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* ptr = base ? base + off : nullptr;
        off += (size_t)round_up((int64_t)bytes, 256);
        return ptr;
    };
    w.X = (float*)take(M * D * 4);
    if (c.gemm_dtype == 2) {      // fp32 verification mode: every intermediate is fp32, q / k / v un-padded [bf, H, S, 64]
        w.A = (op_t*)take(M * D * 4);
        w.AO = (op_t*)take(M * D * 4);
        w.qkv_bytes = M * D * 4;
        w.Q = (op_t*)take(w.qkv_bytes);
        w.K = (op_t*)take(w.qkv_bytes);
        w.Vt = (op_t*)take(w.qkv_bytes);
        w.Hh = (op_t*)take(M * (size_t)p->inner * 4);
        w.f32_wide = (float*)take(M * (size_t)(2 * p->inner > 3 * D ? 2 * p->inner : 3 * D) * 4);     // [M, 3D] qkv / [M, 2 inner] FF-in
        w.ff = (float*)take((size_t)bf * 256 * 4);
        w.h1 = (float*)take((size_t)bf * D * 4);
        w.mo = (float*)take((size_t)bf * c.io_channels * T * 4);
        w.As = nullptr; w.Hs = nullptr; w.AOs = nullptr;
        w.gsum = c.adaln ? (float*)take((size_t)bf * D * 4) : nullptr;
        w.ssg = c.adaln ? (float*)take((size_t)bf * c.depth * 6 * D * 4) : nullptr;
        w.total = off;
        return w;
    }
    w.A = (op_t*)take(M * D * 2);
    w.AO = (op_t*)take(M * D * 2);
    w.qkv_bytes = (size_t)bf * H * Spad * 64 * 2;
    // Q, K, Vt contiguous so that one memset clears all pads
    w.Q = (op_t*)take(w.qkv_bytes);
    w.K = (op_t*)take(w.qkv_bytes);
    w.Vt = (op_t*)take(w.qkv_bytes);
    w.Hh = (op_t*)take(M * (size_t)p->inner * 2);
    w.ff = (float*)take((size_t)bf * 256 * 4);
    w.h1 = (float*)take((size_t)bf * D * 4);
    w.mo = (float*)take((size_t)bf * c.io_channels * T * 4);
    w.As = c.gemm_dtype == 1 ? (float*)take(M * 4) : nullptr;
    w.Hs = c.gemm_dtype == 1 ? (unsigned char*)take(M * (size_t)(p->inner / 32)) : nullptr;
    w.AOs = c.gemm_dtype == 1 ? (unsigned char*)take(M * (size_t)(D / 32)) : nullptr;
    w.gsum = c.adaln ? (float*)take((size_t)bf * D * 4) : nullptr;
    w.ssg = c.adaln ? (float*)take((size_t)bf * c.depth * 6 * D * 4) : nullptr;
    w.ln_part = p->ln_fold ? (float*)take(M * (size_t)(D / 64) * 2 * 4) : nullptr;
    // FF-out (K = inner) is the one fp32-output GEMM with a reduction long enough for the 8-phase kernel's K-split of the remainder
    // round (SA-2.0 shape): its slabs live here, per workspace = per caller and stream
    w.slab_bytes = (c.gemm_dtype == 0 || c.gemm_dtype == 3) ? sat_gemm_ph8_slab_bytes(EPI_RESID, (int)M, D, p->inner) : 0;
    w.slab = w.slab_bytes ? (float*)take(w.slab_bytes) : nullptr;
    w.total = off;
    return w;
}

int run_forward(sat_dit_plan* p, const float* x, int xB, float xscale, const float* t_dev, float t_const, float* out, int bf,
                int T, void* ws, size_t ws_bytes, hipStream_t s) {
    SAT_CHECK_ARG(p && p->finalized, SAT_E_STATE, "dit forward: plan not finalized");
    const sat_dit_cfg& c = p->cfg;
    SAT_CHECK_ARG(x && out && ws && bf > 0 && T > 0, SAT_E_INVALID, "dit forward: bad arguments");
    SAT_CHECK_ARG(T <= c.max_seq_len, SAT_E_INVALID, "dit forward: t_len %d exceeds plan max_seq_len %d", T, c.max_seq_len);
    SAT_CHECK_ARG(((uintptr_t)ws & 255) == 0, SAT_E_INVALID, "dit forward: workspace must be 256-byte aligned");
    const bool cross = c.cond_token_dim > 0;
    SAT_CHECK_ARG(p->ctx_bf == bf, SAT_E_STATE, "dit forward: context prepared for %d sequences, forward called with %d",
                  p->ctx_bf, bf);
    Workspace w = carve(p, bf, T, (char*)ws);
    SAT_CHECK_ARG(ws_bytes >= w.total, SAT_E_WORKSPACE, "dit forward: workspace %zu < required %zu", ws_bytes, w.total);
    const int D = c.embed_dim, H = c.num_heads, C = c.io_channels;
    const bool adaln = c.adaln != 0, f32 = c.gemm_dtype == 2;
    const int f16 = p->f16;
    const int S = T + (adaln ? 0 : 1), M = bf * S, Spad = (int)round_up(S + 3, 128);
    const int ssg_ld = c.depth * 6 * D;      // per-sequence stride of the adaLN modulation vectors

    // pads of q/k/vt must be finite (zero): one memset per forward
    if (!f32) SAT_HIP(hipMemsetAsync(w.Q, 0, 3 * (size_t)round_up((int64_t)w.qkv_bytes, 256), s));

    // timestep embedding (dit.py:176) + global embed (dit.py:179-182) -> prepend token rows X[b,0,:]
    SAT_TRY(glue_fourier(t_dev, t_const, p->ts_w, w.ff, bf, 128, s));
    SAT_TRY(glue_small_linear(w.ff, 256, p->te0_w, p->te0_b, nullptr, 0, w.h1, D, bf, D, 256, 1, 0, s));
    if (!adaln) {
        SAT_TRY(glue_small_linear(w.h1, D, p->te2_w, p->te2_b, p->has_global ? p->ge : nullptr, D, w.X, S * D, bf, D, D, 0, 0, s));
    } else {
        // dit.py:205-206: the summed embedding conditions every block instead of being prepended; transformer.py:667:
        // (scale, shift, gate) x (self, ff) = Linear(SiLU(global)) for all layers in one launch
        SAT_TRY(glue_small_linear(w.h1, D, p->te2_w, p->te2_b, p->has_global ? p->ge : nullptr, D, w.gsum, D, bf, D, D, 2, 0, s));
        SAT_TRY(glue_small_linear(w.gsum, D, p->ssg_w, nullptr, nullptr, 0, w.ssg, ssg_ld, bf, ssg_ld, D, 0, 0, s));
        SAT_TRY(glue_adaln_finish(w.ssg, (int64_t)bf * ssg_ld, D, s));
    }
    // preprocess_conv + residual + project_in (dit.py:197-199, transformer.py:778)
    SAT_TRY(glue_input_proj(x, p->win_eff, w.X, bf, xB, C, T, S, D, xscale, s));

    if (p->dbg) SAT_HIP(hipMemsetAsync(p->dbg, 0, (size_t)c.depth * 3 * 4 * sizeof(float), s));
    GemmArgs g{};
    for (int l = 0; l < c.depth && f32; ++l) {
        // fp32 verification mode: the same block (transformer.py:656-702) on f32_ref.hip, fp32 everywhere
        const LayerW& L = p->layers[l];
        const float* mod = adaln ? w.ssg + (size_t)l * 6 * D : nullptr;
        float *A32 = (float*)w.A, *AO32 = (float*)w.AO, *Q32 = (float*)w.Q, *K32 = (float*)w.K, *V32 = (float*)w.Vt, *H32 = (float*)w.Hh;
        SAT_TRY(sat_launch_layernorm_f32(w.X, L.pre_g, L.pre_b, A32, M, D, mod, mod ? mod + D : nullptr, S, ssg_ld, s));
        SAT_TRY(sat_launch_gemm_f32(A32, (const float*)L.w_qkv, nullptr, w.f32_wide, M, 3 * D, D, 3 * D, 0, nullptr, 1, 0, s));
        SAT_TRY(sat_launch_split_heads_f32(w.f32_wide, Q32, K32, V32, M, S, 3, H, 3, p->rope_cos, p->rope_sin, s));
        SAT_TRY(sat_launch_attention_f32(Q32, K32, V32, AO32, bf, H, H, S, S, s));
        SAT_TRY(sat_launch_gemm_f32(AO32, (const float*)L.w_o, nullptr, w.X, M, D, D, D, 1, adaln ? mod + 2 * D : nullptr, S, ssg_ld, s));
        if (cross) {
            const int bc = (p->ctx_null_from >= 0 && p->ctx_null_from < bf) ? p->ctx_null_from : bf;
            const int Mc = bc * S;
            if (bc > 0) {
                SAT_TRY(sat_launch_layernorm_f32(w.X, L.cross_g, L.cross_b, A32, Mc, D, nullptr, nullptr, 1, 0, s));
                SAT_TRY(sat_launch_gemm_f32(A32, (const float*)L.w_cq, nullptr, w.f32_wide, Mc, D, D, D, 0, nullptr, 1, 0, s));
                SAT_TRY(sat_launch_split_heads_f32(w.f32_wide, Q32, nullptr, nullptr, Mc, S, 1, H, 0, nullptr, nullptr, s));
                const size_t per_layer = (size_t)bf * p->kvh_cross * p->ctx_lc * 64;
                SAT_TRY(sat_launch_attention_f32(Q32, p->kc32 + l * per_layer, p->vc32 + l * per_layer, AO32, bc, H, p->kvh_cross, S, p->ctx_lc, s));
                SAT_TRY(sat_launch_gemm_f32(AO32, (const float*)L.w_co, nullptr, w.X, Mc, D, D, D, 1, nullptr, 1, 0, s));
            }
        }
        SAT_TRY(sat_launch_layernorm_f32(w.X, L.ff_g, L.ff_b, A32, M, D, mod ? mod + 3 * D : nullptr, mod ? mod + 4 * D : nullptr, S, ssg_ld, s));
        SAT_TRY(sat_launch_gemm_f32(A32, (const float*)L.w_ff1, L.b_ff1, w.f32_wide, M, 2 * p->inner, D, 2 * p->inner, 0, nullptr, 1, 0, s));
        SAT_TRY(sat_launch_swiglu_f32(w.f32_wide, H32, M, p->inner, s));
        SAT_TRY(sat_launch_gemm_f32(H32, (const float*)L.w_ff2, L.b_ff2, w.X, M, D, p->inner, D, 1, adaln ? mod + 5 * D : nullptr, S, ssg_ld, s));
    }
    for (int l = 0; l < c.depth && !f32; ++l) {
        const LayerW& L = p->layers[l];
        // ---- self-attention branch (transformer.py:692)
        const float* mod = adaln ? w.ssg + (size_t)l * 6 * D : nullptr;     // + {0..5} * D: scale1p/shift/gate self, then ff
        // ln_fold: from the first to_out on, A holds bf16(X) and ln_part its row statistics, both written by the epilogue of the
        // GEMM that last updated X; the LayerNorms (transformer.py:692, 695, 700) are finished in the epilogues of their consumers
        const bool lf = p->ln_fold;
        auto fold_in = [&](GemmArgs& ga, const float* c1, const float* c2) {
            ga.ln_part = w.ln_part; ga.ln_c1 = c1; ga.ln_c2 = c2; ga.ln_eps = 1e-5f;
        };
        auto fold_out = [&](GemmArgs& ga) {
            if (lf) { ga.xb = w.A; ga.ln_part_out = w.ln_part; }
        };
        if (p->f8_qkv) SAT_TRY(sat_launch_layernorm_fp8(w.X, L.pre_g, L.pre_b, w.A, w.As, M, D, mod, mod ? mod + D : nullptr, S, ssg_ld, s));
        else if (!L.fold_qkv) SAT_TRY(sat_launch_layernorm_mod(w.X, L.pre_g, L.pre_b, w.A, M, D, mod, mod ? mod + D : nullptr, S, ssg_ld, s, f16));
        g = GemmArgs{}; g.f16 = f16; g.variant = p->tile_bits;
        g.A = w.A; g.W = L.w_qkv; g.M = M; g.N = 3 * D; g.K = D;
        if (L.fold_qkv) fold_in(g, L.c1_qkv, L.c2_qkv);
        if (p->f8_qkv) { g.fp8 = p->fp8_mode; g.a_scale = w.As; g.w_scale = L.s_qkv; }
        g.heads.out[0] = w.Q; g.heads.out[1] = w.K; g.heads.out[2] = w.Vt;
        g.heads.kind[0] = 2 | 8; g.heads.kind[1] = 2 | 4; g.heads.kind[2] = 1 | 4; g.heads.qscale = SAT_ATTN_QSCALE;
        g.heads.parts = 3; g.heads.heads = H; g.heads.S = S; g.heads.Spad = Spad;
        g.heads.rope_cos = p->rope_cos; g.heads.rope_sin = p->rope_sin;
        SAT_TRY(sat_launch_gemm(EPI_HEADS, g, s));
        SAT_TRY(sat_launch_attention(w.Q, w.K, w.Vt, w.AO, bf, H, H, S, S, Spad, Spad, s, p->f8_o ? w.AOs : nullptr, 1.0f, f16));
        g = GemmArgs{}; g.f16 = f16; g.variant = p->tile_bits;
        g.A = w.AO; g.W = L.w_o; g.M = M; g.N = D; g.K = D; g.C = w.X; g.ldc = D; g.accumulate = 1;
        if (p->f8_o) { g.fp8 = 3; g.a_bscale = (const unsigned*)w.AOs; g.w_scale = L.s_o; }
        if (adaln) { g.gate = mod + 2 * D; g.gate_rows = S; g.gate_ld = ssg_ld; }
        fold_out(g);
        SAT_TRY(sat_launch_gemm(EPI_RESID, g, s));
        if (p->dbg) SAT_TRY(glue_resid_stats(w.X, M, D, p->dbg + ((size_t)l * 3 + 0) * 4, s));
        // ---- cross-attention branch (transformer.py:694-695).  Sequences whose context is all-zero (the
        // unconditional CFG half, dit.py:294-300) get k = v = 0 from the bias-free to_cond_embed / to_kv, hence an
        // attention output of exactly 0 and, through the bias-free to_out, a branch contribution of exactly 0:
        // the branch runs only on the first `bc` sequences (rows are ordered by sequence).
        if (cross) {
            const int bc = (p->ctx_null_from >= 0 && p->ctx_null_from < bf) ? p->ctx_null_from : bf;
            const int Mc = bc * S;
            if (bc > 0) {
                if (p->f8_cq) SAT_TRY(sat_launch_layernorm_fp8(w.X, L.cross_g, L.cross_b, w.A, w.As, Mc, D, nullptr, nullptr, 1, 0, s));
                else if (!lf) SAT_TRY(sat_launch_layernorm(w.X, L.cross_g, L.cross_b, w.A, Mc, D, s, f16));
                g = GemmArgs{}; g.f16 = f16; g.variant = p->tile_bits;
                g.A = w.A; g.W = L.w_cq; g.M = Mc; g.N = D; g.K = D;
                if (lf) fold_in(g, L.c1_cq, L.c2_cq);
                if (p->f8_cq) { g.fp8 = p->fp8_mode; g.a_scale = w.As; g.w_scale = L.s_cq; }
                g.heads.out[0] = w.Q; g.heads.kind[0] = 8; g.heads.qscale = SAT_ATTN_QSCALE;
                g.heads.parts = 1; g.heads.heads = H; g.heads.S = S; g.heads.Spad = Spad;
                const size_t per_layer = (size_t)bf * p->kvh_cross * p->ctx_lcpad * 64;
                // One launch for to_q + softmax(q k^T) v where the 128 x 64 tile is the choice anyway and its workgroups fit one round
                // (one prompt: 9 x 24 = 216): the projection's epilogue keeps Q in registers and attends to the <= 189 context keys
                // staged in LDS (gemm_bf16.hip, XA_OK).  Saves the attention launch and the Q round trip.
                const bool fuse = p->cross_fusion && !p->f8_cq && !p->f8_o && D >= 192 && p->ctx_lc + 3 <= 192 && cdiv(Mc, 128) * (D / 64) <= 256;
                if (fuse) {
                    g.heads.xa_k = p->kc + l * per_layer; g.heads.xa_vt = p->vct + l * per_layer; g.heads.xa_out = w.AO;
                    g.heads.xa_kvh = p->kvh_cross; g.heads.xa_sk = p->ctx_lc; g.heads.xa_sk_pad = p->ctx_lcpad;
                }
                SAT_TRY(sat_launch_gemm(EPI_HEADS, g, s));
                if (!fuse)
                    SAT_TRY(sat_launch_attention(w.Q, p->kc + l * per_layer, p->vct + l * per_layer, w.AO, bc, H, p->kvh_cross, S,
                                                 p->ctx_lc, Spad, p->ctx_lcpad, s, p->f8_o ? w.AOs : nullptr, 1.0f, f16));
                g = GemmArgs{}; g.f16 = f16; g.variant = p->tile_bits;
                g.A = w.AO; g.W = L.w_co; g.M = Mc; g.N = D; g.K = D; g.C = w.X; g.ldc = D; g.accumulate = 1;
                if (p->f8_o) { g.fp8 = 3; g.a_bscale = (const unsigned*)w.AOs; g.w_scale = L.s_co; }
                fold_out(g);
                SAT_TRY(sat_launch_gemm(EPI_RESID, g, s));
                if (p->dbg) SAT_TRY(glue_resid_stats(w.X, Mc, D, p->dbg + ((size_t)l * 3 + 1) * 4, s));
            }
        }
        // ---- feed-forward branch (transformer.py:700)
        if (p->f8_ff1) SAT_TRY(sat_launch_layernorm_fp8(w.X, L.ff_g, L.ff_b, w.A, w.As, M, D, mod ? mod + 3 * D : nullptr, mod ? mod + 4 * D : nullptr,
                                                       S, ssg_ld, s));
        else if (!lf) SAT_TRY(sat_launch_layernorm_mod(w.X, L.ff_g, L.ff_b, w.A, M, D, mod ? mod + 3 * D : nullptr, mod ? mod + 4 * D : nullptr, S,
                                                       ssg_ld, s, f16));
        g = GemmArgs{}; g.f16 = f16; g.variant = p->tile_bits;
        g.A = w.A; g.W = L.w_ff1; g.bias = L.b_ff1; g.M = M; g.N = 2 * p->inner; g.K = D; g.H = w.Hh;
        if (lf) { g.bias = nullptr; fold_in(g, L.c1_ff1, L.c2_ff1); }
        if (p->f8_ff1) {
            g.fp8 = p->fp8_mode; g.a_scale = w.As; g.w_scale = L.s_ff1;
            if (p->f8_ff2) { g.H8 = (unsigned char*)w.Hh; g.Hs = w.Hs; }          // FF-out's MXFP8 operand; otherwise the e4m3 GEMM writes a bf16 hidden state
        }
        const bool prof = p->prof_on && l == c.depth / 2 && p->prof_n < kProfMaxPairs;
        if (prof) {
            if ((int)p->prof_ev.size() < 2 * (p->prof_n + 1)) {
                hipEvent_t e0, e1;
                SAT_HIP(hipEventCreate(&e0));
                SAT_HIP(hipEventCreate(&e1));
                p->prof_ev.push_back(e0);
                p->prof_ev.push_back(e1);
            }
            SAT_HIP(hipEventRecord(p->prof_ev[2 * p->prof_n], s));
        }
        SAT_TRY(sat_launch_gemm(EPI_SWIGLU, g, s));
        if (prof) {
            SAT_HIP(hipEventRecord(p->prof_ev[2 * p->prof_n + 1], s));
            p->prof_n++;
            p->prof_m = g.M; p->prof_nn = g.N; p->prof_k = g.K;
        }
        g = GemmArgs{}; g.f16 = f16; g.variant = p->tile_bits;
        g.A = w.Hh; g.W = L.w_ff2; g.bias = L.b_ff2; g.M = M; g.N = D; g.K = p->inner; g.C = w.X; g.ldc = D; g.accumulate = 1;
        if (p->f8_ff2) { g.fp8 = 3; g.a_bscale = (const unsigned*)w.Hs; g.w_scale = L.s_ff2; }
        if (adaln) { g.gate = mod + 5 * D; g.gate_rows = S; g.gate_ld = ssg_ld; }
        g.slab = w.slab; g.slab_bytes = w.slab_bytes;
        if (l + 1 < c.depth) fold_out(g);       // nobody normalises the output of the last block
        SAT_TRY(sat_launch_gemm(EPI_RESID, g, s));
        if (p->dbg) SAT_TRY(glue_resid_stats(w.X, M, D, p->dbg + ((size_t)l * 3 + 2) * 4, s));
    }
    // project_out + drop prepend + postprocess_conv + residual (transformer.py:807, dit.py:219-224)
    SAT_TRY(glue_output_proj(w.X, p->wout_eff, out, bf, C, T, S, D, s));
    return 0;
}

}  // namespace

// ------------------------------------------------------------------------------ C ABI
extern "C" int sat_dit_plan_create(const sat_dit_cfg* cfg, sat_dit_plan** out_plan) {
    return sat_dit_plan_create_sized(cfg, SAT_DIT_CFG_BYTES_V5, out_plan);
}

extern "C" int sat_dit_plan_create_sized(const sat_dit_cfg* cfg_in, size_t cfg_bytes, sat_dit_plan** out_plan) {
    SAT_CHECK_ARG(cfg_in && out_plan, SAT_E_INVALID, "dit_plan_create: null argument");
    static_assert(sizeof(sat_dit_cfg) == SAT_DIT_CFG_BYTES_V5, "sat_dit_cfg layout");
    // (the one layout this library knows; when the struct grows again, the older sizes are accepted here and the fields behind them defaulted)
    SAT_CHECK_ARG(cfg_bytes == sizeof(sat_dit_cfg), SAT_E_INVALID, "dit_plan_create: sat_dit_cfg of %zu bytes; this library (ABI version %d) knows %zu",
                  cfg_bytes, sat_version(), sizeof(sat_dit_cfg));
    sat_dit_cfg cfg_local{};
    memcpy(&cfg_local, cfg_in, cfg_bytes);
    const sat_dit_cfg* cfg = &cfg_local;
    SAT_CHECK_ARG(cfg->embed_dim > 0 && cfg->num_heads > 0 && cfg->embed_dim == cfg->num_heads * 64, SAT_E_UNSUPPORTED,
                  "dit_plan_create: dim_heads must be 64 (embed_dim %d, heads %d)", cfg->embed_dim, cfg->num_heads);
    SAT_CHECK_ARG(cfg->embed_dim % 128 == 0 && cfg->embed_dim <= 2048, SAT_E_UNSUPPORTED,
                  "dit_plan_create: embed_dim %d must be a multiple of 128 and <= 2048", cfg->embed_dim);
    // (the input / output projection kernels move 4 channels per lane: glue_output_proj checks the same at forward time)
    SAT_CHECK_ARG(cfg->io_channels > 0 && cfg->io_channels <= 64 && cfg->io_channels % 4 == 0, SAT_E_UNSUPPORTED,
                  "dit_plan_create: io_channels %d must be a multiple of 4 in 4..64", cfg->io_channels);
    SAT_CHECK_ARG(cfg->depth > 0 && cfg->max_seq_len > 0, SAT_E_INVALID, "dit_plan_create: depth/max_seq_len must be positive");
    if (cfg->cond_token_dim > 0) {
        SAT_CHECK_ARG(cfg->cond_embed_dim % 64 == 0 && cfg->cond_embed_dim > 0 && cfg->cond_token_dim % 4 == 0, SAT_E_UNSUPPORTED,
                      "dit_plan_create: cond_embed_dim %d must be a multiple of 64", cfg->cond_embed_dim);
        int kvh = cfg->cond_embed_dim / 64;
        SAT_CHECK_ARG(cfg->num_heads % kvh == 0, SAT_E_UNSUPPORTED, "dit_plan_create: %d query heads not divisible by %d kv heads",
                      cfg->num_heads, kvh);
    }
    SAT_CHECK_ARG(cfg->global_cond_dim % 4 == 0, SAT_E_UNSUPPORTED, "dit_plan_create: global_cond_dim must be a multiple of 4");
    SAT_CHECK_ARG(cfg->gemm_dtype >= 0 && cfg->gemm_dtype <= 3, SAT_E_INVALID,
                  "dit_plan_create: gemm_dtype must be 0 (bf16), 1 (e4m3), 2 (fp32 verification) or 3 (fp16)");
    SAT_CHECK_ARG(cfg->gemm_dtype != 1 || cfg->embed_dim % 256 == 0, SAT_E_UNSUPPORTED, "dit_plan_create: gemm_dtype needs embed_dim %% 256 == 0");
    SAT_CHECK_ARG(cfg->gemm_dtype == 1 || cfg->fp8_families == 0, SAT_E_INVALID,
                  "dit_plan_create: fp8_families = 0x%x with gemm_dtype %d (a caller built against an older sat_dit_cfg layout?)", cfg->fp8_families, cfg->gemm_dtype);
    SAT_CHECK_ARG(cfg->cross_attention == 0 || cfg->cross_attention == 1, SAT_E_INVALID, "dit_plan_create: cross_attention must be 0 (fused where it applies) or 1 (two kernels)");
    SAT_CHECK_ARG(cfg->tile_policy == 0 || cfg->tile_policy == 22 || cfg->tile_policy == 80 || cfg->tile_policy == 81 || cfg->tile_policy == 82, SAT_E_INVALID,
                  "dit_plan_create: tile_policy must be 0 / 80 (default), 22, 81 or 82");
    const int fam = cfg->fp8_families ? cfg->fp8_families : SAT_FP8_DEFAULT;
    if (cfg->gemm_dtype == 1) {
        SAT_CHECK_ARG((fam & ~SAT_FP8_ALL) == 0, SAT_E_INVALID, "dit_plan_create: unknown bits in fp8_families 0x%x", fam);
        SAT_CHECK_ARG(!(fam & SAT_FP8_FF_OUT) || (fam & SAT_FP8_FF_IN), SAT_E_UNSUPPORTED,
                      "dit_plan_create: FF-out's MXFP8 operand is written by the e4m3 FF-in epilogue: SAT_FP8_FF_OUT needs SAT_FP8_FF_IN");
    }
    sat_dit_plan* p = new (std::nothrow) sat_dit_plan();
    SAT_CHECK_ARG(p, SAT_E_INVALID, "dit_plan_create: out of host memory");
    p->cfg = *cfg;
    p->kvh_cross = cfg->cond_token_dim > 0 ? cfg->cond_embed_dim / 64 : 0;
    // the fold lives in the bf16 pipelined GEMM tiles (K >= 192); adaLN modulates between LayerNorm and GEMM per sequence, the e4m3
    // path quantises the LayerNorm output per token: both keep the standalone kernels
    p->f16 = cfg->gemm_dtype == 3 ? 1 : 0;
    if (cfg->gemm_dtype == 1) {
        p->f8_qkv = fam & SAT_FP8_QKV; p->f8_cq = fam & SAT_FP8_CROSS_Q; p->f8_ff1 = fam & SAT_FP8_FF_IN; p->f8_ff2 = fam & SAT_FP8_FF_OUT;
        p->f8_o = fam & SAT_FP8_TO_OUT;
    }
    p->cross_fusion = cfg->cross_attention == 0;
    p->tile_bits = sat_tile_policy_bits(cfg->tile_policy);
    p->ln_fold = cfg->ln_fold != 0 && (cfg->gemm_dtype == 0 || cfg->gemm_dtype == 3) && !cfg->adaln && cfg->embed_dim >= 256;
    *out_plan = p;
    return 0;
}

extern "C" void sat_dit_plan_destroy(sat_dit_plan* p) {
    if (!p) return;
    if (p->arena) (void)hipFree(p->arena);
    if (p->ctx_buf) (void)hipFree(p->ctx_buf);
    if (p->dbg) (void)hipFree(p->dbg);
    for (hipEvent_t e : p->prof_ev) (void)hipEventDestroy(e);
    delete p;
}

extern "C" int sat_dit_plan_set_tensor(sat_dit_plan* p, const char* name, const float* data_dev, int64_t numel) {
    SAT_CHECK_ARG(p && name && data_dev && numel > 0, SAT_E_INVALID, "dit_plan_set_tensor: bad argument");
    p->tensors[name] = {data_dev, numel};
    return 0;
}

extern "C" int sat_dit_plan_finalize(sat_dit_plan* p, sat_stream_t stream) {
    SAT_CHECK_ARG(p, SAT_E_INVALID, "dit_plan_finalize: null plan");
    hipStream_t s = (hipStream_t)stream;
    const int D = p->cfg.embed_dim;
    auto it = p->tensors.find("transformer.layers.0.ff.ff.0.proj.weight");
    SAT_CHECK_ARG(it != p->tensors.end(), SAT_E_MISSING, "dit plan: tensor 'transformer.layers.0.ff.ff.0.proj.weight' was never set");
    SAT_CHECK_ARG(it->second.second % (2 * (int64_t)D) == 0, SAT_E_INVALID, "dit plan: FF weight size not divisible by 2*embed_dim");
    p->inner = (int)(it->second.second / (2 * (int64_t)D));
    SAT_CHECK_ARG(p->inner % 64 == 0, SAT_E_UNSUPPORTED, "dit plan: FF inner dim %d must be a multiple of 64", p->inner);
    SAT_CHECK_ARG(p->cfg.gemm_dtype != 1 || p->inner % 128 == 0, SAT_E_UNSUPPORTED, "dit plan: gemm_dtype needs an FF inner dim that is a multiple of 128");
    if (p->arena) {
        (void)hipFree(p->arena);
        p->arena = nullptr;
    }
    p->finalized = false;
    Arena dry;
    SAT_TRY(build(p, dry, s));
    SAT_HIP(hipMalloc((void**)&p->arena, dry.off));
    Arena real;
    real.base = p->arena;
    real.dry = false;
    SAT_TRY(build(p, real, s));
    SAT_HIP(hipStreamSynchronize(s));   // the caller may free its fp32 tensors once this returns
    p->tensors.clear();
    p->finalized = true;
    return 0;
}

extern "C" int sat_dit_workspace_bytes(const sat_dit_plan* p, int32_t bf, int32_t t_len, size_t* out_bytes) {
    SAT_CHECK_ARG(p && out_bytes && bf > 0 && t_len > 0, SAT_E_INVALID, "dit_workspace_bytes: bad argument");
    SAT_CHECK_ARG(p->finalized, SAT_E_STATE, "dit_workspace_bytes: plan not finalized");
    *out_bytes = carve(p, bf, t_len, nullptr).total;
    return 0;
}

extern "C" int sat_dit_prepare_context(sat_dit_plan* p, const float* cond, int32_t bf, int32_t lc, const float* global_cond,
                                       sat_stream_t stream) {
    SAT_CHECK_ARG(p && p->finalized, SAT_E_STATE, "dit_prepare_context: plan not finalized");
    hipStream_t s = (hipStream_t)stream;
    const sat_dit_cfg& c = p->cfg;
    const int D = c.embed_dim, Dc = c.cond_embed_dim, Dct = c.cond_token_dim, Dg = c.global_cond_dim;
    const bool cross = Dct > 0;
    SAT_CHECK_ARG(bf > 0, SAT_E_INVALID, "dit_prepare_context: bf must be positive");
    SAT_CHECK_ARG(!cross || (cond && lc > 0), SAT_E_INVALID, "dit_prepare_context: model has cross-attention but no cond given");
    SAT_CHECK_ARG(!(global_cond && Dg == 0), SAT_E_INVALID, "dit_prepare_context: model has no global conditioning");
    const int lcpad = cross ? (int)round_up(lc + 3, 64) : 0;
    const int R = bf * lc;
    // layout of the context buffer
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += (size_t)round_up((int64_t)bytes, 256);
        return o;
    };
    const size_t o_ge = take((size_t)bf * D * 4);
    const size_t o_gh = take((size_t)bf * D * 4);
    const size_t o_ch = cross ? take((size_t)R * Dc * 4) : 0;
    const size_t o_ce = cross ? take((size_t)R * Dc * 2) : 0;
    const bool f32 = c.gemm_dtype == 2;
    const size_t kv_elems = cross ? (size_t)c.depth * bf * p->kvh_cross * (f32 ? lc : lcpad) * 64 : 0;
    const size_t o_kc = take(kv_elems * (f32 ? 4 : 2));
    const size_t o_vc = take(kv_elems * (f32 ? 4 : 2));
    const size_t o_kv32 = (cross && f32) ? take((size_t)R * 2 * Dc * 4) : 0;
    const size_t o_ce32 = (cross && f32) ? take((size_t)R * Dc * 4) : 0;
    if (off > p->ctx_cap) {
        SAT_HIP(hipStreamSynchronize(s));
        if (p->ctx_buf) SAT_HIP(hipFree(p->ctx_buf));
        p->ctx_buf = nullptr;
        p->ctx_cap = 0;
        SAT_HIP(hipMalloc((void**)&p->ctx_buf, off));
        p->ctx_cap = off;
    }
    p->ge = (float*)(p->ctx_buf + o_ge);
    float* gh = (float*)(p->ctx_buf + o_gh);
    p->kc = (op_t*)(p->ctx_buf + o_kc);
    p->vct = (op_t*)(p->ctx_buf + o_vc);
    p->has_global = global_cond != nullptr;
    if (global_cond) {   // dit.py:154
        SAT_TRY(glue_small_linear(global_cond, Dg, p->ge0_w, nullptr, nullptr, 0, gh, D, bf, D, Dg, 1, 0, s));
        SAT_TRY(glue_small_linear(gh, D, p->ge2_w, nullptr, nullptr, 0, p->ge, D, bf, D, D, 0, 0, s));
    }
    if (cross && f32) {   // fp32 verification mode: fp32 context embedding, fp32 K / V [bf, kvh, lc, 64] per layer
        float* ch = (float*)(p->ctx_buf + o_ch);
        float* ce32 = (float*)(p->ctx_buf + o_ce32);
        float* kv32 = (float*)(p->ctx_buf + o_kv32);
        p->kc32 = (float*)(p->ctx_buf + o_kc);
        p->vc32 = (float*)(p->ctx_buf + o_vc);
        SAT_TRY(glue_small_linear(cond, Dct, p->ce0_w, nullptr, nullptr, 0, ch, Dc, R, Dc, Dct, 1, 0, s));
        SAT_TRY(glue_small_linear(ch, Dc, p->ce2_w, nullptr, nullptr, 0, ce32, Dc, R, Dc, Dc, 0, 0, s));
        const size_t per_layer = (size_t)bf * p->kvh_cross * lc * 64;
        for (int l = 0; l < c.depth; ++l) {
            SAT_TRY(sat_launch_gemm_f32(ce32, (const float*)p->layers[l].w_ckv, nullptr, kv32, R, 2 * Dc, Dc, 2 * Dc, 0, nullptr, 1, 0, s));
            SAT_TRY(sat_launch_split_heads_f32(kv32, p->kc32 + l * per_layer, p->vc32 + l * per_layer, nullptr, R, lc, 2, p->kvh_cross, 0, nullptr,
                                               nullptr, s));
        }
    } else if (cross) {   // dit.py:150 then per-layer to_kv (transformer.py:420-427)
        float* ch = (float*)(p->ctx_buf + o_ch);
        op_t* ce = (op_t*)(p->ctx_buf + o_ce);
        SAT_TRY(glue_small_linear(cond, Dct, p->ce0_w, nullptr, nullptr, 0, ch, Dc, R, Dc, Dct, 1, 0, s));
        SAT_TRY(glue_small_linear(ch, Dc, p->ce2_w, nullptr, nullptr, 0, ce, Dc, R, Dc, Dc, 0, p->f16 ? 2 : 1, s));
        SAT_HIP(hipMemsetAsync(p->kc, 0, kv_elems * 2, s));
        SAT_HIP(hipMemsetAsync(p->vct, 0, kv_elems * 2, s));
        const size_t per_layer = (size_t)bf * p->kvh_cross * lcpad * 64;
        for (int l = 0; l < c.depth; ++l) {
            GemmArgs g{};
            g.f16 = p->f16;
            g.A = ce; g.W = p->layers[l].w_ckv; g.M = R; g.N = 2 * Dc; g.K = Dc;
            g.heads.out[0] = p->kc + l * per_layer; g.heads.out[1] = p->vct + l * per_layer;
            g.heads.kind[0] = 4; g.heads.kind[1] = 1 | 4; g.heads.parts = 2; g.heads.heads = p->kvh_cross;
            g.heads.S = lc; g.heads.Spad = lcpad;
            g.variant = 1;
            SAT_TRY(sat_launch_gemm(EPI_HEADS, g, s));
        }
    }
    p->ctx_bf = bf;
    p->ctx_null_from = -1;
    p->ctx_lc = cross ? lc : 0;
    p->ctx_lcpad = lcpad;
    return 0;
}

extern "C" int sat_dit_set_null_context_from(sat_dit_plan* p, int32_t first_null_seq) {
    SAT_CHECK_ARG(p && p->finalized, SAT_E_STATE, "dit_set_null_context_from: plan not finalized");
    SAT_CHECK_ARG(first_null_seq >= -1 && first_null_seq <= p->ctx_bf, SAT_E_INVALID, "dit_set_null_context_from: %d not in [-1, %d]",
                  first_null_seq, p->ctx_bf);
    p->ctx_null_from = first_null_seq;
    return 0;
}

extern "C" int sat_dit_forward(sat_dit_plan* p, const float* x_dev, const float* t_dev, float* out_dev, int32_t bf, int32_t t_len,
                               void* ws, size_t ws_bytes, sat_stream_t stream) {
    SAT_CHECK_ARG(t_dev, SAT_E_INVALID, "dit_forward: t_dev is null");
    return run_forward(p, x_dev, bf, 1.0f, t_dev, 0.f, out_dev, bf, t_len, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int sat_dit_denoise_cfg(sat_dit_plan* p, const float* x_dev, float sigma, float cfg_scale, float scale_phi,
                                   float* denoised_dev, int32_t b, int32_t t_len, void* ws, size_t ws_bytes, sat_stream_t stream) {
    SAT_CHECK_ARG(p && p->finalized, SAT_E_STATE, "dit_denoise_cfg: plan not finalized");
    SAT_CHECK_ARG(x_dev && denoised_dev && b > 0 && t_len > 0 && ws, SAT_E_INVALID, "dit_denoise_cfg: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    // models/dit.py:270: CFG only when cfg_scale != 1 and there is cross-attention conditioning
    const int use_cfg = (cfg_scale != 1.0f && p->cfg.cond_token_dim > 0) ? 1 : 0;
    const int bf = use_cfg ? 2 * b : b;
    // k_diffusion.external.VDenoiser, sigma_data = 1
    const double sg = (double)sigma;
    const float c_skip = (float)(1.0 / (sg * sg + 1.0));
    const float c_out = (float)(-sg / sqrt(sg * sg + 1.0));
    const float c_in = (float)(1.0 / sqrt(sg * sg + 1.0));
    const float t = (float)(atan(sg) / M_PI * 2.0);
    size_t need = 0;
    SAT_TRY(sat_dit_workspace_bytes(p, bf, t_len, &need));
    SAT_CHECK_ARG(ws_bytes >= need, SAT_E_WORKSPACE, "dit_denoise_cfg: workspace %zu < required %zu", ws_bytes, need);
    Workspace w = carve(p, bf, t_len, (char*)ws);
    SAT_TRY(run_forward(p, x_dev, b, c_in, nullptr, t, w.mo, bf, t_len, ws, ws_bytes, s));
    return glue_cfg_denoise(w.mo, x_dev, denoised_dev, b, p->cfg.io_channels, t_len, use_cfg, cfg_scale, scale_phi, c_out, c_skip, s);
}

extern "C" int sat_dit_profile(sat_dit_plan* p, int32_t enable) {
    SAT_CHECK_ARG(p, SAT_E_INVALID, "dit_profile: null plan");
    p->prof_on = enable != 0;
    p->prof_n = 0;
    return 0;
}

extern "C" int sat_dit_profile_read(sat_dit_plan* p, double* total_ms, int32_t* launches, int64_t* m, int64_t* n, int64_t* k) {
    SAT_CHECK_ARG(p && total_ms && launches, SAT_E_INVALID, "dit_profile_read: null argument");
    double tot = 0.0;
    for (int i = 0; i < p->prof_n; ++i) {
        SAT_HIP(hipEventSynchronize(p->prof_ev[2 * i + 1]));
        float ms = 0.f;
        SAT_HIP(hipEventElapsedTime(&ms, p->prof_ev[2 * i], p->prof_ev[2 * i + 1]));
        tot += ms;
    }
    *total_ms = tot;
    *launches = p->prof_n;
    if (m) *m = p->prof_m;
    if (n) *n = p->prof_nn;
    if (k) *k = p->prof_k;
    return 0;
}

extern "C" int sat_dit_debug(sat_dit_plan* p, int32_t enable) {
    SAT_CHECK_ARG(p, SAT_E_INVALID, "dit_debug: null plan");
    if (enable && !p->dbg) {
        SAT_CHECK_ARG(p->cfg.gemm_dtype != 2, SAT_E_UNSUPPORTED, "dit_debug: the fp32 verification mode keeps no 16-bit image of the residual stream");
        SAT_HIP(hipMalloc((void**)&p->dbg, (size_t)p->cfg.depth * 3 * 4 * sizeof(float)));
        SAT_HIP(hipMemset(p->dbg, 0, (size_t)p->cfg.depth * 3 * 4 * sizeof(float)));
    } else if (!enable && p->dbg) {
        SAT_HIP(hipDeviceSynchronize());
        SAT_HIP(hipFree(p->dbg));
        p->dbg = nullptr;
    }
    return 0;
}

extern "C" int sat_dit_debug_read(sat_dit_plan* p, float* out_host, int32_t capacity_floats, sat_stream_t stream) {
    SAT_CHECK_ARG(p && out_host, SAT_E_INVALID, "dit_debug_read: null argument");
    SAT_CHECK_ARG(p->dbg, SAT_E_STATE, "dit_debug_read: diagnostics are not enabled (sat_dit_debug)");
    const int n = p->cfg.depth * 3 * 4;
    SAT_CHECK_ARG(capacity_floats >= n, SAT_E_INVALID, "dit_debug_read: room for %d floats, need %d", capacity_floats, n);
    SAT_HIP(hipMemcpyAsync(out_host, p->dbg, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
    SAT_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

extern "C" int sat_cfg_combine(const float* model_out_dev, float* out_dev, int32_t b, int32_t c, int32_t t, float cfg_scale,
                               float scale_phi, sat_stream_t stream) {
    SAT_CHECK_ARG(model_out_dev && out_dev && b > 0 && c > 1 && t > 0, SAT_E_INVALID, "cfg_combine: bad arguments");
    // denoise form with c_out = 1, c_skip = 0 (x is not read when c_skip == 0, but must be a valid pointer)
    return glue_cfg_denoise(model_out_dev, model_out_dev, out_dev, b, c, t, 1, cfg_scale, scale_phi, 1.0f, 0.0f, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------ unit-level entry points
static int layernorm_bf16_impl(int f16, const float* x, const float* gamma, const float* beta, void* y, int32_t m, int32_t d,
                                  sat_stream_t stream) {
    return sat_launch_layernorm(x, gamma, beta, (op_t*)y, m, d, (hipStream_t)stream, f16);
}
extern "C" int sat_layernorm_bf16(const float* x, const float* gamma, const float* beta, void* y, int32_t m, int32_t d,
                                  sat_stream_t stream) {
    return layernorm_bf16_impl(0, x, gamma, beta, y, m, d, stream);
}
extern "C" int sat_layernorm_f16(const float* x, const float* gamma, const float* beta, void* y, int32_t m, int32_t d,
                                  sat_stream_t stream) {
    return layernorm_bf16_impl(1, x, gamma, beta, y, m, d, stream);
}

static int cast_bf16_impl(int f16, const float* x, void* y, int64_t n, sat_stream_t stream) {
    return sat_launch_cast_bf16(x, (op_t*)y, n, (hipStream_t)stream, f16);
}
extern "C" int sat_cast_bf16(const float* x, void* y, int64_t n, sat_stream_t stream) {
    return cast_bf16_impl(0, x, y, n, stream);
}
extern "C" int sat_cast_f16(const float* x, void* y, int64_t n, sat_stream_t stream) {
    return cast_bf16_impl(1, x, y, n, stream);
}

static int gemm_bf16_f32_impl(int f16, const void* a, const void* w, const float* bias, float* c, int32_t m, int32_t n, int32_t k,
                              int32_t accumulate, int32_t variant, void* ws, size_t ws_bytes, sat_stream_t stream) {
    SAT_CHECK_ARG(c, SAT_E_INVALID, "gemm: null output");
    GemmArgs g{};
    g.f16 = f16;
    g.A = (const op_t*)a; g.W = (const op_t*)w; g.bias = bias; g.M = m; g.N = n; g.K = k;
    g.C = c; g.ldc = n; g.accumulate = accumulate; g.variant = variant;
    g.slab = (float*)ws; g.slab_bytes = ws ? ws_bytes : 0;
    return sat_launch_gemm(EPI_F32, g, (hipStream_t)stream);
}
extern "C" int sat_gemm_bf16_f32(const void* a, const void* w, const float* bias, float* c, int32_t m, int32_t n, int32_t k,
                                 int32_t accumulate, int32_t variant, sat_stream_t stream) {
    return gemm_bf16_f32_impl(0, a, w, bias, c, m, n, k, accumulate, variant, nullptr, 0, stream);
}
extern "C" int sat_gemm_f16_f32(const void* a, const void* w, const float* bias, float* c, int32_t m, int32_t n, int32_t k,
                                int32_t accumulate, int32_t variant, sat_stream_t stream) {
    return gemm_bf16_f32_impl(1, a, w, bias, c, m, n, k, accumulate, variant, nullptr, 0, stream);
}
extern "C" int sat_gemm_bf16_f32_ws(const void* a, const void* w, const float* bias, float* c, int32_t m, int32_t n, int32_t k,
                                    int32_t accumulate, int32_t variant, void* ws, size_t ws_bytes, sat_stream_t stream) {
    return gemm_bf16_f32_impl(0, a, w, bias, c, m, n, k, accumulate, variant, ws, ws_bytes, stream);
}
extern "C" int sat_gemm_f16_f32_ws(const void* a, const void* w, const float* bias, float* c, int32_t m, int32_t n, int32_t k,
                                   int32_t accumulate, int32_t variant, void* ws, size_t ws_bytes, sat_stream_t stream) {
    return gemm_bf16_f32_impl(1, a, w, bias, c, m, n, k, accumulate, variant, ws, ws_bytes, stream);
}
extern "C" int sat_gemm_f32_workspace_bytes(int32_t m, int32_t n, int32_t k, int32_t variant, size_t* out_bytes) {
    SAT_CHECK_ARG(out_bytes && m > 0 && n > 0 && k > 0, SAT_E_INVALID, "gemm_f32_workspace_bytes: bad argument");
    const int cus = sat_device_cus();
    SAT_CHECK_ARG(cus > 0, SAT_E_INVALID, "gemm_f32_workspace_bytes: no device");
    // forced K-split (variant bit 16, tests / measurements): one slab per workgroup; otherwise what the automatic schedule would use
    *out_bytes = (variant & 0x10000) ? (size_t)cus * 65536 * sizeof(float) : sat_gemm_ph8_slab_bytes(EPI_F32, m, n, k);
    return 0;
}

static int gemm_swiglu_bf16_impl(int f16, const void* a, const float* w_f32, const float* bias_f32, void* wpack, float* bpack,
                                    void* h, int32_t m, int32_t n, int32_t k, int32_t variant, sat_stream_t stream) {
    SAT_CHECK_ARG(w_f32 && wpack && bpack && h, SAT_E_INVALID, "gemm_swiglu: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (!(variant & 0x4000)) {     // bit 14: wpack / bpack already hold the packed operands of a previous call (benchmarks)
        SAT_TRY(sat_launch_pack_rows_bf16(w_f32, (op_t*)wpack, n, k, 1, s, f16));
        if (bias_f32) SAT_TRY(sat_launch_pack_bias(bias_f32, bpack, n, 1, s));
    }
    GemmArgs g{};
    g.f16 = f16;
    g.A = (const op_t*)a; g.W = (const op_t*)wpack; g.bias = bias_f32 ? bpack : nullptr; g.M = m; g.N = n; g.K = k;
    g.H = (op_t*)h; g.variant = variant;
    return sat_launch_gemm(EPI_SWIGLU, g, s);
}
extern "C" int sat_gemm_swiglu_bf16(const void* a, const float* w_f32, const float* bias_f32, void* wpack, float* bpack,
                                    void* h, int32_t m, int32_t n, int32_t k, int32_t variant, sat_stream_t stream) {
    return gemm_swiglu_bf16_impl(0, a, w_f32, bias_f32, wpack, bpack, h, m, n, k, variant, stream);
}
extern "C" int sat_gemm_swiglu_f16(const void* a, const float* w_f32, const float* bias_f32, void* wpack, float* bpack,
                                    void* h, int32_t m, int32_t n, int32_t k, int32_t variant, sat_stream_t stream) {
    return gemm_swiglu_bf16_impl(1, a, w_f32, bias_f32, wpack, bpack, h, m, n, k, variant, stream);
}

static int attention_bf16_impl(int f16, const void* q, const void* k, const void* vt, void* out, int32_t b, int32_t h, int32_t kvh,
                                  int32_t sq, int32_t sk, int32_t sq_pad, int32_t sk_pad, sat_stream_t stream) {
    return sat_launch_attention((const op_t*)q, (const op_t*)k, (const op_t*)vt, (op_t*)out, b, h, kvh, sq, sk, sq_pad,
                                sk_pad, (hipStream_t)stream, nullptr, SAT_ATTN_QSCALE, f16);
}
extern "C" int sat_attention_bf16(const void* q, const void* k, const void* vt, void* out, int32_t b, int32_t h, int32_t kvh,
                                  int32_t sq, int32_t sk, int32_t sq_pad, int32_t sk_pad, sat_stream_t stream) {
    return attention_bf16_impl(0, q, k, vt, out, b, h, kvh, sq, sk, sq_pad, sk_pad, stream);
}
extern "C" int sat_attention_f16(const void* q, const void* k, const void* vt, void* out, int32_t b, int32_t h, int32_t kvh,
                                  int32_t sq, int32_t sk, int32_t sq_pad, int32_t sk_pad, sat_stream_t stream) {
    return attention_bf16_impl(1, q, k, vt, out, b, h, kvh, sq, sk, sq_pad, sk_pad, stream);
}

// to_q projection + cross-attention in ONE launch (what the plan runs per layer at one prompt): out [b*s, d] = attention(a wq^T, k, v)
static int cross_attention_fused_bf16_impl(int f16, const void* a, const void* wq, const void* k, const void* vt, void* out, int32_t b, int32_t s_len,
                                              int32_t d, int32_t kvh, int32_t sk, int32_t sk_pad, sat_stream_t stream) {
    SAT_CHECK_ARG(a && wq && k && vt && out && b > 0 && s_len > 0 && d > 0 && d % 128 == 0 && kvh > 0, SAT_E_INVALID, "cross_attention_fused: bad argument");
    GemmArgs g{};
    g.f16 = f16;
    g.A = (const op_t*)a; g.W = (const op_t*)wq; g.M = b * s_len; g.N = d; g.K = d;
    g.heads.kind[0] = 8; g.heads.qscale = SAT_ATTN_QSCALE; g.heads.parts = 1; g.heads.heads = d / 64; g.heads.S = s_len; g.heads.Spad = s_len;
    g.heads.xa_k = (const op_t*)k; g.heads.xa_vt = (const op_t*)vt; g.heads.xa_out = (op_t*)out;
    g.heads.xa_kvh = kvh; g.heads.xa_sk = sk; g.heads.xa_sk_pad = sk_pad;
    return sat_launch_gemm(EPI_HEADS, g, (hipStream_t)stream);
}
extern "C" int sat_cross_attention_fused_bf16(const void* a, const void* wq, const void* k, const void* vt, void* out, int32_t b, int32_t s_len,
                                              int32_t d, int32_t kvh, int32_t sk, int32_t sk_pad, sat_stream_t stream) {
    return cross_attention_fused_bf16_impl(0, a, wq, k, vt, out, b, s_len, d, kvh, sk, sk_pad, stream);
}
extern "C" int sat_cross_attention_fused_f16(const void* a, const void* wq, const void* k, const void* vt, void* out, int32_t b, int32_t s_len,
                                              int32_t d, int32_t kvh, int32_t sk, int32_t sk_pad, sat_stream_t stream) {
    return cross_attention_fused_bf16_impl(1, a, wq, k, vt, out, b, s_len, d, kvh, sk, sk_pad, stream);
}

// The layout the DiT plan runs: Q pre-scaled by 1/sqrt(64) * log2(e) by its producer (the QKV / to_q GEMM epilogue)
static int attention_prescaled_bf16_impl(int f16, const void* q, const void* k, const void* vt, void* out, int32_t b, int32_t h, int32_t kvh,
                                            int32_t sq, int32_t sk, int32_t sq_pad, int32_t sk_pad, sat_stream_t stream) {
    return sat_launch_attention((const op_t*)q, (const op_t*)k, (const op_t*)vt, (op_t*)out, b, h, kvh, sq, sk, sq_pad,
                                sk_pad, (hipStream_t)stream, nullptr, 1.0f, f16);
}
extern "C" int sat_attention_prescaled_bf16(const void* q, const void* k, const void* vt, void* out, int32_t b, int32_t h, int32_t kvh,
                                            int32_t sq, int32_t sk, int32_t sq_pad, int32_t sk_pad, sat_stream_t stream) {
    return attention_prescaled_bf16_impl(0, q, k, vt, out, b, h, kvh, sq, sk, sq_pad, sk_pad, stream);
}
extern "C" int sat_attention_prescaled_f16(const void* q, const void* k, const void* vt, void* out, int32_t b, int32_t h, int32_t kvh,
                                            int32_t sq, int32_t sk, int32_t sq_pad, int32_t sk_pad, sat_stream_t stream) {
    return attention_prescaled_bf16_impl(1, q, k, vt, out, b, h, kvh, sq, sk, sq_pad, sk_pad, stream);
}

static int qkv_rope_bf16_impl(int f16, const void* a, const void* w, const float* inv_freq, void* q, void* k, void* vt,
                                 float* rope_scratch, int32_t b, int32_t s_len, int32_t s_pad, int32_t d, int32_t variant,
                                 sat_stream_t stream) {
    SAT_CHECK_ARG(a && w && inv_freq && q && k && vt && rope_scratch, SAT_E_INVALID, "qkv_rope: null pointer");
    SAT_CHECK_ARG(d % 64 == 0 && s_pad >= s_len + 3 && s_pad % 128 == 0, SAT_E_INVALID, "qkv_rope: bad dims (s_pad >= s + 3, %% 128)");
    hipStream_t s = (hipStream_t)stream;
    const int H = d / 64;
    const size_t bytes = (size_t)b * H * s_pad * 64 * 2;
    SAT_HIP(hipMemsetAsync(q, 0, bytes, s));
    SAT_HIP(hipMemsetAsync(k, 0, bytes, s));
    SAT_HIP(hipMemsetAsync(vt, 0, bytes, s));
    float* cs = rope_scratch;
    float* sn = rope_scratch + (size_t)s_len * 16;
    SAT_TRY(sat_launch_rope_table(inv_freq, cs, sn, s_len, s));
    GemmArgs g{};
    g.f16 = f16;
    g.A = (const op_t*)a; g.W = (const op_t*)w; g.M = b * s_len; g.N = 3 * d; g.K = d; g.variant = variant;
    g.heads.out[0] = (op_t*)q; g.heads.out[1] = (op_t*)k; g.heads.out[2] = (op_t*)vt;
    g.heads.kind[0] = 2; g.heads.kind[1] = 2 | 4; g.heads.kind[2] = 1 | 4;
    g.heads.parts = 3; g.heads.heads = H; g.heads.S = s_len; g.heads.Spad = s_pad;
    g.heads.rope_cos = cs; g.heads.rope_sin = sn;
    return sat_launch_gemm(EPI_HEADS, g, s);
}
extern "C" int sat_qkv_rope_bf16(const void* a, const void* w, const float* inv_freq, void* q, void* k, void* vt,
                                 float* rope_scratch, int32_t b, int32_t s_len, int32_t s_pad, int32_t d, int32_t variant,
                                 sat_stream_t stream) {
    return qkv_rope_bf16_impl(0, a, w, inv_freq, q, k, vt, rope_scratch, b, s_len, s_pad, d, variant, stream);
}
extern "C" int sat_qkv_rope_f16(const void* a, const void* w, const float* inv_freq, void* q, void* k, void* vt,
                                 float* rope_scratch, int32_t b, int32_t s_len, int32_t s_pad, int32_t d, int32_t variant,
                                 sat_stream_t stream) {
    return qkv_rope_bf16_impl(1, a, w, inv_freq, q, k, vt, rope_scratch, b, s_len, s_pad, d, variant, stream);
}

// ---- LayerNorm folded into the neighbouring GEMMs (sat_dit_cfg.ln_fold), one entry per role
static int gemm_resid_ln_bf16_impl(int f16, const void* a, const void* w, const float* bias, float* c, void* xb, float* ln_part, int32_t m,
                                   int32_t n, int32_t k, int32_t variant, void* ws, size_t ws_bytes, sat_stream_t stream) {
    SAT_CHECK_ARG(c && xb && ln_part, SAT_E_INVALID, "gemm_resid_ln: null output");
    GemmArgs g{};
    g.f16 = f16;
    g.A = (const op_t*)a; g.W = (const op_t*)w; g.bias = bias; g.M = m; g.N = n; g.K = k;
    g.C = c; g.ldc = n; g.accumulate = 1; g.variant = variant; g.xb = (op_t*)xb; g.ln_part_out = ln_part;
    g.slab = (float*)ws; g.slab_bytes = ws ? ws_bytes : 0;
    return sat_launch_gemm(EPI_RESID, g, (hipStream_t)stream);
}
extern "C" int sat_gemm_resid_ln_bf16(const void* a, const void* w, const float* bias, float* c, void* xb, float* ln_part, int32_t m,
                                      int32_t n, int32_t k, int32_t variant, sat_stream_t stream) {
    return gemm_resid_ln_bf16_impl(0, a, w, bias, c, xb, ln_part, m, n, k, variant, nullptr, 0, stream);
}
extern "C" int sat_gemm_resid_ln_f16(const void* a, const void* w, const float* bias, float* c, void* xb, float* ln_part, int32_t m,
                                     int32_t n, int32_t k, int32_t variant, sat_stream_t stream) {
    return gemm_resid_ln_bf16_impl(1, a, w, bias, c, xb, ln_part, m, n, k, variant, nullptr, 0, stream);
}
extern "C" int sat_gemm_resid_ln_bf16_ws(const void* a, const void* w, const float* bias, float* c, void* xb, float* ln_part, int32_t m,
                                         int32_t n, int32_t k, int32_t variant, void* ws, size_t ws_bytes, sat_stream_t stream) {
    return gemm_resid_ln_bf16_impl(0, a, w, bias, c, xb, ln_part, m, n, k, variant, ws, ws_bytes, stream);
}
extern "C" int sat_gemm_resid_ln_f16_ws(const void* a, const void* w, const float* bias, float* c, void* xb, float* ln_part, int32_t m,
                                        int32_t n, int32_t k, int32_t variant, void* ws, size_t ws_bytes, sat_stream_t stream) {
    return gemm_resid_ln_bf16_impl(1, a, w, bias, c, xb, ln_part, m, n, k, variant, ws, ws_bytes, stream);
}

static int gemm_swiglu_ln_bf16_impl(int f16, const void* xb, const float* ln_part, const float* w_f32, const float* gamma, const float* beta,
                                       const float* bias_f32, void* wpack, float* c12, void* h, int32_t m, int32_t n, int32_t k,
                                       int32_t variant, sat_stream_t stream) {
    SAT_CHECK_ARG(xb && ln_part && w_f32 && gamma && beta && wpack && c12 && h, SAT_E_INVALID, "gemm_swiglu_ln: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (!(variant & 0x4000))       // bit 14: wpack / c12 already hold the packed operands of a previous call (benchmarks)
        SAT_TRY(sat_launch_pack_rows_ln(w_f32, gamma, beta, bias_f32, (op_t*)wpack, c12, c12 + n, n, k, 1, s, f16));
    GemmArgs g{};
    g.f16 = f16;
    g.A = (const op_t*)xb; g.W = (const op_t*)wpack; g.M = m; g.N = n; g.K = k; g.H = (op_t*)h; g.variant = variant & ~0x4000;
    g.ln_part = ln_part; g.ln_c1 = c12; g.ln_c2 = c12 + n; g.ln_eps = 1e-5f;
    return sat_launch_gemm(EPI_SWIGLU, g, s);
}
extern "C" int sat_gemm_swiglu_ln_bf16(const void* xb, const float* ln_part, const float* w_f32, const float* gamma, const float* beta,
                                       const float* bias_f32, void* wpack, float* c12, void* h, int32_t m, int32_t n, int32_t k,
                                       int32_t variant, sat_stream_t stream) {
    return gemm_swiglu_ln_bf16_impl(0, xb, ln_part, w_f32, gamma, beta, bias_f32, wpack, c12, h, m, n, k, variant, stream);
}
extern "C" int sat_gemm_swiglu_ln_f16(const void* xb, const float* ln_part, const float* w_f32, const float* gamma, const float* beta,
                                       const float* bias_f32, void* wpack, float* c12, void* h, int32_t m, int32_t n, int32_t k,
                                       int32_t variant, sat_stream_t stream) {
    return gemm_swiglu_ln_bf16_impl(1, xb, ln_part, w_f32, gamma, beta, bias_f32, wpack, c12, h, m, n, k, variant, stream);
}

static int qkv_rope_ln_bf16_impl(int f16, const void* xb, const float* ln_part, const float* w_f32, const float* gamma, const float* beta,
                                    void* wpack, float* c12, const float* inv_freq, void* q, void* k, void* vt, float* rope_scratch,
                                    int32_t b, int32_t s_len, int32_t s_pad, int32_t d, int32_t variant, sat_stream_t stream) {
    SAT_CHECK_ARG(xb && ln_part && w_f32 && gamma && beta && wpack && c12 && inv_freq && q && k && vt && rope_scratch, SAT_E_INVALID,
                  "qkv_rope_ln: null pointer");
    SAT_CHECK_ARG(d % 64 == 0 && s_pad >= s_len + 3 && s_pad % 128 == 0, SAT_E_INVALID, "qkv_rope_ln: bad dims (s_pad >= s + 3, %% 128)");
    hipStream_t s = (hipStream_t)stream;
    const int H = d / 64;
    const size_t bytes = (size_t)b * H * s_pad * 64 * 2;
    float* cs = rope_scratch;
    float* sn = rope_scratch + (size_t)s_len * 16;
    if (!(variant & 0x4000)) {     // bit 14: pads, tables and packed operands are those of a previous call (benchmarks)
        SAT_HIP(hipMemsetAsync(q, 0, bytes, s));
        SAT_HIP(hipMemsetAsync(k, 0, bytes, s));
        SAT_HIP(hipMemsetAsync(vt, 0, bytes, s));
        SAT_TRY(sat_launch_rope_table(inv_freq, cs, sn, s_len, s));
        SAT_TRY(sat_launch_pack_rows_ln(w_f32, gamma, beta, nullptr, (op_t*)wpack, c12, c12 + 3 * d, 3 * d, d, 0, s, f16));
    }
    GemmArgs g{};
    g.f16 = f16;
    g.A = (const op_t*)xb; g.W = (const op_t*)wpack; g.M = b * s_len; g.N = 3 * d; g.K = d; g.variant = variant & ~0x4000;
    g.heads.out[0] = (op_t*)q; g.heads.out[1] = (op_t*)k; g.heads.out[2] = (op_t*)vt;
    g.heads.kind[0] = 2; g.heads.kind[1] = 2 | 4; g.heads.kind[2] = 1 | 4;
    g.heads.parts = 3; g.heads.heads = H; g.heads.S = s_len; g.heads.Spad = s_pad;
    g.heads.rope_cos = cs; g.heads.rope_sin = sn;
    g.ln_part = ln_part; g.ln_c1 = c12; g.ln_c2 = c12 + 3 * d; g.ln_eps = 1e-5f;
    return sat_launch_gemm(EPI_HEADS, g, s);
}
extern "C" int sat_qkv_rope_ln_bf16(const void* xb, const float* ln_part, const float* w_f32, const float* gamma, const float* beta,
                                    void* wpack, float* c12, const float* inv_freq, void* q, void* k, void* vt, float* rope_scratch,
                                    int32_t b, int32_t s_len, int32_t s_pad, int32_t d, int32_t variant, sat_stream_t stream) {
    return qkv_rope_ln_bf16_impl(0, xb, ln_part, w_f32, gamma, beta, wpack, c12, inv_freq, q, k, vt, rope_scratch, b, s_len, s_pad, d, variant, stream);
}
extern "C" int sat_qkv_rope_ln_f16(const void* xb, const float* ln_part, const float* w_f32, const float* gamma, const float* beta,
                                    void* wpack, float* c12, const float* inv_freq, void* q, void* k, void* vt, float* rope_scratch,
                                    int32_t b, int32_t s_len, int32_t s_pad, int32_t d, int32_t variant, sat_stream_t stream) {
    return qkv_rope_ln_bf16_impl(1, xb, ln_part, w_f32, gamma, beta, wpack, c12, inv_freq, q, k, vt, rope_scratch, b, s_len, s_pad, d, variant, stream);
}

extern "C" int sat_quant_rows_fp8(const float* x, void* out8, float* row_scale, int32_t rows, int32_t k, sat_stream_t stream) {
    return sat_launch_quant_rows_fp8(x, out8, row_scale, rows, k, 0, (hipStream_t)stream);
}

extern "C" int sat_layernorm_fp8(const float* x, const float* gamma, const float* beta, void* y8, float* row_scale, int32_t m,
                                 int32_t d, sat_stream_t stream) {
    return sat_launch_layernorm_fp8(x, gamma, beta, y8, row_scale, m, d, nullptr, nullptr, 1, 0, (hipStream_t)stream);
}

extern "C" int sat_gemm_fp8_f32(const void* a8, const float* a_scale, const void* w8, const float* w_scale, const float* bias,
                                float* c, int32_t m, int32_t n, int32_t k, int32_t accumulate, int32_t variant, sat_stream_t stream) {
    SAT_CHECK_ARG(a8 && w8 && a_scale && w_scale && c, SAT_E_INVALID, "gemm_fp8: null pointer");
    GemmArgs g{};
    g.A = (const op_t*)a8; g.W = (const op_t*)w8; g.bias = bias; g.M = m; g.N = n; g.K = k; g.variant = variant & ~256;
    g.C = c; g.ldc = n; g.accumulate = accumulate; g.a_scale = a_scale; g.w_scale = w_scale;
    g.fp8 = (variant & 256) ? 1 : 2;      // bit 8 of variant: the plain 32x32x16 fp8 MFMA instead of the 2x-rate scaled 32x32x64
    return sat_launch_gemm(EPI_F32, g, (hipStream_t)stream);
}

extern "C" int sat_quant_mx_rows_fp8(const float* x, void* out8, void* scales, int32_t rows, int32_t k, sat_stream_t stream) {
    return sat_launch_quant_mx_rows(x, out8, scales, rows, k, (hipStream_t)stream);
}

extern "C" int sat_gemm_mxfp8_f32(const void* a8, const void* a_scales, const void* w8, const float* w_scale, const float* bias,
                                  float* c, int32_t m, int32_t n, int32_t k, int32_t accumulate, int32_t variant, sat_stream_t stream) {
    SAT_CHECK_ARG(a8 && w8 && a_scales && w_scale && c, SAT_E_INVALID, "gemm_mxfp8: null pointer");
    SAT_CHECK_ARG(((uintptr_t)a_scales & 3) == 0, SAT_E_INVALID, "gemm_mxfp8: the scale array must be 4-byte aligned");
    GemmArgs g{};
    g.A = (const op_t*)a8; g.W = (const op_t*)w8; g.bias = bias; g.M = m; g.N = n; g.K = k; g.variant = variant & 0xff;
    g.C = c; g.ldc = n; g.accumulate = accumulate; g.a_bscale = (const unsigned*)a_scales; g.w_scale = w_scale; g.fp8 = 3;
    return sat_launch_gemm(EPI_F32, g, (hipStream_t)stream);
}
