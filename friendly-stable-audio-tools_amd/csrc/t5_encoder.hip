// T5 encoder stack on the device: the text front-end of the conditioner.
// Replaces the `transformers.T5EncoderModel` call inside the reference's T5Conditioner.forward
// (models/conditioners.py:261-346: tokenizer -> model(input_ids, attention_mask)["last_hidden_state"] -> proj_out -> * mask).
// The algorithm is that of transformers' modeling_t5.py (a third-party dependency of the reference, not vendored in it):
//   T5Stack (encoder): embed_tokens -> N x [ x += SelfAttention(T5LayerNorm(x)) ; x += DenseReluDense(T5LayerNorm(x)) ] -> T5LayerNorm
//   T5LayerNorm   = x * rsqrt(mean(x^2) + eps) * weight                (no mean subtraction, no bias)
//   T5Attention   = softmax(q k^T + position_bias + mask) v, NO 1/sqrt(d) scaling; position_bias = relative_attention_bias of
//                   block 0 gathered through _relative_position_bucket(key - query, bidirectional), shared by all blocks;
//                   mask = (1 - attention_mask) * finfo(float32).min added to the scores
//   DenseReluDense = wo(relu(wi(x)))            ("relu": t5-*)    |  wo(gelu_new(wi_0(x)) * wi_1(x))   ("gated-gelu": flan-t5-*)
// Runs once per generation on B x 128 tokens (a few GFLOP), so everything is fp32 on the exact fp32 MFMA GEMM of f32_ref.hip: the
// reference runs the encoder under fp16 autocast; fp32 here is the more accurate of the two and needs no second set of kernels.
#include <math.h>

#include <map>
#include <string>
#include <vector>

#include "sat_common.h"

namespace {

// rows of the embedding table; ids outside [0, vocab) are clamped (the tokenizer never produces them)
__global__ __launch_bounds__(256) void t5_embed_kernel(const int* __restrict__ ids, const float* __restrict__ emb, float* __restrict__ out,
                                                       int rows, int D, int vocab) {
    const int row = blockIdx.x;
    int id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float4* src = reinterpret_cast<const float4*>(emb + (size_t)id * D);
    float4* dst = reinterpret_cast<float4*>(out + (size_t)row * D);
    for (int i = threadIdx.x; i < D / 4; i += 256) dst[i] = src[i];
}

// one wave per row
__global__ __launch_bounds__(256) void t5_rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int M,
                                                         int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
    float sq = 0.f;
    for (int i = lane; i < D / 4; i += 64) {
        const float4 v = xr[i];
        sq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    sq = wave_sum(sq);
    const float r = rsqrtf(sq / (float)D + eps);
    float4* yr = reinterpret_cast<float4*>(y + (size_t)row * D);
    for (int i = lane; i < D / 4; i += 64) {
        const float4 v = xr[i];
        const float4 g = reinterpret_cast<const float4*>(w)[i];
        yr[i] = make_float4(v.x * r * g.x, v.y * r * g.y, v.z * r * g.z, v.w * r * g.w);
    }
}

// pb[h][delta + L - 1] = relative_attention_bias[bucket(delta)][h],  delta = key - query
__global__ void t5_position_bias_kernel(const float* __restrict__ relb, const int* __restrict__ bucket, float* __restrict__ pb, int H, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * n) return;
    const int h = i / n, d = i - h * n;
    pb[i] = relb[(size_t)bucket[d] * H + h];
}

// One wave per (query, head, sequence).  qkv [B*L][3*inner] (q | k | v), out [B*L][inner].  Keys beyond the attention mask get
// finfo.min added, exactly as the reference model does (so an all-padding row degenerates to the same uniform average).
constexpr int T5_MAX_KEYS_PER_LANE = 8;      // L <= 512
__global__ __launch_bounds__(64) void t5_attention_kernel(const float* __restrict__ qkv, const float* __restrict__ pb, const int* __restrict__ mask,
                                                          float* __restrict__ out, int L, int H, int dkv) {
    extern __shared__ float t5_sm[];          // q [dkv] then p [L]
    float* sq = t5_sm;
    float* sp = t5_sm + dkv;
    const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x;
    const int inner = H * dkv, ld = 3 * inner;
    const float* base = qkv + (size_t)b * L * ld;
    for (int d = lane; d < dkv; d += 64) sq[d] = base[(size_t)i * ld + h * dkv + d];
    __syncthreads();
    float sc[T5_MAX_KEYS_PER_LANE];
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < T5_MAX_KEYS_PER_LANE; ++u) {
        const int j = lane + u * 64;
        sc[u] = -INFINITY;
        if (j < L) {
            const float4* kr = reinterpret_cast<const float4*>(base + (size_t)j * ld + inner + h * dkv);
            float s = 0.f;
            for (int d4 = 0; d4 < dkv / 4; ++d4) {
                const float4 kv = kr[d4];
                const float4 qv = reinterpret_cast<const float4*>(sq)[d4];
                s += (qv.x * kv.x + qv.y * kv.y) + (qv.z * kv.z + qv.w * kv.w);
            }
            s += pb[(size_t)h * (2 * L - 1) + (j - i + L - 1)];
            if (!mask[(size_t)b * L + j]) s += -3.4028234663852886e38f;
            sc[u] = s;
            mx = fmaxf(mx, s);
        }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < T5_MAX_KEYS_PER_LANE; ++u) {
        const int j = lane + u * 64;
        if (j < L) {
            const float p = expf(sc[u] - mx);
            sp[j] = p;
            sum += p;
        }
    }
    sum = wave_sum(sum);
    __syncthreads();
    const float inv = 1.0f / sum;
    for (int d = lane; d < dkv; d += 64) {
        const float* vc = base + 2 * inner + h * dkv + d;
        float o = 0.f;
        for (int j = 0; j < L; ++j) o += sp[j] * vc[(size_t)j * ld];
        out[((size_t)b * L + i) * inner + h * dkv + d] = o * inv;
    }
}

// relu in place on [rows][F]  |  gated: h[r][c] = gelu_new(g[r][c]) * g[r][F + c] from the stacked wi_0 | wi_1 output [rows][2F]
__global__ __launch_bounds__(256) void t5_act_kernel(const float* g, float* h, int64_t rows, int F, int gated) {      // g may alias h (relu)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * F) return;
    if (!gated) {
        h[i] = fmaxf(g[i], 0.f);
        return;
    }
    const int64_t r = i / F;
    const int c = (int)(i - r * F);
    const float x = g[r * 2 * F + c];
    const float t = tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x));
    h[i] = 0.5f * x * (1.0f + t) * g[r * 2 * F + F + c];
}

// embeddings * attention_mask (conditioners.py:341): rows of padding tokens become zero
__global__ __launch_bounds__(256) void t5_mask_rows_kernel(float* __restrict__ y, const int* __restrict__ mask, int64_t rows, int D) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < rows * D && !mask[i / D]) y[i] = 0.f;
}

struct T5Layer {
    float *ln1, *wqkv, *wo, *ln2, *wi, *wo2;
};

}  // namespace

struct sat_t5_plan {
    sat_t5_cfg cfg;
    std::map<std::string, std::pair<const float*, int64_t>> tensors;
    bool finalized = false;
    char* arena = nullptr;
    float *emb = nullptr, *relb = nullptr, *final_ln = nullptr;
    float *proj_w = nullptr, *proj_b = nullptr;       // Conditioner.proj_out (conditioners.py:23) when cfg.proj_dim > 0
    std::vector<T5Layer> layers;
    std::map<int, std::vector<int32_t>> buckets;      // relative-position buckets per sequence length (host copies stay alive for the async upload)
};

// transformers modeling_t5.py, T5Attention._relative_position_bucket (bidirectional = True), for delta = key - query in
// [-(l-1), l-1]: half of the buckets per sign; exact below max_exact = n/2, then logarithmic up to max_distance
extern "C" int sat_t5_relative_buckets(int32_t l, int32_t num_buckets, int32_t max_distance, int32_t* out_host) {
    SAT_CHECK_ARG(l > 0 && num_buckets >= 4 && num_buckets % 2 == 0 && max_distance > num_buckets / 4 && out_host, SAT_E_INVALID,
                  "t5_relative_buckets: bad arguments");
    const int nb = num_buckets / 2, max_exact = nb / 2;
    for (int delta = -(l - 1); delta <= l - 1; ++delta) {
        int ret = delta > 0 ? nb : 0;
        const int n = delta < 0 ? -delta : delta;
        if (n < max_exact) {
            ret += n;
        } else {
            // (the reference evaluates this in float32 and truncates; the small epsilon keeps the exact powers -- n = max_exact * r^k -- on
            // the side float32 puts them: tests/test_t5.py compares every delta with the transformers function)
            const double v = log((double)n / max_exact) / log((double)max_distance / max_exact) * (nb - max_exact);
            int large = max_exact + (int)(v + 1e-6);
            ret += large < nb - 1 ? large : nb - 1;
        }
        out_host[delta + l - 1] = ret;
    }
    return 0;
}

extern "C" int sat_t5_plan_create(const sat_t5_cfg* cfg, sat_t5_plan** out_plan) {
    SAT_CHECK_ARG(cfg && out_plan, SAT_E_INVALID, "t5_plan_create: null argument");
    SAT_CHECK_ARG(cfg->vocab_size > 0 && cfg->num_layers > 0 && cfg->num_heads > 0, SAT_E_INVALID, "t5_plan_create: bad sizes");
    SAT_CHECK_ARG(cfg->d_model % 16 == 0 && cfg->d_ff % 16 == 0 && cfg->d_kv % 16 == 0 && cfg->d_kv <= 256, SAT_E_UNSUPPORTED,
                  "t5_plan_create: d_model %d / d_ff %d / d_kv %d must be multiples of 16 (d_kv <= 256)", cfg->d_model, cfg->d_ff, cfg->d_kv);
    SAT_CHECK_ARG(cfg->rel_buckets >= 4 && cfg->rel_buckets % 2 == 0 && cfg->rel_max_distance > cfg->rel_buckets / 4, SAT_E_INVALID,
                  "t5_plan_create: bad relative-attention parameters");
    SAT_CHECK_ARG(cfg->eps > 0.f && cfg->proj_dim >= 0, SAT_E_INVALID, "t5_plan_create: layer_norm_epsilon must be positive, proj_dim >= 0");
    sat_t5_plan* p = new (std::nothrow) sat_t5_plan();
    SAT_CHECK_ARG(p, SAT_E_INVALID, "t5_plan_create: out of host memory");
    p->cfg = *cfg;
    *out_plan = p;
    return 0;
}

extern "C" void sat_t5_plan_destroy(sat_t5_plan* p) {
    if (!p) return;
    if (p->arena) (void)hipFree(p->arena);
    delete p;
}

extern "C" int sat_t5_plan_set_tensor(sat_t5_plan* p, const char* name, const float* data_dev, int64_t numel) {
    SAT_CHECK_ARG(p && name && data_dev && numel > 0, SAT_E_INVALID, "t5_plan_set_tensor: bad argument");
    p->tensors[name] = {data_dev, numel};
    return 0;
}

namespace {

int t5_get(sat_t5_plan* p, const std::string& name, int64_t numel, const float** out) {
    auto it = p->tensors.find(name);
    SAT_CHECK_ARG(it != p->tensors.end(), SAT_E_MISSING, "t5 plan: tensor '%s' was never set", name.c_str());
    SAT_CHECK_ARG(it->second.second == numel, SAT_E_INVALID, "t5 plan: tensor '%s' has %lld elements, expected %lld", name.c_str(),
                  (long long)it->second.second, (long long)numel);
    *out = it->second.first;
    return 0;
}

// two passes over the same code: sizes first (base == nullptr), then copies
int t5_build(sat_t5_plan* p, char* base, size_t* total, hipStream_t s) {
    const sat_t5_cfg& c = p->cfg;
    const int64_t D = c.d_model, I = (int64_t)c.num_heads * c.d_kv, F = c.d_ff;
    size_t off = 0;
    auto place = [&](const std::string& name, int64_t numel, float** dst, int64_t dst_off_elems = 0, bool advance = true) -> int {
        float* d = base ? reinterpret_cast<float*>(base + off) : nullptr;
        if (dst) *dst = d;
        if (base) {
            const float* src;
            SAT_TRY(t5_get(p, name, numel, &src));
            SAT_HIP(hipMemcpyAsync(d + dst_off_elems, src, numel * 4, hipMemcpyDeviceToDevice, s));
        }
        if (advance) off += (size_t)round_up((numel + dst_off_elems) * 4, 256);
        return 0;
    };
    // "shared.weight" and "encoder.embed_tokens.weight" are the same tensor in a T5 checkpoint; accept either
    const bool has_shared = p->tensors.count("shared.weight") != 0;
    SAT_TRY(place(has_shared ? "shared.weight" : "encoder.embed_tokens.weight", (int64_t)c.vocab_size * D, &p->emb));
    SAT_TRY(place("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", (int64_t)c.rel_buckets * c.num_heads, &p->relb));
    SAT_TRY(place("encoder.final_layer_norm.weight", D, &p->final_ln));
    if (c.proj_dim > 0) {
        SAT_TRY(place("proj_out.weight", (int64_t)c.proj_dim * D, &p->proj_w));
        SAT_TRY(place("proj_out.bias", c.proj_dim, &p->proj_b));
    }
    p->layers.resize(c.num_layers);
    for (int l = 0; l < c.num_layers; ++l) {
        T5Layer& L = p->layers[l];
        const std::string pf = "encoder.block." + std::to_string(l) + ".layer.";
        SAT_TRY(place(pf + "0.layer_norm.weight", D, &L.ln1));
        // q | k | v stacked into one [3I, D] weight: one GEMM
        SAT_TRY(place(pf + "0.SelfAttention.q.weight", I * D, &L.wqkv, 0, false));
        SAT_TRY(place(pf + "0.SelfAttention.k.weight", I * D, nullptr, I * D, false));
        SAT_TRY(place(pf + "0.SelfAttention.v.weight", I * D, nullptr, 2 * I * D, true));
        SAT_TRY(place(pf + "0.SelfAttention.o.weight", D * I, &L.wo));
        SAT_TRY(place(pf + "1.layer_norm.weight", D, &L.ln2));
        if (c.gated_gelu) {
            SAT_TRY(place(pf + "1.DenseReluDense.wi_0.weight", F * D, &L.wi, 0, false));
            SAT_TRY(place(pf + "1.DenseReluDense.wi_1.weight", F * D, nullptr, F * D, true));
        } else {
            SAT_TRY(place(pf + "1.DenseReluDense.wi.weight", F * D, &L.wi));
        }
        SAT_TRY(place(pf + "1.DenseReluDense.wo.weight", D * F, &L.wo2));
    }
    *total = off;
    return 0;
}

struct T5Ws {
    float *hid, *nrm, *qkv, *att, *ff, *ffh, *pb;
    int* bucket;
    size_t total;
};

T5Ws t5_carve(const sat_t5_plan* p, int b, int l, char* base) {
    const sat_t5_cfg& c = p->cfg;
    const size_t M = (size_t)b * l, D = c.d_model, I = (size_t)c.num_heads * c.d_kv, F = c.d_ff;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* q = base ? base + off : nullptr;
        off += (size_t)round_up((int64_t)bytes, 256);
        return q;
    };
    T5Ws w;
    w.hid = (float*)take(M * D * 4);
    w.nrm = (float*)take(M * D * 4);
    w.qkv = (float*)take(M * 3 * I * 4);
    w.att = (float*)take(M * I * 4);
    w.ff = (float*)take(M * F * (c.gated_gelu ? 2 : 1) * 4);
    w.ffh = c.gated_gelu ? (float*)take(M * F * 4) : w.ff;
    w.pb = (float*)take((size_t)c.num_heads * (2 * l - 1) * 4);
    w.bucket = (int*)take((size_t)(2 * l - 1) * 4);
    w.total = off;
    return w;
}

}  // namespace

extern "C" int sat_t5_plan_finalize(sat_t5_plan* p, sat_stream_t stream) {
    SAT_CHECK_ARG(p, SAT_E_INVALID, "t5_plan_finalize: null plan");
    hipStream_t s = (hipStream_t)stream;
    p->finalized = false;
    size_t total = 0;
    SAT_TRY(t5_build(p, nullptr, &total, s));
    if (p->arena) (void)hipFree(p->arena);
    p->arena = nullptr;
    SAT_HIP(hipMalloc((void**)&p->arena, total));
    SAT_TRY(t5_build(p, p->arena, &total, s));
    p->tensors.clear();      // the caller's pointers are not kept
    p->finalized = true;
    return 0;
}

extern "C" int sat_t5_workspace_bytes(const sat_t5_plan* p, int32_t b, int32_t l, size_t* out_bytes) {
    SAT_CHECK_ARG(p && out_bytes && b > 0 && l > 0, SAT_E_INVALID, "t5_workspace_bytes: bad arguments");
    *out_bytes = t5_carve(p, b, l, nullptr).total;
    return 0;
}

extern "C" int sat_t5_encode(sat_t5_plan* p, const int32_t* input_ids_dev, const int32_t* attention_mask_dev, float* out_dev, int32_t b,
                             int32_t l, int32_t mask_output, void* ws, size_t ws_bytes, sat_stream_t stream) {
    SAT_CHECK_ARG(p && p->finalized, SAT_E_STATE, "t5_encode: plan not finalized");
    SAT_CHECK_ARG(input_ids_dev && attention_mask_dev && out_dev && ws && b > 0 && l > 0, SAT_E_INVALID, "t5_encode: bad arguments");
    SAT_CHECK_ARG(l <= 64 * T5_MAX_KEYS_PER_LANE, SAT_E_UNSUPPORTED, "t5_encode: sequence length %d > %d", l, 64 * T5_MAX_KEYS_PER_LANE);
    SAT_CHECK_ARG(((uintptr_t)ws & 255) == 0, SAT_E_INVALID, "t5_encode: workspace must be 256-byte aligned");
    const sat_t5_cfg& c = p->cfg;
    hipStream_t s = (hipStream_t)stream;
    T5Ws w = t5_carve(p, b, l, (char*)ws);
    SAT_CHECK_ARG(ws_bytes >= w.total, SAT_E_WORKSPACE, "t5_encode: workspace %zu < required %zu", ws_bytes, w.total);
    const int M = b * l, D = c.d_model, H = c.num_heads, I = H * c.d_kv, F = c.d_ff, n = 2 * l - 1;

    // position bias of this length (T5Attention.compute_bias), shared by all blocks
    std::vector<int32_t>& bucket = p->buckets[l];
    if (bucket.empty()) {
        bucket.resize(n);
        SAT_TRY(sat_t5_relative_buckets(l, c.rel_buckets, c.rel_max_distance, bucket.data()));
    }
    SAT_HIP(hipMemcpyAsync(w.bucket, bucket.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(t5_position_bias_kernel, dim3(cdiv(H * n, 256)), dim3(256), 0, s, p->relb, w.bucket, w.pb, H, n);
    SAT_LAUNCH_CHECK();

    hipLaunchKernelGGL(t5_embed_kernel, dim3(M), dim3(256), 0, s, input_ids_dev, p->emb, w.hid, M, D, c.vocab_size);
    SAT_LAUNCH_CHECK();
    const size_t att_lds = (size_t)(c.d_kv + l) * 4;
    for (int li = 0; li < c.num_layers; ++li) {
        const T5Layer& L = p->layers[li];
        hipLaunchKernelGGL(t5_rmsnorm_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, w.hid, L.ln1, w.nrm, M, D, c.eps);
        SAT_LAUNCH_CHECK();
        SAT_TRY(sat_launch_gemm_f32(w.nrm, L.wqkv, nullptr, w.qkv, M, 3 * I, D, 3 * I, 0, nullptr, 1, 0, s));
        hipLaunchKernelGGL(t5_attention_kernel, dim3(l, H, b), dim3(64), att_lds, s, w.qkv, w.pb, attention_mask_dev, w.att, l, H, c.d_kv);
        SAT_LAUNCH_CHECK();
        SAT_TRY(sat_launch_gemm_f32(w.att, L.wo, nullptr, w.hid, M, D, I, D, 1, nullptr, 1, 0, s));
        hipLaunchKernelGGL(t5_rmsnorm_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, w.hid, L.ln2, w.nrm, M, D, c.eps);
        SAT_LAUNCH_CHECK();
        const int Fw = c.gated_gelu ? 2 * F : F;
        SAT_TRY(sat_launch_gemm_f32(w.nrm, L.wi, nullptr, w.ff, M, Fw, D, Fw, 0, nullptr, 1, 0, s));
        hipLaunchKernelGGL(t5_act_kernel, dim3((unsigned)cdiv((int64_t)M * F, 256)), dim3(256), 0, s, w.ff, w.ffh, (int64_t)M, F, c.gated_gelu);
        SAT_LAUNCH_CHECK();
        SAT_TRY(sat_launch_gemm_f32(w.ffh, L.wo2, nullptr, w.hid, M, D, F, D, 1, nullptr, 1, 0, s));
    }
    float* last = c.proj_dim > 0 ? w.nrm : out_dev;
    hipLaunchKernelGGL(t5_rmsnorm_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, w.hid, p->final_ln, last, M, D, c.eps);
    SAT_LAUNCH_CHECK();
    const int Dout = c.proj_dim > 0 ? c.proj_dim : D;
    if (c.proj_dim > 0) SAT_TRY(sat_launch_gemm_f32(w.nrm, p->proj_w, p->proj_b, out_dev, M, Dout, D, Dout, 0, nullptr, 1, 0, s));
    if (mask_output) {
        hipLaunchKernelGGL(t5_mask_rows_kernel, dim3((unsigned)cdiv((int64_t)M * Dout, 256)), dim3(256), 0, s, out_dev, attention_mask_dev, (int64_t)M, Dout);
        SAT_LAUNCH_CHECK();
    }
    return 0;
}
