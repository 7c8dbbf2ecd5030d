// LayerNorm (bias-less gamma + persisted beta buffer, eps 1e-5) -> bf16, one wave per row.
// Replaces F.layer_norm at models/transformer.py:205-206 (pre_norm / cross_attend_norm /
// ff_norm, :671/678/685 resp. :692/695/700).  HBM-bound: reads the fp32 residual stream
// once (float4 per lane), two-pass statistics in registers, writes the bf16 GEMM operand.
#include <type_traits>

#include "sat_common.h"

namespace {

// D = 64 * 4 * NV  (NV float4 per lane)
// FP8: the row is quantised to e4m3 with its own scale (amax / 448) -- the A operand of the fp8 GEMMs; y then points to bytes
// OT: the 16-bit output type, bf16 or IEEE fp16 (saturating conversion) -- the operand type of the GEMM kernels that read it
template <int NV, bool FP8 = false, typename OT = bf16_t>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, OT* __restrict__ y,
                                                        int m, int d, const float* __restrict__ sc, const float* __restrict__ sh,
                                                        int rps, int ld, float* __restrict__ row_scale = nullptr) {
    typedef OT otx4 __attribute__((ext_vector_type(4)));
    sat_saturate_for<OT>();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * d);
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = xr[i * 64 + lane];
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
        q += (a * a + b * b) + (c * c + e * e);
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + 1e-5f);
    const float4* gr = reinterpret_cast<const float4*>(gamma);
    const float4* br = reinterpret_cast<const float4*>(beta);
    [[maybe_unused]] otx4* yr = reinterpret_cast<otx4*>(y + (size_t)row * d);
    // adaLN modulation (wave-uniform branch): per-sequence (1 + scale) and shift vectors
    const float4* scr = sc ? reinterpret_cast<const float4*>(sc + (size_t)(row / rps) * ld) : nullptr;
    const float4* shr = sc ? reinterpret_cast<const float4*>(sh + (size_t)(row / rps) * ld) : nullptr;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float4 g = gr[i * 64 + lane];
        float4 b = beta ? br[i * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 r;
        r.x = (v[i].x - mean) * rstd * g.x + b.x;
        r.y = (v[i].y - mean) * rstd * g.y + b.y;
        r.z = (v[i].z - mean) * rstd * g.z + b.z;
        r.w = (v[i].w - mean) * rstd * g.w + b.w;
        if (scr) {
            float4 a = scr[i * 64 + lane], c = shr[i * 64 + lane];
            r.x = r.x * a.x + c.x;
            r.y = r.y * a.y + c.y;
            r.z = r.z * a.z + c.z;
            r.w = r.w * a.w + c.w;
        }
        if constexpr (FP8) {
            v[i] = r;             // keep the normalised row in registers for the amax pass
        } else {
            otx4 o;
            o[0] = (OT)r.x;
            o[1] = (OT)r.y;
            o[2] = (OT)r.z;
            o[3] = (OT)r.w;
            yr[i * 64 + lane] = o;
        }
    }
    if constexpr (FP8) {
        float am = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) am = fmaxf(am, fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w))));
        am = wave_max(am);
        const float scale = am > 0.f ? am * (1.0f / 448.0f) : 1.0f;
        const float inv = 1.0f / scale;
        unsigned* y8 = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(y) + (size_t)row * d);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            unsigned p = 0;
            p = __builtin_amdgcn_cvt_pk_fp8_f32(v[i].x * inv, v[i].y * inv, p, false);
            p = __builtin_amdgcn_cvt_pk_fp8_f32(v[i].z * inv, v[i].w * inv, p, true);
            y8[i * 64 + lane] = p;
        }
        if (lane == 0) row_scale[row] = scale;
    }
}

// generic fallback: any d % 4 == 0 (strided loop, row re-read from L2)
template <typename OT>
__global__ __launch_bounds__(256) void layernorm_generic_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, OT* __restrict__ y,
                                                                int m, int d, const float* __restrict__ sc,
                                                                const float* __restrict__ sh, int rps, int ld) {
    sat_saturate_for<OT>();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    const float* xr = x + (size_t)row * d;
    float s = 0.f;
    for (int i = lane; i < d; i += 64) s += xr[i];
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
    for (int i = lane; i < d; i += 64) {
        float a = xr[i] - mean;
        q += a * a;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + 1e-5f);
    const float* scr = sc ? sc + (size_t)(row / rps) * ld : nullptr;
    const float* shr = sc ? sh + (size_t)(row / rps) * ld : nullptr;
    for (int i = lane; i < d; i += 64) {
        float r = (xr[i] - mean) * rstd * gamma[i] + (beta ? beta[i] : 0.f);
        if (scr) r = r * scr[i] + shr[i];
        y[(size_t)row * d + i] = (OT)r;
    }
}

template <typename OT>
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, OT* __restrict__ y, int64_t n) {
    typedef OT otx4 __attribute__((ext_vector_type(4)));
    sat_saturate_for<OT>();
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (; i + 3 < n; i += stride) {
        float4 v = *reinterpret_cast<const float4*>(x + i);
        otx4 o;
        o[0] = (OT)v.x;
        o[1] = (OT)v.y;
        o[2] = (OT)v.z;
        o[3] = (OT)v.w;
        *reinterpret_cast<otx4*>(y + i) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t j = n & ~(int64_t)3; j < n; ++j) y[j] = (OT)x[j];
}

// out[n'][k] = bf16(w[src(n')][k]); swiglu_interleave: groups of 64 output rows = 32 value
// rows g*32.. followed by the 32 matching gate rows n/2 + g*32..  (see gemm_bf16.hip)
template <typename OT>
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ w, OT* __restrict__ out, int n, int k,
                                                        int interleave) {
    typedef OT otx4 __attribute__((ext_vector_type(4)));
    sat_saturate_for<OT>();
    const int row = blockIdx.x;
    int src = row;
    if (interleave) {
        int g = row >> 6, c = row & 63;
        src = (c < 32) ? (g * 32 + c) : (n / 2 + g * 32 + (c - 32));
    }
    const float* wr = w + (size_t)src * k;
    OT* o = out + (size_t)row * k;
    for (int i = threadIdx.x * 4; i < k; i += 256 * 4) {
        float4 v = *reinterpret_cast<const float4*>(wr + i);
        otx4 p;
        p[0] = (OT)v.x;
        p[1] = (OT)v.y;
        p[2] = (OT)v.z;
        p[3] = (OT)v.w;
        *reinterpret_cast<otx4*>(o + i) = p;
    }
}

// Weights of a GEMM that absorbs the LayerNorm in front of it (GemmArgs::ln_*): out[n'][k] = bf16(gamma[k] * w[src(n')][k]),
// c1[n'] = sum_k float(out[n'][k]) (the sum of what the MFMA will actually multiply), c2[n'] = sum_k beta[k] * w[src][k] + bias.
// One workgroup per output row, same row permutation as pack_rows_kernel.
template <typename OT>
__global__ __launch_bounds__(256) void pack_rows_ln_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ bias,
                                                           OT* __restrict__ out, float* __restrict__ c1, float* __restrict__ c2, int n, int k,
                                                           int interleave) {
    typedef OT otx4 __attribute__((ext_vector_type(4)));
    sat_saturate_for<OT>();
    __shared__ float red[2][4];
    const int row = blockIdx.x;
    int src = row;
    if (interleave) {
        int g = row >> 6, c = row & 63;
        src = (c < 32) ? (g * 32 + c) : (n / 2 + g * 32 + (c - 32));
    }
    const float* wr = w + (size_t)src * k;
    OT* o = out + (size_t)row * k;
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x * 4; i < k; i += 256 * 4) {
        const float4 v = *reinterpret_cast<const float4*>(wr + i);
        const float4 gm = *reinterpret_cast<const float4*>(gamma + i);
        const float4 bt = *reinterpret_cast<const float4*>(beta + i);
        otx4 p;
        p[0] = (OT)(v.x * gm.x);
        p[1] = (OT)(v.y * gm.y);
        p[2] = (OT)(v.z * gm.z);
        p[3] = (OT)(v.w * gm.w);
        *reinterpret_cast<otx4*>(o + i) = p;
        s1 += ((float)p[0] + (float)p[1]) + ((float)p[2] + (float)p[3]);
        s2 += (v.x * bt.x + v.y * bt.y) + (v.z * bt.z + v.w * bt.w);
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s1;
        red[1][threadIdx.x >> 6] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        c1[row] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        c2[row] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]) + (bias ? bias[src] : 0.f);
    }
}

// one workgroup per output row: amax -> scale = amax / 448 -> e4m3 bytes (4 per thread per pass); same row permutation as
// pack_rows_kernel for the SwiGLU weight
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const float* __restrict__ w, unsigned char* __restrict__ out,
                                                             float* __restrict__ row_scale, int n, int k, int interleave) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    int src = row;
    if (interleave) {
        int g = row >> 6, c = row & 63;
        src = (c < 32) ? (g * 32 + c) : (n / 2 + g * 32 + (c - 32));
    }
    const float* wr = w + (size_t)src * k;
    float am = 0.f;
    for (int i = threadIdx.x * 4; i < k; i += 256 * 4) {
        float4 v = *reinterpret_cast<const float4*>(wr + i);
        am = fmaxf(am, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    am = wave_max(am);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = am;
    __syncthreads();
    am = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float scale = am > 0.f ? am * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / scale;
    unsigned* o = reinterpret_cast<unsigned*>(out + (size_t)row * k);
    for (int i = threadIdx.x * 4; i < k; i += 256 * 4) {
        float4 v = *reinterpret_cast<const float4*>(wr + i);
        unsigned p = 0;
        p = __builtin_amdgcn_cvt_pk_fp8_f32(v.x * inv, v.y * inv, p, false);
        p = __builtin_amdgcn_cvt_pk_fp8_f32(v.z * inv, v.w * inv, p, true);
        o[i >> 2] = p;
    }
    if (threadIdx.x == 0) row_scale[row] = scale;
}

// MXFP8 rows (reference producer for the kernel tests; in the plan the SwiGLU epilogue writes this format): one wave per row,
// lane l owns elements l, l + 64, ...; every 32 consecutive elements share one E8M0 scale 2^ceil(log2(amax / 448))
__global__ __launch_bounds__(256) void quant_mx_rows_kernel(const float* __restrict__ x, unsigned char* __restrict__ out,
                                                            unsigned char* __restrict__ scales, int rows, int k) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    for (int c0 = 0; c0 < k; c0 += 64) {
        const float v = x[(size_t)row * k + c0 + lane];
        const float am = half32_max(fabsf(v));
        const float t = am * (1.0f / 448.0f);
        const unsigned tb = __float_as_uint(t);
        int e = (int)((tb >> 23) & 0xff) - 127 + ((tb & 0x7fffff) ? 1 : 0);
        e = am > 0.f ? (e < -127 ? -127 : (e > 127 ? 127 : e)) : -127;
        const float inv = __uint_as_float((unsigned)(127 - e) << 23);
        const unsigned q = __builtin_amdgcn_cvt_pk_fp8_f32(v * inv, 0.f, 0u, false);
        out[(size_t)row * k + c0 + lane] = (unsigned char)(q & 0xff);
        if ((lane & 31) == 0) scales[(size_t)row * (k >> 5) + ((c0 + lane) >> 5)] = (unsigned char)(e + 127);
    }
}

__global__ void pack_bias_kernel(const float* __restrict__ b, float* __restrict__ out, int n, int interleave) {
    int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    int src = row;
    if (interleave) {
        int g = row >> 6, c = row & 63;
        src = (c < 32) ? (g * 32 + c) : (n / 2 + g * 32 + (c - 32));
    }
    out[row] = b[src];
}

// models/transformer.py:130-148: freqs[s][j] = s * inv_freq[j] (fp32), cos/sin in fp32
__global__ void rope_table_kernel(const float* __restrict__ inv_freq, float* __restrict__ cos_t, float* __restrict__ sin_t,
                                  int s_len) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s_len * 16) return;
    int s = i >> 4, j = i & 15;
    float f = (float)s * inv_freq[j];
    cos_t[i] = cosf(f);
    sin_t[i] = sinf(f);
}

}  // namespace

int sat_launch_layernorm_mod(const float* x, const float* gamma, const float* beta, bf16_t* y, int m, int d, const float* sc,
                             const float* sh, int rps, int ld, hipStream_t s, int f16) {
    SAT_CHECK_ARG(x && gamma && y && m > 0 && d > 0 && d % 4 == 0, SAT_E_INVALID, "layernorm: bad args m=%d d=%d", m, d);
    SAT_CHECK_ARG((sc == nullptr) == (sh == nullptr) && (!sc || (rps > 0 && ld % 4 == 0)), SAT_E_INVALID,
                  "layernorm: modulation needs both scale and shift, rows_per_seq > 0 and ld %% 4 == 0");
    dim3 grid(cdiv(m, 4)), block(256);
    auto go = [&](auto* yt) {
        using OT = std::remove_pointer_t<decltype(yt)>;
        switch ((d % 256 == 0 && d / 256 <= 8) ? d / 256 : 0) {
            case 1: hipLaunchKernelGGL((layernorm_kernel<1, false, OT>), grid, block, 0, s, x, gamma, beta, yt, m, d, sc, sh, rps, ld, (float*)nullptr); break;
            case 2: hipLaunchKernelGGL((layernorm_kernel<2, false, OT>), grid, block, 0, s, x, gamma, beta, yt, m, d, sc, sh, rps, ld, (float*)nullptr); break;
            case 3: hipLaunchKernelGGL((layernorm_kernel<3, false, OT>), grid, block, 0, s, x, gamma, beta, yt, m, d, sc, sh, rps, ld, (float*)nullptr); break;
            case 4: hipLaunchKernelGGL((layernorm_kernel<4, false, OT>), grid, block, 0, s, x, gamma, beta, yt, m, d, sc, sh, rps, ld, (float*)nullptr); break;
            case 6: hipLaunchKernelGGL((layernorm_kernel<6, false, OT>), grid, block, 0, s, x, gamma, beta, yt, m, d, sc, sh, rps, ld, (float*)nullptr); break;
            case 8: hipLaunchKernelGGL((layernorm_kernel<8, false, OT>), grid, block, 0, s, x, gamma, beta, yt, m, d, sc, sh, rps, ld, (float*)nullptr); break;
            default: hipLaunchKernelGGL(layernorm_generic_kernel<OT>, grid, block, 0, s, x, gamma, beta, yt, m, d, sc, sh, rps, ld); break;
        }
    };
    if (f16) go(reinterpret_cast<_Float16*>(y));
    else go(y);
    SAT_LAUNCH_CHECK();
    return 0;
}

int sat_launch_layernorm_fp8(const float* x, const float* gamma, const float* beta, void* y8, float* row_scale, int m, int d,
                             const float* sc, const float* sh, int rps, int ld, hipStream_t s) {
    SAT_CHECK_ARG(x && gamma && y8 && row_scale && m > 0 && d > 0, SAT_E_INVALID, "layernorm_fp8: bad args m=%d d=%d", m, d);
    SAT_CHECK_ARG((sc == nullptr) == (sh == nullptr) && (!sc || (rps > 0 && ld % 4 == 0)), SAT_E_INVALID, "layernorm_fp8: bad modulation");
    dim3 grid(cdiv(m, 4)), block(256);
    bf16_t* y = reinterpret_cast<bf16_t*>(y8);
    switch (d % 256 == 0 ? d / 256 : 0) {
        case 1: hipLaunchKernelGGL((layernorm_kernel<1, true>), grid, block, 0, s, x, gamma, beta, y, m, d, sc, sh, rps, ld, row_scale); break;
        case 2: hipLaunchKernelGGL((layernorm_kernel<2, true>), grid, block, 0, s, x, gamma, beta, y, m, d, sc, sh, rps, ld, row_scale); break;
        case 3: hipLaunchKernelGGL((layernorm_kernel<3, true>), grid, block, 0, s, x, gamma, beta, y, m, d, sc, sh, rps, ld, row_scale); break;
        case 4: hipLaunchKernelGGL((layernorm_kernel<4, true>), grid, block, 0, s, x, gamma, beta, y, m, d, sc, sh, rps, ld, row_scale); break;
        case 6: hipLaunchKernelGGL((layernorm_kernel<6, true>), grid, block, 0, s, x, gamma, beta, y, m, d, sc, sh, rps, ld, row_scale); break;
        case 8: hipLaunchKernelGGL((layernorm_kernel<8, true>), grid, block, 0, s, x, gamma, beta, y, m, d, sc, sh, rps, ld, row_scale); break;
        default:
            sat_set_error("layernorm_fp8: d=%d must be 256 * {1,2,3,4,6,8}", d);
            return SAT_E_UNSUPPORTED;
    }
    SAT_LAUNCH_CHECK();
    return 0;
}

int sat_launch_layernorm(const float* x, const float* gamma, const float* beta, bf16_t* y, int m, int d, hipStream_t s, int f16) {
    return sat_launch_layernorm_mod(x, gamma, beta, y, m, d, nullptr, nullptr, 1, 0, s, f16);
}

int sat_launch_cast_bf16(const float* x, bf16_t* y, int64_t n, hipStream_t s, int f16) {
    SAT_CHECK_ARG(x && y && n > 0, SAT_E_INVALID, "cast: bad args");
    SAT_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 8 == 0), SAT_E_INVALID, "cast: unaligned pointers");
    int blocks = (int)((n / 4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    if (f16) hipLaunchKernelGGL(cast_bf16_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, x, reinterpret_cast<_Float16*>(y), n);
    else hipLaunchKernelGGL(cast_bf16_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, x, y, n);
    SAT_LAUNCH_CHECK();
    return 0;
}

int sat_launch_pack_rows_bf16(const float* w, bf16_t* out, int n, int k, int swiglu_interleave, hipStream_t s, int f16) {
    SAT_CHECK_ARG(w && out && n > 0 && k > 0 && k % 4 == 0, SAT_E_INVALID, "pack_rows: bad args");
    SAT_CHECK_ARG(!swiglu_interleave || n % 128 == 0, SAT_E_UNSUPPORTED, "pack_rows: swiglu needs n %% 128 == 0");
    if (f16) hipLaunchKernelGGL(pack_rows_kernel<_Float16>, dim3(n), dim3(256), 0, s, w, reinterpret_cast<_Float16*>(out), n, k, swiglu_interleave);
    else hipLaunchKernelGGL(pack_rows_kernel<bf16_t>, dim3(n), dim3(256), 0, s, w, out, n, k, swiglu_interleave);
    SAT_LAUNCH_CHECK();
    return 0;
}

int sat_launch_pack_rows_ln(const float* w, const float* gamma, const float* beta, const float* bias, bf16_t* out, float* c1, float* c2,
                            int n, int k, int swiglu_interleave, hipStream_t s, int f16) {
    SAT_CHECK_ARG(w && gamma && beta && out && c1 && c2 && n > 0 && k > 0 && k % 4 == 0, SAT_E_INVALID, "pack_rows_ln: bad args");
    SAT_CHECK_ARG(!swiglu_interleave || n % 128 == 0, SAT_E_UNSUPPORTED, "pack_rows_ln: swiglu needs n %% 128 == 0");
    if (f16) hipLaunchKernelGGL(pack_rows_ln_kernel<_Float16>, dim3(n), dim3(256), 0, s, w, gamma, beta, bias, reinterpret_cast<_Float16*>(out), c1, c2, n, k, swiglu_interleave);
    else hipLaunchKernelGGL(pack_rows_ln_kernel<bf16_t>, dim3(n), dim3(256), 0, s, w, gamma, beta, bias, out, c1, c2, n, k, swiglu_interleave);
    SAT_LAUNCH_CHECK();
    return 0;
}

int sat_launch_quant_rows_fp8(const float* w, void* out8, float* row_scale, int n, int k, int swiglu_interleave, hipStream_t s) {
    SAT_CHECK_ARG(w && out8 && row_scale && n > 0 && k > 0 && k % 4 == 0, SAT_E_INVALID, "quant_rows_fp8: bad args");
    SAT_CHECK_ARG(!swiglu_interleave || n % 128 == 0, SAT_E_UNSUPPORTED, "quant_rows_fp8: swiglu needs n %% 128 == 0");
    hipLaunchKernelGGL(quant_rows_fp8_kernel, dim3(n), dim3(256), 0, s, w, reinterpret_cast<unsigned char*>(out8), row_scale, n, k,
                       swiglu_interleave);
    SAT_LAUNCH_CHECK();
    return 0;
}

int sat_launch_quant_mx_rows(const float* x, void* out8, void* scales, int rows, int k, hipStream_t s) {
    SAT_CHECK_ARG(x && out8 && scales && rows > 0 && k > 0 && k % 64 == 0, SAT_E_INVALID, "quant_mx_rows: bad args (k %% 64 == 0)");
    hipLaunchKernelGGL(quant_mx_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, reinterpret_cast<unsigned char*>(out8),
                       reinterpret_cast<unsigned char*>(scales), rows, k);
    SAT_LAUNCH_CHECK();
    return 0;
}

int sat_launch_pack_bias(const float* b, float* out, int n, int swiglu_interleave, hipStream_t s) {
    hipLaunchKernelGGL(pack_bias_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, b, out, n, swiglu_interleave);
    SAT_LAUNCH_CHECK();
    return 0;
}

int sat_launch_rope_table(const float* inv_freq, float* cos_t, float* sin_t, int s_len, hipStream_t s) {
    hipLaunchKernelGGL(rope_table_kernel, dim3(cdiv((int64_t)s_len * 16, 256)), dim3(256), 0, s, inv_freq, cos_t, sin_t, s_len);
    SAT_LAUNCH_CHECK();
    return 0;
}
