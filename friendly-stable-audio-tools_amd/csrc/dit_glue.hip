// Small fp32 kernels around the DiT blocks: conditioning / timestep MLPs,
// the folded 1x1-conv + project_in / project_out, CFG combine + VDenoiser scalings, the
// DPM-Solver++(3M) SDE state update, VAE sampling and int16 quantisation.  All of them are
// HBM/L2-bound elementwise or skinny-GEMV work kept in fp32 (the reference computes them
// in fp32 on CPU); none is reshaped into a GEMM to reach MFMA.
#include "dit_glue.h"

namespace {

// y[r][n] = act( sum_k x[r][k] * W[n][k] + bias[n] ) (+ add[r][n]) ; one wave per n, RB rows
// per wave.  x row stride ldx, y row stride ldy (lets the caller write straight into the
// prepend-token rows of the residual stream).  OUT16: 0 = y is fp32; 1 = bf16, 2 = IEEE fp16 (GEMM A operand).
template <int RB, int OUT16>
__global__ __launch_bounds__(256) void small_linear_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ W,
                                                           const float* __restrict__ bias, const float* __restrict__ add,
                                                           int ldadd, void* __restrict__ yv, int ldy, int R, int N, int K,
                                                           int act) {
    if constexpr (OUT16 == 2) sat_f16_saturate_on();
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r0 = blockIdx.y * RB;
    if (n >= N) return;
    float acc[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) acc[i] = 0.f;
    const float* wr = W + (size_t)n * K;
    for (int k = lane * 4; k < K; k += 256) {
        float4 w = *reinterpret_cast<const float4*>(wr + k);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            int r = r0 + i;
            r = r < R ? r : R - 1;
            float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * ldx + k);
            acc[i] += (w.x * v.x + w.y * v.y) + (w.z * v.z + w.w * v.w);
        }
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) acc[i] = wave_sum(acc[i]);
    if (lane == 0) {
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            int r = r0 + i;
            if (r < R) {
                float v = acc[i] + bv;
                if (act == 1) v = silu_f(v);
                if (add) v += add[(size_t)r * ldadd + n];
                if (act == 2) v = silu_f(v);       // SiLU after the add (input of to_scale_shift_gate, transformer.py:653)
                if constexpr (OUT16 == 1)
                    reinterpret_cast<bf16_t*>(yv)[(size_t)r * ldy + n] = f32_to_bf16(v);
                else if constexpr (OUT16 == 2)
                    reinterpret_cast<_Float16*>(yv)[(size_t)r * ldy + n] = (_Float16)v;
                else
                    reinterpret_cast<float*>(yv)[(size_t)r * ldy + n] = v;
            }
        }
    }
}

// models/blocks.py:95-97: f = 2*pi*t*w ; cat(cos f, sin f)
__global__ void fourier_kernel(const float* __restrict__ t, float t_const, const float* __restrict__ w, float* __restrict__ out,
                               int B, int half_feat) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half_feat) return;
    int b = i / half_feat, j = i - b * half_feat;
    float tv = t ? t[b] : t_const;
    float f = (6.283185307179586f * tv) * w[j];
    out[(size_t)b * 2 * half_feat + j] = cosf(f);
    out[(size_t)b * 2 * half_feat + half_feat + j] = sinf(f);
}

// Weff[n][c] = Win[n][c] + sum_j Win[n][j] * Wpre[j][c]   (project_in o (I + preprocess_conv))
__global__ void fold_in_kernel(const float* __restrict__ Win, const float* __restrict__ Wpre, float* __restrict__ Weff, int D, int C) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D * C) return;
    int n = i / C, c = i - n * C;
    float acc = Win[i];
    for (int j = 0; j < C; ++j) acc += Win[n * C + j] * Wpre[j * C + c];
    Weff[(size_t)c * D + n] = acc;          // stored transposed, [C][D]: the input projection reads it with lane = n
}
// WeffT[n][c] = Wout[c][n] + sum_j Wpost[c][j] * Wout[j][n]   ((I + postprocess_conv) o project_out), stored [D][C]
__global__ void fold_out_kernel(const float* __restrict__ Wout, const float* __restrict__ Wpost, float* __restrict__ WeffT, int D, int C) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * D) return;
    int c = i / D, n = i - c * D;
    float acc = Wout[i];
    for (int j = 0; j < C; ++j) acc += Wpost[c * C + j] * Wout[j * D + n];
    WeffT[(size_t)n * C + c] = acc;
}

// adaLN (transformer.py:667-688): ssg[b][layer][6][D] = (scale_self, shift_self, gate_self, scale_ff, shift_ff, gate_ff) straight
// from to_scale_shift_gate; rewritten in place to what the consumers multiply with: scale -> 1 + scale, gate -> sigmoid(1 - gate).
__global__ __launch_bounds__(256) void adaln_finish_kernel(float* __restrict__ ssg, int64_t n, int D) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int chunk = (int)((i / D) % 6);
    float v = ssg[i];
    if (chunk == 0 || chunk == 3) v = 1.f + v;
    else if (chunk == 2 || chunk == 5) v = 1.f / (1.f + __expf(-(1.f - v)));
    ssg[i] = v;
}

// (S - T = number of prepended rows: 1 for global_cond_type 'prepend', 0 for 'adaLN')
// X[b, (S-T)+t, n] = xscale * sum_c Weff[n][c] * x[b % xB][c][t]       (C <= 64)
// One workgroup = IP_TT time steps x 256 output channels, thread = output channel n with one accumulator per time step.  The
// folded weight is read TRANSPOSED (WeffT[c][n]: consecutive lanes read consecutive floats, once per workgroup).  The inputs of the
// time steps are the SAME for every lane, so they are read through wave-uniform addresses -- scalar loads into SGPRs, consumed as
// the scalar operand of v_fmac: no LDS traffic at all (a first version broadcast them from LDS and was bound by the LDS pipe: one
// 16-byte broadcast read per 4 FMAs is twice what four SIMDs can be fed).  Stores are row-contiguous (1 KiB per wave).
constexpr int IP_TT = 8;       // measured at T = 1024: 32 -> 19.6 us, 16 -> 15.3, 8 -> 12.5, 4 -> 18.0
__global__ __launch_bounds__(256) void input_proj_kernel(const float* __restrict__ x, const float* __restrict__ WeffT,
                                                         float* __restrict__ X, int xB, int C, int T, int S, int D, float xscale) {
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * IP_TT;
    const int n = blockIdx.y * 256 + threadIdx.x;
    const float* __restrict__ xb = x + (size_t)(b % xB) * C * T + t0;      // wave-uniform
    if (n >= D) return;
    float acc[IP_TT];
#pragma unroll
    for (int tt = 0; tt < IP_TT; ++tt) acc[tt] = 0.f;
    if (t0 + IP_TT <= T) {
#pragma unroll 4
        for (int c = 0; c < C; ++c) {
            const float w = WeffT[(size_t)c * D + n];
            const float* __restrict__ xr = xb + (size_t)c * T;
#pragma unroll
            for (int tt = 0; tt < IP_TT; ++tt) acc[tt] += w * xr[tt];
        }
    } else {          // last chunk of a sequence whose length is not a multiple of IP_TT
        for (int c = 0; c < C; ++c) {
            const float w = WeffT[(size_t)c * D + n];
            const float* __restrict__ xr = xb + (size_t)c * T;
#pragma unroll
            for (int tt = 0; tt < IP_TT; ++tt) acc[tt] += w * (t0 + tt < T ? xr[tt] : 0.f);
        }
    }
#pragma unroll
    for (int tt = 0; tt < IP_TT; ++tt)
        if (t0 + tt < T) X[((size_t)b * S + (S - T) + t0 + tt) * D + n] = acc[tt] * xscale;
}

// out[b][c][t] = sum_n WeffT[n][c] * X[b, (S-T)+t, n].  OP_TOK tokens per workgroup: their rows of X are staged in LDS once, the eight
// waves split the D input channels.  Inside a wave, lane = (4 output channels, 4 consecutive input channels): one step covers 16
// input channels with four 16-byte weight loads per lane (each instruction reads four full 256-byte weight rows) and OP_TOK
// broadcast float4 reads of the staged rows for 16 x OP_TOK FMAs -- the round-1 kernel had one 4-byte weight load per 16 FMAs and sat on
// load latency.  The partial sums of the 4 x 8 input-channel slices meet through two lane exchanges and LDS; results are written with
// OP_TOK consecutive time steps per channel.
constexpr int OP_TOK = 8;      // 8 tokens: 256 workgroups at T = 1024 x 2 sequences (16 left half the CUs idle: 26.7 -> 15.9 us; 4: 16.4 us)
__global__ __launch_bounds__(512) void output_proj_kernel(const float* __restrict__ X, const float* __restrict__ WeffT,
                                                          float* __restrict__ out, int Bf, int C, int T, int S, int D) {
    extern __shared__ __attribute__((aligned(16))) char smem_op[];
    float* xs = reinterpret_cast<float*>(smem_op);          // [OP_TOK][D]
    float* red = xs + (size_t)OP_TOK * D;                   // [8 waves][64 channels][OP_TOK]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tok0 = blockIdx.x * OP_TOK;                   // token index b * T + t
    const int total = Bf * T;
    for (int i = threadIdx.x; i < OP_TOK * (D / 4); i += 512) {
        const int tk = i / (D / 4), q = i - tk * (D / 4);
        int row = tok0 + tk;
        row = row < total ? row : total - 1;
        const int b = row / T, t = row - b * T;
        reinterpret_cast<float4*>(xs + (size_t)tk * D)[q] = reinterpret_cast<const float4*>(X + ((size_t)b * S + (S - T) + t) * D)[q];
    }
    __syncthreads();
    const int cg = lane & 15, ns = lane >> 4;               // channels 4cg .. 4cg+3, input channels n0 + 4ns .. +3 of a step
    const int c0 = 4 * cg < C ? 4 * cg : 0;                 // (C % 4 == 0 is checked by the launcher; lanes beyond C recompute group 0)
    float acc[4][OP_TOK];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int tk = 0; tk < OP_TOK; ++tk) acc[e][tk] = 0.f;
    const int nq = D / 8;                                   // input channels per wave (D % 128 == 0: a multiple of 16)
    const float* wp = WeffT + (size_t)(wave * nq + 4 * ns) * C + c0;
    float4 w[4], wn[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] = *reinterpret_cast<const float4*>(wp + (size_t)u * C);
    for (int n0 = 0; n0 < nq; n0 += 16) {
        if (n0 + 16 < nq) {
#pragma unroll
            for (int u = 0; u < 4; ++u) wn[u] = *reinterpret_cast<const float4*>(wp + (size_t)(n0 + 16 + u) * C);
        }
        const float* xrow = xs + wave * nq + n0 + 4 * ns;
#pragma unroll
        for (int tk = 0; tk < OP_TOK; ++tk) {
            const float4 xv = *reinterpret_cast<const float4*>(xrow + (size_t)tk * D);
            acc[0][tk] += (w[0].x * xv.x + w[1].x * xv.y) + (w[2].x * xv.z + w[3].x * xv.w);
            acc[1][tk] += (w[0].y * xv.x + w[1].y * xv.y) + (w[2].y * xv.z + w[3].y * xv.w);
            acc[2][tk] += (w[0].z * xv.x + w[1].z * xv.y) + (w[2].z * xv.z + w[3].z * xv.w);
            acc[3][tk] += (w[0].w * xv.x + w[1].w * xv.y) + (w[2].w * xv.z + w[3].w * xv.w);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = wn[u];
    }
    // fold the four input-channel slices of the wave (lanes l, l+16, l+32, l+48), then the eight waves through LDS
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int tk = 0; tk < OP_TOK; ++tk) {
            float v = acc[e][tk];
            v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));      // lane ^ 16
            unsigned a = __float_as_uint(v), bq = a;
            half_swap(a, bq);                                                                  // lane ^ 32
            v = __uint_as_float(a) + __uint_as_float(bq);
            if (lane < 16) red[((size_t)wave * 64 + 4 * cg + e) * OP_TOK + tk] = v;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < OP_TOK * 64; i += 512) {
        const int ch = i / OP_TOK, tk = i - ch * OP_TOK;    // OP_TOK consecutive threads = consecutive time steps of one channel
        const int row = tok0 + tk;
        if (ch < C && row < total) {
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < 8; ++wv) v += red[((size_t)wv * 64 + ch) * OP_TOK + tk];
            const int b = row / T, t = row - b * T;
            out[((size_t)b * C + ch) * T + t] = v;
        }
    }
}


// CFG combine (models/dit.py:338-339) + VDenoiser without the std rescale: pure elementwise, one thread per element
__global__ __launch_bounds__(256) void cfg_denoise_ew_kernel(const float* __restrict__ mo, const float* __restrict__ x,
                                                             float* __restrict__ den, int64_t n_per_half, int use_cfg,
                                                             float cfg_scale, float c_out, float c_skip) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_per_half) return;
    float vc = mo[i];
    float g = vc;
    if (use_cfg) {
        float vu = mo[i + n_per_half];
        g = vu + (vc - vu) * cfg_scale;
    }
    den[i] = g * c_out + x[i] * c_skip;
}

// models/dit.py:336-345 (CFG combine + optional std rescale) then VDenoiser:
// den = cfg * c_out + x * c_skip.  One thread per (b,t) column, loops the channels.
__global__ __launch_bounds__(256) void cfg_denoise_kernel(const float* __restrict__ mo, const float* __restrict__ x,
                                                          float* __restrict__ den, int B, int C, int T, int use_cfg,
                                                          float cfg_scale, float scale_phi, float c_out, float c_skip) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * T) return;
    int b = i / T, t = i - b * T;
    const float* pc = mo + (size_t)b * C * T + t;
    const float* pu = mo + (size_t)(b + B) * C * T + t;
    const float* px = x + (size_t)b * C * T + t;
    float* pd = den + (size_t)b * C * T + t;
    float ratio_mix = 1.0f;
    if (use_cfg && scale_phi != 0.0f) {
        // unbiased std over channels of cond and of the cfg output (torch.std, dim=1)
        float sc = 0.f, sg = 0.f;
        for (int c = 0; c < C; ++c) {
            float vc = pc[(size_t)c * T], vu = pu[(size_t)c * T];
            sc += vc;
            sg += vu + (vc - vu) * cfg_scale;
        }
        float mc = sc / C, mg = sg / C, qc = 0.f, qg = 0.f;
        for (int c = 0; c < C; ++c) {
            float vc = pc[(size_t)c * T], vu = pu[(size_t)c * T];
            float g = vu + (vc - vu) * cfg_scale;
            qc += (vc - mc) * (vc - mc);
            qg += (g - mg) * (g - mg);
        }
        float stdc = sqrtf(qc / (C - 1)), stdg = sqrtf(qg / (C - 1));
        ratio_mix = scale_phi * (stdc / stdg) + (1.0f - scale_phi);
    }
    for (int c = 0; c < C; ++c) {
        float vc = pc[(size_t)c * T];
        float g = vc;
        if (use_cfg) {
            float vu = pu[(size_t)c * T];
            g = (vu + (vc - vu) * cfg_scale) * ratio_mix;
        }
        pd[(size_t)c * T] = g * c_out + px[(size_t)c * T] * c_skip;
    }
}

__global__ __launch_bounds__(256) void dpmpp3m_update_kernel(float* __restrict__ x, const float* __restrict__ d,
                                                             const float* __restrict__ d1, const float* __restrict__ d2,
                                                             const float* __restrict__ noise, float a, float b, float c1,
                                                             float c2, float cn, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        float dv = d[i];
        float v = a * x[i] + b * dv;
        if (d1) {
            float d1v = d1[i];
            v += c1 * (dv - d1v);
            if (d2) v += c2 * (d1v - d2[i]);
        }
        if (noise) v += cn * noise[i];
        x[i] = v;
    }
}

// models/bottleneck.py:46-52
__global__ __launch_bounds__(256) void vae_sample_kernel(const float* __restrict__ ms, const float* __restrict__ noise,
                                                         float* __restrict__ z, int B, int C, int T) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * C * T) return;
    int64_t ct = (int64_t)C * T;
    int b = (int)(i / ct);
    int64_t r = i - (int64_t)b * ct;
    float mean = ms[(int64_t)b * 2 * ct + r];
    float scale = ms[(int64_t)b * 2 * ct + ct + r];
    float sp = scale > 20.f ? scale : log1pf(expf(scale));   // F.softplus (beta 1, threshold 20)
    z[i] = noise[i] * (sp + 1e-4f) + mean;
}

// utils/audio_utils.py:21-26
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t n, unsigned* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    float m = 0.f;
    for (; i < n; i += stride) m = fmaxf(m, fabsf(x[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));   // non-negative floats order like uints
}
__global__ __launch_bounds__(256) void to_int16_kernel(const float* __restrict__ x, int16_t* __restrict__ y, int64_t n,
                                                       const unsigned* __restrict__ peak_bits, int maximize) {
    float div = __uint_as_float(*peak_bits);
    if (!maximize) div = fmaxf(div, 1.0f);
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < n; i += stride) y[i] = (int16_t)((x[i] / div) * 32767.0f);   // truncation toward zero
}

// models/blocks.py:318-319 -- standalone SnakeBeta (the VAE kernels fuse it on load)
__global__ __launch_bounds__(256) void snake_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                    const float* __restrict__ beta, float* __restrict__ y, int C, int T,
                                                    int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int c = (int)((i / T) % C);
    float a = expf(alpha[c]), ib = 1.0f / (expf(beta[c]) + 0.000000001f);
    float v = x[i];
    float s = sinf(v * a);
    y[i] = v + ib * (s * s);
}

}  // namespace

int glue_small_linear(const float* x, int ldx, const float* W, const float* bias, const float* add, int ldadd, void* y,
                      int ldy, int R, int N, int K, int act, int out16, hipStream_t s) {
    SAT_CHECK_ARG(x && W && y && R > 0 && N > 0 && K > 0 && K % 4 == 0 && ldx % 4 == 0, SAT_E_INVALID,
                  "small_linear: bad args R=%d N=%d K=%d", R, N, K);
    constexpr int RB = 8;
    dim3 grid(cdiv(N, 4), cdiv(R, RB));
    if (out16 == 1)
        hipLaunchKernelGGL((small_linear_kernel<RB, 1>), grid, dim3(256), 0, s, x, ldx, W, bias, add, ldadd, y, ldy, R, N, K, act);
    else if (out16 == 2)
        hipLaunchKernelGGL((small_linear_kernel<RB, 2>), grid, dim3(256), 0, s, x, ldx, W, bias, add, ldadd, y, ldy, R, N, K, act);
    else
        hipLaunchKernelGGL((small_linear_kernel<RB, 0>), grid, dim3(256), 0, s, x, ldx, W, bias, add, ldadd, y, ldy, R, N, K, act);
    SAT_LAUNCH_CHECK();
    return 0;
}

int glue_adaln_finish(float* ssg, int64_t n, int D, hipStream_t s) {
    hipLaunchKernelGGL(adaln_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ssg, n, D);
    SAT_LAUNCH_CHECK();
    return 0;
}

int glue_fourier(const float* t, float t_const, const float* w, float* out, int B, int half_feat, hipStream_t s) {
    hipLaunchKernelGGL(fourier_kernel, dim3(cdiv(B * half_feat, 256)), dim3(256), 0, s, t, t_const, w, out, B, half_feat);
    SAT_LAUNCH_CHECK();
    return 0;
}

int glue_fold_in(const float* Win, const float* Wpre, float* Weff, int D, int C, hipStream_t s) {
    hipLaunchKernelGGL(fold_in_kernel, dim3(cdiv(D * C, 256)), dim3(256), 0, s, Win, Wpre, Weff, D, C);
    SAT_LAUNCH_CHECK();
    return 0;
}
int glue_fold_out(const float* Wout, const float* Wpost, float* Weff, int D, int C, hipStream_t s) {
    hipLaunchKernelGGL(fold_out_kernel, dim3(cdiv(D * C, 256)), dim3(256), 0, s, Wout, Wpost, Weff, D, C);
    SAT_LAUNCH_CHECK();
    return 0;
}

int glue_input_proj(const float* x, const float* Weff, float* X, int Bf, int xB, int C, int T, int S, int D, float xscale,
                    hipStream_t s) {
    SAT_CHECK_ARG(C <= 64, SAT_E_UNSUPPORTED, "input_proj: io_channels %d > 64", C);
    hipLaunchKernelGGL(input_proj_kernel, dim3(cdiv(T, IP_TT), cdiv(D, 256), Bf), dim3(256), 0, s, x, Weff, X, xB, C, T, S, D, xscale);
    SAT_LAUNCH_CHECK();
    return 0;
}

int glue_output_proj(const float* X, const float* WeffT, float* out, int Bf, int C, int T, int S, int D, hipStream_t s) {
    SAT_CHECK_ARG(C <= 64 && C % 4 == 0 && D % 16 == 0, SAT_E_UNSUPPORTED, "output_proj: C=%d (multiple of 4, <= 64) D=%d unsupported", C, D);
    const int lds = (OP_TOK * D + 8 * OP_TOK * 64) * 4;
    SAT_CHECK_ARG(lds <= 160 * 1024 && D % 128 == 0, SAT_E_UNSUPPORTED, "output_proj: embed_dim %d unsupported (multiple of 128, <= 2048)", D);
    SAT_TRY(sat_ensure_dynamic_lds(reinterpret_cast<const void*>(output_proj_kernel), 160 * 1024));
    hipLaunchKernelGGL(output_proj_kernel, dim3(cdiv((int64_t)Bf * T, OP_TOK)), dim3(512), lds, s, X, WeffT, out, Bf, C, T, S, D);
    SAT_LAUNCH_CHECK();
    return 0;
}

int glue_cfg_denoise(const float* mo, const float* x, float* den, int B, int C, int T, int use_cfg, float cfg_scale,
                     float scale_phi, float c_out, float c_skip, hipStream_t s) {
    if (!(use_cfg && scale_phi != 0.0f)) {
        const int64_t n = (int64_t)B * C * T;
        hipLaunchKernelGGL(cfg_denoise_ew_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, mo, x, den, n, use_cfg, cfg_scale,
                           c_out, c_skip);
        SAT_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(cfg_denoise_kernel, dim3(cdiv((int64_t)B * T, 256)), dim3(256), 0, s, mo, x, den, B, C, T, use_cfg,
                       cfg_scale, scale_phi, c_out, c_skip);
    SAT_LAUNCH_CHECK();
    return 0;
}

// out <- sum_i c_i * t_i over up to five terms (NULL terms are skipped; any t_i may alias out): the state update of the
// single-step k-diffusion samplers (Euler / Heun / DPM-2 / LMS / DPM++ 2S ancestral / DPM-Solver fast) and of the
// rectified-flow Euler loop, which are all linear combinations of the state, denoiser outputs and noise
__global__ __launch_bounds__(256) void lincomb_kernel(float* __restrict__ out, const float* t0, const float* t1, const float* t2,
                                                      const float* t3, const float* t4, float c0, float c1, float c2, float c3,
                                                      float c4, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float v = 0.f;
        if (t0) v += c0 * t0[i];
        if (t1) v += c1 * t1[i];
        if (t2) v += c2 * t2[i];
        if (t3) v += c3 * t3[i];
        if (t4) v += c4 * t4[i];
        out[i] = v;
    }
}

extern "C" int sat_lincomb(float* out_dev, const float* t0, float c0, const float* t1, float c1, const float* t2, float c2,
                           const float* t3, float c3, const float* t4, float c4, int64_t n, sat_stream_t stream) {
    SAT_CHECK_ARG(out_dev && n > 0, SAT_E_INVALID, "lincomb: null output or n <= 0");
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(lincomb_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out_dev, t0, t1, t2, t3, t4, c0, c1, c2, c3,
                       c4, n);
    SAT_LAUNCH_CHECK();
    return 0;
}

// DPM-Solver adaptive (k-diffusion dpm_solver_adaptive, selected at sampling.py:222-224): per-block partial sums of
// ((x_low - x_high) / max(atol, rtol * max(|x_low|, |x_prev|)))^2; the host adds the (fixed number of) partials in float64, so the
// accept / reject decision is reproducible
__global__ __launch_bounds__(256) void dpm_error_kernel(const float* __restrict__ x_low, const float* __restrict__ x_high,
                                                        const float* __restrict__ x_prev, float atol, float rtol, int64_t n,
                                                        float* __restrict__ partial) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float lo = x_low[i];
        const float delta = fmaxf(atol, rtol * fmaxf(fabsf(lo), fabsf(x_prev[i])));
        const float e = (lo - x_high[i]) / delta;
        acc += e * e;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

extern "C" int sat_dpm_error_partials(const float* x_low, const float* x_high, const float* x_prev, float atol, float rtol,
                                      int64_t n, float* partial_dev, int32_t n_partials, sat_stream_t stream) {
    SAT_CHECK_ARG(x_low && x_high && x_prev && partial_dev && n > 0 && n_partials > 0 && n_partials <= 4096, SAT_E_INVALID,
                  "dpm_error_partials: bad argument");
    hipLaunchKernelGGL(dpm_error_kernel, dim3(n_partials), dim3(256), 0, (hipStream_t)stream, x_low, x_high, x_prev, atol, rtol, n,
                       partial_dev);
    SAT_LAUNCH_CHECK();
    return 0;
}

// inference/sampling.py:178-190 (inpainting_callback): keep-region of the current step's binary mask is re-noised init data
__global__ __launch_bounds__(256) void inpaint_mix_kernel(float* __restrict__ x, const float* __restrict__ init,
                                                          const float* __restrict__ noise, const float* __restrict__ mask,
                                                          float sigma, float strength, int64_t n, int T) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (mask[i % T] <= strength) x[i] = init[i] + noise[i] * sigma;
    }
}

extern "C" int sat_inpaint_mix(float* x_dev, const float* init_dev, const float* noise_dev, const float* mask_dev, float sigma,
                               float strength, int64_t rows, int32_t t, sat_stream_t stream) {
    SAT_CHECK_ARG(x_dev && init_dev && noise_dev && mask_dev && rows > 0 && t > 0, SAT_E_INVALID, "inpaint_mix: bad argument");
    const int64_t n = rows * t;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(inpaint_mix_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x_dev, init_dev, noise_dev, mask_dev,
                       sigma, strength, n, t);
    SAT_LAUNCH_CHECK();
    return 0;
}

extern "C" int sat_dpmpp3m_update(float* x_dev, const float* d_dev, const float* d1_dev, const float* d2_dev,
                                  const float* noise_dev, float a, float b, float c1, float c2, float cn, int64_t n,
                                  sat_stream_t stream) {
    SAT_CHECK_ARG(x_dev && d_dev && n > 0, SAT_E_INVALID, "dpmpp3m_update: null state or n <= 0");
    SAT_CHECK_ARG(!(c1 != 0.f && !d1_dev) && !(c2 != 0.f && !d2_dev) && !(cn != 0.f && !noise_dev), SAT_E_INVALID,
                  "dpmpp3m_update: non-zero coefficient with a NULL tensor");
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(dpmpp3m_update_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x_dev, d_dev,
                       c1 != 0.f || c2 != 0.f ? d1_dev : nullptr, c2 != 0.f ? d2_dev : nullptr, cn != 0.f ? noise_dev : nullptr, a,
                       b, c1, c2, cn, n);
    SAT_LAUNCH_CHECK();
    return 0;
}

extern "C" int sat_vae_sample(const float* mean_scale_dev, const float* noise_dev, float* z_dev, int32_t b, int32_t c,
                              int32_t t, sat_stream_t stream) {
    SAT_CHECK_ARG(mean_scale_dev && noise_dev && z_dev && b > 0 && c > 0 && t > 0, SAT_E_INVALID, "vae_sample: bad args");
    int64_t n = (int64_t)b * c * t;
    hipLaunchKernelGGL(vae_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mean_scale_dev,
                       noise_dev, z_dev, b, c, t);
    SAT_LAUNCH_CHECK();
    return 0;
}

// Overlap-add of the chunked codec (models/autoencoders.py:476-497, 548-571, 622-645): chunk i of length L starts at i * hop; its
// first `ov` samples are faded in (window[0 .. ov)) unless it is the first chunk, its last `ov` samples faded out (window[ov .. 2 ov))
// unless it is the last one; contributions are summed in chunk order.  One thread per output sample, gather form (no atomics).
__global__ __launch_bounds__(256) void overlap_add_kernel(const float* __restrict__ pieces, const float* __restrict__ window,
                                                          float* __restrict__ out, int n_chunk, int C, int L, int hop, int ov, int total,
                                                          int64_t n) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int t = (int)(idx % total);
    const int64_t bc = idx / total;
    const int c = (int)(bc % C);
    const int64_t b = bc / C;
    int i_hi = t / hop;
    if (i_hi > n_chunk - 1) i_hi = n_chunk - 1;
    int i_lo = t - L + 1 <= 0 ? 0 : (t - L + hop) / hop;           // ceil((t - L + 1) / hop)
    float acc = 0.f;
    for (int i = i_lo; i <= i_hi; ++i) {
        const int off = t - i * hop;
        if (off < 0 || off >= L) continue;
        float v = pieces[((b * n_chunk + i) * C + c) * (int64_t)L + off];
        if (i != 0 && off < ov) v *= window[off];
        if (i != n_chunk - 1 && off >= L - ov) v *= window[ov + off - (L - ov)];
        acc += v;
    }
    out[idx] = acc;
}

extern "C" int sat_overlap_add(const float* pieces_dev, const float* window_dev, float* out_dev, int32_t batch, int32_t n_chunk,
                               int32_t channels, int32_t chunk_len, int32_t hop, int32_t overlap, int32_t total_len, sat_stream_t stream) {
    SAT_CHECK_ARG(pieces_dev && window_dev && out_dev, SAT_E_INVALID, "overlap_add: null pointer");
    SAT_CHECK_ARG(batch > 0 && n_chunk > 0 && channels > 0 && chunk_len > 0 && hop > 0 && overlap >= 0 && overlap <= chunk_len && total_len > 0,
                  SAT_E_INVALID, "overlap_add: bad dims");
    const int64_t n = (int64_t)batch * channels * total_len;
    hipLaunchKernelGGL(overlap_add_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pieces_dev, window_dev, out_dev, n_chunk, channels,
                       chunk_len, hop, overlap, total_len, n);
    SAT_LAUNCH_CHECK();
    return 0;
}

// Sample-rate conversion in front of the encoder (the reference's torchaudio.transforms.Resample at inference/utils.py:25-27,
// models/autoencoders.py:394-397, reconstruct_audios.py:34-35): polyphase windowed-sinc FIR.  With orig / new reduced by their gcd,
// output sample j = frame * new + phase is the dot product of taps = 2 * width + orig input samples starting at frame * orig - width
// (zeros outside the signal) with row `phase` of the filter bank.  The bank is built by the host side exactly as torchaudio builds
// it (float64 -> float32; stable_audio_tools/inference/resample.py); this kernel is the strided convolution.
// One thread per output sample; consecutive threads walk the phases of one frame, so the input window is shared through L1/L2 and the
// bank (147 x 174 floats for 48 kHz -> 44.1 kHz) stays cache-resident.  ~0.7 GFLOP per 47-s stereo clip: nowhere near any roofline.
__global__ __launch_bounds__(256) void resample_sinc_kernel(const float* __restrict__ x, const float* __restrict__ bank, float* __restrict__ y,
                                                            int in_len, int out_len, int orig, int newr, int width, int taps, long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int row = (int)(idx / out_len);            // batch * channels row
    const int j = (int)(idx - (long)row * out_len);
    const int frame = j / newr, phase = j - frame * newr;
    const float* __restrict__ xr = x + (size_t)row * in_len;
    const float* __restrict__ br = bank + (size_t)phase * taps;
    const int start = frame * orig - width;
    float acc = 0.f;
    for (int k = 0; k < taps; ++k) {                 // ascending tap order = the accumulation order of the oracle's reference loop
        const int i = start + k;
        const float v = (i >= 0 && i < in_len) ? xr[i] : 0.f;
        acc = fmaf(v, br[k], acc);
    }
    y[idx] = acc;
}

extern "C" int sat_resample_sinc(const float* x_dev, const float* bank_dev, float* y_dev, int32_t rows, int32_t in_len, int32_t out_len,
                                 int32_t orig, int32_t newr, int32_t width, sat_stream_t stream) {
    SAT_CHECK_ARG(x_dev && bank_dev && y_dev, SAT_E_INVALID, "resample: null pointer");
    SAT_CHECK_ARG(rows > 0 && in_len > 0 && out_len > 0 && orig > 0 && newr > 0 && width >= 0, SAT_E_INVALID, "resample: bad dims");
    SAT_CHECK_ARG((int64_t)out_len <= ((int64_t)in_len * newr + orig - 1) / orig, SAT_E_INVALID,
                  "resample: out_len %d exceeds ceil(in_len * new / orig) = %lld", out_len, (long long)(((int64_t)in_len * newr + orig - 1) / orig));
    const long total = (long)rows * out_len;
    hipLaunchKernelGGL(resample_sinc_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x_dev, bank_dev, y_dev, in_len, out_len,
                       orig, newr, width, 2 * width + orig, total);
    SAT_LAUNCH_CHECK();
    return 0;
}

// NumberConditioner (models/conditioners.py:64-102) = clamp -> (x - min) / (max - min) -> NumberEmbedder (models/adp.py:1495-1514):
// LearnedPositionalEmbedding cat(x, sin(2 pi x w), cos(2 pi x w)) (adp.py:680-694) -> Linear(2*half + 1, features).  A handful of
// scalars per generation: one workgroup per value, the 2*half + 1 features in LDS, one output channel per thread and pass.
__global__ __launch_bounds__(256) void number_embed_kernel(const float* __restrict__ values, float vmin, float vmax,
                                                           const float* __restrict__ w, const float* __restrict__ lw,
                                                           const float* __restrict__ lb, float* __restrict__ out, int half_dim,
                                                           int features) {
    extern __shared__ float feat[];            // [2 * half_dim + 1]
    const int b = blockIdx.x;
    const float x = (fminf(fmaxf(values[b], vmin), vmax) - vmin) / (vmax - vmin);
    if (threadIdx.x == 0) feat[0] = x;
    for (int i = threadIdx.x; i < half_dim; i += 256) {
        const float f = x * w[i] * 2.0f * 3.14159265358979323846f;      // torch: ((x * w) * 2) * pi in fp32
        feat[1 + i] = sinf(f);
        feat[1 + half_dim + i] = cosf(f);
    }
    __syncthreads();
    const int k = 2 * half_dim + 1;
    for (int n = threadIdx.x; n < features; n += 256) {
        const float* wr = lw + (size_t)n * k;
        float acc = 0.f;
        for (int i = 0; i < k; ++i) acc = fmaf(feat[i], wr[i], acc);
        out[(size_t)b * features + n] = acc + lb[n];
    }
}

extern "C" int sat_number_embed(const float* values_dev, int32_t count, float min_val, float max_val, const float* pos_weights_dev,
                                int32_t half_dim, const float* linear_w_dev, const float* linear_b_dev, int32_t features,
                                float* out_dev, sat_stream_t stream) {
    SAT_CHECK_ARG(values_dev && pos_weights_dev && linear_w_dev && linear_b_dev && out_dev, SAT_E_INVALID, "number_embed: null pointer");
    SAT_CHECK_ARG(count > 0 && half_dim > 0 && half_dim <= 4096 && features > 0 && max_val > min_val, SAT_E_INVALID, "number_embed: bad dims");
    hipLaunchKernelGGL(number_embed_kernel, dim3(count), dim3(256), (2 * half_dim + 1) * sizeof(float), (hipStream_t)stream, values_dev, min_val,
                       max_val, pos_weights_dev, linear_w_dev, linear_b_dev, out_dev, half_dim, features);
    SAT_LAUNCH_CHECK();
    return 0;
}

extern "C" int sat_float_to_int16(const float* x_dev, int16_t* out_dev, int64_t n, int32_t maximize, void* scratch_dev,
                                  sat_stream_t stream) {
    SAT_CHECK_ARG(x_dev && out_dev && scratch_dev && n > 0, SAT_E_INVALID, "float_to_int16: bad args");
    hipStream_t s = (hipStream_t)stream;
    SAT_HIP(hipMemsetAsync(scratch_dev, 0, 4, s));
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, s, x_dev, n, (unsigned*)scratch_dev);
    hipLaunchKernelGGL(to_int16_kernel, dim3(blocks), dim3(256), 0, s, x_dev, out_dev, n, (const unsigned*)scratch_dev, maximize);
    SAT_LAUNCH_CHECK();
    return 0;
}

extern "C" int sat_snake_beta(const float* x_dev, const float* alpha_dev, const float* beta_dev, float* y_dev, int32_t b,
                              int32_t c, int32_t t, sat_stream_t stream) {
    SAT_CHECK_ARG(x_dev && alpha_dev && beta_dev && y_dev && b > 0 && c > 0 && t > 0, SAT_E_INVALID, "snake: bad args");
    int64_t n = (int64_t)b * c * t;
    hipLaunchKernelGGL(snake_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_dev, alpha_dev,
                       beta_dev, y_dev, c, t, n);
    SAT_LAUNCH_CHECK();
    return 0;
}

// ---- sat_dit_debug: what the 16-bit image of the residual stream would do to these rows (one wave per row)
namespace {
__global__ __launch_bounds__(256) void resid_stats_kernel(const float* __restrict__ X, int M, int D, float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* x = X + (size_t)row * D;
    float sum = 0.f, sq = 0.f, mx = 0.f, over = 0.f;
    for (int i = lane; i < D; i += 64) {
        const float v = x[i], a = fabsf(v);
        sum += v;
        sq += v * v;
        mx = fmaxf(mx, a);
        over += a > 65504.0f ? 1.f : 0.f;
    }
    sum = wave_sum(sum);
    sq = wave_sum(sq);
    over = wave_sum(over);
    mx = wave_max(mx);
    if (lane == 0) {
        const float mean = sum / (float)D;
        const float var = fmaxf(sq / (float)D - mean * mean, 0.f);
        const float cm = fabsf(mean) * rsqrtf(var + 1e-30f);
        const float crest = mx * rsqrtf(sq / (float)D + 1e-30f);
        // non-negative floats order like their bit patterns
        atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(mx));
        atomicMax(reinterpret_cast<unsigned*>(out + 1), __float_as_uint(cm));
        if (over > 0.f) atomicAdd(out + 2, over);
        atomicMax(reinterpret_cast<unsigned*>(out + 3), __float_as_uint(crest));
    }
}
}  // namespace

int glue_resid_stats(const float* X, int M, int D, float* out, hipStream_t s) {
    hipLaunchKernelGGL(resid_stats_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, X, M, D, out);
    SAT_LAUNCH_CHECK();
    return 0;
}

