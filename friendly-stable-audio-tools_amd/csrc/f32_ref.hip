// fp32 verification path of the DiT ("gemm_dtype = fp32x", sat_dit_cfg.fp8_gemm == 2): the same plan, data flow, index
// arithmetic, RoPE table, prepend token, GQA cross-attention and CFG batching as the bf16 path, but every contraction takes
// fp32 operands on the exact fp32 MFMA (v_mfma_f32_32x32x2_f32 = fmaf chains) and q / k / v / P stay fp32.  NOT a fast path
// (the f32 MFMA runs at 1/16 of the bf16 rate): it exists to show that what separates the bf16 path from the reference
// (3.5e-3 at full size) is operand rounding only -- this path meets north_star's 1e-3 against the reference's own outputs
// (tests/test_gpu_models.py: test_fp32x_mode_vs_reference_golden).
// Replaces the same reference code as gemm_bf16.hip / attention.hip / layernorm.hip (models/transformer.py:188-206, 158-183,
// 222-235, 270, 311-319, 496-536).
#include "sat_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// C[M,N] (+)= A[M,K] . W[N,K]^T (+ bias) (* gate): 128x128 tile, 4 waves (2x2) of 64x64, K-step 16, register-staged double
// buffer.  LDS rows hold 16 floats padded to 20 (80 B): conflict-free ds_read_b128.  A lane half h takes k = 8h .. 8h+7 of the
// K-step for BOTH operands (any k permutation shared by A and W leaves the dot product unchanged).
constexpr int F_BM = 128, F_BN = 128, F_BK = 16, F_LD = 20;

__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* __restrict__ C, int M, int N, int K,
                                                       int ldc, int accumulate, const float* __restrict__ gate, int gate_rows,
                                                       int gate_ld) {
    __shared__ __attribute__((aligned(16))) float sA[2][F_BM * F_LD];
    __shared__ __attribute__((aligned(16))) float sW[2][F_BN * F_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    const int tiles_n = (N + F_BN - 1) / F_BN;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * F_BM, n0 = tn * F_BN;

    // staging: 128 rows x 4 float4 per operand per K-step = 512 float4, two per thread
    int srow[2], sc4[2];
    const float *ap[2], *wp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = i * 256 + tid;
        srow[i] = id >> 2;
        sc4[i] = id & 3;
        int gm = m0 + srow[i];
        gm = gm < M ? gm : M - 1;
        int gn = n0 + srow[i];
        gn = gn < N ? gn : N - 1;
        ap[i] = A + (size_t)gm * K + sc4[i] * 4;
        wp[i] = W + (size_t)gn * K + sc4[i] * 4;
    }
    f32x4 ra[2], rw[2];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ra[i] = *reinterpret_cast<const f32x4*>(ap[i] + kt * F_BK);
            rw[i] = *reinterpret_cast<const f32x4*>(wp[i] + kt * F_BK);
        }
    };
    auto lstore = [&](int st) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<f32x4*>(&sA[st][srow[i] * F_LD + sc4[i] * 4]) = ra[i];
            *reinterpret_cast<f32x4*>(&sW[st][srow[i] * F_LD + sc4[i] * 4]) = rw[i];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto compute = [&](int st) {
        f32x4 af[2][2], wf[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                af[i][q] = *reinterpret_cast<const f32x4*>(&sA[st][(wm * 64 + i * 32 + l31) * F_LD + half * 8 + q * 4]);
                wf[i][q] = *reinterpret_cast<const f32x4*>(&sW[st][(wn * 64 + i * 32 + l31) * F_LD + half * 8 + q * 4]);
            }
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t >> 2][t & 3], wf[j][t >> 2][t & 3], acc[i][j], 0, 0, 0);
    };
    const int nk = K / F_BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk - 1; ++kt) {
        gload(kt + 1);
        compute(kt & 1);
        lstore((kt + 1) & 1);
        __syncthreads();
    }
    compute((nk - 1) & 1);
    // acc[i][j][r]: row = i*32 + (r&3) + 8*(r>>2) + 4*half, col = j*32 + l31
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
            if (n >= N) continue;
            const float bv = bias ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m >= M) continue;
                float v = acc[i][j][r] + bv;
                if (gate) v *= gate[(size_t)(m / gate_rows) * gate_ld + n];
                float* cp = C + (size_t)m * ldc + n;
                *cp = accumulate ? *cp + v : v;
            }
        }
}

// LayerNorm -> fp32 (one wave per row, any d % 4 == 0; optional adaLN modulation)
__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y, int m, int d,
                                                            const float* __restrict__ sc, const float* __restrict__ sh, int rps, int ld) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    const float* xr = x + (size_t)row * d;
    float s = 0.f;
    for (int i = lane; i < d; i += 64) s += xr[i];
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
    for (int i = lane; i < d; i += 64) {
        const float a = xr[i] - mean;
        q += a * a;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + 1e-5f);
    const float* scr = sc ? sc + (size_t)(row / rps) * ld : nullptr;
    const float* shr = sc ? sh + (size_t)(row / rps) * ld : nullptr;
    for (int i = lane; i < d; i += 64) {
        float r = (xr[i] - mean) * rstd * gamma[i] + (beta ? beta[i] : 0.f);
        if (scr) r = r * scr[i] + shr[i];
        y[(size_t)row * d + i] = r;
    }
}

// [M, parts * H * 64] -> per part [B, H, S, 64] fp32; parts with rope: partial rotary on d < 32 (pairs d, d + 16)
__global__ __launch_bounds__(256) void split_heads_f32_kernel(const float* __restrict__ src, float* __restrict__ d0, float* __restrict__ d1,
                                                              float* __restrict__ d2, int M, int S, int parts, int H, int rope_mask,
                                                              const float* __restrict__ rope_cos, const float* __restrict__ rope_sin) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)M * parts * H * 64;
    if (idx >= total) return;
    const int d = (int)(idx & 63);
    int64_t t = idx >> 6;
    const int h = (int)(t % H);
    t /= H;
    const int part = (int)(t % parts);
    const int m = (int)(t / parts);
    const int b = m / S, s = m - b * S;
    const float* row = src + (size_t)m * parts * H * 64 + (size_t)part * H * 64 + h * 64;
    float v = row[d];
    if (((rope_mask >> part) & 1) && d < 32) {
        const float cs = rope_cos[(size_t)s * 16 + (d & 15)], sn = rope_sin[(size_t)s * 16 + (d & 15)];
        v = d < 16 ? v * cs - row[d + 16] * sn : v * cs + row[d - 16] * sn;
    }
    float* dst = part == 0 ? d0 : (part == 1 ? d1 : d2);
    dst[(((size_t)b * H + h) * S + s) * 64 + d] = v;
}

// x, gate = chunk(2); x * silu(gate)  (transformer.py:232-235), fp32
__global__ __launch_bounds__(256) void swiglu_f32_kernel(const float* __restrict__ hg, float* __restrict__ h, int64_t M, int inner) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * inner) return;
    const int64_t m = idx / inner;
    const int c = (int)(idx - m * inner);
    const float v = hg[m * 2 * inner + c], g = hg[m * 2 * inner + inner + c];
    h[idx] = v * (g / (1.0f + expf(-g)));
}

// softmax(q k^T / 8) v in fp32, one query per lane (q and the output row in registers), K / V rows are wave-uniform loads.
// Online softmax over chunks of 8 keys.  q [B,H,Sq,64], k / v [B,KVH,Sk,64] -> out [B*Sq, H*64].
__global__ __launch_bounds__(64) void attention_f32_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, float* __restrict__ out, int H, int KVH, int Sq,
                                                           int Sk) {
    const int b = blockIdx.z, h = blockIdx.y;
    const int kvh = h / (H / KVH);
    const int qi = blockIdx.x * 64 + threadIdx.x;
    const int qc = qi < Sq ? qi : Sq - 1;
    float qr[64], o[64];
    const f32x4* qp = reinterpret_cast<const f32x4*>(q + (((size_t)b * H + h) * Sq + qc) * 64);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const f32x4 x = qp[t];
        qr[4 * t] = x[0] * 0.125f;
        qr[4 * t + 1] = x[1] * 0.125f;
        qr[4 * t + 2] = x[2] * 0.125f;
        qr[4 * t + 3] = x[3] * 0.125f;
    }
#pragma unroll
    for (int d = 0; d < 64; ++d) o[d] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float* kb = k + ((size_t)b * KVH + kvh) * Sk * 64;
    const float* vb = v + ((size_t)b * KVH + kvh) * Sk * 64;
    for (int j0 = 0; j0 < Sk; j0 += 8) {
        float sc[8];
        float mx = m_run;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + u;
            float s = -INFINITY;
            if (j < Sk) {
                const float* kr = kb + (size_t)j * 64;
                s = 0.f;
#pragma unroll
                for (int d = 0; d < 64; ++d) s = fmaf(qr[d], kr[d], s);
            }
            sc[u] = s;
            mx = fmaxf(mx, s);
        }
        const float alpha = expf(m_run - mx);
        m_run = mx;
        l_run *= alpha;
#pragma unroll
        for (int d = 0; d < 64; ++d) o[d] *= alpha;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + u;
            if (j < Sk) {
                const float p = expf(sc[u] - mx);
                l_run += p;
                const float* vr = vb + (size_t)j * 64;
#pragma unroll
                for (int d = 0; d < 64; ++d) o[d] = fmaf(p, vr[d], o[d]);
            }
        }
    }
    if (qi < Sq) {
        const float inv = 1.0f / l_run;
        f32x4* op = reinterpret_cast<f32x4*>(out + ((size_t)b * Sq + qi) * ((size_t)H * 64) + h * 64);
#pragma unroll
        for (int t = 0; t < 16; ++t) op[t] = f32x4{o[4 * t] * inv, o[4 * t + 1] * inv, o[4 * t + 2] * inv, o[4 * t + 3] * inv};
    }
}

}  // namespace

int sat_launch_gemm_f32(const float* A, const float* W, const float* bias, float* C, int M, int N, int K, int ldc, int accumulate,
                        const float* gate, int gate_rows, int gate_ld, hipStream_t s) {
    SAT_CHECK_ARG(A && W && C && M > 0 && N > 0 && K > 0, SAT_E_INVALID, "gemm_f32: bad arguments");
    SAT_CHECK_ARG(K % F_BK == 0, SAT_E_UNSUPPORTED, "gemm_f32: K=%d must be a multiple of %d", K, F_BK);
    const int tiles = cdiv(M, F_BM) * cdiv(N, F_BN);
    hipLaunchKernelGGL(gemm_f32_kernel, dim3(tiles), dim3(256), 0, s, A, W, bias, C, M, N, K, ldc, accumulate, gate, gate_rows > 0 ? gate_rows : 1,
                       gate_ld);
    SAT_LAUNCH_CHECK();
    return 0;
}

int sat_launch_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y, int m, int d, const float* sc,
                             const float* sh, int rps, int ld, hipStream_t s) {
    SAT_CHECK_ARG(x && gamma && y && m > 0 && d > 0, SAT_E_INVALID, "layernorm_f32: bad args");
    hipLaunchKernelGGL(layernorm_f32_kernel, dim3(cdiv(m, 4)), dim3(256), 0, s, x, gamma, beta, y, m, d, sc, sh, rps > 0 ? rps : 1, ld);
    SAT_LAUNCH_CHECK();
    return 0;
}

int sat_launch_split_heads_f32(const float* src, float* d0, float* d1, float* d2, int M, int S, int parts, int H, int rope_mask,
                               const float* rope_cos, const float* rope_sin, hipStream_t s) {
    SAT_CHECK_ARG(src && d0 && parts >= 1 && parts <= 3 && (rope_mask == 0 || (rope_cos && rope_sin)), SAT_E_INVALID, "split_heads_f32: bad args");
    const int64_t total = (int64_t)M * parts * H * 64;
    hipLaunchKernelGGL(split_heads_f32_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, src, d0, d1, d2, M, S, parts, H, rope_mask, rope_cos,
                       rope_sin);
    SAT_LAUNCH_CHECK();
    return 0;
}

int sat_launch_swiglu_f32(const float* hg, float* h, int64_t M, int inner, hipStream_t s) {
    hipLaunchKernelGGL(swiglu_f32_kernel, dim3(cdiv(M * inner, 256)), dim3(256), 0, s, hg, h, M, inner);
    SAT_LAUNCH_CHECK();
    return 0;
}

int sat_launch_attention_f32(const float* q, const float* k, const float* v, float* out, int b, int h, int kvh, int sq, int sk,
                             hipStream_t s) {
    SAT_CHECK_ARG(q && k && v && out && b > 0 && h > 0 && kvh > 0 && h % kvh == 0 && sq > 0 && sk > 0, SAT_E_INVALID, "attention_f32: bad args");
    hipLaunchKernelGGL(attention_f32_kernel, dim3(cdiv(sq, 64), h, b), dim3(64), 0, s, q, k, v, out, h, kvh, sq, sk);
    SAT_LAUNCH_CHECK();
    return 0;
}
