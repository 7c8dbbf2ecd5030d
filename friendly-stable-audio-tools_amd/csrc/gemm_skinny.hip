// Skinny GEMM for the M-tail of the block GEMMs (round 6):   C[Mt, N] = A[Mt, K] . W[N, K]^T   with Mt = a few rows (16 per workgroup row).
// The block GEMMs of the DiT run M = bf * 1025 rows = whole 256-row tiles + 2 (one prompt) / 16 (eight prompts) rows, and the near-empty extra
// row of tiles costs the big-tile kernels 4-10 % at eight prompts (profiles/r06_mtail_pricing.txt): every column tile streams its whole W panel
// through the LDS ring of one compute unit for 16 rows.  Those rows are ordinary tokens -- nn.Linear is per row (models/transformer.py:222, 270, 319)
// -- so the plan (dit_plan.hip, launch2) runs the big tiles on the rows of the whole tiles and THIS kernel on the rest: no LDS ring, no barriers in
// the K loop -- a weight-streaming kernel.
//   workgroup = 16 rows x 64 columns, 8 (16) waves, wave w takes the k-range [w K/8, (w+1) K/8): per 32 k one 16-byte load of the lane's A row piece
//   and four of its W row pieces (v_mfma_f32_16x16x32 fragments straight from global memory), six k-steps = 30 loads in flight per lane;
//   the partial 16 x 64 blocks meet in LDS (32 / 64 KiB) and wave 0 runs the epilogue on the sums.
//   N / 64 workgroups per 16 rows: FF-in 192, to_qkv 72, FF-out / to_out 24 -- every W row is read exactly once, 64 contiguous bytes per lane quad.
// Epilogues = those of the big tiles on TRANSPOSED accumulators (weight fragment = MFMA A operand: lane (l15, q4) holds token row l15 and channels
// 16 nb + 4 q4 + r of block nb): SwiGLU (value / gate = blocks nb / nb + 2 of the same lane) and the fp32 residual update with the LayerNorm-fold
// producer (16-bit image + (sum, sum of squares) of the rounded values over the workgroup's 64 columns), both with the LayerNorm-fold consumer
// constants where the GEMM sits behind a LayerNorm.  The heads epilogue (RoPE, q / k / v^T layouts) is not built: to_qkv gains 1 % from whole tiles.
#include "sat_common.h"

namespace {

// Epilogue of one 16 x 64 block on its summed accumulators (lane (l15, q4): token row m0 + l15, channels n0 + 16 nb + 4 q4 + r): LayerNorm-fold
// consumer, then SwiGLU or the fp32 residual update with the LayerNorm-fold producer
template <int EPI>
__device__ __forceinline__ void skinny_epilogue(const GemmArgs& g, f32x4 (&acc)[4], const int m0, const int n0, const int lane) {
    const int l15 = lane & 15, q4 = lane >> 4;
    const int M = g.M, N = g.N, K = g.K;
    const int m = m0 + l15;
    const int mc = m < M ? m : M - 1;
    // ---- LayerNorm fold, consumer side (GemmArgs::ln_part): (mean, rstd) of the lane's token row from the producer's per-64-column partial sums;
    //      the four lanes of a token (q4 = 0..3) each add every fourth pair
    float mean = 0.f, rstd = 1.f;
    const bool fold = g.ln_part != nullptr;
    if (fold) {
        const int np = K >> 6;
        const float2* pp = reinterpret_cast<const float2*>(g.ln_part) + (size_t)mc * np;
        float sum = 0.f, sq = 0.f;
        for (int i = q4; i < np; i += 4) {
            const float2 v = pp[i];
            sum += v.x;
            sq += v.y;
        }
        sum += __shfl_xor(sum, 16, 64);
        sq += __shfl_xor(sq, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        sq += __shfl_xor(sq, 32, 64);
        const float inv_k = 1.0f / (float)K;
        mean = sum * inv_k;
        rstd = rsqrtf(fmaxf(sq * inv_k - mean * mean, 0.f) + g.ln_eps);
    }
    // x = rstd * (acc - mean * c1) + c2 with (c1, c2) = (rowsum(gamma W), W beta + b) under the fold, (0, bias) without it
    f32x4 x[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int ch = n0 + 16 * nb + 4 * q4;
        f32x4 c1 = {0.f, 0.f, 0.f, 0.f}, c2 = {0.f, 0.f, 0.f, 0.f};
        if (fold) {
            c1 = *reinterpret_cast<const f32x4*>(g.ln_c1 + ch);
            c2 = *reinterpret_cast<const f32x4*>(g.ln_c2 + ch);
        } else if (g.bias) {
            c2 = *reinterpret_cast<const f32x4*>(g.bias + ch);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) x[nb][e] = rstd * (acc[nb][e] - mean * c1[e]) + c2[e];
    }
    if constexpr (EPI == EPI_SWIGLU) {
        // packed FF-in rows: columns [n0, n0 + 32) are values, [n0 + 32, n0 + 64) their gates (models/transformer.py:232-235)
        if (m < M) {
            op_t* hrow = g.H + (size_t)m * (N >> 1) + (n0 >> 1) + 4 * q4;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                float h[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = x[nb][e] * silu_f(x[nb + 2][e]);
                *reinterpret_cast<u32x2*>(hrow + 16 * nb) = u32x2{pack_op2(h[0], h[1]), pack_op2(h[2], h[3])};
            }
        }
    } else {
        // fp32 output / residual update (models/transformer.py:692-700) + LayerNorm-fold producer
        float sum = 0.f, sq = 0.f;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int ch = n0 + 16 * nb + 4 * q4;
            f32x4 v = x[nb];
            if (m < M) {
                float* crow = g.C + (size_t)m * g.ldc + ch;
                if (g.accumulate) v += *reinterpret_cast<const f32x4*>(crow);
                *reinterpret_cast<f32x4*>(crow) = v;
                if (g.xb) {
                    opx4 xr;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        xr[e] = f32_to_op(v[e]);
                        const float f = op_to_f32(xr[e]);
                        sum += f;
                        sq += f * f;
                    }
                    *reinterpret_cast<opx4*>(g.xb + (size_t)m * N + ch) = xr;
                }
            }
        }
        if (g.xb) {
            sum += __shfl_xor(sum, 16, 64);
            sq += __shfl_xor(sq, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            sq += __shfl_xor(sq, 32, 64);
            if (q4 == 0 && m < M) *reinterpret_cast<float2*>(g.ln_part_out + ((size_t)m * (N >> 6) + (n0 >> 6)) * 2) = make_float2(sum, sq);
        }
    }
}

// ---- LDS-staged variant (K a multiple of 512).  The direct variant below loads MFMA fragments lane-per-row: 16 rows x 64 bytes per wave
// instruction = 16 address requests, ~64 cycles of the address path each, 240 instructions per workgroup -- 12 us per launch whatever the batching
// (profiles/r06_mtail_split.txt).  Here a wave instruction moves ONE KiB OF ONE W ROW (512 k, fully coalesced) straight into LDS
// (global_load_lds_dwordx4): a chunk = 64 rows x 512 k = 64 instructions per workgroup, two chunks resident (rows 1056 bytes apart: the 16 lanes of a
// ds_read_b128 group land on 16 distinct 16-byte slots), one barrier pair per chunk.  Inside a chunk wave w takes k in [64 w, 64 w + 64): two k-steps,
// 8 ds_read_b128 + 8 MFMAs; its A fragments (lane-per-row, but only one per k-step) are fetched for the whole K up front -- the oldest entries of the
// vector-memory queue, so the counted waits on the chunks hold.
constexpr int SK2_ROW = 1056;                        // LDS bytes per W row of a chunk
constexpr int SK2_BUF = 64 * SK2_ROW;                // 67584
constexpr int SK2_LDS = 2 * SK2_BUF;                 // 132 KiB; the partial sums reuse buffer 0 afterwards
template <int EPI, int NCH>                          // NCH = K / 512 chunks (3: K = 1536; 12: K = 6144)
__global__ __launch_bounds__(512) void gemm_skinny_lds_kernel(GemmArgs g) {
    sat_f16_saturate();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q4 = lane >> 4;
    const int M = g.M, K = g.K;
    const int n0 = blockIdx.x * 64;
    const int m0 = blockIdx.y * 16;
    const int m = m0 + l15;
    const int mc = m < M ? m : M - 1;
    // A fragments of this wave for every chunk: k = 512 c + 64 w + 32 ks + 8 q4
    opx8 fa[NCH][2];
    {
        const op_t* ap = g.A + (size_t)mc * K + 64 * wave + 8 * q4;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[c][ks] = *reinterpret_cast<const opx8*>(ap + 512 * c + 32 * ks);
    }
    // W rows 8 w .. 8 w + 7 of the column group are this wave's to stage: lane l moves bytes [16 l, 16 l + 16) of the row's chunk
    const op_t* wsrc = g.W + (size_t)(n0 + 8 * wave) * K + 8 * lane;
    auto stage = [&](int c, int buf) {
        char* dst = smem + buf * SK2_BUF + (8 * wave) * SK2_ROW;
#pragma unroll
        for (int r = 0; r < 8; ++r)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (size_t)r * K + 512 * c),
                                             (__attribute__((address_space(3))) void*)(dst + r * SK2_ROW), 16, 0, 0);
    };
    f32x4 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    stage(0, 0);
    if (NCH > 1) stage(1, 1);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) wait_vmcnt<8>();          // chunk c + 1 (8 instructions of this wave) may stay in flight
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        const char* base = smem + (c & 1) * SK2_BUF + l15 * SK2_ROW + 128 * wave + 16 * q4;
        opx8 fw[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) fw[ks][nb] = *reinterpret_cast<const opx8*>(base + 16 * nb * SK2_ROW + 64 * ks);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[nb] = mfma_16x16x32(fw[ks][nb], fa[c][ks], acc[nb]);
        if (c + 2 < NCH) {
            __builtin_amdgcn_s_barrier();          // every wave has read buffer c & 1
            stage(c + 2, c & 1);
        }
    }
    // the eight partial blocks meet in LDS (buffer 0 is free: the last chunks' reads are behind the barrier below)
    __syncthreads();
    f32x4* part = reinterpret_cast<f32x4*>(smem);          // [wave][block][lane]
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) part[(wave * 4 + nb) * 64 + lane] = acc[nb];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll 4
    for (int w = 1; w < 8; ++w)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[nb] += part[(w * 4 + nb) * 64 + lane];
    skinny_epilogue<EPI>(g, acc, m0, n0, lane);
}

// NW waves share the reduction (K / NW each); U k-steps per batch: U x (1 + 4) 16-byte loads in flight per lane, then U x 4 MFMAs, no branch inside
template <int EPI, int NW, int U>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(GemmArgs g) {
    sat_f16_saturate();
    __shared__ f32x4 part[NW][4][64];                // [wave][block][lane]: 32 KiB (64 KiB with 16 waves)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q4 = lane >> 4;
    const int M = g.M, N = g.N, K = g.K;
    const int n0 = blockIdx.x * 64;
    const int m0 = blockIdx.y * 16;
    const int m = m0 + l15;
    const int mc = m < M ? m : M - 1;
    const int kw = K / NW;                           // k-range of this wave: a multiple of 32 U (the launcher picks U)
    const int k0 = wave * kw + 8 * q4;
    const op_t* ap = g.A + (size_t)mc * K + k0;
    const op_t* wp[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) wp[nb] = g.W + (size_t)(n0 + 16 * nb + l15) * K + k0;

    f32x4 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int batches = kw / (32 * U);
    for (int bt = 0; bt < batches; ++bt) {
        opx8 fa[U], fw[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            fa[u] = *reinterpret_cast<const opx8*>(ap + u * 32);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) fw[u][nb] = *reinterpret_cast<const opx8*>(wp[nb] + u * 32);
        }
        ap += 32 * U;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) wp[nb] += 32 * U;
        __builtin_amdgcn_sched_barrier(0);          // every load of the batch is issued before its first MFMA (the scheduler would otherwise trade
                                                    // memory-level parallelism for registers: 64 VGPRs, a dozen loads in flight)
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[nb] = mfma_16x16x32(fw[u][nb], fa[u], acc[nb]);
    }
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) part[wave][nb][lane] = acc[nb];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll 4          // (fully unrolled, sixteen waves' partials would all be live at once: 240 registers)
    for (int w = 1; w < NW; ++w)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[nb] += part[w][nb][lane];

    skinny_epilogue<EPI>(g, acc, m0, n0, lane);
}

}  // namespace

bool SAT_OPNS::sat_gemm_skinny_supports(int epi, const GemmArgs& a) {
    return (epi == EPI_SWIGLU || epi == EPI_F32 || epi == EPI_RESID) && !a.fp8 && !a.H8 && !a.gate && a.N % 64 == 0 && a.K % 256 == 0 && a.M > 0 && a.M <= 64;
}

namespace {
template <int EPI, int NW>
int launch_skinny(const GemmArgs& a, hipStream_t stream) {
    const dim3 grid(a.N / 64, cdiv(a.M, 16));
    const int steps = a.K / NW / 32;          // k-steps per wave
    if constexpr (NW == 8) {          // (sixteen waves have 128 registers each: four k-steps = 20 loads in flight)
        if (steps % 6 == 0) {
            hipLaunchKernelGGL((gemm_skinny_kernel<EPI, NW, 6>), grid, dim3(NW * 64), 0, stream, a);
            SAT_LAUNCH_CHECK();
            return 0;
        }
    }
    if (steps % 4 == 0) hipLaunchKernelGGL((gemm_skinny_kernel<EPI, NW, 4>), grid, dim3(NW * 64), 0, stream, a);
    else if (steps % 2 == 0) hipLaunchKernelGGL((gemm_skinny_kernel<EPI, NW, 2>), grid, dim3(NW * 64), 0, stream, a);
    else hipLaunchKernelGGL((gemm_skinny_kernel<EPI, NW, 1>), grid, dim3(NW * 64), 0, stream, a);
    SAT_LAUNCH_CHECK();
    return 0;
}
}  // namespace

int SAT_OPNS::sat_launch_gemm_skinny(int epi, const GemmArgs& a, hipStream_t stream) {
    SAT_CHECK_ARG(sat_gemm_skinny_supports(epi, a), SAT_E_UNSUPPORTED,
                  "gemm(skinny): 16-bit operands, SwiGLU or fp32 output without gate, N %% 64 == 0, K %% 256 == 0, at most 64 rows (M=%d N=%d K=%d)", a.M, a.N, a.K);
    SAT_CHECK_ARG(!a.ln_part || (a.ln_c1 && a.ln_c2), SAT_E_INVALID, "gemm(skinny): LayerNorm fold needs ln_c1 / ln_c2");
    SAT_CHECK_ARG((!a.xb && !a.ln_part_out) || (a.xb && a.ln_part_out && epi != EPI_SWIGLU), SAT_E_INVALID, "gemm(skinny): xb and ln_part_out come together, from the fp32 epilogue");
    // the LDS-staged variant wherever K is a whole number of 512-wide chunks it is built for
    {
        const int nch = a.K % 512 == 0 ? a.K / 512 : 0;
        const dim3 grid(a.N / 64, cdiv(a.M, 16));
        auto go = [&](auto kern) -> int {
            SAT_TRY(sat_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), SK2_LDS));
            hipLaunchKernelGGL(kern, grid, dim3(512), SK2_LDS, stream, a);
            SAT_LAUNCH_CHECK();
            return 0;
        };
        if (!(a.variant & 0x100)) {          // (variant bit 8: force the direct variant -- tests, A/B)
            if (epi == EPI_SWIGLU) {
                SAT_CHECK_ARG(a.H, SAT_E_INVALID, "gemm(skinny): null output");
                if (nch == 3) return go(gemm_skinny_lds_kernel<EPI_SWIGLU, 3>);
                if (nch == 2) return go(gemm_skinny_lds_kernel<EPI_SWIGLU, 2>);
                if (nch == 1) return go(gemm_skinny_lds_kernel<EPI_SWIGLU, 1>);
            } else {
                SAT_CHECK_ARG(a.C, SAT_E_INVALID, "gemm(skinny): null output");
                if (nch == 3) return go(gemm_skinny_lds_kernel<EPI_F32, 3>);
                if (nch == 2) return go(gemm_skinny_lds_kernel<EPI_F32, 2>);
                if (nch == 1) return go(gemm_skinny_lds_kernel<EPI_F32, 1>);
                if (nch == 12) return go(gemm_skinny_lds_kernel<EPI_F32, 12>);
            }
        }
    }
    // few workgroups and a long reduction (FF-out: 24 column groups, K = 6144): sixteen waves share it
    const bool wide = a.K % 512 == 0 && a.K >= 4096 && (long)(a.N / 64) * cdiv(a.M, 16) < 128;
    if (epi == EPI_SWIGLU) {
        SAT_CHECK_ARG(a.H, SAT_E_INVALID, "gemm(skinny): null output");
        return wide ? launch_skinny<EPI_SWIGLU, 16>(a, stream) : launch_skinny<EPI_SWIGLU, 8>(a, stream);
    }
    SAT_CHECK_ARG(a.C, SAT_E_INVALID, "gemm(skinny): null output");
    return wide ? launch_skinny<EPI_F32, 16>(a, stream) : launch_skinny<EPI_F32, 8>(a, stream);
}
