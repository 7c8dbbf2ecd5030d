// Flash-style attention core for the DiT: softmax(q k^T / sqrt(64)) v,
// non-causal, no mask, GQA by head index (models/transformer.py:496-536; the reference's
// CPU branch is einsum + softmax(fp32) + einsum, :525-536; GQA repeat_interleave :512-515).
//
// gfx950 design.  One workgroup = 4 waves = 128 queries of one (batch, head); each wave owns
// 32 queries.  K and V^T tiles of 64 keys are staged through LDS (register-staged double
// buffer, one barrier per tile, XOR-swizzled rows -> conflict-free ds_read_b128) and shared
// by the 4 waves.  Both products are computed TRANSPOSED so that everything that belongs
// to one query lives in one lane (plus its lane+32 partner):
//     S^T[key, q] = K[key, :] . Q[q, :]          (A = K fragment from LDS, B = Q in registers)
//     O^T[d,  q] = V^T[d, key] . P^T[key, q]     (A = V^T fragment from LDS, B = P in registers)
// With v_mfma_f32_32x32x16_bf16 the C/D layout is col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5):
// a lane's 16 S^T registers are 16 keys of ITS query, and they are exactly the k-operand
// elements (k = 8*(lane>>5)+j) the second MFMA wants for the key order
//     key(kb,u,half,j) = 32*kb + 16*u + 8*(j>>2) + 4*half + (j&3)
// so P never moves between lanes; V^T is laid out in LDS in that key order (bits 2 and 3
// of the key index swapped) when it is staged.  Row max needs ONE wavefront shuffle
// (lane <-> lane+32) per tile, row sums one at the end; the O rescale is lane-local.
// V^T ([B,KVH,64,Spad]) is produced directly by the QKV GEMM epilogue (gemm_bf16.hip).
#include "sat_common.h"

namespace {

constexpr int KV_TILE = 64;
constexpr int Q_BLOCK = 128;

// MX8: fp8_gemm mode -- the output is the A operand of the to_out GEMM and is written as MXFP8 (e4m3 bytes at `out`, one E8M0
// scale per 32 channels = half a head at `out_scales` [B*Sq][H*2]) instead of bf16
template <bool MX8>
__global__ __launch_bounds__(256, 2) void attention_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                        const bf16_t* __restrict__ vt, bf16_t* __restrict__ out,
                                                        unsigned char* __restrict__ out_scales,
                                                        int H, int KVH, int Sq, int Sk, int Sq_pad, int Sk_pad,
                                                        float scale_log2) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * KV_TILE * 128];   // [stage][K | Vt][64 rows * 128 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y;
    const int kvh = h / (H / KVH);
    const int qi = blockIdx.x * Q_BLOCK + wave * 32 + l31;   // < Sq_pad

    // Q fragments (B operand of S^T): Q[qi][16t + 8*half .. +8]
    bf16x8 qf[4];
    {
        const bf16_t* qp = q + ((size_t)(b * H + h) * Sq_pad + qi) * 64 + half * 8;
#pragma unroll
        for (int t = 0; t < 4; ++t) qf[t] = *reinterpret_cast<const bf16x8*>(qp + t * 16);
    }

    const bf16_t* kbase = k + (size_t)(b * KVH + kvh) * Sk_pad * 64;
    const bf16_t* vbase = vt + (size_t)(b * KVH + kvh) * 64 * Sk_pad;

    // staging coordinates: 512 16-B chunks per operand per tile, 2 per thread
    int srow[2], schk[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int id = i * 256 + tid;
        srow[i] = id >> 3;
        schk[i] = id & 7;
    }
    u32x4 rk[2], rv[2];
    auto gload = [&](int tile) {
        const int key0 = tile * KV_TILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            rk[i] = *reinterpret_cast<const u32x4*>(kbase + (size_t)(key0 + srow[i]) * 64 + schk[i] * 8);
            rv[i] = *reinterpret_cast<const u32x4*>(vbase + (size_t)srow[i] * Sk_pad + key0 + schk[i] * 8);
        }
    };
    auto lstore = [&](int stage) {
        char* sk = smem + stage * (2 * KV_TILE * 128);
        char* sv = sk + KV_TILE * 128;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<u32x4*>(sk + lds_tile_off(srow[i], schk[i])) = rk[i];
            // keys 8c..8c+3 -> chunk 2*(c>>1), keys 8c+4..8c+7 -> chunk 2*(c>>1)+1, byte 8*(c&1)
            const int c = schk[i];
            const int c0 = (c >> 1) * 2;
            *reinterpret_cast<u32x2*>(sv + lds_tile_off(srow[i], c0) + 8 * (c & 1)) = u32x2{rv[i][0], rv[i][1]};
            *reinterpret_cast<u32x2*>(sv + lds_tile_off(srow[i], c0 + 1) + 8 * (c & 1)) = u32x2{rv[i][2], rv[i][3]};
        }
    };

    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -1e30f;   // running max, in log2-scaled units
    float l_run = 0.f;      // this lane's partial row sum (its 32 of every 64 keys)

    // K rows / V^T columns of sequence b start at ob = (b*Sk) & 3 (see EPI_HEADS in gemm_bf16.hip)
    const int ob = (b * Sk) & 3;
    const int k_end = ob + Sk;
    const int n_tiles = (k_end + KV_TILE - 1) / KV_TILE;
    // a wave whose 32 queries are all beyond Sq (tail workgroup) still stages tiles and joins the barriers,
    // but skips the matrix and softmax work
    const bool wave_active = __builtin_amdgcn_readfirstlane(blockIdx.x * Q_BLOCK + wave * 32) < Sq;
    gload(0);
    lstore(0);
    __syncthreads();

    auto process = [&](int tile) {
        if (!wave_active) return;
        const int cur = tile & 1;
        const char* sk = smem + cur * (2 * KV_TILE * 128);
        const char* sv = sk + KV_TILE * 128;

        // ---- S^T = K Q^T : two 32-key blocks.  All 8 K fragments are requested up front so that the MFMAs never
        // wait on a just-issued ds_read.
        bf16x8 kf[2][4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                kf[kb][t] = *reinterpret_cast<const bf16x8*>(sk + lds_tile_off(kb * 32 + l31, t * 2 + half));
        __builtin_amdgcn_sched_barrier(0);
        f32x16 sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][t], qf[t], sacc[kb], 0, 0, 0);
        }
        // V^T fragments do not depend on the softmax: request them now, their LDS latency hides behind the VALU work
        bf16x8 vf[2][4];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int ku = 0; ku < 4; ++ku)
                vf[db][ku] = *reinterpret_cast<const bf16x8*>(sv + lds_tile_off(db * 32 + l31, ku * 2 + half));
        __builtin_amdgcn_sched_barrier(0);
        // ---- mask the columns outside [ob, ob + Sk) (wave-uniform branch: first and last tile only)
        if ((tile == 0 && ob != 0) || (tile == n_tiles - 1 && (k_end & (KV_TILE - 1)) != 0)) {
            const int key0 = tile * KV_TILE + 4 * half;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2);
                    if (key < ob || key >= k_end) sacc[kb][r] = -INFINITY;
                }
        }
        // ---- online softmax (per query = per lane pair)
        float mloc = sacc[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, sacc[kb][r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc * scale_log2);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
        bf16x8 pb[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r], scale_log2, -m_new));
                psum += p;
                pb[kb][r >> 3][r & 7] = f32_to_bf16(p);
            }
        l_run = l_run * alpha + psum;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
        }
        // ---- O^T += V^T P^T
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[db][kb * 2 + u], pb[kb][u], oacc[db], 0, 0, 0);
    };
    for (int tile = 0; tile < n_tiles - 1; ++tile) {
        gload(tile + 1);
        process(tile);
        lstore((tile + 1) & 1);
        __syncthreads();
    }
    process(n_tiles - 1);

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if constexpr (MX8) {
        // block db (32 channels of this head): this lane holds 16 of them, lane ^ 32 the other 16
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            float am = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) am = fmaxf(am, fabsf(oacc[db][r] * inv));
            am = fmaxf(am, __shfl_xor(am, 32, 64));
            const float t = am * (1.0f / 448.0f);
            const unsigned tb = __float_as_uint(t);
            int e = (int)((tb >> 23) & 0xff) - 127 + ((tb & 0x7fffff) ? 1 : 0);
            e = am > 0.f ? (e < -127 ? -127 : (e > 127 ? 127 : e)) : -127;
            const float qs = inv * __uint_as_float((unsigned)(127 - e) << 23);
            if (qi < Sq) {
                unsigned char* op = reinterpret_cast<unsigned char*>(out) + ((size_t)b * Sq + qi) * ((size_t)H * 64) + h * 64 + db * 32;
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    unsigned p = 0;
                    p = __builtin_amdgcn_cvt_pk_fp8_f32(oacc[db][rq * 4] * qs, oacc[db][rq * 4 + 1] * qs, p, false);
                    p = __builtin_amdgcn_cvt_pk_fp8_f32(oacc[db][rq * 4 + 2] * qs, oacc[db][rq * 4 + 3] * qs, p, true);
                    *reinterpret_cast<unsigned*>(op + rq * 8 + half * 4) = p;
                }
                if (half == 0) out_scales[((size_t)b * Sq + qi) * ((size_t)H * 2) + h * 2 + db] = (unsigned char)(e + 127);
            }
        }
    } else if (qi < Sq) {
        bf16_t* op = out + ((size_t)b * Sq + qi) * ((size_t)H * 64) + h * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f32_to_bf16(oacc[db][rq * 4 + e] * inv);
                *reinterpret_cast<bf16x4*>(op + db * 32 + rq * 8 + half * 4) = o;
            }
    }
}

}  // namespace

int sat_launch_attention(const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* out, int b, int h, int kvh,
                         int sq, int sk, int sq_pad, int sk_pad, hipStream_t s, unsigned char* out_scales) {
    SAT_CHECK_ARG(q && k && vt && out, SAT_E_INVALID, "attention: null pointer");
    SAT_CHECK_ARG(b > 0 && h > 0 && kvh > 0 && h % kvh == 0, SAT_E_INVALID, "attention: bad heads %d/%d", h, kvh);
    SAT_CHECK_ARG(sq > 0 && sk > 0 && sq_pad >= sq && sk_pad >= sk, SAT_E_INVALID, "attention: bad lengths");
    SAT_CHECK_ARG(sq_pad % Q_BLOCK == 0 && sk_pad % KV_TILE == 0, SAT_E_INVALID,
                  "attention: sq_pad %% 128 and sk_pad %% 64 must be 0 (got %d, %d)", sq_pad, sk_pad);
    SAT_CHECK_ARG(sk_pad >= sk + 3, SAT_E_INVALID, "attention: sk_pad must be >= sk + 3 (key-side shift), got %d for sk=%d", sk_pad, sk);
    const float scale_log2 = 0.125f * 1.4426950408889634f;   // 1/sqrt(64) * log2(e)
    dim3 grid(cdiv(sq, Q_BLOCK), h, b);
    if (out_scales)
        hipLaunchKernelGGL(attention_kernel<true>, grid, dim3(256), 0, s, q, k, vt, out, out_scales, h, kvh, sq, sk, sq_pad, sk_pad, scale_log2);
    else
        hipLaunchKernelGGL(attention_kernel<false>, grid, dim3(256), 0, s, q, k, vt, out, out_scales, h, kvh, sq, sk, sq_pad, sk_pad, scale_log2);
    SAT_LAUNCH_CHECK();
    return 0;
}
