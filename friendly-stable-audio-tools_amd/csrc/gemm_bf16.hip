// bf16 MFMA GEMM for the DiT projections and the SwiGLU FFN (SURVEY.md K2,K5,K6,K7,K8):
//     C[M,N] = A[M,K] . W[N,K]^T      A, W bf16 (K contiguous), fp32 accumulate
// replacing the nn.Linear calls at models/transformer.py:222,270,314,311-312,319 of the
// reference.  gfx950 only: v_mfma_f32_32x32x16_bf16, 64-wide wavefronts, LDS-staged tiles
// (XOR-swizzled, conflict-free ds_read_b128), register-staged double buffering with one
// barrier per K-tile, XCD-aware tile rasterisation.
//
// Fused epilogues
//   EPI_F32    C (fp32) = acc (+bias) (+C)           -> to_out / FF-out + residual add
//   EPI_SWIGLU H (bf16) = (acc_v+b_v) * silu(acc_g+b_g)   (W rows interleaved 32 value /
//              32 gate so both land in one wave tile; models/transformer.py:232-235)
//   EPI_HEADS  split into heads, partial RoPE (models/transformer.py:158-183,438-452) on
//              q/k, store q/k as [B,H,Spad,64] and v transposed as [B,H,64,Spad]
#include <stdlib.h>

#include <type_traits>

#include "attn_core.h"
#include <algorithm>

#include "sat_common.h"

namespace {

template <int EPI, int MI, int NI, bool LNC = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x16 (&acc)[MI][NI], const int mw, const int nw, const int half,
                                              const int l31, const float2* ln = nullptr, const float* lc1 = nullptr,
                                              const float* lc2 = nullptr) {
    // LNC (bf16 pipelined kernels, EPI_HEADS / EPI_SWIGLU): out = rstd * (acc - mean * c1) + c2 with (mean, rstd) of the wave's TM
    // token rows at `ln` and the wave's 64 channel constants at lc1 / lc2, all in LDS -- the LayerNorm fold of GemmArgs, or
    // (0, 1), 0, bias without it
    const int M = g.M, N = g.N;
    (void)N;
    // acc[i][j][r]: row = i*32 + (r&3) + 8*(r>>2) + 4*half ; col = j*32 + l31   (guide section 3)

    if constexpr (EPI == EPI_F32) {
        float* __restrict__ C = g.C;
        const int ldc = g.ldc;
        float bia[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) bia[j] = g.bias ? g.bias[nw + j * 32 + l31] : 0.f;
        const bool accum = g.accumulate != 0;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                // residual reads are issued as one batch per 16-row half block (16 loads in flight), then added
                float old[8][NI];
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) {
                    const int r = rh * 8 + r8;
                    int m = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        old[r8][j] = (accum && m < M) ? C[(size_t)m * ldc + nw + j * 32 + l31] : 0.f;
                }
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) {
                    const int r = rh * 8 + r8;
                    int m = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (m < M) {
                        if (g.gate) {      // adaLN gated branch (transformer.py:674, 688): wave-uniform pointer test
                            const float* gr = g.gate + (size_t)(m / g.gate_rows) * g.gate_ld + nw + l31;
#pragma unroll
                            for (int j = 0; j < NI; ++j)
                                C[(size_t)m * ldc + nw + j * 32 + l31] = (acc[i][j][r] + bia[j]) * gr[j * 32] + old[r8][j];
                        } else {
#pragma unroll
                            for (int j = 0; j < NI; ++j) C[(size_t)m * ldc + nw + j * 32 + l31] = acc[i][j][r] + bia[j] + old[r8][j];
                        }
                    }
                }
            }
    } else if constexpr (EPI == EPI_SWIGLU) {
        op_t* __restrict__ H = g.H;
        const int ldh = N >> 1;
        const float bv = g.bias ? g.bias[nw + l31] : 0.f;
        const float bg = g.bias ? g.bias[nw + 32 + l31] : 0.f;
        const int hc = (nw >> 1) + l31;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float hv = 0.f;
                if (m < M) {
                    float v = acc[i][0][r] + bv;
                    float gt = acc[i][1][r] + bg;
                    hv = v * silu_f(gt);
                }
                if (g.H8) {
                    // MXFP8: the 32 lanes of this half hold the 32 consecutive hidden columns of one block of row m.
                    // scale = 2^e with e = ceil(log2(amax / 448)) (so that amax / scale <= 448), stored as E8M0 = e + 127
                    const float am = half32_max(fabsf(hv));
                    const float t = am * (1.0f / 448.0f);
                    const unsigned tb = __float_as_uint(t);
                    int e = (int)((tb >> 23) & 0xff) - 127 + ((tb & 0x7fffff) ? 1 : 0);
                    e = am > 0.f ? (e < -127 ? -127 : (e > 127 ? 127 : e)) : -127;
                    const float inv = __uint_as_float((unsigned)(127 - e) << 23);          // 2^-e (e = -127 -> 2^254: only for an all-zero block)
                    const unsigned q = __builtin_amdgcn_cvt_pk_fp8_f32(hv * inv, 0.f, 0u, false);
                    if (m < M) {
                        g.H8[(size_t)m * ldh + hc] = (unsigned char)(q & 0xff);
                        if (l31 == 0) g.Hs[(size_t)m * (ldh >> 5) + (hc >> 5)] = (unsigned char)(e + 127);
                    }
                } else if (m < M) {
                    H[(size_t)m * ldh + hc] = f32_to_op(hv);
                }
            }
    } else {   // EPI_HEADS
        const HeadsEpi& he = g.heads;
        const int hp = he.heads * 64;
        const int part = nw / hp;
        const int head = (nw - part * hp) >> 6;
        const int kind = he.kind[part];
        op_t* __restrict__ dst = he.out[part];
        const int S = he.S, Spad = he.Spad;
        // K rows / V^T columns of sequence b are shifted by o_b = (b*S) & 3 (kind bit 2) so that the 4 consecutive
        // tokens a lane holds (rows 4q..4q+3 of the GEMM) land on an 8-byte aligned V^T span: one dwordx2 store
        // instead of four 2-byte scatters.  The attention kernel applies the same shift to its key index.
        const bool shift = (kind & 4) != 0;
        float c1a = 0.f, c1b = 0.f, c2a = 0.f, c2b = 0.f;
        if constexpr (LNC) {
            c1a = lc1[l31];
            c1b = lc1[32 + l31];
            c2a = lc2[l31];
            c2b = lc2[32 + l31];
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int mb = mw + i * 32 + 8 * rq + 4 * half;          // first of 4 consecutive rows, multiple of 4
                float v0[4], v1[4];
                int bb[4], ss[4];
                f32x4 st01 = {0.f, 1.f, 0.f, 1.f}, st23 = {0.f, 1.f, 0.f, 1.f};      // (mean, rstd) of rows e = 0,1 / 2,3
                if constexpr (LNC) {
                    const f32x4* sp = reinterpret_cast<const f32x4*>(ln + i * 32 + 8 * rq + 4 * half);
                    st01 = sp[0];
                    st23 = sp[1];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = rq * 4 + e;
                    int mm = mb + e;
                    mm = mm < M ? mm : M - 1;
                    bb[e] = mm / S;
                    ss[e] = mm - bb[e] * S;
                    v0[e] = acc[i][0][r];
                    v1[e] = acc[i][1][r];
                    if constexpr (LNC) {
                        const float mean = e < 2 ? st01[2 * e] : st23[2 * e - 4];
                        const float rstd = e < 2 ? st01[2 * e + 1] : st23[2 * e - 3];
                        v0[e] = rstd * (v0[e] - mean * c1a) + c2a;
                        v1[e] = rstd * (v1[e] - mean * c1b) + c2b;
                    }
                    if (kind & 2) {   // wave-uniform: partial RoPE on d < 32 (pairs d, d^16)
                        float p = __shfl_xor(v0[e], 16, 64);
                        int jf = l31 & 15;
                        float cs = he.rope_cos[ss[e] * 16 + jf], sn = he.rope_sin[ss[e] * 16 + jf];
                        v0[e] = (l31 < 16) ? (v0[e] * cs - p * sn) : (v0[e] * cs + p * sn);
                    }
                }
                if (kind & 1) {
                    if (mb + 3 < M && bb[0] == bb[3]) {
                        const int ob = shift ? ((bb[0] * S) & 3) : 0;
                        const size_t hbase = ((size_t)(bb[0] * he.heads + head) * 64) * Spad;
                        const size_t base = hbase + vt_pos(ss[0] + ob);         // aligned group of 4 keys: stays a group of 4
                        opx4 p0, p1;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            p0[e] = f32_to_op(v0[e]);
                            p1[e] = f32_to_op(v1[e]);
                        }
                        if (shift) {     // aligned: (ss[0] + ob) % 4 == mb % 4 == 0
                            *reinterpret_cast<opx4*>(dst + base + (size_t)l31 * Spad) = p0;
                            *reinterpret_cast<opx4*>(dst + base + (size_t)(32 + l31) * Spad) = p1;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const size_t pe = hbase + vt_pos(ss[0] + e);
                                dst[pe + (size_t)l31 * Spad] = p0[e];
                                dst[pe + (size_t)(32 + l31) * Spad] = p1[e];
                            }
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (mb + e < M) {
                                const int ob = shift ? ((bb[e] * S) & 3) : 0;
                                const size_t base = ((size_t)(bb[e] * he.heads + head) * 64) * Spad + vt_pos(ss[e] + ob);
                                dst[base + (size_t)l31 * Spad] = f32_to_op(v0[e]);
                                dst[base + (size_t)(32 + l31) * Spad] = f32_to_op(v1[e]);
                            }
                    }
                } else {
                    const float qs = (kind & 8) ? he.qscale : 1.0f;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (mb + e < M) {
                            const int ob = shift ? ((bb[e] * S) & 3) : 0;
                            const size_t base = ((size_t)(bb[e] * he.heads + head) * Spad + ss[e] + ob) * 64;
                            dst[base + l31] = f32_to_op(v0[e] * qs);
                            dst[base + 32 + l31] = f32_to_op(v1[e] * qs);
                        }
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// Epilogues on TRANSPOSED accumulators.  The deep-prefetch kernels issue their MFMAs with the operands swapped
// (W fragment as the A operand, activation fragment as the B operand), so a 32x32 block holds C^T: lane l31 owns TOKEN row
// m = mw + i*32 + l31, and its 16 registers are output channels n = nw + j*32 + 8*(r>>2) + 4*half + (r&3) -- four runs of
// four consecutive channels.  Everything a token needs is then lane-local:
//   * fp32 output / residual: 16-byte loads and stores (4 per 32x32 block instead of 16 dword accesses);
//   * bf16 / e4m3 output: two runs are exchanged between the wave halves with v_permlane32_swap (lane l takes the partner's
//     low run, lane l+32 the high runs), so every lane writes 8 consecutive channels = one 16-byte (8-byte for e4m3) store
//     instead of sixteen 2-byte stores -- the epilogue is store-issue bound (guide T21), not bandwidth bound;
//   * RoPE: the rotation partner d^16 is register r^8 of the same lane, no cross-lane traffic, cos/sin are two 16-byte loads;
//   * SwiGLU / MXFP8 block maximum: value and gate, resp. the 32 channels of a block, sit in one lane pair.
// V^T (token-contiguous destination) keeps the un-swapped orientation and the epilogue above.
// ---------------------------------------------------------------------------------------------
// ROPE_PRE: the rotation table rows of all MI row blocks are fetched up front (not in the 128-register budget of the 16-wave tile)
template <int EPI, int MI, int NI, bool LNC = false, bool ROPE_PRE = true>
__device__ __forceinline__ void gemm_epilogue_t(const GemmArgs& g, f32x16 (&acc)[MI][NI], const int mw, const int nw, const int half,
                                                const int l31, const float2* ln = nullptr, const float* lc1 = nullptr,
                                                const float* lc2 = nullptr, unsigned (*qfrag)[8] = nullptr) {
    const int M = g.M, N = g.N;
    (void)N;
    if constexpr (EPI == EPI_F32) {
        const bool accum = g.accumulate != 0;
        const int c0 = nw + 4 * half;                  // first channel of this lane's run 0
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = mw + i * 32 + l31;
            const int mc = m < M ? m : M - 1;
            float* __restrict__ crow = g.C + (size_t)mc * g.ldc + c0;
            f32x4 old[NI][4];
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    old[j][q] = accum ? *reinterpret_cast<const f32x4*>(crow + j * 32 + q * 8) : f32x4{0.f, 0.f, 0.f, 0.f};
            const float* grow = g.gate ? g.gate + (size_t)(mc / g.gate_rows) * g.gate_ld + c0 : nullptr;
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                    if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + c0 + j * 32 + q * 8);
                    if (grow) v *= *reinterpret_cast<const f32x4*>(grow + j * 32 + q * 8);      // adaLN gate (transformer.py:674, 688)
                    v += old[j][q];
                    if (m < M) *reinterpret_cast<f32x4*>(crow + j * 32 + q * 8) = v;
                }
        }
    } else if constexpr (EPI == EPI_SWIGLU) {
        static_assert(NI == 2, "value block + gate block");
        const int ldh = N >> 1;
        const int hc0 = nw >> 1;                       // first hidden column of this wave's 32
        [[maybe_unused]] f32x4 bv[LNC ? 1 : 4], bg[LNC ? 1 : 4];
        if constexpr (!LNC) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bv[q] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + nw + 4 * half + q * 8) : f32x4{0.f, 0.f, 0.f, 0.f};
                bg[q] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + nw + 32 + 4 * half + q * 8) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = mw + i * 32 + l31;
            float hv[16];
            if constexpr (LNC) {      // LayerNorm fold: finish the normalisation of token row m (bias is part of ln_c2)
                const float2 st = ln[i * 32 + l31];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = 4 * half + q * 8;
                    const f32x4 c1v = *reinterpret_cast<const f32x4*>(lc1 + c), c1g = *reinterpret_cast<const f32x4*>(lc1 + c + 32);
                    const f32x4 c2v = *reinterpret_cast<const f32x4*>(lc2 + c), c2g = *reinterpret_cast<const f32x4*>(lc2 + c + 32);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = st.y * (acc[i][0][4 * q + e] - st.x * c1v[e]) + c2v[e];
                        const float gt = st.y * (acc[i][1][4 * q + e] - st.x * c1g[e]) + c2g[e];
                        hv[4 * q + e] = m < M ? v * silu_f(gt) : 0.f;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[i][0][r] + bv[r >> 2][r & 3];
                    const float gt = acc[i][1][r] + bg[r >> 2][r & 3];
                    hv[r] = m < M ? v * silu_f(gt) : 0.f;
                }
            }
            if (g.H8) {
                // MXFP8: the 32 hidden columns of this wave are ONE block of row m, held by the lane pair (l31, l31 + 32)
                float am = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) am = fmaxf(am, fabsf(hv[r]));
                {
                    unsigned a = __float_as_uint(am), b = a;
                    half_swap(a, b);                   // a: this value of the low half in both halves, b: of the high half
                    am = fmaxf(__uint_as_float(a), __uint_as_float(b));
                }
                const float t = am * (1.0f / 448.0f);
                const unsigned tb = __float_as_uint(t);
                int e = (int)((tb >> 23) & 0xff) - 127 + ((tb & 0x7fffff) ? 1 : 0);
                e = am > 0.f ? (e < -127 ? -127 : (e > 127 ? 127 : e)) : -127;
                const float inv = __uint_as_float((unsigned)(127 - e) << 23);          // 2^-e
                unsigned q8[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned w = __builtin_amdgcn_cvt_pk_fp8_f32(hv[4 * q] * inv, hv[4 * q + 1] * inv, 0u, false);
                    q8[q] = __builtin_amdgcn_cvt_pk_fp8_f32(hv[4 * q + 2] * inv, hv[4 * q + 3] * inv, w, true);
                }
                half_swap(q8[0], q8[1]);
                half_swap(q8[2], q8[3]);
                if (m < M) {
                    unsigned char* hrow = g.H8 + (size_t)m * ldh + hc0 + 8 * half;
                    *reinterpret_cast<u32x2*>(hrow) = u32x2{q8[0], q8[1]};
                    *reinterpret_cast<u32x2*>(hrow + 16) = u32x2{q8[2], q8[3]};
                    if (half == 0) g.Hs[(size_t)m * (ldh >> 5) + (hc0 >> 5)] = (unsigned char)(e + 127);
                }
            } else {
                unsigned pk[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    pk[2 * q] = pack_op2(hv[4 * q], hv[4 * q + 1]);
                    pk[2 * q + 1] = pack_op2(hv[4 * q + 2], hv[4 * q + 3]);
                }
                half_swap(pk[0], pk[2]);
                half_swap(pk[1], pk[3]);
                half_swap(pk[4], pk[6]);
                half_swap(pk[5], pk[7]);
                if (m < M) {
                    op_t* hrow = g.H + (size_t)m * ldh + hc0 + 8 * half;
                    *reinterpret_cast<u32x4*>(hrow) = u32x4{pk[0], pk[1], pk[2], pk[3]};
                    *reinterpret_cast<u32x4*>(hrow + 16) = u32x4{pk[4], pk[5], pk[6], pk[7]};
                }
            }
        }
    } else {   // EPI_HEADS, row-major destinations only ([B,H,Spad,64]: q, k); V^T takes the un-swapped orientation
        static_assert(NI == 2, "one head = two 32-column blocks");
        const HeadsEpi& he = g.heads;
        const int hp = he.heads * 64;
        const int part = nw / hp;
        const int head = (nw - part * hp) >> 6;
        const int kind = he.kind[part];
        op_t* __restrict__ dst = he.out[part];
        const int S = he.S, Spad = he.Spad;
        // the rotation table rows of ALL of the lane's row blocks up front: one round trip instead of one per row block behind the previous
        // block's stores (round 5)
        [[maybe_unused]] f32x4 rcs[MI][2], rsn[MI][2];
        if (ROPE_PRE && (kind & 2)) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = mw + i * 32 + l31;
                const int s = (m < M ? m : M - 1) % S;
                rcs[i][0] = *reinterpret_cast<const f32x4*>(he.rope_cos + (size_t)s * 16 + 4 * half);
                rcs[i][1] = *reinterpret_cast<const f32x4*>(he.rope_cos + (size_t)s * 16 + 8 + 4 * half);
                rsn[i][0] = *reinterpret_cast<const f32x4*>(he.rope_sin + (size_t)s * 16 + 4 * half);
                rsn[i][1] = *reinterpret_cast<const f32x4*>(he.rope_sin + (size_t)s * 16 + 8 + 4 * half);
            }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = mw + i * 32 + l31;
            const int mc = m < M ? m : M - 1;
            const int b = mc / S;
            const int s = mc - b * S;
            const int ob = (kind & 4) ? ((b * S) & 3) : 0;
            if constexpr (LNC) {         // LayerNorm fold (before the rotation: both are linear, this one is per channel)
                const float2 st = ln[i * 32 + l31];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 c1 = *reinterpret_cast<const f32x4*>(lc1 + j * 32 + q * 8 + 4 * half);
                        const f32x4 c2 = *reinterpret_cast<const f32x4*>(lc2 + j * 32 + q * 8 + 4 * half);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] = st.y * (acc[i][j][4 * q + e] - st.x * c1[e]) + c2[e];
                    }
            }
            if (kind & 2) {   // partial RoPE on d < 32 (block j = 0): partner of d < 16 is d + 16 = register r + 8 of this lane
                if constexpr (!ROPE_PRE) {
                    rcs[i][0] = *reinterpret_cast<const f32x4*>(he.rope_cos + (size_t)s * 16 + 4 * half);
                    rcs[i][1] = *reinterpret_cast<const f32x4*>(he.rope_cos + (size_t)s * 16 + 8 + 4 * half);
                    rsn[i][0] = *reinterpret_cast<const f32x4*>(he.rope_sin + (size_t)s * 16 + 4 * half);
                    rsn[i][1] = *reinterpret_cast<const f32x4*>(he.rope_sin + (size_t)s * 16 + 8 + 4 * half);
                }
                const f32x4 cs0 = rcs[i][0], cs1 = rcs[i][1], sn0 = rsn[i][0], sn1 = rsn[i][1];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float cs = (r < 4) ? cs0[r & 3] : cs1[r & 3];
                    const float sn = (r < 4) ? sn0[r & 3] : sn1[r & 3];
                    const float x1 = acc[i][0][r], x2 = acc[i][0][r + 8];
                    acc[i][0][r] = x1 * cs - x2 * sn;
                    acc[i][0][r + 8] = x2 * cs + x1 * sn;
                }
            }
            if (kind & 8) {   // query: pre-scaled for the attention kernel (one rounding, here)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] *= he.qscale;
            }
            op_t* row = dst + ((size_t)(b * he.heads + head) * Spad + s + ob) * 64 + 8 * half;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                unsigned pk[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    pk[2 * q] = pack_op2(acc[i][j][4 * q], acc[i][j][4 * q + 1]);
                    pk[2 * q + 1] = pack_op2(acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                }
                half_swap(pk[0], pk[2]);
                half_swap(pk[1], pk[3]);
                half_swap(pk[4], pk[6]);
                half_swap(pk[5], pk[7]);
                if (qfrag) {       // fused cross-attention (MI == 1): channels [32 j + 8 half, +8) and [32 j + 16 + 8 half, +8) of the lane's row
                                   // ARE the B-operand fragments 2 j and 2 j + 1 of the score MFMA
#pragma unroll
                    for (int e = 0; e < 8; ++e) qfrag[j][e] = pk[e];
                    continue;
                }
                if (m < M) {
                    *reinterpret_cast<u32x4*>(row + j * 32) = u32x4{pk[0], pk[1], pk[2], pk[3]};
                    *reinterpret_cast<u32x4*>(row + j * 32 + 16) = u32x4{pk[4], pk[5], pk[6], pk[7]};
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fp32 output / residual epilogue staged through LDS (un-swapped orientation, wave tile TM x 64).  Each wave writes one
// 32 x 64 fp32 block of its accumulators into a private 8 KiB LDS region (row-major, ds_write_b32: the 32 lanes of a half cover
// one 128-byte row segment, conflict-free) and reads it back as 16-byte pieces of whole rows, so that the read-modify-write of
// the residual stream becomes 8 + 8 fully coalesced 16-byte accesses per block (4 rows x 256 contiguous bytes per instruction)
// instead of 32 + 32 dword accesses.  The LDS ring is free at that point (the caller passes a barrier after the last K-tile).
// ---------------------------------------------------------------------------------------------
// PRE: the residual values were loaded at kernel start (`pre`, MI == 1 only): their HBM / L2 latency is then hidden behind
// the whole K loop instead of sitting between the last MFMA and the first store.
template <int MI, bool PRE = false>
__device__ __forceinline__ void gemm_epilogue_f32_staged(const GemmArgs& g, f32x16 (&acc)[MI][2], float* ep, const int mw, const int nw,
                                                         const int lane, const f32x4* pre = nullptr) {
    const int half = lane >> 5, l31 = lane & 31;
    const int M = g.M;
    const int prow = lane >> 4;            // row of this lane inside a 4-row pass
    const int c4 = (lane & 15) * 4;        // first of its 4 columns
    const bool accum = g.accumulate != 0;
    const f32x4 bia = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + nw + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) ep[((r & 3) + 8 * (r >> 2) + 4 * half) * 64 + j * 32 + l31] = acc[i][j][r];
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {     // two batches of four passes: 16 registers of residual values in flight
            f32x4 old[4];
#pragma unroll
            for (int p4 = 0; p4 < 4; ++p4) {
                const int m = mw + i * 32 + (grp * 4 + p4) * 4 + prow;
                if constexpr (PRE) old[p4] = pre[grp * 4 + p4];
                else old[p4] = (accum && m < M) ? *reinterpret_cast<const f32x4*>(g.C + (size_t)m * g.ldc + nw + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int p4 = 0; p4 < 4; ++p4) {
                const int row = (grp * 4 + p4) * 4 + prow;
                const int m = mw + i * 32 + row;
                f32x4 v = *reinterpret_cast<const f32x4*>(ep + row * 64 + c4) + bia;
                if (m < M && g.gate) v *= *reinterpret_cast<const f32x4*>(g.gate + (size_t)(m / g.gate_rows) * g.gate_ld + nw + c4);
                v += old[p4];
                if (m < M) *reinterpret_cast<f32x4*>(g.C + (size_t)m * g.ldc + nw + c4) = v;
                if (g.xb) {
                    // LayerNorm fold, producer side (wave-uniform test): bf16 image of the updated row piece + the statistics of the
                    // ROUNDED values over this wave's 64-column block (the 16 lanes of a DPP row hold one row of the pass)
                    opx4 xr;
                    float sum = 0.f, sq = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        xr[e] = f32_to_op(v[e]);
                        const float f = op_to_f32(xr[e]);
                        sum += f;
                        sq += f * f;
                    }
                    sum = row16_sum(sum);
                    sq = row16_sum(sq);
                    if (m < M) {
                        *reinterpret_cast<opx4*>(g.xb + (size_t)m * g.N + nw + c4) = xr;
                        if ((lane & 15) == 0)
                            *reinterpret_cast<float2*>(g.ln_part_out + ((size_t)m * (g.N >> 6) + (nw >> 6)) * 2) = make_float2(sum, sq);
                    }
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel(GemmArgs g) {
    sat_f16_saturate();
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM;
    constexpr int TN = BN / WN;
    static_assert(TN == 64, "wave tile is TM x 64");
    constexpr int MI = TM / 32;
    constexpr int NI = 2;
    constexpr int A_CH = BM * 8 / NT;
    constexpr int B_CH = BN * 8 / NT;
    static_assert(A_CH >= 1 && B_CH >= 1, "tile too small for the block");
    constexpr int STAGE_BYTES = (BM + BN) * 128;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    const int M = g.M, N = g.N, K = g.K;
    const int tiles_m = (M + BM - 1) / BM;
    const int tiles_n = N / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    if (tiles_m <= tiles_n) {   // M is the short dimension: m fastest, W panels stay in one XCD's L2
        tn = bid / tiles_m;
        tm = bid - tn * tiles_m;
    } else {
        tm = bid / tiles_n;
        tn = bid - tm * tiles_n;
    }
    const int m0 = tm * BM;
    const int n0 = tn * BN;

    const op_t* __restrict__ A = g.A;
    const op_t* __restrict__ W = g.W;

    // per-thread staging coordinates
    int a_row[A_CH], a_chk[A_CH];
    const op_t* a_ptr[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        int id = i * NT + tid;
        a_row[i] = id >> 3;
        a_chk[i] = id & 7;
        int gm = m0 + a_row[i];
        gm = gm < M ? gm : M - 1;
        a_ptr[i] = A + (size_t)gm * K + a_chk[i] * 8;
    }
    int b_row[B_CH], b_chk[B_CH];
    const op_t* b_ptr[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        int id = i * NT + tid;
        b_row[i] = id >> 3;
        b_chk[i] = id & 7;
        b_ptr[i] = W + (size_t)(n0 + b_row[i]) * K + b_chk[i] * 8;
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 ra[A_CH], rb[B_CH];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) ra[i] = *reinterpret_cast<const u32x4*>(a_ptr[i] + kt * 64);
#pragma unroll
        for (int i = 0; i < B_CH; ++i) rb[i] = *reinterpret_cast<const u32x4*>(b_ptr[i] + kt * 64);
    };
    auto lstore = [&](int stage) {
        char* sa = smem + stage * STAGE_BYTES;
        char* sb = sa + BM * 128;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) *reinterpret_cast<u32x4*>(sa + lds_tile_off(a_row[i], a_chk[i])) = ra[i];
#pragma unroll
        for (int i = 0; i < B_CH; ++i) *reinterpret_cast<u32x4*>(sb + lds_tile_off(b_row[i], b_chk[i])) = rb[i];
    };

    auto compute = [&](int stage) {
        const char* sa = smem + stage * STAGE_BYTES;
        const char* sb = sa + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            opx8 af[MI], bfr[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                af[i] = *reinterpret_cast<const opx8*>(sa + lds_tile_off(wm * TM + i * 32 + l31, ks * 2 + half));
#pragma unroll
            for (int j = 0; j < NI; ++j)
                bfr[j] = *reinterpret_cast<const opx8*>(sb + lds_tile_off(wn * TN + j * 32 + l31, ks * 2 + half));
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = mfma_32x32x16(af[i], bfr[j], acc[i][j]);
        }
    };

    // register-staged double buffer, one barrier per K-tile; the last tile is peeled so that the
    // staging registers are written/read unconditionally (keeps them out of scratch)
    const int nk = K / 64;
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

    gemm_epilogue<EPI, MI, NI>(g, acc, m0 + wm * TM, n0 + wn * TN, half, l31);
}

// ---------------------------------------------------------------------------------------------
// Direct-to-LDS variant: tiles are staged with global_load_lds_dwordx4 (LDS-DMA, no VGPR round
// trip, no ds_write pass).  The LDS image is the same XOR-swizzled one; because an LDS-DMA writes
// wave-uniform base + lane*16, the swizzle is applied to the per-lane SOURCE address (guide
// section 5.4 rule 21): lane l of the piece that covers rows 8p..8p+7 lands at row 8p + l/8,
// position l%8, so it fetches logical chunk (l%8) ^ ((row>>1)&7) of that row.
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM * WN * 64) void gemm_glds_kernel(GemmArgs g) {
    sat_f16_saturate();
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM;
    constexpr int TN = BN / WN;
    static_assert(TN == 64, "wave tile is TM x 64");
    constexpr int MI = TM / 32;
    constexpr int NI = 2;
    constexpr int A_CH = BM * 8 / NT;
    constexpr int B_CH = BN * 8 / NT;
    constexpr int STAGE_BYTES = (BM + BN) * 128;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    const int M = g.M, N = g.N, K = g.K;
    const int tiles_m = (M + BM - 1) / BM;
    const int tiles_n = N / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    if (tiles_m <= tiles_n) {
        tn = bid / tiles_m;
        tm = bid - tn * tiles_m;
    } else {
        tm = bid / tiles_n;
        tn = bid - tm * tiles_n;
    }
    const int m0 = tm * BM;
    const int n0 = tn * BN;

    // per-lane source pointers (swizzle folded in); LDS destinations are wave-uniform
    const op_t* a_ptr[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        int q = i * NT + tid;
        int row = q >> 3, pos = q & 7;
        int c = pos ^ ((row >> 1) & 7);
        int gm = m0 + row;
        gm = gm < M ? gm : M - 1;
        a_ptr[i] = g.A + (size_t)gm * K + c * 8;
    }
    const op_t* b_ptr[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        int q = i * NT + tid;
        int row = q >> 3, pos = q & 7;
        int c = pos ^ ((row >> 1) & 7);
        b_ptr[i] = g.W + (size_t)(n0 + row) * K + c * 8;
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto stage_in = [&](int kt, int stage) {
        char* sa = smem + stage * STAGE_BYTES;
        char* sb = sa + BM * 128;
#pragma unroll
        for (int i = 0; i < A_CH; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_ptr[i] + kt * 64),
                                             (__attribute__((address_space(3))) void*)(sa + (i * NT + wave * 64) * 16), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < B_CH; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_ptr[i] + kt * 64),
                                             (__attribute__((address_space(3))) void*)(sb + (i * NT + wave * 64) * 16), 16, 0, 0);
    };
    auto compute = [&](int stage) {
        const char* sa = smem + stage * STAGE_BYTES;
        const char* sb = sa + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            opx8 af[MI], bfr[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                af[i] = *reinterpret_cast<const opx8*>(sa + lds_tile_off(wm * TM + i * 32 + l31, ks * 2 + half));
#pragma unroll
            for (int j = 0; j < NI; ++j)
                bfr[j] = *reinterpret_cast<const opx8*>(sb + lds_tile_off(wn * TN + j * 32 + l31, ks * 2 + half));
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = mfma_32x32x16(af[i], bfr[j], acc[i][j]);
        }
    };

    const int nk = K / 64;
    stage_in(0, 0);
    __syncthreads();          // hipcc emits s_waitcnt vmcnt(0) in front of the barrier while an LDS-DMA is in flight
    for (int kt = 0; kt < nk - 1; ++kt) {
        stage_in(kt + 1, (kt + 1) & 1);
        compute(kt & 1);
        __syncthreads();
    }
    compute((nk - 1) & 1);

    gemm_epilogue<EPI, MI, NI>(g, acc, m0 + wm * TM, n0 + wn * TN, half, l31);
}

// ---------------------------------------------------------------------------------------------
// Deep-prefetch variant: NS-stage LDS ring filled by LDS-DMA, prefetch distance NS-1, ONE raw
// s_barrier per K-tile and COUNTED s_waitcnt vmcnt (never 0 in steady state) so that NS-2 tiles
// stay in flight across every barrier (guide T3+T4).  This is what hides the L2/HBM -> LDS latency
// when only one workgroup fits a CU (B=1: 204..432 workgroups on 256 CUs).
//   iteration k:  wait(tile k landed) ; barrier ; issue tile k+NS-1 -> stage (k-1)%NS ; compute k
// BK = 64 (128-B rows, 8 chunks) or 32 (64-B rows, 4 chunks; swizzle c ^ ((row>>2)&3)).
// ---------------------------------------------------------------------------------------------
template <int BK>
__device__ __forceinline__ int lds_off_bk(int row, int chunk) {
    if constexpr (BK == 128) return row * 256 + ((chunk ^ (row & 15)) << 4);      // 256-B rows: 16 chunks, XOR with the row
    else if constexpr (BK == 64) return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
    else return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
}

// (The ablation modes of rounds 1-2 -- no LDS-DMA / no ds_read / no barrier / no epilogue builds of this kernel -- are in the git
// history at 9661dcc; what they measured is in profiles/r01_gemm_ablation.txt, r02_gemm_ablation.txt.)
// FP8: A and W are e4m3 bytes.  The launcher hands the kernel K/2 "bf16 columns", so staging, LDS layout and swizzle are
// byte-for-byte those of the bf16 kernel (128-B rows now hold 128 k); a 16-B fragment is two 8-byte MFMA operands:
// v_mfma_f32_32x32x16_fp8_fp8 runs at the bf16 rate, but every LDS / L2 / HBM byte carries twice the k.
// KG = 2 (round 4, fp32 output): TWO K-groups of WM x WN waves work on the same BM x BN tile, group g on the k-half g of every
// 2 x BK slice (its own A / W rows in the stage), and exchange half of their partial sums through LDS at the end -- each group then
// finishes 32 of its waves' 64 rows with the 32 x 64 staged epilogue.  A one-round launch (<= one tile per CU) gets 8 waves of 64 x 64
// per CU instead of 8 waves of 32 x 64: a third less LDS fragment traffic per MFMA, which is what bounds the narrow tiles
// (profiles/r04_narrow_tiles_negative.txt: two co-resident 4-wave workgroups of 64 x 64 waves beat the 8-wave tile by 17-21 % wherever
// there are two workgroups per CU; at one prompt there is only one).
// (Measured and not kept, profiles/r04_kgroup_tile.txt: four PRODUCER waves -- one per SIMD, two per K-group -- that issue all of the LDS-DMA while the eight
// MFMA waves never touch the vector-memory queue.  Bit-identical results, FF-out 66.5 us against 53.7: the CU's vector-memory path takes one 1-KiB piece per
// ~16 clocks whoever issues it, and with a two-stage ring the issue of tile k + 1 can only start at boundary k and has to land by boundary k + 1 -- sixteen
// pieces in a row per producer wave stretch that chain, eight per wave in parallel with the MFMAs do not.  The ablation without any DMA runs 36.9 us.)
// DIL ("DMA in loop", bf16 / fp16 operands): the iteration's LDS-DMA pieces are issued inside compute(), behind the MFMAs of the first k-steps, instead of
// in front of it -- always on with K-groups; a per-tile choice otherwise (measured, see launch_epi)
template <int BM, int BN, int BK, int WM, int WN, int NS, int EPI, int FP8 = 0, int KG = 1, bool DIL = false>
__global__ __launch_bounds__(KG * WM * WN * 64) void gemm_pipe_kernel(GemmArgs g) {
    sat_f16_saturate();
    static_assert(KG == 1 || (KG == 2 && EPI == EPI_F32 && FP8 == 0 && BK == 64 && BM / WM == 64 && BN / WN == 64),
                  "K-groups: fp32 output, 64 x 64 wave tiles, 128-byte rows");
    constexpr int NT = KG * WM * WN * 64;
    constexpr int TM = BM / WM;
    constexpr int TN = BN / WN;
    // the fused epilogues need a whole head / a value-gate pair per wave (64 columns); plain fp32 output takes any 32-multiple
    static_assert(TN % 32 == 0 && (TN == 64 || EPI == EPI_F32), "wave tile is TM x 64 (TM x 32k for EPI_F32)");
    constexpr int MI = TM / 32;
    constexpr int NI = TN / 32;
    constexpr int CPR = BK / 8;                    // 16-B chunks per row
    // One LDS-DMA instruction of one wave moves 64 chunks = 1 KiB.  A stage holds WL_A + WL_B of them (A rows first, then
    // W rows, contiguous); wave w issues wave-loads w, w+NW, w+2NW, ...  When NW does not divide WL (256x192 on 12 waves:
    // 56 wave-loads) the first WL%NW waves carry one more than the rest, and the counted vmcnt wait is per wave.
    constexpr int NW = KG * WM * WN;                  // every wave loads
    constexpr int WL_A = BM * CPR / 64;
    constexpr int WLG = (BM + BN) * CPR / 64;         // wave-loads of one K-group's rows of a stage
    constexpr int WL = KG * WLG;
    constexpr int LPT = (WL + NW - 1) / NW;        // max LDS-DMA instructions per tile per wave
    constexpr int N_FULL = WL - (LPT - 1) * NW;    // waves [0, N_FULL) issue LPT, the others LPT-1
    constexpr bool UNIFORM = (WL % NW) == 0;
    constexpr int ROWB = BK * 2;
    constexpr bool MXA = FP8 == 3;                 // MXFP8 A operand: + one dword of E8M0 block scales per A row per stage
    constexpr int SCALE_WAVES = MXA ? BM / 64 : 0; // the first BM/64 waves each DMA 64 scale dwords per stage
    constexpr int GROUP_BYTES = (BM + BN) * ROWB + (MXA ? BM * 4 : 0);
    constexpr int STAGE_BYTES = KG * GROUP_BYTES;
    constexpr int D = NS - 1;                      // prefetch distance
    constexpr bool DIL_ON = (DIL || KG == 2) && FP8 == 0;
    constexpr int NDIL = !DIL_ON ? 0 : UNIFORM ? LPT : LPT - 1;          // pieces EVERY wave issues in the loop body (a ragged tile's extra piece stays in front)
    constexpr int DIL_STEPS = KG == 2 ? BK / 32 : BK / 16 - 1;            // k-steps that carry them (K-groups: the half in front of the mid-iteration barrier)
    constexpr int DIL_PER = (NDIL + DIL_STEPS - 1) / (DIL_STEPS > 0 ? DIL_STEPS : 1);
    static_assert(!MXA || D == 1 || UNIFORM, "MXFP8: counted waits are built for uniform tiles or 2-stage rings");
    static_assert((BM * CPR) % 64 == 0 && (BN * CPR) % 64 == 0 && (D - 1) * LPT < 64, "bad pipeline geometry");
    static_assert(!FP8 || BK == 64, "fp8: 128-byte rows only");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = KG == 2 ? wave / (WM * WN) : 0;           // K-group
    const int wv = KG == 2 ? wave % (WM * WN) : wave;
    const int wm = wv / WN;
    const int wn = wv % WN;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    const int M = g.M, N = g.N, K = g.K;
    const int tiles_m = (M + BM - 1) / BM;
    const int tiles_n = N / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    if (tiles_m <= tiles_n) {
        tn = bid / tiles_m;
        tm = bid - tn * tiles_m;
    } else {
        tm = bid / tiles_n;
        tn = bid - tm * tiles_n;
    }
    const int m0 = tm * BM;
    const int n0 = tn * BN;
    // waves whose TM rows lie entirely beyond M (the M-tail tile: 2 valid rows of 256 at B=1) keep staging tiles and
    // joining barriers but skip their LDS reads, MFMAs and epilogue
    const bool wave_rows_valid = (m0 + wm * TM) < M;
    // Fused cross-attention (HeadsEpi::xa_k, 128 x 64 tile: one head per workgroup column, one 32-query block per wave): the K / V^T
    // tiles of the (sequence, kv-head) go to LDS behind the ring and the LayerNorm constants NOW -- they are the oldest entries of the
    // vector-memory queue, so every counted vmcnt wait of the K loop still holds, and the loop's barriers publish them.
    constexpr bool XA_OK = EPI == EPI_HEADS && MI == 1 && NI == 2 && NW == 4 && BN == 64 && BK == 64 && NS == 3 && FP8 == 0;
    constexpr int XA_OFF = (NS * STAGE_BYTES + (BM + BN) * 8 + 1023) & ~1023;
    [[maybe_unused]] bool xa_on = false;
    [[maybe_unused]] auto xa_stage = [&](int b) {
        const HeadsEpi& he = g.heads;
        const int kvh = (n0 >> 6) / (he.heads / he.xa_kvh);
        const op_t* kbase = he.xa_k + (size_t)(b * he.xa_kvh + kvh) * he.xa_sk_pad * 64;
        const op_t* vbase = he.xa_vt + (size_t)(b * he.xa_kvh + kvh) * 64 * he.xa_sk_pad;
        const int n_t = (((b * he.xa_sk) & 3) + he.xa_sk + 63) >> 6;
        for (int t = 0; t < n_t; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int p = wave + 4 * i;                          // 1-KiB piece = 8 rows of 128 B
                const int row = p * 8 + (lane >> 3);
                const int c = (lane & 7) ^ ((row >> 1) & 7);         // source-side XOR swizzle (lds_tile_off)
                char* dst = smem + XA_OFF + t * 16384 + p * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kbase + (size_t)(t * 64 + row) * 64 + c * 8),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vbase + (size_t)row * he.xa_sk_pad + t * 64 + c * 8),
                                                 (__attribute__((address_space(3))) void*)(dst + 8192), 16, 0, 0);
            }
    };
    if constexpr (XA_OK) {
        xa_on = g.heads.xa_k != nullptr;
        if (xa_on) xa_stage(m0 / g.heads.S);
    }
    // fp32 residual epilogue of the small tiles (one 32-row block per wave, registers to spare): fetch the residual values NOW.
    // They are the oldest entries of the vector-memory queue, so every counted vmcnt wait of the K loop still holds.
    constexpr bool PRE_RESID = EPI == EPI_F32 && (MI == 1 || KG == 2) && NI == 2 && NT <= 512;
    // K-groups: the 32 rows this wave finishes
    const int epi_m = m0 + wm * TM + (KG == 2 ? grp * 32 : 0);
    const bool epi_rows_valid = KG == 2 ? epi_m < M : wave_rows_valid;
    [[maybe_unused]] f32x4 resid[PRE_RESID ? 8 : 1];
    if constexpr (PRE_RESID) {
        // (exactly the condition under which the staged epilogue runs, see the end of the kernel)
#ifdef SAT_GEMM_EXPERIMENTS
        const bool staged = !(!MXA && !(g.variant & 0x1000) && (g.variant & 0x2000)) && !(g.variant & 0x8000);
#else
        const bool staged = true;
#endif
        const bool want = g.accumulate != 0 && epi_rows_valid && staged;
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int m = epi_m + ps * 4 + (lane >> 4);
            resid[ps] = (want && m < M) ? *reinterpret_cast<const f32x4*>(g.C + (size_t)m * g.ldc + n0 + wn * TN + (lane & 15) * 4)
                                        : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    // LayerNorm fold, consumer side: mean and 1/std of the BM rows of this tile from the producer's per-block partial sums, into
    // LDS behind the ring (filled right after the prologue's LDS-DMA below, read by the epilogue).
    constexpr bool LN_CONS = (EPI == EPI_SWIGLU || EPI == EPI_HEADS) && FP8 == 0;
    // LDS behind the ring: (mean, 1/std) of the BM rows, then the BN (c1, c2) pairs of this tile's output channels
    [[maybe_unused]] float2* lnst = reinterpret_cast<float2*>(smem + NS * STAGE_BYTES);
    [[maybe_unused]] float* lnc = reinterpret_cast<float*>(smem + NS * STAGE_BYTES + BM * 8);      // c1[BN] then c2[BN]
    const op_t* ld_ptr[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int L = KG == 2 ? grp * WLG + i * (WM * WN) + wv : i * NW + wave;               // wave-uniform; K-groups: a group stages its own rows
        const int lg = KG == 2 ? grp : 0;
        const int Lg = KG == 2 ? i * (WM * WN) + wv : L;
        const bool is_a = Lg < WL_A;
        int q = (is_a ? Lg : Lg - WL_A) * 64 + lane;
        int row = q / CPR, pos = q % CPR;
        int c = (BK == 128) ? (pos ^ (row & 15)) : (BK == 64) ? (pos ^ ((row >> 1) & 7)) : (pos ^ ((row >> 2) & 3));
        int gm = m0 + row;
        gm = gm < M ? gm : M - 1;
        int gn = n0 + row;
        gn = gn < N ? gn : N - 1;                  // only reachable by the unused slot of a short wave
        ld_ptr[i] = (is_a ? g.A + (size_t)gm * K + c * 8 : g.W + (size_t)gn * K + c * 8) + lg * BK;
    }
    const bool wave_full = UNIFORM || wave < N_FULL;
    [[maybe_unused]] const unsigned* sc_ptr = nullptr;
    if constexpr (MXA) {
        int gm = m0 + wave * 64 + lane;
        gm = gm < M ? gm : M - 1;
        sc_ptr = g.a_bscale + (size_t)gm * (K / BK);      // [M][K/128 bytes] dwords; K counts 16-bit columns here
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto stage_in = [&](int kt, int stage) {
        char* sa = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < LPT; ++i)
            if (UNIFORM || i + 1 < LPT || wave_full)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ld_ptr[i] + kt * (BK * KG)),
                                                 (__attribute__((address_space(3))) void*)(sa + (KG == 2 ? grp * WLG + i * (WM * WN) + wv : i * NW + wave) * 1024), 16, 0, 0);
        if constexpr (MXA) {
            if (wave < SCALE_WAVES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sc_ptr + kt),
                                                 (__attribute__((address_space(3))) void*)(sa + (BM + BN) * ROWB + wave * 256), 4, 0, 0);
        }
    };
    // tiles k+1..k+D-1 may stay in flight: this wave issued LPT (or LPT-1) loads for each of them
    [[maybe_unused]] auto wait_steady = [&]() {
        if constexpr (MXA && D > 1) {
            if (wave < SCALE_WAVES) wait_vmcnt<(D - 1) * (LPT + 1)>();
            else wait_vmcnt<(D - 1) * LPT>();
        } else if constexpr (UNIFORM || D == 1) {
            wait_vmcnt<(D - 1) * LPT>();
        } else {
            if (wave_full) wait_vmcnt<(D - 1) * LPT>();
            else wait_vmcnt<(D - 1) * (LPT - 1)>();
        }
    };
    // fragments are double-buffered across the k-steps: the ds_reads of step ks+1 are issued BEFORE the MFMAs of
    // step ks, so the wave waits with a counted lgkmcnt and LDS latency hides behind the matrix pipe
    typedef long i64x2 __attribute__((ext_vector_type(2)));
    // TRc: operands swapped -> acc holds C^T (lane = token row), see gemm_epilogue_t
    auto compute = [&](int stage, auto trc, [[maybe_unused]] int kt_issue = -1, [[maybe_unused]] int st_issue = 0) {
        constexpr bool TRc = decltype(trc)::value;
        const char* sa = smem + stage * STAGE_BYTES + grp * GROUP_BYTES;
        const char* sb = sa + BM * ROWB;
        constexpr int KS = BK / 16;
        if constexpr (FP8 >= 2) {
            // v_mfma_scale_f32_32x32x64_f8f6f4: twice the MFMA rate of bf16 / plain fp8.  One instruction consumes 64 k = two scale
            // blocks of 32.  Measured layout (tools/mx_probe.cpp): the FIRST 16 bytes of every lane belong to block 0, the LAST 16 to
            // block 1, and block g takes its E8M0 scale from the lanes of half g.  So lane (row, half) loads the 16-byte chunks
            // `half` and `2 + half` of the 64-byte step (k = half*16.. and 32 + half*16..), identically for A and W.
            typedef int i32x8 __attribute__((ext_vector_type(8)));
            constexpr int KS2 = BK / 32;               // 64-byte steps per 128-byte row
            i32x8 af[2][MI], bfr[2][NI];
            auto frag = [&](int ks, int buf) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int row = wm * TM + i * 32 + l31;
                    u32x4 lo = *reinterpret_cast<const u32x4*>(sa + lds_off_bk<BK>(row, ks * 4 + half));
                    u32x4 hi = *reinterpret_cast<const u32x4*>(sa + lds_off_bk<BK>(row, ks * 4 + 2 + half));
                    af[buf][i] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int row = wn * TN + j * 32 + l31;
                    u32x4 lo = *reinterpret_cast<const u32x4*>(sb + lds_off_bk<BK>(row, ks * 4 + half));
                    u32x4 hi = *reinterpret_cast<const u32x4*>(sb + lds_off_bk<BK>(row, ks * 4 + 2 + half));
                    bfr[buf][j] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                }
            };
            // 16-wave workgroups have 128 VGPRs per lane: 64 accumulators + one set of 32-byte fragments (32) fit, two sets do not;
            // four waves per SIMD cover the fragment-read latency instead
            constexpr bool DBUF = NT < 1024;
            // MXFP8 A: dword of this row's four E8M0 block scales of the K-tile, shifted so that byte 0 / byte 2 are the blocks this
            // lane half feeds in 64-byte step 0 / 1 (block = 2 * step + half)
            [[maybe_unused]] int sdw[MI];
            if constexpr (MXA) {
                const char* ssc = sb + BN * ROWB;
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    sdw[i] = (int)(*reinterpret_cast<const unsigned*>(ssc + (wm * TM + i * 32 + l31) * 4) >> (8 * half));
            }
            auto mx = [&](const i32x8& af_, const i32x8& bf_, f32x16 c_, int ks, int i) -> f32x16 {
                if constexpr (MXA) {
                    static_assert(!TRc, "MXFP8 A operand: block scales were probed on the A side only (tools/mx_probe.cpp)");
                    if (ks == 0) return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af_, bf_, c_, 0, 0, 0, sdw[i], 0, 0x7F7F7F7F);
                    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af_, bf_, c_, 0, 0, 2, sdw[i], 0, 0x7F7F7F7F);
                } else if constexpr (TRc) {
                    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bf_, af_, c_, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                } else {
                    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af_, bf_, c_, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                }
            };
            if constexpr (!DBUF) {
#pragma unroll
                for (int ks = 0; ks < KS2; ++ks) {
                    frag(ks, 0);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            acc[i][j] = mx(af[0][i], bfr[0][j], acc[i][j], ks, i);
                }
                return;
            }
            frag(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MI + NI), 0);
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                if (ks + 1 < KS2) frag(ks + 1, (ks + 1) & 1);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = mx(af[ks & 1][i], bfr[ks & 1][j], acc[i][j], ks, i);
                if (ks + 1 < KS2) {
                    constexpr int NR = 2 * (MI + NI), NM = MI * NI;
#pragma unroll
                    for (int r = 0; r < NM; ++r) {
                        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, NR / NM, 0);
                    }
                    if (NR % NM) __builtin_amdgcn_sched_group_barrier(0x100, NR % NM, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x8, MI * NI, 0);
                }
            }
            return;
        }
        if constexpr (FP8 == 1) {
            // fragment step ks = 32 k (two MFMAs): lane (row, half) holds bytes half*16 .. +15 of it, the first 8 feed one MFMA and
            // the last 8 the next -- A and W use the same k permutation, so the dot product is unchanged
            i64x2 af[2][MI], bfr[2][NI];
            auto frag = [&](int ks, int buf) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    af[buf][i] = *reinterpret_cast<const i64x2*>(sa + lds_off_bk<BK>(wm * TM + i * 32 + l31, ks * 2 + half));
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    bfr[buf][j] = *reinterpret_cast<const i64x2*>(sb + lds_off_bk<BK>(wn * TN + j * 32 + l31, ks * 2 + half));
            };
            frag(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) frag(ks + 1, (ks + 1) & 1);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            acc[i][j] = TRc ? __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(bfr[ks & 1][j][h2], af[ks & 1][i][h2], acc[i][j], 0, 0, 0)
                                            : __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(af[ks & 1][i][h2], bfr[ks & 1][j][h2], acc[i][j], 0, 0, 0);
                if (ks + 1 < KS) {
#pragma unroll
                    for (int r = 0; r < MI + NI; ++r) {
                        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x8, 2 * MI * NI - (MI + NI), 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x8, 2 * MI * NI, 0);
                }
            }
            return;
        }
        opx8 af[2][MI], bfr[2][NI];
        auto frag = [&](int ks, int buf) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
                af[buf][i] = *reinterpret_cast<const opx8*>(sa + lds_off_bk<BK>(wm * TM + i * 32 + l31, ks * 2 + half));
#pragma unroll
            for (int j = 0; j < NI; ++j)
                bfr[buf][j] = *reinterpret_cast<const opx8*>(sb + lds_off_bk<BK>(wn * TN + j * 32 + l31, ks * 2 + half));
        };
        frag(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);        // the first fragments: DS reads only
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) frag(ks + 1, (ks + 1) & 1);
            if constexpr (DIL_ON) {
                // the iteration's LDS-DMA pieces ride in the MFMA stream of the first k-steps -- in front of the first MFMA they cost their
                // issue time, 60-180 cycles a piece
                if (ks < DIL_STEPS && kt_issue >= 0) {
                    char* sdst = smem + st_issue * STAGE_BYTES;
#pragma unroll
                    for (int i = ks * DIL_PER; i < (ks + 1) * DIL_PER && i < NDIL; ++i)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ld_ptr[i] + kt_issue * (BK * KG)),
                                                         (__attribute__((address_space(3))) void*)(sdst + (KG == 2 ? grp * WLG + i * (WM * WN) + wv : i * NW + wave) * 1024), 16, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = TRc ? mfma_32x32x16(bfr[ks & 1][j], af[ks & 1][i], acc[i][j])
                                    : mfma_32x32x16(af[ks & 1][i], bfr[ks & 1][j], acc[i][j]);
            // pin the interleave: one ds_read of the NEXT step's fragments behind each MFMA of this step
            if (ks + 1 < KS) {
                constexpr int NR = MI + NI, NM = MI * NI;
                if constexpr (KG == 2) {
                    // (measured on the 12-wave 256 x 192 tile as well: 42.9-43.4 us against 41.1-41.5 for one read per MFMA -- three waves per SIMD
                    // cover the trailing read, two do not)
                    // all of the next step's fragment reads behind the FIRST MFMA: they get three MFMAs (~100 cycles) of head start on the
                    // wait in front of the next step instead of one read trailing the last MFMA
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
                    if (ks < KS / 2 && DIL_PER == 4) {
                        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x20, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x20, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x8, NM - 1, 0);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < (NR < NM ? NR : NM); ++r) {
                        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    if (NM > NR) __builtin_amdgcn_sched_group_barrier(0x8, NM - NR, 0);
                    if (NR > NM) __builtin_amdgcn_sched_group_barrier(0x100, NR - NM, 0);
                    if (DIL_ON && ks < DIL_STEPS) __builtin_amdgcn_sched_group_barrier(0x20, DIL_PER, 0);          // this step's DMA pieces behind its MFMAs
                }
            } else {
                __builtin_amdgcn_sched_group_barrier(0x8, MI * NI, 0);
            }
            if constexpr (KG == 2) {
                if (ks == KS / 2 - 1) {          // the OTHER group's iteration boundary (it runs half an iteration out of phase)
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };

    const int nk = K / (BK * KG);
    // prologue: tiles 0..D-1 in flight (nk >= D is guaranteed by the launcher)
#pragma unroll
    for (int s = 0; s < D; ++s) stage_in(s, s);
    // (after the prologue's LDS-DMA: the constants' global-memory round trip overlaps the first tiles' instead of preceding it; the
    // K loop's barriers order these LDS writes before the epilogue's reads)
    if constexpr (LN_CONS) {
        // Without the fold the same epilogue runs on (mean, rstd) = (0, 1), c1 = 0, c2 = bias.
        const bool ln_fold = g.ln_part != nullptr;
        // TPR threads per row, each summing every TPR-th partial pair (independent loads, one round trip), combined with DPP
        constexpr int TPR = (NT / BM >= 4) ? 4 : (NT / BM >= 2 ? 2 : 1);
        const int np = K >> 6;
        const int r = tid / TPR, sub = tid % TPR;
        float sum = 0.f, sq = 0.f;
        if (ln_fold && r < BM) {
            int m = m0 + r;
            m = m < M ? m : M - 1;
            const float2* pp = reinterpret_cast<const float2*>(g.ln_part) + (size_t)m * np;
#pragma unroll 6
            for (int i = sub; i < np; i += TPR) {
                const float2 v = pp[i];
                sum += v.x;
                sq += v.y;
            }
        }
        if constexpr (TPR >= 2) {
            sum += dpp_move<0xB1>(sum);
            sq += dpp_move<0xB1>(sq);
        }
        if constexpr (TPR >= 4) {
            sum += dpp_move<0x4E>(sum);
            sq += dpp_move<0x4E>(sq);
        }
        if (r < BM && sub == 0) {
            const float inv_k = 1.0f / (float)K;
            const float mean = sum * inv_k;
            const float var = fmaxf(sq * inv_k - mean * mean, 0.f);
            lnst[r] = ln_fold ? make_float2(mean, rsqrtf(var + g.ln_eps)) : make_float2(0.f, 1.f);
        }
        // the last 2 BN / 4 threads bring in the channel constants, 16 bytes each
        const int ct = NT - 1 - tid;
        if (ct < BN / 2) {
            const bool first = ct < BN / 4;
            const float* src = ln_fold ? (first ? g.ln_c1 + n0 + ct * 4 : g.ln_c2 + n0 + (ct - BN / 4) * 4)
                                       : ((first || !g.bias) ? nullptr : g.bias + n0 + (ct - BN / 4) * 4);
            *reinterpret_cast<f32x4*>(lnc + ct * 4) = src ? *reinterpret_cast<const f32x4*>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    int rd = 0;          // stage of tile k
    int wr = D;          // stage that receives tile k+D (== stage of tile k-1)
    // steady state: tiles k+1..k+D-1 stay in flight across the barrier
    // Orientation (uniform over the workgroup: tiles never straddle a q / k / v part): transposed accumulators everywhere except
    // for a V^T destination (token-contiguous stores want lane = channel), the MXFP8 A operand and the fp32 output.
    // Measured (profiles/r02_epilogue_ab.txt): bf16 outputs gain 3-4 % (SwiGLU) / 13 % (heads) from the transposed orientation, the
    // fp32 residual epilogue LOSES 6-10 % at 8 prompts (a lane-per-token store instruction touches 32 cache lines; in the legacy
    // orientation every store instruction writes two full 128-byte lines) -> fp32 output stays un-swapped.
    // The shipped build fixes the orientation at compile time wherever it can (ORI 0: un-swapped only -- fp32 output, MXFP8 A
    // operand; 1: transposed only -- SwiGLU; 2: per workgroup -- heads: q / k transposed, V^T un-swapped): one main loop instead of
    // two keeps the 128-VGPR kernels (16 waves, e4m3 fragments) out of scratch.  The experiments build keeps both behind variant
    // bits 12 (force un-swapped), 13 (transposed fp32 epilogue) and 15 (direct dword fp32 epilogue) for A/B measurements.
#ifdef SAT_GEMM_EXPERIMENTS
    constexpr int ORI = MXA ? 0 : 2;
    bool tr = !MXA && !(g.variant & 0x1000) && (EPI != EPI_F32 || (g.variant & 0x2000));
    const bool direct_f32 = (g.variant & 0x8000) != 0;
#else
    constexpr int ORI = (MXA || EPI == EPI_F32) ? 0 : (EPI == EPI_SWIGLU ? 1 : 2);
    bool tr = ORI != 0;
    constexpr bool direct_f32 = false;
#endif
    if constexpr (EPI == EPI_HEADS) {
        const int hp = g.heads.heads * 64;
        tr = tr && !(g.heads.kind[(n0 + wn * TN) / hp] & 1);
    }
    auto main_loop = [&](auto trc) {
        // K-groups: a group stages and reads only its own rows, so every barrier only has to hold for the group -- group 1 runs half an
        // iteration behind group 0 (one barrier more up front, one less at the end; compute() passes one in mid-iteration), and on every
        // SIMD one wave is inside its MFMA stream while the other waits for its tile, issues the next and reads its first fragments
        if (KG == 2 && grp == 1) __builtin_amdgcn_s_barrier();
        for (int k = 0; k < nk - D; ++k) {
            wait_steady();
            __builtin_amdgcn_s_barrier();
            if constexpr (DIL_ON) {
                if (wave_rows_valid) {
                    if constexpr (!UNIFORM) {          // the extra piece of the first waves of a ragged tile
                        if (wave_full)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ld_ptr[LPT - 1] + (k + D) * (BK * KG)),
                                                             (__attribute__((address_space(3))) void*)(smem + wr * STAGE_BYTES + ((LPT - 1) * NW + wave) * 1024), 16, 0, 0);
                    }
                    compute(rd, trc, k + D, wr);
                } else {
                    stage_in(k + D, wr);
                    if (KG == 2) __builtin_amdgcn_s_barrier();
                }
            } else {
                stage_in(k + D, wr);
                if (wave_rows_valid) compute(rd, trc);
                else if (KG == 2) __builtin_amdgcn_s_barrier();
            }
            rd = (rd + 1 == NS) ? 0 : rd + 1;
            wr = (wr + 1 == NS) ? 0 : wr + 1;
        }
        // drain: nothing left to issue
        for (int k = nk - D; k < nk; ++k) {
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (wave_rows_valid) compute(rd, trc);
            else if (KG == 2) __builtin_amdgcn_s_barrier();
            rd = (rd + 1 == NS) ? 0 : rd + 1;
        }
        if (KG == 2 && grp == 0) __builtin_amdgcn_s_barrier();
    };
    if constexpr (ORI == 0) {
        main_loop(std::false_type{});
    } else if constexpr (ORI == 1) {
        main_loop(std::true_type{});
    } else {
        if (tr) main_loop(std::true_type{});
        else main_loop(std::false_type{});
    }

    if constexpr (XA_OK) {
        if (xa_on) {
            const HeadsEpi& he = g.heads;
            unsigned qp[2][8];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) qp[j][e] = 0u;
            if (wave_rows_valid) gemm_epilogue_t<EPI, MI, NI, LN_CONS>(g, acc, m0 + wm * TM, n0, half, l31, lnst + wm * TM, lnc, lnc + BN, qp);
            opx8 qf[4];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                qf[2 * j] = __builtin_bit_cast(opx8, u32x4{qp[j][0], qp[j][1], qp[j][2], qp[j][3]});
                qf[2 * j + 1] = __builtin_bit_cast(opx8, u32x4{qp[j][4], qp[j][5], qp[j][6], qp[j][7]});
            }
            const int S = he.S;
            const int m = m0 + wm * TM + l31;
            const int b_me = (m < M ? m : M - 1) / S;
            const int last = (m0 + BM - 1 < M ? m0 + BM - 1 : M - 1);
            const int b_lo = m0 / S, b_hi = last / S;
            const int wlast = (m0 + wm * TM + 31 < M ? m0 + wm * TM + 31 : M - 1);
            const int w_lo = (m0 + wm * TM) / S, w_hi = wlast / S;
            for (int bb = b_lo; bb <= b_hi; ++bb) {
                if (bb > b_lo) {        // the tile's rows straddle two sequences: second pass on the next sequence's keys
                    __builtin_amdgcn_s_barrier();
                    xa_stage(bb);
                    wait_vmcnt<0>();
                    __builtin_amdgcn_s_barrier();
                }
                if (!(wave_rows_valid && w_lo <= bb && bb <= w_hi)) continue;
                const int ob = (bb * he.xa_sk) & 3, k_end = ob + he.xa_sk;
                const int n_t = (k_end + 63) >> 6;
                attn::State<2> st;
                st.init(half);
                for (int t = 0; t < n_t; ++t) {
                    const char* sk = smem + XA_OFF + t * 16384;
                    const bool edge = (t == 0 && ob != 0) || (t == n_t - 1 && (k_end & 63) != 0);
                    attn::tile<2>(st, qf, sk, sk + 8192, edge, t * 64, ob, k_end, t == 0, 1.0f, l31, half);
                }
                const float inv = 1.0f / attn::half_sum(st.l_run);
                op_t* op = he.xa_out + (size_t)m * ((size_t)he.heads * 64) + n0 + 8 * half;      // out row = b * S + s = m
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    unsigned pk[8];
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        pk[2 * rq] = pack_op2(st.oacc[db][rq * 4] * inv, st.oacc[db][rq * 4 + 1] * inv);
                        pk[2 * rq + 1] = pack_op2(st.oacc[db][rq * 4 + 2] * inv, st.oacc[db][rq * 4 + 3] * inv);
                    }
                    half_swap(pk[0], pk[2]);            // 8 consecutive channels per lane: 16-byte stores
                    half_swap(pk[1], pk[3]);
                    half_swap(pk[4], pk[6]);
                    half_swap(pk[5], pk[7]);
                    if (m < M && b_me == bb) {
                        *reinterpret_cast<u32x4*>(op + db * 32) = u32x4{pk[0], pk[1], pk[2], pk[3]};
                        *reinterpret_cast<u32x4*>(op + db * 32 + 16) = u32x4{pk[4], pk[5], pk[6], pk[7]};
                    }
                }
            }
            return;
        }
    }
    if constexpr (FP8 != 0) {
        // dequantise: per-token scale of A x per-output-channel scale of W (kept out of the last K-tile's MFMA schedule)
        __builtin_amdgcn_sched_barrier(0);
        if (wave_rows_valid) {
            if (tr) {      // lane = token row, registers = channels 8*(r>>2) + 4*half + (r&3)
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    int row = m0 + wm * TM + i * 32 + l31;
                    row = row < M ? row : M - 1;
                    const float sa_r = g.a_scale[row];
#pragma unroll
                    for (int j = 0; j < NI; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 sw = *reinterpret_cast<const f32x4*>(g.w_scale + n0 + wn * TN + j * 32 + q * 8 + 4 * half);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] *= sa_r * sw[e];
                        }
                }
            } else {
                float sw[NI];
#pragma unroll
                for (int j = 0; j < NI; ++j) sw[j] = g.w_scale[n0 + wn * TN + j * 32 + l31];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        row = row < M ? row : M - 1;
                        const float sa_r = MXA ? 1.0f : g.a_scale[row];
#pragma unroll
                        for (int j = 0; j < NI; ++j) acc[i][j][r] *= sa_r * sw[j];
                    }
            }
        }
    }
    if constexpr (KG == 2) {
        // exchange: group 0 keeps row block 0 of its 64 x 64 wave tile and sends block 1, group 1 the other way round; lane-contiguous
        // dwords, 8 KiB per wave in the first half of the (free) ring, the staged epilogue's areas in the second half
        __builtin_amdgcn_s_barrier();
        float* mine = reinterpret_cast<float*>(smem) + wave * 2048;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[(j * 16 + r) * 64 + lane] = grp ? acc[0][j][r] : acc[1][j][r];
        __syncthreads();
        const float* theirs = reinterpret_cast<const float*>(smem) + (wave ^ (WM * WN)) * 2048;
        f32x16 fin[1][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) fin[0][j][r] = (grp ? acc[1][j][r] : acc[0][j][r]) + theirs[(j * 16 + r) * 64 + lane];
        static_assert(KG == 1 || NS * STAGE_BYTES >= 2 * NW * 8192, "exchange + staging areas");
        if (epi_rows_valid)
            gemm_epilogue_f32_staged<1, true>(g, fin, reinterpret_cast<float*>(smem + NW * 8192) + wave * 2048, epi_m, n0 + wn * TN, lane, resid);
        return;
    }
    if constexpr (EPI == EPI_F32 && NI == 2) {
        if (!tr && !direct_f32) {
            __builtin_amdgcn_s_barrier();       // every wave is done reading the ring: it becomes the staging area
            if (wave_rows_valid) {
                if constexpr (PRE_RESID)
                    gemm_epilogue_f32_staged<MI, true>(g, acc, reinterpret_cast<float*>(smem) + wave * 2048, m0 + wm * TM, n0 + wn * TN, lane, resid);
                else
                    gemm_epilogue_f32_staged<MI>(g, acc, reinterpret_cast<float*>(smem) + wave * 2048, m0 + wm * TM, n0 + wn * TN, lane);
            }
            return;
        }
    }
    if (wave_rows_valid) {
        if constexpr (NI == 2 || EPI == EPI_F32) {
            if (tr) gemm_epilogue_t<EPI, MI, NI, LN_CONS, (NT < 1024)>(g, acc, m0 + wm * TM, n0 + wn * TN, half, l31, lnst + wm * TM, lnc + wn * TN, lnc + BN + wn * TN);
            else gemm_epilogue<EPI, MI, NI, LN_CONS>(g, acc, m0 + wm * TM, n0 + wn * TN, half, l31, lnst + wm * TM, lnc + wn * TN, lnc + BN + wn * TN);
        } else {
            gemm_epilogue<EPI, MI, NI>(g, acc, m0 + wm * TM, n0 + wn * TN, half, l31);
        }
    }
}

template <int BM, int BN, int BK, int WM, int WN, int NS, int EPI, int FP8 = 0, int KG = 1, bool DIL = false>
int launch_pipe(const GemmArgs& a, hipStream_t stream) {
    constexpr int NT = KG * WM * WN * 64;
    constexpr bool LN_CONS = (EPI == EPI_SWIGLU || EPI == EPI_HEADS) && FP8 == 0;
    constexpr int LDS = NS * KG * ((BM + BN) * BK * 2 + (FP8 == 3 ? BM * 4 : 0)) + (LN_CONS ? (BM + BN) * 8 : 0);     // + (mean, rstd) per row, (c1, c2) per column
    static_assert(LDS <= 160 * 1024, "LDS ring exceeds 160 KiB");
    SAT_CHECK_ARG(LN_CONS || !a.ln_part, SAT_E_UNSUPPORTED, "gemm: LayerNorm fold needs bf16 operands and a SwiGLU / heads epilogue");
    SAT_CHECK_ARG((!a.xb && !a.ln_part_out) || (EPI == EPI_F32 && BN / WN == 64 && FP8 == 0 && a.xb && a.ln_part_out), SAT_E_UNSUPPORTED,
                  "gemm: the bf16 image / row statistics come from the bf16 fp32-output tiles with 64-column wave tiles");
    static_assert(EPI != EPI_F32 || BN / WN != 64 || LDS >= WM * WN * 8192, "the staged fp32 epilogue needs 8 KiB of LDS per wave");
    auto kern = gemm_pipe_kernel<BM, BN, BK, WM, WN, NS, EPI, FP8, KG, DIL>;
    // fused cross-attention (HeadsEpi::xa_k): three K / V^T tiles of 64 keys behind the ring and the LayerNorm constants
    constexpr bool XA_OK = EPI == EPI_HEADS && BM == 128 && BN == 64 && BK == 64 && WM == 4 && WN == 1 && NS == 3 && FP8 == 0;
    constexpr int XA_LDS = XA_OK ? ((NS * (BM + BN) * BK * 2 + (BM + BN) * 8 + 1023) & ~1023) + 3 * 16384 : LDS;
    static_assert(XA_LDS <= 160 * 1024, "fused cross-attention: K / V^T tiles do not fit behind the ring");
    bool xa = false;
    if constexpr (EPI == EPI_HEADS) {
        xa = a.heads.xa_k != nullptr;
        if (xa) {
            const HeadsEpi& he = a.heads;
            SAT_CHECK_ARG(XA_OK, SAT_E_UNSUPPORTED, "gemm: the fused cross-attention epilogue lives in the 128 x 64 tile (variant 16)");
            SAT_CHECK_ARG(he.parts == 1 && he.kind[0] == 8 && he.qscale == SAT_ATTN_QSCALE && he.xa_vt && he.xa_out && he.xa_kvh > 0 &&
                              he.heads % he.xa_kvh == 0 && he.xa_sk > 0 && he.xa_sk + 3 <= 192 && he.xa_sk_pad >= he.xa_sk + 3 && he.xa_sk_pad % 64 == 0,
                          SAT_E_UNSUPPORTED, "gemm: fused cross-attention needs one pre-scaled row-major part and at most 189 keys (got %d)", he.xa_sk);
        }
    }
    SAT_TRY(sat_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), XA_LDS));
    GemmArgs b = a;
    if (FP8) {
        SAT_CHECK_ARG(a.K % 128 == 0 && a.w_scale && (FP8 == 3 ? (const void*)a.a_bscale : (const void*)a.a_scale), SAT_E_UNSUPPORTED,
                      "gemm(fp8): K=%d must be a multiple of 128 and both scale vectors given", a.K);
        b.K = a.K / 2;       // the kernel counts 16-bit columns: 128-byte LDS rows = 128 e4m3
    }
    SAT_CHECK_ARG(b.N % BN == 0, SAT_E_UNSUPPORTED, "gemm: N=%d not a multiple of the %d-column tile", b.N, BN);
    SAT_CHECK_ARG(b.K % (BK * KG) == 0 && b.K / (BK * KG) >= NS, SAT_E_UNSUPPORTED, "gemm: K=%d too small for the %d-stage pipeline", a.K, NS);
    int tiles = cdiv(b.M, BM) * (b.N / BN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(NT), xa ? XA_LDS : LDS, stream, b);
    SAT_LAUNCH_CHECK();
    return 0;
}

template <int BM, int BN, int WM, int WN, int EPI, bool GLDS = false>
int launch_cfg(const GemmArgs& a, hipStream_t stream) {
    constexpr int NT = WM * WN * 64;
    constexpr int LDS = 2 * (BM + BN) * 128;
    SAT_CHECK_ARG(!a.ln_part && !a.xb && !a.ln_part_out, SAT_E_UNSUPPORTED, "gemm: LayerNorm fold is built into the pipelined tiles only (K >= 192)");
    auto kern = GLDS ? gemm_glds_kernel<BM, BN, WM, WN, EPI> : gemm_kernel<BM, BN, WM, WN, EPI>;
    SAT_TRY(sat_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS));
    SAT_CHECK_ARG(a.N % BN == 0, SAT_E_UNSUPPORTED, "gemm: N=%d not a multiple of the %d-column tile", a.N, BN);
    int tiles = cdiv(a.M, BM) * (a.N / BN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(NT), LDS, stream, a);
    SAT_LAUNCH_CHECK();
    return 0;
}

// Shipped tile configurations (variant ids as in round 1; everything else was an experiment and lives behind
// -DSAT_GEMM_EXPERIMENTS, see profiles/r01_gemm_variants*.txt for what they measured):
//    1  128x128, 4 waves, register-staged double buffer   (reference tile; tiny per-generation GEMMs: cross-attention to_kv)
//    5  128x128, 4 waves, LDS-DMA double buffer            (K < 192: too short for a 3-stage ring)
//   15  128x128x64, 8 waves, 3-stage LDS-DMA ring          (to_out at 1 prompt)
//   44  the same with a 4-stage ring, fp32 output only     (FF-out at 1 prompt: K = 6144)
//   16  128x64x64,  4 waves, 3-stage ring                  (cross-attention projections, M = 1025)
//   22  256x256x64, 16 waves, 2-stage ring                 (FF-in at 1 prompt; every GEMM from 4 prompts on.  The 4-stage BK = 32
//       variant with cross-tile fragment prefetch and grouped raster of round 1 measured within 2 % of it at 8 prompts after the
//       epilogue rewrite -- profiles/r02_b8_tiles.txt -- and was removed)
//   30  256x192x64, 12 waves, 2-stage ring                 (to_qkv at 1 prompt)
// fp8 (e4m3) operands: 15 / 16 / 22 / 30 in three flavours (plain fp8 MFMA, 2x-rate block-scaled MFMA, MXFP8 A operand).
// experiments build: SAT_GEMM_NO_DEEP=1 keeps the 3-stage ring for FF-out (A/B measurements)
inline bool deep_ring_off() {
#ifdef SAT_GEMM_EXPERIMENTS
    static const bool off = [] {
        const char* e = getenv("SAT_GEMM_NO_DEEP");
        return e && e[0] == '1';
    }();
    return off;
#else
    return false;
#endif
}

template <int EPI>
int launch_epi(const GemmArgs& a, hipStream_t stream) {
    int v = a.variant & 0xff;
    if constexpr (EPI == EPI_HEADS) {
        if (a.heads.xa_k) {       // fused cross-attention: built into the 128 x 64 tile only (the caller asks for it where that tile is the choice)
            SAT_CHECK_ARG(!a.fp8 && a.K >= 192, SAT_E_UNSUPPORTED, "gemm: fused cross-attention needs bf16 operands and K >= 192");
            return launch_pipe<128, 64, 64, 4, 1, 3, EPI>(a, stream);
        }
    }
    // fill of the last round of the device's CUs (256 on MI355X) x measured in-kernel rate of the tile, relative to the 256x256 tile
    const long cus = std::max(1, sat_device_cus());
    auto score = [&](int bm, int bn, double rate) {
        if (a.N % bn) return 0.0;
        long t = (long)cdiv(a.M, bm) * (a.N / bn);
        return rate * (double)t / (double)(((t + cus - 1) / cus) * cus);
    };
#ifndef SAT_OPERAND_F16          // e4m3 operands ride in the bf16 build (sat_launch_gemm rejects f16 && fp8): the fp16 build does not instantiate them
    if (a.fp8) {
        if (v == 0) {
            const double s256 = score(256, 256, 1.0), s192 = score(256, 192, 0.95), s128 = score(128, 128, 0.7), s64 = score(128, 64, 0.6);
            const double best = s256 > s192 ? (s256 > s128 ? s256 : s128) : (s192 > s128 ? s192 : s128);
            v = (s64 > best) ? 16 : (best == s256) ? 22 : (best == s192) ? 30 : 15;
            if (a.K < 384 && (v == 15 || v == 16)) v = 22;     // the 3-stage tiles need K >= 384 bytes
        }
        if (a.fp8 == 3) {      // MXFP8 A operand (hardware block scales), fp32 output only: FF-out, to_out
            if constexpr (EPI == EPI_F32) {
                switch (v) {
                    case 15: return launch_pipe<128, 128, 64, 4, 2, 3, EPI, 3>(a, stream);
                    case 16: return launch_pipe<128, 64, 64, 4, 1, 3, EPI, 3>(a, stream);
                    case 22: return launch_pipe<256, 256, 64, 4, 4, 2, EPI, 3>(a, stream);
                    case 30: return launch_pipe<256, 192, 64, 4, 3, 2, EPI, 3>(a, stream);
                }
            }
        } else if (a.fp8 == 2) {      // block-scaled MFMA with unit scales: 2x the MFMA rate
            const int fv = a.variant & 0xff;
            if ((fv == 80 || (fv == 0 && v == 22 && sat_wide_tile_of(a.variant) >= 80)) && sat_gemm_ph8_supports(EPI, a)) return sat_launch_gemm_ph8(EPI, a, stream);
            switch (v) {
                case 15: return launch_pipe<128, 128, 64, 4, 2, 3, EPI, 2>(a, stream);
                case 16: return launch_pipe<128, 64, 64, 4, 1, 3, EPI, 2>(a, stream);
                case 22: return launch_pipe<256, 256, 64, 4, 4, 2, EPI, 2>(a, stream);
                case 30: return launch_pipe<256, 192, 64, 4, 3, 2, EPI, 2>(a, stream);
            }
        } else {
            switch (v) {
                case 15: return launch_pipe<128, 128, 64, 4, 2, 3, EPI, 1>(a, stream);
                case 16: return launch_pipe<128, 64, 64, 4, 1, 3, EPI, 1>(a, stream);
                case 22: return launch_pipe<256, 256, 64, 4, 4, 2, EPI, 1>(a, stream);
                case 30: return launch_pipe<256, 192, 64, 4, 3, 2, EPI, 1>(a, stream);
            }
        }
        sat_set_error("gemm(fp8): variant %d has no e4m3 build (15, 16, 22, 30)", v);
        return SAT_E_INVALID;
    }
#endif
    if (v == 0) {
        // At 1 prompt (M = 2050) this gives FF-in 256x256 (432 workgroups, 2 rounds), to_qkv 256x192 (216 instead of 162
        // workgroups), to_out / FF-out 128x128 (204) and the cross-attention projections (M = 1025) 128x64 (216); from 4 prompts on
        // everything takes the 256x256 tile.
        if (a.K >= 192) {
            // The 256 x 256 tile is the 8-phase kernel where it applies; its rate relative to the 16-wave tile, measured at 8 prompts
            // (profiles/r03_ph8_streamk.txt): SwiGLU 1.26, heads 1.07, fp32 output with a long reduction 1.02 -- and with the K-split of the
            // remainder round (sat_gemm_ph8_splits) the last round costs ~0.35 of a round instead of 1.
            double s256 = score(256, 256, 1.0);
            if (sat_wide_tile_of(a.variant) >= 80 && sat_gemm_ph8_supports(EPI, a)) {
                const double rate = EPI == EPI_SWIGLU ? 1.26 : EPI == EPI_HEADS ? 1.07 : 1.02;
                const long t = (long)cdiv(a.M, 256) * (a.N / 256);
                const double rounds = sat_gemm_ph8_splits(EPI, a) ? (double)(t / cus) + 0.35 : (double)((t + cus - 1) / cus);
                s256 = rate * (double)t / (rounds * (double)cus);
            }
            const double s192 = score(256, 192, 0.95), s128 = score(128, 128, 0.7), s64 = score(128, 64, 0.6);
            const double best = s256 > s192 ? (s256 > s128 ? s256 : s128) : (s192 > s128 ? s192 : s128);
            if (best == 0.0 && s64 == 0.0) v = 15;      // N is not a tile multiple: let the launcher report it
            else if (s64 > best) v = 16;
            else if (best == s256) v = 22;
            else if (best == s192) v = 30;
            else v = 15;
            // long reductions (FF-out: 96 K-tiles) gain 4 % from a fourth ring stage (prefetch distance 3); K = 1536 does not care
            if (v == 15 && EPI == EPI_F32 && a.K >= 4096 && !deep_ring_off()) v = 44;
            // one round of 128 x 128 tiles (to_out / FF-out at one prompt: 204 workgroups on 256 CUs): the two-K-group build puts 8 waves of
            // 64 x 64 on every CU instead of 8 waves of 32 x 64 -- FF-out 56.5 us against 60.7, to_out 20.6 against 21.4 (tools/ph8_probe.py narrow)
            if ((v == 15 || v == 44) && EPI == EPI_F32 && !a.fp8 && a.K % 128 == 0 && a.K >= 256 && (long)cdiv(a.M, 128) * (a.N / 128) <= cus &&
                !(a.variant & 0x800000) && sat_wide_tile_of(a.variant) != 82)
                v = 49;
        } else {
            v = 5;
        }
    }
    // the 256x256 tile is the 8-wave / 8-phase kernel of gemm_ph8.hip wherever it applies (bf16 operands, K % 128 == 0);
    // tile policy 22 (sat_dit_cfg.tile_policy) brings the 16-wave 2-stage tile back for A/B measurements
    if (v == 22 && !(a.variant & 0xff) && sat_wide_tile_of(a.variant) >= 80 && sat_gemm_ph8_supports(EPI, a)) return sat_launch_gemm_ph8(EPI, a, stream);
    switch (v) {
        case 1: return launch_cfg<128, 128, 2, 2, EPI>(a, stream);
        case 5: return launch_cfg<128, 128, 2, 2, EPI, true>(a, stream);
        case 15: return launch_pipe<128, 128, 64, 4, 2, 3, EPI>(a, stream);
        case 44:
            if constexpr (EPI == EPI_F32) return launch_pipe<128, 128, 64, 4, 2, 4, EPI>(a, stream);
            break;
        case 16: return launch_pipe<128, 64, 64, 4, 1, 3, EPI, 0, 1, true>(a, stream);      // four waves = one per SIMD: LDS-DMA pieces in the MFMA stream (15.1 vs 16.2 us, cross to_out)
        case 49:           // 128 x 128 on two K-groups of 2 x 2 waves (64 x 64 each), 2 x 64 k per stage, 2 stages
            if constexpr (EPI == EPI_F32) return launch_pipe<128, 128, 64, 2, 2, 2, EPI, 0, 2>(a, stream);
            break;
        case 22: return launch_pipe<256, 256, 64, 4, 4, 2, EPI>(a, stream);
        case 30: return launch_pipe<256, 192, 64, 4, 3, 2, EPI>(a, stream);
#ifdef SAT_GEMM_EXPERIMENTS
        case 55: return launch_pipe<128, 128, 64, 4, 2, 3, EPI, 0, 1, true>(a, stream);      // tiles 15 / 16 / 30 / 44 with the LDS-DMA pieces in the MFMA stream (A/B)
        case 56: return launch_pipe<128, 64, 64, 4, 1, 3, EPI>(a, stream);                   // tile 16 with its pieces in front of the loop body (A/B)
        case 60: return launch_pipe<256, 192, 64, 4, 3, 2, EPI, 0, 1, true>(a, stream);
        case 54:
            if constexpr (EPI == EPI_F32) return launch_pipe<128, 128, 64, 4, 2, 4, EPI, 0, 1, true>(a, stream);
            break;
        case 2: return launch_cfg<256, 128, 4, 2, EPI>(a, stream);
        case 3: return launch_cfg<256, 256, 2, 4, EPI>(a, stream);
        case 7: return launch_cfg<256, 256, 2, 4, EPI, true>(a, stream);
        case 10: return launch_pipe<128, 128, 64, 2, 2, 3, EPI>(a, stream);
        case 12: return launch_pipe<256, 128, 64, 4, 2, 3, EPI>(a, stream);
        case 13: return launch_pipe<256, 256, 32, 2, 4, 3, EPI>(a, stream);
        case 39: return launch_pipe<128, 128, 128, 4, 2, 2, EPI>(a, stream);       // 256-B rows: half the barriers per k
        case 48: return launch_pipe<128, 128, 128, 2, 2, 2, EPI>(a, stream);       // the same on 4 waves of 64x64 (the vendor library's pick for FF-out at one prompt)
        case 41: return launch_pipe<256, 128, 32, 4, 2, 3, EPI>(a, stream);        // 72 KiB, <= 128 VGPRs: two workgroups per CU
        case 45:                                                                    //                               distance 4 (160 KiB)
            if constexpr (EPI == EPI_F32) return launch_pipe<128, 128, 64, 4, 2, 5, EPI>(a, stream);
            break;
        case 46: return launch_pipe<128, 64, 64, 4, 1, 5, EPI>(a, stream);         // tile 16 with prefetch distance 4
        case 47: return launch_pipe<128, 64, 64, 4, 1, 6, EPI>(a, stream);         //                               distance 5 (144 KiB)
        case 42: return launch_pipe<128, 128, 64, 2, 2, 4, EPI>(a, stream);        // 4 waves of 64x64 (half the LDS reads per MFMA of tile 15), 4 stages
        case 43: return launch_pipe<128, 128, 64, 2, 2, 2, EPI>(a, stream);        // same, 2 stages = 64 KiB: two workgroups per CU
#endif
        default: break;
    }
    sat_set_error("gemm: unknown variant %d (or not built for this epilogue)", v);
    return SAT_E_INVALID;
}

}  // namespace

#ifdef SAT_OPERAND_F16
// entry of the fp16 build for the bf16 build's dispatcher (GemmArgs differs only in the pointer element type)
int sat_launch_gemm_f16(int epi, const void* gemm_args, hipStream_t stream) {
    return f16::sat_launch_gemm(epi, *static_cast<const f16::GemmArgs*>(gemm_args), stream);
}
#endif

int SAT_OPNS::sat_launch_gemm(int epi, const GemmArgs& a, hipStream_t stream) {
#ifndef SAT_OPERAND_F16
    if (a.f16) return sat_launch_gemm_f16(epi, &a, stream);
#else
    SAT_CHECK_ARG(a.f16, SAT_E_INVALID, "gemm: bf16 operands handed to the fp16 build");
#endif
    SAT_CHECK_ARG(a.A && a.W, SAT_E_INVALID, "gemm: null operand");
    SAT_CHECK_ARG(!a.f16 || !a.fp8, SAT_E_UNSUPPORTED, "gemm: e4m3 operands ride in the bf16 build (fp8 mode keeps bf16 for everything else)");
    SAT_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0, SAT_E_INVALID, "gemm: bad shape %d %d %d", a.M, a.N, a.K);
    SAT_CHECK_ARG(a.K % (a.fp8 ? 128 : 64) == 0, SAT_E_UNSUPPORTED, "gemm: K=%d must be a multiple of %d", a.K, a.fp8 ? 128 : 64);
    SAT_CHECK_ARG(a.N % 128 == 0, SAT_E_UNSUPPORTED, "gemm: N=%d must be a multiple of 128", a.N);
    // the epilogues move 16 bytes per lane
    SAT_CHECK_ARG((((uintptr_t)a.bias | (uintptr_t)a.C | (uintptr_t)a.H | (uintptr_t)a.gate | (uintptr_t)a.w_scale) & 15) == 0 && a.ldc % 4 == 0 &&
                      a.gate_ld % 4 == 0,
                  SAT_E_INVALID, "gemm: bias / output / gate / scale pointers must be 16-byte aligned and ldc a multiple of 4");
    SAT_CHECK_ARG(!a.ln_part || (a.ln_c1 && a.ln_c2 && a.ln_eps > 0.f && !a.bias && !a.fp8 && a.K % 64 == 0 &&
                                 (((uintptr_t)a.ln_part | (uintptr_t)a.ln_c1 | (uintptr_t)a.ln_c2) & 15) == 0),
                  SAT_E_INVALID, "gemm: LayerNorm fold needs ln_c1 / ln_c2 / ln_eps, no separate bias (it is part of ln_c2), 16-byte aligned vectors");
    SAT_CHECK_ARG((((uintptr_t)a.xb | (uintptr_t)a.ln_part_out) & 15) == 0, SAT_E_INVALID, "gemm: xb / ln_part_out must be 16-byte aligned");
    if ((a.variant & 0xfff) % 100 == 80 || (a.variant & 0xfff) % 100 == 81) return sat_launch_gemm_ph8(epi, a, stream);      // 256x256x64, 8 waves, 8-phase schedule (gemm_ph8.hip)
    switch (epi) {
        case EPI_F32:
        case EPI_RESID: return launch_epi<EPI_F32>(a, stream);
        case EPI_SWIGLU: return launch_epi<EPI_SWIGLU>(a, stream);
        case EPI_HEADS: return launch_epi<EPI_HEADS>(a, stream);
    }
    sat_set_error("gemm: unknown epilogue %d", epi);
    return SAT_E_INVALID;
}
