// The per-wave attention step shared by the attention kernel (attention.hip) and the fused to_q + cross-attention epilogue of the
// 128 x 64 GEMM tile (gemm_bf16.hip): one wave = 32 queries (lane l and l ^ 32 share query l & 31), one call = one LDS tile of 64
// keys = two blocks of 32 keys.  Reference: models/transformer.py:496-536 (softmax(q k^T / sqrt(64)) v, fp32 softmax).
//
// Both products are computed transposed (see attention.hip): S^T = K Q^T with the K fragment from LDS as the A operand and Q in
// registers as B; O^T = V^T P^T with the V^T fragment from LDS as A and the lane's own 16 probabilities as B.
//
// Online softmax WITHOUT a per-block maximum.  Any per-query reference m_ref gives the same quotient sum(p v) / sum(p), p =
// 2^(s - m_ref), as long as nothing overflows; so the reference stands still and the row sum doubles as the overflow check (every p
// is <= its lane's sum).  Only when some lane's sum leaves [0, 2^12] -- always in the first block, later only for a score more than
// 12 octaves above the reference -- the block is redone the classic way (true maximum, accumulators rescaled).  The kernel is bound
// by VALU issue (profiles/r03_issue_rates.txt: 16 v_exp_f32 + 16 v_fma_f32 + 16 v_add_f32 + 8 v_cvt_pk_bf16_f32 per 32 keys against
// 8 MFMAs); this removes the 8 v_max3_f32, the exchange, the second exponential and -- for random scores in about every second
// tile -- the 32 accumulator multiplies of the textbook recurrence.
//   MODE 1: p = exp2(fma(s, scale_log2, -m_ref)).
//   MODE 2 (Q pre-scaled by scale_log2 by its producer): the reference rides in the matrix pipe -- a fifth K-step
//           [1, 0, ...] x [-m_ref, 0, ...] makes the accumulator come out as log2-domain score minus reference, p = exp2(acc): the 16
//           v_fma_f32 disappear from the VALU stream, the matrix pipe has the slack.  m_ref is kept representable in the operand type (bf16 / fp16).
//   MODE 0: the textbook recurrence (experiments build, A/B).
#pragma once
#include "sat_common.h"

namespace attn {

constexpr int KV_TILE = 64;

__device__ __forceinline__ float half_max(float v) {     // max over the lane pair (l, l ^ 32), in both lanes
    unsigned a = __float_as_uint(v), b = a;
    u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float v) {
    unsigned a = __float_as_uint(v), b = a;
    u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <int MODE>
struct State {
    f32x16 oacc[2];         // O^T: channels d = db*32 + 8*(r>>2) + 4*half + (r&3) of the lane's query
    float m_run;            // reference, log2-scaled units
    float l_run;            // this lane's partial row sum (its 16 of every 32 keys)
    opx8 ones_a, mref_b;  // MODE 2: the fifth K-step

    __device__ __forceinline__ void init(int half) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
        m_run = MODE == 2 ? 0.f : -1e30f;
        l_run = 0.f;
        if constexpr (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                ones_a[j] = f32_to_op(0.f);
                mref_b[j] = f32_to_op(0.f);
            }
            if (half == 0) ones_a[0] = f32_to_op(1.0f);
        }
    }
};

// One tile of 64 keys at LDS addresses sk (K tile: 64 rows of 128 B) / sv (V^T tile: 64 channel rows of 128 B), both in the XOR-swizzled
// layout of lds_tile_off.  edge: some keys of the tile lie outside [k_lo, k_hi) and are masked; key_base = the tile's first key.
// first: the sequence's first tile for this wave (MODE 2 fixes a real reference there whatever the sums say: underflow safety).
// DBG (experiments build, wrong results): 1 no exp / max / sum; 3 no MFMA; 5 fragments not read from LDS.
template <int MODE, int DBG = 0>
__device__ __forceinline__ void tile(State<MODE>& st, const opx8 (&qf)[4], const char* sk, const char* sv, const bool edge, const int key_base,
                                     const int k_lo, const int k_hi, const bool first, const float scale_log2, const int l31, const int half) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        // ---- S^T = K Q^T for 32 keys
        opx8 kf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if constexpr (DBG == 5) kf[t] = qf[t];
            else kf[t] = *reinterpret_cast<const opx8*>(sk + lds_tile_off(kb * 32 + l31, t * 2 + half));
        }
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = DBG == 3 ? (float)kf[r & 3][r >> 2] : 0.f;
        if constexpr (DBG != 3) {
            if constexpr (MODE == 2) sacc = mfma_32x32x16(st.ones_a, st.mref_b, sacc);
#pragma unroll
            for (int t = 0; t < 4; ++t) sacc = mfma_32x32x16(kf[t], qf[t], sacc);
        }
        // V^T fragments do not depend on the softmax: request them now, their LDS latency hides behind the VALU work
        opx8 vf[2][2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if constexpr (DBG == 5) vf[db][u] = qf[db * 2 + u];
                else vf[db][u] = *reinterpret_cast<const opx8*>(sv + lds_tile_off(db * 32 + l31, (kb * 2 + u) * 2 + half));
        // ---- mask the keys outside [k_lo, k_hi) (wave-uniform branch: first and last tile only)
        if (edge) {
            const int key0 = key_base + kb * 32 + 4 * half;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + (r & 3) + 8 * (r >> 2);
                if (key < k_lo || key >= k_hi) sacc[r] = -INFINITY;
            }
        }
        // ---- online softmax step (per query = per lane pair)
        opx8 pb[2];
        if constexpr (DBG == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pb[r >> 3][r & 7] = f32_to_op(sacc[r]);
            st.l_run += sacc[0];
        } else if constexpr (MODE == 0) {
            float mloc = sacc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, sacc[r]);
            mloc = half_max(mloc);
            const float m_new = fmaxf(st.m_run, mloc * scale_log2);
            const float alpha = __builtin_amdgcn_exp2f(st.m_run - m_new);
            st.m_run = m_new;
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(sacc[r], scale_log2, -m_new));
                psum += p;
                pb[r >> 3][r & 7] = f32_to_op(p);
            }
            st.l_run = st.l_run * alpha + psum;
            if (!__all(alpha == 1.0f)) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) st.oacc[i][r] *= alpha;
            }
        } else {
            // fast path: exponentials against the standing reference (MODE 2: already subtracted by the matrix pipe)
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = MODE == 2 ? __builtin_amdgcn_exp2f(sacc[r]) : __builtin_amdgcn_exp2f(fmaf(sacc[r], scale_log2, -st.m_run));
                psum += p;
                pb[r >> 3][r & 7] = f32_to_op(p);
            }
            const bool redo = (MODE == 2 && first && kb == 0) || !__all(psum <= 4096.0f);       // wave-uniform; NaN-safe (inf - inf cannot arise)
            if (redo) {
                float mloc = sacc[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, sacc[r]);
                float shift;              // what to add to the fast path's exponent
                if constexpr (MODE == 2) {
                    // the sequence's first block SETS the reference to the true maximum (the initial 0 is not a lower bound: a query whose
                    // log2-domain scores all lie below -126 would otherwise flush every p to 0 and divide by a zero row sum);
                    // later blocks only ever raise it
                    const float cand = half_max(mloc) + st.m_run;
                    const float m_new = op_to_f32(f32_to_op((first && kb == 0) ? cand : fmaxf(st.m_run, cand)));
                    shift = st.m_run - m_new;                      // <= 0 up to the operand-type rounding of m_new (first block: any sign)
                    st.m_run = m_new;
                    st.mref_b[0] = f32_to_op(half == 0 ? -m_new : 0.f);
                } else {
                    const float m_new = fmaxf(st.m_run, half_max(mloc) * scale_log2);
                    shift = st.m_run - m_new;
                    st.m_run = m_new;
                }
                // (MODE 2, first block: nothing is accumulated yet and the shift may have either sign -- 0 * 2^shift must not become 0 * inf)
                const float alpha = (MODE == 2 && first && kb == 0) ? 1.0f : __builtin_amdgcn_exp2f(shift);
                st.l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) st.oacc[i][r] *= alpha;
                psum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = MODE == 2 ? __builtin_amdgcn_exp2f(sacc[r] + shift) : __builtin_amdgcn_exp2f(fmaf(sacc[r], scale_log2, -st.m_run));
                    psum += p;
                    pb[r >> 3][r & 7] = f32_to_op(p);
                }
            }
            st.l_run += psum;
        }
        // ---- O^T += V^T P^T
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if constexpr (DBG == 3) st.oacc[db][u] += (float)vf[db][u][0] * (float)pb[u][0];
                else st.oacc[db] = mfma_32x32x16(vf[db][u], pb[u], st.oacc[db]);
            }
    }
}

}  // namespace attn
