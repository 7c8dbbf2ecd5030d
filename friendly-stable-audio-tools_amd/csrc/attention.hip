// Flash-style attention core for the DiT: softmax(q k^T / sqrt(64)) v,
// non-causal, no mask, GQA by head index (models/transformer.py:496-536; the reference's
// CPU branch is einsum + softmax(fp32) + einsum, :525-536; GQA repeat_interleave :512-515).
//
// gfx950 design.  One workgroup = 8 waves = 128 queries of one (batch, head) x 2 key ranges: waves 0-3 ("group 0") walk the first
// half of the KV tiles, waves 4-7 the second half, for the SAME 4 x 32 queries; the two partial results (running max, row sum,
// un-normalised output) are merged once through LDS.  At S = 1025 this doubles the number of waves in flight (432 workgroups of
// 8 waves, two per CU = 4 waves per SIMD) and halves the length of every wave's dependent tile chain.
//
// K and V^T tiles of 64 keys are copied to LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write pass) into a
// 2-stage ring per group, one raw s_barrier per tile, the copy of tile i+1 in flight while tile i is consumed.  LDS rows are 128 B
// with the 16-byte chunk index XORed by (row >> 1) & 7 (conflict-free ds_read_b128 fragments); the swizzle is folded into the
// per-lane SOURCE address because LDS-DMA writes lane-linear.
//
// Both products are computed TRANSPOSED so that everything that belongs to one query lives in one lane (plus its lane+32 partner):
//     S^T[key, q] = K[key, :] . Q[q, :]          (A = K fragment from LDS, B = Q in registers)
//     O^T[d,  q] = V^T[d, key] . P^T[key, q]     (A = V^T fragment from LDS, B = P in registers)
// With v_mfma_f32_32x32x16_bf16 the C/D layout is col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5): a lane's 16 S^T registers
// are 16 keys of ITS query, and they are exactly the k-operand elements the second MFMA wants if V^T stores every aligned group of
// 16 keys as [0-3, 8-11, 4-7, 12-15] (vt_pos, sat_common.h) -- which is how the QKV GEMM epilogue writes it.  P never moves between
// lanes; the row max needs one exchange between the wave halves per 32 keys, the row sum one at the end; the O rescale is lane-local.
// The online softmax advances in steps of 32 keys (one S^T accumulator block): 16 score registers live instead of 32, which is
// what lets the kernel run at 4 waves per SIMD (<= 128 VGPRs).
#include <stdlib.h>

#include "attn_core.h"
#include "sat_common.h"

namespace {

constexpr int KV_TILE = attn::KV_TILE;
constexpr int Q_BLOCK = 128;
constexpr int STAGE_BYTES = 2 * KV_TILE * 128;            // K tile + V^T tile
constexpr int GROUP_BYTES = 2 * STAGE_BYTES;              // 2-stage ring per KV group
constexpr int ATT_LDS = 2 * GROUP_BYTES;                  // 64 KiB: two workgroups per CU

using attn::half_max;
using attn::half_sum;
__device__ __forceinline__ void swap_halves(unsigned& lo_run, unsigned& hi_run) {
    u32x2 r = __builtin_amdgcn_permlane32_swap(lo_run, hi_run, false, false);
    lo_run = r[0];
    hi_run = r[1];
}

// MX8: fp8_gemm mode -- the output is the A operand of the to_out GEMM and is written as MXFP8 (e4m3 bytes at `out`, one E8M0
// scale per 32 channels = half a head at `out_scales` [B*Sq][H*2]) instead of bf16
// DBG (experiments build only, wrong results): 1 no exp / max / sum (P = bf16(S)); 2 no LDS-DMA in the loop; 3 no MFMA; 4 = 2 + no
// barrier; 5 = 4 + K / V^T fragments read from LDS once
// NGRP: 2 = the layout above (128 queries x 2 key ranges); 1 = 256 queries per workgroup, every wave walks ALL the KV tiles of one shared
// ring (long sequences / many sequences: half the LDS-DMA and K / V^T traffic per query, no merge; chosen by the launcher when the grid
// still fills the chip)
// MODE: the softmax recurrence of attn_core.h -- 1 (default): standing reference, the row sum is the overflow check; 2: the same with
// the reference carried through the matrix pipe, for a Q its producer wrote pre-scaled (scale_log2 == 1; anything else is multiplied
// into Q here and rounded to bf16 a second time: experiments); 0: the textbook recurrence (experiments build).
template <bool MX8, int DBG = 0, int NGRP = 2, int MODE = 1>
__global__ __launch_bounds__(512, 2) void attention_kernel(const op_t* __restrict__ q, const op_t* __restrict__ k,
                                                           const op_t* __restrict__ vt, op_t* __restrict__ out,
                                                           unsigned char* __restrict__ out_scales,
                                                           int H, int KVH, int Sq, int Sk, int Sq_pad, int Sk_pad,
                                                           float scale_log2) {
    sat_f16_saturate();
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int WPG = 8 / NGRP;         // waves per KV group
    constexpr int QB = 32 * WPG;          // queries per workgroup
    const int grp = NGRP == 2 ? wave >> 2 : 0;            // KV group
    const int wq = NGRP == 2 ? (wave & 3) : wave;         // query sub-block of this wave
    const int half = lane >> 5;
    const int l31 = lane & 31;
    // XCD-aware placement: the hardware deals consecutive workgroup ids round-robin to the 8 XCDs, so the query blocks of one (batch, head)
    // -- which all stream the SAME K / V^T -- would each pull their own copy through a different L2.  The linear id is remapped so that
    // consecutive LOGICAL ids share an XCD; x (query block) is the fastest logical coordinate.
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int lid = xcd_remap(lin, gridDim.x * gridDim.y * gridDim.z);
    const int bx = lid % gridDim.x;
    const int h = (lid / gridDim.x) % gridDim.y, b = lid / (gridDim.x * gridDim.y);
    const int kvh = h / (H / KVH);
    const int qi = bx * QB + wq * 32 + l31;                // NGRP == 2: < Sq_pad; NGRP == 1: the last workgroup may reach beyond it

    // Q fragments (B operand of S^T): Q[qi][16t + 8*half .. +8]
    opx8 qf[4];
    {
        const op_t* qp = q + ((size_t)(b * H + h) * Sq_pad + (qi < Sq_pad ? qi : Sq_pad - 1)) * 64 + half * 8;
#pragma unroll
        for (int t = 0; t < 4; ++t) qf[t] = *reinterpret_cast<const opx8*>(qp + t * 16);
    }

    if constexpr (MODE == 2) {
        if (scale_log2 != 1.0f) {          // (experiments: a plain Q, rounded a second time)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[t][j] = f32_to_op(op_to_f32(qf[t][j]) * scale_log2);
        }
    }

    // K rows / V^T columns of sequence b start at ob = (b*Sk) & 3 (see EPI_HEADS in gemm_bf16.hip)
    const int ob = (b * Sk) & 3;
    const int k_end = ob + Sk;
    const int n_tiles = (k_end + KV_TILE - 1) / KV_TILE;
    const int n0 = NGRP == 2 ? (n_tiles + 1) >> 1 : n_tiles;   // group 0: tiles [0, n0), group 1: [n0, n_tiles)
    const int t_first = grp ? n0 : 0;
    const int t_count = grp ? n_tiles - n0 : n0;
    const int n_iter = n0;                                 // >= the other group's count: both groups pass the same barriers

    // LDS-DMA pieces of this wave: 1 KiB = 8 rows of 128 B; wave wq of the group copies pieces wq and wq + 4 of the K tile and of the
    // V^T tile.  Lane l lands at row 8p + l/8, position l%8, so it fetches logical chunk (l%8) ^ ((row >> 1) & 7) of that row.
    const op_t* kbase = k + (size_t)(b * KVH + kvh) * Sk_pad * 64;
    const op_t* vbase = vt + (size_t)(b * KVH + kvh) * 64 * Sk_pad;
    constexpr int PPW = 8 / WPG;          // 1-KiB pieces of the K tile (and of the V^T tile) this wave copies
    const op_t* ksrc[PPW];
    const op_t* vsrc[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int row = (wq + WPG * i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        ksrc[i] = kbase + (size_t)row * 64 + c * 8;                 // + tile * 64 rows
        vsrc[i] = vbase + (size_t)row * Sk_pad + c * 8;             // + tile * 64 keys
    }
    char* ring = smem + grp * GROUP_BYTES;
    auto stage_in = [&](int tile, int stage) {
        char* sk = ring + stage * STAGE_BYTES;
        char* sv = sk + KV_TILE * 128;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ksrc[i] + (size_t)tile * KV_TILE * 64),
                                             (__attribute__((address_space(3))) void*)(sk + (wq + WPG * i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vsrc[i] + (size_t)tile * KV_TILE),
                                             (__attribute__((address_space(3))) void*)(sv + (wq + WPG * i) * 1024), 16, 0, 0);
        }
    };

    attn::State<MODE> st;
    st.init(half);
    f32x16 (&oacc)[2] = st.oacc;
    float& m_run = st.m_run;
    float& l_run = st.l_run;

    // a wave whose 32 queries are all beyond Sq (tail workgroup) still copies tiles and joins the barriers, but skips the matrix and
    // softmax work
    const bool wave_active = __builtin_amdgcn_readfirstlane(bx * QB + wq * 32) < Sq;

    auto process = [&](int tile, int stage) {
        const char* sk = ring + stage * STAGE_BYTES;
        const char* sv = sk + KV_TILE * 128;
        const bool edge = (tile == 0 && ob != 0) || (tile == n_tiles - 1 && (k_end & (KV_TILE - 1)) != 0);
        attn::tile<MODE, DBG>(st, qf, sk, sv, edge, tile * KV_TILE, ob, k_end, tile == t_first, scale_log2, l31, half);
    };

    if constexpr (NGRP == 1) {
        // three stages (48 KiB), prefetch distance 2: tile it + 1 stays in flight across the barrier (counted vmcnt: this wave issued
        // 2 * PPW pieces for it); tile it + 2 goes into the stage tile it - 1 occupied, which every wave left before this barrier.
        // (Measured, profiles/r03_attention_groups.txt: SA-2.0 609 -> 599 us against two stages; ONE barrier per pair of tiles with a
        // 2 x 2-tile ring instead: 619 us -- the distance-1 prefetch costs more than the saved barriers.)
        stage_in(t_first, 0);
        if (t_count > 1) stage_in(t_first + 1, 1);
        int st = 0;
        [[maybe_unused]] unsigned long long t_sync = 0, t_work = 0, t_a = 0, t_b = 0;
        for (int it = 0; it < n_iter; ++it) {
            if constexpr (DBG == 9) t_a = __builtin_readcyclecounter();
            if (it + 1 < t_count) wait_vmcnt<2 * PPW>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if constexpr (DBG == 9) { t_b = __builtin_readcyclecounter(); t_sync += t_b - t_a; }
            const int st2 = st >= 1 ? st - 1 : 2;                 // (it + 2) % 3
            if (it + 2 < t_count) stage_in(t_first + it + 2, st2);
            if (wave_active) process(t_first + it, st);
            if constexpr (DBG == 9) t_work += __builtin_readcyclecounter() - t_b;
            st = st == 2 ? 0 : st + 1;
        }
        if constexpr (DBG == 9) {          // experiments build: per-wave cycles parked at the tile barrier vs issuing the tile's work
            if (lane == 0 && wave_active) {
                unsigned long long* dbg = reinterpret_cast<unsigned long long*>(out_scales);
                atomicAdd(dbg, t_sync);
                atomicAdd(dbg + 1, t_work);
                atomicAdd(dbg + 2, 1ull);
                atomicAdd(dbg + 3, (unsigned long long)n_iter);
            }
        }
    } else {
    if (t_count > 0) stage_in(t_first, 0);
    for (int it = 0; it < n_iter; ++it) {
        if constexpr (DBG < 4) {
            wait_vmcnt<0>();                          // this wave's pieces of tile `it` have landed ...
            __builtin_amdgcn_s_barrier();             // ... everybody's have, and everybody is done with the other stage (tile it - 1)
        }
        if (DBG != 2 && DBG < 4 && it + 1 < t_count) stage_in(t_first + it + 1, (it + 1) & 1);
        if (wave_active && it < t_count) process(t_first + it, it & 1);
    }
    }

    // ---- merge the two key ranges: group 1 hands (m, l, O) to group 0 through LDS ([wave][34 values][64 lanes], conflict-free)
    if constexpr (NGRP == 2) {
    __builtin_amdgcn_s_barrier();                 // the rings are free
    float* mg = reinterpret_cast<float*>(smem) + wq * (34 * 64) + lane;
    if (grp == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) mg[(i * 16 + r) * 64] = oacc[i][r];
        mg[32 * 64] = m_run;
        mg[33 * 64] = l_run;
    }
    __syncthreads();
    if (grp == 1) return;
    {
        const float m1 = mg[32 * 64], l1 = mg[33 * 64];
        const float m = fmaxf(m_run, m1);
        const float a0 = __builtin_amdgcn_exp2f(m_run - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
        l_run = l_run * a0 + l1 * a1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[i][r] = oacc[i][r] * a0 + mg[(i * 16 + r) * 64] * a1;
    }
    }

    const float inv = 1.0f / half_sum(l_run);
    // O^T registers of a lane: channels d = db*32 + 8*(r>>2) + 4*half + (r&3) of ITS query (the layout of gemm_epilogue_t)
    if constexpr (MX8) {
        // block db (32 channels of this head): this lane holds 16 of them, lane ^ 32 the other 16
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            float am = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) am = fmaxf(am, fabsf(oacc[db][r] * inv));
            am = half_max(am);
            const float t = am * (1.0f / 448.0f);
            const unsigned tb = __float_as_uint(t);
            int e = (int)((tb >> 23) & 0xff) - 127 + ((tb & 0x7fffff) ? 1 : 0);
            e = am > 0.f ? (e < -127 ? -127 : (e > 127 ? 127 : e)) : -127;
            const float qs = inv * __uint_as_float((unsigned)(127 - e) << 23);
            unsigned q8[4];
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                unsigned p = __builtin_amdgcn_cvt_pk_fp8_f32(oacc[db][rq * 4] * qs, oacc[db][rq * 4 + 1] * qs, 0u, false);
                q8[rq] = __builtin_amdgcn_cvt_pk_fp8_f32(oacc[db][rq * 4 + 2] * qs, oacc[db][rq * 4 + 3] * qs, p, true);
            }
            swap_halves(q8[0], q8[1]);            // lanes 0-31: channels 0-7 | 16-23, lanes 32-63: 8-15 | 24-31
            swap_halves(q8[2], q8[3]);
            if (qi < Sq) {
                unsigned char* op = reinterpret_cast<unsigned char*>(out) + ((size_t)b * Sq + qi) * ((size_t)H * 64) + h * 64 + db * 32 + 8 * half;
                *reinterpret_cast<u32x2*>(op) = u32x2{q8[0], q8[1]};
                *reinterpret_cast<u32x2*>(op + 16) = u32x2{q8[2], q8[3]};
                if (half == 0) out_scales[((size_t)b * Sq + qi) * ((size_t)H * 2) + h * 2 + db] = (unsigned char)(e + 127);
            }
        }
    } else {
        op_t* op = out + ((size_t)b * Sq + qi) * ((size_t)H * 64) + h * 64 + 8 * half;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            unsigned pk[8];
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                opx2 lo, hi;
                lo[0] = f32_to_op(oacc[db][rq * 4] * inv);
                lo[1] = f32_to_op(oacc[db][rq * 4 + 1] * inv);
                hi[0] = f32_to_op(oacc[db][rq * 4 + 2] * inv);
                hi[1] = f32_to_op(oacc[db][rq * 4 + 3] * inv);
                pk[2 * rq] = __builtin_bit_cast(unsigned, lo);
                pk[2 * rq + 1] = __builtin_bit_cast(unsigned, hi);
            }
            swap_halves(pk[0], pk[2]);            // 8 consecutive channels per lane: one 16-byte store instead of two 8-byte ones
            swap_halves(pk[1], pk[3]);
            swap_halves(pk[4], pk[6]);
            swap_halves(pk[5], pk[7]);
            if (qi < Sq) {
                *reinterpret_cast<u32x4*>(op + db * 32) = u32x4{pk[0], pk[1], pk[2], pk[3]};
                *reinterpret_cast<u32x4*>(op + db * 32 + 16) = u32x4{pk[4], pk[5], pk[6], pk[7]};
            }
        }
    }
}

}  // namespace

#ifdef SAT_GEMM_EXPERIMENTS
static unsigned long long* g_attn_dbg = nullptr;
#endif
#if defined(SAT_GEMM_EXPERIMENTS) && !defined(SAT_OPERAND_F16)
extern "C" __attribute__((visibility("default"))) int sat_attention_dbg_read(unsigned long long* out4) {
    SAT_CHECK_ARG(g_attn_dbg, SAT_E_INVALID, "no attention counters");
    SAT_HIP(hipDeviceSynchronize());
    SAT_HIP(hipMemcpy(out4, g_attn_dbg, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    SAT_HIP(hipMemset(g_attn_dbg, 0, 4 * sizeof(unsigned long long)));
    return 0;
}
#endif

#ifdef SAT_OPERAND_F16
int sat_launch_attention_f16(const void* q, const void* k, const void* vt, void* out, int b, int h, int kvh, int sq, int sk, int sq_pad,
                             int sk_pad, hipStream_t s, unsigned char* out_scales, float q_scale) {
    return f16::sat_launch_attention((const op_t*)q, (const op_t*)k, (const op_t*)vt, (op_t*)out, b, h, kvh, sq, sk, sq_pad, sk_pad, s, out_scales,
                                     q_scale, 1);
}
#endif

int SAT_OPNS::sat_launch_attention(const op_t* q, const op_t* k, const op_t* vt, op_t* out, int b, int h, int kvh,
                                   int sq, int sk, int sq_pad, int sk_pad, hipStream_t s, unsigned char* out_scales, float q_scale, int f16) {
#ifndef SAT_OPERAND_F16
    if (f16) return sat_launch_attention_f16(q, k, vt, out, b, h, kvh, sq, sk, sq_pad, sk_pad, s, out_scales, q_scale);
#else
    SAT_CHECK_ARG(f16 && !out_scales, SAT_E_INVALID, "attention: the fp16 build takes fp16 tensors and writes fp16");
#endif
    SAT_CHECK_ARG(q && k && vt && out, SAT_E_INVALID, "attention: null pointer");
    SAT_CHECK_ARG(b > 0 && h > 0 && kvh > 0 && h % kvh == 0, SAT_E_INVALID, "attention: bad heads %d/%d", h, kvh);
    SAT_CHECK_ARG(sq > 0 && sk > 0 && sq_pad >= sq && sk_pad >= sk, SAT_E_INVALID, "attention: bad lengths");
    SAT_CHECK_ARG(sq_pad % Q_BLOCK == 0 && sk_pad % KV_TILE == 0, SAT_E_INVALID,
                  "attention: sq_pad %% 128 and sk_pad %% 64 must be 0 (got %d, %d)", sq_pad, sk_pad);
    SAT_CHECK_ARG(sk_pad >= sk + 3, SAT_E_INVALID, "attention: sk_pad must be >= sk + 3 (key-side shift), got %d for sk=%d", sk_pad, sk);
    SAT_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt | (uintptr_t)out) & 15) == 0, SAT_E_INVALID, "attention: pointers must be 16-byte aligned");
    const float scale_log2 = q_scale;       // SAT_ATTN_QSCALE = 1/sqrt(64) * log2(e), or 1 for a pre-scaled Q
    dim3 grid(cdiv(sq, Q_BLOCK), h, b);
    // One KV group (256 queries per workgroup, every wave walks all the KV tiles of one shared ring: half the LDS-DMA and K / V^T traffic per
    // query, no merge) or two (128 queries x 2 key ranges merged through LDS: twice the waves in flight, half the tile chain per wave)?
    // Round 3's rule -- one group only from 1024 workgroups on, or for <= 512 keys -- predates MODE 2 (the softmax reference carried through the
    // matrix pipe), which only the one-group layout can afford in registers.  Re-measured in round 5 for a pre-scaled Q
    // (profiles/r05_attention_layout_sweep.txt, S = 1025, 24 heads): 2 sequences (one prompt with CFG, 240 workgroups) 23.7 us against 25.9 with
    // two groups, 4 / 6 / 8 / 16 sequences 38.8 / 56.1 / 72.8 / 137 against 46.0 / 65.3 / 78.7 / 154; ONE sequence (120 workgroups for 256 CUs)
    // 21.5 against 15.8 -- there the split stays.  A plain Q (MODE 1 in both layouts) keeps round 3's rule.  SAT_ATTN_GROUPS = 1 | 2 forces a layout (A/B).
#ifdef SAT_GEMM_EXPERIMENTS
    const int force_grp = [] { const char* e = getenv("SAT_ATTN_GROUPS"); return e ? atoi(e) : 0; }();      // re-read per launch (A/B in one process)
#else
    static const int force_grp = [] { const char* e = getenv("SAT_ATTN_GROUPS"); return e ? atoi(e) : 0; }();
#endif
    const long wg1 = (long)cdiv(sq, 256) * h * b;
    // (the pre-scaled rule holds where it was measured: S = 1025; a lone long sequence -- S = 6145, 600 workgroups -- measured 2 % FASTER split,
    // profiles/r05_attention_layout_sweep.txt -- keeps round 3's rule: ADVICE r5)
    const bool one_group = force_grp ? force_grp == 1 : (wg1 >= 1024 || sk <= 512 || (q_scale == 1.0f && !out_scales && wg1 >= 200 && sk <= 2048));
#ifdef SAT_GEMM_EXPERIMENTS
    if (one_group && !out_scales && getenv("SAT_ATTN_DBG") && atoi(getenv("SAT_ATTN_DBG")) == 9) {
        if (!g_attn_dbg) {
            SAT_HIP(hipMalloc(&g_attn_dbg, 4 * sizeof(unsigned long long)));
            SAT_HIP(hipMemset(g_attn_dbg, 0, 4 * sizeof(unsigned long long)));
        }
        SAT_TRY(sat_ensure_dynamic_lds(reinterpret_cast<const void*>(attention_kernel<false, 9, 1>), 3 * STAGE_BYTES));
        hipLaunchKernelGGL((attention_kernel<false, 9, 1>), dim3(cdiv(sq, 256), h, b), dim3(512), 3 * STAGE_BYTES, s, q, k, vt, out,
                           reinterpret_cast<unsigned char*>(g_attn_dbg), h, kvh, sq, sk, sq_pad, sk_pad, scale_log2);
        SAT_LAUNCH_CHECK();
        return 0;
    }
#endif
#ifdef SAT_GEMM_EXPERIMENTS
    if (const char* eo = getenv("SAT_ATTN_MODE"); eo && !out_scales) {          // A/B of the softmax recurrences (tools/attn_opt_probe.py)
        auto launch = [&](auto kern, dim3 g, int lds) {
            (void)sat_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
            hipLaunchKernelGGL(kern, g, dim3(512), lds, s, q, k, vt, out, out_scales, h, kvh, sq, sk, sq_pad, sk_pad, scale_log2);
        };
        const dim3 g1(cdiv(sq, 256), h, b);
        switch (atoi(eo) + (one_group ? 100 : 0)) {
            case 0: launch(attention_kernel<false, 0, 2, 0>, grid, ATT_LDS); return 0;
            case 1: launch(attention_kernel<false, 0, 2, 1>, grid, ATT_LDS); return 0;
            case 2: launch(attention_kernel<false, 0, 2, 2>, grid, ATT_LDS); return 0;
            case 100: launch(attention_kernel<false, 0, 1, 0>, g1, 3 * STAGE_BYTES); return 0;
            case 101: launch(attention_kernel<false, 0, 1, 1>, g1, 3 * STAGE_BYTES); return 0;
            case 102: launch(attention_kernel<false, 0, 1, 2>, g1, 3 * STAGE_BYTES); return 0;
        }
    }
#endif
    if (one_group && !out_scales && q_scale == 1.0f) {      // pre-scaled Q: the reference rides in the matrix pipe (MODE 2)
        SAT_TRY(sat_ensure_dynamic_lds(reinterpret_cast<const void*>(attention_kernel<false, 0, 1, 2>), 3 * STAGE_BYTES));
        hipLaunchKernelGGL((attention_kernel<false, 0, 1, 2>), dim3(cdiv(sq, 256), h, b), dim3(512), 3 * STAGE_BYTES, s, q, k, vt, out, out_scales, h, kvh,
                           sq, sk, sq_pad, sk_pad, scale_log2);
        SAT_LAUNCH_CHECK();
        return 0;
    }
    if (one_group && !out_scales) {
        SAT_TRY(sat_ensure_dynamic_lds(reinterpret_cast<const void*>(attention_kernel<false, 0, 1>), 3 * STAGE_BYTES));
        hipLaunchKernelGGL((attention_kernel<false, 0, 1>), dim3(cdiv(sq, 256), h, b), dim3(512), 3 * STAGE_BYTES, s, q, k, vt, out, out_scales, h, kvh, sq, sk,
                           sq_pad, sk_pad, scale_log2);
        SAT_LAUNCH_CHECK();
        return 0;
    }
#ifdef SAT_GEMM_EXPERIMENTS
    if (const char* dbg = getenv("SAT_ATTN_DBG")) {
        auto launch = [&](auto kern) {
            (void)sat_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), ATT_LDS);
            hipLaunchKernelGGL(kern, grid, dim3(512), ATT_LDS, s, q, k, vt, out, out_scales, h, kvh, sq, sk, sq_pad, sk_pad, scale_log2);
        };
        switch (atoi(dbg)) {
            case 1: launch(attention_kernel<false, 1>); return 0;
            case 2: launch(attention_kernel<false, 2>); return 0;
            case 3: launch(attention_kernel<false, 3>); return 0;
            case 4: launch(attention_kernel<false, 4>); return 0;
            case 5: launch(attention_kernel<false, 5>); return 0;
        }
    }
#endif
    if (out_scales) {
        SAT_TRY(sat_ensure_dynamic_lds(reinterpret_cast<const void*>(attention_kernel<true>), ATT_LDS));
        hipLaunchKernelGGL(attention_kernel<true>, grid, dim3(512), ATT_LDS, s, q, k, vt, out, out_scales, h, kvh, sq, sk, sq_pad, sk_pad, scale_log2);
    } else {
        SAT_TRY(sat_ensure_dynamic_lds(reinterpret_cast<const void*>(attention_kernel<false>), ATT_LDS));
        hipLaunchKernelGGL(attention_kernel<false>, grid, dim3(512), ATT_LDS, s, q, k, vt, out, out_scales, h, kvh, sq, sk, sq_pad, sk_pad, scale_log2);
    }
    SAT_LAUNCH_CHECK();
    return 0;
}
