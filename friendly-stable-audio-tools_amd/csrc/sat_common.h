// Shared device/host helpers for libsat_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/sat_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------
// The 16-bit OPERAND type of the matrix kernels.  gemm_bf16.hip, gemm_ph8.hip, attention.hip (+ attn_core.h) and oobleck.hip are
// compiled TWICE (csrc/Makefile): once with op_t = bf16 and once with -DSAT_OPERAND_F16, op_t = IEEE fp16 -- gfx950 runs
// v_mfma_f32_*_f16 at exactly the rate of the bf16 instructions (MI355X_MICROARCH.md), fp16 carries three more significand bits
// (8x less operand rounding), and fp16 is what the reference itself computes in on a GPU (inference/sampling.py:210 autocast,
// models/transformer.py:496-504 flash-attn fp16, models/pretransforms.py:39-59 model_half).  Everything that names the element type
// in those files goes through op_t / opx8 / f32_to_op / mfma_*; the host-facing launchers of the two builds live in namespace
// bf16 / f16 (SAT_OPNS), the bf16 build owns the un-namespaced dispatchers and every extern "C" symbol.  Files compiled once
// (dit_plan.hip, layernorm.hip, ...) see op_t = bf16 and treat operand buffers as opaque 16-bit storage.
// fp16 range policy: conversions SATURATE at +-65504 (MODE.FP16_OVFL, set by sat_f16_saturate() at the top of every kernel of the
// fp16 build) instead of producing infinities; inf / NaN inputs still propagate.
#ifdef SAT_OPERAND_F16
typedef _Float16 op_t;
#define SAT_OPNS f16
#define SAT_OP_IS_F16 1
#else
typedef __bf16 op_t;
#define SAT_OPNS bf16
#define SAT_OP_IS_F16 0
#endif
typedef op_t opx8 __attribute__((ext_vector_type(8)));
typedef op_t opx4 __attribute__((ext_vector_type(4)));
typedef op_t opx2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

void sat_set_error(const char* fmt, ...);

#define SAT_CHECK_ARG(cond, code, ...)            \
    do {                                          \
        if (!(cond)) {                            \
            sat_set_error(__VA_ARGS__);           \
            return (code);                        \
        }                                         \
    } while (0)

#define SAT_HIP(call)                                                                       \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess) {                                                            \
            sat_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return (int)e__;                                                                \
        }                                                                                   \
    } while (0)

#define SAT_LAUNCH_CHECK()                                                                  \
    do {                                                                                    \
        hipError_t e__ = hipGetLastError();                                                 \
        if (e__ != hipSuccess) {                                                            \
            sat_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
            return (int)e__;                                                                \
        }                                                                                   \
    } while (0)

#define SAT_TRY(expr)            \
    do {                         \
        int rc__ = (expr);       \
        if (rc__ != 0) return rc__; \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: raise it once per (kernel, device), not once
// per process (a second GPU in the same process would otherwise launch with the 64 KiB default and fail).  Thread-safe.
int sat_ensure_dynamic_lds(const void* kernel, int bytes);
int sat_device_cus();          // compute units of the current device (cached per device; 0 + sat_last_error on failure)

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f32_to_bf16(float v) { return (bf16_t)v; }   // RNE
__device__ __forceinline__ float op_to_f32(op_t v) { return (float)v; }
__device__ __forceinline__ op_t f32_to_op(float v) { return (op_t)v; }          // RNE (fp16: saturating under sat_f16_saturate)

// fp16 build: overflowing float -> half conversions clamp to +-65504 (MODE.FP16_OVFL, bit 23 of the MODE register) instead of
// returning infinity.  One scalar instruction at kernel entry; a no-op in the bf16 build.
__device__ __forceinline__ void sat_f16_saturate_on() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }
__device__ __forceinline__ void sat_f16_saturate() {
#ifdef SAT_OPERAND_F16
    sat_f16_saturate_on();
#endif
}
// kernels templated on their 16-bit output type (layernorm.hip, dit_glue.hip: compiled once)
template <typename OT>
__device__ __forceinline__ void sat_saturate_for() {
    if constexpr (__is_same(OT, _Float16)) sat_f16_saturate_on();
}

// the two dense 16-bit MFMAs of gfx950 on the operand type of this build (same rate for bf16 and fp16)
__device__ __forceinline__ f32x16 mfma_32x32x16(opx8 a, opx8 b, f32x16 c) {
#ifdef SAT_OPERAND_F16
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ f32x4 mfma_16x16x32(opx8 a, opx8 b, f32x4 c) {
#ifdef SAT_OPERAND_F16
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// Wave-wide reductions without LDS traffic: four DPP steps reduce every row of 16 lanes (quad xor 1, quad xor 2, row_half_mirror,
// row_mirror), row_bcast:15 / row_bcast:31 fold the four rows into lane 63, v_readlane broadcasts the result through an SGPR.
// 7 VALU instructions instead of six ds_bpermute round trips (__shfl_xor).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move_masked(float v) {      // lanes outside ROW_MASK receive 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ float dpp_row_move(float v) {
    const int i = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(i, i, CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_row_move<0xB1>(v);                    // quad_perm [1,0,3,2]
    v += dpp_row_move<0x4E>(v);                    // quad_perm [2,3,0,1]
    v += dpp_row_move<0x141>(v);                   // row_half_mirror
    v += dpp_row_move<0x140>(v);                   // row_mirror: every lane of a row holds the row sum
    v += dpp_move_masked<0x142, 0xA>(v);           // row_bcast:15 into rows 1 and 3
    v += dpp_move_masked<0x143, 0xC>(v);           // row_bcast:31 into rows 2 and 3: lane 63 holds the total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {   // for non-negative inputs or any inputs: the masked moves use v itself as filler
    v = fmaxf(v, dpp_row_move<0xB1>(v));
    v = fmaxf(v, dpp_row_move<0x4E>(v));
    v = fmaxf(v, dpp_row_move<0x141>(v));
    v = fmaxf(v, dpp_row_move<0x140>(v));
    {
        const int i = __float_as_int(v);
        v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(i, i, 0x142, 0xA, 0xf, false)));
    }
    {
        const int i = __float_as_int(v);
        v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(i, i, 0x143, 0xC, 0xf, false)));
    }
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// max over the 32 lanes of a wave half (lanes 0-31 / 32-63), result in every lane: four DPP steps (quad xor 1, quad xor 2,
// row_half_mirror, row_mirror -> max of each row of 16) and one ds_swizzle that swaps the two rows of a half.  No address
// VGPRs, no ds_bpermute: 6 instructions against ~15 for five __shfl_xor steps.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    const int i = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(i, i, CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float half32_max(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v));       // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_move<0x4E>(v));       // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_move<0x141>(v));      // row_half_mirror
    v = fmaxf(v, dpp_move<0x140>(v));      // row_mirror
    return fmaxf(v, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F)));   // lane ^ 16
}

// sum over each aligned group of 16 lanes (one DPP row), result in all 16 lanes
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    return v;
}

// Accumulators of a 32x32 MFMA block issued with the operands swapped (weight fragment first) hold C^T: lane l31 owns one output ROW,
// its 16 registers are channels 8*(r>>2) + 4*half + (r&3) -- four runs of four.  half_swap exchanges one dword of runs g and g+1
// between the wave halves (v_permlane32_swap): afterwards lanes 0-31 hold (run g of half 0, run g of half 1) = 8 consecutive
// channels from 8g, lanes 32-63 (run g+1 of half 0, run g+1 of half 1) = 8 consecutive channels from 8(g+1).
__device__ __forceinline__ void half_swap(unsigned& lo_run, unsigned& hi_run) {
    u32x2 r = __builtin_amdgcn_permlane32_swap(lo_run, hi_run, false, false);
    lo_run = r[0];
    hi_run = r[1];
}
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    bf16x2 v;
    v[0] = f32_to_bf16(a);
    v[1] = f32_to_bf16(b);
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned pack_op2(float a, float b) {
    opx2 v;
    v[0] = f32_to_op(a);
    v[1] = f32_to_op(b);
    return __builtin_bit_cast(unsigned, v);
}
// the same exchange on the fp32 values of a whole block: afterwards v[0..7] are channels c0 .. c0+7 and v[8..15] channels
// c0+16 .. c0+23 of the lane's row, c0 = 8 * half (16-byte residual reads, 16-byte bf16 stores after packing)
__device__ __forceinline__ void gather_channel_runs(const f32x16& a, float (&v)[16]) {
#pragma unroll
    for (int g = 0; g < 4; g += 2)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned lo = __float_as_uint(a[4 * g + e]), hi = __float_as_uint(a[4 * (g + 1) + e]);
            half_swap(lo, hi);
            v[4 * g + e] = __uint_as_float(lo);
            v[4 * (g + 1) + e] = __uint_as_float(hi);
        }
}

// XCD-aware bijective remap of a linear workgroup id (guide T1): consecutive logical ids
// land on the same XCD (hardware places block b on XCD b % 8), so neighbouring tiles share
// one L2.  Speed only -- never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int NX = 8;
    int xcd = bid % NX, idx = bid / NX;
    int q = nwg / NX, r = nwg % NX;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// LDS tile layout used by the MFMA kernels: rows of 64 bf16 (128 B = eight 16-B chunks);
// chunk c of row r lives at chunk (c ^ ((r >> 1) & 7)).  With this XOR every 16-lane group
// of a ds_read_b128 fragment read (16 distinct rows, same logical chunk) covers all 16
// 16-B slots of the 256-B bank row: conflict-free; an 8-lane ds_write_b128 group (one row,
// 8 chunks) is conflict-free as well.
__device__ __forceinline__ int lds_tile_off(int row, int chunk) {
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// Position of key s inside V^T ([.., 64, Spad], key index contiguous): bits 2 and 3 of the index are swapped, i.e. every aligned
// group of 16 keys is stored as [0-3, 8-11, 4-7, 12-15].  That is the order in which the second attention MFMA consumes the
// probabilities a lane holds after the first one (attention.hip), so a V^T tile can be copied to LDS verbatim (LDS-DMA, 16-byte
// pieces) and read as fragments without any re-arrangement.  Involution; groups of 4 consecutive keys stay together.
__host__ __device__ __forceinline__ int vt_pos(int s) { return (s & ~12) | ((s & 4) << 1) | ((s & 8) >> 1); }

// counted wait on outstanding vector-memory operations (LDS-DMA pieces included); hipcc does not see inside the asm,
// which is the point: the compiler never drains the in-flight prefetch ring with a vmcnt(0)
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate is 6 bits");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---------------------------------------------------------------------------------------
// internal launchers shared between translation units
// ---------------------------------------------------------------------------------------
enum { EPI_F32 = 0, EPI_RESID = 1, EPI_SWIGLU = 2, EPI_HEADS = 3 };

namespace SAT_OPNS {

struct HeadsEpi {
    op_t* out[3];      // per part destination
    int kind[3];         // bit0: transposed ([B,Hh,64,Spad]) else [B,Hh,Spad,64]; bit1: apply RoPE;
                         // bit2: key-side tensor (K / V^T): rows/columns of sequence b shifted by (b*S)&3
                         // bit3: query tensor written PRE-SCALED by `qscale` (row-major destinations only): the attention
                         //       kernel then gets log2-domain scores straight out of its first MFMA (SAT_ATTN_QSCALE)
    float qscale;
    // Fused cross-attention (one part, row-major, pre-scaled, 128 x 64 tile = one head per workgroup column): the epilogue does not
    // store Q at all -- each wave keeps its 32 queries x 64 channels as MFMA fragments and runs softmax(q k^T) v against the
    // (batch, kv-head)'s keys / values, which the workgroup staged in LDS behind the GEMM ring at kernel start
    // (models/transformer.py:496-536 behind :430-437).  xa_k == nullptr: off.
    const op_t* xa_k;      // [B, kvh, sk_pad, 64], key-side layout of sat_attention_bf16
    const op_t* xa_vt;     // [B, kvh, 64, sk_pad]
    op_t* xa_out;          // [M, heads * 64]
    int xa_kvh, xa_sk, xa_sk_pad;
    int parts;           // N == parts * heads * 64
    int heads;           // heads per part
    int S;               // valid rows per sequence (row m -> b = m / S, s = m % S)
    int Spad;            // padded sequence length of the destination
    const float* rope_cos;   // [>=S][16]
    const float* rope_sin;
};

struct GemmArgs {
    int f16;             // operand format of A / W / H / xb / heads.*: 0 = bf16, 1 = IEEE fp16 (routes to the fp16 build of the kernels)
    const op_t* A;     // [M,K]
    const op_t* W;     // [N,K]
    const float* bias;   // [N] or nullptr
    int M, N, K;
    int variant;
    // EPI_F32 / EPI_RESID
    float* C;            // [M,ldc]
    int ldc;
    int accumulate;
    // fp8 (e4m3) operands: A and W hold one byte per element (row-major, K contiguous), the fp32 accumulator is multiplied by
    // a_scale[row] * w_scale[col] (per-token x per-output-channel scales) before the epilogue; K % 128 == 0
    int fp8;
    const float* a_scale;   // [M]
    const float* w_scale;   // [N]
    // fp8 == 3 (MXFP8 A operand): one E8M0 scale per 32 k of every A row, [M][K/32] bytes = [M][K/128] dwords, applied by the
    // block-scaled MFMA itself; a_scale is unused
    const unsigned* a_bscale;
    // EPI_SWIGLU with H8 != nullptr: the hidden activation is written as MXFP8 (e4m3 bytes [M, N/2] + E8M0 [M, N/64]) instead of bf16
    unsigned char* H8;
    unsigned char* Hs;
    const float* gate;   // adaLN: (acc + bias) * gate[(row / gate_rows) * gate_ld + col] before the residual add; or nullptr
    int gate_rows, gate_ld;
    // EPI_SWIGLU
    op_t* H;           // [M, N/2]
    // EPI_HEADS
    HeadsEpi heads;
    // ---- LayerNorm folded into the GEMMs either side of the residual stream (bf16 operands, pipelined tiles only).
    // LN(x) W^T = rstd * (x (gamma (.) W)^T - mean * rowsum(gamma (.) W)) + beta W^T: the GEMM that UPDATES the residual stream
    // (EPI_RESID, the "producer") also writes the bf16 image of the new rows and, per row and 64-column block, the sum and the
    // sum of squares of those rounded values; the GEMM that CONSUMES the normalised rows (EPI_SWIGLU / EPI_HEADS) takes the bf16
    // image as its A operand, gamma-scaled weights, and finishes the normalisation on its fp32 accumulators.
    op_t* xb;                // producer: [M, N] bf16(C after the update), or nullptr
    float* ln_part_out;        // producer: [M][N / 64][2]
    const float* ln_part;      // consumer: [M][K / 64][2] partial (sum, sum of squares) of A's rows, or nullptr (no fold)
    const float* ln_c1;        // consumer: [N] sum_k bf16(gamma_k W_nk), epilogue channel order
    const float* ln_c2;        // consumer: [N] sum_k beta_k W_nk (+ bias_n)
    float ln_eps;
    // ---- scratch of the 8-phase kernel's K-split (fp32 output, long reductions: FF-out at the SA-2.0 shape): raw accumulator images
    // of the remainder round's partial K-ranges, [workgroups][65536] floats, added up by a second launch.  Caller-owned (the plan's
    // workspace), used by this launch only; nullptr = the remainder round's tiles stay whole.
    float* slab;
    size_t slab_bytes;
};

// bf16 build: checks a.f16 and forwards fp16 work to sat_launch_gemm_f16 (the fp16 build of the same file)
int sat_launch_gemm(int epi, const GemmArgs& a, hipStream_t stream);
bool sat_gemm_ph8_supports(int epi, const GemmArgs& a);
bool sat_gemm_ph8_splits(int epi, const GemmArgs& a);       // the automatic schedule would split the remainder round along K
size_t sat_gemm_ph8_slab_bytes(int epi, int M, int N, int K);      // slab workspace that makes it do so (0: never for this shape)
int sat_launch_gemm_ph8(int epi, const GemmArgs& a, hipStream_t stream);     // gemm_ph8.hip: the 8-wave / 8-phase 256x256 tile
// out_scales != nullptr: MXFP8 output (e4m3 bytes at `out`, E8M0 per 32 channels at out_scales [b*sq][h*2]) instead of bf16
// q_scale: what the kernel still has to multiply in -- SAT_ATTN_QSCALE = 1/sqrt(64) * log2(e) for a plain Q, 1.0f for a Q the
// producer already wrote pre-scaled (HeadsEpi kind bit 3): only then the single-KV-group kernel carries its softmax reference
// through the matrix pipe (a plain Q would have to be rounded to bf16 a second time).
// f16 != 0: q / k / vt / out hold IEEE fp16 (bf16 build: forwards to sat_launch_attention_f16)
#define SAT_ATTN_QSCALE (0.125f * 1.4426950408889634f)
int sat_launch_attention(const op_t* q, const op_t* k, const op_t* vt, op_t* out, int b, int h, int kvh,
                         int sq, int sk, int sq_pad, int sk_pad, hipStream_t s, unsigned char* out_scales = nullptr,
                         float q_scale = SAT_ATTN_QSCALE, int f16 = 0);
}  // namespace SAT_OPNS
using namespace SAT_OPNS;
// entry points of the fp16 build for the dispatchers of the bf16 build (GemmArgs is layout-identical in both builds: the pointer
// element type is the only difference)
int sat_launch_gemm_f16(int epi, const void* gemm_args, hipStream_t stream);
int sat_launch_attention_f16(const void* q, const void* k, const void* vt, void* out, int b, int h, int kvh, int sq, int sk, int sq_pad,
                             int sk_pad, hipStream_t s, unsigned char* out_scales, float q_scale);
// Tile policy of a launch: bits 24-26 of GemmArgs::variant (sat_dit_cfg.tile_policy puts them there for every GEMM of a plan; the
// unit-level entry points leave them 0).  0 = the default (80); the others are A/B measurement switches:
//   22: the 16-wave 2-stage 256 x 256 tile of rounds 1-2 instead of the 8-phase kernel
//   81: the 8-phase kernel also for the fp32-output GEMMs with K < 4096        82: no two-K-group 128 x 128 tile
#define SAT_TILE_POLICY_SHIFT 24
static inline int sat_tile_policy_bits(int policy) { return (policy == 22 ? 1 : policy == 81 ? 2 : policy == 82 ? 3 : 0) << SAT_TILE_POLICY_SHIFT; }
static inline int sat_wide_tile_of(int variant) {
    const int p = (variant >> SAT_TILE_POLICY_SHIFT) & 7;
    return p == 1 ? 22 : p == 2 ? 81 : p == 3 ? 82 : 80;
}

// f16 (last argument of the launchers below): the 16-bit output is IEEE fp16 (saturating) instead of bf16
int sat_launch_layernorm(const float* x, const float* gamma, const float* beta, op_t* y, int m, int d, hipStream_t s, int f16 = 0);
// adaLN: y = LN(x) * scale1p[b] + shift[b] with b = row / rows_per_seq and per-sequence vectors ld apart (transformer.py:671-672)
int sat_launch_layernorm_mod(const float* x, const float* gamma, const float* beta, op_t* y, int m, int d, const float* scale1p,
                             const float* shift, int rows_per_seq, int ld, hipStream_t s, int f16 = 0);
// LayerNorm (+ optional adaLN modulation) quantised per row to fp8 e4m3: y8 = rne(r / s), s = amax(r) / 448 -> row_scale[m]
int sat_launch_layernorm_fp8(const float* x, const float* gamma, const float* beta, void* y8, float* row_scale, int m, int d,
                             const float* scale1p, const float* shift, int rows_per_seq, int ld, hipStream_t s);
// rows of an fp32 matrix -> fp8 e4m3 with one scale per row (weights: per output channel; optional SwiGLU interleave)
int sat_launch_quant_rows_fp8(const float* w, void* out8, float* row_scale, int n, int k, int swiglu_interleave, hipStream_t s);
int sat_launch_quant_mx_rows(const float* x, void* out8, void* scales_e8m0, int rows, int k, hipStream_t s);
// fp32 verification path (f32_ref.hip)
int sat_launch_gemm_f32(const float* A, const float* W, const float* bias, float* C, int M, int N, int K, int ldc, int accumulate,
                        const float* gate, int gate_rows, int gate_ld, hipStream_t s);
int sat_launch_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y, int m, int d, const float* sc,
                             const float* sh, int rps, int ld, hipStream_t s);
int sat_launch_split_heads_f32(const float* src, float* d0, float* d1, float* d2, int M, int S, int parts, int H, int rope_mask,
                               const float* rope_cos, const float* rope_sin, hipStream_t s);
int sat_launch_swiglu_f32(const float* hg, float* h, int64_t M, int inner, hipStream_t s);
int sat_launch_attention_f32(const float* q, const float* k, const float* v, float* out, int b, int h, int kvh, int sq, int sk,
                             hipStream_t s);
int sat_launch_cast_bf16(const float* x, op_t* y, int64_t n, hipStream_t s, int f16 = 0);
int sat_launch_pack_rows_bf16(const float* w, op_t* out, int n, int k, int swiglu_interleave, hipStream_t s, int f16 = 0);
int sat_launch_pack_bias(const float* b, float* out, int n, int swiglu_interleave, hipStream_t s);
// weights of a GEMM that absorbs the LayerNorm in front of it: out = bf16(gamma (.) w) (rows permuted like pack_rows),
// c1[n] = sum_k float(out[n][k]), c2[n] = sum_k beta[k] w[n][k] (+ bias[n])
int sat_launch_pack_rows_ln(const float* w, const float* gamma, const float* beta, const float* bias, op_t* out, float* c1, float* c2,
                            int n, int k, int swiglu_interleave, hipStream_t s, int f16 = 0);
int sat_launch_rope_table(const float* inv_freq, float* cos_t, float* sin_t, int s_len, hipStream_t s);
