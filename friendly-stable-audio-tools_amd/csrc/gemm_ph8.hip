// 256 x 256 x 64 bf16 MFMA GEMM in the 8-wave / 256-register / 8-phase regime (round 3):
//     C[M,N] = A[M,K] . W[N,K]^T      A, W bf16 (K contiguous), fp32 accumulate
// for the wide DiT projections (FF-in SwiGLU, to_qkv; every block GEMM from 4 prompts per GPU on) -- the nn.Linear calls at
// models/transformer.py:222,270,314,311-312,319 of the reference.  gfx950 only.
//
// Structure (one workgroup = 8 waves = 2 (M) x 4 (N), wave tile 128 x 64, v_mfma_f32_16x16x32_bf16, 128 accumulator registers):
//   * a K-tile (64 k) is FOUR 16-KiB half-tiles in LDS, in the order a wave consumes them:
//         kind 0  W-lo : the first 32 channels of every wave's 64       kind 1  A-lo : the first 64 rows of every wave's 128
//         kind 2  W-hi : the last 32 channels                           kind 3  A-hi : the last 64 rows
//     two K-tiles are resident (128 KiB ring); rows are 128 B with the 16-byte chunk XOR ((row >> 1) & 7), applied on the
//     SOURCE side of the LDS-DMA (buffer_load_dwordx4 ... lds writes lane-linear), so ds_read_b128 fragment reads are
//     conflict-free;
//   * a K-tile is four PHASES, one 64 x 32 quadrant of the wave tile (16 MFMAs) each:
//         phase 0: read W-lo (4) + A-lo (8) fragments   -> A-lo x W-lo        phase 2: read A-hi (8)  -> A-hi x W-hi
//         phase 1: read W-hi (4)                        -> A-lo x W-hi        phase 3: nothing        -> A-hi x W-lo
//     every phase is { ds_reads ; one half-tile of LDS-DMA (2 instructions per wave) ; s_barrier ; lgkmcnt(0) ; 16 MFMA ; s_barrier };
//   * the two wave rows run staggered by one barrier (wave row 1 passes one extra s_barrier up front), so on every SIMD one
//     wave is in its MFMA section while its partner reads fragments and issues DMA;
//   * half-tile h = 4 t + j is issued in phase h - 7 (seven half-tiles ahead); the only vector-memory wait of the loop is one
//     COUNTED s_waitcnt vmcnt(6) in phase 3 of every K-tile (three half-tiles stay in flight across the barriers), and the
//     K-tile it retires is first read one phase later.  Restaging is WAR-safe by construction: slot j of the current buffer is
//     rewritten in phase j + 1 -- W-lo after an lgkmcnt that retired its reads before phase 0's first barrier, the others two
//     phases after their last read.
// Epilogues run on TRANSPOSED accumulators (W fragment = MFMA A operand): lane l owns token row (l & 15) of a 16 x 16 block and
// four consecutive output channels; which channels a wave's W rows are is a free permutation applied where the DMA picks its
// source rows (chan_of), chosen per epilogue so that stores are 16 bytes and SwiGLU / RoPE partners are lane-local.
#include <type_traits>

#include "sat_common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_void_p;
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

// Channel (0..63 inside the wave's 64) held by W half-tile `ni` (0 lo / 1 hi), fragment nf (0/1), fragment row i (0..15).
// After the MFMA lane (i' = l & 15, q = l >> 4) holds rows 4q..4q+3 of the fragment, i.e. fragment rows i = 4q + r.
//   PERM 0 (fp32 output):  natural order ni*32 + nf*16 + i      -> 4 lanes q write 64 contiguous bytes of a row per store
//   PERM 1 (bf16 output):  ni*32 + q*8 + nf*4 + r                -> a lane holds 8 consecutive channels per ni: one 16-byte store;
//                          value (ni = 0) and gate (ni = 1) of a SwiGLU pair sit in the same lane
//   PERM 2 (heads):        ni = 0 natural (RoPE partner d + 16 = fragment nf + 1 of the same lane), ni = 1 as PERM 1
template <int PERM>
__device__ __forceinline__ int chan_of(int ni, int nf, int i) {
    const int q = i >> 2, r = i & 3;
    if constexpr (PERM == 0) return ni * 32 + nf * 16 + i;
    else if constexpr (PERM == 1) return ni * 32 + q * 8 + nf * 4 + r;
    else return ni == 0 ? nf * 16 + i : 32 + q * 8 + nf * 4 + r;
}

// DBG (tools/gpu_probe.py ablations, wrong results): 1 no LDS-DMA in the loop, 2 no ds_read (fragments stay), 3 no MFMA
// Measured and dropped (profiles/r03_ph8_schedule_options.txt): without the explicit lgkmcnt(0) behind the barrier, without s_setprio,
// with a static priority for waves 4-7 -- all within 1 %; LDS-DMA issued at the head of the MFMA section instead of the load
// section -- 5 % slower.
#ifdef SAT_GEMM_EXPERIMENTS
unsigned long long* g_ts_buf = nullptr;       // DBG 9: per workgroup (start, prologue done, main loop done, end) in 100 MHz ticks + HW id
#endif

template <int EPI, int DBG = 0>
__global__ __launch_bounds__(512) void gemm_ph8_kernel(GemmArgs g, unsigned long long* ts = nullptr) {
    [[maybe_unused]] unsigned long long t_start = 0, t_pro = 0, t_main = 0;
    if constexpr (DBG == 9) t_start = __builtin_amdgcn_s_memrealtime();
    constexpr int BUF_BYTES = 65536, HALF_BYTES = 16384, RING_BYTES = 131072;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, q4 = lane >> 4;

    const int M = g.M, N = g.N, K = g.K;
    const int tiles_m = (M + 255) >> 8;
    const int tiles_n = N >> 8;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    if (tiles_m <= 12) {               // short M: m fastest, the W panel of a column tile stays in one XCD's L2
        tn = bid / tiles_m;
        tm = bid - tn * tiles_m;
    } else {                           // long M: bands of 8 row tiles, n-major inside a band (8 A panels + the W panels in flight)
        const int band_sz = 8 * tiles_n;
        const int band = bid / band_sz;
        const int rem = bid - band * band_sz;
        const int gm = min(8, tiles_m - band * 8);
        tn = rem / gm;
        tm = band * 8 + (rem - tn * gm);
    }
    const int m0 = tm << 8, n0 = tn << 8;
    // Accumulator orientation, uniform over the workgroup (a 256-column tile never straddles a q / k / v part): transposed
    // (lane = token) everywhere except for a V^T destination, whose token-contiguous stores want lane = channel.
    bool tr = true;
    if constexpr (EPI == EPI_HEADS) tr = !(g.heads.kind[n0 / (g.heads.heads * 64)] & 1);
    // waves whose 128 rows lie entirely beyond M (the M-tail tile) keep staging and joining barriers, nothing else
    const bool rows_valid = (m0 + wr * 128) < M;

    // ---- LDS-DMA sources.  One instruction of one wave fills 8 LDS rows (1 KiB); round i of wave w covers rows 64 i + 8 w + (lane >> 3).
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)((unsigned)M * (unsigned)K * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(g.W + (size_t)n0 * K), 0, 256 * K * 2, 0x00020000);
    int voffA[2], voffW[2][2];          // [round] (A: lo; hi = + 64 rows), [ni][round]
    {
        const int sub = lane >> 3, pos = lane & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = i * 64 + wave * 8 + sub;                   // LDS row of the half-tile
            const int c = pos ^ ((r >> 1) & 7);
            // A: LDS rows [0,64) belong to wave row 0, [64,128) to wave row 1
            voffA[i] = (m0 + (r >> 6) * 128 + (r & 63)) * (K * 2) + c * 16;
            // W: LDS rows [32 w', 32 w' + 32) belong to wave column w'; row = 16 nf + fragment row
            const int wcol = r >> 5, nf = (r >> 4) & 1, fi = r & 15;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                int ch;
                if constexpr (EPI == EPI_F32) ch = chan_of<0>(ni, nf, fi);
                else if constexpr (EPI == EPI_SWIGLU) ch = chan_of<1>(ni, nf, fi);
                else ch = tr ? chan_of<2>(ni, nf, fi) : chan_of<0>(ni, nf, fi);
                voffW[ni][i] = (wcol * 64 + ch) * (K * 2) + c * 16;
            }
        }
    }
    const int hiA = 64 * K * 2;
    auto issue = [&](int kind, int buf, int kt) {          // kind: 0 W-lo, 1 A-lo, 2 W-hi, 3 A-hi (compile-time after inlining)
        if constexpr (DBG == 1) return;
        char* dst = smem + buf * BUF_BYTES + kind * HALF_BYTES + wave * 1024;
        const int soff = kt * 128;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (kind & 1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_p)(dst + i * 8192), 16, voffA[i] + (kind == 3 ? hiA : 0), soff, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void_p)(dst + i * 8192), 16, voffW[kind >> 1][i], soff, 0, 0);
        }
    };

    // ---- fragment addresses: row = (wave's first row of the half-tile) + 16 f + l15, chunk (4 ks + q4) ^ (l15 >> 1)
    const int swz = l15 >> 1;
    int offA[2], offW[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ch = ((ks * 4 + q4) ^ swz) << 4;
        offA[ks] = (wr * 64 + l15) * 128 + ch;
        offW[ks] = (wc * 32 + l15) * 128 + ch;
    }

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    bf16x8 fa[4][2], fwl[2][2], fwh[2][2];
    auto read_a = [&](int buf, int hi) {
        const char* base = smem + buf * BUF_BYTES + (hi ? 3 : 1) * HALF_BYTES;
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[f][ks] = *reinterpret_cast<const bf16x8*>(base + offA[ks] + f * 2048);
    };
    auto read_w = [&](int buf, int hi, bf16x8 (&fw)[2][2]) {
        const char* base = smem + buf * BUF_BYTES + (hi ? 2 : 0) * HALF_BYTES;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fw[f][ks] = *reinterpret_cast<const bf16x8*>(base + offW[ks] + f * 2048);
    };

    const int nk = K >> 6;             // launcher: even

    // SWAP: W fragment as the MFMA A operand -> a 16 x 16 block holds C^T (lane = token l15, registers = channels 4 q4 + r)
    auto main_loop = [&](auto swap_c) {
        constexpr bool SWAP = decltype(swap_c)::value;
        auto mfma_quadrant = [&](int mi, int ni, bf16x8 (&fw)[2][2]) {
            if constexpr (DBG == 3) return;
            if (!rows_valid) return;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[mi * 4 + f][ni * 2 + n] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[n][ks], fa[f][ks], acc[mi * 4 + f][ni * 2 + n], 0, 0, 0)
                                                           : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[f][ks], fw[n][ks], acc[mi * 4 + f][ni * 2 + n], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        // One K-tile = four phases.  BUF is the LDS buffer of tile t; tile t + 2 restages the same buffer.
        // ISSUE: 4 = issue all four half-tiles (steady state), 1 = only phase 0's (tile nk - 2), 0 = none (tile nk - 1)
        auto k_tile = [&](auto buf_c, auto issue_c, auto wait_c, int t) {
            constexpr int BUF = decltype(buf_c)::value;
            constexpr int ISSUE = decltype(issue_c)::value;
            constexpr int WAIT = decltype(wait_c)::value;          // vmcnt at phase 3: 6 steady, 0 for tile nk - 2, -1 none
            // ---- phase 0
            if constexpr (DBG != 2) {
                read_w(BUF, 0, fwl);
                __builtin_amdgcn_sched_barrier(0);
                read_a(BUF, 0);
            }
            if constexpr (ISSUE >= 1) issue(3, BUF ^ 1, t + 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DBG != 2) wait_lgkmcnt<8>();              // the W-lo reads (issued first) have returned: slot 0 may be restaged in phase 1
            __builtin_amdgcn_s_barrier();
            wait_lgkmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
            mfma_quadrant(0, 0, fwl);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- phase 1
            if constexpr (DBG != 2) read_w(BUF, 1, fwh);
            if constexpr (ISSUE >= 4) issue(0, BUF, t + 2);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            wait_lgkmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
            mfma_quadrant(0, 1, fwh);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- phase 2
            if constexpr (DBG != 2) read_a(BUF, 1);
            if constexpr (ISSUE >= 4) issue(1, BUF, t + 2);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            wait_lgkmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
            mfma_quadrant(1, 1, fwh);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- phase 3
            if constexpr (ISSUE >= 4) issue(2, BUF, t + 2);
            // K-tile t + 1 has landed (this wave's pieces; the barrier covers the others'); it is first read in the next phase
            if constexpr (WAIT >= 0 && DBG != 1) wait_vmcnt<WAIT>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            mfma_quadrant(1, 0, fwl);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I4 = std::integral_constant<int, 4>;
        using I6 = std::integral_constant<int, 6>;
        using IM = std::integral_constant<int, -1>;
        for (int t = 0; t < nk - 2; t += 2) {
            k_tile(I0{}, I4{}, I6{}, t);
            k_tile(I1{}, I4{}, I6{}, t + 1);
        }
        k_tile(I0{}, I1{}, I0{}, nk - 2);
        k_tile(I1{}, I0{}, IM{}, nk - 1);
    };

    // ---- prologue: half-tiles 0..6 in flight
#pragma unroll
    for (int j = 0; j < 4; ++j) issue(j, 0, 0);
#pragma unroll
    for (int j = 0; j < 3; ++j) issue(j, 1, 1);

    // LayerNorm fold, consumer side (GemmArgs): (mean, 1/std) of the tile's 256 rows from the producer's per-64-column partial sums
    // and the tile's 256 (c1, c2) channel constants, into LDS behind the ring while the first tiles are in flight; the K loop's
    // barriers order these writes before the epilogue's reads.  Without the fold the same epilogue runs on (0, 1), 0, bias.
    constexpr bool LN_CONS = (EPI == EPI_SWIGLU || EPI == EPI_HEADS) && (DBG == 0 || DBG == 9);
    [[maybe_unused]] float2* lnst = reinterpret_cast<float2*>(smem + RING_BYTES);               // [256] (mean, rstd)
    [[maybe_unused]] float* lnc = reinterpret_cast<float*>(smem + RING_BYTES + 2048);           // c1[256] then c2[256]
    if constexpr (LN_CONS) {
        const bool ln_fold = g.ln_part != nullptr;
        const int np = K >> 6;
        const int r = tid >> 1, sub = tid & 1;           // two threads per row
        float sum = 0.f, sq = 0.f;
        if (ln_fold) {
            int m = m0 + r;
            m = m < M ? m : M - 1;
            const float2* pp = reinterpret_cast<const float2*>(g.ln_part) + (size_t)m * np;
#pragma unroll 6
            for (int i = sub; i < np; i += 2) {
                const float2 v = pp[i];
                sum += v.x;
                sq += v.y;
            }
        }
        sum += dpp_move<0xB1>(sum);
        sq += dpp_move<0xB1>(sq);
        if (sub == 0) {
            const float inv_k = 1.0f / (float)K;
            const float mean = sum * inv_k;
            const float var = fmaxf(sq * inv_k - mean * mean, 0.f);
            lnst[r] = ln_fold ? make_float2(mean, rsqrtf(var + g.ln_eps)) : make_float2(0.f, 1.f);
        }
        const int ct = 511 - tid;           // the last 128 threads bring in the channel constants, 16 bytes each
        if (ct < 128) {
            const bool first = ct < 64;
            const float* src = ln_fold ? (first ? g.ln_c1 + n0 + ct * 4 : g.ln_c2 + n0 + (ct - 64) * 4)
                                       : ((first || !g.bias) ? nullptr : g.bias + n0 + (ct - 64) * 4);
            *reinterpret_cast<f32x4_t*>(lnc + ct * 4) = src ? *reinterpret_cast<const f32x4_t*>(src) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    }

    if constexpr (DBG == 2) {       // ablation: fragments are read once
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        read_w(0, 0, fwl);
        read_w(0, 1, fwh);
        read_a(0, 0);
    }
    // tile 0 has landed.  (With the fold the plain loads above were consumed already, so they are not part of the count.)
    wait_vmcnt<6>();
    __builtin_amdgcn_s_barrier();
    if constexpr (DBG == 9) t_pro = __builtin_amdgcn_s_memrealtime();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // stagger: wave row 1 runs one barrier behind wave row 0
    if constexpr (EPI == EPI_HEADS) {
        if (tr) main_loop(std::true_type{});
        else main_loop(std::false_type{});
    } else {
        main_loop(std::true_type{});
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();          // re-align the two wave rows
    if constexpr (DBG == 9) t_main = __builtin_amdgcn_s_memrealtime();
    if (!rows_valid) return;

    // ---- epilogues.  Transposed: lane (l15, q4) holds, for block (mb = 0..7, nb = 2 ni + nf): token row m0 + wr*128 + mb*16 + l15,
    //      channels n0 + wc*64 + chan_of(ni, nf, 4 q4 + r), r = 0..3
    const int mrow0 = m0 + wr * 128 + l15;
    const int ncol0 = n0 + wc * 64;
    [[maybe_unused]] const float2* ln = lnst + wr * 128;
    [[maybe_unused]] const float* lc1 = lnc + wc * 64;
    [[maybe_unused]] const float* lc2 = lnc + 256 + wc * 64;
    if constexpr (EPI == EPI_F32) {
        // fp32 output / residual update (transformer.py:692-700), adaLN gate (:674, 688); LayerNorm fold, producer side: bf16 image of
        // the updated rows + (sum, sum of squares) of the ROUNDED values over this wave's 64-column block
        const bool accum = g.accumulate != 0;
        const bool prod = g.xb != nullptr;
        f32x4_t bia[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
            bia[nb] = g.bias ? *reinterpret_cast<const f32x4_t*>(g.bias + ncol0 + 4 * q4 + nb * 16) : f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
            const int m = mrow0 + mb * 16;
            const int mc = m < M ? m : M - 1;
            float* __restrict__ crow = g.C + (size_t)mc * g.ldc + ncol0 + 4 * q4;
            const float* grow = g.gate ? g.gate + (size_t)(mc / g.gate_rows) * g.gate_ld + ncol0 + 4 * q4 : nullptr;
            f32x4_t old[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) old[nb] = accum ? *reinterpret_cast<const f32x4_t*>(crow + nb * 16) : f32x4_t{0.f, 0.f, 0.f, 0.f};
            float sum = 0.f, sq = 0.f;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                f32x4_t v = acc[mb][nb] + bia[nb];
                if (grow) v *= *reinterpret_cast<const f32x4_t*>(grow + nb * 16);
                v += old[nb];
                if (m < M) *reinterpret_cast<f32x4_t*>(crow + nb * 16) = v;
                if (prod) {
                    bf16x4 xr;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        xr[e] = f32_to_bf16(v[e]);
                        const float f = bf16_to_f32(xr[e]);
                        sum += f;
                        sq += f * f;
                    }
                    if (m < M) *reinterpret_cast<bf16x4*>(g.xb + (size_t)m * N + ncol0 + 4 * q4 + nb * 16) = xr;
                }
            }
            if (prod) {       // add the four lanes (q4 = 0..3) that share the token row
                sum += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(sum), 0x401F));       // lane ^ 16
                sq += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(sq), 0x401F));
                u32x2 a = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum), __float_as_uint(sum), false, false);
                u32x2 b = __builtin_amdgcn_permlane32_swap(__float_as_uint(sq), __float_as_uint(sq), false, false);
                sum = __uint_as_float(a[0]) + __uint_as_float(a[1]);
                sq = __uint_as_float(b[0]) + __uint_as_float(b[1]);
                if (q4 == 0 && m < M)
                    *reinterpret_cast<float2*>(g.ln_part_out + ((size_t)m * (N >> 6) + (ncol0 >> 6)) * 2) = make_float2(sum, sq);
            }
        }
    } else if constexpr (EPI == EPI_SWIGLU) {
        // H = (v + b_v) * silu(gate + b_g) (transformer.py:232-235): value rows are channels [0, 32) of the wave's 64, gate rows
        // [32, 64) (pack_rows interleave); PERM 1 puts value and gate of hidden columns hc0 + 8 q4 + 4 nf + r into this lane
        const int ldh = N >> 1;
        f32x4_t c1v[2], c1g[2], c2v[2], c2g[2];
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
            c1v[nf] = *reinterpret_cast<const f32x4_t*>(lc1 + q4 * 8 + nf * 4);
            c1g[nf] = *reinterpret_cast<const f32x4_t*>(lc1 + 32 + q4 * 8 + nf * 4);
            c2v[nf] = *reinterpret_cast<const f32x4_t*>(lc2 + q4 * 8 + nf * 4);
            c2g[nf] = *reinterpret_cast<const f32x4_t*>(lc2 + 32 + q4 * 8 + nf * 4);
        }
        bf16_t* __restrict__ hbase = g.H + (ncol0 >> 1) + q4 * 8;
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
            const int m = mrow0 + mb * 16;
            const float2 st = ln[mb * 16 + l15];
            unsigned pk[4];
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) {
                float hv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = st.y * (acc[mb][nf][e] - st.x * c1v[nf][e]) + c2v[nf][e];
                    const float gt = st.y * (acc[mb][2 + nf][e] - st.x * c1g[nf][e]) + c2g[nf][e];
                    hv[e] = v * silu_f(gt);
                }
                pk[2 * nf] = pack_bf16x2(hv[0], hv[1]);
                pk[2 * nf + 1] = pack_bf16x2(hv[2], hv[3]);
            }
            if (m < M) *reinterpret_cast<u32x4*>(hbase + (size_t)m * ldh) = u32x4{pk[0], pk[1], pk[2], pk[3]};
        }
    } else {   // EPI_HEADS: split into heads, LayerNorm fold, partial RoPE on d < 32 (transformer.py:158-183, 438-452)
        const HeadsEpi& he = g.heads;
        const int hp = he.heads * 64;
        const int part = ncol0 / hp;
        const int head = (ncol0 - part * hp) >> 6;
        const int kind = he.kind[part];
        bf16_t* __restrict__ dst = he.out[part];
        const int S = he.S, Spad = he.Spad;
        if (tr) {
            // q / k, row-major [B, H, Spad, 64].  PERM 2: block ni = 0 holds d = 16 nf + 4 q4 + r (the rotation partner d + 16 is block
            // nf + 1 of the same lane), block ni = 1 holds d = 32 + 8 q4 + 4 nf + r (8 consecutive channels)
            f32x4_t c1[4], c2[4];
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) {
                c1[nf] = *reinterpret_cast<const f32x4_t*>(lc1 + nf * 16 + 4 * q4);
                c2[nf] = *reinterpret_cast<const f32x4_t*>(lc2 + nf * 16 + 4 * q4);
                c1[2 + nf] = *reinterpret_cast<const f32x4_t*>(lc1 + 32 + q4 * 8 + nf * 4);
                c2[2 + nf] = *reinterpret_cast<const f32x4_t*>(lc2 + 32 + q4 * 8 + nf * 4);
            }
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) {
                const int m = mrow0 + mb * 16;
                const int mc = m < M ? m : M - 1;
                const int b = mc / S;
                const int sq_ = mc - b * S;
                const int ob = (kind & 4) ? ((b * S) & 3) : 0;
                const float2 st = ln[mb * 16 + l15];
                f32x4_t x[4];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[nb][e] = st.y * (acc[mb][nb][e] - st.x * c1[nb][e]) + c2[nb][e];
                if (kind & 2) {
                    const f32x4_t cs = *reinterpret_cast<const f32x4_t*>(he.rope_cos + (size_t)sq_ * 16 + 4 * q4);
                    const f32x4_t sn = *reinterpret_cast<const f32x4_t*>(he.rope_sin + (size_t)sq_ * 16 + 4 * q4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x1 = x[0][e], x2 = x[1][e];
                        x[0][e] = x1 * cs[e] - x2 * sn[e];
                        x[1][e] = x2 * cs[e] + x1 * sn[e];
                    }
                }
                if (m < M) {
                    bf16_t* row = dst + ((size_t)(b * he.heads + head) * Spad + sq_ + ob) * 64;
                    *reinterpret_cast<u32x2*>(row + 4 * q4) = u32x2{pack_bf16x2(x[0][0], x[0][1]), pack_bf16x2(x[0][2], x[0][3])};
                    *reinterpret_cast<u32x2*>(row + 16 + 4 * q4) = u32x2{pack_bf16x2(x[1][0], x[1][1]), pack_bf16x2(x[1][2], x[1][3])};
                    *reinterpret_cast<u32x4*>(row + 32 + 8 * q4) = u32x4{pack_bf16x2(x[2][0], x[2][1]), pack_bf16x2(x[2][2], x[2][3]),
                                                                         pack_bf16x2(x[3][0], x[3][1]), pack_bf16x2(x[3][2], x[3][3])};
                }
            }
        } else {
            // V^T [B, H, 64, Spad] (no rotation): un-swapped accumulators, lane = channel d = 16 nb + l15, registers = the four consecutive
            // token rows mb*16 + 4 q4 + e.  Columns of sequence b are shifted by (b S) & 3 (kind bit 2) so that those four tokens are
            // an 8-byte aligned group of vt_pos order (see gemm_bf16.hip / sat_common.h vt_pos)
            const bool shift = (kind & 4) != 0;
            float c1[4], c2[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                c1[nb] = lc1[nb * 16 + l15];
                c2[nb] = lc2[nb * 16 + l15];
            }
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) {
                const int mbase = m0 + wr * 128 + mb * 16 + 4 * q4;          // multiple of 4
                const f32x4_t* sp = reinterpret_cast<const f32x4_t*>(ln + mb * 16 + 4 * q4);
                const f32x4_t st01 = sp[0], st23 = sp[1];                    // (mean, rstd) of rows e = 0, 1 / 2, 3
                int bb[4], ss[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int mm = mbase + e;
                    mm = mm < M ? mm : M - 1;
                    bb[e] = mm / S;
                    ss[e] = mm - bb[e] * S;
                }
                const bool whole = mbase + 3 < M && bb[0] == bb[3];
                const int ob0 = shift ? ((bb[0] * S) & 3) : 0;
                const size_t hb0 = ((size_t)(bb[0] * he.heads + head) * 64) * Spad;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float mean = e < 2 ? st01[2 * e] : st23[2 * e - 4];
                        const float rstd = e < 2 ? st01[2 * e + 1] : st23[2 * e - 3];
                        v[e] = rstd * (acc[mb][nb][e] - mean * c1[nb]) + c2[nb];
                    }
                    const size_t drow = (size_t)(nb * 16 + l15) * Spad;
                    if (whole && shift) {         // aligned: (ss[0] + ob) % 4 == mbase % 4 == 0
                        *reinterpret_cast<u32x2*>(dst + hb0 + vt_pos(ss[0] + ob0) + drow) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (mbase + e < M) {
                                const int ob = shift ? ((bb[e] * S) & 3) : 0;
                                dst[((size_t)(bb[e] * he.heads + head) * 64) * Spad + vt_pos(ss[e] + ob) + drow] = f32_to_bf16(v[e]);
                            }
                    }
                }
            }
        }
    }
    if constexpr (DBG == 9) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) {
            unsigned hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* o = ts + (size_t)blockIdx.x * 6;
            o[0] = t_start; o[1] = t_pro; o[2] = t_main; o[3] = __builtin_amdgcn_s_memrealtime(); o[4] = hw; o[5] = xcc;
        }
    }
}

template <int EPI, int DBG = 0>
int launch_ph8(const GemmArgs& a, hipStream_t stream) {
    constexpr int LDS = 131072 + 4096;          // ring + (mean, rstd) per row + (c1, c2) per column
    SAT_CHECK_ARG(a.N % 256 == 0, SAT_E_UNSUPPORTED, "gemm(8-phase): N=%d not a multiple of 256", a.N);
    SAT_CHECK_ARG(a.K % 128 == 0, SAT_E_UNSUPPORTED, "gemm(8-phase): K=%d must be a multiple of 128", a.K);
    SAT_CHECK_ARG((uint64_t)a.M * (uint64_t)a.K * 2u < (1ull << 31), SAT_E_UNSUPPORTED, "gemm(8-phase): A larger than 2 GiB");
    SAT_CHECK_ARG(!a.fp8 && !a.H8, SAT_E_UNSUPPORTED, "gemm(8-phase): bf16 operands only");
    constexpr bool LN_CONS = EPI == EPI_SWIGLU || EPI == EPI_HEADS;
    SAT_CHECK_ARG(LN_CONS || !a.ln_part, SAT_E_UNSUPPORTED, "gemm(8-phase): the LayerNorm fold is finished by the SwiGLU / heads epilogues");
    SAT_CHECK_ARG((!a.xb && !a.ln_part_out) || (EPI == EPI_F32 && a.xb && a.ln_part_out), SAT_E_UNSUPPORTED,
                  "gemm(8-phase): the bf16 image / row statistics come from the fp32-output epilogue");
    if constexpr (EPI == EPI_HEADS) {
        SAT_CHECK_ARG((a.heads.heads * 64) % 256 == 0 && a.N == a.heads.parts * a.heads.heads * 64, SAT_E_UNSUPPORTED,
                      "gemm(8-phase): a 256-column tile must not straddle q / k / v (heads=%d)", a.heads.heads);
        for (int p = 0; p < a.heads.parts; ++p)
            SAT_CHECK_ARG((a.heads.kind[p] & 3) != 3, SAT_E_UNSUPPORTED, "gemm(8-phase): no rotation on a transposed destination");
    }
    auto kern = gemm_ph8_kernel<EPI, DBG>;
    SAT_TRY(sat_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS));
    const int tiles = cdiv(a.M, 256) * (a.N / 256);
    unsigned long long* ts = nullptr;
#ifdef SAT_GEMM_EXPERIMENTS
    if constexpr (DBG == 9) {
        if (!g_ts_buf) SAT_HIP(hipMalloc(&g_ts_buf, 8192 * 6 * sizeof(unsigned long long)));
        SAT_CHECK_ARG(tiles <= 8192, SAT_E_UNSUPPORTED, "timestamp buffer holds 8192 workgroups");
        ts = g_ts_buf;
    }
#endif
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), LDS, stream, a, ts);
    SAT_LAUNCH_CHECK();
    return 0;
}

}  // namespace

bool sat_gemm_ph8_supports(int epi, const GemmArgs& a) {
    if (a.fp8 || a.H8 || a.N % 256 || a.K % 128 || (uint64_t)a.M * (uint64_t)a.K * 2u >= (1ull << 31)) return false;
    if (epi == EPI_HEADS) return (a.heads.heads * 64) % 256 == 0;
    return true;
}

#ifdef SAT_GEMM_EXPERIMENTS
extern "C" int sat_gemm_ph8_timestamps(unsigned long long* out_host, int n_wg) {
    SAT_CHECK_ARG(g_ts_buf && n_wg <= 8192, SAT_E_INVALID, "no timestamps recorded");
    SAT_HIP(hipDeviceSynchronize());
    SAT_HIP(hipMemcpy(out_host, g_ts_buf, (size_t)n_wg * 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}
#endif

int sat_launch_gemm_ph8(int epi, const GemmArgs& a, hipStream_t stream) {
    const int dbg = (a.variant & 0xfff) / 100;
    switch (epi) {
        case EPI_F32:
        case EPI_RESID:
            switch (dbg) {
                case 0: return launch_ph8<EPI_F32>(a, stream);
#ifdef SAT_GEMM_EXPERIMENTS
                case 1: return launch_ph8<EPI_F32, 1>(a, stream);
                case 2: return launch_ph8<EPI_F32, 2>(a, stream);
                case 3: return launch_ph8<EPI_F32, 3>(a, stream);
#endif
            }
            break;
        case EPI_SWIGLU:
            if (dbg == 0) return launch_ph8<EPI_SWIGLU>(a, stream);
#ifdef SAT_GEMM_EXPERIMENTS
            if (dbg == 9) return launch_ph8<EPI_SWIGLU, 9>(a, stream);
#endif
            break;
        case EPI_HEADS:
            if (dbg == 0) return launch_ph8<EPI_HEADS>(a, stream);
            break;
    }
    sat_set_error("gemm(8-phase): epilogue %d / ablation %d not built", epi, dbg);
    return SAT_E_UNSUPPORTED;
}
