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

template <int EPI>
struct Ph8Traits {
    static constexpr int PERM = EPI == EPI_F32 ? 0 : (EPI == EPI_SWIGLU ? 1 : 2);
};

// DBG (tools/gpu_probe.py ablations, wrong results): 1 no LDS-DMA in the loop, 2 no ds_read (fragments stay), 3 no MFMA
// OPT (experiments build, correct results): bit 0 no explicit lgkmcnt(0) behind the barrier (the compiler's counted waits only),
// bit 1 no s_setprio, bit 2 LDS-DMA issued at the head of the MFMA section instead of the load section,
// bit 3 static priority for waves 4-7 instead of flips
template <int EPI, int DBG = 0, int OPT = 0>
__global__ __launch_bounds__(512) void gemm_ph8_kernel(GemmArgs g) {
    constexpr int PERM = Ph8Traits<EPI>::PERM;
    constexpr int BUF_BYTES = 65536, HALF_BYTES = 16384;
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
            for (int ni = 0; ni < 2; ++ni) voffW[ni][i] = (wcol * 64 + chan_of<PERM>(ni, nf, fi)) * (K * 2) + c * 16;
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
    auto mfma_quadrant = [&](int mi, int ni, bf16x8 (&fw)[2][2]) {
        if constexpr (DBG == 3) return;
        if constexpr (!(OPT & 10)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[mi * 4 + f][ni * 2 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[n][ks], fa[f][ks], acc[mi * 4 + f][ni * 2 + n], 0, 0, 0);
        if constexpr (!(OPT & 10)) __builtin_amdgcn_s_setprio(0);
    };

    const int nk = K >> 6;             // launcher: even, >= 4

    // One K-tile = four phases.  BUF is the LDS buffer of tile t; tile t + 2 restages the same buffer.
    // ISSUE: 4 = issue all four half-tiles (steady state), 1 = only phase 0's (tile nk - 2), 0 = none (tile nk - 1)
    auto k_tile = [&](auto buf_c, auto issue_c, auto wait_c, int t) {
        constexpr int BUF = decltype(buf_c)::value;
        constexpr int ISSUE = decltype(issue_c)::value;
        constexpr int WAIT = decltype(wait_c)::value;          // vmcnt at phase 3: 6 steady, 0 for tile nk - 2, -1 none
        constexpr bool LATE = (OPT & 4) != 0;
        // ---- phase 0
        if constexpr (DBG != 2) {
            read_w(BUF, 0, fwl);
            __builtin_amdgcn_sched_barrier(0);
            read_a(BUF, 0);
        }
        if constexpr (ISSUE >= 1 && !LATE) issue(3, BUF ^ 1, t + 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DBG != 2) wait_lgkmcnt<8>();              // the W-lo reads (issued first) have returned: slot 0 may be restaged in phase 1
        __builtin_amdgcn_s_barrier();
        if constexpr (ISSUE >= 1 && LATE) issue(3, BUF ^ 1, t + 1);
        if constexpr (!(OPT & 1)) wait_lgkmcnt<0>();
        __builtin_amdgcn_sched_barrier(0);
        mfma_quadrant(0, 0, fwl);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 1
        if constexpr (DBG != 2) read_w(BUF, 1, fwh);
        if constexpr (ISSUE >= 4 && !LATE) issue(0, BUF, t + 2);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        if constexpr (ISSUE >= 4 && LATE) issue(0, BUF, t + 2);
        if constexpr (!(OPT & 1)) wait_lgkmcnt<0>();
        __builtin_amdgcn_sched_barrier(0);
        mfma_quadrant(0, 1, fwh);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 2
        if constexpr (DBG != 2) read_a(BUF, 1);
        if constexpr (ISSUE >= 4 && !LATE) issue(1, BUF, t + 2);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        if constexpr (ISSUE >= 4 && LATE) issue(1, BUF, t + 2);
        if constexpr (!(OPT & 1)) wait_lgkmcnt<0>();
        __builtin_amdgcn_sched_barrier(0);
        mfma_quadrant(1, 1, fwh);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 3
        if constexpr (ISSUE >= 4 && !LATE) issue(2, BUF, t + 2);
        // K-tile t + 1 has landed (this wave's pieces; the barrier covers the others').  LATE: this phase's half-tile is not issued yet
        if constexpr (WAIT >= 0 && DBG != 1) wait_vmcnt<(LATE && WAIT == 6) ? 4 : WAIT>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        if constexpr (ISSUE >= 4 && LATE) issue(2, BUF, t + 2);
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

    // ---- prologue: half-tiles 0..6 in flight, tile 0 landed
#pragma unroll
    for (int j = 0; j < 4; ++j) issue(j, 0, 0);
#pragma unroll
    for (int j = 0; j < 3; ++j) issue(j, 1, 1);
    if constexpr (DBG == 2) {       // ablation: fragments are read once
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        read_w(0, 0, fwl);
        read_w(0, 1, fwh);
        read_a(0, 0);
    }
    wait_vmcnt<6>();
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // stagger: wave row 1 runs one barrier behind wave row 0
    if constexpr ((OPT & 8) != 0) {
        if (wr == 1) __builtin_amdgcn_s_setprio(1);
    }

    for (int t = 0; t < nk - 2; t += 2) {
        k_tile(I0{}, I4{}, I6{}, t);
        k_tile(I1{}, I4{}, I6{}, t + 1);
    }
    k_tile(I0{}, I1{}, I0{}, nk - 2);
    k_tile(I1{}, I0{}, IM{}, nk - 1);
    if (wr == 0) __builtin_amdgcn_s_barrier();          // re-align the two wave rows

    // ---- epilogue: lane (l15, q4) holds, for block (mb = 0..7, nb = 0..3): token row m0 + wr*128 + mb*16 + l15,
    //      channels n0 + wc*64 + chan_of(nb >> 1, nb & 1, 4 q4 + r), r = 0..3
    const int mrow0 = m0 + wr * 128 + l15;
    const int ncol0 = n0 + wc * 64;
    if constexpr (EPI == EPI_F32) {
        const bool accum = g.accumulate != 0;
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
            const int m = mrow0 + mb * 16;
            if (m >= M) continue;
            float* __restrict__ crow = g.C + (size_t)m * g.ldc + ncol0 + 4 * q4;
            const float* grow = g.gate ? g.gate + (size_t)(m / g.gate_rows) * g.gate_ld + ncol0 + 4 * q4 : nullptr;
            f32x4_t old[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) old[nb] = accum ? *reinterpret_cast<const f32x4_t*>(crow + nb * 16) : f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                f32x4_t v = acc[mb][nb];
                if (g.bias) v += *reinterpret_cast<const f32x4_t*>(g.bias + ncol0 + 4 * q4 + nb * 16);
                if (grow) v *= *reinterpret_cast<const f32x4_t*>(grow + nb * 16);
                v += old[nb];
                *reinterpret_cast<f32x4_t*>(crow + nb * 16) = v;
            }
        }
    }
}

template <int EPI, int DBG = 0, int OPT = 0>
int launch_ph8(const GemmArgs& a, hipStream_t stream) {
    constexpr int LDS = 131072;
    SAT_CHECK_ARG(a.N % 256 == 0, SAT_E_UNSUPPORTED, "gemm(8-phase): N=%d not a multiple of 256", a.N);
    SAT_CHECK_ARG(a.K % 128 == 0 && a.K >= 256, SAT_E_UNSUPPORTED, "gemm(8-phase): K=%d must be a multiple of 128, >= 256", a.K);
    SAT_CHECK_ARG((uint64_t)a.M * (uint64_t)a.K * 2u < (1ull << 31), SAT_E_UNSUPPORTED, "gemm(8-phase): A larger than 2 GiB");
    SAT_CHECK_ARG(!a.fp8 && !a.ln_part && !a.xb, SAT_E_UNSUPPORTED, "gemm(8-phase): bf16 operands, no LayerNorm fold yet");
    auto kern = gemm_ph8_kernel<EPI, DBG, OPT>;
    SAT_TRY(sat_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS));
    const int tiles = cdiv(a.M, 256) * (a.N / 256);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), LDS, stream, a);
    SAT_LAUNCH_CHECK();
    return 0;
}

}  // namespace

int sat_launch_gemm_ph8(int epi, const GemmArgs& a, hipStream_t stream) {
    const int dbg = (a.variant & 0xfff) / 100;
    switch (epi) {
        case EPI_F32:
        case EPI_RESID:
            switch (dbg) {
                case 0:
#ifdef SAT_GEMM_EXPERIMENTS
                    switch ((a.variant >> 16) & 0xff) {
                        case 1: return launch_ph8<EPI_F32, 0, 1>(a, stream);
                        case 2: return launch_ph8<EPI_F32, 0, 2>(a, stream);
                        case 3: return launch_ph8<EPI_F32, 0, 3>(a, stream);
                        case 4: return launch_ph8<EPI_F32, 0, 4>(a, stream);
                        case 5: return launch_ph8<EPI_F32, 0, 5>(a, stream);
                        case 8: return launch_ph8<EPI_F32, 0, 8>(a, stream);
                        case 9: return launch_ph8<EPI_F32, 0, 9>(a, stream);
                    }
#endif
                    return launch_ph8<EPI_F32>(a, stream);
#ifdef SAT_GEMM_EXPERIMENTS
                case 1: return launch_ph8<EPI_F32, 1>(a, stream);
                case 2: return launch_ph8<EPI_F32, 2>(a, stream);
                case 3: return launch_ph8<EPI_F32, 3>(a, stream);
#endif
            }
            break;
    }
    sat_set_error("gemm(8-phase): epilogue %d / ablation %d not built", epi, dbg);
    return SAT_E_UNSUPPORTED;
}
