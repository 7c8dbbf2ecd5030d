// 256 x 256 x 64 bf16 MFMA GEMM in the 8-wave / 256-register / 8-phase regime (round 3):
//     C[M,N] = A[M,K] . W[N,K]^T      A, W bf16 (K contiguous), fp32 accumulate
// for the DiT projections (FF-in SwiGLU, to_qkv, FF-out at one prompt; every block GEMM from 4 prompts per GPU on) -- the nn.Linear
// calls at models/transformer.py:222,270,314,311-312,319 of the reference.  gfx950 only.
//
// Main loop (one workgroup = 8 waves = 2 (M) x 4 (N), wave tile 128 x 64, v_mfma_f32_16x16x32_bf16, 128 accumulator registers):
//   * a K-tile (64 k) is FOUR 16-KiB half-tiles in LDS, in the order a wave consumes them:
//         kind 0  W-lo : the first 32 channels of every wave's 64       kind 1  A-lo : the first 64 rows of every wave's 128
//         kind 2  W-hi : the last 32 channels                           kind 3  A-hi : the last 64 rows
//     two K-tiles are resident (128 KiB ring); rows are 128 B with the 16-byte chunk XOR ((row >> 1) & 7), applied on the
//     SOURCE side of the LDS-DMA (buffer_load_dwordx4 ... lds writes lane-linear), so ds_read_b128 fragment reads are
//     conflict-free;
//   * a K-tile is TWO PHASES of 32 MFMAs (two 64 x 32 quadrants of the wave tile each), four barriers per K-tile:
//         phase A: read W-lo, W-hi (4 + 4) and A-lo (8) fragments; issue W-hi, A-hi of tile t + 1   -> A-lo x W-lo, A-lo x W-hi
//         phase B: read A-hi (8);                                  issue W-lo, A-lo of tile t + 2   -> A-hi x W-hi, A-hi x W-lo
//     every phase is { ds_reads ; two half-tiles of LDS-DMA (4 instructions per wave) ; counted vmcnt ; lgkmcnt(0) ; s_barrier ;
//     32 MFMA ; s_barrier };
//   * the two wave rows run staggered by one barrier (wave row 1 passes one extra s_barrier up front), so on every SIMD one
//     wave is in its MFMA section while its partner reads fragments and issues DMA;
//   * the only vector-memory waits of the loop are COUNTED: vmcnt(8) in phase A (A-hi of this tile has landed: W-lo, A-lo, W-hi,
//     A-hi of the next stay in flight), vmcnt(6) in phase B (W-lo, A-lo, W-hi of the next tile have landed); a half-tile is
//     first read one barrier after the wait that covers it.  Every load section retires its ds_reads BEFORE its barrier, so a
//     slot may be restaged by the partner wave row in the very next interval: W-lo / A-lo / W-hi of tile t (read in phase A) are
//     rewritten from phase B on, A-hi from phase A of tile t + 1.
//   (The first version of this kernel ran FOUR phases of 16 MFMAs, 8 barriers per K-tile, half-tiles issued seven ahead, one
//   vmcnt(6) per K-tile -- the schedule of the in-image guide; two phases measured +5..7 % on the SwiGLU GEMM, +3 % on fp32 output
//   (profiles/r03_ph8_two_phases.txt).  It is kept as `main_loop` for A/B, variant bit 18, experiments build.)
//   Measured in the loop: 1.36 us per K-tile on 256 CUs = 1575 TFLOP/s (profiles/r03_ph8_ksweep_fixed_overhead.txt).
//
// Schedule (PERSISTENT workgroups, one per CU; Ph8Sched).  With one 136-KiB workgroup per CU nothing overlaps a tile's prologue and
// epilogue, and a launch of T tiles costs ceil(T / 256) rounds: measured 8-10 us of a 42-52 us tile
// (profiles/r03_ph8_workgroup_timeline.txt) and, at one prompt, 2 rounds for 1.5 rounds of work.  So:
//   * workgroup i walks `dp_rounds` whole tiles (logical tile s G + i in round s: the same neighbourhood per round as hardware
//     dispatch order, XCD-aware), then its share of the REMAINDER round: whole tiles (contiguous shares), or -- fp32 output with a long
//     reduction behind a whole round, sat_gemm_ph8_splits -- ONE K-range of a remainder tile: every remainder tile is cut along K into
//     equal parts, one workgroup per part, parts in proportion to cost (row tiles with <= 64 valid rows -- the 2 leftover rows of
//     M = 2 x 1025 -- count half);
//   * a partial K-range stores its raw accumulators to the workgroup's slab (lane-for-lane the register image, 1-KiB coalesced
//     pieces); a second launch (ph8_reduce_f32_kernel) adds a tile's slabs in ascending workgroup order -- bit-deterministic -- and
//     runs the fp32 / residual / LayerNorm-producer epilogue on the sums.  The slabs live in the CALLER's workspace (GemmArgs::slab):
//     nothing here allocates, synchronises or keeps state between launches, so plans on different streams are independent and a
//     forward can be captured into a hipGraph.  (Round 3 also built the in-kernel last-arriver fix-up with arrival tickets, and
//     contiguous stream-K shares: both measured slower than this -- profiles/r03_ph8_streamk_timeline.txt,
//     r03_ph8_ffout_streamk_negative.txt -- and are gone from the source.)
//   * the whole schedule is a handful of integers computed on the host per launch in closed form and passed by value: no device
//     tables, no cache, no lock;
//   * the LDS-DMA prologue of the next K-range (7 half-tiles) is issued BEFORE the epilogue of the current one -- the ring is
//     free after the last barrier of the main loop and the DMA needs no registers -- so its latency hides behind the epilogue.
//
// Epilogues run on TRANSPOSED accumulators (W fragment = MFMA A operand): lane l owns token row (l & 15) of a 16 x 16 block and
// four consecutive output channels; which channels a wave's W rows are is a free permutation applied where the DMA picks its
// source rows (chan_of), chosen per epilogue so that stores are 16 bytes and SwiGLU / RoPE partners are lane-local.
#include <algorithm>
#include <type_traits>

#include "sat_common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_void_p;
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

// x * sigmoid(x) with v_rcp_f32 (1 ulp) instead of the IEEE division sequence (10 instructions per element in the epilogue)
__device__ __forceinline__ float silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// LDS accesses of the LayerNorm / rotation constants that the COMPILER MUST NOT SEE (round 5).  While an LDS-DMA is in flight -- and
// in the persistent loop the next range's prologue always is, from `prepare` to the top of the next main loop -- the compiler's wait
// insertion treats every ds_read / ds_write it emits as a possible access to what the DMA is writing and puts s_waitcnt vmcnt(0) in front of
// it (it cannot tell the constants behind the ring from the ring).  That cost: the LayerNorm statistics of the next range were written only
// after its whole DMA prologue had LANDED (the 2 us of "prepare-next" in profiles/r04_ph8_heads_swiglu_timeline.txt), and every row block of
// an epilogue waited for the write acknowledgement of the previous row block's store before it read its (mean, rstd) -- eight dependent store
// round trips per tile.  Inline assembly keeps the vector-memory counter out of it; the lgkmcnt waits are then ours (lds_wait).
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(const char*)p;
}
__device__ __forceinline__ float lds_ld32(const void* p) {
    float v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(lds_addr(p)) : "memory");
    return v;
}
__device__ __forceinline__ float2 lds_ld64(const void* p) {
    float2 v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(lds_addr(p)) : "memory");
    return v;
}
__device__ __forceinline__ f32x4_t lds_ld128(const void* p) {
    f32x4_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(lds_addr(p)) : "memory");
    return v;
}
__device__ __forceinline__ void lds_st64(void* p, float2 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(lds_addr(p)), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_st128(void* p, f32x4_t v) { asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr(p)), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_tie() {}
template <typename T, typename... Ts>
__device__ __forceinline__ void lds_tie(T& v, Ts&... vs) {
    asm volatile("" : "+v"(v));          // (a use of v cannot be scheduled in front of this, and this not in front of the wait)
    lds_tie(vs...);
}
// all LDS reads issued so far have returned; every later use of the listed values sits behind the wait
template <typename... Ts>
__device__ __forceinline__ void lds_wait(Ts&... vs) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    lds_tie(vs...);
}

// Channel (0..63 inside the wave's 64) held by W half-tile `ni` (0 lo / 1 hi), fragment nf (0/1), fragment row i (0..15).
// After the MFMA lane (i' = l & 15, q = l >> 4) holds rows 4q..4q+3 of the fragment, i.e. fragment rows i = 4q + r.
//   PERM 0 (fp32 output, V^T):  natural order ni*32 + nf*16 + i  -> 4 lanes q write 64 contiguous bytes of a row per store
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

struct Ph8Sched {          // host-computed per launch (ph8_schedule), passed by value
    int G;                  // workgroups (== gridDim.x)
    int tiles_n;
    int tiles_m_full;       // row tiles of the "full" logical tile space
    int light;              // 1: one more row of tiles with <= 64 valid rows ("light": about half the time of a full tile: the W panel still streams)
    int light_first;        // work order: light tiles before the full ones (K-split schedules with whole rounds) instead of behind them
    int dp_rounds;          // whole tiles per workgroup
    int nkp;                // K-pair units (128 k) per tile
    int sk_tiles;           // tiles of the remainder space (the full tiles left over by the whole rounds, and the light tiles)
    int rem0;               // work-order position of the first remainder tile (remainder tile j = position rem0 + j)
    int split;              // 0: remainder tiles stay whole; 1: every remainder tile is cut along K, one workgroup per part
    int sk_q, sk_r;         // split 0: workgroup i takes sk_q (+ 1 if i < sk_r) consecutive remainder tiles
    int cls_n[3], cls_p[3]; // split 1: three consecutive classes of remainder tiles, cls_n[c] tiles of cls_p[c] parts each
    float* sk_slab;         // split 1: [G][65536] raw accumulator images (caller's workspace)
};

// units [b, e) of the remainder space (unit = 128 k of one tile, tile j = units [j nkp, (j + 1) nkp)) that workgroup i works on
__host__ __device__ __forceinline__ void ph8_wg_units(const Ph8Sched& sc, int i, int& b, int& e) {
    if (!sc.split) {
        const int lo = i < sc.sk_r ? i : sc.sk_r, hi = i + 1 < sc.sk_r ? i + 1 : sc.sk_r;
        b = (i * sc.sk_q + lo) * sc.nkp;
        e = ((i + 1) * sc.sk_q + hi) * sc.nkp;
        return;
    }
    int w0 = 0, j0 = 0;
    b = e = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int n = sc.cls_n[c], p = sc.cls_p[c];
        if (i >= w0 && i < w0 + n * p) {
            const int j = j0 + (i - w0) / p, k = (i - w0) % p;
            b = j * sc.nkp + sc.nkp * k / p;
            e = j * sc.nkp + sc.nkp * (k + 1) / p;
        }
        w0 += n * p;
        j0 += n;
    }
}
// split 1: the workgroups first .. first + parts - 1 that hold remainder tile j's K-ranges
__host__ __device__ __forceinline__ void ph8_tile_parts(const Ph8Sched& sc, int j, int& first, int& parts) {
    int w0 = 0, j0 = 0;
    first = 0;
    parts = 1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int n = sc.cls_n[c], p = sc.cls_p[c];
        if (j >= j0 && j < j0 + n) {
            first = w0 + (j - j0) * p;
            parts = p;
        }
        w0 += n * p;
        j0 += n;
    }
}

struct Seg {                // one K-range of one tile
    int m0, n0;
    int kt0, nk;            // first K-tile, K-tiles (even)
    bool whole;             // the K-range covers the tile: plain epilogue
    bool tr;                // accumulator orientation (transposed unless a V^T destination)
};


// position in the work order -> tile.  Work order: the full tiles (short M: m fastest, the W panel of a column tile stays in one
// XCD's L2; long M: bands of 8 row tiles, n-major inside a band -- 8 A panels + the W panels in flight) with the light row behind
// them, or (light_first) in front of them.
__device__ __forceinline__ void ph8_tile_of(const Ph8Sched& sc, int id, int& tm, int& tn) {
    const int nl = sc.light ? sc.tiles_n : 0;
    const int lid = sc.light_first ? id : id - sc.tiles_m_full * sc.tiles_n;
    if (lid >= 0 && lid < nl) {
        tm = sc.tiles_m_full;
        tn = lid;
        return;
    }
    const int L = sc.light_first ? id - nl : id;
    const int tiles_m = sc.tiles_m_full, tiles_n = sc.tiles_n;
    if (tiles_m <= 12) {
        tn = L / tiles_m;
        tm = L - tn * tiles_m;
    } else {
        const int band_sz = 8 * tiles_n;
        const int band = L / band_sz;
        const int rem = L - band * band_sz;
        const int gm = min(8, tiles_m - band * 8);
        tn = rem / gm;
        tm = band * 8 + (rem - tn * gm);
    }
}

// fp32 output / residual update of ONE token row piece (transformer.py:692-700), adaLN gate (:674, 688), and the producer side of the
// LayerNorm fold: 16-bit image of the updated row + (sum, sum of squares) of the ROUNDED values over the wave's 64-column block.
// v[nb] = the accumulators of channels ncol0 + 16 nb + 4 q4 .. + 3 of token row m (lane (l15, q4) of a transposed 16 x 16 block).
// `old` = the residual values of the same pieces, loaded by the caller in batches (ph8_load_resid): one load -> use -> store round trip
// per row block would serialise eight memory latencies per tile (measured: 17 us of epilogue per tile at 8 prompts).
// Every access goes through a buffer descriptor whose range ends at row M: rows beyond M read zeros and drop their stores in
// hardware, so the code is straight-line -- no exec-masked branch around a store, and the compiler can COUNT the outstanding
// vector-memory operations (vmcnt) instead of draining them all whenever the next batch's loads are already in flight (round 4).
struct Ph8F32Epi {
    __amdgpu_buffer_rsrc_t rsC, rsXb, rsPart;
    int ldc, N;
    bool accumulate, prod;
    const float* gate;
    int gate_rows, gate_ld, M;
};
__device__ __forceinline__ Ph8F32Epi ph8_f32_epi(const GemmArgs& g) {
    Ph8F32Epi e;
    e.rsC = __builtin_amdgcn_make_buffer_rsrc((void*)g.C, 0, (int)((unsigned)g.M * (unsigned)g.ldc * 4u), 0x00020000);
    e.rsXb = __builtin_amdgcn_make_buffer_rsrc((void*)g.xb, 0, g.xb ? (int)((unsigned)g.M * (unsigned)g.N * 2u) : 0, 0x00020000);
    e.rsPart = __builtin_amdgcn_make_buffer_rsrc((void*)g.ln_part_out, 0, g.ln_part_out ? (int)((unsigned)g.M * (unsigned)(g.N >> 6) * 8u) : 0, 0x00020000);
    e.ldc = g.ldc; e.N = g.N; e.accumulate = g.accumulate != 0; e.prod = g.xb != nullptr;
    e.gate = g.gate; e.gate_rows = g.gate_rows; e.gate_ld = g.gate_ld; e.M = g.M;
    return e;
}
typedef unsigned int ph8_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int ph8_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ph8_load_resid(const Ph8F32Epi& e, f32x4_t (&old)[4], int m, int ncol0, int q4) {
    const int off = (m * e.ldc + ncol0 + 4 * q4) * 4;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
        old[nb] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(e.rsC, off + nb * 64, 0, 0));      // (always issued: a uniform branch
}                                                                                                                       //  here would hide the count)
// RT = true: gate / producer decided at run time (the adaLN path and the reduce kernel: uniform branches, conservative waits);
// RT = false: no gate, PROD compile-time -- the straight-line code the pipelined tile epilogue needs
template <bool RT, bool PROD_C>
__device__ __forceinline__ void ph8_epi_f32_row(const Ph8F32Epi& e, f32x4_t (&v)[4], const f32x4_t (&old)[4], int m, int ncol0, int q4) {
    const int off = (m * e.ldc + ncol0 + 4 * q4) * 4;
    [[maybe_unused]] const float* grow = nullptr;
    bool prod = PROD_C;
    if constexpr (RT) {
        prod = e.prod;
        if (e.gate) {
            const int mc = m < e.M ? m : e.M - 1;
            grow = e.gate + (size_t)(mc / e.gate_rows) * e.gate_ld + ncol0 + 4 * q4;
        }
    }
    float sum = 0.f, sq = 0.f;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        f32x4_t x = v[nb];          // (the bias is already in: the accumulators of a whole tile start from it, the reduce kernel adds it to its sums)
        if constexpr (RT) {
            if (grow) x *= *reinterpret_cast<const f32x4_t*>(grow + nb * 16);
        }
        const f32x4_t o = old[nb];
        x += e.accumulate ? o : f32x4_t{0.f, 0.f, 0.f, 0.f};          // (a select on the loaded value, not a branch around the load)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ph8_u32x4, x), e.rsC, off + nb * 64, 0, 0);
        if (prod) {
            opx4 xr;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                xr[k] = f32_to_op(x[k]);
                const float f = op_to_f32(xr[k]);
                sum += f;
                sq += f * f;
            }
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ph8_u32x2, xr), e.rsXb, (m * e.N + ncol0 + 4 * q4 + nb * 16) * 2, 0, 0);
        }
    }
    if (prod) {       // add the four lanes (q4 = 0..3) that share the token row
        sum += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(sum), 0x401F));       // lane ^ 16
        sq += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(sq), 0x401F));
        u32x2 a = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum), __float_as_uint(sum), false, false);
        u32x2 b = __builtin_amdgcn_permlane32_swap(__float_as_uint(sq), __float_as_uint(sq), false, false);
        sum = __uint_as_float(a[0]) + __uint_as_float(a[1]);
        sq = __uint_as_float(b[0]) + __uint_as_float(b[1]);
        // lanes q4 != 0 aim beyond the descriptor's range: their store is dropped (no branch)
        const int poff = q4 == 0 ? (m * (e.N >> 6) + (ncol0 >> 6)) * 8 : 0x7ffffff0;
        __builtin_amdgcn_raw_buffer_store_b64(ph8_u32x2{__float_as_uint(sum), __float_as_uint(sq)}, e.rsPart, poff, 0, 0);
    }
}

// DBG (tools/ph8_probe.py ablations, wrong results): 1 no LDS-DMA in the loop, 2 no ds_read (fragments stay), 3 no MFMA
// Measured and dropped (profiles/r03_ph8_schedule_options.txt): without the explicit lgkmcnt(0) behind the barrier, without s_setprio,
// with a static priority for waves 4-7 -- all within 1 %; LDS-DMA issued at the head of the MFMA section instead of the load
// section -- 5 % slower.
// DBG 9 (experiments build): per workgroup and K-range the 100 MHz timestamps (range start, main loop done, fix-up done, epilogue
// done) + (whole, K-tiles, last arriver) -> tools/ph8_probe.py timeline
// Tile geometry: 2 wave rows x WN wave columns, a wave = 2 quadrant rows of MFQ 16-row blocks x 64 columns.
//   WN = 4, MFQ = 4: 256 x 256, 8 waves, 136 KiB of LDS, one workgroup per CU   (the tile everything above describes)
//   WN = 2, MFQ = 2: 128 x 128, 4 waves,  68 KiB of LDS, two workgroups per CU  (experiment for the narrow one-prompt GEMMs: slower than
//                    the 16-wave-family tiles everywhere, see sat_launch_gemm_ph8; compiled in the experiments build only)
// FP8 = 2 (BASELINE config 5, the LayerNorm-fed GEMMs: to_qkv, cross to_q, FF-in): A and W hold e4m3 bytes; the launcher hands the kernel
// K / 2 "16-bit columns", so DMA, LDS image and swizzle are byte-for-byte those of the bf16 kernel -- a 128-byte LDS row now carries 128 k.
// One v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales, twice the bf16 MFMA rate) consumes a whole row of the K-tile: a lane's 32
// operand bytes are logical chunks 2 q and 2 q + 1 of its row, loaded identically for A and W, so the instruction's internal k order does
// not matter.  The fp32 accumulators are multiplied by a_scale[token] * w_scale[channel] in front of the epilogue.
// GATED (fp32 output only): the adaLN build of the residual epilogue -- (acc + bias) * gate[sequence] before the add
// (transformer.py:674, 688).  A kernel of its own: a second copy of the straight-line epilogue inside one kernel made the register
// allocator spill, a uniform branch inside it hides the vmcnt bookkeeping.
template <int EPI, int DBG = 0, bool PH2 = true, int PH2V = 1, int WN = 4, int MFQ = 4, int FP8 = 0, bool GATED = false>
__global__ __launch_bounds__(2 * WN * 64) void gemm_ph8_kernel(GemmArgs g, Ph8Sched sc, unsigned long long* ts = nullptr) {
    sat_f16_saturate();
    constexpr int NW = 2 * WN, NT = NW * 64;
    constexpr int QR = MFQ * 16;                 // rows of one quadrant of a wave
    constexpr int WR = 2 * QR;                   // rows of a wave
    constexpr int BM = 2 * WR, BN = WN * 64;
    constexpr int MB = 2 * MFQ;                  // 16-row blocks of a wave
    constexpr int AH = (BM / 2) * 128, WH = (BN / 2) * 128;          // bytes of an A / W half-tile
    static_assert(NT / 2 == BM, "two threads per row in the LayerNorm prologue");
    [[maybe_unused]] int ts_n = 0;
    constexpr int BUF_BYTES = 2 * (AH + WH), RING_BYTES = 2 * BUF_BYTES;
    auto koff = [](int kind) { return kind == 0 ? 0 : kind == 1 ? WH : kind == 2 ? WH + AH : 2 * WH + AH; };          // W-lo, A-lo, W-hi, A-hi
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid_ = threadIdx.x;
    const int lane = tid_ & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int l15_ = lane & 15, q4_ = lane >> 4;
    const int M = g.M, N = g.N, K = g.K;
    const int wgi = xcd_remap(blockIdx.x, sc.G);          // consecutive logical workgroups share an XCD (and so the tiles they split)

    // ---- the walk over this workgroup's K-ranges
    auto tile_of = [&](int id, int& tm, int& tn) { ph8_tile_of(sc, id, tm, tn); };
    int dp_s = 0;
    int sk_b = 0, sk_e = 0;
    if (sc.sk_tiles) ph8_wg_units(sc, wgi, sk_b, sk_e);
    int sk_u = sk_b;
    auto next_seg = [&](Seg& s) -> bool {
        int tm, tn;
        if (dp_s < sc.dp_rounds) {
            tile_of(dp_s * sc.G + wgi, tm, tn);
            ++dp_s;
            s.kt0 = 0; s.nk = 2 * sc.nkp; s.whole = true;
        } else {
            if (sk_u >= sk_e) return false;
            const int j = sk_u / sc.nkp;
            const int ub = sk_u - j * sc.nkp;
            const int ue = min(sc.nkp, ub + (sk_e - sk_u));
            tile_of(sc.rem0 + j, tm, tn);
            s.kt0 = 2 * ub; s.nk = 2 * (ue - ub); s.whole = (ub == 0 && ue == sc.nkp);
            sk_u += ue - ub;
        }
        s.m0 = tm * BM; s.n0 = tn * BN;
        // Accumulator orientation, uniform over the workgroup (a 256-column tile never straddles a q / k / v part): transposed
        // (lane = token) everywhere except for a V^T destination, whose token-contiguous stores want lane = channel.
        s.tr = true;
        if constexpr (EPI == EPI_HEADS) s.tr = !(g.heads.kind[s.n0 / (g.heads.heads * 64)] & 1);
        return true;
    };

    // ---- LDS-DMA sources.  One instruction of one wave fills 8 LDS rows (1 KiB); round i of wave w covers rows 64 i + 8 w + (lane >> 3).
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)((unsigned)M * (unsigned)K * 2u), 0x00020000);
    __amdgpu_buffer_rsrc_t rsW = rsA;
    int voffA[2], voffW[2][2];          // [round] (A: lo; hi = + 64 rows), [ni][round]
    auto setup_dma = [&](const Seg& s) {
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));
        rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(g.W + (size_t)s.n0 * K), 0, BN * K * 2, 0x00020000);
        const int sub = lane_l >> 3, pos = lane_l & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = i * (NW * 8) + wave * 8 + sub;             // LDS row of the half-tile
            const int c = pos ^ ((r >> 1) & 7);
            // A: LDS rows [0,64) belong to wave row 0, [64,128) to wave row 1; rows beyond M are out of range of rsA: zeros
            voffA[i] = (s.m0 + (r / QR) * WR + (r % QR)) * (K * 2) + c * 16;
            // W: LDS rows [32 w', 32 w' + 32) belong to wave column w'; row = 16 nf + fragment row
            const int wcol = r >> 5, nf = (r >> 4) & 1, fi = r & 15;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                int ch;
                if constexpr (EPI == EPI_F32) ch = chan_of<0>(ni, nf, fi);
                else if constexpr (EPI == EPI_SWIGLU) ch = chan_of<1>(ni, nf, fi);
                else ch = s.tr ? chan_of<2>(ni, nf, fi) : chan_of<0>(ni, nf, fi);
                voffW[ni][i] = (wcol * 64 + ch) * (K * 2) + c * 16;
            }
        }
    };
    const int hiA = QR * K * 2;
    auto issue = [&](int kind, int buf, int kt) {          // kind: 0 W-lo, 1 A-lo, 2 W-hi, 3 A-hi (compile-time after inlining)
        char* dst = smem + buf * BUF_BYTES + koff(kind) + wave * 1024;
        const int soff = __builtin_amdgcn_readfirstlane(kt * 128);      // (stays scalar even if the K-tile counter was spilled to a VGPR lane)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (kind & 1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_p)(dst + i * (NW * 1024)), 16, voffA[i] + (kind == 3 ? hiA : 0), soff, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void_p)(dst + i * (NW * 1024)), 16, voffW[kind >> 1][i], soff, 0, 0);
        }
    };

    // ---- fragment addresses: row = (wave's first row of the half-tile) + 16 f + l15, chunk (4 ks + q4) ^ (l15 >> 1)
    const int swz = l15_ >> 1;
    [[maybe_unused]] const int swz_ = swz;
    int offA[2], offW[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ch = ((ks * 4 + q4_) ^ swz) << 4;
        offA[ks] = (wr * QR + l15_) * 128 + ch;
        offW[ks] = (wc * 32 + l15_) * 128 + ch;
    }

    f32x4_t acc[MB][4];
    opx8 fa[MFQ][2], fwl[2][2], fwh[2][2];
    typedef int i32x8_t __attribute__((ext_vector_type(8)));
    [[maybe_unused]] i32x8_t fa8[MFQ], fw8l[2], fw8h[2];          // FP8: one 32-byte fragment per 16-row block and K-tile
    // logical chunks 2 q4 and 2 q4 + 1 of the lane's row (the XOR swizzle may swap their physical order: put them back)
    [[maybe_unused]] const int off8[2] = {(((2 * q4_) ^ swz_) << 4), (((2 * q4_ + 1) ^ swz_) << 4)};
    auto read8 = [&](const char* rowbase) -> i32x8_t {
        const u32x4 lo = *reinterpret_cast<const u32x4*>(rowbase + off8[0]);
        const u32x4 hi = *reinterpret_cast<const u32x4*>(rowbase + off8[1]);
        return i32x8_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    };
    [[maybe_unused]] auto read_a8 = [&](int buf, int hi) {
        const char* base = smem + buf * BUF_BYTES + koff(hi ? 3 : 1) + (wr * QR + l15_) * 128;
#pragma unroll
        for (int f = 0; f < MFQ; ++f) fa8[f] = read8(base + f * 2048);
    };
    [[maybe_unused]] auto read_w8 = [&](int buf, int hi, i32x8_t (&fw)[2]) {
        const char* base = smem + buf * BUF_BYTES + koff(hi ? 2 : 0) + (wc * 32 + l15_) * 128;
#pragma unroll
        for (int f = 0; f < 2; ++f) fw[f] = read8(base + f * 2048);
    };
    auto read_a = [&](int buf, int hi) {
        const char* base = smem + buf * BUF_BYTES + koff(hi ? 3 : 1);
#pragma unroll
        for (int f = 0; f < MFQ; ++f)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[f][ks] = *reinterpret_cast<const opx8*>(base + offA[ks] + f * 2048);
    };
    auto read_w = [&](int buf, int hi, opx8 (&fw)[2][2]) {
        const char* base = smem + buf * BUF_BYTES + koff(hi ? 2 : 0);
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fw[f][ks] = *reinterpret_cast<const opx8*>(base + offW[ks] + f * 2048);
    };

    // SWAP: W fragment as the MFMA A operand -> a 16 x 16 block holds C^T (lane = token l15, registers = channels 4 q4 + r).
    // K-tiles kt0 .. kt0 + nk - 1 (nk even); on entry half-tiles 0..6 of the range are in flight or landed and tile kt0 is visible.
    // Quadrant mi of a wave is skipped when its 64 rows lie beyond M (M-tail tiles: the wave keeps staging and joining barriers).
    auto main_loop = [&](auto swap_c, const int kt0, const int nk, const bool q_valid0, const bool q_valid1) {
        constexpr bool SWAP = decltype(swap_c)::value;
        auto mfma_quadrant = [&](int mi, int ni, opx8 (&fw)[2][2]) {
            if constexpr (DBG == 3) return;
            if (!(mi ? q_valid1 : q_valid0)) return;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int f = 0; f < MFQ; ++f)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[mi * MFQ + f][ni * 2 + n] = SWAP ? mfma_16x16x32(fw[n][ks], fa[f][ks], acc[mi * MFQ + f][ni * 2 + n])
                                                             : mfma_16x16x32(fa[f][ks], fw[n][ks], acc[mi * MFQ + f][ni * 2 + n]);
            __builtin_amdgcn_s_setprio(0);
        };
        // One K-tile = four phases.  BUF is the LDS buffer of tile t; tile t + 2 restages the same buffer.
        // ISSUE: 4 = issue all four half-tiles (steady state), 1 = only phase 0's (the range's tile nk - 2), 0 = none (tile nk - 1)
        auto k_tile = [&](auto buf_c, auto issue_c, auto wait_c, int t) {
            constexpr int BUF = decltype(buf_c)::value;
            constexpr int ISSUE = (DBG == 1) ? 0 : decltype(issue_c)::value;
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
        if (wr == 1) __builtin_amdgcn_s_barrier();          // stagger: wave row 1 runs one barrier behind wave row 0
        const int kt_last = kt0 + nk - 2;
        for (int t = kt0; t < kt_last; t += 2) {
            k_tile(I0{}, I4{}, I6{}, t);
            k_tile(I1{}, I4{}, I6{}, t + 1);
        }
        k_tile(I0{}, I1{}, I0{}, kt_last);
        k_tile(I1{}, I0{}, IM{}, kt_last + 1);
        if (wr == 0) __builtin_amdgcn_s_barrier();          // re-align the two wave rows: every wave is done with the ring
    };

    // PH2 (experiment): the same K-tile in TWO phases of 32 MFMAs -- 4 barriers per K-tile instead of 8.
    //   phase A: read W-lo, W-hi (4 + 4) and A-lo (8); issue W-hi, A-hi of tile t + 1; vmcnt(8): A-hi of tile t has landed
    //            -> A-lo x W-lo, A-lo x W-hi
    //   phase B: read A-hi (8); issue W-lo, A-lo of tile t + 2; vmcnt(6): W-lo, A-lo, W-hi of tile t + 1 have landed
    //            -> A-hi x W-hi, A-hi x W-lo
    // every load section retires its ds_reads BEFORE its barrier, so a slot may be restaged by the partner wave row in the very next
    // interval; a landed half-tile is first read one barrier after the wait that covers it.
    auto main_loop2 = [&](auto swap_c, const int kt0, const int nk, const bool q_valid0, const bool q_valid1) {
        constexpr bool SWAP = decltype(swap_c)::value;
        auto mfma_half = [&](int mi, int ni_first) {
            if (!(mi ? q_valid1 : q_valid0)) return;
            __builtin_amdgcn_s_setprio(1);
            if constexpr (FP8 != 0) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int ni = ni_first ^ h;
                    i32x8_t (&fw)[2] = ni ? fw8h : fw8l;
#pragma unroll
                    for (int f = 0; f < MFQ; ++f)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[mi * MFQ + f][ni * 2 + n] =
                                SWAP ? __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw[n], fa8[f], acc[mi * MFQ + f][ni * 2 + n], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F)
                                     : __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa8[f], fw[n], acc[mi * MFQ + f][ni * 2 + n], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                }
                __builtin_amdgcn_s_setprio(0);
                return;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ni = ni_first ^ h;
                opx8 (&fw)[2][2] = ni ? fwh : fwl;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int f = 0; f < MFQ; ++f)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[mi * MFQ + f][ni * 2 + n] = SWAP ? mfma_16x16x32(fw[n][ks], fa[f][ks], acc[mi * MFQ + f][ni * 2 + n])
                                                                 : mfma_16x16x32(fa[f][ks], fw[n][ks], acc[mi * MFQ + f][ni * 2 + n]);
            }
            __builtin_amdgcn_s_setprio(0);
        };
        // MODE 0: steady state (tiles t + 1 and t + 2 exist); 1: tile t + 1 is the range's last; 2: tile t is the last
        // EARLY_WHI: W-hi of tile t + 2 is issued in phase B of tile t together with W-lo and A-lo (its slot is free since phase A)
        // instead of phase A of tile t + 1: one phase more of flight time, and the DMA issues balance the ds_reads (A: 16 + 2, B: 8 + 6)
        constexpr bool EARLY_WHI = (PH2V == 2);
        auto k_tile2 = [&](auto buf_c, auto mode_c, int t) {
            constexpr int BUF = decltype(buf_c)::value;
            constexpr int MODE = decltype(mode_c)::value;
            // ---- phase A
            if constexpr (FP8 != 0) {
                read_w8(BUF, 0, fw8l);
                read_w8(BUF, 1, fw8h);
                read_a8(BUF, 0);
            } else {
                read_w(BUF, 0, fwl);
                read_w(BUF, 1, fwh);
                read_a(BUF, 0);
            }
            if constexpr (MODE <= 1) {
                if constexpr (!EARLY_WHI) issue(2, BUF ^ 1, t + 1);
                issue(3, BUF ^ 1, t + 1);
                wait_vmcnt<8>();
            } else {
                wait_vmcnt<0>();
            }
            __builtin_amdgcn_sched_barrier(0);
            wait_lgkmcnt<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            mfma_half(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- phase B
            if constexpr (FP8 != 0) read_a8(BUF, 1);
            else read_a(BUF, 1);
            if constexpr (MODE == 0) {
                issue(0, BUF, t + 2);
                issue(1, BUF, t + 2);
                if constexpr (EARLY_WHI) issue(2, BUF, t + 2);
                wait_vmcnt<EARLY_WHI ? 8 : 6>();
            } else if constexpr (MODE == 1) {
                wait_vmcnt<2>();
            }
            __builtin_amdgcn_sched_barrier(0);
            wait_lgkmcnt<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            mfma_half(1, 1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        if (wr == 1) __builtin_amdgcn_s_barrier();
        const int kt_last = kt0 + nk - 2;
        for (int t = kt0; t < kt_last; t += 2) {
            k_tile2(I0{}, I0{}, t);
            k_tile2(I1{}, I0{}, t + 1);
        }
        k_tile2(I0{}, I1{}, kt_last);
        k_tile2(I1{}, I2{}, kt_last + 1);
        if (wr == 0) __builtin_amdgcn_s_barrier();
    };

    // ---- LayerNorm fold, consumer side (GemmArgs): (mean, 1/std) of the tile's 256 rows from the producer's per-64-column partial
    // sums and the tile's 256 (c1, c2) channel constants go to LDS behind the ring (two buffers: the next K-range's are written
    // while the current epilogue still reads its own).  Without the fold the same epilogues run on (0, 1), 0, bias.
    // The loads are ISSUED in front of the range's LDS-DMA prologue and consumed behind it: the vector-memory counter retires in
    // order, so a load issued after the 14 DMA pieces could only be used once all of them have landed.
    constexpr bool LN_CONS = (EPI == EPI_SWIGLU || EPI == EPI_HEADS);
    constexpr int LN_BYTES = (BM + BN) * 8;          // (mean, rstd) per row, then c1[BN], c2[BN]
    const bool ln_fold = LN_CONS && g.ln_part != nullptr;
    const int np = K >> 6;
    const bool ln_fast = ln_fold && np == 24;             // 12 partial pairs per thread, held in registers across the DMA issue
    // bf16: rstd (acc - mean c1) + c2.  e4m3 (no fold): the same slots carry the dequantisation -- st = (-, a_scale[row]),
    // c1 = w_scale[channel], c2 = bias -- and the epilogues compute a_scale w_scale acc + bias
    auto fold = [](float a, float mean, float rstd, float c1, float c2) {
        if constexpr (FP8 != 0) return fmaf(a * rstd, c1, c2);
        else return rstd * (a - mean * c1) + c2;
    };
    // Everything a K-range needs before its main loop, in two steps.  ISSUE: the loads of the LayerNorm partial sums / channel constants, the
    // DMA addresses, half-tiles 0..6 in flight.  FINISH: (mean, 1 / std) and the constants into LDS.  For the FIRST range of a workgroup the
    // two run back to back; inside the persistent loop ISSUE runs in front of the current range's epilogue and -- LN_DEFER -- FINISH behind it,
    // at the top of the next range: the round trip of the partial-sum loads (1.2-1.5 us of the 1.7 us "prepare-next" that was left after the
    // LDS accesses went out of the compiler's sight, profiles/r05_ph8_timeline.txt) then hides behind the epilogue instead of in front of it.
    // The loaded values (28 registers) stay live across the epilogue: affordable next to the SwiGLU epilogue, not next to the heads one.
    struct LnPre {
        float2 lnp[12];
        f32x4_t lncst;
        float a_sc;
    };
    constexpr bool LN_DEFER = LN_CONS && EPI == EPI_SWIGLU;
    // Half-tiles of a range's DMA prologue: all four of its first K-tile + PRO_T1 of the second.  ONE constant drives the issue loop below and the
    // counted waits that let exactly these stay in flight (PRO_DMA; ADVICE r5: the two used to spell the expression out separately)
    constexpr int PRO_T1 = (PH2 && PH2V != 2) ? 2 : 3;
    auto prepare_issue = [&](const Seg& s, [[maybe_unused]] LnPre& st) {
        int tid = tid_;
        asm volatile("" : "+v"(tid));            // (keeps this block's address arithmetic inside the persistent loop, see the epilogue)
        [[maybe_unused]] const int ct = NT - 1 - tid;        // the last BN / 2 threads bring in the channel constants, 16 bytes each
        if constexpr (LN_CONS) {
            st.lncst = f32x4_t{0.f, 0.f, 0.f, 0.f};
            st.a_sc = 1.f;
            if (ln_fast) {
                int m = s.m0 + (tid >> 1);
                m = m < M ? m : M - 1;
                const float2* pp = reinterpret_cast<const float2*>(g.ln_part) + (size_t)m * np + (tid & 1);
#pragma unroll
                for (int i = 0; i < 12; ++i) st.lnp[i] = pp[2 * i];
            } else {
#pragma unroll
                for (int i = 0; i < 12; ++i) st.lnp[i] = make_float2(0.f, 0.f);
            }
            if (ct < BN / 2) {
                const bool first = ct < BN / 4;
                const float* src = ln_fold ? (first ? g.ln_c1 + s.n0 + ct * 4 : g.ln_c2 + s.n0 + (ct - BN / 4) * 4)
                                           : (first ? (FP8 != 0 ? g.w_scale + s.n0 + ct * 4 : nullptr)
                                                    : (g.bias ? g.bias + s.n0 + (ct - BN / 4) * 4 : nullptr));
                if (src) st.lncst = *reinterpret_cast<const f32x4_t*>(src);
            }
            if constexpr (FP8 != 0) {
                const int m = s.m0 + (tid >> 1);
                st.a_sc = g.a_scale[m < M ? m : M - 1];
            }
        }
        setup_dma(s);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) issue(j, 0, s.kt0);
#pragma unroll
        for (int j = 0; j < PRO_T1; ++j) issue(j, 1, s.kt0 + 1);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto prepare_finish = [&](const Seg& s, int lb, [[maybe_unused]] LnPre& st) {
        if constexpr (LN_CONS) {
            int tid = tid_;
            asm volatile("" : "+v"(tid));
            const int ct = NT - 1 - tid;
            float2* lnst = reinterpret_cast<float2*>(smem + RING_BYTES + lb * LN_BYTES);
            float* lnc = reinterpret_cast<float*>(smem + RING_BYTES + lb * LN_BYTES + BM * 8);
            const int r = tid >> 1, sub = tid & 1;           // two threads per row
            float sum = 0.f, sq = 0.f;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                sum += st.lnp[i].x;
                sq += st.lnp[i].y;
            }
            if (ln_fold && !ln_fast) {                       // any other K: plain loop (behind the DMA pieces in the memory queue)
                int m = s.m0 + r;
                m = m < M ? m : M - 1;
                const float2* pp = reinterpret_cast<const float2*>(g.ln_part) + (size_t)m * np;
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
                lds_st64(lnst + r, ln_fold ? make_float2(mean, rsqrtf(var + g.ln_eps)) : make_float2(0.f, st.a_sc));
            }
            if (ct < BN / 2) lds_st128(lnc + ct * 4, st.lncst);
        }
    };

    // ---- EPI_HEADS, q / k tiles: the rotation (transformer.py:158-183).  Loaded per row block inside the epilogue, the table rows were sixteen
    // DEPENDENT round trips per tile -- the compiler put an s_waitcnt vmcnt(0) behind every one of them: 8.5 us of epilogue per q / k tile
    // against 2.8 for SwiGLU (profiles/r04_ph8_heads_swiglu_timeline.txt).  Now a lane fetches the (cos, sin) of its FIRST row block and of
    // position 16 right after the main loop, in FRONT of the next range's LDS-DMA prologue (oldest entries of the in-order vector-memory
    // queue; inline assembly, so the compiler's conservative wait insertion neither sees nor serialises them), waits for them with ONE counted
    // wait behind it, and walks to the next row block -- 16 positions on -- by the angle-addition rotation (16 FMAs).  Against exact arithmetic the
    // walked values are as accurate as the table (1.5e-5 vs 2.7e-5 at S = 1025); they differ from the TABLE -- the reference's fp32 pos * inv_freq --
    // by its own angle rounding: <= 4e-5 (rms 3e-6) at S = 1025, <= 2.4e-4 (rms 1.5e-5) at S = 6145, against 16-bit rounding of q / k of 1.4e-4 rms
    // (tests/test_host_logic.py::test_rope_angle_addition_recurrence_error).  A row block in which some lane crosses into the next sequence re-reads the table.
    // The (cos, sin) of position 16 -- 32 floats, the same for every tile -- sit in LDS behind the LayerNorm constants, written once per workgroup.
    struct RopePre {
        f32x4_t cs, sn;
    };
    constexpr int ROPE16_OFF = RING_BYTES + 2 * LN_BYTES;          // cos[16], sin[16] of position 16 (min(16, S - 1))
    if constexpr (EPI == EPI_HEADS) {
        if (tid_ < 32 && g.heads.rope_cos) {
            const int r16 = g.heads.S > 16 ? 16 : g.heads.S - 1;
            reinterpret_cast<float*>(smem + ROPE16_OFF)[tid_] = (tid_ < 16 ? g.heads.rope_cos : g.heads.rope_sin)[r16 * 16 + (tid_ & 15)];
        }          // (published by the barriers of the first main loop)
    }
    constexpr int PRO_DMA = 2 * (4 + PRO_T1);          // LDS-DMA instructions per wave of a range's prologue (issue(): two per half-tile)
    static_assert(PRO_DMA == 12 || PRO_DMA == 14, "prologue = 6 or 7 half-tiles");
    [[maybe_unused]] auto rope_load = [&](f32x4_t& cs, f32x4_t& sn, int row, int q4) {
        const unsigned voff = (unsigned)(row * 16 + 4 * q4) * 4u;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(cs) : "v"(voff), "s"(g.heads.rope_cos) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(sn) : "v"(voff), "s"(g.heads.rope_sin) : "memory");
    };
    [[maybe_unused]] auto rope_prefetch = [&](const Seg& s, RopePre& rp) {
        int l15 = l15_, q4 = q4_;
        asm volatile("" : "+v"(l15), "+v"(q4));
        const int S = g.heads.S;
        const int m = s.m0 + wr * WR + l15;
        rope_load(rp.cs, rp.sn, (m < M ? m : M - 1) % S, q4);
    };
    [[maybe_unused]] auto rope_wait = [](auto n_c, RopePre& rp) {
        wait_vmcnt<decltype(n_c)::value>();
        asm volatile("" : "+v"(rp.cs), "+v"(rp.sn));          // every use sits behind the wait
    };

    // ---- epilogues.  Transposed: lane (l15, q4) holds, for block (mb = 0..7, nb = 2 ni + nf): token row m0 + wr*128 + mb*16 + l15,
    //      channels n0 + wc*64 + chan_of(ni, nf, 4 q4 + r), r = 0..3
    auto epilogue = [&](const Seg& s, int lb, [[maybe_unused]] RopePre& rp, [[maybe_unused]] const bool dma_behind) {
        // (laundered copies: keeps the segment-invariant address arithmetic of the epilogue from being hoisted out of the persistent
        // loop, where it would stay live across the main loop and push its 128 + 64 registers into scratch)
        int l15 = l15_, q4 = q4_;
        asm volatile("" : "+v"(l15), "+v"(q4));
        const int mrow0 = s.m0 + wr * WR + l15;
        const int ncol0 = s.n0 + wc * 64;
        [[maybe_unused]] const float2* ln = reinterpret_cast<const float2*>(smem + RING_BYTES + lb * LN_BYTES) + wr * WR;
        [[maybe_unused]] const float* lc1 = reinterpret_cast<const float*>(smem + RING_BYTES + lb * LN_BYTES + BM * 8) + wc * 64;
        [[maybe_unused]] const float* lc2 = lc1 + BN;
        if constexpr (EPI == EPI_F32) {
            // Batches of two row blocks, double-buffered: the residual loads of batch h + 1 are issued before batch h is finished, so 16
            // loads per lane are in flight instead of 8 (the fragment registers are dead here).  The read-modify-write of X is latency-
            // bound per CU -- 640 KB per tile in four dependent round trips took 17 us of a 54-us tile at 8 prompts
            // (profiles/r03_ph8_ksweep_fixed_overhead.txt) -- and every load in flight is time off that.
            const Ph8F32Epi fe = ph8_f32_epi(g);
            auto run = [&](auto rt_c, auto prod_c) {
                constexpr bool RT = decltype(rt_c)::value, PROD = decltype(prod_c)::value;
                f32x4_t old[2][2][4];
#pragma unroll
                for (int i = 0; i < 2; ++i) ph8_load_resid(fe, old[0][i], mrow0 + i * 16, ncol0, q4);
#pragma unroll
                for (int h = 0; h < MB / 2; ++h) {
                    if (h + 1 < MB / 2) {
#pragma unroll
                        for (int i = 0; i < 2; ++i) ph8_load_resid(fe, old[(h + 1) & 1][i], mrow0 + ((h + 1) * 2 + i) * 16, ncol0, q4);
                    }
                    __builtin_amdgcn_sched_barrier(0);          // (two batches in flight, not four: straight-line code lets the scheduler hoist every
#pragma unroll                                                  //  load to the top, which is 128 registers of residual next to 128 accumulators)
                    for (int i = 0; i < 2; ++i) ph8_epi_f32_row<RT, PROD>(fe, acc[h * 2 + i], old[h & 1][i], mrow0 + (h * 2 + i) * 16, ncol0, q4);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            // (compile-time variants of the common cases: a wave-uniform branch inside the batch loop hides the number of outstanding
            // stores from the compiler's vmcnt bookkeeping and brings the drain-everything waits back.  adaLN's gated update keeps them.)
            // Without a producer role (no xb / ln_part_out) the same code runs: those descriptors then have an empty range and the
            // hardware drops the stores -- a second straight-line copy next to this one made the register allocator spill.
            if constexpr (GATED) run(std::true_type{}, std::false_type{});
            else run(std::false_type{}, std::true_type{});
        } else if constexpr (EPI == EPI_SWIGLU) {
            // H = (v + b_v) * silu(gate + b_g) (transformer.py:232-235): value rows are channels [0, 32) of the wave's 64, gate rows
            // [32, 64) (pack_rows interleave); PERM 1 puts value and gate of hidden columns hc0 + 8 q4 + 4 nf + r into this lane
            const int ldh = N >> 1;
            f32x4_t c1v[2], c1g[2], c2v[2], c2g[2];
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) {
                c1v[nf] = lds_ld128(lc1 + q4 * 8 + nf * 4);
                c1g[nf] = lds_ld128(lc1 + 32 + q4 * 8 + nf * 4);
                c2v[nf] = lds_ld128(lc2 + q4 * 8 + nf * 4);
                c2g[nf] = lds_ld128(lc2 + 32 + q4 * 8 + nf * 4);
            }
            float2 stv[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) stv[mb] = lds_ld64(ln + mb * 16 + l15);
            lds_wait(c1v[0], c1v[1], c1g[0], c1g[1], c2v[0], c2v[1], c2g[0], c2g[1]);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) lds_tie(stv[mb]);
            // The 16-bit output goes through a buffer descriptor whose range ends at row M: rows beyond it are dropped by the hardware, the
            // store is UNCONDITIONAL -- so a wave issues exactly MB vector-memory operations behind the next range's DMA prologue, and the
            // wait at the top of that range can be counted (EPI_STORES): it no longer waits for these stores to be acknowledged.
            const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc((void*)g.H, 0, (int)((unsigned)M * (unsigned)ldh * 2u), 0x00020000);
            const int hcol = (ncol0 >> 1) + q4 * 8;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int m = mrow0 + mb * 16;
                const float2 st = stv[mb];
                unsigned pk[4];
                [[maybe_unused]] float hv8[8];
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) {
                    float hv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = fold(acc[mb][nf][e], st.x, st.y, c1v[nf][e], c2v[nf][e]);
                        const float gt = fold(acc[mb][2 + nf][e], st.x, st.y, c1g[nf][e], c2g[nf][e]);
                        hv[e] = v * silu_fast(gt);
                        if constexpr (FP8 != 0) hv8[4 * nf + e] = hv[e];
                    }
                    pk[2 * nf] = pack_op2(hv[0], hv[1]);
                    pk[2 * nf + 1] = pack_op2(hv[2], hv[3]);
                }
                if constexpr (FP8 != 0) {
                    if (g.H8) {
                        // MXFP8 (the A operand of FF-out): the wave's 32 hidden columns are ONE block of token row m, 8 values in each of the
                        // four lanes q4 = 0..3; scale = 2^e, e = ceil(log2(amax / 448)), stored as E8M0 = e + 127
                        float am = 0.f;
#pragma unroll
                        for (int i = 0; i < 8; ++i) am = fmaxf(am, fabsf(hv8[i]));
                        am = m < M ? am : 0.f;
                        am = fmaxf(am, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(am), 0x401F)));              // lane ^ 16
                        am = fmaxf(am, __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __float_as_int(am))));   // lane ^ 32
                        const float t = am * (1.0f / 448.0f);
                        const unsigned tb = __float_as_uint(t);
                        int e = (int)((tb >> 23) & 0xff) - 127 + ((tb & 0x7fffff) ? 1 : 0);
                        e = am > 0.f ? (e < -127 ? -127 : (e > 127 ? 127 : e)) : -127;
                        const float inv = __uint_as_float((unsigned)(127 - e) << 23);
                        unsigned q0 = __builtin_amdgcn_cvt_pk_fp8_f32(hv8[0] * inv, hv8[1] * inv, 0u, false);
                        q0 = __builtin_amdgcn_cvt_pk_fp8_f32(hv8[2] * inv, hv8[3] * inv, q0, true);
                        unsigned q1 = __builtin_amdgcn_cvt_pk_fp8_f32(hv8[4] * inv, hv8[5] * inv, 0u, false);
                        q1 = __builtin_amdgcn_cvt_pk_fp8_f32(hv8[6] * inv, hv8[7] * inv, q1, true);
                        if (m < M) {
                            *reinterpret_cast<u32x2*>(g.H8 + (size_t)m * ldh + (ncol0 >> 1) + q4 * 8) = u32x2{q0, q1};
                            if (q4 == 0) g.Hs[(size_t)m * (ldh >> 5) + (ncol0 >> 6)] = (unsigned char)(e + 127);
                        }
                        continue;
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b128(ph8_u32x4{pk[0], pk[1], pk[2], pk[3]}, rsH, (m * ldh + hcol) * 2, 0, 0);
            }
        } else {   // EPI_HEADS: split into heads, LayerNorm fold, partial RoPE on d < 32 (transformer.py:158-183, 438-452)
            const HeadsEpi& he = g.heads;
            const int hp = he.heads * 64;
            const int part = ncol0 / hp;
            const int head = (ncol0 - part * hp) >> 6;
            const int kind = he.kind[part];
            op_t* __restrict__ dst = he.out[part];
            const int S = he.S, Spad = he.Spad;
            if (s.tr) {
                // q / k, row-major [B, H, Spad, 64].  PERM 2: block ni = 0 holds d = 16 nf + 4 q4 + r (the rotation partner d + 16 is
                // block nf + 1 of the same lane), block ni = 1 holds d = 32 + 8 q4 + 4 nf + r (8 consecutive channels)
                f32x4_t c1[4], c2[4];
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) {
                    c1[nf] = lds_ld128(lc1 + nf * 16 + 4 * q4);
                    c2[nf] = lds_ld128(lc2 + nf * 16 + 4 * q4);
                    c1[2 + nf] = lds_ld128(lc1 + 32 + q4 * 8 + nf * 4);
                    c2[2 + nf] = lds_ld128(lc2 + 32 + q4 * 8 + nf * 4);
                }
                float2 stv[2];          // (mean, rstd) of the lane's row: read one row block ahead
                stv[0] = lds_ld64(ln + l15);
                [[maybe_unused]] f32x4_t c16 = lds_ld128(smem + ROPE16_OFF + 16 * q4), s16 = lds_ld128(smem + ROPE16_OFF + 64 + 16 * q4);
                lds_wait(c1[0], c1[1], c1[2], c1[3], c2[0], c2[1], c2[2], c2[3], c16, s16, stv[0]);
                // (sequence, position) of the lane's eight rows: one division, then steps of 16 rows (rows beyond M keep walking: they are never
                // stored, and their table index stays inside [0, S))
                int b = 0, sq_ = 0;
                {
                    const int mc = mrow0 < M ? mrow0 : M - 1;
                    b = mc / S;
                    sq_ = mc - b * S;
                }
                if (kind & 2) {
                    // (requested in front of the next range's prologue: this wait does not hold the epilogue until its `dma_behind` pieces land)
                    if (dma_behind) rope_wait(std::integral_constant<int, PRO_DMA>{}, rp);
                    else rope_wait(std::integral_constant<int, 0>{}, rp);
                }
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const int m = mrow0 + mb * 16;
                    const int ob = (kind & 4) ? ((b * S) & 3) : 0;
                    if (mb) lds_wait(stv[mb & 1]);
                    const float2 st = stv[mb & 1];
                    if (mb + 1 < MB) stv[(mb + 1) & 1] = lds_ld64(ln + (mb + 1) * 16 + l15);
                    f32x4_t x[4];
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[nb][e] = fold(acc[mb][nb][e], st.x, st.y, c1[nb][e], c2[nb][e]);
                    if (kind & 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float x1 = x[0][e], x2 = x[1][e];
                            x[0][e] = x1 * rp.cs[e] - x2 * rp.sn[e];
                            x[1][e] = x2 * rp.cs[e] + x1 * rp.sn[e];
                        }
                    }
                    if (kind & 8) {   // query: pre-scaled for the attention kernel (one rounding, here)
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb) x[nb] *= he.qscale;
                    }
                    if (m < M) {
                        op_t* row = dst + ((size_t)(b * he.heads + head) * Spad + sq_ + ob) * 64;
                        *reinterpret_cast<u32x2*>(row + 4 * q4) = u32x2{pack_op2(x[0][0], x[0][1]), pack_op2(x[0][2], x[0][3])};
                        *reinterpret_cast<u32x2*>(row + 16 + 4 * q4) = u32x2{pack_op2(x[1][0], x[1][1]), pack_op2(x[1][2], x[1][3])};
                        *reinterpret_cast<u32x4*>(row + 32 + 8 * q4) = u32x4{pack_op2(x[2][0], x[2][1]), pack_op2(x[2][2], x[2][3]),
                                                                             pack_op2(x[3][0], x[3][1]), pack_op2(x[3][2], x[3][3])};
                    }
                    sq_ += 16;
                    while (sq_ >= S) {
                        sq_ -= S;
                        ++b;
                    }
                    if ((kind & 2) && mb + 1 < MB) {
                        // the next row block is 16 positions on: rotate (cos, sin) by the angle of position 16 -- unless a lane of the wave just
                        // crossed into the next sequence (its position fell below 16), then the wave re-reads the table (rare: one row block per sequence)
                        if (__builtin_amdgcn_ballot_w64(sq_ < 16) != 0 || S <= 16) {
                            rope_load(rp.cs, rp.sn, sq_, q4);
                            rope_wait(std::integral_constant<int, 0>{}, rp);
                        } else {
                            const f32x4_t c = rp.cs, sn_ = rp.sn;
                            rp.cs = c * c16 - sn_ * s16;
                            rp.sn = sn_ * c16 + c * s16;
                        }
                    }
                }
            } else {
                // V^T [B, H, 64, Spad] (no rotation): un-swapped accumulators, lane = channel d = 16 nb + l15, registers = the four
                // consecutive token rows mb*16 + 4 q4 + e.  Columns of sequence b are shifted by (b S) & 3 (kind bit 2) so that those
                // four tokens are an 8-byte aligned group of vt_pos order (sat_common.h)
                const bool shift = (kind & 4) != 0;
                float c1[4], c2[4];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    c1[nb] = lds_ld32(lc1 + nb * 16 + l15);
                    c2[nb] = lds_ld32(lc2 + nb * 16 + l15);
                }
                // (mean, rstd) of the lane's four rows of a row block: read one row block ahead (two register sets)
                f32x4_t stq[2][2];
                stq[0][0] = lds_ld128(ln + 4 * q4);
                stq[0][1] = lds_ld128(ln + 4 * q4 + 2);
                lds_wait(c1[0], c1[1], c1[2], c1[3], c2[0], c2[1], c2[2], c2[3], stq[0][0], stq[0][1]);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const int mbase = s.m0 + wr * WR + mb * 16 + 4 * q4;          // multiple of 4
                    if (mb) lds_wait(stq[mb & 1][0], stq[mb & 1][1]);
                    const f32x4_t st01 = stq[mb & 1][0], st23 = stq[mb & 1][1];                    // (mean, rstd) of rows e = 0, 1 / 2, 3
                    if (mb + 1 < MB) {
                        stq[(mb + 1) & 1][0] = lds_ld128(ln + (mb + 1) * 16 + 4 * q4);
                        stq[(mb + 1) & 1][1] = lds_ld128(ln + (mb + 1) * 16 + 4 * q4 + 2);
                    }
                    int bb[4], ss[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int mm = mbase + e;
                        mm = mm < M ? mm : M - 1;
                        bb[e] = mm / S;
                        ss[e] = mm - bb[e] * S;
                    }
                    const bool whole4 = mbase + 3 < M && bb[0] == bb[3];
                    const int ob0 = shift ? ((bb[0] * S) & 3) : 0;
                    const size_t hb0 = ((size_t)(bb[0] * he.heads + head) * 64) * Spad;
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float mean = e < 2 ? st01[2 * e] : st23[2 * e - 4];
                            const float rstd = e < 2 ? st01[2 * e + 1] : st23[2 * e - 3];
                            v[e] = fold(acc[mb][nb][e], mean, rstd, c1[nb], c2[nb]);
                        }
                        const size_t drow = (size_t)(nb * 16 + l15) * Spad;
                        if (whole4 && shift) {         // aligned: (ss[0] + ob) % 4 == mbase % 4 == 0
                            *reinterpret_cast<u32x2*>(dst + hb0 + vt_pos(ss[0] + ob0) + drow) = u32x2{pack_op2(v[0], v[1]), pack_op2(v[2], v[3])};
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (mbase + e < M) {
                                    const int ob = shift ? ((bb[e] * S) & 3) : 0;
                                    dst[((size_t)(bb[e] * he.heads + head) * 64) * Spad + vt_pos(ss[e] + ob) + drow] = f32_to_op(v[e]);
                                }
                        }
                    }
                }
            }
        }
    };

    // ---- the persistent loop
    Seg cur, nxt;
    if (!next_seg(cur)) return;
    // (Starting the workgroups (i & 3) * {2, 4, 6} us apart, so that the fp32 epilogues of a round do not hit the memory system at the
    // same moment, only adds the delay: profiles/r03_ph8_stagger_negative.txt -- the 17-us epilogue of a 256 x 256 fp32 tile is bound
    // per CU, not by the sum.)
    int lb = 0;
    // vector-memory operations a wave issues between a range's DMA prologue and the top of that range, when they are a constant: the MB
    // unconditional stores of the SwiGLU epilogue (16-bit output; the e4m3 build writes a data-dependent mix).  -1: not a constant.
    // (DBG 9, the timeline build: wave 0 issues its timestamp stores on top -- more operations behind the prologue, so the counted wait only gets stricter)
    // the 16-bit SwiGLU epilogue issues exactly ONE unconditional buffer store per 16-row block of the wave (the `mb` loop around the store to rsH)
    // and no other vector-memory operation: that is the number the counted wait at the top of the next range adds to its 6.  A store made
    // conditional again, or a second one per block, has to change this line (a build with vmcnt(0) there must give bit-identical results)
    constexpr int SWIGLU_STORES_PER_BLOCK = 1;
    constexpr int EPI_STORES = (EPI == EPI_SWIGLU && FP8 == 0 && (DBG == 0 || DBG == 9) && PH2 && PH2V == 1) ? MB * SWIGLU_STORES_PER_BLOCK : -1;
    static_assert(EPI_STORES < 0 || 6 + EPI_STORES <= 63, "vmcnt is a 6-bit counter");
    [[maybe_unused]] bool stores_behind = false;
    LnPre lnpre;
    prepare_issue(cur, lnpre);
    prepare_finish(cur, lb, lnpre);
    while (true) {
        {
            // fp32 output: the accumulators of a WHOLE tile start from the bias (lane (l15, q4) owns channels 16 nb + 4 q4 .. + 3 of every
            // row block) -- sixteen registers the epilogue then does not need next to its residual batches; a partial K-range starts
            // from zero, ph8_reduce_f32_kernel adds the bias to the sums
            f32x4_t b0[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            if constexpr (EPI == EPI_F32) {
                if (g.bias && cur.whole) {
                    int q4 = q4_;
                    asm volatile("" : "+v"(q4));
#pragma unroll
                    for (int j = 0; j < 4; ++j) b0[j] = *reinterpret_cast<const f32x4_t*>(g.bias + cur.n0 + wc * 64 + 4 * q4 + j * 16);
                }
            }
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = b0[j];
        }
        // (materialised HERE: left to itself the compiler turns the initialisation into 64 register moves behind the barrier, in front of the
        // first fragment reads of the main loop; in front of the wait they overlap with the landing of the range's first tiles and the previous
        // epilogue's stores)
        if constexpr (EPI != EPI_HEADS) {          // (the heads kernel has no registers to spare for it: it spills)
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[i][j]));
        }
        // This wave's pieces of W-lo, A-lo, W-hi of the range's first K-tile have landed (the barrier covers the other waves'); LayerNorm
        // constants are visible.  The prologue is 12 LDS-DMA instructions per wave and phase A's own vmcnt(8) covers the rest of it, so the
        // six youngest may stay in flight -- and where the epilogue in between issued a KNOWN number of stores (EPI_STORES) those stay in
        // flight too: the wait used to be vmcnt(0), i.e. every store of the previous epilogue acknowledged before the next main loop starts
        // (the 1.9 us between two ranges in profiles/r05_ph8_timeline.txt).
        if constexpr (EPI_STORES >= 0) {
            if (stores_behind) wait_vmcnt<6 + (EPI_STORES >= 0 ? EPI_STORES : 0)>();
            else wait_vmcnt<6>();
        } else {
            wait_vmcnt<0>();
        }
        wait_lgkmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if constexpr (DBG == 2) {       // ablation: fragments are read once
            read_w(0, 0, fwl);
            read_w(0, 1, fwh);
            read_a(0, 0);
        }
        [[maybe_unused]] unsigned long long t0 = 0, t1 = 0, t2 = 0;
        if constexpr (DBG == 9) t0 = __builtin_amdgcn_s_memrealtime();
        const int mq = cur.m0 + wr * WR;
        const bool rows_valid = mq < M;
        if constexpr (PH2) {
            if (cur.tr) main_loop2(std::true_type{}, cur.kt0, cur.nk, rows_valid, mq + QR < M);
            else main_loop2(std::false_type{}, cur.kt0, cur.nk, rows_valid, mq + QR < M);
        } else {
            if (cur.tr) main_loop(std::true_type{}, cur.kt0, cur.nk, rows_valid, mq + QR < M);
            else main_loop(std::false_type{}, cur.kt0, cur.nk, rows_valid, mq + QR < M);
        }
        if constexpr (DBG == 9) t1 = __builtin_amdgcn_s_memrealtime();
        const bool more = next_seg(nxt);
        [[maybe_unused]] RopePre rp;
        if constexpr (EPI == EPI_HEADS) {
            if (rows_valid && cur.tr && (g.heads.kind[cur.n0 / (g.heads.heads * 64)] & 2)) rope_prefetch(cur, rp);
        }
        if (more) {                           // the ring is free: the next range's DMA latency hides behind this epilogue
            prepare_issue(nxt, lnpre);
            if constexpr (!LN_DEFER) prepare_finish(nxt, lb ^ 1, lnpre);
        }
        bool fin = true;
        if (EPI == EPI_F32 && !cur.whole) {           // a part of a K-split tile (fp32 output only): plain stores of the raw accumulators, the kernel
            int lane_l = lane;                            // boundary publishes them to ph8_reduce_f32_kernel
            asm volatile("" : "+v"(lane_l));              // (keeps the per-lane slab address out of the state carried across the main loop)
            float* mine = sc.sk_slab + (size_t)wgi * (BM * BN) + (size_t)(wave * MB * 4) * 256 + lane_l * 4;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) *reinterpret_cast<f32x4_t*>(mine + (mb * 4 + nb) * 256) = acc[mb][nb];
            fin = false;
        }
        if constexpr (DBG == 9) t2 = __builtin_amdgcn_s_memrealtime();
        if (fin && rows_valid) epilogue(cur, lb, rp, more);
        stores_behind = fin && rows_valid;
        if constexpr (DBG == 9) {
            if (tid_ == 0 && ts_n < 4) {
                unsigned long long* o = ts + ((size_t)blockIdx.x * 4 + ts_n) * 8;
                o[0] = t0; o[1] = t1; o[2] = t2; o[3] = __builtin_amdgcn_s_memrealtime();
                o[4] = cur.whole; o[5] = cur.nk; o[6] = fin; o[7] = 1;
            }
            ++ts_n;
        }
        if (!more) break;
        cur = nxt;
        lb ^= 1;
        if constexpr (LN_DEFER) prepare_finish(cur, lb, lnpre);          // (published by the barrier at the top of the loop)
    }
}

// Second launch of a K-split fp32-output GEMM: workgroup (j, mb) adds the slabs of remainder tile j for the row blocks mb of all
// eight waves -- every contributor's accumulator image in ascending workgroup order, bit-deterministic -- and runs the fp32 /
// residual / LayerNorm-producer epilogue on the sums.  Same lane <-> element map as the GEMM, so the slab reads are 1-KiB coalesced.
__global__ __launch_bounds__(512) void ph8_reduce_f32_kernel(GemmArgs g, Ph8Sched sc) {
    sat_f16_saturate();
    const int j = blockIdx.x >> 3, mb = blockIdx.x & 7;
    int first, parts;
    ph8_tile_parts(sc, j, first, parts);
    if (parts <= 1) return;                          // a whole tile: finished by the GEMM launch itself
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3, l15 = lane & 15, q4 = lane >> 4;
    int tm, tn;
    ph8_tile_of(sc, sc.rem0 + j, tm, tn);
    const int m = (tm << 8) + wr * 128 + mb * 16 + l15;
    const int ncol0 = (tn << 8) + wc * 64;
    if ((tm << 8) + wr * 128 >= g.M) return;
    f32x4_t v[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int w = first; w < first + parts; ++w) {
        const float* src = sc.sk_slab + (size_t)w * 65536 + (size_t)(wave * 32 + mb * 4) * 256 + lane * 4;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) v[nb] += *reinterpret_cast<const f32x4_t*>(src + nb * 256);
    }
    f32x4_t bia[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
        bia[nb] = g.bias ? *reinterpret_cast<const f32x4_t*>(g.bias + ncol0 + 4 * q4 + nb * 16) : f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) v[nb] += bia[nb];
    f32x4_t old[4];
    const Ph8F32Epi fe = ph8_f32_epi(g);
    ph8_load_resid(fe, old, m, ncol0, q4);
    ph8_epi_f32_row<true, false>(fe, v, old, m, ncol0, q4);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// host side: the persistent schedule of a shape -- a few integers in closed form, rebuilt per launch (no state, no allocation)
// ---------------------------------------------------------------------------------------------------------------------------------
int ph8_cus(int& out) {
    out = sat_device_cus();
    return out > 0 ? 0 : SAT_E_INVALID;
}

// Measured policy (profiles/r03_ph8_streamk.txt): split only fp32-output GEMMs with a long reduction (K >= 4096: FF-out) behind at
// least one whole round, and only when every remainder tile gets >= 2 parts (otherwise the whole tiles set the makespan and the slab
// traffic -- 256 KiB per part written and read back at HBM speed, everybody at the same time -- is pure loss): SA-2.0 FF-out -25 %.
// 8 prompts (134 remainder tiles on 256 CUs) and every K = 1536 GEMM stay whole; below one whole round the 128 x 128 tiles of
// gemm_bf16.hip are faster (FF-out at 1 prompt: 60 us against 67).
bool ph8_auto_split(const GemmArgs& a, bool epi_f32, int cus) {
    const long t_all = (long)cdiv(a.M, 256) * (a.N / 256);
    const long rem = t_all % cus;
    return epi_f32 && a.K >= 4096 && t_all > cus && rem > 0 && 2 * rem <= cus;
}

// split: 0 = the remainder round's tiles stay whole (contiguous shares, light tiles last), 1 = every remainder tile is cut along K
// (needs GemmArgs::slab), -1 = the measured policy above
int ph8_schedule(const GemmArgs& a, int split, bool epi_f32, int bm, int bn, int wgs_per_cu, Ph8Sched& out) {
    int cus = 0;
    SAT_TRY(ph8_cus(cus));
    // the automatic policy only splits when the caller's slab holds one accumulator image per workgroup -- the same condition the tile score
    // assumes (sat_gemm_ph8_splits); an undersized workspace runs the unsplit schedule.  A FORCED split (variant bit 16) keeps the hard error.
    const bool have_slab = a.slab != nullptr && (split >= 0 || a.slab_bytes >= (size_t)cus * 65536 * sizeof(float));
    if (split < 0) split = (bm == 256 && epi_f32 && have_slab && ph8_auto_split(a, epi_f32, cus)) ? 1 : 0;
    if (bm != 256 || !epi_f32) split = 0;          // the K-split machinery (slabs, reduce kernel) is built for the 256 x 256 fp32-output tile
    Ph8Sched s{};
    const int tiles_m = cdiv(a.M, bm), tail = a.M % bm;
    s.tiles_n = a.N / bn;
    s.light = (tail != 0 && tail <= bm / 4 && tiles_m > 1) ? 1 : 0;          // only the first quadrant of the first wave row has rows
    s.tiles_m_full = tiles_m - s.light;
    s.nkp = a.K / 128;
    const long t_full = (long)s.tiles_m_full * s.tiles_n, t_light = s.light ? s.tiles_n : 0;
    const long t_all = t_full + t_light;
    s.G = (int)std::min<long>((long)cus * wgs_per_cu, split ? t_all * s.nkp : t_all);
    // Balanced rounds: a launch of more than one and at most two rounds runs on FEWER workgroups, every one with two tiles (390 tiles:
    // 2 x 195 instead of 256 + 134; FF-in at one prompt: 2 x 216, which is also what the vendor library's stream-K launches for this
    // shape).  The chip is power-limited under MFMA load -- a tile runs faster when fewer CUs are active -- so the idle CUs cost
    // less than a half-empty second round: FF-out at 8 prompts 336 -> 311 us, FF-in at one prompt 85.4 -> 82.6 us.  With many rounds it
    // loses (FF-in at 8 prompts, 12.2 rounds: 479 -> 488 us): profiles/r04_ph8_balanced_rounds.txt.  Variant bit 21 switches it off (A/B).
    if (!split && !(a.variant & 0x200000) && t_all > s.G && t_all <= 2L * s.G) s.G = (int)((t_all + 1) / 2);
#ifdef SAT_GEMM_EXPERIMENTS
    if (!split && (a.variant & 0x400000) && t_all > s.G) {          // bit 22 (A/B): balanced rounds at any round count
        const long rounds = (t_all + s.G - 1) / s.G;
        s.G = (int)((t_all + rounds - 1) / rounds);
    }
#endif
    s.dp_rounds = (int)((split ? t_all : t_full) / s.G);
    // K-split with at least one whole round: the light tiles go FIRST (they idle their workgroup for half of round 0 -- a handful of
    // them) so that the remainder round holds full tiles only and splits evenly
    s.light_first = (split && s.dp_rounds >= 1 && t_light) ? 1 : 0;
    s.rem0 = (int)((long)s.dp_rounds * s.G);
    s.sk_tiles = (int)(t_all - s.rem0);
    s.split = (split && s.sk_tiles) ? 1 : 0;
    if (s.sk_tiles && !s.split) {
        s.sk_q = s.sk_tiles / s.G;
        s.sk_r = s.sk_tiles % s.G;
    } else if (s.sk_tiles) {
        // remainder tiles in work order: full ones (cost 2), then -- unless they went first -- the light ones (cost 1).  Parts per tile in
        // proportion to cost, at most min(nkp, 8); leftover workgroups give the first full tiles one more part: three classes.
        const int n_light = s.light_first ? 0 : (int)std::min<long>(t_light, s.sk_tiles);
        const int n_full = s.sk_tiles - n_light;
        const long cost2 = 2L * n_full + n_light;
        const int cap = std::min(s.nkp, 8);
        const int pf = (int)std::max<long>(1, std::min<long>(cap, (long)s.G * 2 / cost2));
        const int pl = (int)std::max<long>(1, std::min<long>(cap, (long)s.G * 1 / cost2));
        long used = (long)n_full * pf + (long)n_light * pl;
        int extra = 0;
        if (pf < cap && used < s.G) extra = (int)std::min<long>(n_full, s.G - used);
        used += extra;
        SAT_CHECK_ARG(used <= s.G, SAT_E_INVALID, "gemm(8-phase): the K-split needs %ld workgroups, has %d", used, s.G);
        s.cls_n[0] = extra; s.cls_p[0] = pf + 1;
        s.cls_n[1] = n_full - extra; s.cls_p[1] = pf;
        s.cls_n[2] = n_light; s.cls_p[2] = pl;
        const size_t need = (size_t)s.G * 65536 * sizeof(float);
        SAT_CHECK_ARG(a.slab && a.slab_bytes >= need, SAT_E_WORKSPACE, "gemm(8-phase): the K-split of the remainder round needs %zu bytes of slab workspace, got %zu",
                      need, a.slab ? a.slab_bytes : (size_t)0);
        SAT_CHECK_ARG(((uintptr_t)a.slab & 15) == 0, SAT_E_INVALID, "gemm(8-phase): the slab workspace must be 16-byte aligned");
        s.sk_slab = a.slab;
    }
    out = s;
    return 0;
}

#ifdef SAT_GEMM_EXPERIMENTS
unsigned long long* g_ts_buf = nullptr;
#endif

template <int EPI, int DBG = 0, bool PH2 = true, int PH2V = 1, int WN = 4, int MFQ = 4, int FP8 = 0, bool GATED = false>
int launch_ph8(const GemmArgs& a0, hipStream_t stream) {
    GemmArgs a = a0;
    if constexpr (FP8 != 0) {
        // e4m3 operands with per-token / per-output-channel scales (unit block scales in the MFMA): the kernel counts 16-bit columns,
        // one 128-byte LDS row = 128 e4m3 = one v_mfma_scale_f32_16x16x128_f8f6f4 step
        SAT_CHECK_ARG(a0.fp8 == 2 && a0.K % 256 == 0 && a0.a_scale && a0.w_scale, SAT_E_UNSUPPORTED,
                      "gemm(8-phase, e4m3): K=%d must be a multiple of 256 and both scale vectors given", a0.K);
        SAT_CHECK_ARG(!a0.ln_part && (EPI == EPI_SWIGLU || !a0.H8), SAT_E_UNSUPPORTED, "gemm(8-phase, e4m3): no LayerNorm fold; MXFP8 output from SwiGLU only");
        a.K = a0.K / 2;
    } else {
        SAT_CHECK_ARG(!a.fp8 && !a.H8, SAT_E_UNSUPPORTED, "gemm(8-phase): built for bf16 operands");
    }
    constexpr int BM = 64 * MFQ, BN = 64 * WN, NT = 2 * WN * 64;
    constexpr int LDS = 2 * 2 * (BM / 2 + BN / 2) * 128 + 2 * (BM + BN) * 8 + (EPI == EPI_HEADS ? 128 : 0);          // ring + 2 x ((mean, rstd) per row + (c1, c2) per column) + rotation constants
    SAT_CHECK_ARG(a.N % BN == 0, SAT_E_UNSUPPORTED, "gemm(8-phase): N=%d not a multiple of %d", a.N, BN);
    SAT_CHECK_ARG(a.K % 128 == 0, SAT_E_UNSUPPORTED, "gemm(8-phase): K=%d must be a multiple of 128", a.K);
    SAT_CHECK_ARG((uint64_t)a.M * (uint64_t)a.K * 2u < (1ull << 31), SAT_E_UNSUPPORTED, "gemm(8-phase): A larger than 2 GiB");
    if constexpr (EPI == EPI_F32)         // the epilogue addresses C / xb / ln_part through buffer descriptors with 32-bit byte offsets
        SAT_CHECK_ARG(((uint64_t)a.M + 256) * (uint64_t)a.ldc * 4u < (1ull << 31) && a.ldc >= a.N, SAT_E_UNSUPPORTED, "gemm(8-phase): C larger than 2 GiB");
    if constexpr (EPI == EPI_SWIGLU)      // the 16-bit output is addressed through a buffer descriptor with 32-bit byte offsets
        SAT_CHECK_ARG(((uint64_t)a.M + 256) * (uint64_t)(a.N / 2) * 2u < (1ull << 31), SAT_E_UNSUPPORTED, "gemm(8-phase): H larger than 2 GiB");
    constexpr bool LN_CONS = EPI == EPI_SWIGLU || EPI == EPI_HEADS;
    SAT_CHECK_ARG(LN_CONS || !a.ln_part, SAT_E_UNSUPPORTED, "gemm(8-phase): the LayerNorm fold is finished by the SwiGLU / heads epilogues");
    SAT_CHECK_ARG((!a.xb && !a.ln_part_out) || (EPI == EPI_F32 && a.xb && a.ln_part_out), SAT_E_UNSUPPORTED,
                  "gemm(8-phase): the bf16 image / row statistics come from the fp32-output epilogue");
    if constexpr (EPI == EPI_HEADS) {
        SAT_CHECK_ARG((a.heads.heads * 64) % BN == 0 && a.N == a.heads.parts * a.heads.heads * 64, SAT_E_UNSUPPORTED,
                      "gemm(8-phase): a %d-column tile must not straddle q / k / v (heads=%d)", BN, a.heads.heads);
        for (int p = 0; p < a.heads.parts; ++p)
            SAT_CHECK_ARG((a.heads.kind[p] & 3) != 3, SAT_E_UNSUPPORTED, "gemm(8-phase): no rotation on a transposed destination");
        SAT_CHECK_ARG(!a.heads.xa_k, SAT_E_UNSUPPORTED, "gemm(8-phase): the fused cross-attention epilogue lives in the 128 x 64 tile");
    }
    Ph8Sched sc;
    // bits 16 / 17 of the variant force / forbid the K-split of the remainder round (measurements, tests)
    const int split = (a.variant & 0x10000) ? 1 : (a.variant & 0x20000) ? 0 : -1;
    SAT_TRY(ph8_schedule(a, split, EPI == EPI_F32, BM, BN, BM == 256 ? 1 : 2, sc));
    SAT_CHECK_ARG(GATED == (a0.gate != nullptr), SAT_E_INVALID, "gemm(8-phase): gated / plain build mismatch");
    auto kern = gemm_ph8_kernel<EPI, DBG, PH2, PH2V, WN, MFQ, FP8, GATED>;
    SAT_TRY(sat_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS));
    unsigned long long* ts = nullptr;
#ifdef SAT_GEMM_EXPERIMENTS
    if constexpr (DBG == 9) {
        if (!g_ts_buf) SAT_HIP(hipMalloc(&g_ts_buf, 256 * 4 * 8 * sizeof(unsigned long long)));
        SAT_HIP(hipMemsetAsync(g_ts_buf, 0, 256 * 4 * 8 * sizeof(unsigned long long), stream));
        ts = g_ts_buf;
    }
#endif
    hipLaunchKernelGGL(kern, dim3(sc.G), dim3(NT), LDS, stream, a, sc, ts);
    if (sc.split) hipLaunchKernelGGL(ph8_reduce_f32_kernel, dim3(sc.sk_tiles * 8), dim3(512), 0, stream, a, sc);
    SAT_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// whether the launcher's automatic choice of the 256 x 256 tile should land here (a forced variant 80 always does)
bool SAT_OPNS::sat_gemm_ph8_supports(int epi, const GemmArgs& a) {
    if (a.N % 256 || a.K % 128 || (uint64_t)a.M * (uint64_t)a.K * 2u >= (1ull << 31)) return false;
    if (a.fp8 || a.H8) {      // e4m3: the LayerNorm-fed GEMMs (to_qkv, cross to_q, FF-in) with per-token scales
        if (a.fp8 != 2 || a.K % 256 || a.ln_part || !(epi == EPI_SWIGLU || epi == EPI_HEADS) || (a.H8 && epi != EPI_SWIGLU)) return false;
    }
    // fp32-output GEMMs with a short reduction (to_out, cross to_out: K = 1536) spend a third of their time in the residual
    // read-modify-write at HBM speed; persistent workgroups run those epilogues in lockstep, the 16-wave tile's independent workgroups
    // drift apart and overlap them with other tiles' main loops: measured 111 us against 122 at 8 prompts (profiles/r03_ph8_streamk.txt)
    if ((epi == EPI_F32 || epi == EPI_RESID) && a.K < 4096 && sat_wide_tile_of(a.variant) != 81) return false;          // (81: sat_dit_cfg.tile_policy, A/B)
    if (epi == EPI_HEADS) return (a.heads.heads * 64) % 256 == 0;
    return true;
}

#if defined(SAT_GEMM_EXPERIMENTS) && !defined(SAT_OPERAND_F16)
extern "C" __attribute__((visibility("default"))) int sat_gemm_ph8_timestamps(unsigned long long* out_host) {
    SAT_CHECK_ARG(g_ts_buf, SAT_E_INVALID, "no timestamps recorded");
    SAT_HIP(hipDeviceSynchronize());
    SAT_HIP(hipMemcpy(out_host, g_ts_buf, 256 * 4 * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}
#endif

bool SAT_OPNS::sat_gemm_ph8_splits(int epi, const GemmArgs& a) {
    int cus = 0;
    if (ph8_cus(cus) != 0 || !a.slab || a.slab_bytes < (size_t)cus * 65536 * sizeof(float)) return false;
    return ph8_auto_split(a, epi == EPI_F32 || epi == EPI_RESID, cus);
}

// bytes of slab workspace (GemmArgs::slab) with which the launcher's automatic schedule splits the remainder round of this shape
// along K; 0 = it would not split (callers size their workspace with this: sat_dit_workspace_bytes)
size_t SAT_OPNS::sat_gemm_ph8_slab_bytes(int epi, int M, int N, int K) {
    int cus = 0;
    if (ph8_cus(cus) != 0) return 0;
    GemmArgs a{};
    a.M = M; a.N = N; a.K = K;
    if (N % 256 || K % 128 || !ph8_auto_split(a, epi == EPI_F32 || epi == EPI_RESID, cus)) return 0;
    return (size_t)cus * 65536 * sizeof(float);
}

int SAT_OPNS::sat_launch_gemm_ph8(int epi, const GemmArgs& a, hipStream_t stream) {
    const int dbg = (a.variant & 0xfff) / 100;
#ifdef SAT_GEMM_EXPERIMENTS
    // The 128 x 128 geometry (4 waves, two workgroups per CU), experiments build only: measured SLOWER than the 16-wave-family tiles at
    // every one-prompt shape (FF-out 69.5 us vs 62.3, to_out 26.8 vs 22.4, cross 25.0 vs 16.2, QKV 68 vs 51;
    // profiles/r03_ph8_128x128_geometry_negative.txt) -- 16 MFMAs between barriers and half the operand reuse per LDS byte.
    if ((a.variant & 0xfff) % 100 == 81) {
        switch (epi) {
            case EPI_F32:
            case EPI_RESID: return launch_ph8<EPI_F32, 0, true, 1, 2, 2>(a, stream);
            case EPI_SWIGLU: return launch_ph8<EPI_SWIGLU, 0, true, 1, 2, 2>(a, stream);
            case EPI_HEADS: return launch_ph8<EPI_HEADS, 0, true, 1, 2, 2>(a, stream);
        }
    }
#endif
    switch (epi) {
        case EPI_F32:
        case EPI_RESID:
            switch (dbg) {
                case 0:
#ifdef SAT_GEMM_EXPERIMENTS
                    if (a.variant & 0x40000) return launch_ph8<EPI_F32, 0, false>(a, stream);          // bit 18: the four-phase loop (A/B)
                    if (a.variant & 0x80000) return launch_ph8<EPI_F32, 0, true, 2>(a, stream);        // bit 19: W-hi issued one phase earlier
#endif
                    if (a.gate) return launch_ph8<EPI_F32, 0, true, 1, 4, 4, 0, true>(a, stream);          // adaLN
                    return launch_ph8<EPI_F32>(a, stream);
#ifdef SAT_GEMM_EXPERIMENTS
                case 1: return launch_ph8<EPI_F32, 1, false>(a, stream);
                case 2: return launch_ph8<EPI_F32, 2, false>(a, stream);
                case 3: return launch_ph8<EPI_F32, 3, false>(a, stream);
                case 9: return launch_ph8<EPI_F32, 9>(a, stream);
#endif
            }
            break;
        case EPI_SWIGLU:
#ifdef SAT_GEMM_EXPERIMENTS
            if (dbg == 0 && (a.variant & 0x40000)) return launch_ph8<EPI_SWIGLU, 0, false>(a, stream);
            if (dbg == 0 && (a.variant & 0x80000)) return launch_ph8<EPI_SWIGLU, 0, true, 2>(a, stream);
#endif
#ifndef SAT_OPERAND_F16          // (e4m3 operands ride in the bf16 build: sat_launch_gemm rejects f16 && fp8)
            if (dbg == 0 && a.fp8) return launch_ph8<EPI_SWIGLU, 0, true, 1, 4, 4, 2>(a, stream);
#endif
            if (dbg == 0) return launch_ph8<EPI_SWIGLU>(a, stream);
#ifdef SAT_GEMM_EXPERIMENTS
            if (dbg == 9) return launch_ph8<EPI_SWIGLU, 9>(a, stream);
#endif
            break;
        case EPI_HEADS:
#ifndef SAT_OPERAND_F16
            if (dbg == 0 && a.fp8) return launch_ph8<EPI_HEADS, 0, true, 1, 4, 4, 2>(a, stream);
#endif
            if (dbg == 0) return launch_ph8<EPI_HEADS>(a, stream);
#ifdef SAT_GEMM_EXPERIMENTS
            if (dbg == 9) return launch_ph8<EPI_HEADS, 9>(a, stream);
#endif
            break;
    }
    sat_set_error("gemm(8-phase): epilogue %d / ablation %d not built", epi, dbg);
    return SAT_E_UNSUPPORTED;
}
