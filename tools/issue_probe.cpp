// Issue-rate probe for gfx950 (build: hipcc --offload-arch=gfx950 -O3 tools/issue_probe.cpp -o tools/issue_probe; run on the GPU box).
// Answers, per SIMD: clocks per v_fma_f32 / v_pk_fma_f32 / v_exp_f32 / v_max3_f32 / v_cvt_pk_bf16_f32 / 32x32x16 bf16 MFMA,
// and whether MFMA and VALU work overlap (a) inside one wave when interleaved in program order, (b) across two waves of a SIMD.
// Everything is timed with HIP events over a long loop; the engine clock is derived from the v_fma_f32 line (4 clocks per wave64 instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define REP8(x) x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, int iters, int waves_mfma) {
    const int wave = threadIdx.x >> 6;
    float a0 = threadIdx.x * 1e-6f, a1 = a0 + 1e-3f, a2 = a0 + 2e-3f, a3 = a0 + 3e-3f, a4 = a0 + 4e-3f, a5 = a0 + 5e-3f, a6 = a0 + 6e-3f, a7 = a0 + 7e-3f;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(a0 + i); fb[i] = (__bf16)(a1 - i); }
    const float k = 0.999f;
    const bool do_mfma = (MODE == 6) ? (wave < waves_mfma) : true;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {        // 64 v_fma_f32
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                              "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
        } else if constexpr (MODE == 1) { // 64 v_exp_f32
            REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                              "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if constexpr (MODE == 2) { // 32 v_pk_fma_f32
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0));)
        } else if constexpr (MODE == 3) { // 64: 32 v_max3_f32 + 32 v_cvt_pk_bf16_f32
            REP8(asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %4, %5\n"
                              "v_cvt_pk_bf16_f32 %4, %0, %1\n v_cvt_pk_bf16_f32 %5, %2, %3\n v_cvt_pk_bf16_f32 %6, %0, %2\n v_cvt_pk_bf16_f32 %7, %1, %3\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if constexpr (MODE == 4) { // 16 MFMA 32x32x16 bf16, four independent accumulators
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c3, 0, 0, 0);
            }
        } else if constexpr (MODE == 5) { // one wave: 16 MFMA interleaved with 64 v_exp + 64 v_fma (4 + 4 per MFMA), program order
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#define VBLK asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %4, %4, %8, %8\n v_exp_f32 %1, %1\n v_fma_f32 %5, %5, %8, %8\n" \
                          "v_exp_f32 %2, %2\n v_fma_f32 %6, %6, %8, %8\n v_exp_f32 %3, %3\n v_fma_f32 %7, %7, %8, %8\n" \
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c0, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); VBLK __builtin_amdgcn_sched_barrier(0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c1, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); VBLK __builtin_amdgcn_sched_barrier(0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c2, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); VBLK __builtin_amdgcn_sched_barrier(0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c3, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); VBLK __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (MODE == 6) { // two kinds of waves: `waves_mfma` waves run 16 MFMA, the others 64 v_exp + 64 v_fma
            if (do_mfma) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c3, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) { VBLK }
            }
        } else if constexpr (MODE == 7) { // one wave, phases: 16 MFMA, then 64 v_exp + 64 v_fma that DEPEND on nothing (no interleave): serial?
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c3, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) { VBLK }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[1] + p2[0] + p3[1];
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE>
double run(int threads, int iters, int waves_mfma, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, out, iters / 8, waves_mfma);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, waves_mfma);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 / iters;     // seconds per loop iteration
}

int main() {
    float* out; hipMalloc(&out, 4096);
    const int IT = 200000;
    // one wave per SIMD (256 threads), one workgroup per CU
    double t_fma = run<0>(256, IT, 0, out);
    double clk = 64 * 4 / t_fma;          // v_fma_f32: 4 clocks per wave64 instruction
    printf("engine clock derived from v_fma_f32 (4 clk/instr): %.3f GHz\n", clk * 1e-9);
    auto cyc = [&](double t) { return t * clk; };
    printf("v_fma_f32            : %6.2f clk/instr\n", cyc(t_fma) / 64);
    printf("v_exp_f32            : %6.2f clk/instr\n", cyc(run<1>(256, IT, 0, out)) / 64);
    printf("v_pk_fma_f32         : %6.2f clk/instr\n", cyc(run<2>(256, IT, 0, out)) / 32);
    printf("v_max3 + v_cvt_pk_bf16: %6.2f clk/instr\n", cyc(run<3>(256, IT, 0, out)) / 64);
    double t_m = run<4>(256, IT, 0, out);
    printf("mfma 32x32x16 bf16   : %6.2f clk/instr (1 wave/SIMD)\n", cyc(t_m) / 16);
    printf("mfma 32x32x16 bf16   : %6.2f clk/instr per SIMD (2 waves/SIMD)\n", cyc(run<4>(512, IT, 0, out)) / 32);
    double t5 = run<5>(256, IT, 0, out), t7 = run<7>(256, IT, 0, out);
    printf("1 wave: 16 mfma + 64 exp + 64 fma interleaved: %7.1f clk/iter   phased: %7.1f   (mfma alone %7.1f)\n", cyc(t5), cyc(t7), cyc(t_m));
    printf("2 waves/SIMD, both interleaved             : %7.1f clk/iter (2 iterations of work)\n", cyc(run<5>(512, IT, 0, out)));
    printf("2 waves/SIMD, both phased                  : %7.1f clk/iter (2 iterations of work)\n", cyc(run<7>(512, IT, 0, out)));
    printf("2 waves/SIMD: waves 0-3 mfma, 4-7 valu      : %7.1f clk/iter (16 mfma + 128 valu per SIMD)\n", cyc(run<6>(512, IT, 4, out)));
    printf("2 waves/SIMD: all valu (64 exp + 64 fma each): %7.1f clk/iter\n", cyc(run<6>(512, IT, 0, out)));
    printf("1 wave/SIMD : valu only (64 exp + 64 fma)    : %7.1f clk/iter\n", cyc(run<6>(256, IT, 0, out)));
    return 0;
}
