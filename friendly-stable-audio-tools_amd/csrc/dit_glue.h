#pragma once
#include "sat_common.h"

int glue_small_linear(const float* x, int ldx, const float* W, const float* bias, const float* add, int ldadd, void* y,
                      int ldy, int R, int N, int K, int act, int out16, hipStream_t s);      // out16: 0 fp32, 1 bf16, 2 fp16
int glue_adaln_finish(float* ssg, int64_t n, int D, hipStream_t s);
int glue_fourier(const float* t, float t_const, const float* w, float* out, int B, int half_feat, hipStream_t s);
int glue_fold_in(const float* Win, const float* Wpre, float* Weff, int D, int C, hipStream_t s);
int glue_fold_out(const float* Wout, const float* Wpost, float* Weff, int D, int C, hipStream_t s);
int glue_input_proj(const float* x, const float* Weff, float* X, int Bf, int xB, int C, int T, int S, int D, float xscale,
                    hipStream_t s);
int glue_output_proj(const float* X, const float* Weff, float* out, int Bf, int C, int T, int S, int D, hipStream_t s);
int glue_cfg_denoise(const float* mo, const float* x, float* den, int B, int C, int T, int use_cfg, float cfg_scale,
                     float scale_phi, float c_out, float c_skip, hipStream_t s);
// diagnostics of the residual stream (sat_dit_debug): per launch 4 floats at `out`, combined with atomics over the M rows:
// [0] max |x|, [1] max over rows of |mean| / std, [2] number of elements with |x| > 65504 (what an fp16 image saturates), [3] max over rows of max|x| / rms
int glue_resid_stats(const float* X, int M, int D, float* out, hipStream_t s);

