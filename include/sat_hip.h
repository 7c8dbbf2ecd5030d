/*
 * sat_hip.h -- C ABI of libsat_hip.so: the MI355X (gfx950) implementation of the
 * diffusion-sampling hot path of stable-audio-tools (DiT denoiser + DPM-Solver++ update +
 * Oobleck VAE encode/decode).
 *
 * The reference (yukara-ikemiya/friendly-stable-audio-tools) is pure Python on stock
 * PyTorch ops and has NO FFI / plugin boundary of its own: the
 * boundary it exposes is its Python API.  This header is what a Python (ctypes), C++ or
 * any-other-language host binds *underneath* that API.  Each entry point names the
 * reference interface it replaces (paths relative to stable_audio_tools/ in the reference).
 *
 * Conventions
 *   - return 0 on success; negative SAT_E_* on invalid argument / unsupported config;
 *     positive = hipError_t of a failed HIP call.  Never throws, never exits.
 *     sat_last_error() gives a thread-local message for the last non-zero return.
 *   - all pointers named *_dev are DEVICE pointers owned by the caller (e.g. PyTorch
 *     storage); tensors are dense, row-major ("PyTorch contiguous"), fp32 unless noted.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  All compute
 *     calls are asynchronous on that stream; no call synchronises the device.
 *   - no allocation on the hot calls (forward / denoise / step / decode / encode): the
 *     caller supplies the workspace; plans own only re-packed (bf16) weights and the
 *     per-generation cross-attention K/V cache (allocated in *_prepare_context).
 *   - a plan is not thread-safe; distinct plans are independent.  One process per GPU.
 */
#ifndef SAT_HIP_H
#define SAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the functions declared in this header are exported (tests/test_cabi.py). */
#pragma GCC visibility push(default)

#define SAT_OK 0
#define SAT_E_INVALID (-1)      /* bad argument (NULL, shape, size) */
#define SAT_E_UNSUPPORTED (-2)  /* configuration outside the kernels' supported set */
#define SAT_E_MISSING (-3)      /* a required tensor was never set */
#define SAT_E_WORKSPACE (-4)    /* workspace too small */
#define SAT_E_STATE (-5)        /* call order violated (e.g. forward before finalize) */

typedef void* sat_stream_t;     /* hipStream_t */

/* ABI version: 6.  A loader MUST check it (stable_audio_tools/_hip.py does): the cfg structs grow at the END between versions.  Since version 6
 * the size of the caller's sat_dit_cfg travels with the call (sat_dit_plan_create_sized), so a struct laid out for another version is rejected
 * deterministically instead of being read past its end; sat_dit_plan_create keeps reading exactly the version-5 layout (14 int32 fields). */
int sat_version(void);
const char* sat_last_error(void);

/* ------------------------------------------------------------------------------------
 * DiT denoiser.  Replaces models/dit.py:135-364 (DiffusionTransformer._forward/.forward),
 * models/transformer.py:99-809 (ContinuousTransformer, TransformerBlock, Attention,
 * FeedForward, LayerNorm, RotaryEmbedding), models/diffusion.py:482-529 (DiTWrapper).
 * Supported set: transformer_type "continuous_transformer", global_cond_type "prepend",
 * dim_heads 64, patch_size 1, no input_concat / prepend_cond, non-causal, no masks
 * (the reference discards them at inference: models/dit.py:250-252).
 * ---------------------------------------------------------------------------------- */
typedef struct sat_dit_plan sat_dit_plan;

#define SAT_GEMM_BF16 0
#define SAT_GEMM_FP8 1
#define SAT_GEMM_FP32X 2
#define SAT_GEMM_FP16 3
/* sat_dit_cfg.fp8_families */
#define SAT_FP8_QKV 1       /* self-attention to_qkv   (transformer.py:314), LayerNorm-fed: one scale per token */
#define SAT_FP8_CROSS_Q 2   /* cross-attention to_q    (transformer.py:311), LayerNorm-fed */
#define SAT_FP8_FF_IN 4     /* FF-in (SwiGLU)          (transformer.py:222), LayerNorm-fed */
#define SAT_FP8_FF_OUT 8    /* FF-out                  (transformer.py:270), MXFP8 A operand from the FF-in epilogue */
#define SAT_FP8_TO_OUT 16   /* to_out, self and cross  (transformer.py:319), MXFP8 A operand from the attention kernels */
#define SAT_FP8_ALL 31
#define SAT_FP8_DEFAULT (SAT_FP8_CROSS_Q | SAT_FP8_FF_IN | SAT_FP8_FF_OUT)

typedef struct sat_dit_cfg {
    int32_t io_channels;       /* config "io_channels" (64) */
    int32_t embed_dim;         /* "embed_dim" (1536); multiple of 128 */
    int32_t depth;             /* "depth" (24) */
    int32_t num_heads;         /* "num_heads" (24); embed_dim / num_heads must be 64 */
    int32_t cond_token_dim;    /* "cond_token_dim" (768); 0 = no cross-attention */
    int32_t cond_embed_dim;    /* cond_token_dim if project_cond_tokens=false else embed_dim */
    int32_t global_cond_dim;   /* "global_cond_dim" (1536); 0 = timestep embedding only */
    int32_t max_seq_len;       /* largest latent length T the plan will see (+1 prepend) */
    int32_t adaln;             /* 0: "global_cond_type": "prepend" (models/dit.py:186-197, the shipped configs);
                                  1: "adaLN" (models/dit.py:205-206, models/transformer.py:665-689): no prepend token, the
                                  global embedding drives per-layer scale/shift/gate of the self-attention and FF branches;
                                  needs "transformer.layers.N.to_scale_shift_gate.1.weight" [6*embed_dim, embed_dim] */
    int32_t gemm_dtype;        /* operand format of the block GEMMs and the attention kernels (fp32 accumulation throughout):
                                  0 (SAT_GEMM_BF16): bf16 operands everywhere (the C default of a zeroed struct; 3-4 % faster than fp16, 8x its operand rounding);
                                  3 (SAT_GEMM_FP16): IEEE fp16 operands everywhere (what the Python package selects by default and bench.py measures) -- the same
                                  kernels built on v_mfma_f32_*_f16, which
                                  gfx950 issues at the bf16 rate; three more significand bits (8x less operand rounding), and what the
                                  reference computes in on a GPU (inference/sampling.py:210 autocast, transformer.py:496-504).  Range
                                  policy: every fp32 -> fp16 conversion SATURATES at +-65504 (MODE.FP16_OVFL); values below 6e-8 flush;
                                  1 (SAT_GEMM_FP8): BASELINE config 5 -- the GEMMs fed by a LayerNorm (self-attention to_qkv, cross-attention
                                  to_q, FF-in; transformer.py:314, 311, 222) take OCP e4m3 operands with one scale per token
                                  (activations) and per output channel (weights), fp32 accumulation; FF-out (transformer.py:270)
                                  takes the SwiGLU output as MXFP8 (e4m3 + one E8M0 scale per 32 hidden channels, written by the
                                  FF-in epilogue, consumed as hardware block scales) and e4m3 weights; the attention kernels write
                                  MXFP8 too (one scale per half head) for the to_out projections (transformer.py:319); rest as 0;
                                  2 (SAT_GEMM_FP32X): fp32 VERIFICATION mode -- every contraction of the blocks on the exact fp32 MFMA
                                  (v_mfma_f32_32x32x2_f32), fp32 LayerNorm output, fp32 q / k / v / P, fp32 weights: same plan, data flow
                                  and index arithmetic, no operand rounding (~20x slower; meets 1e-3 vs the reference's outputs) */
    int32_t fp8_families;      /* gemm_dtype == 1 only: which GEMM families take e4m3 operands, an OR of SAT_FP8_* (0 = SAT_FP8_DEFAULT).  The
                                  full-size 12-step CFG-7 trajectory (tests/golden/traj_full.npz; tools/fp8_budget.py) prices each family's
                                  quantisation separately (rel-L2 vs fp32 after 12 steps; bf16 everywhere: 8e-3): the to_out projections
                                  on MXFP8 attention outputs 3.3e-1 -- that one family was round 3's whole fidelity problem --, to_qkv 6e-2
                                  (attention scores under CFG 7 do not tolerate 3-bit mantissas), cross to_q 1.2e-2, FF-in 9.5e-3, FF-out
                                  1.3e-2.  The default therefore quantises cross to_q, FF-in and FF-out (56 % of the block's FLOPs) and keeps
                                  to_qkv / to_out in bf16; SAT_FP8_ALL is round 3's mode.  SAT_FP8_FF_OUT needs SAT_FP8_FF_IN (the
                                  MXFP8 hidden state is written by the e4m3 FF-in epilogue) */
    int32_t ln_fold;           /* 1 (gemm_dtype 0 or 3, adaln == 0, embed_dim >= 256; ignored otherwise): no standalone LayerNorm launches
                                  after the first one.  LN(x) W^T = rstd (x (gamma.W)^T - mean rowsum(gamma.W)) + beta W^T: the GEMM
                                  that updates the residual stream (to_out, FF-out; transformer.py:692-700) also writes bf16(x) and
                                  per-row partial sums, the GEMM behind the LayerNorm (to_qkv, to_q, FF-in) multiplies bf16(x) with
                                  bf16(gamma.W) and finishes the normalisation on its fp32 accumulators.  Same arithmetic up to WHERE
                                  the one bf16 rounding of the activation happens (before instead of after the normalisation);
                                  0: three LayerNorm kernels per block */
    /* ---- version 5: the two A/B switches that used to be process-wide setters (sat_set_cross_attention_fusion, sat_gemm_set_wide_tile) are
     * per plan -- "distinct plans are independent" now also holds for them.  Appended at the END of the struct; sat_version() == 5. */
    int32_t cross_attention;   /* 0 (default): to_q projection + cross-attention core in ONE launch where the projection's 128 x 64 tiles fit one
                                  round of workgroups (one prompt; transformer.py:430-437 + 496-536), two kernels otherwise; 1: always two kernels */
    int32_t tile_policy;       /* 0 / 80 (default): the measured tile choice; A/B measurement switches: 22 = the 16-wave 256 x 256 tile of rounds
                                  1-2 instead of the 8-phase kernel, 81 = the 8-phase kernel also for fp32-output GEMMs with K < 4096, 82 = no
                                  two-K-group 128 x 128 tile */
} sat_dit_cfg;
#define SAT_DIT_CFG_BYTES_V5 56          /* the layout sat_dit_plan_create reads: 14 int32 fields, up to and including tile_policy */

/* cfg_bytes = sizeof(sat_dit_cfg) of the header the CALLER was built against.  This version knows one layout (SAT_DIT_CFG_BYTES_V5: the version-5
 * layout, unchanged in version 6); any other size is SAT_E_INVALID -- a caller built against another header is told so instead of having its struct
 * read past its end.  When the struct grows again, the older sizes stay accepted and the new fields take their defaults.
 * sat_dit_plan_create(cfg, out) == sat_dit_plan_create_sized(cfg, SAT_DIT_CFG_BYTES_V5, out). */
int sat_dit_plan_create_sized(const sat_dit_cfg* cfg, size_t cfg_bytes, sat_dit_plan** out_plan);
int sat_dit_plan_create(const sat_dit_cfg* cfg, sat_dit_plan** out_plan);
void sat_dit_plan_destroy(sat_dit_plan* plan);

/* Hand one fp32 tensor of the reference state dict to the plan.  `name` is the key
 * RELATIVE to the DiffusionTransformer module ("model.model." stripped), e.g.
 * "transformer.layers.3.ff.ff.0.proj.weight" (models/dit.py, models/transformer.py module
 * tree).  The pointer is only read inside sat_dit_plan_finalize. */
int sat_dit_plan_set_tensor(sat_dit_plan* plan, const char* name, const float* data_dev, int64_t numel);

/* Checks that every required tensor was set, converts GEMM weights to bf16 (SwiGLU rows
 * interleaved), folds the 1x1 pre/post convs into the in/out projections, builds the RoPE
 * table.  Replaces nn.Module.load_state_dict for the DiT. */
int sat_dit_plan_finalize(sat_dit_plan* plan, sat_stream_t stream);

/* Bytes of caller workspace needed by sat_dit_forward / sat_dit_denoise_cfg for `bf`
 * sequences (bf = 2*B with CFG) of latent length `t_len`. */
int sat_dit_workspace_bytes(const sat_dit_plan* plan, int32_t bf, int32_t t_len, size_t* out_bytes);

/* Per-generation constants (models/dit.py:150,154 to_cond_embed / to_global_embed and the
 * per-layer cross-attention to_kv projection, models/transformer.py:420-427): computed
 * once, reused by all sampler steps.  cross_attn_cond_dev [bf, lc, cond_token_dim] (NULL
 * if the model has no cross-attention), global_cond_dev [bf, global_cond_dim] or NULL. */
int sat_dit_prepare_context(sat_dit_plan* plan, const float* cross_attn_cond_dev, int32_t bf, int32_t lc,
                            const float* global_cond_dev, sat_stream_t stream);

/* Declares that sequences first_null_seq .. bf-1 of the prepared context are ALL-ZERO (the null embed of the
 * unconditional CFG half, models/dit.py:294-300).  Their cross-attention branch contributes exactly 0
 * (bias-free to_cond_embed / to_kv / to_out) and is skipped; results are unchanged.  -1 = none (default
 * after every sat_dit_prepare_context). */
int sat_dit_set_null_context_from(sat_dit_plan* plan, int32_t first_null_seq);

/* DiffusionTransformer._forward (models/dit.py:135-226) on bf sequences:
 * x_dev [bf, io_channels, t_len], t_dev [bf] (timestep in [0,1]) -> out_dev [bf, io_channels, t_len]. */
int sat_dit_forward(sat_dit_plan* plan, const float* x_dev, const float* t_dev, float* out_dev,
                    int32_t bf, int32_t t_len, void* workspace_dev, size_t workspace_bytes, sat_stream_t stream);

/* One k-diffusion VDenoiser evaluation with batched CFG (k_diffusion.external.VDenoiser
 * called at inference/sampling.py:159 around DiTWrapper.forward; CFG models/dit.py:270-349):
 *   denoised = cfg(DiT(x*c_in, t(sigma))) * c_out + x * c_skip
 * x_dev, denoised_dev [b, io_channels, t_len]; the context prepared must hold bf = 2*b
 * sequences (cond half first, uncond half second) when cfg_scale != 1, else bf = b.
 * scale_phi: CFG rescale (models/dit.py:342-345); 0 = off. */
int sat_dit_denoise_cfg(sat_dit_plan* plan, const float* x_dev, float sigma, float cfg_scale, float scale_phi,
                        float* denoised_dev, int32_t b, int32_t t_len,
                        void* workspace_dev, size_t workspace_bytes, sat_stream_t stream);

/* Measurement hook for bench.py: while enabled, every forward brackets the FFN-in (SwiGLU) GEMM
 * launch of ONE transformer layer (depth/2) with a hipEvent pair on the launch stream (at most
 * 4096 pairs; enabling resets the count).  sat_dit_profile_read synchronises on the recorded
 * events and returns the summed elapsed time, the number of launches and the GEMM shape. */
int sat_dit_profile(sat_dit_plan* plan, int32_t enable);
int sat_dit_profile_read(sat_dit_plan* plan, double* total_ms, int32_t* launches, int64_t* m, int64_t* n, int64_t* k);

/* Diagnostics of the residual stream, for checkpoints this build could not be run on (the fp16 default rounds the UN-normalised
 * residual row to 16 bits for the LayerNorm fold and saturates at +-65504 -- models/transformer.py:692-700; real DiT checkpoints are known for
 * massive-activation channels, the synthetic weights of the test-suite have none).  While enabled (16-bit operand modes), every forward runs one
 * small reduction kernel behind each of the three residual updates of every block (self-attention to_out, cross-attention to_out, FF-out) and
 * keeps, over the rows of that update: [0] max |x|, [1] max over rows of |mean| / std (the common-mode ratio the fold's error grows with:
 * 3.7e-3 at 0, 2.1e-2 at 8, tests/test_gpu_kernels.py::test_ln_fold_rows_with_common_mode), [2] the number of elements beyond +-65504,
 * [3] max over rows of max |x| / rms.  sat_dit_debug_read copies the [depth][3][4] floats of the LAST forward to the host (synchronises the
 * stream).  Enabling allocates, disabling frees: not for the timed path.  What to do with it: see README "Checking a real checkpoint". */
int sat_dit_debug(sat_dit_plan* plan, int32_t enable);
int sat_dit_debug_read(sat_dit_plan* plan, float* out_host, int32_t capacity_floats, sat_stream_t stream);

/* Batched-CFG combine alone (models/dit.py:336-345): model_out_dev [2*b, c, t] (cond half, then
 * uncond half) -> out_dev [b, c, t] = uncond + (cond - uncond) * cfg_scale, with the optional
 * std rescale when scale_phi != 0. */
int sat_cfg_combine(const float* model_out_dev, float* out_dev, int32_t b, int32_t c, int32_t t,
                    float cfg_scale, float scale_phi, sat_stream_t stream);

/* One DPM-Solver++(3M) SDE update (k_diffusion.sampling.sample_dpmpp_3m_sde, called at
 * inference/sampling.py:228), fused elementwise over n elements:
 *   x <- a*x + b*d + c1*(d - d1) + c2*(d1 - d2) + cn*noise     (in place on x_dev)
 * The host computes the scalars from the sigma schedule (see the Python host code);
 * d1_dev / d2_dev / noise_dev may be NULL when their coefficient is 0. */
int sat_dpmpp3m_update(float* x_dev, const float* d_dev, const float* d1_dev, const float* d2_dev,
                       const float* noise_dev, float a, float b, float c1, float c2, float cn,
                       int64_t n, sat_stream_t stream);

/* out <- c0*t0 + c1*t1 + c2*t2 + c3*t3 + c4*t4 over n floats; NULL terms are skipped, terms may alias out_dev.
 * State update of the single-step k-diffusion samplers selected at inference/sampling.py:212-225 (sample_heun, sample_lms,
 * sample_dpmpp_2s_ancestral, sample_dpm_2, sample_dpm_fast) and of sample_discrete_euler (:28-60): each is a linear
 * combination of the state, denoiser outputs and noise with host-computed scalars. */
int sat_lincomb(float* out_dev, const float* t0, float c0, const float* t1, float c1, const float* t2, float c2,
                const float* t3, float c3, const float* t4, float c4, int64_t n, sat_stream_t stream);

/* fp8 building blocks of gemm_dtype = 1, exported for the kernel-level parity tests:
 *   sat_quant_rows_fp8: x [rows, k] fp32 -> out8 [rows, k] OCP e4m3 bytes, row_scale[rows] = amax(row) / 448 (1 if the row is 0)
 *   sat_layernorm_fp8 : LayerNorm (eps 1e-5, transformer.py:205-206) fused with that row quantisation
 *   sat_gemm_fp8_f32  : c [m, n] (+)= (a8 . w8^T) * a_scale[m] * w_scale[n] (+ bias), k % 128 == 0, n % 128 == 0 */
int sat_quant_rows_fp8(const float* x_dev, void* out8_dev, float* row_scale_dev, int32_t rows, int32_t k, sat_stream_t stream);
int sat_layernorm_fp8(const float* x_dev, const float* gamma_dev, const float* beta_dev, void* y8_dev, float* row_scale_dev,
                      int32_t m, int32_t d, sat_stream_t stream);
int sat_gemm_fp8_f32(const void* a8_dev, const float* a_scale_dev, const void* w8_dev, const float* w_scale_dev,
                     const float* bias_dev, float* c_dev, int32_t m, int32_t n, int32_t k, int32_t accumulate, int32_t variant,
                     sat_stream_t stream);
/*   sat_quant_mx_rows_fp8: x [rows, k] fp32 -> MXFP8: out8 [rows, k] e4m3 bytes + scales [rows, k/32] E8M0 bytes
 *                          (scale = 2^ceil(log2(amax_block / 448)), byte = exponent + 127); k % 64 == 0
 *   sat_gemm_mxfp8_f32   : as sat_gemm_fp8_f32 with an MXFP8 A operand: the block scales are applied by the MFMA itself */
int sat_quant_mx_rows_fp8(const float* x_dev, void* out8_dev, void* scales_e8m0_dev, int32_t rows, int32_t k, sat_stream_t stream);
int sat_gemm_mxfp8_f32(const void* a8_dev, const void* a_scales_e8m0_dev, const void* w8_dev, const float* w_scale_dev,
                       const float* bias_dev, float* c_dev, int32_t m, int32_t n, int32_t k, int32_t accumulate, int32_t variant,
                       sat_stream_t stream);

/* Error estimate of one DPM-Solver adaptive step (k-diffusion sample_dpm_adaptive, inference/sampling.py:222-224):
 * partial_dev[j], j < n_partials, receive block-wise partial sums of ((x_low - x_high) / delta)^2 with
 * delta = max(atol, rtol * max(|x_low|, |x_prev|)); error = sqrt(sum(partials) / n), summed by the caller (fixed order). */
int sat_dpm_error_partials(const float* x_low_dev, const float* x_high_dev, const float* x_prev_dev, float atol, float rtol,
                           int64_t n, float* partial_dev, int32_t n_partials, sat_stream_t stream);

/* Inpainting re-injection (inference/sampling.py:98-103 get_bmask, :178-190 inpainting_callback, :168-172 initial mix):
 *   x[r, i] <- init[r, i] + noise[r, i] * sigma     wherever mask[i] <= strength      (in place on x_dev)
 * x/init/noise are [rows, t] (rows = batch * channels), mask is the [t] soft mask of generation.py:269-290 and
 * strength = (step + 1) / steps.  Called right after the denoiser evaluation of every step, as the reference's callback. */
int sat_inpaint_mix(float* x_dev, const float* init_dev, const float* noise_dev, const float* mask_dev,
                    float sigma, float strength, int64_t rows, int32_t t, sat_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Oobleck VAE.  Replaces models/autoencoders.py:45-194 (ResidualUnit, EncoderBlock,
 * DecoderBlock, OobleckEncoder, OobleckDecoder), models/blocks.py:318-358 (SnakeBeta),
 * dac.nn.layers.WNConv1d/WNConvTranspose1d (weight norm folded at finalize).
 * ---------------------------------------------------------------------------------- */
typedef struct sat_oobleck_plan sat_oobleck_plan;

typedef struct sat_oobleck_cfg {
    int32_t is_decoder;        /* 1 = OobleckDecoder, 0 = OobleckEncoder */
    int32_t io_channels;       /* decoder out_channels / encoder in_channels (2) */
    int32_t channels;          /* "channels" (128) */
    int32_t latent_dim;        /* decoder input (64) / encoder output (128) channels */
    int32_t n_blocks;          /* len(strides) (5) */
    int32_t c_mults[8];        /* WITHOUT the implicit leading 1: e.g. {1,2,4,8,16} */
    int32_t strides[8];        /* e.g. {2,4,4,8,8} */
    int32_t gemm_dtype;        /* SAT_GEMM_BF16 (0): bf16 activations / weights in the convolutions; SAT_GEMM_FP16 (3): IEEE fp16 (the
                                  reference's `model_half`, models/pretransforms.py:39-59: encoder / decoder in half precision) -- the
                                  fp16 build of the same kernels, same MFMA rate, saturating conversions; fp32 accumulation either way */
} sat_oobleck_cfg;

int sat_oobleck_plan_create(const sat_oobleck_cfg* cfg, sat_oobleck_plan** out_plan);
void sat_oobleck_plan_destroy(sat_oobleck_plan* plan);
/* `name` relative to the OobleckEncoder/OobleckDecoder module, e.g.
 * "layers.1.layers.2.layers.1.weight_v". */
int sat_oobleck_plan_set_tensor(sat_oobleck_plan* plan, const char* name, const float* data_dev, int64_t numel);
int sat_oobleck_plan_finalize(sat_oobleck_plan* plan, sat_stream_t stream);
/* latent length `t_len` for both directions (audio length = t_len * prod(strides)). */
int sat_oobleck_workspace_bytes(const sat_oobleck_plan* plan, int32_t b, int32_t t_len, size_t* out_bytes);

/* OobleckDecoder.forward (models/autoencoders.py:193-194): z_dev [b, latent_dim, t_len]
 * -> audio_dev [b, io_channels, t_len * prod(strides)] fp32. */
int sat_oobleck_decode(sat_oobleck_plan* plan, const float* z_dev, float* audio_dev, int32_t b, int32_t t_len,
                       void* workspace_dev, size_t workspace_bytes, sat_stream_t stream);
/* OobleckEncoder.forward (models/autoencoders.py:152-153): audio_dev [b, io_channels, t_len*ratio]
 * -> out_dev [b, latent_dim, t_len] fp32 (for the VAE: mean | scale halves). */
int sat_oobleck_encode(sat_oobleck_plan* plan, const float* audio_dev, float* out_dev, int32_t b, int32_t t_len,
                       void* workspace_dev, size_t workspace_bytes, sat_stream_t stream);

/* VAEBottleneck.encode / vae_sample (models/bottleneck.py:46-62) with the Gaussian noise
 * supplied by the caller: z = noise * (softplus(scale) + 1e-4) + mean.
 * mean_scale_dev [b, 2*c, t], noise_dev/z_dev [b, c, t]. */
int sat_vae_sample(const float* mean_scale_dev, const float* noise_dev, float* z_dev,
                   int32_t b, int32_t c, int32_t t, sat_stream_t stream);

/* float_to_int16_audio (utils/audio_utils.py:21-26) for one item of n samples:
 * peak = max|x| (device reduction), div = maximize ? peak : max(peak, 1),
 * out = (int16) trunc(x / div * 32767).  scratch_dev: >= 4 bytes. */
int sat_float_to_int16(const float* x_dev, int16_t* out_dev, int64_t n, int32_t maximize,
                       void* scratch_dev, sat_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Unit-level entry points (used by the parity tests and micro-benchmarks; the plans above
 * are built from exactly these kernels).
 * ---------------------------------------------------------------------------------- */
/* LayerNorm (models/transformer.py:205-206): x [m,d] fp32 -> y [m,d] bf16 (eps 1e-5). */
int sat_layernorm_bf16(const float* x_dev, const float* gamma_dev, const float* beta_dev, void* y_bf16_dev,
                       int32_t m, int32_t d, sat_stream_t stream);
/* fp32 -> bf16 (round to nearest even) */
int sat_cast_bf16(const float* x_dev, void* y_bf16_dev, int64_t n, sat_stream_t stream);
/* C[m,n] (fp32) = A[m,k] (bf16) * W[n,k]^T (bf16)  (+ bias[n]) ; accumulate != 0 adds into C.
 * n % 128 == 0, k % 64 == 0.  variant selects a tile configuration (0 = default). */
int sat_gemm_bf16_f32(const void* a_bf16_dev, const void* w_bf16_dev, const float* bias_dev, float* c_dev,
                      int32_t m, int32_t n, int32_t k, int32_t accumulate, int32_t variant, sat_stream_t stream);
/* The same GEMM with caller-supplied scratch: the 8-phase 256 x 256 tile cuts the remainder round of a long reduction (k >= 4096 behind
 * at least one whole round of tiles, e.g. FF-out at the SA-2.0 shape; or forced with variant bit 16) along K -- every partial K-range
 * stores its raw accumulators to a slab in `ws`, a second launch adds them in a fixed order.  sat_gemm_f32_workspace_bytes gives the
 * size (0: this shape does not split).  Without workspace (the entry above) nothing is ever split.  The DiT plan carves this scratch
 * out of the workspace the caller hands to sat_dit_forward: no call allocates, and plans on different streams never share scratch. */
int sat_gemm_f32_workspace_bytes(int32_t m, int32_t n, int32_t k, int32_t variant, size_t* out_bytes);
int sat_gemm_bf16_f32_ws(const void* a_bf16_dev, const void* w_bf16_dev, const float* bias_dev, float* c_dev,
                         int32_t m, int32_t n, int32_t k, int32_t accumulate, int32_t variant, void* ws_dev, size_t ws_bytes,
                         sat_stream_t stream);
/* SwiGLU GEMM (models/transformer.py:211-235): h[m,n/2] (bf16) = (A W_v^T + b_v) * silu(A W_g^T + b_g)
 * with W [n,k] in the REFERENCE row order (value rows then gate rows); re-packed internally
 * into wpack_dev [n,k] bf16 and bpack_dev [n] fp32 scratch. */
int sat_gemm_swiglu_bf16(const void* a_bf16_dev, const float* w_f32_dev, const float* bias_f32_dev,
                         void* wpack_dev, float* bpack_dev, void* h_bf16_dev,
                         int32_t m, int32_t n, int32_t k, int32_t variant, sat_stream_t stream);
/* Attention core (models/transformer.py:496-536): q [b,h,sq_pad,64], k [b,kvh,sk_pad,64],
 * vt [b,kvh,64,sk_pad] bf16 -> out [b*sq, h*64] bf16; softmax(q k^T / 8) v, GQA h/kvh.
 * Key-side layout: the sk keys of sequence i occupy rows (of k) / columns (of vt) [o_i, o_i + sk) with
 * o_i = (i*sk) & 3; everything outside must be finite (zero).  sq_pad % 128 == 0, sk_pad % 64 == 0,
 * sk_pad >= sk + 3.  (The shift lets the QKV GEMM epilogue store V^T with aligned 8-byte stores.)
 * Key order inside vt: every aligned group of 16 key columns is stored as [0-3, 8-11, 4-7, 12-15], i.e. key s sits at column
 * (s & ~12) | ((s & 4) << 1) | ((s & 8) >> 1) -- the order in which the second attention MFMA consumes the probabilities a lane
 * holds, so that a V^T tile is copied to LDS verbatim.  k keeps the natural order.  sat_qkv_rope_bf16 writes this layout. */
int sat_attention_bf16(const void* q_dev, const void* k_dev, const void* vt_dev, void* out_dev,
                       int32_t b, int32_t h, int32_t kvh, int32_t sq, int32_t sk, int32_t sq_pad, int32_t sk_pad,
                       sat_stream_t stream);
/* The same with q PRE-SCALED by its producer: q holds bf16(q_fp32 * log2(e) / 8), i.e. the scores come out of the first MFMA
 * in the log2 domain.  This is the layout the DiT plan runs (its QKV / to_q GEMM epilogues write it with one rounding); it lets
 * the single-KV-group kernel carry the softmax reference through the matrix pipe as a fifth K-step, which removes the per-score
 * multiply-add from the issue-bound VALU stream (attention). */
int sat_attention_prescaled_bf16(const void* q_dev, const void* k_dev, const void* vt_dev, void* out_dev,
                                 int32_t b, int32_t h, int32_t kvh, int32_t sq, int32_t sk, int32_t sq_pad, int32_t sk_pad,
                                 sat_stream_t stream);
/* Cross-attention query projection and attention core in one launch (models/transformer.py:430-437 + 496-536; what the DiT plan
 * runs per layer while the 128 x 64 GEMM tiles of the projection fit one round of workgroups, i.e. at one prompt):
 * out [b*s, d] bf16 = softmax((a wq^T) k^T / 8) v per head of 64, a [b*s, d] bf16, wq [d, d] bf16, k / vt in the key-side layout of
 * sat_attention_bf16 with sk + 3 <= 192 keys (they are staged in LDS as a whole); GQA d/64 over kvh.  Q never reaches memory. */
int sat_cross_attention_fused_bf16(const void* a_bf16_dev, const void* wq_bf16_dev, const void* k_dev, const void* vt_dev, void* out_dev,
                                   int32_t b, int32_t s, int32_t d, int32_t kvh, int32_t sk, int32_t sk_pad, sat_stream_t stream);
/* Fused QKV projection + partial RoPE + head split (models/transformer.py:430-452):
 * a [b*s, d] bf16, w_qkv [3d, d] bf16 -> q,k [b,h,s_pad,64], vt [b,h,64,s_pad] bf16 (k / vt in the key-side
 * layout of sat_attention_bf16; s_pad % 128 == 0, s_pad >= s + 3).
 * inv_freq_dev [16] fp32 (RotaryEmbedding.inv_freq, models/transformer.py:114-115). */
int sat_qkv_rope_bf16(const void* a_bf16_dev, const void* w_bf16_dev, const float* inv_freq_dev,
                      void* q_dev, void* k_dev, void* vt_dev, float* rope_scratch_dev,
                      int32_t b, int32_t s, int32_t s_pad, int32_t d, int32_t variant, sat_stream_t stream);
/* LayerNorm folded into the GEMMs either side of it (sat_dit_cfg.ln_fold; models/transformer.py:692-700: x + f(LayerNorm(x))).
 * Producer = the GEMM that updates the residual stream (to_out, FF-out; transformer.py:319, 270):
 *   c [m, n] fp32 += a [m, k] . w [n, k]^T + bias; xb [m, n] bf16 = bf16(c); ln_part [m][n / 64][2] fp32 = per 64-column block
 *   (sum, sum of squares) of the rounded row.  n % 128 == 0, k % 64 == 0, k >= 192. */
int sat_gemm_resid_ln_bf16(const void* a_bf16_dev, const void* w_bf16_dev, const float* bias_dev, float* c_dev, void* xb_dev,
                           float* ln_part_dev, int32_t m, int32_t n, int32_t k, int32_t variant, sat_stream_t stream);
int sat_gemm_resid_ln_bf16_ws(const void* a_bf16_dev, const void* w_bf16_dev, const float* bias_dev, float* c_dev, void* xb_dev,
                              float* ln_part_dev, int32_t m, int32_t n, int32_t k, int32_t variant, void* ws_dev, size_t ws_bytes,
                              sat_stream_t stream);      /* with K-split scratch, as sat_gemm_bf16_f32_ws */
/* Consumers: xb / ln_part as written by the producer (k = its n).  SwiGLU FF-in (transformer.py:222, 232-235) of LayerNorm(x):
 * w_f32 [n, k], gamma / beta [k], bias [n] fp32 (reference layout: value rows then gate rows) -> h [m, n / 2] bf16.
 * wpack [n, k] bf16 and c12 [2 n] fp32 receive the re-packed operands: bf16(gamma (.) w) (value / gate rows interleaved by 32),
 * c1 = its row sums, c2 = w beta + bias; the epilogue computes rstd (acc - mean c1) + c2 before value * silu(gate). */
int sat_gemm_swiglu_ln_bf16(const void* xb_dev, const float* ln_part_dev, const float* w_f32_dev, const float* gamma_dev,
                            const float* beta_dev, const float* bias_f32_dev, void* wpack_dev, float* c12_dev, void* h_dev, int32_t m,
                            int32_t n, int32_t k, int32_t variant, sat_stream_t stream);
/* to_qkv + partial RoPE + head split of LayerNorm(x) (transformer.py:314, 430-452); outputs as sat_qkv_rope_bf16.
 * w_f32 [3d, d]; wpack [3d, d] bf16 and c12 [6 d] fp32 receive the re-packed operands. */
int sat_qkv_rope_ln_bf16(const void* xb_dev, const float* ln_part_dev, const float* w_f32_dev, const float* gamma_dev,
                         const float* beta_dev, void* wpack_dev, float* c12_dev, const float* inv_freq_dev, void* q_dev, void* k_dev,
                         void* vt_dev, float* rope_scratch_dev, int32_t b, int32_t s, int32_t s_pad, int32_t d, int32_t variant,
                         sat_stream_t stream);
/* ------------------------------------------------------------------------------------
 * The same unit-level entry points on IEEE fp16 operands (gemm_dtype = 3): identical signatures and layouts, every
 * "bf16" tensor holds fp16 instead, the kernels are the fp16 build of the same sources (v_mfma_f32_32x32x16_f16 /
 * v_mfma_f32_16x16x32_f16; conversions saturate at +-65504).
 * ---------------------------------------------------------------------------------- */
int sat_layernorm_f16(const float* x_dev, const float* gamma_dev, const float* beta_dev, void* y_f16_dev,
                      int32_t m, int32_t d, sat_stream_t stream);
int sat_cast_f16(const float* x_dev, void* y_f16_dev, int64_t n, sat_stream_t stream);
int sat_gemm_f16_f32(const void* a_f16_dev, const void* w_f16_dev, const float* bias_dev, float* c_dev,
                     int32_t m, int32_t n, int32_t k, int32_t accumulate, int32_t variant, sat_stream_t stream);
int sat_gemm_f16_f32_ws(const void* a_f16_dev, const void* w_f16_dev, const float* bias_dev, float* c_dev,
                        int32_t m, int32_t n, int32_t k, int32_t accumulate, int32_t variant, void* ws_dev, size_t ws_bytes,
                        sat_stream_t stream);
int sat_gemm_resid_ln_f16_ws(const void* a_f16_dev, const void* w_f16_dev, const float* bias_dev, float* c_dev, void* xb_dev,
                             float* ln_part_dev, int32_t m, int32_t n, int32_t k, int32_t variant, void* ws_dev, size_t ws_bytes,
                             sat_stream_t stream);
int sat_gemm_swiglu_f16(const void* a_f16_dev, const float* w_f32_dev, const float* bias_f32_dev,
                        void* wpack_dev, float* bpack_dev, void* h_f16_dev,
                        int32_t m, int32_t n, int32_t k, int32_t variant, sat_stream_t stream);
int sat_attention_f16(const void* q_dev, const void* k_dev, const void* vt_dev, void* out_dev,
                      int32_t b, int32_t h, int32_t kvh, int32_t sq, int32_t sk, int32_t sq_pad, int32_t sk_pad,
                      sat_stream_t stream);
int sat_attention_prescaled_f16(const void* q_dev, const void* k_dev, const void* vt_dev, void* out_dev,
                                int32_t b, int32_t h, int32_t kvh, int32_t sq, int32_t sk, int32_t sq_pad, int32_t sk_pad,
                                sat_stream_t stream);
int sat_cross_attention_fused_f16(const void* a_f16_dev, const void* wq_f16_dev, const void* k_dev, const void* vt_dev, void* out_dev,
                                  int32_t b, int32_t s, int32_t d, int32_t kvh, int32_t sk, int32_t sk_pad, sat_stream_t stream);
int sat_qkv_rope_f16(const void* a_f16_dev, const void* w_f16_dev, const float* inv_freq_dev,
                     void* q_dev, void* k_dev, void* vt_dev, float* rope_scratch_dev,
                     int32_t b, int32_t s, int32_t s_pad, int32_t d, int32_t variant, sat_stream_t stream);
int sat_gemm_resid_ln_f16(const void* a_f16_dev, const void* w_f16_dev, const float* bias_dev, float* c_dev, void* xb_dev,
                          float* ln_part_dev, int32_t m, int32_t n, int32_t k, int32_t variant, sat_stream_t stream);
int sat_gemm_swiglu_ln_f16(const void* xb_dev, const float* ln_part_dev, const float* w_f32_dev, const float* gamma_dev,
                           const float* beta_dev, const float* bias_f32_dev, void* wpack_dev, float* c12_dev, void* h_dev, int32_t m,
                           int32_t n, int32_t k, int32_t variant, sat_stream_t stream);
int sat_qkv_rope_ln_f16(const void* xb_dev, const float* ln_part_dev, const float* w_f32_dev, const float* gamma_dev,
                        const float* beta_dev, void* wpack_dev, float* c12_dev, const float* inv_freq_dev, void* q_dev, void* k_dev,
                        void* vt_dev, float* rope_scratch_dev, int32_t b, int32_t s, int32_t s_pad, int32_t d, int32_t variant,
                        sat_stream_t stream);

/* ------------------------------------------------------------------------------------
 * T5 encoder stack: the text front-end of the conditioner.  Replaces the transformers.T5EncoderModel call in
 * T5Conditioner.forward (models/conditioners.py:317-339): last_hidden_state for tokenised prompts.  The algorithm is
 * transformers' modeling_t5.py (T5Stack encoder: RMS T5LayerNorm, un-scaled attention with bucketed relative position bias shared
 * from block 0, ReLU or gated-GELU feed-forward); fp32 throughout (the reference runs it under fp16 autocast).
 * Tensors are named as in a Hugging Face T5 checkpoint: "shared.weight" (or "encoder.embed_tokens.weight"),
 * "encoder.block.N.layer.0.SelfAttention.{q,k,v,o}.weight", "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight",
 * "encoder.block.N.layer.0.layer_norm.weight", "encoder.block.N.layer.1.DenseReluDense.{wi | wi_0,wi_1}.weight", "....wo.weight",
 * "encoder.block.N.layer.1.layer_norm.weight", "encoder.final_layer_norm.weight".
 * ---------------------------------------------------------------------------------- */
typedef struct sat_t5_plan sat_t5_plan;
typedef struct sat_t5_cfg {
    int32_t vocab_size;        /* T5Config.vocab_size (32128) */
    int32_t d_model;           /* 768 for t5-base */
    int32_t d_kv;              /* 64; inner dim = num_heads * d_kv */
    int32_t d_ff;              /* 3072 */
    int32_t num_layers;        /* 12 */
    int32_t num_heads;         /* 12 */
    int32_t rel_buckets;       /* relative_attention_num_buckets (32) */
    int32_t rel_max_distance;  /* relative_attention_max_distance (128) */
    int32_t gated_gelu;        /* feed_forward_proj: 0 = "relu" (t5-*), 1 = "gated-gelu" (flan-t5-*) */
    int32_t proj_dim;          /* > 0: Conditioner.proj_out = Linear(d_model, proj_dim) (conditioners.py:23) applied to the output,
                                  tensors "proj_out.weight" [proj_dim, d_model] / "proj_out.bias"; 0: identity */
    float eps;                 /* layer_norm_epsilon (1e-6) */
} sat_t5_cfg;
int sat_t5_plan_create(const sat_t5_cfg* cfg, sat_t5_plan** out_plan);
void sat_t5_plan_destroy(sat_t5_plan* plan);
int sat_t5_plan_set_tensor(sat_t5_plan* plan, const char* name, const float* data_dev, int64_t numel);
int sat_t5_plan_finalize(sat_t5_plan* plan, sat_stream_t stream);
int sat_t5_workspace_bytes(const sat_t5_plan* plan, int32_t b, int32_t l, size_t* out_bytes);
/* input_ids_dev / attention_mask_dev [b, l] int32 (tokenizer output, padding = mask 0) -> out_dev [b, l, proj_dim or d_model] fp32
 * = proj_out(model(input_ids, attention_mask)["last_hidden_state"]), and with mask_output != 0 multiplied by the mask
 * (conditioners.py:333-341); l <= 512 */
int sat_t5_encode(sat_t5_plan* plan, const int32_t* input_ids_dev, const int32_t* attention_mask_dev, float* out_dev, int32_t b,
                  int32_t l, int32_t mask_output, void* workspace_dev, size_t workspace_bytes, sat_stream_t stream);
/* HOST helper (no GPU): T5Attention._relative_position_bucket(key - query, bidirectional=True) for key - query in
 * [-(l-1), l-1] -> out_host[2l - 1] */
int sat_t5_relative_buckets(int32_t l, int32_t num_buckets, int32_t max_distance, int32_t* out_host);

/* SnakeBeta (models/blocks.py:318-319): y = x + sin^2(x*exp(alpha_c)) / (exp(beta_c)+1e-9); x,y [b,c,t] fp32 */
int sat_snake_beta(const float* x_dev, const float* alpha_dev, const float* beta_dev, float* y_dev,
                   int32_t b, int32_t c, int32_t t, sat_stream_t stream);

/* Sample-rate conversion, polyphase windowed-sinc FIR = torchaudio.transforms.Resample with its defaults (inference/utils.py:25-27,
 * models/autoencoders.py:394-397, reconstruct_audios.py:34-35 of the reference).  x [rows, in_len] fp32 -> y [rows, out_len] fp32,
 * out_len <= ceil(in_len * new / orig); orig / new are the two rates divided by their gcd; bank [new, 2 * width + orig] fp32 is the
 * filter bank (row p = output phase p; built on the host, stable_audio_tools/inference/resample.py: sinc_resample_bank). */
int sat_resample_sinc(const float* x_dev, const float* bank_dev, float* y_dev, int32_t rows, int32_t in_len, int32_t out_len,
                      int32_t orig, int32_t new_rate, int32_t width, sat_stream_t stream);
/* Windowed overlap-add of the chunked codec paths, AudioAutoencoder.encode_audio / decode_audio / reconstruct_audio
 * (models/autoencoders.py:476-497, 548-571, 622-645): pieces_dev [batch, n_chunk, channels, chunk_len], chunk i placed at i * hop,
 * faded in / out over `overlap` samples with window_dev [2 * overlap] (torch.bartlett_window(2 * overlap)) except at the outer
 * edges of the first / last chunk -> out_dev [batch, channels, total_len] (every sample written). */
int sat_overlap_add(const float* pieces_dev, const float* window_dev, float* out_dev, int32_t batch, int32_t n_chunk, int32_t channels,
                    int32_t chunk_len, int32_t hop, int32_t overlap, int32_t total_len, sat_stream_t stream);

/* NumberConditioner.forward (models/conditioners.py:64-102) with its NumberEmbedder (models/adp.py:1495-1514, 680-694):
 *   x = (clamp(v, min_val, max_val) - min_val) / (max_val - min_val)
 *   out[b] = Linear(cat(x, sin(2 pi x w), cos(2 pi x w)))     w = embedder.embedding.0.weights [half_dim],
 *   linear_w = embedder.embedding.1.weight [features, 2*half_dim + 1], linear_b = ...bias [features]
 * values_dev [count] -> out_dev [count, features] (the caller adds the singleton token axis and the all-ones mask). */
int sat_number_embed(const float* values_dev, int32_t count, float min_val, float max_val, const float* pos_weights_dev,
                     int32_t half_dim, const float* linear_w_dev, const float* linear_b_dev, int32_t features, float* out_dev,
                     sat_stream_t stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* SAT_HIP_H */
