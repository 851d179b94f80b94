/*
 * xsd.h -- C ABI of the MI355X-native RRDB-generator engine (libxsd_hip.so).
 *
 * Drop-in boundary for the hot path of SamSweere/xmm-superres-denoise.  All pointers named
 * "dev" are device (HBM) pointers owned by the caller; the engine borrows them for the duration of a call and
 * launches asynchronously on the given hipStream_t (passed as void*; NULL = the null stream).  Every entry point
 * returns 0 on success or a negative xsd_status; xsd_last_error() returns a thread-local message.  No C++ types,
 * no torch types, never throws across the ABI.  One engine per process/GPU; calls on one engine are stream-ordered,
 * not thread-safe.
 *
 * Parameter vector layout ("flat params"): fp32, the reference's state_dict order, each tensor OIHW:
 *   conv_first.{weight,bias}, rrdb.{i}.RDB{r}.conv{c}.{weight,bias} (i<blocks, r=1..3, c=1..5), trunk_conv.*,
 *   conv_last.*, and for SR: upsampling.{3u}.* (u<num_upsample), HRconv.*
 *   (xmm_superres_denoise/models/modules/generator_rrdb.py:10-64,73-101; rrdb_blocks.py:23-32,60-64).
 * Images are NCHW fp32: x [B][in_channels][H][W], y [B][out_channels][sH][sW] (the shipped models have one channel: plain [B][H][W]).
 */
#ifndef XSD_H
#define XSD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct xsd_engine xsd_engine;

enum xsd_status {
    XSD_OK = 0,
    XSD_ERR_ARG = -1,     /* bad argument / unsupported configuration */
    XSD_ERR_HIP = -2,     /* a HIP runtime call failed */
    XSD_ERR_STATE = -3,   /* call sequence error (e.g. backward without a saved forward) */
    XSD_ERR_NOMEM = -4
};

enum xsd_kind { XSD_KIND_DN = 0, XSD_KIND_SR = 1 };

/* Mirrors the constructor arguments of GeneratorRRDB_DN / GeneratorRRDB_SR
 * (generator_rrdb.py:114-121 / :73-81) as mapped from RrdbCfg by Model.configure_model (models/model.py:157-186). */
typedef struct xsd_config {
    int32_t kind;          /* xsd_kind */
    int32_t in_channels;   /* 1..1024 (models.toml: 1).  DN: must equal out_channels or be 1 (`out + x`, generator_rrdb.py:134) */
    int32_t out_channels;  /* 1..1024 (models.toml: 1) */
    int32_t num_filters;   /* 1..1024 (models.toml: filters = 32).  up to 256 filters with up to 8 image channels in and out run on the
                              split-precision MFMA kernels (a feature tensor = 1..8 planes of 32 channels; widths that are no
                              multiple of 32 are zero-padded internally, the flat vectors keep the reference's layout); beyond
                              that the exact-fp32 kernels of csrc/generic_net.hip (xsd_set_math does not apply there) */
    int32_t num_res_blocks;/* >= 1 (models.toml: residual_blocks = 4) */
    int32_t num_upsample;  /* SR only: (hr_res/lr_res)/2, 1 or 2 */
    int32_t memory_efficient; /* rrdb_blocks.py:39-47 recompute policy; numerics identical. Enforced by the host (chunked recompute), not here */
    int32_t reserved;
} xsd_config;

const char* xsd_last_error(void);
const char* xsd_version(void);

/* replaces GeneratorRRDB_*.__init__ (engine state only; weights stay in the caller's flat buffer) */
int xsd_create(const xsd_config* cfg, xsd_engine** out);
void xsd_destroy(xsd_engine* e);
int64_t xsd_param_count(const xsd_engine* e);

/* Math mode of the MFMA convs (forward, input-gradient, weight-gradient).  Three modes, all carrying fp32 planes:
 * 0 = "fp32": exact fp32 on v_mfma_f32_32x32x2_f32 (bitwise an fp32 fma chain; 157 TFLOP/s peak).
 * 3 = "bf16x6" (strict): every fp32 operand is split EXACTLY into three bf16 terms (8+8+8 = fp32's 24 bits) and a product is
 *     six bf16 MFMA products (dropped terms <= 2^-23 relative); v_mfma_f32_32x32x16_bf16 sums its 16 products and the fp32
 *     accumulator exactly and rounds once.  Measured against float64 (tests/test_hip_precision.py, several seeds and sizes):
 *     error <= torch's fp32 CPU path and <= mode 0, forward and backward, on every tensor.
 * 4 = "f16x3" (default, the benchmark headline): every operand TENSOR is scaled by a power of two chosen from its max |x|
 *     (reported by the kernel that produced it) and split into two fp16 terms, x * s = h + l * 2^-11: 22-23 of fp32's 24
 *     significant bits per OPERAND (not 24); a product is three fp16 MFMA products, accumulation as in mode 3, the epilogue
 *     undoes the scales exactly.  What the tests hold it to, against float64: forward error <= torch fp32 and <= mode 0;
 *     backward (every parameter-gradient tensor and dL/dx) within 2x of torch's fp32 CPU path and within 1.25x of mode 0
 *     -- i.e. at the level of an fp32 fma chain, NOT below torch's blocked-summation kernel (measured 1.2-1.6x of it).  Three
 *     to four orders of magnitude inside the 1e-3 parity tolerance of the task; the label "fp32" without qualification
 *     belongs to modes 0 and 3 only.
 * (Modes 1 and 2 -- two-term bf16 splits with 16-bit significands -- existed in rounds 1-2 and were removed.)
 * Default: 4 (f16x3), or the environment variable XSD_MATH ("fp32" | "bf16x6" | "f16x3"; anything else fails xsd_create).
 * Changing it invalidates the packed weights and the plan.
 * XSD_MATH is the ONLY environment variable the production library (lib/libxsd_hip.so) reads.  Since round 6 the A/B switches and
 * test hooks below are compiled only into the test-hooks variant (make -C csrc hooks -> lib/libxsd_hip_hooks.so = the product's
 * objects with xsd_engine.hip built -DXSD_TEST_HOOKS; selected with XSD_LIB=<path>, as the diagnostic variant is) -- a shipped
 * library does not replan on an environment variable (tests/test_hip_network.py holds the product to that):
 *   XSD_WGRAD_BLOCK=0   the weight gradients of a dense block as one launch per G (five launches) instead of ONE pair-list launch
 *                       over its 15 (X, G) pairs (default 1; A/B switch, tools/ab_env.sh; tests hold the two forms to 2e-6);
 *   XSD_WGRAD_TAIL=0    no extra part on the CUs the block launch's 8 m x 15 workgroups leave (default 1; MI355X: a 17th part);
 *   XSD_TEST_NCU=n      plan the block launch as if the device had n compute units (n < 120: one launch per G);
 *   XSD_TEST_AMAX_CAP=n initial capacity of the max-|x| slot array (default 65536 floats), so that a small net exercises the
 *                       grow / copy / rebuild path a 256-filter x 64-block net would take.
 * Modes 3 and 4 address a plane's batch slice with 32-bit byte offsets: images of 2^24 or more output pixels are rejected
 * by xsd_forward (mode 0 takes them). */
int xsd_set_math(xsd_engine* e, int mode);
/* The mode the kernels of THIS engine compute in: the setting above on the plane kernels; 0 on the exact-fp32 path that serves
 * more than 256 filters / more than 8 image channels, whatever was set. */
int xsd_get_math(const xsd_engine* e);

/* Repack the caller's flat OIHW parameters into MFMA fragment-order panels (forward + transposed/flipped for the
 * input-gradient).  Must be called after every parameter update and before forward.  Replaces nothing in the
 * reference (torch reads OIHW directly); it is the engine's weight-layout step. */
int xsd_pack_weights(xsd_engine* e, const float* dev_params, void* stream);

/* replaces Model.forward = clamp(GeneratorRRDB_*.forward(x), 0, 1) (models/model.py:48-49;
 * generator_rrdb.py:66-69,103-110,130-137).  x: [B][in_channels][H][W]; y: [B][out_channels][sH][sW], s = 2^num_upsample (SR) or 1 (DN).
 * save_for_backward != 0 keeps every activation needed by xsd_backward (about 7.7 KB per LR pixel). */
int xsd_forward(xsd_engine* e, const float* dev_x, float* dev_y, int B, int H, int W, int save_for_backward, void* stream);

/* replaces torch autograd through the module for the last xsd_forward(save_for_backward=1):
 * dy: gradient wrt the (clamped) output y; dx_or_null: gradient wrt x; dev_grads: flat gradient vector in the flat-params
 * layout (overwritten, not accumulated). */
int xsd_backward(xsd_engine* e, const float* dev_dy, float* dev_dx_or_null, float* dev_grads, void* stream);

/* Data-parallel overlap support: the backward pass split into num_res_blocks + 2 stages, executed in order
 * stage 0 (output head + trunk_conv), stages 1..blocks (RRDB blocks-1 .. 0), stage blocks+1 (conv_first).
 * After stage s returns, the gradient ranges reported by xsd_grad_range(stage) are final on the stream, so the caller
 * can start their all-reduce on a side stream while later stages run. */
/* Several backwards after ONE forward (different dy) are allowed: stage 0 resets what the previous backward left. */
int xsd_backward_num_stages(const xsd_engine* e);
int xsd_backward_stage(xsd_engine* e, int stage, const float* dev_dy, float* dev_dx_or_null, float* dev_grads, void* stream);
int xsd_grad_range(const xsd_engine* e, int stage, int range_idx, int64_t* offset, int64_t* count); /* returns number of ranges */

/* mean-L1 loss (torchmetrics MeanAbsoluteError / F.l1_loss; utils/loss_functions.py:16): writes *dev_loss (float) and,
 * if dev_dy != NULL, d loss / d y = sign(y - t) / n. */
int xsd_l1_loss(xsd_engine* e, const float* dev_y, const float* dev_target, float* dev_dy_or_null, float* dev_loss,
                int64_t n, void* stream);

/* replaces create_loss + the per-batch Metric.forward of the composed loss (utils/loss_functions.py:11-47;
 * models/model.py:78; constants res/configs/loss_functions.toml:5-42).  w_* are the EFFECTIVE weights of the terms
 * (relative percentage x paper scaling; 0 = term absent), correction is added to the total when > 0
 * (loss_functions.py:44-45).  ssim / ms_ssim follow torchmetrics 1.x with gaussian_kernel=True: window
 * int(3.5*sigma+0.5)*2+1 taps, data_range from the images, k1/k2 as given, kernel_size only for the MS-SSIM size check
 * (the reference passes kernel_size=13, sigma=2.5, k2=0.05; k1 defaults to 0.01).  psnr/ssim/ms_ssim parity is unpinned
 * (torchmetrics absent here); see oracle/loss.py for the restated algorithm.
 * y, target: [B][H][W].  dev_out (XSD_LOSS_OUT = 12 device floats): [0] total, [1] l1, [2] poisson, [3] psnr, [4] ssim,
 * [5] ms_ssim (inactive terms 0), [6] mean squared error, [7] min(target), [8] max(target) (the last three when any of
 * l1 / poisson / psnr is active; they are the per-batch states the epoch-level metrics accumulate,
 * metrics/xmm_metric_collection.py:14-38), [9..11] reserved.  dev_dy_or_null receives d total / d y. */
#define XSD_LOSS_OUT 12
typedef struct xsd_loss_config {
    float w_l1, w_poisson, w_psnr, w_ssim, w_ms_ssim;
    float correction;
    float sigma, k1, k2;
    int32_t kernel_size;
} xsd_loss_config;
typedef struct xsd_loss_fn xsd_loss_fn;   /* the object create_loss returns; owns a device workspace */
int xsd_loss_create(const xsd_loss_config* cfg, xsd_loss_fn** out);
void xsd_loss_destroy(xsd_loss_fn* f);
int xsd_loss_eval(xsd_loss_fn* f, const float* dev_y, const float* dev_target, float* dev_dy_or_null, float* dev_out,
                  int B, int H, int W, void* stream);
/* Multi-channel images ([B][C][H][W] contiguous = B*C images of H x W): tell the loss how many consecutive images form one SAMPLE
 * (default 1).  Two terms reduce per sample, as the reference's metrics do: the Poisson term divides the element mean by the number
 * of samples (metrics/metrics.py:30-39: `self.total += preds.size()[0]`), and MS-SSIM averages every scale's statistic over a
 * sample's channels before the product over scales (torchmetrics: `.reshape(B, -1).mean(-1)` over C, H, W).  l1, psnr and ssim are
 * the same either way.  xsd_loss_eval then requires B to be a multiple of `channels`. */
int xsd_loss_set_channels(xsd_loss_fn* f, int channels);

/* torch.optim.Adam(lr, betas, eps=1e-8) single fused step over flat buffers (models/model.py:241-245).
 * step is 1-based; grad_scale multiplies the gradient on read (1/world_size for data-parallel mean). */
int xsd_adam_step(xsd_engine* e, float* dev_params, const float* dev_grads, float* dev_m, float* dev_v, int64_t n,
                  int step, float lr, float beta1, float beta2, float eps, float grad_scale, void* stream);

/* Input pipeline: counts (int32 or fp32, [B][Hin][Win]) * detector mask (uint8 {0,1} [Hin][Win] or NULL)
 * -> centred zero pad / crop to [B][res][res] -> optional Normalize.normalize_image(max_val, stretch)
 * (data/dataset.py:41-47; data/tools.py:103-126; transforms/normalize.py:66-82).
 * stretch: 0 linear, 1 sqrt, 2 asinh, 3 log.  do_normalize = 0 returns the masked, padded counts (bit-exact). */
int xsd_mask_pad_normalize(const void* dev_counts, int counts_is_int32, const uint8_t* dev_mask_or_null, float* dev_out,
                           int B, int Hin, int Win, int res, int do_normalize, float max_val, int stretch, void* stream);
/* Whole sample composition of XmmDataset (data/dataset.py:24-49 + :267-268) in one kernel, straight from FITS payload
 * words: img (+ agn) (+ background) summed in fp32 like load_fits' float images, * detector mask, optional nearest
 * upsample x s with / s^2 (ImageUpsample, dataset.py:44-45), centred pad / crop to res, optional normalize.
 * is_int32: BITPIX 32 counts (else IEEE float32); big_endian != 0: words are in FITS byte order (byte-swapped on load),
 * so a primary HDU's data block can be copied to the device unmodified.  Bit-exact when do_normalize = 0. */
int xsd_compose_input(const void* dev_img, const void* dev_agn_or_null, const void* dev_bkg_or_null, int is_int32, int big_endian,
                      const uint8_t* dev_mask_or_null, float* dev_out, int B, int Hin, int Win, int upsample, int res,
                      int do_normalize, float max_val, int stretch, void* stream);
/* Normalize.normalize_image (inverse = 0) / denormalize_image (inverse = 1), max_val > 0 (transforms/normalize.py:66-92) */
int xsd_normalize(const float* dev_in, float* dev_out, int64_t n, float max_val, int stretch, int inverse, void* stream);
/* ImageUpsample: nearest x scale then / scale^2 (transforms/imageupsample.py:10-26); in [N][H][W] */
int xsd_image_upsample(const float* dev_in, float* dev_out, int N, int H, int W, int scale, void* stream);

/* ---- measurement / test hooks ------------------------------------------------------------------------------- */
/* Per-kernel-class HIP-event timing of the kernels launched by this engine (bench.py roofline block), with each launch's
 * ALGORITHMIC flop and bytes (SURVEY.md 8d counting rule: every operand once).  MFMA-bound classes: 0 = conv (forward +
 * input-gradient), 1 = weight gradient.  HBM-bound classes (round 6): 2 = edge_expand (conv_first forward, conv_last input-gradient:
 * 1 -> 32 channels), 3 = edge_reduce (conv_last forward + skip + clamp, conv_first input-gradient: 32 -> 1), 4 = edge_wgrad
 * (weight gradients of the two edge layers), 5 = xsd_l1_loss, 6 = xsd_adam_step, 7 = clamp backward, 8 = max-|x| sweeps of
 * planes no producer reported (f16x3).  enable resets the counters; unknown classes read as zero launches. */
int xsd_profile_enable(xsd_engine* e, int enable);
int xsd_profile_read(xsd_engine* e, int klass, double* total_ms, int64_t* launches, double* total_flop, double* total_bytes);

/* What the matrix pipes of the current device SUSTAIN on the conv kernels' own MFMA stream (csrc/mfma_stream_probe.hip): eight
 * waves per CU issuing the v_mfma_f32_32x32x16_{f16 (fmt 0), bf16 (fmt 1)} sequence of one conv half-step with its LDS fragment
 * reads, on realistic split operands, no staging, no global traffic -- for `seconds` (0 < seconds <= 30) of back-to-back
 * launches; reported over the second half (the package-power governor has settled by then): dense 16-bit MFMA TFLOP/s and the
 * in-kernel shader clock in GHz (sclk_ghz may be NULL).  bench.py's `roofline.sustained_peak` = this rate / products per
 * multiply, measured in the bench process on the bench's device.  Blocks the host until done. */
int xsd_probe_mfma_stream(int fmt, double seconds, double* mfma_tflops, double* sclk_ghz, void* stream);

/* Diagnostic only: accumulated shader-cycle stamps of the kernels' phases (32 slots; read with enable = 0).  Conv:
 * [0] prologue, [1] prefetch issue, [2] MFMA loop, [3] epilogue, [4] wait+barrier, [5] split+LDS write+barrier, [6] items,
 * [7] s_memrealtime ticks, [8..12] staging wave, [13..15] youngest MFMA wave; weight gradient: [16] staging rounds,
 * [17] MFMA walk, [18] MFMA wave at the barrier, [19] staging wave at the barrier, [21] tiles. */
int xsd_debug_stamps(xsd_engine* e, int enable, unsigned long long* out32);
/* Diagnostic only (pure host arithmetic, no device needed): the number of workgroups a persistent split-mode conv launch over `ntiles`
 * tiles uses on a device of `ncu` compute units (csrc/xsd_kernels.h: persistent_grid -- the full grid, or the balanced one when the
 * launch has at most 8 rounds and its last round would fill at most 0.35 of the CUs). */
int xsd_debug_persistent_grid(int ntiles, int ncu);
/* Diagnostic only: hipOccupancyMaxActiveBlocksPerMultiprocessor of the split forward conv kernel at a dynamic LDS size. */
int xsd_debug_occupancy(int lds_bytes);
/* Diagnostic only: wall time (ms) of a grid of `grid` workgroups that each sleep `us` microseconds holding `lds_bytes` of LDS
 * (a census of how many such workgroups are resident at once). */
float xsd_debug_residency_ms(int grid, int threads, int lds_bytes, int us);

/* Single-layer entry points used by the kernel-level parity tests (one 3x3 conv over NHWC 32-channel planes).
 * dev_in: [n_in] plane pointers on the host (each plane [B][H][W][32]); w_oihw: device OIHW [32*n_out][32*n_in][3][3]. */
int xsd_test_conv3x3(xsd_engine* e, const float* const* host_in_planes, int n_in, const float* dev_w_oihw, const float* dev_bias,
                     float* const* host_out_planes, int n_out, float slope, int B, int H, int W, void* stream);
int xsd_test_conv3x3_bwd(xsd_engine* e, const float* const* host_in_planes, int n_in, const float* dev_w_oihw,
                         const float* dev_g_plane, float* const* host_dx_planes, float* dev_dw_oihw, float* dev_db,
                         int B, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XSD_H */
