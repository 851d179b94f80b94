// xsd_loss.h -- host launch interface of the loss kernels (loss_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace xsd {

constexpr int LOSS_RMAX = 12;        // gaussian radius limit (sigma <= 3.4); sigma = 2.5 -> R = 9, 19 taps
constexpr int LOSS_OUT_FLOATS = 12;  // result vector of launch_loss
constexpr int LOSS_MAX_SCALES = 5;   // MS-SSIM betas (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)

// effective weight of each term (relative percentage x paper scaling, utils/loss_functions.py:25-36), order
// l1, poisson, psnr, ssim, ms_ssim; correction is added when > 0 (loss_functions.py:44-45)
struct LossWeights {
    float w[5];
    float correction;
    float sigma, k1, k2;
    int kernel_size;                 // only used by the MS-SSIM size check, like torchmetrics
};

size_t loss_workspace_bytes(int B, int H, int W);
int loss_check(const LossWeights& w, int B, int H, int W, const char** why);
// out8 (device, LOSS_OUT_FLOATS floats): [0] total, [1..5] l1, poisson, psnr, ssim, ms_ssim (0 for inactive terms),
// [6] mean squared error, [7] min(target), [8] max(target) (written when any of l1/poisson/psnr is active); dy may be null
// B images of H x W = B / channels samples (NCHW folded): the Poisson term and MS-SSIM reduce per sample
hipError_t launch_loss(const LossWeights& w, const float* y, const float* t, float* dy, float* out8, int B, int channels, int H, int W,
                       void* workspace, hipStream_t s);

} // namespace xsd
