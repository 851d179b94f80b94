// xsd_aux.h -- host launch prototypes of the engine's kernels + transform parameter blocks.
#pragma once
#include "xsd_kernels.h"

namespace xsd {

struct MaskPadParams {
    const int32_t* counts_i32; // exactly one of counts_i32 / counts_f32 is non-null
    const float* counts_f32;
    const uint8_t* mask;       // [Hin][Win] of {0,1} or null
    float* out;                // [B][res][res]
    int B, Hin, Win, res;
    int y_top, x_left;         // floor((res-Hin)/2), floor((res-Win)/2)  (data/tools.py:111-116); negative = crop
    int do_norm, mode;         // mode: 0 linear, 1 sqrt, 2 asinh, 3 log
    float max_val;
    // optional extra count images summed before the mask (img += agn; img += background, data/dataset.py:33-39);
    // same dtype/shape/endianness as the main image
    const void* extra1;
    const void* extra2;
    int big_endian;            // int32/float32 words are FITS big-endian: byte-swap on load
    int upsample;              // nearest x s then / s^2 before the pad (ImageUpsample, dataset.py:44-45); 1 = none
};

hipError_t launch_conv3x3_mfma(const ConvParams& p, hipStream_t stream);
hipError_t launch_wgrad_mfma(const WgradParams& p, hipStream_t stream);
hipError_t launch_wgrad_reduce(const WgradReduceParams& r, hipStream_t stream);
hipError_t launch_wgrad_reduce_multi(const WgradReduceBatch& rb, hipStream_t stream);
hipError_t launch_edge_expand(const EdgeExpandParams& p, hipStream_t s);
hipError_t launch_edge_reduce(const EdgeReduceParams& p, hipStream_t s);
hipError_t launch_add_channels(float* dst, const float* src, int C, long long HW, int B, hipStream_t s);
hipError_t launch_edge_wgrad(const EdgeWgradParams& p, int mode, float* dw, float* db, hipStream_t s, int cstride = 9);
hipError_t launch_clamp_bwd(const float* pre, const float* dy, float* dpre, long long n, hipStream_t s);
hipError_t launch_l1_loss(const float* y, const float* t, float* dy, double* partial, int nblocks, float* loss,
                          long long n, hipStream_t s);
hipError_t launch_adam(float* p, const float* g, float* m, float* v, long long n, int step, float lr, float b1, float b2,
                       float eps, float gscale, hipStream_t s);
hipError_t launch_pack_weights(const float* params, const PackDesc* descs_dev, int ndesc, float* fwd, float* bwd,
                               hipStream_t s);
hipError_t launch_pack_weights_s3(const float* params, const PackDesc* descs_dev, int ndesc, float* fwd, float* bwd,
                                  hipStream_t s);
hipError_t launch_conv3x3_s3x(const ConvParams& p, hipStream_t stream);
hipError_t launch_wgrad_s3x(const WgradParams& p, hipStream_t stream);
hipError_t launch_wgrad_h2x(const WgradParams& p, hipStream_t stream);
hipError_t launch_plane_amax(const PlaneIn& v, int B, int H, int W, float* slot, hipStream_t s);   // math mode 4
hipError_t launch_buffer_amax(const float* v, long long n, float* slot, hipStream_t s);
hipError_t launch_split_panels_f16(const float* src, void* dst, long long nfloats, const float* amax, hipStream_t s);
hipError_t launch_conv3x3_h2x(const ConvParams& p, hipStream_t stream);
hipError_t launch_pack_edge(const float* w_first, const float* w_last, float* ff, float* fb, float* lf, float* lb,
                            hipStream_t s, int first_cstride = 9);
hipError_t launch_mask_pad_normalize(const MaskPadParams& p, hipStream_t s);
hipError_t launch_normalize(const float* in, float* out, long long n, float max_val, int mode, int inverse, hipStream_t s);
hipError_t launch_upsample_nearest(const float* in, float* out, int N, int H, int W, int sc, hipStream_t s);

} // namespace xsd
