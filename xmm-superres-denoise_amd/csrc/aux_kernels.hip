// aux_kernels.hip -- the HBM-bound kernels around the MFMA convs: edge layers (Cin=1 / Cout=1), loss, Adam,
// weight repack, and the input transforms (detector mask, centred pad, Normalize, ImageUpsample).
// Each kernel cites the reference code it replaces.
#include "xsd_kernels.h"
#include "xsd_aux.h"
#include "xsd_split.h"

namespace xsd {

// max |x| of a workgroup into a slot with ONE global atomic (round 6: same-address atomics are served one after the other at the memory
// side -- 2,048 of them at the end of a launch cost 22 us, tools/launch_floor_probe.hip; these kernels used to issue one per WAVE of up to
// 4,096 workgroups).  Every thread of the workgroup must call it (it holds a barrier); NaN never wins fmaxf.
__device__ __forceinline__ void publish_block_max(float m, float* slot)
{
    __shared__ float wave_max[16];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = wave_max[0];
        for (int i = 1; i < (int)((blockDim.x + 63) >> 6); ++i) t = fmaxf(t, wave_max[i]);
        atomicMax(reinterpret_cast<unsigned int*>(slot), __float_as_uint(t));      // non-negative floats order as unsigned integers
    }
}

// torch.clamp propagates NaN (clamp(nan, 0, 1) = nan; +-inf go to the bounds); fminf / fmaxf alone return the non-NaN operand and
// would turn a NaN pixel into 0.  Identical to fminf(fmaxf(v, lo), hi) for every non-NaN v (round 6: tests/test_hip_abi_errors.py).
__device__ __forceinline__ float clamp_nan(float v, float lo, float hi) { return v != v ? v : fminf(fmaxf(v, lo), hi); }


typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------
// 1 -> 32 conv (conv_first forward, generator_rrdb.py:31-37,67; also the input-gradient of conv_last, :48-54).
// Thread = (pixel, 4-channel quad): 8 consecutive lanes write one pixel's 128 B, a wave writes 1 KiB contiguous.
// Algorithmic bytes/px: 4 read + 128 written.
// ---------------------------------------------------------------------------------------------------------------
constexpr int EE_STAGE = 6;      // staging loads in flight per thread: 1536 elements per batch = the three rows of a 510-pixel-wide image at once
__global__ __launch_bounds__(256) void edge_expand_kernel(const EdgeExpandParams P)
{
    // One image row per block iteration: the three rows of s around it are staged zero-padded in LDS, the thread's 9 x 4
    // weights live in registers, and the row is swept two pixels per thread at a time (no per-pixel division).
    // (Round 6 measured this kernel at 2.8 - 3.7 TB/s against the 6.85 TB/s a pure write stream reaches here (tools/hbm_stream_probe.py)
    // and tried two rewrites, same device, alternating: a 32-pixel chunk per iteration without LDS (nine loads, then the store): 2.2 TB/s,
    // one memory latency per 4 KiB; row bands of four with the next item's loads in flight: 3.2 / 4.0 TB/s at batch 32 / SR 1024^2 (+10 %)
    // but 110 instead of 92 us at batch 4 x 416 x 416, where a launch is 2.6 items per workgroup -- a loss at the reference's own batch for
    // 0.03 % of the bench step.  Not adopted: tools/attic/edge_expand_row_band_pipeline.patch, profiles/r06_ab_edge_expand.txt.)
    __shared__ float srow[3][EDGE_MAX_W + 2];
    const int q = threadIdx.x & 7, px0 = threadIdx.x >> 3;
    f32x4 w4[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) w4[tap] = *reinterpret_cast<const f32x4*>(P.w + tap * 32 + q * 4);
    const f32x4 b4 = P.bias ? *reinterpret_cast<const f32x4*>(P.bias + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int nrows = P.B * P.H;
    float vmax = 0.f;      // max |value written| by this thread (reported only when P.amax is set)
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int y = row % P.H;
        const float* sb = P.s + (P.s_bs ? (long long)(row / P.H) * P.s_bs : (long long)(row - y) * P.W);
        __syncthreads();
        // The three input rows into LDS, EE_STAGE loads per thread IN FLIGHT at once (round 6: the loop used to be load -> wait -> LDS write
        // per element, i.e. five memory latencies behind each other per 416-pixel row: 92 us for the 91 MB of a batch-4 x 416 x 416 plane)
        const int n3 = 3 * (P.W + 2);
        for (int base = 0; base < n3; base += 256 * EE_STAGE) {
            float tmp[EE_STAGE];
#pragma unroll
            for (int u = 0; u < EE_STAGE; ++u) {
                const int i = base + u * 256 + (int)threadIdx.x;
                const int k = i / (P.W + 2), xx = i - k * (P.W + 2) - 1, yy = y + k - 1;
                tmp[u] = (i < n3 && yy >= 0 && yy < P.H && xx >= 0 && xx < P.W) ? sb[(long long)yy * P.W + xx] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < EE_STAGE; ++u) {
                const int i = base + u * 256 + (int)threadIdx.x;
                const int k = i / (P.W + 2);
                if (i < n3) srow[k][i - k * (P.W + 2)] = tmp[u];
            }
        }
        __syncthreads();
        const long long rbase = (long long)row * P.W * 32;
        // the per-pixel side operand (the plane this launch adds to, or the mask plane: 16 B per thread and pixel) one iteration ahead
        const bool side = P.accumulate || (P.mask && !P.bits);
        const float* sp = P.accumulate ? P.out : P.mask;
        f32x4 nxt = {0.f, 0.f, 0.f, 0.f};
        if (side && px0 < P.W) nxt = *reinterpret_cast<const f32x4*>(sp + rbase + (long long)px0 * 32 + q * 4);
        for (int x = px0; x < P.W; x += 32) {
            const f32x4 cur = nxt;
            if (side && x + 32 < P.W) nxt = *reinterpret_cast<const f32x4*>(sp + rbase + (long long)(x + 32) * 32 + q * 4);
            f32x4 v = b4;
            if (P.accumulate) v += cur;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) v += srow[tap / 3][x + tap % 3] * w4[tap];
            const long long o = rbase + (long long)x * 32;
            if (P.bits) {
                const unsigned int m = (unsigned int)P.bits[((long long)row * P.W + x) * 2 + (q & 1)] >> (4 * (q >> 1));
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = ((m >> i) & 1u) ? v[i] : v[i] * P.mslope;
            } else if (P.mask) {
                const f32x4 m = P.accumulate ? *reinterpret_cast<const f32x4*>(P.mask + o + q * 4) : cur;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = m[i] > 0.f ? v[i] : v[i] * P.mslope;
            }
            *reinterpret_cast<f32x4*>(P.out + o + q * 4) = v;
#pragma unroll
            for (int i = 0; i < 4; ++i) vmax = fmaxf(vmax, fabsf(v[i]));
        }
    }
    if (P.amax) publish_block_max(vmax, P.amax);      // (P.amax is uniform: every thread arrives)
}

// ---------------------------------------------------------------------------------------------------------------
// 32 -> 1 conv (conv_last forward + DN skip + clamp, generator_rrdb.py:107-108,132-135 and model.py:49; also the
// input-gradient of conv_first).  8 lanes per pixel, each 4 channels x 9 taps, xor-shuffle reduction over the 8 lanes.
// Algorithmic bytes/px: 128 read (+4 skip) + 4..8 written.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void edge_reduce_kernel(const EdgeReduceParams P)
{
    // Block = (image, 32-pixel column strip, band of EDGE_BAND rows).  Thread = (column, 4-channel quad) walking down
    // the band with a three-row register window: every feature pixel is fetched 3 times (left / centre / right thread)
    // instead of 9, and no per-pixel division is needed.
    const int q = threadIdx.x & 7, px = threadIdx.x >> 3;
    f32x4 w4[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) w4[tap] = *reinterpret_cast<const f32x4*>(P.w + tap * 32 + q * 4);
    const float bias = P.bias ? P.bias[0] : 0.f;
    const int stripsX = (P.W + 31) / 32, bandsY = (P.H + EDGE_BAND - 1) / EDGE_BAND;
    const int nblk = P.B * stripsX * bandsY;
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int sx = blk % stripsX, by = (blk / stripsX) % bandsY, b = blk / (stripsX * bandsY);
        const int x = sx * 32 + px, y0 = by * EDGE_BAND;
        const float* fb = P.f + (long long)b * P.H * P.W * 32;
        const long long ob = P.y_bs ? (long long)b * P.y_bs : (long long)b * P.H * P.W;
        const long long sob = P.skip_bs ? (long long)b * P.skip_bs : (long long)b * P.H * P.W;
        auto load = [&](int yy, int xx) { // branch-free: out-of-image taps read the image's first pixel and are zeroed
            const bool ok = yy >= 0 && yy < P.H && xx >= 0 && xx < P.W;
            const float* pxp = fb + (ok ? ((long long)yy * P.W + xx) * 32 : 0);
            const f32x4 v = *reinterpret_cast<const f32x4*>(pxp + q * 4);
            return ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};      // a select, not a product with 0: the image's first pixel may hold a NaN / inf (0 * nan = nan)
        };
        f32x4 r0[3], r1[3], r2[3];   // rows y-1, y, y+1 at columns x-1, x, x+1
#pragma unroll
        for (int d = 0; d < 3; ++d) { r0[d] = load(y0 - 1, x - 1 + d); r1[d] = load(y0, x - 1 + d); }
        const int yend = y0 + EDGE_BAND < P.H ? y0 + EDGE_BAND : P.H;
        for (int y = y0; y < yend; ++y) {
#pragma unroll
            for (int d = 0; d < 3; ++d) r2[d] = load(y + 1, x - 1 + d);
            // the scalar side inputs of this output pixel travel with the feature loads (one memory latency per row, not two)
            const bool owner = q == 0 && x < P.W;
            const long long pix = ob + (long long)y * P.W + x;
            const float sk = (owner && P.skip) ? P.skip[sob + (long long)y * P.W + x] : 0.f;
            const float ad = (owner && P.addto) ? P.addto[pix] : 0.f;
            f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int d = 0; d < 3; ++d) a4 += r0[d] * w4[d] + r1[d] * w4[3 + d] + r2[d] * w4[6 + d];
            float acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
            acc += __shfl_xor(acc, 1);
            acc += __shfl_xor(acc, 2);
            acc += __shfl_xor(acc, 4);
            if (owner) {
                float v = acc + bias;
                if (P.skip) v += sk;
                if (P.addto) v += ad;
                if (P.pre) P.pre[pix] = v;
                if (P.clamp01) v = clamp_nan(v, 0.f, 1.f);
                P.y[pix] = v;
            }
#pragma unroll
            for (int d = 0; d < 3; ++d) { r0[d] = r1[d]; r1[d] = r2[d]; }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Edge weight gradients: out[tap][c] = sum_p f[p][c] * s[p+tap];  bsum[c] = sum_p f[p][c];  ssum = sum_p s[p].
//  conv_first: f = d(fea), s = x          -> dW[c][0][tap] = out[tap][c],   db[c] = bsum[c]
//  conv_last : f = trunk features, s = dy -> dW[0][c][tap] = out[8-tap][c], db[0] = ssum
// Deterministic two-stage reduction (per-block partials, then edge_wgrad_final_kernel).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void edge_wgrad_kernel(const EdgeWgradParams P)
{
    // One image row per block iteration: the three rows of s around it are staged (zero padded) in LDS, then 8 threads
    // per pixel (4 channels each) sweep the row.  No per-pixel division, 2 pixels in flight per thread.
    __shared__ float red[256 * 4];
    __shared__ float srow[3][EDGE_MAX_W + 2];
    const int q = threadIdx.x & 7, px0 = threadIdx.x >> 3;
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bs = {0.f, 0.f, 0.f, 0.f};
    float ss = 0.f;
    const int nrows = P.B * P.H;
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int y = row % P.H;
        const float* sb = P.s + (P.s_bs ? (long long)(row / P.H) * P.s_bs : (long long)(row - y) * P.W);     // image base
        __syncthreads();
        for (int i = threadIdx.x; i < 3 * (P.W + 2); i += 256) {
            const int k = i / (P.W + 2), xx = i - k * (P.W + 2) - 1, yy = y + k - 1;
            srow[k][xx + 1] = (yy >= 0 && yy < P.H && xx >= 0 && xx < P.W) ? sb[(long long)yy * P.W + xx] : 0.f;
        }
        __syncthreads();
        const float* frow = P.f + (long long)row * P.W * 32;
        auto load = [&](int x) {
            return *reinterpret_cast<const f32x4*>(frow + (long long)x * 32 + q * 4);
        };
        auto use = [&](int x, const f32x4& fv) {
            bs += fv;
            if (q == 0) ss += srow[1][x + 1];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) acc[tap] += fv * srow[tap / 3][x + tap % 3];
        };
        int x = px0;
        for (; x + 96 < P.W; x += 128) {
            const f32x4 f0 = load(x), f1 = load(x + 32), f2 = load(x + 64), f3 = load(x + 96);
            use(x, f0);
            use(x + 32, f1);
            use(x + 64, f2);
            use(x + 96, f3);
        }
        for (; x < P.W; x += 32) use(x, load(x));
    }
    float* outp = P.partial + (long long)blockIdx.x * (9 * 32 + 32 + 1);
    // reduce over the 32 threads of the block that share q
#pragma unroll
    for (int t = 0; t < 10; ++t) {
        const f32x4 v = t < 9 ? acc[t] : bs;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) red[threadIdx.x * 4 + i] = v[i];
        __syncthreads();
        if (threadIdx.x < 32) {
            const int qq = threadIdx.x >> 2, i = threadIdx.x & 3;
            float s = 0.f;
            for (int k = 0; k < 32; ++k) s += red[(k * 8 + qq) * 4 + i];
            outp[t * 32 + threadIdx.x] = s;
        }
    }
    __syncthreads();
    red[threadIdx.x] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int k = 0; k < 256; ++k) s += red[k];
        outp[320] = s;
    }
}

__global__ void edge_wgrad_final_kernel(const float* partial, int nblocks, int mode, float* dw, float* db, int cstride)
{
    // one 64-lane wave per output element: lane l sums blocks l, l+64, ... then a fixed-order shuffle tree
    const int e = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) s += (double)partial[(long long)b * 321 + e];
    for (int k = 32; k > 0; k >>= 1) s += __shfl_down(s, k);
    if (threadIdx.x != 0) return;
    if (mode == 0) { // conv_first: dW[c][0][tap], db[c]
        if (e < 288) dw[(e & 31) * cstride + (e >> 5)] = (float)s;      // cstride = 9 * in_channels: W[c][ch][tap], ch folded into dw
        else if (e < 320) db[e - 288] = (float)s;
    } else {         // conv_last: dW[0][c][8-tap], db[0] = ssum
        if (e < 288) dw[(e & 31) * 9 + (8 - (e >> 5))] = (float)s;
        else if (e == 320) db[0] = (float)s;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Output-side elementwise: clamp backward (torch.clamp passes the gradient iff 0 <= v <= 1; applied for both the
// generator's clamp and Model.forward's, generator_rrdb.py:108,135 + model.py:49) and the mean-L1 loss
// (torchmetrics MeanAbsoluteError / F.l1_loss, utils/loss_functions.py:16).
// ---------------------------------------------------------------------------------------------------------------
__global__ void clamp_bwd_kernel(const float* pre, const float* dy, float* dpre, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = pre[i];
        dpre[i] = (v >= 0.f && v <= 1.f) ? dy[i] : 0.f;
    }
}

__global__ __launch_bounds__(256) void l1_loss_kernel(const float* y, const float* t, float* dy, double* partial,
                                                      long long n, float inv_n)
{
    __shared__ double red[256];
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float d = y[i] - t[i];
        s += (double)fabsf(d);
        if (dy) dy[i] = d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ void l1_loss_final_kernel(const double* partial, int nblocks, float* loss, double inv_n)
{
    __shared__ double red[256];
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[b];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(red[0] * inv_n);
}

// ---------------------------------------------------------------------------------------------------------------
// torch.optim.Adam single step (model.py:241-245; lr/betas from res/configs/models.toml:7-8), fused over the flat
// parameter buffer.  gscale folds the data-parallel 1/world_size mean into the read of the gradient.
// ---------------------------------------------------------------------------------------------------------------
__global__ void adam_kernel(float* p, const float* g, float* m, float* v, long long n, float lr_over_bc1, float inv_sqrt_bc2,
                            float b1, float b2, float eps, float gscale)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        p[i] = p[i] - lr_over_bc1 * (mi / denom);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight repack: OIHW (state-dict layout) -> MFMA fragment-order panels.
//  fwd panel [n][s][tap][j][h][co][t]     = W[oc(n,co)][32s + 16h+4j+t][tap]
//  bwd panel [s][n][tap'][j][h][ci][t]    = W[oc(n,16h+4j+t)][32s + ci][8-tap']     (transposed + flipped: dgrad)
//  oc(n,c) = 32n + c, or 4c + n for the pixel-shuffle conv (so that chunk n = sub-pixel (i,j), generator_rrdb.py:95-97)
// ---------------------------------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(const float* params, const PackDesc* descs, float* fwd, float* bwd)
{
    const PackDesc d = descs[blockIdx.y];
    const int ns = d.cin / 32, nn = d.cout / 32;
    const long long total = (long long)ns * nn * PANEL_FLOATS;
    const float* W = params + d.src_w;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        int r = (int)(e % PANEL_FLOATS);
        const int panel = (int)(e / PANEL_FLOATS);
        const int t = r & 3; r >>= 2;
        const int c = r & 31; r >>= 5;
        const int h = r & 1; r >>= 1;
        const int j = r & 3; r >>= 2;
        const int tap = r;
        const int k = 16 * h + 4 * j + t;
        { // forward: panel = n*ns + s
            const int n = panel / ns, s = panel % ns;
            const int oc = d.shuffle ? (4 * (32 * (n % (nn / 4)) + c) + n / (nn / 4)) : (32 * n + c);   // shuffle: chunk n = sub-pixel * P + plane
            fwd[d.dst_fwd + e] = W[((long long)oc * d.cin + 32 * s + k) * 9 + tap];
        }
        { // dgrad: panel = s*nn + n
            const int s = panel / nn, n = panel % nn;
            const int oc = d.shuffle ? (4 * (32 * (n % (nn / 4)) + k) + n / (nn / 4)) : (32 * n + k);
            bwd[d.dst_bwd + e] = d.bwd_scale * W[((long long)oc * d.cin + 32 * s + c) * 9 + (8 - tap)];
        }
    }
}

// math mode 3 / 4 panels (conv3x3_s3x.hip "bf16x6"; source of split_panels_f16_kernel for "f16x3"): fp32 in the fragment order of the bf16 MFMA, [s2][tap][lane = h*32 + m][j]:
// row m is the output channel (input channel for the dgrad panels), the k index of element j is channel 16*s2 + 8h + j of
// the 32-channel K-chunk.  The conv kernel splits them into three bf16 terms on the way into LDS (streaming 4 B instead of
// 6 B per weight).  PANEL_FLOATS per panel, like the fp32 panels.
__global__ void pack_weights_s3_kernel(const float* params, const PackDesc* descs, float* fwd, float* bwd)
{
    const PackDesc d = descs[blockIdx.y];
    const int ns = d.cin / 32, nn = d.cout / 32;
    const long long total = (long long)ns * nn * PANEL_FLOATS;
    const float* W = params + d.src_w;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        int r = (int)(e % PANEL_FLOATS);
        const int panel = (int)(e / PANEL_FLOATS);
        const int j = r & 7; r >>= 3;
        const int m = r & 31; r >>= 5;
        const int h = r & 1; r >>= 1;
        const int tap = r % 9;
        const int s2 = r / 9;
        const int k = 16 * s2 + 8 * h + j;
        {
            const int n = panel / ns, s = panel % ns;
            const int oc = d.shuffle ? (4 * (32 * (n % (nn / 4)) + m) + n / (nn / 4)) : (32 * n + m);   // shuffle: chunk n = sub-pixel * P + plane
            fwd[d.dst_fwd + e] = W[((long long)oc * d.cin + 32 * s + k) * 9 + tap];
        }
        {
            const int s = panel / nn, n = panel % nn;
            const int oc = d.shuffle ? (4 * (32 * (n % (nn / 4)) + k) + n / (nn / 4)) : (32 * n + k);
            bwd[d.dst_bwd + e] = d.bwd_scale * W[((long long)oc * d.cin + 32 * s + m) * 9 + (8 - tap)];
        }
    }
}

// math mode 4 (f16x3): max |x| of a plane view / of a flat buffer into a slot (atomic max of the float bits; the slot is zeroed by the
// caller).  Used for planes written by kernels that do not report it themselves (edge layers, test hooks) and for the packed
// weight buffers; the consumers scale their operands into the fp16 range with it (xsd_split.h, scale_for_amax).
__global__ void plane_amax_kernel(PlaneIn v, int B, int H, int W, float* slot)
{
    const long long total = (long long)B * H * W * 8;        // (pixel, channel quad)
    float m = 0.f;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(g & 7);
        long long pix = g >> 3;
        const int x = (int)(pix % W); pix /= W;
        const int y = (int)(pix % H);
        const int b = (int)(pix / H);
        const f32x4 t = *reinterpret_cast<const f32x4*>(v.p + (long long)b * v.bs + (long long)y * v.rs + (long long)x * v.ps + q * 4);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(t[0]), fabsf(t[1]))), fmaxf(fabsf(t[2]), fabsf(t[3])));
    }
    publish_block_max(m, slot);
}
__global__ void buffer_amax_kernel(const float* v, long long n, float* slot)
{
    float m = 0.f;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(v[g]));
    publish_block_max(m, slot);
}

// math mode 4 (f16x3): fp32 fragment-order panels [panel][s2][tap][lane][8 floats] (pack_weights_s3_kernel) -> the two-term fp16
// image the conv kernel copies into LDS as it is: [panel][s2][tap][term h | l][lane][8 x f16], the same 36,864 B per panel.
// One scale for the whole buffer (its max |w|, slot written by buffer_amax_kernel); thread = (panel, s2, tap, lane).
__global__ void split_panels_f16_kernel(const float* src, unsigned int* dst, long long nfrag /* panels * 2 * 9 * 64 */, const float* amax)
{
    float inv;
    const float sw = scale_for_amax(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, *amax))), inv);
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < nfrag; g += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(g & 63);
        const long long blk = g >> 6;                           // (panel, s2, tap)
        const f32x4 a = *reinterpret_cast<const f32x4*>(src + g * 8), b = *reinterpret_cast<const f32x4*>(src + g * 8 + 4);
        split_u32x2 ha, la, hb, lb;
        split2_f16x4(a, sw, ha, la);
        split2_f16x4(b, sw, hb, lb);
        unsigned int* d = dst + blk * 512 + lane * 4;           // 2 terms x 1 KiB per block = 512 words
        d[0] = ha[0]; d[1] = ha[1]; d[2] = hb[0]; d[3] = hb[1];
        d[256 + 0] = la[0]; d[256 + 1] = la[1]; d[256 + 2] = lb[0]; d[256 + 3] = lb[1];
    }
}

// edge-layer weights: conv_first W[c][0][tap] -> [tap][c] (forward) ; conv_last W[0][c][tap] -> [tap][c] (forward)
// and the flipped forms used by their input-gradients.
__global__ void pack_edge_kernel(const float* w_first, const float* w_last, float* first_fwd, float* first_bwd,
                                 float* last_fwd, float* last_bwd, int first_cstride)
{
    // first_cstride: floats between consecutive output channels of conv_first = 9 * in_channels (w_first points at the
    // input channel being packed); either side may be null (in_channels and out_channels differ)
    const int e = threadIdx.x + blockIdx.x * blockDim.x;
    if (e >= 288) return;
    const int tap = e >> 5, c = e & 31;
    if (w_first) {
        first_fwd[e] = w_first[c * first_cstride + tap];       // out[p][c] += x[p+tap] * W[c][ch][tap]
        first_bwd[e] = w_first[c * first_cstride + (8 - tap)]; // dx[p] += sum_c g[p+tap'][c] * W[c][ch][8-tap']
    }
    if (w_last) {
        last_fwd[e] = w_last[c * 9 + tap];         // y[p] += f[p+tap][c] * W[co][c][tap]
        last_bwd[e] = w_last[c * 9 + (8 - tap)];   // dT[p][c] += dy[p+tap'] * W[co][c][8-tap']
    }
}

// bias of the pixel-shuffle conv (4 * 32 P outputs) in chunk order: chunk n = sub-pixel * P + plane q holds the channels
// 4 (32 q + c) + sub-pixel:  out[n*32 + c] = b[4 (32 q + c) + sub]
__global__ void pack_shuffle_bias_kernel(const float* b, float* out, int P)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 128 * P) return;
    const int n = e >> 5, c = e & 31;
    out[e] = b[4 * (32 * (n % P) + c) + n / P];
}

// ---------------------------------------------------------------------------------------------------------------
// Input transforms
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float stretch_fwd(float v, int mode)
{
    switch (mode) {
    case 1: return sqrtf(v);                                        // torch.sqrt     (normalize.py:56-57)
    case 2: return asinhf(v / 0.02f) / asinhf(1.0f / 0.02f);        // _asinh         (normalize.py:4-11)
    case 3: return logf(1000.f * v + 1.f) / logf(1000.f);           // _log           (normalize.py:23-26)
    default: return v;
    }
}
__device__ __forceinline__ float stretch_inv(float v, int mode)
{
    switch (mode) {
    case 1: return v * v;
    case 2: return 0.02f * sinhf(v * asinhf(1.0f / 0.02f));         // _asinh_inv     (normalize.py:14-20)
    case 3: return (powf(1000.f, v) - 1.f) / 1000.f;                // _log_inv       (normalize.py:29-32)
    default: return v;
    }
}

// counts (int32 or float32) [B][Hin][Win] (* mask uint8 [Hin][Win]) -> centred zero pad/crop -> [B][res][res]
// -> optional Normalize.  Mask multiply by exactly 0/1 is a select, so the masked/padded image is bit-exact
// (data/dataset.py:41-47, data/tools.py:103-126, transforms/normalize.py:66-82).
__device__ __forceinline__ float mpn_load(const MaskPadParams& P, const void* img, long long idx)
{
    unsigned int w = reinterpret_cast<const unsigned int*>(img)[idx];
    if (P.big_endian) w = __builtin_bswap32(w);
    return P.counts_i32 ? (float)(int)w : __builtin_bit_cast(float, w);
}
__global__ void mask_pad_normalize_kernel(MaskPadParams P)
{
    const int s = P.upsample > 1 ? P.upsample : 1;
    const int Hs = P.Hin * s, Ws = P.Win * s; // size after the optional nearest upsample
    const long long total = (long long)P.B * P.res * P.res;
    const void* img0 = P.counts_i32 ? (const void*)P.counts_i32 : (const void*)P.counts_f32;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % P.res);
        const long long r = i / P.res;
        const int oy = (int)(r % P.res);
        const int b = (int)(r / P.res);
        const int uy = oy - P.y_top, ux = ox - P.x_left;
        float v = 0.f;
        if (uy >= 0 && uy < Hs && ux >= 0 && ux < Ws) {
            const int iy = uy / s, ix = ux / s;
            const long long src = ((long long)b * P.Hin + iy) * P.Win + ix;
            v = mpn_load(P, img0, src);
            if (P.extra1) v += mpn_load(P, P.extra1, src);
            if (P.extra2) v += mpn_load(P, P.extra2, src);
            if (P.mask) v = v * (float)P.mask[(long long)iy * P.Win + ix];
            if (s > 1) v = v / (float)(s * s);
        }
        if (P.do_norm) {
            v = clamp_nan(v, 0.f, P.max_val) / P.max_val;
            v = stretch_fwd(v, P.mode);
            v = clamp_nan(v, 0.f, 1.f);
        }
        P.out[i] = v;
    }
}

// Normalize.normalize_image with max_val > 0 / denormalize_image (normalize.py:66-92), elementwise.
__global__ void normalize_kernel(const float* in, float* out, long long n, float max_val, int mode, int inverse)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = in[i];
        if (!inverse) {
            v = clamp_nan(v, 0.f, max_val) / max_val;
            v = stretch_fwd(v, mode);
            v = clamp_nan(v, 0.f, 1.f);
        } else {
            v = max_val * stretch_inv(v, mode);
            v = clamp_nan(v, 0.f, max_val);
        }
        out[i] = v;
    }
}

// ImageUpsample: nearest x s then / s^2 (transforms/imageupsample.py:10-26); in [N][H][W] -> out [N][sH][sW]
__global__ void upsample_nearest_kernel(const float* in, float* out, int N, int H, int W, int s)
{
    const long long total = (long long)N * H * s * W * s;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % (W * s));
        const long long r = i / (W * s);
        const int oy = (int)(r % (H * s));
        const int nimg = (int)(r / (H * s));
        out[i] = in[((long long)nimg * H + oy / s) * W + ox / s] / (float)(s * s);
    }
}

// diagnostic: residency census.  Every workgroup sleeps ~`us` microseconds; with R workgroups resident per CU a grid of
// G workgroups takes ceil(G / (R * CUs)) * us.
__global__ void sleep_kernel(int us, int* sink)
{
    extern __shared__ char dyn[];
    const long long t0 = wall_clock64(); // 100 MHz
    while (wall_clock64() - t0 < (long long)us * 100) __builtin_amdgcn_s_sleep(32);
    if (us < 0) sink[0] = dyn[threadIdx.x];
}
float debug_residency_ms(int grid, int threads, int lds_bytes, int us)
{
    hipFuncSetAttribute(reinterpret_cast<const void*>(&sleep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(sleep_kernel, dim3(grid), dim3(threads), lds_bytes, 0, 10, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(sleep_kernel, dim3(grid), dim3(threads), lds_bytes, 0, us, nullptr);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms;
}

// ---------------------------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------------------------
static inline int grid_for(long long n, int threads, int cap = 2048)
{
    long long g = (n + threads - 1) / threads;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

hipError_t launch_edge_expand(const EdgeExpandParams& p, hipStream_t s)
{
    // the kernel reports the maximum of every value it WRITES: with `accumulate` those include partial sums of a plane other
    // launches complete, which is not the plane's max |x| -- no caller combines the two, and none may
    if (p.amax && p.accumulate) return hipErrorInvalidValue;
    const long long rows = (long long)p.B * p.H;
    // 768 workgroups = the resident set (three per CU at 49 KB of LDS each); every workgroup publishes one atomic when it ends
    hipLaunchKernelGGL(edge_expand_kernel, dim3((unsigned)(rows < 768 ? rows : 768)), dim3(256), 0, s, p);
    return hipGetLastError();
}
hipError_t launch_edge_reduce(const EdgeReduceParams& p, hipStream_t s)
{
    const long long nblk = (long long)p.B * ((p.W + 31) / 32) * ((p.H + EDGE_BAND - 1) / EDGE_BAND);
    hipLaunchKernelGGL(edge_reduce_kernel, dim3((unsigned)(nblk < 8192 ? nblk : 8192)), dim3(256), 0, s, p);
    return hipGetLastError();
}
hipError_t launch_edge_wgrad(const EdgeWgradParams& p, int mode, float* dw, float* db, hipStream_t s, int cstride)
{
    hipLaunchKernelGGL(edge_wgrad_kernel, dim3(p.nblocks), dim3(256), 0, s, p);
    hipLaunchKernelGGL(edge_wgrad_final_kernel, dim3(321), dim3(64), 0, s, p.partial, p.nblocks, mode, dw, db, cstride);
    return hipGetLastError();
}
// dst[b][p] += sum over the C channels of src[b][c][p]: the skip gradient of GeneratorRRDB_DN's `out + x` when a one-channel x
// broadcasts over several output channels (generator_rrdb.py:134)
__global__ void add_channels_kernel(float* dst, const float* src, int C, long long HW, int B)
{
    const long long total = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / HW, p = i - b * HW;
        float t = 0.f;
        for (int c = 0; c < C; ++c) t += src[(b * C + c) * HW + p];
        dst[i] += t;
    }
}
hipError_t launch_add_channels(float* dst, const float* src, int C, long long HW, int B, hipStream_t s)
{
    hipLaunchKernelGGL(add_channels_kernel, dim3(grid_for((long long)B * HW, 256)), dim3(256), 0, s, dst, src, C, HW, B);
    return hipGetLastError();
}
hipError_t launch_clamp_bwd(const float* pre, const float* dy, float* dpre, long long n, hipStream_t s)
{
    hipLaunchKernelGGL(clamp_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, pre, dy, dpre, n);
    return hipGetLastError();
}
hipError_t launch_l1_loss(const float* y, const float* t, float* dy, double* partial, int nblocks, float* loss,
                          long long n, hipStream_t s)
{
    hipLaunchKernelGGL(l1_loss_kernel, dim3(nblocks), dim3(256), 0, s, y, t, dy, partial, n, 1.0f / (float)n);
    hipLaunchKernelGGL(l1_loss_final_kernel, dim3(1), dim3(256), 0, s, partial, nblocks, loss, 1.0 / (double)n);
    return hipGetLastError();
}
hipError_t launch_adam(float* p, const float* g, float* m, float* v, long long n, int step, float lr, float b1, float b2,
                       float eps, float gscale, hipStream_t s)
{
    const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, p, g, m, v, n, (float)(lr / bc1),
                       (float)(1.0 / sqrt(bc2)), b1, b2, eps, gscale);
    return hipGetLastError();
}
hipError_t launch_pack_weights(const float* params, const PackDesc* descs_dev, int ndesc, float* fwd, float* bwd,
                               hipStream_t s)
{
    hipLaunchKernelGGL(pack_weights_kernel, dim3(36, ndesc), dim3(256), 0, s, params, descs_dev, fwd, bwd);
    return hipGetLastError();
}
hipError_t launch_pack_weights_s3(const float* params, const PackDesc* descs_dev, int ndesc, float* fwd, float* bwd,
                                  hipStream_t s)
{
    hipLaunchKernelGGL(pack_weights_s3_kernel, dim3(36, ndesc), dim3(256), 0, s, params, descs_dev, fwd, bwd);
    return hipGetLastError();
}
hipError_t launch_plane_amax(const PlaneIn& v, int B, int H, int W, float* slot, hipStream_t s)
{
    hipLaunchKernelGGL(plane_amax_kernel, dim3(grid_for((long long)B * H * W * 8, 256, 1024)), dim3(256), 0, s, v, B, H, W, slot);
    return hipGetLastError();
}
hipError_t launch_buffer_amax(const float* v, long long n, float* slot, hipStream_t s)
{
    hipLaunchKernelGGL(buffer_amax_kernel, dim3(grid_for(n, 256, 256)), dim3(256), 0, s, v, n, slot);
    return hipGetLastError();
}
hipError_t launch_split_panels_f16(const float* src, void* dst, long long nfloats, const float* amax, hipStream_t s)
{
    const long long nfrag = nfloats / 8;
    hipLaunchKernelGGL(split_panels_f16_kernel, dim3(grid_for(nfrag, 256)), dim3(256), 0, s, src, reinterpret_cast<unsigned int*>(dst), nfrag, amax);
    return hipGetLastError();
}
hipError_t launch_pack_edge(const float* w_first, const float* w_last, float* ff, float* fb, float* lf, float* lb,
                            hipStream_t s, int first_cstride)
{
    hipLaunchKernelGGL(pack_edge_kernel, dim3(2), dim3(256), 0, s, w_first, w_last, ff, fb, lf, lb, first_cstride);
    return hipGetLastError();
}
hipError_t launch_pack_shuffle_bias(const float* b, float* out, int planes, hipStream_t s)
{
    hipLaunchKernelGGL(pack_shuffle_bias_kernel, dim3(planes), dim3(128), 0, s, b, out, planes);
    return hipGetLastError();
}
hipError_t launch_mask_pad_normalize(const MaskPadParams& p, hipStream_t s)
{
    const long long total = (long long)p.B * p.res * p.res;
    hipLaunchKernelGGL(mask_pad_normalize_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, p);
    return hipGetLastError();
}
hipError_t launch_normalize(const float* in, float* out, long long n, float max_val, int mode, int inverse, hipStream_t s)
{
    hipLaunchKernelGGL(normalize_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, in, out, n, max_val, mode, inverse);
    return hipGetLastError();
}
hipError_t launch_upsample_nearest(const float* in, float* out, int N, int H, int W, int sc, hipStream_t s)
{
    const long long total = (long long)N * H * sc * W * sc;
    hipLaunchKernelGGL(upsample_nearest_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, in, out, N, H, W, sc);
    return hipGetLastError();
}

} // namespace xsd
