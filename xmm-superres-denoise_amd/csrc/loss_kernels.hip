// loss_kernels.hip -- the training loss of the reference on one-channel images [B][H][W].
//
// Replaces create_loss (xmm_superres_denoise/utils/loss_functions.py:11-47): a weighted sum of torchmetrics metrics
// evaluated per batch (models/model.py:78), with the paper's scaling/correction constants
// (res/configs/loss_functions.toml:5-42):  l1 (MeanAbsoluteError), poisson (metrics/metrics.py:30-39), psnr
// (PeakSignalNoiseRatio), ssim / ms_ssim (kernel_size=13, sigma=2.5, k2=0.05).  The SSIM arithmetic follows the
// published torchmetrics 1.x algorithm (parity unpinned: torchmetrics is not installed):
// 19-tap gaussian for sigma 2.5 over the valid interior, data_range taken from the current scale's images (a
// differentiable function of the prediction), variances clamped at 0, relu-normalised contrast terms, five scales with
// 2x2 average pooling, per-image product, batch mean.
//
// All of it is HBM-bound stencil / reduction work on one-channel images (about 1/1000 of the conv stack's bytes), so
// the kernels are plain LDS-tiled separable filters with deterministic two-stage reductions (no atomics, no host
// synchronisation: every scalar the next kernel needs stays on the device).
#include "xsd_loss.h"

namespace xsd {

namespace {

constexpr int TS = 32;                 // outputs per tile edge
constexpr int TW = TS + 2 * LOSS_RMAX; // 56: tile edge incl. filter support
constexpr int NT = 256;

// per-scale scalars (device)
struct ScaleScal {
    float pmin, pmax, tmin, tmax;
    float dr, c1, c2;
    int from_p;          // data_range comes from the prediction -> it carries gradient
    float nmax, nmin;    // tie counts of pmax / pmin
    float ddr;           // d loss / d data_range (backward)
    float pad_;
};

__device__ __forceinline__ double block_sum(double v, double* red)
{
    red[threadIdx.x] = v;
    __syncthreads();
    for (int k = NT / 2; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

// ---- min / max of both images, then data_range, c1, c2 --------------------------------------------------------------
__global__ __launch_bounds__(NT) void minmax_kernel(const float* p, const float* t, long long n, float* partial)
{
    __shared__ float red[4][NT];
    float a = INFINITY, b = -INFINITY, c = INFINITY, d = -INFINITY;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        const float x = p[i], y = t[i];
        a = fminf(a, x); b = fmaxf(b, x); c = fminf(c, y); d = fmaxf(d, y);
    }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b; red[2][threadIdx.x] = c; red[3][threadIdx.x] = d;
    __syncthreads();
    for (int k = NT / 2; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) {
            red[0][threadIdx.x] = fminf(red[0][threadIdx.x], red[0][threadIdx.x + k]);
            red[1][threadIdx.x] = fmaxf(red[1][threadIdx.x], red[1][threadIdx.x + k]);
            red[2][threadIdx.x] = fminf(red[2][threadIdx.x], red[2][threadIdx.x + k]);
            red[3][threadIdx.x] = fmaxf(red[3][threadIdx.x], red[3][threadIdx.x + k]);
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) partial[blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}
__global__ __launch_bounds__(NT) void minmax_final_kernel(const float* partial, int nblocks, ScaleScal* sc, float k1, float k2)
{
    __shared__ float red[4][NT];
    float a = INFINITY, b = -INFINITY, c = INFINITY, d = -INFINITY;
    for (int i = threadIdx.x; i < nblocks; i += NT) {
        a = fminf(a, partial[i * 4]); b = fmaxf(b, partial[i * 4 + 1]);
        c = fminf(c, partial[i * 4 + 2]); d = fmaxf(d, partial[i * 4 + 3]);
    }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b; red[2][threadIdx.x] = c; red[3][threadIdx.x] = d;
    __syncthreads();
    for (int k = NT / 2; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) {
            red[0][threadIdx.x] = fminf(red[0][threadIdx.x], red[0][threadIdx.x + k]);
            red[1][threadIdx.x] = fmaxf(red[1][threadIdx.x], red[1][threadIdx.x + k]);
            red[2][threadIdx.x] = fminf(red[2][threadIdx.x], red[2][threadIdx.x + k]);
            red[3][threadIdx.x] = fmaxf(red[3][threadIdx.x], red[3][threadIdx.x + k]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ScaleScal s;
        s.pmin = red[0][0]; s.pmax = red[1][0]; s.tmin = red[2][0]; s.tmax = red[3][0];
        const float rp = s.pmax - s.pmin, rt = s.tmax - s.tmin;
        s.from_p = rp >= rt;                  // python max(a, b) keeps a on ties
        s.dr = s.from_p ? rp : rt;
        s.c1 = (k1 * s.dr) * (k1 * s.dr);
        s.c2 = (k2 * s.dr) * (k2 * s.dr);
        s.nmax = s.nmin = 1.f; s.ddr = 0.f; s.pad_ = 0.f;
        *sc = s;
    }
}
__global__ __launch_bounds__(NT) void ties_kernel(const float* p, long long n, const ScaleScal* sc, double* partial)
{
    __shared__ double red[NT];
    const float hi = sc->pmax, lo = sc->pmin;
    double a = 0, b = 0;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        const float x = p[i];
        a += x == hi; b += x == lo;
    }
    a = block_sum(a, red);
    b = block_sum(b, red);
    if (threadIdx.x == 0) { partial[blockIdx.x * 2] = a; partial[blockIdx.x * 2 + 1] = b; }
}

// ---- separable gaussian over an LDS tile -----------------------------------------------------------------------------
struct Taps { float g[2 * LOSS_RMAX + 1]; int R; };

// ---- SSIM statistics per valid output pixel; forward sums or backward gradient maps ---------------------------------
struct SsimParams {
    const float* p; const float* t;
    int B, H, W, Hv, Wv, tilesX, tilesY;
    const ScaleScal* sc;
    double* partial;     // [B*tiles][2]: forward (sum ssim, sum cs); backward (sum d/dc1, sum d/dc2)
    const float* gup;    // backward: [B][2] upstream (d loss / d sim_b, d loss / d cs_b)
    float* maps;         // backward: [3][B][Hv][Wv]  d/d mu_p, d/d E[pp], d/d E[pt]
    Taps taps;
};

template <bool BWD>
__global__ __launch_bounds__(NT) void ssim_kernel(const SsimParams P)
{
    __shared__ float sp[TW][TW + 1];
    __shared__ float st[TW][TW + 1];
    __shared__ float hz[5][TW][TS];
    __shared__ double red[NT];
    const int R2 = 2 * P.taps.R, ext = TS + R2;
    const int b = blockIdx.z, oy0 = blockIdx.y * TS, ox0 = blockIdx.x * TS;
    const float* pb = P.p + (long long)b * P.H * P.W;
    const float* tb = P.t + (long long)b * P.H * P.W;
    for (int i = threadIdx.x; i < ext * ext; i += NT) {
        const int r = i / ext, c = i - r * ext;
        const int y = oy0 + r, x = ox0 + c;
        const bool in = y < P.H && x < P.W;
        sp[r][c] = in ? pb[(long long)y * P.W + x] : 0.f;
        st[r][c] = in ? tb[(long long)y * P.W + x] : 0.f;
    }
    __syncthreads();
    // Variances and the covariance are shift invariant: the moments are taken about the tile's centre pixel (cp, ct), which
    // removes the E[x^2] - mu^2 cancellation on nearly flat images (the coarse MS-SSIM scales); the means get it added back.
    const float cp = sp[ext / 2][ext / 2], ct = st[ext / 2][ext / 2];
    for (int i = threadIdx.x; i < ext * TS; i += NT) {
        const int r = i / TS, c = i - r * TS;
        float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
        for (int k = 0; k <= R2; ++k) {
            const float g = P.taps.g[k], x = sp[r][c + k] - cp, y = st[r][c + k] - ct;
            a0 += g * x; a1 += g * y; a2 += g * (x * x); a3 += g * (y * y); a4 += g * (x * y);
        }
        hz[0][r][c] = a0; hz[1][r][c] = a1; hz[2][r][c] = a2; hz[3][r][c] = a3; hz[4][r][c] = a4;
    }
    __syncthreads();
    const float c1 = P.sc->c1, c2 = P.sc->c2;
    const long long npix = (long long)P.Hv * P.Wv;
    float g_sim = 0.f, g_cs = 0.f;
    if (BWD) { g_sim = P.gup[b * 2] / (float)npix; g_cs = P.gup[b * 2 + 1] / (float)npix; }
    double acc0 = 0, acc1 = 0;
    const int c = threadIdx.x & (TS - 1);
    for (int y = threadIdx.x / TS; y < TS; y += NT / TS) {
        float mp = 0, mt = 0, epp = 0, ett = 0, ept = 0;
        for (int k = 0; k <= R2; ++k) {
            const float g = P.taps.g[k];
            mp += g * hz[0][y + k][c]; mt += g * hz[1][y + k][c]; epp += g * hz[2][y + k][c];
            ett += g * hz[3][y + k][c]; ept += g * hz[4][y + k][c];
        }
        const int oy = oy0 + y, ox = ox0 + c;
        if (oy >= P.Hv || ox >= P.Wv) continue;
        const float vp_raw = epp - mp * mp, vt_raw = ett - mt * mt;   // moments about (cp, ct)
        const float vp = fmaxf(vp_raw, 0.f), vt = fmaxf(vt_raw, 0.f);
        const float cov = ept - mp * mt;
        mp += cp; mt += ct;                                           // the taps sum to 1
        const float U = 2.f * cov + c2, L = vp + vt + c2;
        const float A = 2.f * mp * mt + c1, Bq = mp * mp + mt * mt + c1;
        const float cs = U / L, lum = A / Bq;
        if (!BWD) {
            acc0 += (double)(lum * cs);
            acc1 += (double)cs;
        } else {
            const float d_cs = g_cs + g_sim * lum;
            const float d_lum = g_sim * cs;
            const float dA = d_lum / Bq, dB = -d_lum * A / (Bq * Bq);
            const float dU = d_cs / L, dL = -d_cs * U / (L * L);
            const float d_cov = 2.f * dU;
            const float d_vp = vp_raw > 0.f ? dL : 0.f;
            const float d_mp = dA * 2.f * mt + dB * 2.f * mp - d_cov * mt - d_vp * 2.f * mp;
            const long long o = ((long long)b * P.Hv + oy) * P.Wv + ox, plane = (long long)P.B * npix;
            P.maps[o] = d_mp;
            P.maps[plane + o] = d_vp;
            P.maps[2 * plane + o] = d_cov;
            // d/dc1 = dA + dB and d/dc2 = dU + dL, written without the cancellation of their nearly opposite terms:
            // dA + dB = d_lum (Bq - A) / Bq^2 with Bq - A = (mu_p - mu_t)^2;  dU + dL = d_cs (L - U) / L^2
            const float dm = mp - mt;
            acc0 += (double)(d_lum * (dm * dm) / (Bq * Bq));
            acc1 += (double)(d_cs * (vp + vt - 2.f * cov) / (L * L));
        }
    }
    acc0 = block_sum(acc0, red);
    acc1 = block_sum(acc1, red);
    if (threadIdx.x == 0) {
        const long long tile = ((long long)b * P.tilesY + blockIdx.y) * P.tilesX + blockIdx.x;
        P.partial[tile * 2] = acc0;
        P.partial[tile * 2 + 1] = acc1;
    }
}

// per-image means of the forward partials: stat[b][0] = mean ssim, stat[b][1] = mean cs
__global__ __launch_bounds__(NT) void image_stat_kernel(const double* partial, int tiles, double inv_n, double* stat)
{
    __shared__ double red[NT];
    const int b = blockIdx.x;
    double a = 0, c = 0;
    for (int i = threadIdx.x; i < tiles; i += NT) { a += partial[((long long)b * tiles + i) * 2]; c += partial[((long long)b * tiles + i) * 2 + 1]; }
    a = block_sum(a, red);
    c = block_sum(c, red);
    if (threadIdx.x == 0) { stat[b * 2] = a * inv_n; stat[b * 2 + 1] = c * inv_n; }
}

// backward scalars of one scale: d loss / d data_range from the c1/c2 partials, tie counts from ties_kernel
__global__ __launch_bounds__(NT) void scale_bwd_scalars_kernel(const double* dc_partial, int ndc, const double* tie_partial, int nties,
                                                               ScaleScal* sc, float k1, float k2)
{
    __shared__ double red[NT];
    double a = 0, c = 0, hi = 0, lo = 0;
    for (int i = threadIdx.x; i < ndc; i += NT) { a += dc_partial[i * 2]; c += dc_partial[i * 2 + 1]; }
    for (int i = threadIdx.x; i < nties; i += NT) { hi += tie_partial[i * 2]; lo += tie_partial[i * 2 + 1]; }
    a = block_sum(a, red); c = block_sum(c, red); hi = block_sum(hi, red); lo = block_sum(lo, red);
    if (threadIdx.x == 0) {
        const double dr = sc->dr;
        sc->ddr = sc->from_p ? (float)(2.0 * k1 * k1 * dr * a + 2.0 * k2 * k2 * dr * c) : 0.f;
        sc->nmax = (float)fmax(hi, 1.0);
        sc->nmin = (float)fmax(lo, 1.0);
    }
}

// MS-SSIM / SSIM value and the upstream gradient of every (scale, image) statistic
struct CombineParams {
    const double* stat;  // [nscales][B][2]
    int nscales, B;
    int group;           // images per SAMPLE (the channels of an NCHW batch folded into B): MS-SSIM averages a scale's statistic over them
                         // before the product over scales, as torchmetrics does (`ssim_idx.reshape(B, -1).mean(-1)` over C, H, W)
    float betas[LOSS_MAX_SCALES];
    int ms;              // 1: MS-SSIM product; 0: plain SSIM (mean of stat[0][b][0])
    float weight;        // d total / d value
    float* value;        // out
    float* gup;          // out [nscales][B][2]
};
__global__ __launch_bounds__(NT) void combine_kernel(const CombineParams P)
{
    __shared__ double red[NT];
    double acc = 0;
    const int G = P.group, S = P.B / G;          // S samples of G images
    if (!P.ms) {
        for (int b = threadIdx.x; b < P.B; b += NT) {       // plain SSIM: the mean over all images = the mean over samples of their channel means
            acc += P.stat[b * 2];
            P.gup[b * 2] = P.weight / (float)P.B;
            P.gup[b * 2 + 1] = 0.f;
        }
        acc = block_sum(acc, red);
        if (threadIdx.x == 0) *P.value = (float)(acc / P.B);
        return;
    }
    for (int smp = threadIdx.x; smp < S; smp += NT) {
        double v[LOSS_MAX_SCALES], M = 1.0;
        for (int s = 0; s < P.nscales; ++s) {
            double raw = 0.0;
            for (int c = 0; c < G; ++c) raw += P.stat[((long long)s * P.B + (long long)smp * G + c) * 2 + (s == P.nscales - 1 ? 0 : 1)];
            raw /= G;
            v[s] = raw > 0 ? raw : 0.0;                       // normalize="relu"
            M *= pow(v[s], (double)P.betas[s]);
        }
        acc += M;
        for (int s = 0; s < P.nscales; ++s) {
            const float g = v[s] > 0 ? (float)((double)P.betas[s] * M / v[s] * P.weight / S / G) : 0.f;
            const bool last = s == P.nscales - 1;
            for (int c = 0; c < G; ++c) {
                P.gup[((long long)s * P.B + (long long)smp * G + c) * 2] = last ? g : 0.f;
                P.gup[((long long)s * P.B + (long long)smp * G + c) * 2 + 1] = last ? 0.f : g;
            }
        }
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) *P.value = (float)(acc / S);
    return;
}

// ---- transpose filter of the three gradient maps back onto the image, + pooled-scale and data_range terms ----------
struct BackFilterParams {
    const float* p; const float* t;
    const float* maps;    // [3][B][Hv][Wv]
    int B, H, W, Hv, Wv;
    const ScaleScal* sc;
    const float* coarse;  // gradient of the next (pooled) scale [B][Hc][Wc] or null
    int Hc, Wc;
    float* out;           // [B][H][W]
    int accumulate;
    Taps taps;
};
__global__ __launch_bounds__(NT) void back_filter_kernel(const BackFilterParams P)
{
    __shared__ float sm[3][TW][TW + 1];
    __shared__ float hz[3][TW][TS];
    const int R2 = 2 * P.taps.R, ext = TS + R2;
    const int b = blockIdx.z, y0 = blockIdx.y * TS, x0 = blockIdx.x * TS;
    const long long npix = (long long)P.Hv * P.Wv, plane = (long long)P.B * npix;
    for (int i = threadIdx.x; i < ext * ext; i += NT) {
        const int r = i / ext, c = i - r * ext;
        const int oy = y0 - R2 + r, ox = x0 - R2 + c;        // valid-map coordinates
        const bool in = oy >= 0 && oy < P.Hv && ox >= 0 && ox < P.Wv;
        const long long o = ((long long)b * P.Hv + oy) * P.Wv + ox;
        sm[0][r][c] = in ? P.maps[o] : 0.f;
        sm[1][r][c] = in ? P.maps[plane + o] : 0.f;
        sm[2][r][c] = in ? P.maps[2 * plane + o] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ext * TS; i += NT) {
        const int r = i / TS, c = i - r * TS;
        float a0 = 0, a1 = 0, a2 = 0;
        for (int k = 0; k <= R2; ++k) {                        // taps are symmetric: correlation == convolution
            const float g = P.taps.g[k];
            a0 += g * sm[0][r][c + k]; a1 += g * sm[1][r][c + k]; a2 += g * sm[2][r][c + k];
        }
        hz[0][r][c] = a0; hz[1][r][c] = a1; hz[2][r][c] = a2;
    }
    __syncthreads();
    const ScaleScal s = *P.sc;
    const int c = threadIdx.x & (TS - 1);
    for (int yy = threadIdx.x / TS; yy < TS; yy += NT / TS) {
        const int y = y0 + yy, x = x0 + c;
        if (y >= P.H || x >= P.W) continue;
        float f0 = 0, f1 = 0, f2 = 0;
        for (int k = 0; k <= R2; ++k) {
            const float g = P.taps.g[k];
            f0 += g * hz[0][yy + k][c]; f1 += g * hz[1][yy + k][c]; f2 += g * hz[2][yy + k][c];
        }
        const long long i = ((long long)b * P.H + y) * P.W + x;
        const float pv = P.p[i], tv = P.t[i];
        float d = f0 + 2.f * pv * f1 + tv * f2;
        if (s.from_p) {
            if (pv == s.pmax) d += s.ddr / s.nmax;
            if (pv == s.pmin) d -= s.ddr / s.nmin;
        }
        if (P.coarse && (y >> 1) < P.Hc && (x >> 1) < P.Wc)
            d += 0.25f * P.coarse[((long long)b * P.Hc + (y >> 1)) * P.Wc + (x >> 1)];
        P.out[i] = P.accumulate ? P.out[i] + d : d;
    }
}

__global__ void pool2_kernel(const float* p, const float* t, float* po, float* to, int B, int H, int W, int Ho, int Wo)
{
    const long long n = (long long)B * Ho * Wo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho), b = (int)(i / ((long long)Wo * Ho));
        const long long s = ((long long)b * H + 2 * y) * W + 2 * x;
        po[i] = (p[s] + p[s + 1] + p[s + W] + p[s + W + 1]) * 0.25f;
        to[i] = (t[s] + t[s + 1] + t[s + W] + t[s + W + 1]) * 0.25f;
    }
}

// ---- pointwise terms: l1, poisson, psnr -------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void pointwise_sums_kernel(const float* p, const float* t, long long n, double* partial)
{
    __shared__ double red[NT];
    double a = 0, b = 0, c = 0;
    float lo = INFINITY, hi = -INFINITY;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        const float x = p[i], y = t[i], d = x - y;
        a += (double)fabsf(d);
        b += (double)(x - y * logf(x + 1e-8f));
        c += (double)(d * d);
        lo = fminf(lo, y); hi = fmaxf(hi, y);
    }
    a = block_sum(a, red); b = block_sum(b, red); c = block_sum(c, red);
    __shared__ float fr[2][NT];
    fr[0][threadIdx.x] = lo; fr[1][threadIdx.x] = hi;
    __syncthreads();
    for (int k = NT / 2; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) {
            fr[0][threadIdx.x] = fminf(fr[0][threadIdx.x], fr[0][threadIdx.x + k]);
            fr[1][threadIdx.x] = fmaxf(fr[1][threadIdx.x], fr[1][threadIdx.x + k]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* o = partial + (long long)blockIdx.x * 5;
        o[0] = a; o[1] = b; o[2] = c; o[3] = fr[0][0]; o[4] = fr[1][0];
    }
}
// values[0..2] = l1, poisson, psnr; values[3] = mse (for the psnr gradient)
__global__ __launch_bounds__(NT) void pointwise_final_kernel(const double* partial, int nblocks, long long n, int B, float* v_l1,
                                                             float* v_poisson, float* v_psnr, float* mse_out, float* extra)
{
    __shared__ double red[NT];
    double a = 0, b = 0, c = 0, lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < nblocks; i += NT) {
        a += partial[i * 5]; b += partial[i * 5 + 1]; c += partial[i * 5 + 2];
        lo = fmin(lo, partial[i * 5 + 3]); hi = fmax(hi, partial[i * 5 + 4]);
    }
    a = block_sum(a, red); b = block_sum(b, red); c = block_sum(c, red);
    __shared__ double mm[2][NT];
    mm[0][threadIdx.x] = lo; mm[1][threadIdx.x] = hi;
    __syncthreads();
    for (int k = NT / 2; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) {
            mm[0][threadIdx.x] = fmin(mm[0][threadIdx.x], mm[0][threadIdx.x + k]);
            mm[1][threadIdx.x] = fmax(mm[1][threadIdx.x], mm[1][threadIdx.x + k]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mse = c / (double)n;
        const double dr = fmax(mm[1][0], 0.0) - fmin(mm[0][0], 0.0);   // metric states start at 0 (torchmetrics psnr.py)
        *v_l1 = (float)(a / (double)n);
        *v_poisson = (float)(b / (double)n / (double)B);
        *v_psnr = (float)((2.0 * log(dr) - log(mse)) * (10.0 / log(10.0)));
        *mse_out = (float)mse;
        extra[0] = (float)mse; extra[1] = (float)mm[0][0]; extra[2] = (float)mm[1][0];
    }
}
__global__ void pointwise_grad_kernel(const float* p, const float* t, float* dy, long long n, float w_l1, float w_po, float w_ps,
                                      const float* mse, int accumulate)
{
    const float inv_n = 1.f / (float)n;
    const float kps = w_ps != 0.f ? -w_ps * (float)(10.0 / 2.302585092994046) * 2.f * inv_n / *mse : 0.f;
    const float kl1 = w_l1 * inv_n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float x = p[i], y = t[i], d = x - y;
        float g = kps * d;
        if (w_l1 != 0.f) g += d > 0.f ? kl1 : (d < 0.f ? -kl1 : 0.f);
        if (w_po != 0.f) g += w_po * (1.f - y / (x + 1e-8f));
        dy[i] = accumulate ? dy[i] + g : g;
    }
}
__global__ void total_kernel(float* out, LossWeights w)
{
    float tot = 0.f;
    for (int i = 0; i < 5; ++i)
        if (w.w[i] != 0.f) tot += w.w[i] * out[1 + i];
        else out[1 + i] = 0.f;
    if (w.correction > 0.f) tot += w.correction;           // loss_functions.py:44-45
    out[0] = tot;
}
__global__ void zero_kernel(float* p, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = 0.f;
}

int grid_for(long long n, int per) { long long g = (n + per - 1) / per; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

} // namespace

// ---- workspace -------------------------------------------------------------------------------------------------------
size_t loss_workspace_bytes(int B, int H, int W)
{
    size_t img = (size_t)B * H * W * sizeof(float);
    // pooled pyramids of p, t and of the gradient (sum of 1/4^s < 1/3 each), three maps, partials and scalars
    return img * 3 + img + img * 3 + (size_t)(4 << 20);
}

static Taps make_taps(float sigma)
{
    Taps t;
    const int R = (int)(3.5f * sigma + 0.5f);
    t.R = R;
    double e[2 * LOSS_RMAX + 1], sum = 0;
    for (int k = 0; k <= 2 * R && k <= 2 * LOSS_RMAX; ++k) { const double d = (k - R) / (double)sigma; e[k] = exp(-d * d / 2); sum += e[k]; }
    for (int k = 0; k <= 2 * R && k <= 2 * LOSS_RMAX; ++k) t.g[k] = (float)(e[k] / sum);
    return t;
}

int loss_check(const LossWeights& w, int B, int H, int W, const char** why)
{
    const int R = (int)(3.5f * w.sigma + 0.5f);
    if (B < 1 || H < 1 || W < 1) { *why = "loss: empty batch"; return -1; }
    if ((w.w[3] != 0.f || w.w[4] != 0.f) && (R < 1 || R > LOSS_RMAX)) { *why = "loss: sigma outside the supported gaussian support (<= 25 taps)"; return -1; }
    if (w.w[3] != 0.f && (H <= 2 * R || W <= 2 * R)) { *why = "loss: image smaller than the SSIM window"; return -1; }
    if (w.w[4] != 0.f) {
        // the ValueErrors of torchmetrics' _multiscale_ssim_update, plus the 19-tap window at the coarsest scale
        const int ns = LOSS_MAX_SCALES, div = (ns - 1) * (ns - 1);
        if (H < (1 << ns) || W < (1 << ns) || H / div <= w.kernel_size - 1 || W / div <= w.kernel_size - 1) {
            *why = "loss: image too small for 5 MS-SSIM scales with this kernel_size"; return -1;
        }
        if ((H >> (ns - 1)) <= 2 * R || (W >> (ns - 1)) <= 2 * R) { *why = "loss: coarsest MS-SSIM scale is smaller than the gaussian window"; return -1; }
    }
    return 0;
}

// One SSIM chain (nscales = 1: plain SSIM, 5: MS-SSIM).  ws is carved linearly; everything is stream-ordered.
static hipError_t ssim_chain(const LossWeights& w, int term, int nscales, const float* y, const float* t, float* dy, int accumulate,
                             float* out8, int B, int group, int H, int W, char* ws, hipStream_t s)
{
    static const float kBetas[LOSS_MAX_SCALES] = {0.0448f, 0.2856f, 0.3001f, 0.2363f, 0.1333f};
    const Taps taps = make_taps(w.sigma);
    const int R2 = 2 * taps.R;
    size_t off = 0;
    auto carve = [&](size_t bytes) { char* p = ws + off; off += (bytes + 255) & ~(size_t)255; return p; };
    const float* ps[LOSS_MAX_SCALES]; const float* ts[LOSS_MAX_SCALES]; float* gs[LOSS_MAX_SCALES];
    int Hs[LOSS_MAX_SCALES], Ws[LOSS_MAX_SCALES];
    ps[0] = y; ts[0] = t; gs[0] = dy; Hs[0] = H; Ws[0] = W;
    for (int k = 1; k < nscales; ++k) {
        Hs[k] = Hs[k - 1] / 2; Ws[k] = Ws[k - 1] / 2;
        const size_t b = (size_t)B * Hs[k] * Ws[k] * 4;
        float* pk = (float*)carve(b); float* tk = (float*)carve(b);
        ps[k] = pk; ts[k] = tk; gs[k] = dy ? (float*)carve(b) : nullptr;
        hipLaunchKernelGGL(pool2_kernel, dim3(grid_for((long long)B * Hs[k] * Ws[k], 256)), dim3(256), 0, s, ps[k - 1], ts[k - 1], pk, tk, B,
                           Hs[k - 1], Ws[k - 1], Hs[k], Ws[k]);
    }
    ScaleScal* sc = (ScaleScal*)carve(sizeof(ScaleScal) * LOSS_MAX_SCALES);
    double* stat = (double*)carve(sizeof(double) * 2 * B * LOSS_MAX_SCALES);
    float* gup = (float*)carve(sizeof(float) * 2 * B * LOSS_MAX_SCALES);
    const int maxtiles = ((H + TS - 1) / TS) * ((W + TS - 1) / TS);
    double* partial = (double*)carve(sizeof(double) * 2 * (size_t)B * maxtiles);
    float* mm_partial = (float*)carve(sizeof(float) * 4 * 1024);
    double* tie_partial = (double*)carve(sizeof(double) * 2 * 1024);
    float* maps = dy ? (float*)carve((size_t)3 * B * (H - R2) * (W - R2) * 4) : nullptr;

    for (int k = 0; k < nscales; ++k) {
        const long long n = (long long)B * Hs[k] * Ws[k];
        const int nb = grid_for(n, NT * 8) > 1024 ? 1024 : grid_for(n, NT * 8);
        hipLaunchKernelGGL(minmax_kernel, dim3(nb), dim3(NT), 0, s, ps[k], ts[k], n, mm_partial);
        hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(NT), 0, s, mm_partial, nb, sc + k, w.k1, w.k2);
        SsimParams P;
        P.p = ps[k]; P.t = ts[k]; P.B = B; P.H = Hs[k]; P.W = Ws[k]; P.Hv = Hs[k] - R2; P.Wv = Ws[k] - R2;
        P.tilesX = (P.Wv + TS - 1) / TS; P.tilesY = (P.Hv + TS - 1) / TS;
        P.sc = sc + k; P.partial = partial; P.gup = nullptr; P.maps = nullptr; P.taps = taps;
        hipLaunchKernelGGL(ssim_kernel<false>, dim3(P.tilesX, P.tilesY, B), dim3(NT), 0, s, P);
        hipLaunchKernelGGL(image_stat_kernel, dim3(B), dim3(NT), 0, s, partial, P.tilesX * P.tilesY, 1.0 / ((double)P.Hv * P.Wv),
                           stat + (size_t)k * 2 * B);
    }
    CombineParams C;
    C.stat = stat; C.nscales = nscales; C.B = B; C.group = group; C.ms = nscales > 1; C.weight = w.w[term]; C.value = out8 + 1 + term; C.gup = gup;
    for (int k = 0; k < LOSS_MAX_SCALES; ++k) C.betas[k] = kBetas[k];
    hipLaunchKernelGGL(combine_kernel, dim3(1), dim3(NT), 0, s, C);
    if (dy) {
        for (int k = nscales - 1; k >= 0; --k) {
            const long long n = (long long)B * Hs[k] * Ws[k];
            const int nb = grid_for(n, NT * 8) > 1024 ? 1024 : grid_for(n, NT * 8);
            hipLaunchKernelGGL(ties_kernel, dim3(nb), dim3(NT), 0, s, ps[k], n, sc + k, tie_partial);
            SsimParams P;
            P.p = ps[k]; P.t = ts[k]; P.B = B; P.H = Hs[k]; P.W = Ws[k]; P.Hv = Hs[k] - R2; P.Wv = Ws[k] - R2;
            P.tilesX = (P.Wv + TS - 1) / TS; P.tilesY = (P.Hv + TS - 1) / TS;
            P.sc = sc + k; P.partial = partial; P.gup = gup + (size_t)k * 2 * B; P.maps = maps; P.taps = taps;
            hipLaunchKernelGGL(ssim_kernel<true>, dim3(P.tilesX, P.tilesY, B), dim3(NT), 0, s, P);
            hipLaunchKernelGGL(scale_bwd_scalars_kernel, dim3(1), dim3(NT), 0, s, partial, B * P.tilesX * P.tilesY, tie_partial, nb, sc + k,
                               w.k1, w.k2);
            BackFilterParams F;
            F.p = ps[k]; F.t = ts[k]; F.maps = maps; F.B = B; F.H = Hs[k]; F.W = Ws[k]; F.Hv = P.Hv; F.Wv = P.Wv; F.sc = sc + k;
            F.coarse = k + 1 < nscales ? gs[k + 1] : nullptr; F.Hc = k + 1 < nscales ? Hs[k + 1] : 0; F.Wc = k + 1 < nscales ? Ws[k + 1] : 0;
            F.out = gs[k]; F.accumulate = k == 0 ? accumulate : 0; F.taps = taps;
            hipLaunchKernelGGL(back_filter_kernel, dim3((Ws[k] + TS - 1) / TS, (Hs[k] + TS - 1) / TS, B), dim3(NT), 0, s, F);
        }
    }
    return hipGetLastError();
}

hipError_t launch_loss(const LossWeights& w, const float* y, const float* t, float* dy, float* out8, int B, int channels, int H, int W,
                       void* workspace, hipStream_t s)
{
    // B images = B / channels SAMPLES of `channels` images each (an NCHW batch folded).  Two terms see samples, not images:
    // the Poisson term (the reference divides the element mean by preds.size()[0], metrics/metrics.py:30-39) and MS-SSIM
    // (per-sample channel mean of every scale's statistic before the product over scales)
    if (channels < 1 || B % channels) return hipErrorInvalidValue;
    const int samples = B / channels;
    char* ws = (char*)workspace;
    const long long n = (long long)B * H * W;
    int wrote = 0;
    hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(64), 0, s, out8, (long long)LOSS_OUT_FLOATS);
    if (w.w[0] != 0.f || w.w[1] != 0.f || w.w[2] != 0.f) {
        double* partial = (double*)ws;
        float* mse = (float*)(ws + 5 * 1024 * sizeof(double));
        const int nb = grid_for(n, NT * 8) > 1024 ? 1024 : grid_for(n, NT * 8);
        hipLaunchKernelGGL(pointwise_sums_kernel, dim3(nb), dim3(NT), 0, s, y, t, n, partial);
        hipLaunchKernelGGL(pointwise_final_kernel, dim3(1), dim3(NT), 0, s, partial, nb, n, samples, out8 + 1, out8 + 2, out8 + 3, mse, out8 + 6);
        if (dy) {
            hipLaunchKernelGGL(pointwise_grad_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, y, t, dy, n, w.w[0],
                               w.w[1] / ((float)n * (float)samples), w.w[2], mse, 0);
            wrote = 1;
        }
    }
    char* chain_ws = ws + (64 << 10);
    if (w.w[3] != 0.f) {
        hipError_t e = ssim_chain(w, 3, 1, y, t, dy, wrote, out8, B, channels, H, W, chain_ws, s);
        if (e != hipSuccess) return e;
        wrote = dy != nullptr;
    }
    if (w.w[4] != 0.f) {
        hipError_t e = ssim_chain(w, 4, LOSS_MAX_SCALES, y, t, dy, wrote, out8, B, channels, H, W, chain_ws, s);
        if (e != hipSuccess) return e;
        wrote = dy != nullptr;
    }
    if (dy && !wrote) hipLaunchKernelGGL(zero_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, dy, n);
    hipLaunchKernelGGL(total_kernel, dim3(1), dim3(1), 0, s, out8, w);
    return hipGetLastError();
}

} // namespace xsd
