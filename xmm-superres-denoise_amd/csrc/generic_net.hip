// generic_net.hip -- the RRDB generators at ARBITRARY widths: any num_filters, any in/out channel counts.
//
// The reference's constructors accept any widths (generator_rrdb.py:10-54; config/config.py:164-203 PositiveInt; the dense
// block's own default is nf = 64, rrdb_blocks.py:23); the MFMA path of this library is specialised for the shipped
// configuration (32 filters, one image channel, res/configs/models.toml).  Every other configuration runs here: hand-written
// HIP direct-convolution kernels in exact fp32 (fmaf chains on the vector ALUs, no matrix cores, no split arithmetic; the
// math mode of xsd_set_math does not apply), NCHW tensors like the reference's.  Same C ABI, same flat parameter layout, same
// backward stages as the MFMA path (xsd_engine.hip dispatches on the configuration).
//
// Layout.  torch.cat of the dense block (rrdb_blocks.py:49-52) is a channel PREFIX of one slab [B][5 nf][H][W] per dense
// block: conv_k reads channels [0, k nf) and writes [k nf, (k+1) nf); conv5 of a block writes block 0 of the next slab with
// its residual adds fused (x5 * 0.2 + x, and out * 0.2 + x for the third block: rrdb_blocks.py:54,70).  PixelShuffle
// (generator_rrdb.py:97) is the upsample conv's store addressing.
//
// Backward = reverse-mode restatement of the same graph with read-modify-write gradient slabs (two, ping-pong): for a dense
// block, G5 = 0.2 dOut, dS = conv5^T(G5), dS[block 0] += dOut, then for c = 4..1: G_c = dS[block c] * lrelu'(x_c),
// dW_c = X^T G_c, dS[0, c nf) += conv_c^T(G_c).  Weight gradients: per-block partial sums over a pixel range, combined in a
// fixed order in double (bitwise reproducible, no atomics).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "generic_net.h"

namespace xsd {

// ---------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------
struct GView {          // [B][C][H][W] view: channel stride H*W, batch stride bs (floats); p may point at a channel offset
    float* p;
    long long bs;
};

struct GConvP {
    const float* x; long long xbs; int cin;     // input channels [0, cin)
    const float* w;                               // [cout][cin][9] (forward: OIHW; input-gradient: transposed + flipped)
    const float* bias;                            // [cout] or null
    float* y; long long ybs; int cout;
    int B, H, W;
    float a1;                                     // v = (acc + bias) * a1
    const float* e1; long long e1bs; float s1; int e1c;   // v += s1 * e1[ch]   for ch < e1c
    float a2;                                     // v *= a2
    const float* e2; long long e2bs; float s2;    // v += s2 * e2[ch]
    float slope;                                  // v = v > 0 ? v : v * slope
    int accumulate;                               // v += y (read-modify-write; after everything else)
    int shuffle;                                  // y is [B][cout/4][2H][2W]: channel oc -> (oc >> 2, sub-pixel (oc >> 1) & 1, oc & 1)
    const float* skip; long long skipbs; int skipc;       // head: v += skip[ch or 0]  (skipc == 1: broadcast over channels)
    float* pre;                                   // head: pre-clamp value (same layout as y) or null
    int clamp01;
};

constexpr int GT = 16;              // 16 x 16 pixel tile per workgroup
constexpr int GCO = 8, GCI = 8;     // output channels per thread / input channels per LDS round

__global__ __launch_bounds__(256) void gconv3x3_kernel(const GConvP P)
{
    __shared__ float xin[GCI][GT + 2][GT + 2];
    __shared__ float wl[GCO][GCI][9];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int tilesX = (P.W + GT - 1) / GT;
    const int x0 = ((int)blockIdx.x % tilesX) * GT, y0 = ((int)blockIdx.x / tilesX) * GT;
    const int co0 = blockIdx.y * GCO, b = blockIdx.z;
    const long long HW = (long long)P.H * P.W;
    const float* xb = P.x + (long long)b * P.xbs;
    float acc[GCO];
#pragma unroll
    for (int k = 0; k < GCO; ++k) acc[k] = 0.f;
    for (int ci0 = 0; ci0 < P.cin; ci0 += GCI) {
        __syncthreads();
        for (int i = threadIdx.x; i < GCI * (GT + 2) * (GT + 2); i += 256) {
            const int c = i / ((GT + 2) * (GT + 2)), r = i % ((GT + 2) * (GT + 2));
            const int hy = r / (GT + 2), hx = r % (GT + 2);
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx, ci = ci0 + c;
            const bool ok = ci < P.cin && gy >= 0 && gy < P.H && gx >= 0 && gx < P.W;
            xin[c][hy][hx] = ok ? xb[(long long)ci * HW + (long long)gy * P.W + gx] : 0.f;
        }
        for (int i = threadIdx.x; i < GCO * GCI * 9; i += 256) {
            const int k = i / (GCI * 9), r = i % (GCI * 9), c = r / 9, t = r % 9;
            const int co = co0 + k, ci = ci0 + c;
            wl[k][c][t] = (co < P.cout && ci < P.cin) ? P.w[((long long)co * P.cin + ci) * 9 + t] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < GCI; ++c) {
            float v[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) v[t] = xin[c][ty + t / 3][tx + t % 3];
#pragma unroll
            for (int k = 0; k < GCO; ++k)
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[k] = fmaf(wl[k][c][t], v[t], acc[k]);
        }
    }
    const int gx = x0 + tx, gy = y0 + ty;
    if (gx >= P.W || gy >= P.H) return;
    const long long pix = (long long)gy * P.W + gx;
#pragma unroll
    for (int k = 0; k < GCO; ++k) {
        const int co = co0 + k;
        if (co >= P.cout) break;
        float v = (acc[k] + (P.bias ? P.bias[co] : 0.f)) * P.a1;
        if (P.e1 && co < P.e1c) v += P.s1 * P.e1[(long long)b * P.e1bs + co * HW + pix];
        v *= P.a2;
        if (P.e2) v += P.s2 * P.e2[(long long)b * P.e2bs + co * HW + pix];
        v = v > 0.f ? v : v * P.slope;
        if (P.skip) v += P.skip[(long long)b * P.skipbs + (P.skipc == 1 ? 0 : co) * HW + pix];
        long long o;
        if (P.shuffle) o = (long long)b * P.ybs + (long long)(co >> 2) * 4 * HW + (long long)(2 * gy + ((co >> 1) & 1)) * (2 * P.W) + 2 * gx + (co & 1);
        else o = (long long)b * P.ybs + co * HW + pix;
        if (P.accumulate) v += P.y[o];
        if (P.pre) P.pre[o] = v;
        if (P.clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
        P.y[o] = v;
    }
}

// weight gradient, stage 1: partial[part][co][ci][9] = sum over the part's pixels of g[b,co,p] * x[b,ci,p + tap];
// bias_partial[part][co] = sum g (blocks with ci == 0).  grid (cin, cout, parts), 256 threads, fixed-order LDS tree.
struct GWgradP {
    const float* x; long long xbs; int cin;
    const float* g; long long gbs; int cout;
    int B, H, W, parts;
    float* partial; float* bias_partial;
};
__global__ __launch_bounds__(256) void gwgrad_kernel(const GWgradP P)
{
    __shared__ float red[256];
    const int ci = blockIdx.x, co = blockIdx.y, part = blockIdx.z;
    const long long HW = (long long)P.H * P.W, N = (long long)P.B * HW;
    const long long per = (N + P.parts - 1) / P.parts, n0 = part * per, n1 = n0 + per < N ? n0 + per : N;
    float acc[9], bs = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    for (long long n = n0 + threadIdx.x; n < n1; n += 256) {
        const int b = (int)(n / HW);
        const long long pix = n - (long long)b * HW;
        const int y = (int)(pix / P.W), x = (int)(pix - (long long)y * P.W);
        const float g = P.g[(long long)b * P.gbs + co * HW + pix];
        bs += g;
        const float* xp = P.x + (long long)b * P.xbs + ci * HW;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            const float xv = (yy >= 0 && yy < P.H && xx >= 0 && xx < P.W) ? xp[(long long)yy * P.W + xx] : 0.f;
            acc[t] = fmaf(g, xv, acc[t]);
        }
    }
    for (int t = 0; t < 10; ++t) {
        __syncthreads();
        red[threadIdx.x] = t < 9 ? acc[t] : bs;
        __syncthreads();
        for (int k = 128; k > 0; k >>= 1) {
            if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            if (t < 9) P.partial[(((long long)part * P.cout + co) * P.cin + ci) * 9 + t] = red[0];
            else if (ci == 0) P.bias_partial[(long long)part * P.cout + co] = red[0];
        }
    }
}
// stage 2: fixed-order sum over the parts (double) into the flat gradient (OIHW, bias)
__global__ void gwgrad_reduce_kernel(const float* partial, const float* bias_partial, int parts, int cout, int cin, float* dw, float* db)
{
    const int n = cout * cin * 9;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) {
        double s = 0.0;
        for (int p = 0; p < parts; ++p) s += (double)partial[(long long)p * n + e];
        dw[e] = (float)s;
    }
    if (e < cout) {
        double s = 0.0;
        for (int p = 0; p < parts; ++p) s += (double)bias_partial[(long long)p * cout + e];
        db[e] = (float)s;
    }
}

// elementwise over a [B][C][HW] view
__global__ void gew_kernel(int op, float* d, long long dbs, const float* s, long long sbs, int C, long long HW, int B, float a)
{
    const long long total = (long long)B * C * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / (C * HW));
        const long long r = i - (long long)b * C * HW;
        float* dp = d + (long long)b * dbs + r;
        const float sv = s[(long long)b * sbs + r];
        if (op == 0) *dp = a * sv;                           // copy * a
        else if (op == 1) *dp += a * sv;                     // axpy
        else *dp = sv > 0.f ? *dp : *dp * a;                 // d *= lrelu'(s), slope a
    }
}
// G[b][4c + 2i + j][y][x] = dU[b][c][2y+i][2x+j] * lrelu'(U[b][c][2y+i][2x+j])   (PixelShuffle^T with the upsample LeakyReLU(0.01))
__global__ void gunshuffle_kernel(const float* dU, const float* U, float* G, int B, int C4, int H, int W, float slope)
{
    const long long total = (long long)B * C4 * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        long long r = i / W;
        const int y = (int)(r % H); r /= H;
        const int oc = (int)(r % C4);
        const int b = (int)(r / C4);
        const long long o = (((long long)b * (C4 / 4) + (oc >> 2)) * (2 * H) + 2 * y + ((oc >> 1) & 1)) * (2 * W) + 2 * x + (oc & 1);
        const float u = U[o], g = dU[o];
        G[i] = u > 0.f ? g : g * slope;
    }
}
// dst[co][ci][t] -> wT[ci][co][8 - t]  for every conv of the table
struct GPackDesc { long long w_off, t_off; int cout, cin; };
__global__ void gpack_kernel(const float* params, float* wt, const GPackDesc* descs)
{
    const GPackDesc d = descs[blockIdx.y];
    const int n = d.cout * d.cin * 9;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const int t = e % 9, ci = (e / 9) % d.cin, co = e / (9 * d.cin);
        wt[d.t_off + ((long long)ci * d.cout + co) * 9 + (8 - t)] = params[d.w_off + e];
    }
}
// dx[b][ci][p] (+)= sum over the out channels of s (skip gradient of the DN head when the image channels broadcast)
__global__ void gsum_channels_kernel(float* d, const float* s, int C, long long HW, int B)
{
    const long long total = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW);
        const long long p = i - (long long)b * HW;
        float t = 0.f;
        for (int c = 0; c < C; ++c) t += s[((long long)b * C + c) * HW + p];
        d[i] += t;
    }
}
__global__ void gclamp_bwd_kernel(const float* pre, const float* dy, float* dpre, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = pre[i];
        dpre[i] = (v >= 0.f && v <= 1.f) ? dy[i] : 0.f;     // torch.clamp passes the gradient at the bounds (twice the same mask)
    }
}

static inline int ew_grid(long long n) { long long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

// ---------------------------------------------------------------------------------------------------------------
// the net
// ---------------------------------------------------------------------------------------------------------------
#define GCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

GenericNet::~GenericNet()
{
    hipFree(wt); hipFree(descs_dev); hipFree(ws); hipFree(wg_partial); hipFree(wg_bias_partial);
}

GenericNet* GenericNet::create(const xsd_config& cfg)
{
    GenericNet* n = new GenericNet();
    n->cfg = cfg;
    n->nf = cfg.num_filters; n->cin = cfg.in_channels; n->cout = cfg.out_channels; n->blocks = cfg.num_res_blocks;
    n->sr = cfg.kind == XSD_KIND_SR;
    n->nup = n->sr ? cfg.num_upsample : 0;
    long long off = 0, toff = 0;
    auto mk = [&](int co, int ci) {
        GConvW c; c.w = off; off += (long long)co * ci * 9; c.b = off; off += co; c.cout = co; c.cin = ci; c.t = toff; toff += (long long)co * ci * 9;
        return c;
    };
    n->first = mk(n->nf, n->cin);
    for (int i = 0; i < n->blocks; ++i) {
        n->rrdb_begin.push_back(off);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 5; ++c) n->rdb.push_back(mk(n->nf, n->nf * (c + 1)));
    }
    n->rrdb_begin.push_back(off);
    n->trunk = mk(n->nf, n->nf);
    n->last = mk(n->cout, n->nf);
    for (int u = 0; u < n->nup; ++u) n->up.push_back(mk(4 * n->nf, n->nf));
    if (n->sr) n->hr = mk(n->nf, n->nf);
    n->nparams = off;
    n->wt_floats = toff;
    std::vector<GPackDesc> d;
    auto add = [&](const GConvW& c) { GPackDesc q; q.w_off = c.w; q.t_off = c.t; q.cout = c.cout; q.cin = c.cin; d.push_back(q); };
    add(n->first);
    for (auto& c : n->rdb) add(c);
    add(n->trunk); add(n->last);
    for (auto& c : n->up) add(c);
    if (n->sr) add(n->hr);
    n->ndesc = (int)d.size();
    int maxw = 0;
    for (auto& q : d) maxw = std::max(maxw, q.cout * q.cin * 9);
    n->max_w = maxw;
    if (hipMalloc((void**)&n->wt, sizeof(float) * toff) != hipSuccess ||
        hipMalloc((void**)&n->descs_dev, sizeof(GPackDesc) * d.size()) != hipSuccess ||
        hipMemcpy(n->descs_dev, d.data(), sizeof(GPackDesc) * d.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc((void**)&n->wg_partial, sizeof(float) * (size_t)GenericNet::MAX_PARTS * maxw) != hipSuccess ||
        hipMalloc((void**)&n->wg_bias_partial, sizeof(float) * (size_t)GenericNet::MAX_PARTS * 4 * std::max(n->nf, n->cout)) != hipSuccess) {
        delete n;
        return nullptr;
    }
    return n;
}

hipError_t GenericNet::pack(const float* dev_params, hipStream_t s)
{
    params = dev_params;
    hipLaunchKernelGGL(gpack_kernel, dim3(8, ndesc), dim3(256), 0, s, dev_params, wt, reinterpret_cast<const GPackDesc*>(descs_dev));
    packed = true;
    return hipGetLastError();
}

// workspace carving
struct GArena {
    char* base; size_t top = 0;
    float* take(size_t floats) { float* p = base ? reinterpret_cast<float*>(base + top) : nullptr; top += (floats * sizeof(float) + 255) & ~(size_t)255; return p; }
};

hipError_t GenericNet::plan(int B_, int H_, int W_, bool train_)
{
    if (B_ == B && H_ == H && W_ == W && (int)train_ == train) return hipSuccess;
    B = 0; train = -1;
    const long long HW = (long long)H_ * W_;
    const int nslab = train_ ? blocks * 3 : 2;
    for (int pass = 0; pass < 2; ++pass) {
        GArena a; a.base = pass ? ws : nullptr;
        fea = a.take((size_t)B_ * nf * HW);
        rin = a.take((size_t)B_ * nf * HW);
        slabs.assign(nslab, nullptr);
        for (int i = 0; i < nslab; ++i) slabs[i] = a.take((size_t)B_ * 5 * nf * HW);
        rout = a.take((size_t)B_ * nf * HW);          // output of the last dense block (input of trunk_conv)
        T = a.take((size_t)B_ * nf * HW);
        U.assign(nup, nullptr);
        for (int u = 0; u < nup; ++u) U[u] = a.take((size_t)B_ * nf * (HW << (2 * (u + 1))));
        H1 = sr ? a.take((size_t)B_ * nf * (HW << (2 * nup))) : nullptr;
        pre = train_ ? a.take((size_t)B_ * cout * (HW << (2 * nup))) : nullptr;
        if (train_) {
            const long long HWo = HW << (2 * nup);
            dpre = a.take((size_t)B_ * cout * HWo);
            dS[0] = a.take((size_t)B_ * 5 * nf * HW);
            dS[1] = a.take((size_t)B_ * 5 * nf * HW);
            gtmp = a.take((size_t)B_ * nf * HW);
            dRR = a.take((size_t)B_ * nf * HW);
            dT = a.take((size_t)B_ * nf * HW);
            dHi[0] = sr ? a.take((size_t)B_ * nf * HWo) : nullptr;      // gradient planes at the output resolution (ping-pong)
            dHi[1] = sr ? a.take((size_t)B_ * nf * HWo) : nullptr;
            gup = sr ? a.take((size_t)B_ * 4 * nf * (HWo >> 2)) : nullptr;
        }
        if (!pass) {
            if (a.top > ws_bytes) {
                if (ws) { hipDeviceSynchronize(); hipFree(ws); ws = nullptr; ws_bytes = 0; }
                GCHK(hipMalloc((void**)&ws, a.top));
                ws_bytes = a.top;
            }
        }
    }
    B = B_; H = H_; W = W_; train = (int)train_;
    return hipSuccess;
}

hipError_t GenericNet::conv(hipStream_t s, const GConvW& c, bool transposed, const float* x, long long xbs, float* y, long long ybs, int B_, int H_, int W_,
                            const std::function<void(void*)>& tweak)
{
    GConvP p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.xbs = xbs; p.y = y; p.ybs = ybs; p.B = B_; p.H = H_; p.W = W_;
    if (!transposed) { p.cin = c.cin; p.cout = c.cout; p.w = params + c.w; p.bias = params + c.b; }
    else { p.cin = c.cout; p.cout = c.cin; p.w = wt + c.t; p.bias = nullptr; }
    p.a1 = 1.f; p.a2 = 1.f; p.slope = 1.f;
    if (tweak) tweak(&p);
    const int tiles = ((W_ + GT - 1) / GT) * ((H_ + GT - 1) / GT);
    hipLaunchKernelGGL(gconv3x3_kernel, dim3(tiles, (p.cout + GCO - 1) / GCO, B_), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t GenericNet::wgrad(hipStream_t s, const GConvW& c, const float* x, long long xbs, const float* g, long long gbs, int B_, int H_, int W_, float* grads)
{
    GWgradP p;
    const long long N = (long long)B_ * H_ * W_;
    int parts = (int)std::min<long long>(MAX_PARTS, std::max<long long>(1, N / 2048));
    p.x = x; p.xbs = xbs; p.cin = c.cin; p.g = g; p.gbs = gbs; p.cout = c.cout; p.B = B_; p.H = H_; p.W = W_; p.parts = parts;
    p.partial = wg_partial; p.bias_partial = wg_bias_partial;
    hipLaunchKernelGGL(gwgrad_kernel, dim3(c.cin, c.cout, parts), dim3(256), 0, s, p);
    const int n = c.cout * c.cin * 9;
    hipLaunchKernelGGL(gwgrad_reduce_kernel, dim3((std::max(n, c.cout) + 255) / 256), dim3(256), 0, s, wg_partial, wg_bias_partial, parts, c.cout, c.cin,
                       grads + c.w, grads + c.b);
    return hipGetLastError();
}

hipError_t GenericNet::ew(hipStream_t s, int op, float* d, long long dbs, const float* src, long long sbs, int C, long long HW, float a)
{
    hipLaunchKernelGGL(gew_kernel, dim3(ew_grid((long long)B * C * HW)), dim3(256), 0, s, op, d, dbs, src, sbs, C, HW, B, a);
    return hipGetLastError();
}

hipError_t GenericNet::forward(const float* x, float* y, int B_, int H_, int W_, bool save, hipStream_t s)
{
    GCHK(plan(B_, H_, W_, save));
    const long long HW = (long long)H * W, nfHW = nf * HW, slab_bs = 5 * nfHW;
    b_x = x;
    GCHK(conv(s, first, false, x, cin * HW, fea, nfHW, B, H, W, nullptr));                               // generator_rrdb.py:67
    const int nslab = (int)slabs.size();
    GCHK(ew(s, 0, slabs[0], slab_bs, fea, nfHW, nf, HW, 1.f));
    for (int i = 0; i < blocks; ++i) {
        float* S0 = slabs[(3 * i) % nslab];
        GCHK(ew(s, 0, rin, nfHW, S0, slab_bs, nf, HW, 1.f));                                               // the RRDB's input (residual of :70)
        for (int r = 0; r < 3; ++r) {
            const int k = 3 * i + r;
            float* S = slabs[k % nslab];
            const bool lastk = k == 3 * blocks - 1;
            float* nxt = lastk ? rout : slabs[(k + 1) % nslab];
            const long long nxt_bs = lastk ? nfHW : slab_bs;
            for (int c = 0; c < 4; ++c)                                                                   // rrdb_blocks.py:38-52
                GCHK(conv(s, rdb[k * 5 + c], false, S, slab_bs, S + (c + 1) * nfHW, slab_bs, B, H, W, [](void* q) { ((GConvP*)q)->slope = 0.2f; }));
            const float* rin_ = rin;
            GCHK(conv(s, rdb[k * 5 + 4], false, S, slab_bs, nxt, nxt_bs, B, H, W, [=](void* q) {               // x5 * 0.2 + x (:54)
                GConvP* p = (GConvP*)q;
                p->a1 = 0.2f; p->e1 = S; p->e1bs = slab_bs; p->s1 = 1.f; p->e1c = nf;
                if (r == 2) { p->a2 = 0.2f; p->e2 = rin_; p->e2bs = nfHW; p->s2 = 1.f; }                     // out * 0.2 + x (:70)
            }));
        }
    }
    const float* fea_ = fea;
    GCHK(conv(s, trunk, false, rout, nfHW, T, nfHW, B, H, W, [=](void* q) {                                  // fea + trunk_conv(...) (:68-69)
        GConvP* p = (GConvP*)q; p->e1 = fea_; p->e1bs = nfHW; p->s1 = 1.f; p->e1c = nf; }));
    const float* feat = T;
    int lv = 0;
    for (int u = 0; u < nup; ++u) {                                                                        // :93-99
        GCHK(conv(s, up[u], false, feat, nfHW << (2 * u), U[u], nfHW << (2 * (u + 1)), B, H << u, W << u, [](void* q) {
            GConvP* p = (GConvP*)q; p->slope = 0.01f; p->shuffle = 1; }));
        feat = U[u]; lv = u + 1;
    }
    const long long HWo = HW << (2 * lv);
    if (sr) {
        GCHK(conv(s, hr, false, feat, nf * HWo, H1, nf * HWo, B, H << lv, W << lv, [](void* q) { ((GConvP*)q)->slope = 0.2f; }));   // :107
        feat = H1;
    }
    float* pre_ = pre;
    const bool dn = !sr;
    const int cin_ = cin;
    GCHK(conv(s, last, false, feat, nf * HWo, y, cout * HWo, B, H << lv, W << lv, [=](void* q) {               // :107-108 / :132-135, model.py:49
        GConvP* p = (GConvP*)q;
        if (dn) { p->skip = x; p->skipbs = cin_ * HW; p->skipc = cin_; }
        p->pre = pre_; p->clamp01 = 1;
    }));
    saved = save;
    return hipSuccess;
}

hipError_t GenericNet::rdb_backward(hipStream_t s, int k, const float* dOut, long long dOut_bs, float* dSk, float* grads)
{
    const long long HW = (long long)H * W, nfHW = nf * HW, slab_bs = 5 * nfHW;
    const float* S = slabs[k];
    GCHK(ew(s, 0, gtmp, nfHW, dOut, dOut_bs, nf, HW, 0.2f));                                               // G5 = 0.2 * dOut
    GCHK(wgrad(s, rdb[k * 5 + 4], S, slab_bs, gtmp, nfHW, B, H, W, grads));
    GCHK(conv(s, rdb[k * 5 + 4], true, gtmp, nfHW, dSk, slab_bs, B, H, W, nullptr));
    GCHK(ew(s, 1, dSk, slab_bs, dOut, dOut_bs, nf, HW, 1.f));                                              // the "+ x" path
    for (int c = 3; c >= 0; --c) {
        float* G = dSk + (c + 1) * nfHW;
        GCHK(ew(s, 2, G, slab_bs, S + (c + 1) * nfHW, slab_bs, nf, HW, 0.2f));                             // * lrelu'(x_{c+1})
        GCHK(wgrad(s, rdb[k * 5 + c], S, slab_bs, G, slab_bs, B, H, W, grads));
        GCHK(conv(s, rdb[k * 5 + c], true, G, slab_bs, dSk, slab_bs, B, H, W, [](void* q) { ((GConvP*)q)->accumulate = 1; }));
    }
    return hipSuccess;
}

hipError_t GenericNet::backward_stage(int stage, const float* dy, float* dx, float* grads, hipStream_t s)
{
    const long long HW = (long long)H * W, nfHW = nf * HW, slab_bs = 5 * nfHW;
    const int lv = nup;
    const long long HWo = HW << (2 * lv);
    const int Ho = H << lv, Wo = W << lv;
    if (stage == 0) {
        hipLaunchKernelGGL(gclamp_bwd_kernel, dim3(ew_grid((long long)B * cout * HWo)), dim3(256), 0, s, pre, dy, dpre, (long long)B * cout * HWo);
        const float* feat = sr ? H1 : T;
        GCHK(wgrad(s, last, feat, nf * HWo, dpre, cout * HWo, B, Ho, Wo, grads));
        if (!sr) {
            GCHK(conv(s, last, true, dpre, cout * HWo, dT, nfHW, B, H, W, nullptr));
        } else {
            float* dH1 = dHi[0];
            GCHK(conv(s, last, true, dpre, cout * HWo, dH1, nf * HWo, B, Ho, Wo, nullptr));
            {   // * lrelu'(H1, 0.2)  (views at the output resolution: reuse ew with B tiles of nf channels)
                hipLaunchKernelGGL(gew_kernel, dim3(ew_grid((long long)B * nf * HWo)), dim3(256), 0, s, 2, dH1, nf * HWo, (const float*)H1, nf * HWo, nf, HWo, B, 0.2f);
            }
            const float* hr_in = nup > 0 ? U[nup - 1] : T;
            GCHK(wgrad(s, hr, hr_in, nf * HWo, dH1, nf * HWo, B, Ho, Wo, grads));
            float* dU = nup > 0 ? dHi[1] : dT;
            GCHK(conv(s, hr, true, dH1, nf * HWo, dU, nf * HWo, B, Ho, Wo, nullptr));
            for (int u = nup - 1; u >= 0; --u) {
                const long long HWu = HW << (2 * u);
                const int Hu = H << u, Wu = W << u;
                hipLaunchKernelGGL(gunshuffle_kernel, dim3(ew_grid((long long)B * 4 * nf * HWu)), dim3(256), 0, s, (const float*)dU, (const float*)U[u], gup, B, 4 * nf, Hu, Wu, 0.01f);
                const float* xin = u > 0 ? U[u - 1] : T;
                GCHK(wgrad(s, up[u], xin, nf * HWu, gup, 4 * nf * HWu, B, Hu, Wu, grads));
                float* dn_ = u > 0 ? (dU == dHi[1] ? dHi[0] : dHi[1]) : dT;
                GCHK(conv(s, up[u], true, gup, 4 * nf * HWu, dn_, nf * HWu, B, Hu, Wu, nullptr));
                dU = dn_;
            }
        }
        GCHK(wgrad(s, trunk, rout, nfHW, dT, nfHW, B, H, W, grads));
        GCHK(conv(s, trunk, true, dT, nfHW, dRR, nfHW, B, H, W, nullptr));          // gradient wrt the last RRDB's output
        return hipSuccess;
    }
    if (stage <= blocks) {
        const int i = blocks - stage;
        // dRR = gradient wrt this RRDB's output; RDB3's output gradient is 0.2 * dRR (out * 0.2 + x, rrdb_blocks.py:70)
        const float* dOut = nullptr; long long dOut_bs = 0;
        for (int r = 2; r >= 0; --r) {
            const int k = 3 * i + r;
            float* dSk = dS[k & 1];
            if (r == 2) {
                float* d3 = dS[(k + 1) & 1];          // block 0 of the OTHER gradient slab holds 0.2 * dRR
                GCHK(ew(s, 0, d3, slab_bs, dRR, nfHW, nf, HW, 0.2f));
                dOut = d3; dOut_bs = slab_bs;
            }
            GCHK(rdb_backward(s, k, dOut, dOut_bs, dSk, grads));
            dOut = dSk; dOut_bs = slab_bs;            // block 0 = gradient wrt this dense block's input
        }
        GCHK(ew(s, 1, dRR, nfHW, dOut, dOut_bs, nf, HW, 1.f));      // "+ x" of the RRDB: d(rin) = dRR + d(RDB1 input)
        return hipSuccess;
    }
    // last stage: conv_first.  gradient wrt fea = dRR (through the RRDBs) + dT (the fea + trunk skip)
    GCHK(ew(s, 1, dRR, nfHW, dT, nfHW, nf, HW, 1.f));
    GCHK(wgrad(s, first, b_x, cin * HW, dRR, nfHW, B, H, W, grads));
    if (dx) {
        GCHK(conv(s, first, true, dRR, nfHW, dx, cin * HW, B, H, W, nullptr));
        if (!sr) {      // skip path of the DN head: d(x) += d(out) (summed over the output channels when x broadcasts)
            if (cin == cout) GCHK(ew(s, 1, dx, cin * HW, dpre, cout * HW, cin, HW, 1.f));
            else hipLaunchKernelGGL(gsum_channels_kernel, dim3(ew_grid((long long)B * HW)), dim3(256), 0, s, dx, (const float*)dpre, cout, HW, B);
        }
    }
    return hipGetLastError();
}

void GenericNet::grad_range(int stage, long long* off, long long* cnt) const
{
    long long a, b;
    if (stage == 0) { a = rrdb_begin[blocks]; b = nparams; }
    else if (stage <= blocks) { const int i = blocks - stage; a = rrdb_begin[i]; b = rrdb_begin[i + 1]; }
    else { a = 0; b = rrdb_begin[0]; }
    *off = a; *cnt = b - a;
}

} // namespace xsd
