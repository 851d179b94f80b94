// wgrad_mfma.hip -- weight gradient of the 3x3 convs on the fp32 matrix cores.
//
// Replaces autograd's conv weight-gradient (ATen/MIOpen wgrad) for the reference's nn.Conv2d(32k -> 32n, 3,1,1)
// layers (rrdb_blocks.py:27-31; generator_rrdb.py:38-44,95,101):
//     dW[co][ci][tap] = sum_{b,y,x} G[b,y,x,co] * X[b,y+dy-1,x+dx-1,ci],   db[co] = sum G[..,co]
// GEMM view: M = 32 input channels (one plane), N = 32 output channels, K = pixels; nine independent 32x32
// accumulators (one per tap) per wave.  A workgroup (512 threads, 8 waves, 1 per CU) walks a strided subset of
// the 8x32-pixel tiles for ONE (input plane j, G chunk n) pair and keeps its 9 accumulators in registers across all
// of them; wave w owns row w of each tile (32 pixels = 16 MFMA k-steps of 2 pixels).  Partial sums are written once
// per workgroup ([nparts][n][j][9][32][32]) and combined in a fixed order by wgrad_reduce_kernel, so the result is
// bitwise reproducible (no float atomics).
#include <algorithm>
#include "xsd_kernels.h"

namespace xsd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WG_THREADS = 512;
constexpr int X_SLOTS = HALO_PX * 8;                           // 2720
constexpr int X_ROUNDS = (X_SLOTS + WG_THREADS - 1) / WG_THREADS; // 6
constexpr int G_SLOTS = TILE_H * TILE_W * 8;                   // 2048
constexpr int G_ROUNDS = G_SLOTS / WG_THREADS;                 // 4
constexpr int G_LDS_BYTES = TILE_H * TILE_W * 128;             // 32,768
constexpr int WGRAD_LDS_BYTES = IN_LDS_BYTES + G_LDS_BYTES;    // 76,288


__global__ __launch_bounds__(WG_THREADS, 2) void wgrad_mfma_kernel(const WgradParams P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* x_lds = smem;
    char* g_lds = smem + IN_LDS_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6; // 0..7 = tile row
    const int h = lane >> 5;
    const int l31 = lane & 31;

    const int part = blockIdx.x;
    const int j = blockIdx.y;  // input plane
    const int n = blockIdx.z;  // G chunk
    const PlaneIn xp = P.x[j];
    const PlaneIn gp = P.g[n];
    const int ntiles = P.B * P.tilesY * P.tilesX;

    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    float bsum = 0.f;

    f32x4 px[X_ROUNDS];
    f32x4 pg[G_ROUNDS];

    auto load_tile = [&](int t) {
        const int tx = t % P.tilesX;
        const int t2 = t / P.tilesX;
        const int ty = t2 % P.tilesY;
        const int b = t2 / P.tilesY;
        const int x0 = tx * TILE_W, y0 = ty * TILE_H;
        const float* xb = xp.p + (long long)b * xp.bs;
        const float* gb = gp.p + (long long)b * gp.bs;
#pragma unroll
        for (int r = 0; r < X_ROUNDS; ++r) {
            const int slot = r * WG_THREADS + tid;
            const int p = slot >> 3, c = slot & 7;
            const int hy = p / HALO_W, hx = p - hy * HALO_W;
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            const bool ok = (slot < X_SLOTS) && gy >= 0 && gy < P.H && gx >= 0 && gx < P.W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(xb + (long long)gy * xp.rs + gx * xp.ps + c * 4);
            px[r] = v;
        }
#pragma unroll
        for (int r = 0; r < G_ROUNDS; ++r) {
            const int slot = r * WG_THREADS + tid;
            const int p = slot >> 3, c = slot & 7;
            const int gy = y0 + (p >> 5), gx = x0 + (p & 31);
            const bool ok = gy < P.H && gx < P.W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(gb + (long long)gy * gp.rs + gx * gp.ps + c * 4);
            pg[r] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int r = 0; r < X_ROUNDS; ++r) {
            const int slot = r * WG_THREADS + tid;
            if (slot < X_SLOTS) *reinterpret_cast<f32x4*>(x_lds + slot * 16) = px[r]; // linear: b32 reads need no swizzle
        }
#pragma unroll
        for (int r = 0; r < G_ROUNDS; ++r) {
            const int slot = r * WG_THREADS + tid;
            *reinterpret_cast<f32x4*>(g_lds + slot * 16) = pg[r];
        }
    };

    int t = part;
    if (t < ntiles) {
        load_tile(t);
        store_tile();
    }
    __syncthreads();
#pragma unroll 1
    for (; t < ntiles; t += P.nparts) {
        const bool more = (t + P.nparts < ntiles);
        if (more) load_tile(t + P.nparts);
        // B fragments: G[pixel (wv, 2m+h)][co = l31], m = 0..15
        float gf[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            gf[m] = *reinterpret_cast<const float*>(g_lds + (wv * 32 + 2 * m + h) * 128 + l31 * 4);
            bsum += gf[m];
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int p = (wv + dy) * HALO_W + 2 * m + h + dx;
                const float a = *reinterpret_cast<const float*>(x_lds + p * 128 + l31 * 4);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, gf[m], acc[tap], 0, 0, 0);
            }
        }
        __syncthreads();
        if (more) store_tile();
        __syncthreads();
    }

    // ---- cross-wave reduction through LDS, one tap at a time (8 waves x 4 KiB), then one coalesced store per tap
    float* red = reinterpret_cast<float*>(smem); // 8 * 1024 floats = 32 KiB
    float* outp = P.partial + ((((long long)part * P.n_g + n) * P.n_in + j) * 9) * 1024;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ci = (i & 3) + 8 * (i >> 2) + 4 * h;
            red[wv * 1024 + ci * 32 + l31] = acc[tap][i];
        }
        __syncthreads();
        for (int e = tid; e < 1024; e += WG_THREADS) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += red[w * 1024 + e];
            outp[tap * 1024 + e] = s;
        }
        __syncthreads();
    }
    if (j == 0) {
        red[tid] = bsum; // [wave][h][co]
        __syncthreads();
        if (tid < 32) {
            float s = 0.f;
            for (int w = 0; w < 16; ++w) s += red[w * 32 + tid];
            P.bias_partial[((long long)part * P.n_g + n) * 32 + tid] = s;
        }
    }
}

// Fixed-order combination of the per-workgroup partials into the OIHW gradient (state-dict layout).
// Block = 1024 threads = 16 waves over 256 consecutive elements: wave w sums partials [w*P/16, (w+1)*P/16) with one
// float4 per lane (1 KiB contiguous per wave-load, 16 loads in flight), then the 16 wave sums are combined in fixed
// order through LDS: deterministic, and the same summation order for any launch geometry.
__device__ __forceinline__ void wgrad_reduce_body(const WgradReduceParams& R)
{
    __shared__ double red[16][256];
    const int total = R.n_g * R.n_in * 9 * 1024;        // multiple of 256
    if ((int)blockIdx.x * 256 >= total) return;         // (multi-conv launches: the grid is the largest conv's)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int e4 = blockIdx.x * 256 + lane * 4;
    const long long stride = R.part_stride ? R.part_stride : (long long)R.n_g * R.n_in * 9 * 1024;
    const int per = (R.nparts + 15) / 16;
    const int p0 = w * per, p1 = min(R.nparts, p0 + per);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (e4 < total) {
        int p = p0;
        for (; p + 8 <= p1; p += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(R.partial + (p + u) * stride + e4);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s0 += (double)v[u][0]; s1 += (double)v[u][1]; s2 += (double)v[u][2]; s3 += (double)v[u][3]; }
        }
        for (; p < p1; ++p) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(R.partial + p * stride + e4);
            s0 += (double)v[0]; s1 += (double)v[1]; s2 += (double)v[2]; s3 += (double)v[3];
        }
    }
    red[w][lane * 4 + 0] = s0; red[w][lane * 4 + 1] = s1; red[w][lane * 4 + 2] = s2; red[w][lane * 4 + 3] = s3;
    __syncthreads();
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x < 256 && e < total) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
        int r = e;
        const int co = r & 31; r >>= 5;
        const int ci = r & 31; r >>= 5;
        const int tap = r % 9; r /= 9;
        const int j = r % R.n_in;
        const int n = r / R.n_in;
        const int cch = co, ich = ci;
        const int oc = R.shuffle ? (4 * (32 * R.plane + cch) + n) : (32 * (R.n0 + n) + cch);
        R.dw[((long long)oc * R.cin_total + (32 * (R.j0 + j) + ich)) * 9 + tap] = (float)(t * (double)R.scale);
    }
    // bias: block 0; the same 16 x (P/16) fixed-order scheme (a serial 256-load chain here used to cost 60 us per launch)
    if (blockIdx.x == 0) {
        const int nb = R.n_g * 32; // <= 128 entries (n, co)
        const long long bstride = R.bias_stride ? R.bias_stride : nb;
        __syncthreads();
        for (int base = 0; base < nb; base += 64) {
            const int idx = base + lane;
            double t = 0.0;
            if (idx < nb) {
                int p = p0;
                for (; p + 8 <= p1; p += 8) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = R.bias_partial[(long long)(p + u) * bstride + idx];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t += (double)v[u];
                }
                for (; p < p1; ++p) t += (double)R.bias_partial[(long long)p * bstride + idx];
            }
            if (idx < 256) red[w][idx] = t;
        }
        __syncthreads();
        if ((int)threadIdx.x < nb) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
            const int co = threadIdx.x & 31, n = threadIdx.x >> 5;
            const int cch = co;
            const int oc = R.shuffle ? (4 * (32 * R.plane + cch) + n) : (32 * (R.n0 + n) + cch);
            R.db[oc] = (float)(t * (double)R.scale);
        }
    }
}

__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const WgradReduceParams R) { wgrad_reduce_body(R); }

// The reductions of SEVERAL convs in one launch (blockIdx.y = conv): a dense block's pair-list weight-gradient launch is followed by
// five of them (conv1 .. conv5), each 5 us of work behind 4 us of launch floor (tools/launch_floor_probe.hip) -- at the reference's
// batch sizes (1 / 4 / 8) a train step's 60 reduce launches were 3.3 % / 1.3 % / 0.7 % of it.  Same blocks, same summation order per
// element: bitwise the result of the separate launches.
__global__ __launch_bounds__(1024) void wgrad_reduce_multi_kernel(const WgradReduceBatch RB)
{
    WgradReduceParams R = RB.r[0];
#pragma unroll
    for (int k = 1; k < WgradReduceBatch::MAXN; ++k) if ((int)blockIdx.y == k) R = RB.r[k];      // static indices into the kernel arguments
    wgrad_reduce_body(R);
}

static PerDevice g_once;

hipError_t launch_wgrad_mfma(const WgradParams& p, hipStream_t stream)
{
    hipError_t e = g_once.once([]() {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_mfma_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, WGRAD_LDS_BYTES);
    }, nullptr);
    if (e != hipSuccess) return e;
    const dim3 g(p.nparts, p.n_in, p.n_g), b(WG_THREADS);
    hipLaunchKernelGGL(wgrad_mfma_kernel, g, b, WGRAD_LDS_BYTES, stream, p);
    return hipGetLastError();
}

hipError_t launch_wgrad_reduce(const WgradReduceParams& r, hipStream_t stream)
{
    const int total = r.n_g * r.n_in * 9 * 1024;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(1024), 0, stream, r);
    return hipGetLastError();
}

hipError_t launch_wgrad_reduce_multi(const WgradReduceBatch& rb, hipStream_t stream)
{
    if (rb.n < 1 || rb.n > WgradReduceBatch::MAXN) return hipErrorInvalidValue;
    int blocks = 0;
    for (int k = 0; k < rb.n; ++k) blocks = std::max(blocks, (rb.r[k].n_g * rb.r[k].n_in * 9 * 1024 + 255) / 256);
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(blocks, rb.n), dim3(1024), 0, stream, rb);
    return hipGetLastError();
}

} // namespace xsd
