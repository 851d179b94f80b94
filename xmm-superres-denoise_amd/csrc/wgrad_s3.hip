// wgrad_s3.hip -- math mode 3 ("bf16x6"): weight gradient of the 3x3 convs over fp32 planes with fp32-class arithmetic
// on the bf16 matrix cores (3-term exact split of BOTH operands, six products, single-rounding MFMA accumulation: see
// conv3x3_s3.hip).  Replaces autograd's conv weight-gradient for the reference's nn.Conv2d(32k -> 32n, 3,1,1) layers
// (rrdb_blocks.py:27-31; generator_rrdb.py:38-44,95,101):
//     dW[co][ci][tap] = sum_{b,y,x} G[b,y,x,co] * X[b,y+dy-1,x+dx-1,ci],   db[co] = sum G[..,co]
// GEMM view: M = 32 input channels (one plane), N = 32 output channels, K = pixels; nine 32x32 accumulators (one per
// tap) per wave, kept in registers across all tiles of the workgroup.  Workgroup = 256 threads on a 4 x 32-pixel tile
// (wave w owns row w), TWO workgroups per CU: the staging of a tile (fp32 -> registers a tile ahead; VALU split and LDS
// writes between two barriers) is serial inside a workgroup, and the co-resident workgroup's MFMAs run beside it.  Both
// MFMA operands need K (8 consecutive pixels) contiguous per lane while memory is [pixel][channel], so LDS holds six
// images [pixel][32 x bf16] (X_hi, X_mid, X_lo over the 6 x 34 halo, G_hi, G_mid, G_lo; 63,744 B) read with the
// transposing ds_read_b64_tr_b16.  Fixed-order two-stage reduction (wgrad_reduce_kernel): bitwise reproducible, no atomics.
#include "xsd_kernels.h"
#include "xsd_split.h"

namespace xsd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) f32x4* gf32x4p;

constexpr int W3_TH = 4;                                              // tile rows = waves per workgroup
constexpr int W3_THREADS = 64 * W3_TH;                                // 256
constexpr int W3_HPX = (W3_TH + 2) * HALO_W;                          // 204 halo pixels
constexpr int W3_X_SLOTS = W3_HPX * 8;                                // 1632 (pixel, channel quad) slots
constexpr int W3_X_ROUNDS = (W3_X_SLOTS + W3_THREADS - 1) / W3_THREADS; // 7
constexpr int W3_G_SLOTS = W3_TH * TILE_W * 8;                        // 1024
constexpr int W3_G_ROUNDS = W3_G_SLOTS / W3_THREADS;                  // 4
constexpr int W3_XT = W3_HPX * 64;                                    // 13,056 B per X term image
constexpr int W3_GT = W3_TH * TILE_W * 64;                            // 8,192 B per G term image
constexpr int W3_G_OFF = 3 * W3_XT;                                   // 39,168
constexpr int W3_LDS_BYTES = (W3_G_OFF + 3 * W3_GT) > W3_TH * 4096 ? (W3_G_OFF + 3 * W3_GT) : W3_TH * 4096;   // 63,744

// exact 3-term split of 4 fp32 values into packed bf16 pairs
__device__ __forceinline__ void w3_split4(const f32x4& a, u32x2& hi, u32x2& mid, u32x2& lo) { split3_f32x4(a, hi, mid, lo); }   // xsd_split.h

// 8 consecutive pixels (k = 8h + 0..7) of this lane's channel from a [pixel][32 x bf16] image
__device__ __forceinline__ bf16x8 w3_tr_frag(const char* lds_lane_base, int byte_off)
{
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds_lane_base + byte_off));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds_lane_base + byte_off + 4 * 64));
    s16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return __builtin_bit_cast(bf16x8, r);
}

__global__ __launch_bounds__(W3_THREADS, 2) void wgrad_s3_kernel(const WgradParams P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];

#ifdef XSD_DIAG   // ablation bits of the diagnostic library (timing experiments only; results are garbage when set)
    const int abl = P.ablate;
#else
    constexpr int abl = 0;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6; // tile row
    const int h = lane >> 5;
    const int l31 = lane & 31;

    // 1-D grid, decoded so that the n_in workgroups that read the SAME G tiles (same `part`, input planes j = 0..n_in-1)
    // have linear ids 8 apart: the dispatcher deals consecutive ids round-robin over the 8 XCDs, so they land on one XCD
    // back to back and the G tile is fetched from HBM once per group and served from that XCD's L2 to the others.
    const int lin = blockIdx.x;
    const int xcd = lin & 7, qq = lin >> 3;
    const int j = qq % P.n_in;                        // input plane
    const int rest = qq / P.n_in;
    const int parts8 = P.nparts >> 3;
    const int part = (rest % parts8) * 8 + xcd;
    const int n = rest / parts8;                      // G chunk
    const PlaneIn xp = P.x[j];
    const PlaneIn gp = P.g[n];
    const int tilesY = (P.H + W3_TH - 1) / W3_TH;
    const int ntiles = P.B * tilesY * P.tilesX;

    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f}; // this thread's 4 channels (tid & 7) of the G tiles it stages

    f32x4 px[W3_X_ROUNDS];
    f32x4 pg[W3_G_ROUNDS];
    const float* zero = reinterpret_cast<const float*>(P.zero);

    // staging slots of this thread, decoded once: X round r -> halo pixel (hy, hx) and channel quad c, packed hy | hx<<8 | c<<16
    int xslot[W3_X_ROUNDS];
#pragma unroll
    for (int r = 0; r < W3_X_ROUNDS; ++r) {
        const int slot = r * W3_THREADS + tid;
        const int p = slot >> 3, c = slot & 7;
        const int hy = p / HALO_W, hx = p - hy * HALO_W;
        xslot[r] = slot < W3_X_SLOTS ? (hy | (hx << 8) | (c << 16)) : -1;
    }
    auto load_tile = [&](int t) {
        const int tx = t % P.tilesX;
        const int t2 = t / P.tilesX;
        const int ty = t2 % tilesY;
        const int b = t2 / tilesY;
        const int x0 = tx * TILE_W, y0 = ty * W3_TH;
        const float* xb = xp.p + (long long)b * xp.bs;
        const float* gb = gp.p + (long long)b * gp.bs;
#pragma unroll
        for (int r = 0; r < W3_X_ROUNDS; ++r) {
            const int w = xslot[r];
            const int gy = y0 - 1 + (w & 0xff), gx = x0 - 1 + ((w >> 8) & 0xff);
            const bool ok = w >= 0 && (unsigned)gy < (unsigned)P.H && (unsigned)gx < (unsigned)P.W;
            // unconditional load (padding reads the zero page): a branch around it would make hipcc wait for the whole
            // prefetch at the join, i.e. BEFORE the MFMAs it is meant to overlap.  In-image offsets fit 32 bits
            // (xsd_forward rejects larger images).
            px[r] = *(gf32x4p)(ok ? xb + (gy * xp.rs + gx * xp.ps + ((w >> 16) & 7) * 4) : zero);
        }
#pragma unroll
        for (int r = 0; r < W3_G_ROUNDS; ++r) {
            const int slot = r * W3_THREADS + tid;
            const int p = slot >> 3, c = slot & 7;
            const int gy = y0 + (p >> 5), gx = x0 + (p & 31);
            const bool ok = gy < P.H && gx < P.W;
            pg[r] = *(gf32x4p)(ok ? gb + (gy * gp.rs + gx * gp.ps + c * 4) : zero);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int r = 0; r < W3_X_ROUNDS; ++r) {
            const int slot = r * W3_THREADS + tid; // = pixel*8 + quad: 8 B per slot in each image
            if (slot < W3_X_SLOTS) {
                u32x2 hi, mid, lo;
                w3_split4(px[r], hi, mid, lo);
                *reinterpret_cast<u32x2*>(smem + slot * 8) = hi;
                *reinterpret_cast<u32x2*>(smem + W3_XT + slot * 8) = mid;
                *reinterpret_cast<u32x2*>(smem + 2 * W3_XT + slot * 8) = lo;
            }
        }
#pragma unroll
        for (int r = 0; r < W3_G_ROUNDS; ++r) {
            const int slot = r * W3_THREADS + tid;
            u32x2 hi, mid, lo;
            w3_split4(pg[r], hi, mid, lo);
            *reinterpret_cast<u32x2*>(smem + W3_G_OFF + slot * 8) = hi;
            *reinterpret_cast<u32x2*>(smem + W3_G_OFF + W3_GT + slot * 8) = mid;
            *reinterpret_cast<u32x2*>(smem + W3_G_OFF + 2 * W3_GT + slot * 8) = lo;
            bsum += pg[r];
        }
    };

    // per-lane base of the transposing reads: lane i of a 16-lane group addresses block row q = i>>2 (pixel) and
    // columns 4p..4p+3 (p = i&3) of channel group (lane>>4)&1; the lane half h selects pixels +8.
    const int i16 = lane & 15;
    const int lane_off = (8 * h + (i16 >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (i16 & 3) * 8;
    const char* xbase = smem + wv * (HALO_W * 64) + lane_off; // + term image + ((dy*34 + dx + 16*mf) * 64)
    const char* gbase = smem + W3_G_OFF + wv * (TILE_W * 64) + lane_off;

#ifdef XSD_DIAG   // phase stamps (diagnostic library variant only; tools/stamps_train.py)
    unsigned long long st[5] = {0, 0, 0, 0, 0}, ntile = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    const bool stamp = P.dbg != nullptr;
#define W3_TICK(i) do { if (stamp) { const unsigned long long t_ = __builtin_readcyclecounter(); st[i] += t_ - t0; t0 = t_; } } while (0)
#else
#define W3_TICK(i) do { } while (0)
#endif
    int t = part;
    if (t < ntiles) {
        load_tile(t);
        store_tile();
    }
    __syncthreads();
    W3_TICK(0);
#pragma unroll 1
    for (; t < ntiles; t += P.nparts) {
        const bool more = (t + P.nparts < ntiles);
        if (more && !(abl & 16)) load_tile(t + P.nparts);
        W3_TICK(0);
        // Running accumulators take every product: 12 roundings per tile row and tap, against 32 for an fp32 fma chain over
        // the same 32 pixels (single-layer error vs float64: tools/dbg_layer.py).
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
            bf16x8 g[3];
#pragma unroll
            for (int term = 0; term < 3; ++term) g[term] = w3_tr_frag(gbase, term * W3_GT + 16 * mf * 64);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap % 3;
                const int off = (dy * HALO_W + dx + 16 * mf) * 64;
                const bf16x8 xh = w3_tr_frag(xbase, off);
                const bf16x8 xm = w3_tr_frag(xbase, W3_XT + off);
                const bf16x8 xl = w3_tr_frag(xbase, 2 * W3_XT + off);
                if (abl & 8) continue;
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, g[0], acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, g[2], acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, g[1], acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, g[0], acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, g[1], acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, g[0], acc[tap], 0, 0, 0);
            }
        }
        W3_TICK(1);
        __syncthreads();
        W3_TICK(2);
        if (more && !(abl & 1)) store_tile();
        W3_TICK(3);
        __syncthreads();
        W3_TICK(4);
#ifdef XSD_DIAG
        ++ntile;
#endif
    }
#ifdef XSD_DIAG
    if (stamp && tid == 0) {
#pragma unroll
        for (int q = 0; q < 5; ++q) atomicAdd(&P.dbg[8 + q], st[q]);
        atomicAdd(&P.dbg[13], ntile);
    }
#endif

    // ---- cross-wave reduction through LDS, one tap at a time (one 4 KiB slab per wave), then one coalesced store per tap
    float* red = reinterpret_cast<float*>(smem);
    float* outp = P.partial + ((((long long)part * P.n_g + n) * P.n_in + j) * 9) * 1024;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ci = (i & 3) + 8 * (i >> 2) + 4 * h;
            red[wv * 1024 + ci * 32 + l31] = acc[tap][i];
        }
        __syncthreads();
        for (int e = tid; e < 1024; e += W3_THREADS) {
            float sacc = 0.f;
#pragma unroll
            for (int w = 0; w < W3_THREADS / 64; ++w) sacc += red[w * 1024 + e];
            outp[tap * 1024 + e] = sacc;
        }
        __syncthreads();
    }
    if (j == 0) { // bias gradient: thread tid staged channels 4*(tid&7)..+3 of the G tiles
#pragma unroll
        for (int i = 0; i < 4; ++i) red[tid * 4 + i] = bsum[i];
        __syncthreads();
        if (tid < 32) {
            const int q = tid >> 2, i = tid & 3;
            float sacc = 0.f;
            for (int w = 0; w < W3_THREADS / 8; ++w) sacc += red[(w * 8 + q) * 4 + i];
            P.bias_partial[((long long)part * P.n_g + n) * 32 + tid] = sacc;
        }
    }
}

hipError_t launch_wgrad_s3(const WgradParams& p, hipStream_t stream)
{
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_s3_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, W3_LDS_BYTES);
        if (e != hipSuccess) return e;
        done = true;
    }
    if (!p.zero) return hipErrorInvalidValue;
    if (p.nparts & 7) return hipErrorInvalidValue;
    const dim3 g(p.nparts * p.n_in * p.n_g), b(W3_THREADS);
    hipLaunchKernelGGL(wgrad_s3_kernel, g, b, W3_LDS_BYTES, stream, p);
    return hipGetLastError();
}

} // namespace xsd
