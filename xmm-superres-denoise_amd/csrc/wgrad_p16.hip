// wgrad_p16.hip -- math mode 2: weight gradient over P16 planes (p16.h), bf16x3 MFMA, LDS-DMA double buffering.
// Reference: autograd's conv weight-gradient for nn.Conv2d(32k -> 32n, 3,1,1) (rrdb_blocks.py:27-31;
// generator_rrdb.py:38-44,95,101):  dW[co][ci][tap] = sum_px G[px][co] X[px+tap][ci],  db[co] = sum_px G[px][co].
//
// Same GEMM as wgrad_bf16x3_kernel (M = X channel position, N = G channel position, K = pixels, nine 32x32
// accumulators per wave, ds_read_b64_tr_b16 transposing reads), but X and G are already stored as hi|lo bf16, so a
// tile is four plain copies HBM -> LDS ([pixel][64 B] planes X_hi, X_lo, G_hi, G_lo) issued as LDS-DMA into the buffer
// set the previous tile released: no staging registers, no split, no ds_write, one barrier per tile.
// Rows/columns of the result are P16 positions; wgrad_reduce_kernel (p16 = 1) maps them back to channels.
#include "p16.h"
#include "xsd_kernels.h"

namespace xsd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int RT = 512;
constexpr int RX_CHUNKS = (HALO_PX * 4 + 63) / 64;  // 22 (340 px x 64 B = 21,760 -> 22 KiB)
constexpr int RX_BYTES = RX_CHUNKS * 1024;          // 22,528
constexpr int RG_CHUNKS = TILE_H * TILE_W * 64 / 1024; // 16
constexpr int RG_BYTES = RG_CHUNKS * 1024;          // 16,384
constexpr int R_XH = 0, R_XL = RX_BYTES, R_GH = 2 * RX_BYTES, R_GL = 2 * RX_BYTES + RG_BYTES;
constexpr int R_BUF = 2 * RX_BYTES + 2 * RG_BYTES;  // 77,824 per buffer set
constexpr int R_LDS_BYTES = 2 * R_BUF;              // 155,648

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ bf16x8 tr_frag16(const char* lds_lane_base, int byte_off)
{
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds_lane_base + byte_off));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds_lane_base + byte_off + 4 * 64));
    s16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return __builtin_bit_cast(bf16x8, r);
}

__global__ __launch_bounds__(RT, 2) void wgrad_p16_kernel(const WgradParams P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef XSD_DIAG   // diagnostic library variant only (make diag; selected with XSD_LIB): ablation knobs are compiled out otherwise
    const int abl = P.ablate;
#else
    constexpr int abl = 0;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6); // tile row
    const int h = lane >> 5;
    const int l31 = lane & 31;

    const int part = blockIdx.x;
    const int j = blockIdx.y;
    const int n = blockIdx.z;
    const PlaneIn xp = P.x[j];
    const PlaneIn gp = P.g[n];
    const int ntiles = P.B * P.tilesY * P.tilesX;
    const char* zero = reinterpret_cast<const char*>(P.zero) + (lane & 3) * 16;

    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    float bsum = 0.f; // sum over this lane's pixels of G[.][position l31]

    // DMA: X chunks g = wv, wv+8, wv+16 (< 22); G chunks g = wv, wv+8 (< 16).  slot = g*64 + lane -> pixel slot>>2,
    // 16-B piece slot&3 of the pixel's 64-B hi (or lo) half.
    auto dma_tile = [&](int t, int buf) {
        const int tx = t % P.tilesX;
        const int t2 = t / P.tilesX;
        const int ty = t2 % P.tilesY;
        const int b = t2 / P.tilesY;
        const int x0 = tx * TILE_W, y0 = ty * TILE_H;
        const char* xb = reinterpret_cast<const char*>(xp.p + (long long)b * xp.bs);
        const char* gb = reinterpret_cast<const char*>(gp.p + (long long)b * gp.bs);
        char* dst = smem + buf * R_BUF;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int g = wv + 8 * k;
            if (g < RX_CHUNKS) {
                const int slot = g * 64 + lane;
                const int px = slot >> 2, pc = slot & 3;
                const int hy = px / HALO_W, hx = px - hy * HALO_W;
                const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
                const bool ok = px < HALO_PX && gy >= 0 && gy < P.H && gx >= 0 && gx < P.W;
                const char* src = xb + 4 * ((long long)gy * xp.rs + gx * xp.ps) + pc * 16;
                __builtin_amdgcn_global_load_lds((gptr_t)(ok ? src : zero), (lptr_t)(dst + R_XH + g * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)(ok ? src + 64 : zero), (lptr_t)(dst + R_XL + g * 1024), 16, 0, 0);
            }
        }
        if ((abl & 4096) && t != part) return; // diagnostic: stale (but realistic) G tiles after the first
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int g = wv + 8 * k;
            const int slot = g * 64 + lane;
            const int px = slot >> 2, pc = slot & 3;
            const int gy = y0 + (px >> 5), gx = x0 + (px & 31);
            const bool ok = gy < P.H && gx < P.W;
            const char* src = gb + 4 * ((long long)gy * gp.rs + gx * gp.ps) + pc * 16;
            __builtin_amdgcn_global_load_lds((gptr_t)(ok ? src : zero), (lptr_t)(dst + R_GH + g * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(ok ? src + 64 : zero), (lptr_t)(dst + R_GL + g * 1024), 16, 0, 0);
        }
    };

    const int i16 = lane & 15;
    const int lane_off = (8 * h + (i16 >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (i16 & 3) * 8;
    const int xoff = wv * (HALO_W * 64) + lane_off;
    const int goff = wv * (TILE_W * 64) + lane_off;

    int t = part, buf = 0;
    if (t < ntiles) dma_tile(t, 0);
    __syncthreads();
#pragma unroll 1
    for (; t < ntiles; t += P.nparts, buf ^= 1) {
        if (t + P.nparts < ntiles) dma_tile(t + P.nparts, buf ^ 1);
        const char* base = smem + buf * R_BUF;
        bf16x8 gh[2], gl[2];
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
            gh[mf] = tr_frag16(base + goff, R_GH + 16 * mf * 64);
            gl[mf] = tr_frag16(base + goff, R_GL + 16 * mf * 64);
        }
        if (j == 0) {
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum += (float)gh[mf][e] + (float)gl[mf][e];
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
                const int off = (dy * HALO_W + dx + 16 * mf) * 64;
                const bf16x8 xh = tr_frag16(base + xoff, R_XH + off);
                const bf16x8 xl = tr_frag16(base + xoff, R_XL + off);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, gh[mf], acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, gl[mf], acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, gh[mf], acc[tap], 0, 0, 0);
            }
        }
        __syncthreads(); // all waves are done with `buf`; the other buffer's DMA has landed (vmcnt(0) precedes the barrier)
    }

    // ---- cross-wave reduction through LDS (fixed order), one tap at a time
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
        for (int e = tid; e < 1024; e += RT) {
            float sacc = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) sacc += red[w * 1024 + e];
            outp[tap * 1024 + e] = sacc;
        }
        __syncthreads();
    }
    if (j == 0) {
        red[tid] = bsum; // [wave][h][position]
        __syncthreads();
        if (tid < 32) {
            float sacc = 0.f;
            for (int w = 0; w < 16; ++w) sacc += red[w * 32 + tid];
            P.bias_partial[((long long)part * P.n_g + n) * 32 + tid] = sacc;
        }
    }
}

hipError_t launch_wgrad_p16(const WgradParams& p, hipStream_t stream)
{
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_p16_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS_BYTES);
        if (e != hipSuccess) return e;
        done = true;
    }
    if (!p.zero) return hipErrorInvalidValue;
    hipLaunchKernelGGL(wgrad_p16_kernel, dim3(p.nparts, p.n_in, p.n_g), dim3(RT), R_LDS_BYTES, stream, p);
    return hipGetLastError();
}

} // namespace xsd
