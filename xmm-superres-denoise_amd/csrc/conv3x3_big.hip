// conv3x3_big.hip -- the "big tile" variant of the 3x3 conv (same math, same ConvParams, same epilogue as
// conv3x3_mfma.hip; reference layers: rrdb_blocks.py:27-31, generator_rrdb.py:38-44,95,101 and their input-gradients).
//
// Why a second structure: with the bf16x3 math the MFMA time per 8x32-pixel step drops to ~3.5k cycles and the kernel
// becomes bound by bytes moved through L2 -> L1: every 256-pixel step re-reads a 36 KB weight panel (45 % of the
// traffic) and a halo that is 1.33x the tile.  This variant trades the second co-resident workgroup for a tile twice as
// tall and a weight ring that never stalls the MFMAs:
//   * workgroup = 512 threads (8 waves, 2 per SIMD), tile = 16 x 32 pixels, wave w owns rows 2w, 2w+1
//     -> per 256 pixels: weights 18 KB (was 36), halo factor 1.195 (was 1.33);
//   * LDS 152,064 B = one 18x34x32ch input tile (78,336 B) + TWO weight panels (2 x 36,864 B);
//   * weight panels arrive by LDS-DMA (global_load_lds_dwordx4, no VGPR staging, no ds_write) into the panel buffer that
//     the previous step released, while the MFMAs of the current step read the other one;
//   * the input tile is prefetched into registers during the MFMAs and written (split into hi|lo bf16 when SPLIT) after
//     the step's barrier; in the one-input/many-output mode (dense-block input-gradients) the tile is staged once per
//     tile, so those steps run barrier -> MFMA -> epilogue back to back.
#include "conv_core.h"
#include "xsd_kernels.h"

namespace xsd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

constexpr int BW = 8;                         // waves
constexpr int BT = BW * 64;                   // 512 threads
constexpr int BTH = 2 * BW;                   // 16 tile rows
constexpr int BHH = BTH + 2;                  // 18 halo rows
constexpr int BPX = BHH * HALO_W;             // 612 halo pixels
constexpr int BIN_BYTES = BPX * 128;          // 78,336
constexpr int BIAS_OFF = BIN_BYTES + 2 * W_LDS_BYTES;       // 152,064
constexpr int BIG_LDS_BYTES = BIAS_OFF + 5 * 32 * 4;       // + bias[n_out*32] (read through LDS: lgkmcnt, not vmcnt)
constexpr int BSLOTS = BPX * 8;               // 4896 (pixel, 4-channel quad) staging slots of 16 B
constexpr int BROUNDS = (BSLOTS + BT - 1) / BT; // 10
constexpr int BROW_BYTES = HALO_W * 128;      // 4352

__device__ __forceinline__ int bswz(int hy, int hx, int c) { return hy * BROW_BYTES + hx * 128 + ((c ^ ((hx >> 1) & 7)) << 4); }

__device__ __forceinline__ void bsplit4(const f32x4& a, u16x4& hi, u16x4& lo)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 h = (__bf16)a[i];
        const __bf16 l = (__bf16)(a[i] - (float)h);
        hi[i] = __builtin_bit_cast(unsigned short, h);
        lo[i] = __builtin_bit_cast(unsigned short, l);
    }
}

template <bool MULTI_OUT, bool SPLIT>
__global__ __launch_bounds__(BT, 2) void conv3x3_big_kernel(const ConvParams P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* in_lds = smem;
    char* w_lds = smem + BIN_BYTES; // two panels

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int l31 = lane & 31;

    const int tilesY = (P.H + BTH - 1) / BTH;
    const int ntiles = P.B * tilesY * P.tilesX;
    const int nsteps = MULTI_OUT ? P.n_out : P.n_in;
    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
    const int items = my_tiles * nsteps;
    if (items <= 0) return;

    struct TileXY { int b, y0, x0; };
    auto tile_of = [&](int k) {
        int t = (int)blockIdx.x + k * G;
        TileXY r;
        const int tx = t % P.tilesX; t /= P.tilesX;
        r.x0 = tx * TILE_W; r.y0 = (t % tilesY) * BTH; r.b = t / tilesY;
        return r;
    };

    int abase[3][4];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = SPLIT ? ((k >> 1) * 4 + 2 * (k & 1) + h) : (4 * h + k);
            abase[dx][k] = bswz(wv * 2, l31 + dx, c);
        }

    // ---- input-tile staging through registers (coalesced: 8 lanes = one pixel's 128 B)
    f32x4 pin[BROUNDS];
    int goff[BROUNDS];
    auto tile_offsets = [&](const TileXY& T, int rs, int ps) {
#pragma unroll
        for (int r = 0; r < BROUNDS; ++r) {
            const int slot = r * BT + tid;
            const int p = slot >> 3, m = slot & 7;
            const int hy = p / HALO_W, hx = p - hy * HALO_W;
            const int gy = T.y0 - 1 + hy, gx = T.x0 - 1 + hx;
            const bool ok = (slot < BSLOTS) && (gy >= 0) && (gy < P.H) && (gx >= 0) && (gx < P.W);
            goff[r] = ok ? gy * rs + gx * ps + m * 4 : -1;
        }
    };
    auto load_in = [&](int s, const TileXY& T) {
        const PlaneIn pl = P.in[s];
        const float* base = pl.p + (long long)T.b * pl.bs;
#pragma unroll
        for (int r = 0; r < BROUNDS; ++r) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (goff[r] >= 0) v = *reinterpret_cast<const f32x4*>(base + goff[r]);
            pin[r] = v;
        }
    };
    auto store_in = [&]() {
#pragma unroll
        for (int r = 0; r < BROUNDS; ++r) {
            const int slot = r * BT + tid;
            const int p = slot >> 3, m = slot & 7;
            const int hy = p / HALO_W, hx = p - hy * HALO_W;
            if (slot < BSLOTS) {
                if constexpr (!SPLIT) {
                    *reinterpret_cast<f32x4*>(in_lds + bswz(hy, hx, m)) = pin[r];
                } else {
                    u16x4 hi, lo;
                    bsplit4(pin[r], hi, lo);
                    *reinterpret_cast<u16x4*>(in_lds + bswz(hy, hx, m >> 1) + (m & 1) * 8) = hi;
                    *reinterpret_cast<u16x4*>(in_lds + bswz(hy, hx, 4 + (m >> 1)) + (m & 1) * 8) = lo;
                }
            }
        }
    };
    // ---- weight panel by LDS-DMA: 36 chunks of 1 KiB, wave w moves chunks w, w+8, ...
    auto dma_w = [&](int s, int buf) {
        const char* src = reinterpret_cast<const char*>(P.wstep[s]) + lane * 16;
        char* dst = w_lds + buf * W_LDS_BYTES;
#pragma unroll
        for (int c0 = 0; c0 < 40; c0 += BW) {
            const int c = c0 + wv;
            if (c < 36)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c * 1024),
                                                 (__attribute__((address_space(3))) void*)(dst + c * 1024), 16, 0, 0);
        }
    };

    // bias goes through LDS: a global load issued after the prefetch would make its consumer wait for every older
    // VMEM op (vmcnt retires in order) and serialise the prefetch in front of the MFMAs.
    float* bias_lds = reinterpret_cast<float*>(smem + BIAS_OFF);
    if (tid < 160) bias_lds[tid] = (P.bias && tid < 32 * (MULTI_OUT ? P.n_out : 1)) ? P.bias[tid] : 0.f;

    f32x16 acc[2];
    auto init_acc = [&](int j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_lds + j * 32 + 8 * q + 4 * h);
#pragma unroll
            for (int t = 0; t < 4; ++t) { acc[0][4 * q + t] = bv[t]; acc[1][4 * q + t] = bv[t]; }
        }
    };

    auto compute = [&](int buf) {
        conv_compute<SPLIT, BROW_BYTES>(in_lds, w_lds + buf * W_LDS_BYTES + lane * 16, abase, acc);
    };

    auto epilogue = [&](int j, const TileXY& T) {
        const OutDesc o = P.out[j];
        float* dst = o.p + (long long)T.b * o.bs;
        const long long sb = (long long)T.b * P.std_bs;
        const int x = T.x0 + l31;
        if (x >= P.W) return;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int y = T.y0 + wv * 2 + r;
            if (y >= P.H) continue;
            float* dp = dst + (long long)y * o.rs + (long long)x * o.ps + 4 * h;
            const long long os = sb + (long long)y * P.std_rs + x * 32 + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = acc[r][4 * q + t] * o.a1;
                if (o.accumulate) v += *reinterpret_cast<const f32x4*>(dp + 8 * q);
                if (o.e1) v += o.s1 * *reinterpret_cast<const f32x4*>(o.e1 + os + 8 * q);
                v *= o.a2;
                if (o.e2) v += o.s2 * *reinterpret_cast<const f32x4*>(o.e2 + os + 8 * q);
                if (o.e3) v += o.s3 * *reinterpret_cast<const f32x4*>(o.e3 + os + 8 * q);
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = v[t] > 0.f ? v[t] : v[t] * o.slope;
                if (o.mask) {
                    const f32x4 m = *reinterpret_cast<const f32x4*>(o.mask + os + 8 * q);
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = m[t] > 0.f ? v[t] : v[t] * o.mslope;
                }
                *reinterpret_cast<f32x4*>(dp + 8 * q) = v;
            }
        }
    };

    // ---- prologue: stage item 0 (tile + panel 0 into buffer 0)
    TileXY cur = tile_of(0);
    tile_offsets(cur, P.in[0].rs, P.in[0].ps);
    dma_w(0, 0);
    load_in(0, cur);
    store_in();
    __syncthreads(); // also waits the DMA (vmcnt(0) before the barrier)

    int s = 0, k = 0;
#pragma unroll 1
    for (int it = 0; it < items; ++it) {
        const int buf = it & 1;
        const bool more = (it + 1 < items);
        const int s_next = (s + 1 == nsteps) ? 0 : s + 1;
        const bool new_in = more && (!MULTI_OUT || s_next == 0);
        TileXY nxt = cur;
        if (more) {
            dma_w(s_next, buf ^ 1); // the other panel was released by the barrier that ended the previous step
            if (s_next == 0) { nxt = tile_of(k + 1); tile_offsets(nxt, P.in[0].rs, P.in[0].ps); }
            if (new_in) load_in(MULTI_OUT ? 0 : s_next, nxt);
        }
        if (MULTI_OUT || s == 0) init_acc(MULTI_OUT ? s : 0);
        compute(buf);
        if (MULTI_OUT) epilogue(s, cur);
        else if (s == nsteps - 1) epilogue(0, cur);
        if (more) {
            __syncthreads(); // every wave is done with this step's tile + panel; next panel has landed
            if (new_in) {
                store_in();
                __syncthreads();
            }
        }
        if (s_next == 0) { cur = nxt; ++k; }
        s = s_next;
    }
}

template <bool M, bool S>
static hipError_t big_set_lds()
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_big_kernel<M, S>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS_BYTES);
}

hipError_t launch_conv3x3_big(const ConvParams& p, int split, hipStream_t stream)
{
    static bool done = false;
    static int ncu = 256;
    if (!done) {
        hipError_t e;
        if ((e = big_set_lds<false, false>()) != hipSuccess) return e;
        if ((e = big_set_lds<true, false>()) != hipSuccess) return e;
        if ((e = big_set_lds<false, true>()) != hipSuccess) return e;
        if ((e = big_set_lds<true, true>()) != hipSuccess) return e;
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        done = true;
    }
    const int tilesY = (p.H + BTH - 1) / BTH;
    const int ntiles = p.B * p.tilesX * tilesY;
    if (ntiles <= 0) return hipSuccess;
    const dim3 g(ntiles < ncu ? ntiles : ncu), b(BT);
    if (p.n_out > 1) {
        if (split) hipLaunchKernelGGL((conv3x3_big_kernel<true, true>), g, b, BIG_LDS_BYTES, stream, p);
        else hipLaunchKernelGGL((conv3x3_big_kernel<true, false>), g, b, BIG_LDS_BYTES, stream, p);
    } else {
        if (split) hipLaunchKernelGGL((conv3x3_big_kernel<false, true>), g, b, BIG_LDS_BYTES, stream, p);
        else hipLaunchKernelGGL((conv3x3_big_kernel<false, false>), g, b, BIG_LDS_BYTES, stream, p);
    }
    return hipGetLastError();
}

} // namespace xsd
