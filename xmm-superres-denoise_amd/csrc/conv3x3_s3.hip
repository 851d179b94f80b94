// conv3x3_s3.hip -- math mode 3 ("bf16x6"): the 3x3 conv (forward + input-gradient) over fp32 feature planes with
// fp32-CLASS arithmetic on the bf16 matrix cores.
// Reference layers: nn.Conv2d(32k -> 32, 3, 1, 1) of rrdb_blocks.py:27-31, generator_rrdb.py:38-44,95,101 (fp32) and
// their autograd input-gradients.
//
// Arithmetic.  Every fp32 operand is split EXACTLY into three bf16 terms, x = hi + mid + lo (round-to-nearest at each
// step; 8 + 8 + 8 significant bits cover fp32's 24), and a product w*x is evaluated as the six bf16 products
//     wh*xh + wh*xm + wm*xh + wh*xl + wl*xh + wm*xm        (dropped: wm*xl + wl*xm + wl*xl <= 2^-23 |w*x|)
// on v_mfma_f32_32x32x16_bf16.  That instruction adds its 16 exact products and the fp32 accumulator in a wide internal
// format and rounds ONCE (measured: profiles/r02_mfma_probe.txt, tools/mfma_probe.hip), so a K = 288..1440 reduction
// sees 16x fewer roundings than an fp32 fma chain: against a float64 evaluation this mode is MORE accurate than exact
// fp32 MFMA / torch's fp32 conv (tests/test_hip_precision.py), at 16/6 = 2.7x the fp32 matrix rate
// (2500 / 6 = 417 TFLOP/s effective peak).  Planes, biases, residual adds and the epilogue stay fp32.
//
// Structure (same skeleton as conv3x3_mfma.hip, cut into 16-channel HALF-steps so that two workgroups fit a CU):
//   * workgroup = 256 threads (4 waves) -> 8 x 32 output pixels, wave w owns rows 2w, 2w+1; 2 workgroups per CU, whose
//     independent barriers let one stage while the other computes;
//   * half-step = 16 input channels: the (8+2) x (32+2) halo tile is fetched as fp32 (64 B per pixel) into registers one
//     half-step ahead, split by the VALU on the way into LDS as three [pixel][16 x bf16] images (32 B per pixel each, 16-B
//     slots XOR-swizzled by (hx >> 3) & 1 -> conflict-free ds_read_b128 for every tap shift), 32,640 B;
//   * weights arrive pre-split from pack_weights_s3_kernel as half-panels [tap][term][lane][8 bf16] = 27,648 B;
//   * per half-step and wave: 9 taps x 2 rows x 6 MFMAs, software pipelined one (tap,row) stage ahead;
//   * K-loop over input planes (n_in x 1) and one-input/many-output (1 x n_out, re-staging the input per output chunk)
//     run through the same step sequence.
#include <cstdlib>
#include "xsd_kernels.h"

namespace xsd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int S3_XT = HALO_PX * 32;                 // 10,880 B per term image
constexpr int S3_X_BYTES = 3 * S3_XT;               // 32,640
constexpr int S3_BIAS = S3_X_BYTES + S3_WH_BYTES;   // 60,288
constexpr int S3_LDS_BYTES = S3_BIAS + 5 * 32 * 4;  // 60,928 -> 2 workgroups per CU
constexpr int S3_ROWB = HALO_W * 32;                // 1088 B per halo row of a term image
constexpr int S3_SLOTS = HALO_PX * 2;               // (pixel, 8-channel octet) staging slots of 32 B fp32
constexpr int S3_XR = (S3_SLOTS + 255) / 256;       // 3
constexpr int S3_WCH = S3_WH_BYTES / 16;            // 1728 16-B chunks per half-panel
constexpr int S3_WR = (S3_WCH + 255) / 256;         // 7

// LDS byte offset (inside one term image) of octet slot o of halo pixel (hy, hx)
__device__ __forceinline__ int s3_off(int hy, int hx, int o) { return hy * S3_ROWB + hx * 32 + ((o ^ ((hx >> 3) & 1)) << 4); }

// exact 3-term split of 8 fp32 values into packed bf16 (element i in bits [16(i&1), +16) of word i>>1)
__device__ __forceinline__ void s3_split8(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& mid, u32x4& lo)
{
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float x0 = w < 2 ? a[2 * w] : b[2 * w - 4], x1 = w < 2 ? a[2 * w + 1] : b[2 * w - 3];
        const unsigned int h0 = __builtin_bit_cast(unsigned short, (__bf16)x0), h1 = __builtin_bit_cast(unsigned short, (__bf16)x1);
        const float r0 = x0 - __builtin_bit_cast(float, h0 << 16), r1 = x1 - __builtin_bit_cast(float, h1 << 16);
        const unsigned int m0 = __builtin_bit_cast(unsigned short, (__bf16)r0), m1 = __builtin_bit_cast(unsigned short, (__bf16)r1);
        const float q0 = r0 - __builtin_bit_cast(float, m0 << 16), q1 = r1 - __builtin_bit_cast(float, m1 << 16);
        const unsigned int l0 = __builtin_bit_cast(unsigned short, (__bf16)q0), l1 = __builtin_bit_cast(unsigned short, (__bf16)q1);
        hi[w] = h0 | (h1 << 16);
        mid[w] = m0 | (m1 << 16);
        lo[w] = l0 | (l1 << 16);
    }
}

__global__ __launch_bounds__(256, 2) void conv3x3_s3_kernel(const ConvParams P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* x_lds = smem;
    char* w_lds = smem + S3_X_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int h = lane >> 5;
    const int l31 = lane & 31;

    const int ntiles = P.B * P.tilesY * P.tilesX;
    const int n_in = P.n_in, n_out = P.n_out;
    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
    const int per_tile = n_out * n_in * 2;     // half-steps per tile
    const int items = my_tiles * per_tile;
    if (items <= 0) return;

    struct TileXY { int b, y0, x0; };
    auto tile_of = [&](int k) {
        int t = (int)blockIdx.x + k * G;
        TileXY r;
        const int tx = t % P.tilesX; t /= P.tilesX;
        r.x0 = tx * TILE_W; r.y0 = (t % P.tilesY) * TILE_H; r.b = t / P.tilesY;
        return r;
    };

    // per-lane LDS read bases (one per tap column): lane = (pixel column l31, k-half h) reads octet slot h
    int abase[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) abase[dx] = s3_off(wv * 2, l31 + dx, h);

    // staging slots of this thread: LDS offsets are tile independent, global offsets are per tile
    int lds_slot[S3_XR];
#pragma unroll
    for (int r = 0; r < S3_XR; ++r) {
        const int slot = r * 256 + tid;
        const int p = slot >> 1, o = slot & 1;
        const int hy = p / HALO_W, hx = p - hy * HALO_W;
        lds_slot[r] = slot < S3_SLOTS ? s3_off(hy, hx, o) : -1;
    }
    int goff[S3_XR];
    auto tile_offsets = [&](const TileXY& T, int rs, int ps) {
#pragma unroll
        for (int r = 0; r < S3_XR; ++r) {
            const int slot = r * 256 + tid;
            const int p = slot >> 1, o = slot & 1;
            const int hy = p / HALO_W, hx = p - hy * HALO_W;
            const int gy = T.y0 - 1 + hy, gx = T.x0 - 1 + hx;
            const bool ok = (slot < S3_SLOTS) && (gy >= 0) && (gy < P.H) && (gx >= 0) && (gx < P.W);
            goff[r] = ok ? gy * rs + gx * ps + o * 8 : -1;
        }
    };

    f32x4 pin[2 * S3_XR];
    u32x4 pw[S3_WR];
    // Loads are UNCONDITIONAL (padding / out-of-range slots read a page of zeros): a branch around a load makes hipcc wait
    // for it before the next one, which serialises the prefetch into dependent round trips (measured: 3,900 cycles).
    const float* zero = reinterpret_cast<const float*>(P.zero);
    auto load_in = [&](int i, int s2, const TileXY& T) {
        const PlaneIn pl = P.in[i];
        const float* base = pl.p + (long long)T.b * pl.bs + s2 * 16;
#pragma unroll
        for (int r = 0; r < S3_XR; ++r) {
            const f32x4* src = reinterpret_cast<const f32x4*>(goff[r] >= 0 ? base + goff[r] : zero);
            pin[2 * r] = src[0];
            pin[2 * r + 1] = src[1];
        }
    };
    auto store_in = [&]() {
#pragma unroll
        for (int r = 0; r < S3_XR; ++r) {
            if (lds_slot[r] >= 0) {
                u32x4 hi, mid, lo;
                s3_split8(pin[2 * r], pin[2 * r + 1], hi, mid, lo);
                *reinterpret_cast<u32x4*>(x_lds + lds_slot[r]) = hi;
                *reinterpret_cast<u32x4*>(x_lds + S3_XT + lds_slot[r]) = mid;
                *reinterpret_cast<u32x4*>(x_lds + 2 * S3_XT + lds_slot[r]) = lo;
            }
        }
    };
    auto load_w = [&](int widx, int s2) {
        const u32x4* src = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(P.wstep[widx]) + s2 * S3_WH_BYTES);
#pragma unroll
        for (int r = 0; r < S3_WR; ++r) {
            const int c = r * 256 + tid;
            pw[r] = src[c < S3_WCH ? c : S3_WCH - 1];   // unconditional (the last round is partial: clamped, not stored)
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int r = 0; r < S3_WR; ++r)
            if (r * 256 + tid < S3_WCH) *reinterpret_cast<u32x4*>(w_lds + (r * 256 + tid) * 16) = pw[r];
    };

    // bias through LDS (a global load behind the prefetch would make its consumer wait for every older VMEM op)
    float* bias_lds = reinterpret_cast<float*>(smem + S3_BIAS);
    if (tid < 160) bias_lds[tid] = (P.bias && tid < 32 * n_out) ? P.bias[tid] : 0.f;

    // Two accumulators per tile row: acc takes the hi*hi products (one rounding per 16-channel k-step: 16x fewer than an
    // fp32 fma chain), accx the five cross products, whose sum is 2^-8 of acc's, so its roundings do not count; they are
    // added once in the epilogue.  A single accumulator would be rounded by all six MFMAs of every k-step.
    f32x16 acc[2], accx[2];
    // D = W (rows = output channel) x X (cols = pixel): lane = (pixel l31, half h); register i holds channel
    // co(i) = (i&3) + 8*(i>>2) + 4h, i.e. four float4 groups q = 0..3 at channels 8q + 4h .. +3.
    auto init_acc = [&](int j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_lds + j * 32 + 8 * q + 4 * h);
#pragma unroll
            for (int t = 0; t < 4; ++t) { acc[0][4 * q + t] = bv[t]; acc[1][4 * q + t] = bv[t]; accx[0][4 * q + t] = 0.f; accx[1][4 * q + t] = 0.f; }
        }
    };

    // one half-step: 9 taps x 2 rows x 6 products; fragments of stage i+1 are requested before the MFMAs of stage i
    const char* wl = w_lds + lane * 16;
    auto compute = [&]() {
        bf16x8 bf[2][3], af[2][3];   // [set][term]: 0 = hi, 1 = mid, 2 = lo
        auto load_b = [&](int tap, bf16x8 (&b)[3]) {
#pragma unroll
            for (int t = 0; t < 3; ++t) b[t] = *reinterpret_cast<const bf16x8*>(wl + (tap * 3 + t) * 1024);
        };
        auto load_a = [&](int tap, int r, bf16x8 (&a)[3]) {
            const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
            for (int t = 0; t < 3; ++t) a[t] = *reinterpret_cast<const bf16x8*>(x_lds + t * S3_XT + abase[dx] + (r + dy) * S3_ROWB);
        };
        load_b(0, bf[0]);
        load_a(0, 0, af[0]);
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const int tap = i >> 1, r = i & 1;
            if (i + 1 < 18) {
                if (((i + 1) & 1) == 0) load_b((i + 1) >> 1, bf[((i + 1) >> 1) & 1]);
                load_a((i + 1) >> 1, (i + 1) & 1, af[(i + 1) & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 (&w)[3] = bf[tap & 1];
            const bf16x8 (&x)[3] = af[i & 1];
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[2], accx[r], 0, 0, 0);   // Wh * Xl
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], x[0], accx[r], 0, 0, 0);   // Wl * Xh
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[1], accx[r], 0, 0, 0);   // Wm * Xm
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[0], acc[r], 0, 0, 0);     // Wh * Xh
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[1], accx[r], 0, 0, 0);   // Wh * Xm
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[0], accx[r], 0, 0, 0);   // Wm * Xh
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Epilogue over fp32 planes: each lane owns one pixel and 16 channels as four float4 groups -> 16-B loads/stores;
    // lanes l and l+32 cover adjacent 16-B chunks, so every store instruction writes 32 x 32 contiguous bytes.
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
                for (int t = 0; t < 4; ++t) v[t] = (acc[r][4 * q + t] + accx[r][4 * q + t]) * o.a1;
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

#ifdef XSD_DIAG   // phase stamps (diagnostic library variant only; tools/stamps.py): accumulated shader cycles per phase
    unsigned long long st[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long t0 = __builtin_readcyclecounter();
    const bool stamp = P.dbg != nullptr;
#define S3_TICK(i) do { if (stamp) { const unsigned long long t_ = __builtin_readcyclecounter(); st[i] += t_ - t0; t0 = t_; } } while (0)
#else
#define S3_TICK(i) do { } while (0)
#endif
    // ---- step sequence inside a tile: for j in n_out: for i in n_in: for s2 in {0,1}
    TileXY cur = tile_of(0);
    tile_offsets(cur, P.in[0].rs, P.in[0].ps);
    load_in(0, 0, cur);
    load_w(0, 0);
    store_in();
    store_w();
    __syncthreads();
    S3_TICK(0);

    int j = 0, i = 0, s2 = 0, k = 0;
#pragma unroll 1
    for (int it = 0; it < items; ++it) {
        const bool more = (it + 1 < items);
        // successor of (j, i, s2)
        int nj = j, ni = i, ns2 = s2 ^ 1;
        bool new_tile = false;
        if (s2 == 1) {
            ni = i + 1;
            if (ni == n_in) { ni = 0; nj = j + 1; if (nj == n_out) { nj = 0; new_tile = true; } }
        }
        TileXY nxt = cur;
        if (more) {
            if (new_tile) { nxt = tile_of(k + 1); tile_offsets(nxt, P.in[0].rs, P.in[0].ps); }
            load_in(ni, ns2, nxt);
            load_w(nj * n_in + ni, ns2);
        }
        S3_TICK(1);
        if (i == 0 && s2 == 0) init_acc(j);
        compute();
        S3_TICK(2);
        if (i == n_in - 1 && s2 == 1) epilogue(j, cur);
        S3_TICK(3);
        if (more) {
            __syncthreads();
            S3_TICK(4);
            store_in();
            store_w();
            __syncthreads();
            S3_TICK(5);
        }
        if (new_tile) { cur = nxt; ++k; }
        j = nj; i = ni; s2 = ns2;
    }
#ifdef XSD_DIAG
    if (stamp && tid == 0) {
#pragma unroll
        for (int q = 0; q < 6; ++q) atomicAdd(&P.dbg[q], st[q]);
        atomicAdd(&P.dbg[6], (unsigned long long)items);
    }
#endif
}

hipError_t launch_conv3x3_s3(const ConvParams& p, hipStream_t stream)
{
    static bool done = false;
    static int ncu = 256;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_s3_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, S3_LDS_BYTES);
        if (e != hipSuccess) return e;
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        done = true;
    }
    if (p.n_in < 1 || p.n_out < 1 || p.n_in * p.n_out > 5 || !p.zero) return hipErrorInvalidValue;
    const int ntiles = p.B * p.tilesX * p.tilesY;
    if (ntiles <= 0) return hipSuccess;
    const int resident = 2 * ncu;
    const dim3 g(ntiles < resident ? ntiles : resident), b(256);
    hipLaunchKernelGGL(conv3x3_s3_kernel, g, b, S3_LDS_BYTES, stream, p);
    return hipGetLastError();
}

} // namespace xsd
