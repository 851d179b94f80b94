// conv3x3_mfma.hip -- the hot kernel: 3x3 / stride 1 / zero-pad 1 convolution over 32-channel feature planes as an
// implicit GEMM on the CDNA4 fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fmaf chain).
//
// Replaces, for the reference's nn.Conv2d(32k -> 32, 3, 1, 1) layers (rrdb_blocks.py:27-31, generator_rrdb.py:38-44,
// 95, 101) and their autograd input-gradients, the ATen/MIOpen convolution + the torch.cat copies + the elementwise
// LeakyReLU / residual kernels.  GEMM view:  M = pixels (32 per MFMA = one tile row), N = 32 output channels,
// K = 9 taps x 32 input channels per plane.
//
// Workgroup = 256 threads (4 waves) -> 8 x 32 output pixels; wave w owns rows 2w, 2w+1 (two 32x32 accumulators).
// LDS (80,384 B -> 2 workgroups/CU): one (8+2) x (32+2) x 32ch input halo tile, 16-B chunks XOR-swizzled by
// (pixel>>1)&7 so the ds_read_b128 A-fragment reads (16 lanes = 16 consecutive pixels, same chunk) are conflict-free
// without padding; plus one 9x32x32 weight panel stored in fragment order [tap][j][h][co][4].
// K-steps are software pipelined through registers (issue the next plane's global loads before the MFMAs of the
// current one, write LDS after the barrier); the second co-resident workgroup covers the short write phase.
//
// Two step structures share the code:
//   MULTI_OUT=false : n_in input planes (K-loop), one output chunk          (forward dense convs; dgrad of 32->128)
//   MULTI_OUT=true  : one input plane staged once, n_out output chunks      (dgrad of dense convs; forward 32->128)
#include "xsd_kernels.h"

namespace xsd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int IN_SLOTS = HALO_PX * 8;              // 2720 16-byte chunks
constexpr int IN_ROUNDS = (IN_SLOTS + 255) / 256;  // 11
constexpr int W_ROUNDS = PANEL_FLOATS / 4 / 256;   // 9

__device__ __forceinline__ int swz_off(int p, int c) { return p * 128 + ((c ^ ((p >> 1) & 7)) << 4); }

// bijective XCD-aware remap: workgroups that share an XCD (bid % 8) get a contiguous range of tiles, so halo rows and
// the weight panels are re-read from that XCD's own L2.
__device__ __forceinline__ int xcd_remap(int bid, int n)
{
    const int q = n >> 3, r = n & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <bool MULTI_OUT>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(const ConvParams P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* in_lds = smem;
    char* w_lds = smem + IN_LDS_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int h = lane >> 5;
    const int l31 = lane & 31;

    const int ntiles = P.B * P.tilesY * P.tilesX;
    int t = xcd_remap(blockIdx.x, ntiles);
    const int tx = t % P.tilesX; t /= P.tilesX;
    const int ty = t % P.tilesY;
    const int b = t / P.tilesY;
    const int x0 = tx * TILE_W, y0 = ty * TILE_H;

    const int nsteps = MULTI_OUT ? P.n_out : P.n_in;

    f32x4 pin[IN_ROUNDS];
    f32x4 pw[W_ROUNDS];

    auto load_in = [&](int s) {
        const PlaneIn pl = P.in[s];
        const float* base = pl.p + (long long)b * pl.bs;
#pragma unroll
        for (int r = 0; r < IN_ROUNDS; ++r) {
            const int slot = r * 256 + tid;
            const int p = slot >> 3, c = slot & 7;
            const int hy = p / HALO_W, hx = p - hy * HALO_W;
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            const bool ok = (slot < IN_SLOTS) && (gy >= 0) && (gy < P.H) && (gx >= 0) && (gx < P.W);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(base + (long long)gy * pl.rs + gx * pl.ps + c * 4);
            pin[r] = v;
        }
    };
    auto store_in = [&]() {
#pragma unroll
        for (int r = 0; r < IN_ROUNDS; ++r) {
            const int slot = r * 256 + tid;
            const int p = slot >> 3, c = slot & 7;
            if (slot < IN_SLOTS) *reinterpret_cast<f32x4*>(in_lds + swz_off(p, c)) = pin[r];
        }
    };
    auto load_w = [&](int s) {
        const f32x4* src = reinterpret_cast<const f32x4*>(P.wpanel + (long long)s * PANEL_FLOATS);
#pragma unroll
        for (int r = 0; r < W_ROUNDS; ++r) pw[r] = src[r * 256 + tid];
    };
    auto store_w = [&]() {
#pragma unroll
        for (int r = 0; r < W_ROUNDS; ++r) *reinterpret_cast<f32x4*>(w_lds + (r * 256 + tid) * 16) = pw[r];
    };

    f32x16 acc[2];
    auto init_acc = [&](int j) {
        const float bv = P.bias ? P.bias[j * 32 + l31] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[0][i] = bv; acc[1][i] = bv; }
    };

    auto compute = [&]() {
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
            f32x4 bf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bf[j] = *reinterpret_cast<const f32x4*>(w_lds + ((tap * 4 + j) * 64 + lane) * 16);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int p = (wv * 2 + r + dy) * HALO_W + l31 + dx;
                f32x4 af[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const f32x4*>(in_lds + swz_off(p, 4 * h + j));
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][q], bf[j][q], acc[r], 0, 0, 0);
            }
        }
    };

    auto epilogue = [&](int j) {
        const OutDesc o = P.out[j];
        float* dst = o.p + (long long)b * o.bs;
        const long long sb = (long long)b * P.std_bs;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int y = y0 + wv * 2 + r;
            if (y >= P.H) continue;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int x = x0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (x >= P.W) continue;
                const long long od = (long long)y * o.rs + (long long)x * o.ps + l31;
                const long long os = sb + (long long)y * P.std_rs + x * 32 + l31;
                float v = acc[r][i] * o.a1;
                if (o.accumulate) v += dst[od];
                if (o.e1) v += o.s1 * o.e1[os];
                v *= o.a2;
                if (o.e2) v += o.s2 * o.e2[os];
                if (o.e3) v += o.s3 * o.e3[os];
                v = v > 0.f ? v : v * o.slope;
                if (o.mask) v = o.mask[os] > 0.f ? v : v * o.mslope;
                dst[od] = v;
            }
        }
    };

    // ---- prologue: stage step 0
    load_in(0);
    load_w(0);
    store_in();
    store_w();
    __syncthreads();

    if (!MULTI_OUT) init_acc(0);
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
        const bool more = (s + 1 < nsteps);
        if (more) {
            if (!MULTI_OUT) load_in(s + 1);
            load_w(s + 1);
        }
        if (MULTI_OUT) init_acc(s);
        compute();
        if (MULTI_OUT) epilogue(s);
        if (more) {
            __syncthreads();
            if (!MULTI_OUT) store_in();
            store_w();
            __syncthreads();
        }
    }
    if (!MULTI_OUT) epilogue(0);
}

static hipError_t set_lds_once()
{
    static bool done = false;
    if (done) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mfma_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, CONV_LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mfma_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, CONV_LDS_BYTES);
    if (e != hipSuccess) return e;
    done = true;
    return hipSuccess;
}

hipError_t launch_conv3x3_mfma(const ConvParams& p, hipStream_t stream)
{
    hipError_t e = set_lds_once();
    if (e != hipSuccess) return e;
    const int ntiles = p.B * p.tilesX * p.tilesY;
    if (ntiles <= 0) return hipSuccess;
    if (p.n_out > 1)
        hipLaunchKernelGGL(conv3x3_mfma_kernel<true>, dim3(ntiles), dim3(256), CONV_LDS_BYTES, stream, p);
    else
        hipLaunchKernelGGL(conv3x3_mfma_kernel<false>, dim3(ntiles), dim3(256), CONV_LDS_BYTES, stream, p);
    return hipGetLastError();
}

} // namespace xsd
