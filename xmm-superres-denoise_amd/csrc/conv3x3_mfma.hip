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
#include <cstdlib>
#include "conv_core.h"
#include "xsd_kernels.h"

namespace xsd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int IN_SLOTS = HALO_PX * 8;              // 2720 16-byte chunks
constexpr int IN_ROUNDS = (IN_SLOTS + 255) / 256;  // 11
constexpr int W_ROUNDS = PANEL_FLOATS / 4 / 256;   // 9

constexpr int CONV_LDS_TOTAL = CONV_LDS_BYTES + 5 * 32 * 4; // + bias -> 81,024 B, still 2 workgroups per CU
constexpr int ROW_BYTES = HALO_W * 128; // 4352 = 17 x 256: every halo row starts on a bank-row boundary

// LDS byte offset of 16-B chunk c (0..7) of halo pixel (hy, hx).  The XOR swizzle depends on hx only, so a tap shift
// (dy, dx) is "+ dy*ROW_BYTES" (an instruction immediate) on top of one of three per-lane bases (dx = 0,1,2):
// no address arithmetic is left in the MFMA loop.  A ds_read_b128 lane group (16 lanes = 16 consecutive hx, same c)
// touches 16 distinct 16-B slots of the 256-B bank row -> conflict-free.
__device__ __forceinline__ int swz_off(int hy, int hx, int c) { return hy * ROW_BYTES + hx * 128 + ((c ^ ((hx >> 1) & 7)) << 4); }

template <bool MULTI_OUT>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(const ConvParams P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* in_lds = smem;
    char* w_lds = smem + IN_LDS_BYTES;

#ifdef XSD_DIAG   // diagnostic library variant only (make diag; selected with XSD_LIB): ablation knobs are compiled out otherwise
    const int abl = P.ablate;
#else
    constexpr int abl = 0;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int h = lane >> 5;
    const int l31 = lane & 31;

    const int ntiles = P.B * P.tilesY * P.tilesX;
    const int nsteps = MULTI_OUT ? P.n_out : P.n_in;
    // persistent: workgroup g walks tiles g, g + G, g + 2G, ... and software-pipelines across tile boundaries
    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
    const int items = my_tiles * nsteps;

    struct TileXY { int b, y0, x0; };
    auto tile_of = [&](int k) {
        int t = (int)blockIdx.x + k * G;
        TileXY r;
        const int tx = t % P.tilesX; t /= P.tilesX;
        r.x0 = tx * TILE_W; r.y0 = (t % P.tilesY) * TILE_H; r.b = t / P.tilesY;
        return r;
    };

    // per-lane LDS read bases: [dx][k], k = the 4 A-fragment chunks of one (row, tap)
    int abase[3][4];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            abase[dx][k] = swz_off(wv * 2, l31 + dx, 4 * h + k);
        }

    f32x4 pin[IN_ROUNDS];
    f32x4 pw[W_ROUNDS];

    // per-thread element offset of staging slot r inside one image of the tile (-1 = zero padding / out of range)
    auto slot_offset = [&](int r, const TileXY& T, int rs, int ps) {
        const int slot = r * 256 + tid;
        const int p = slot >> 3;
        const int sub = (slot & 7) * 4;
        const int hy = p / HALO_W, hx = p - hy * HALO_W;
        const int gy = T.y0 - 1 + hy, gx = T.x0 - 1 + hx;
        const bool ok = (slot < IN_SLOTS) && (gy >= 0) && (gy < P.H) && (gx >= 0) && (gx < P.W);
        return (ok && !(abl & 1)) ? gy * rs + gx * ps + sub : -1;
    };
    auto load_in = [&](int s, const TileXY& T) {
        const PlaneIn pl = P.in[s];
        const float* base = pl.p + (long long)T.b * pl.bs;
#pragma unroll
        for (int r = 0; r < IN_ROUNDS; ++r) {
            const int off = slot_offset(r, T, pl.rs, pl.ps);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (off >= 0) v = *reinterpret_cast<const f32x4*>(base + off);
            pin[r] = v;
        }
    };
    auto store_in = [&]() {
#pragma unroll
        for (int r = 0; r < IN_ROUNDS; ++r) {
            const int slot = r * 256 + tid;
            const int p = slot >> 3, c = slot & 7;
            const int hy = p / HALO_W, hx = p - hy * HALO_W;
            if (slot < IN_SLOTS) *reinterpret_cast<f32x4*>(in_lds + swz_off(hy, hx, c)) = pin[r];
        }
    };
    auto load_w = [&](int s) {
        const f32x4* src = reinterpret_cast<const f32x4*>(P.wstep[s]);
        if (abl & 2) return;
#pragma unroll
        for (int r = 0; r < W_ROUNDS; ++r) pw[r] = src[r * 256 + tid];
    };
    auto store_w = [&]() {
#pragma unroll
        for (int r = 0; r < W_ROUNDS; ++r) *reinterpret_cast<f32x4*>(w_lds + (r * 256 + tid) * 16) = pw[r];
    };

    f32x16 acc[2];
    // D = W (rows = output channel) x X (cols = pixel): lane = (pixel l31, half h); register i holds channel
    // co(i) = (i&3) + 8*(i>>2) + 4h, i.e. four float4 groups q = 0..3 at channels 8q + 4h .. +3.
    // bias goes through LDS: a global load issued after the prefetch would make its consumer wait for every older
    // VMEM op (vmcnt retires in order) and serialise the prefetch in front of the MFMAs.
    float* bias_lds = reinterpret_cast<float*>(smem + CONV_LDS_BYTES);
    if (tid < 160) bias_lds[tid] = (P.bias && tid < 32 * (MULTI_OUT ? P.n_out : 1)) ? P.bias[tid] : 0.f;
    auto init_acc = [&](int j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_lds + j * 32 + 8 * q + 4 * h);
#pragma unroll
            for (int t = 0; t < 4; ++t) { acc[0][4 * q + t] = bv[t]; acc[1][4 * q + t] = bv[t]; }
        }
    };

    const char* wl = w_lds + lane * 16;
    auto compute = [&]() { if (!(abl & 8)) conv_compute<ROW_BYTES>(in_lds, wl, abase, acc); };

    // Epilogue: each lane owns one pixel and 16 channels as four float4 groups -> 16-B loads/stores; lanes l and l+32
    // cover adjacent 16-B chunks, so every store instruction writes 32 x 32 contiguous bytes.
    auto epilogue = [&](int j, const TileXY& T) {
        const OutDesc o = P.out[j];
        float* dst = o.p + (long long)T.b * o.bs;
        const long long sb = (long long)T.b * P.std_bs;
        const int x = T.x0 + l31;
        if (x >= P.W || (abl & 4)) return;
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

    if (items <= 0) return;
    // stagger: the two workgroups that share a CU run the same program; started together they phase-lock (both in the
    // MFMA loop, then both staging).  Delay the second half of the grid by about half a step (diagnostic knob bit 16+).
    if ((abl >> 4) && (int)blockIdx.x >= (G >> 1)) {
        for (int q = 0; q < (abl >> 4); ++q) __builtin_amdgcn_s_sleep(127);
    }
    // diagnostic phase stamps: compiled in only under XSD_DIAG (diagnostic library variant)
    unsigned long long st[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long t0 = 0;
#ifdef XSD_DIAG
    const bool stamp = P.dbg != nullptr;
#else
    constexpr bool stamp = false;
#endif
    auto tick = [&](int i) {
        if (stamp) { const unsigned long long t = __builtin_readcyclecounter(); st[i] += t - t0; t0 = t; }
    };
    if (stamp) t0 = __builtin_readcyclecounter();
    // ---- prologue: stage item 0
    TileXY cur = tile_of(0);
    load_in(0, cur);
    load_w(0);
    store_in();
    store_w();
    __syncthreads();
    tick(0);

    int s = 0, k = 0; // step within tile, tile ordinal
#pragma unroll 1
    for (int it = 0; it < items; ++it) {
        const bool more = (it + 1 < items);
        const int s_next = (s + 1 == nsteps) ? 0 : s + 1;
        TileXY nxt = cur;
        if (more) {
            if (s_next == 0) nxt = tile_of(k + 1);
            if (!MULTI_OUT || s_next == 0) load_in(MULTI_OUT ? 0 : s_next, nxt);
            load_w(s_next);
        }
        tick(1);
        if (MULTI_OUT || s == 0) init_acc(MULTI_OUT ? s : 0);
        compute();
        tick(2);
        if (MULTI_OUT) epilogue(s, cur);
        else if (s == nsteps - 1) epilogue(0, cur);
        tick(3);
        if (more) {
            __syncthreads();
            tick(4);
            if (!MULTI_OUT || s_next == 0) store_in();
            store_w();
            __syncthreads();
            tick(5);
        }
        if (s_next == 0) { cur = nxt; ++k; }
        s = s_next;
    }
    if (stamp && tid == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) atomicAdd(&P.dbg[i], st[i]);
        atomicAdd(&P.dbg[6], (unsigned long long)items);
    }
}

static PerDevice g_once;

hipError_t launch_conv3x3_mfma(const ConvParams& p, hipStream_t stream)
{
    int ncu = 256;
    hipError_t e = g_once.once([]() {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mfma_kernel<false>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, CONV_LDS_TOTAL);
        if (e1 != hipSuccess) return e1;
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mfma_kernel<true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, CONV_LDS_TOTAL);
    }, &ncu);
    if (e != hipSuccess) return e;
    if (p.n_in < 1 || p.n_out < 1 || p.n_in * p.n_out > 5) return hipErrorInvalidValue;
    const int ntiles = p.B * p.tilesX * p.tilesY;
    if (ntiles <= 0) return hipSuccess;
    const int resident = 2 * ncu; // 2 workgroups per CU (LDS 81,024 B each)
    const dim3 g(ntiles < resident ? ntiles : resident), b(256);
    if (p.n_out > 1) hipLaunchKernelGGL((conv3x3_mfma_kernel<true>), g, b, CONV_LDS_TOTAL, stream, p);
    else hipLaunchKernelGGL((conv3x3_mfma_kernel<false>), g, b, CONV_LDS_TOTAL, stream, p);
    return hipGetLastError();
}

// diagnostic: occupancy API answer for the forward kernel at a given dynamic LDS size
int debug_conv_occupancy(int lds_bytes)
{
    int n = -1;
    if (g_once.once([]() { return hipSuccess; }, nullptr) != hipSuccess) return -2;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mfma_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return -2;
    const hipError_t q = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&conv3x3_mfma_kernel<false>), 256, lds_bytes);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mfma_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, CONV_LDS_TOTAL);
    return q == hipSuccess ? n : -3;
}

} // namespace xsd
