// generic_net.hip -- the RRDB generators at ARBITRARY widths: any num_filters, any in/out channel counts.
//
// The reference's constructors accept any widths (generator_rrdb.py:10-54; config/config.py:164-203 PositiveInt; the dense
// block's own default is nf = 64, rrdb_blocks.py:23); the split-precision MFMA path of this library takes whole 32-channel
// planes (32 filters = the shipped configuration, res/configs/models.toml; any width up to 256 filters, zero-padded to a multiple
// of 32; up to 8 image channels).  What lies beyond runs here, in exact fp32
// (the math mode of xsd_set_math does not apply), on NCHW tensors like the reference's: convs with at least 16 channels on
// both sides on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32, an fmaf chain: gconv_mfma_kernel, gwgrad_mfma_kernel),
// narrower ones (the image-side convs; every conv of an 8-filter net) as direct convolutions on the vector ALUs.  Same C
// ABI, same flat parameter layout, same backward stages as the MFMA path (xsd_engine.hip dispatches on the configuration).
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

// one output element through the fused epilogue (shared by the direct and the MFMA kernel)
__device__ __forceinline__ void gconv_store(const GConvP& P, int b, int co, int gy, int gx, float acc)
{
    const long long HW = (long long)P.H * P.W, pix = (long long)gy * P.W + gx;
    float v = (acc + (P.bias ? P.bias[co] : 0.f)) * P.a1;
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
    if (P.clamp01) v = v != v ? v : fminf(fmaxf(v, 0.f), 1.f);      // torch.clamp propagates NaN
    P.y[o] = v;
}

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
#pragma unroll
    for (int k = 0; k < GCO; ++k) {
        if (co0 + k >= P.cout) break;
        gconv_store(P, b, co0 + k, gy, gx, acc[k]);
    }
}

// The same convolution on the exact-fp32 matrix instruction, for convs with at least 16 channels on both sides:
// D[co][px] = sum over (ci, tap) of W[co][ci][tap] X[ci][px + tap] as v_mfma_f32_32x32x2_f32 products with K = two input
// channels per instruction: lane l supplies W[co = l % 32][ci = 2 c2 + l / 32][tap] and X[that ci][pixel l % 32 + tap shift].
// A workgroup (4 waves) owns 32 output channels x an 8 x 32 pixel tile, wave w the rows 2w, 2w + 1 (two accumulators); per
// block of 32 input channels the X halo tile ([32][10 x 34], channel stride 341 words) and the weight block ([9][32 ci][32 co],
// pre-arranged and zero-padded by gpack_blocks_kernel: one contiguous 36,864-B copy) go through LDS: 80.5 KB, two workgroups
// per CU.  Summation order: input channels ascending in pairs, taps inside; exact fp32 (the MFMA is an fmaf chain).
typedef float gf32x16 __attribute__((ext_vector_type(16)));
constexpr int GW_ROWS = 8, GW_COLS = 32;
constexpr int GW_XCH = (GW_ROWS + 2) * (GW_COLS + 2) + 1;     // 341 words per X channel
constexpr int GW_GCH = GW_ROWS * GW_COLS + 1;                 // 257 words per G channel (weight gradient)
constexpr int GM_LDS_BYTES = (32 * GW_XCH + 9 * 1024) * 4;    // 80,512

// Tile staging of the matrix-instruction kernels.  Loads are BUFFER loads against a descriptor of the (at most 32) channels
// being staged: channels past the tensor fail the hardware range check and read as zero, pixels outside the image get an
// offset that fails it too -- no select on the data, no access outside the tensor.  A thread requests several elements
// before it writes the first to LDS (memory-level parallelism; one element at a time is a round trip each): the weight
// gradient with its 144 accumulator registers affords GS_CHUNK = 4 and recomputes the offsets per tile; the conv, whose tile
// is fixed per workgroup, keeps its 43 offsets in registers and has 22 requests in flight.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t gchan_rsrc(const float* base, int nch, long long HW)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)((long long)(nch < 32 ? nch : 32) * HW * 4), 0x00020000);
}
// X halo tile [32][10 x 34] (channel stride GW_XCH) of the channels the descriptor covers
template <int GS_CHUNK>
__device__ __forceinline__ void gstage_x(float* xs, __amdgpu_buffer_rsrc_t rs, int y0, int x0, int H, int W, int HW, int tid)
{
    constexpr int PX = (GW_ROWS + 2) * (GW_COLS + 2), N = 32 * PX, K = (N + 255) / 256;
#pragma unroll 1
    for (int k0 = 0; k0 < K; k0 += GS_CHUNK) {
        float v[GS_CHUNK];
        int li[GS_CHUNK];
#pragma unroll
        for (int q = 0; q < GS_CHUNK; ++q) {
            if (k0 + q >= K) continue;
            const int i = tid + 256 * (k0 + q);
            const int c = i / PX, r = i - c * PX, hy = r / (GW_COLS + 2), hx = r - hy * (GW_COLS + 2);
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            const bool ok = i < N && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            v[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? (c * HW + gy * W + gx) * 4 : (int)0x80000000, 0, 0));
            li[q] = i < N ? i + c : -1;                    // c * GW_XCH + r
        }
#pragma unroll
        for (int q = 0; q < GS_CHUNK; ++q)
            if (k0 + q < K && li[q] >= 0) xs[li[q]] = v[q];
    }
}
// G tile [32][8 x 32] (channel stride GW_GCH)
template <int GS_CHUNK>
__device__ __forceinline__ void gstage_g(float* gs, int gch, __amdgpu_buffer_rsrc_t rs, int y0, int x0, int H, int W, int HW, int tid)
{
    constexpr int PX = GW_ROWS * GW_COLS, K = 32 * PX / 256;     // 32 elements per thread
    static_assert(K % GS_CHUNK == 0, "chunk");
#pragma unroll 1
    for (int k0 = 0; k0 < K; k0 += GS_CHUNK) {
        float v[GS_CHUNK];
#pragma unroll
        for (int q = 0; q < GS_CHUNK; ++q) {
            const int i = tid + 256 * (k0 + q), c = i / PX, r = i & (PX - 1);
            const int gy = y0 + (r >> 5), gx = x0 + (r & 31);
            v[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (gy < H && gx < W) ? (c * HW + gy * W + gx) * 4 : (int)0x80000000, 0, 0));
        }
#pragma unroll
        for (int q = 0; q < GS_CHUNK; ++q) {
            const int i = tid + 256 * (k0 + q), c = i / PX, r = i & (PX - 1);
            gs[c * gch + r] = v[q];
        }
    }
}

__global__ __launch_bounds__(256, 2) void gconv_mfma_kernel(const GConvP P)     // P.w: the conv's block-packed weights
{
    extern __shared__ __attribute__((aligned(16))) float gm_lds[];
    float* xs = gm_lds;
    float* wl = gm_lds + 32 * GW_XCH;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kk = lane >> 5;
    const int tilesX = (P.W + GW_COLS - 1) / GW_COLS;
    const int x0 = ((int)blockIdx.x % tilesX) * GW_COLS, y0 = ((int)blockIdx.x / tilesX) * GW_ROWS;
    const int cob = blockIdx.y, b = blockIdx.z;
    const int nci = (P.cin + 31) / 32;
    const long long HW = (long long)P.H * P.W;
    const float* xb = P.x + (long long)b * P.xbs;
    gf32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[r][v] = 0.f;
    // this thread's 43 elements of the X halo tile: byte offset inside a 32-channel block (fails the range check outside the image)
    // and LDS word, the same for every block of input channels -- computed once, kept in registers
    constexpr int PX = (GW_ROWS + 2) * (GW_COLS + 2), NX = 32 * PX, KX = (NX + 255) / 256;
    int xoff[KX], xli[KX];
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        const int i = tid + 256 * k;
        const int c = i / PX, r = i - c * PX, hy = r / (GW_COLS + 2), hx = r - hy * (GW_COLS + 2);
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool ok = i < NX && (unsigned)gy < (unsigned)P.H && (unsigned)gx < (unsigned)P.W;
        xoff[k] = ok ? (c * (int)HW + gy * P.W + gx) * 4 : (int)0x80000000;
        xli[k] = i < NX ? i + c : 32 * GW_XCH - 1;        // c * GW_XCH + r; past the tile: the last padding word
    }
    for (int cib = 0; cib < nci; ++cib) {
        __syncthreads();
        {
            const __amdgpu_buffer_rsrc_t rs = gchan_rsrc(xb + (long long)cib * 32 * HW, P.cin - cib * 32, HW);
#pragma unroll
            for (int k0 = 0; k0 < KX; k0 += 22) {        // 22 requests in flight per thread
                float v[22];
#pragma unroll
                for (int q = 0; q < 22; ++q)
                    if (k0 + q < KX) v[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, xoff[k0 + q], 0, 0));
#pragma unroll
                for (int q = 0; q < 22; ++q)
                    if (k0 + q < KX) xs[xli[k0 + q]] = v[q];
            }
        }
        const float4* wsrc = reinterpret_cast<const float4*>(P.w + ((long long)cob * nci + cib) * (9 * 1024));
        for (int i = tid; i < 9 * 256; i += 256) reinterpret_cast<float4*>(wl)[i] = wsrc[i];
        __syncthreads();
        const float* xl = xs + kk * GW_XCH + (2 * wave) * (GW_COLS + 2) + j;
        const float* wp = wl + kk * 32 + j;
#pragma unroll 2
        for (int c2 = 0; c2 < 16; ++c2) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float a = wp[t * 1024 + c2 * 64];
                const float b0 = xl[c2 * 2 * GW_XCH + (t / 3) * (GW_COLS + 2) + t % 3];
                const float b1 = xl[c2 * 2 * GW_XCH + (t / 3 + 1) * (GW_COLS + 2) + t % 3];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
            }
        }
    }
    // accumulator register v of lane l: output channel 8 (v / 4) + 4 (l / 32) + v % 4, pixel l % 32
    const int gx = x0 + j;
    if (gx >= P.W) return;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int gy = y0 + 2 * wave + r;
        if (gy >= P.H) continue;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int co = cob * 32 + 8 * (v >> 2) + 4 * kk + (v & 3);
            if (co < P.cout) gconv_store(P, b, co, gy, gx, acc[r][v]);
        }
    }
}

// weight gradient, stage 1: partial[part][co][ci][9] = sum over the part's tiles of g[b,co,p] * x[b,ci,p + tap];
// bias_partial[part][co] = sum g (workgroups of input-channel block 0).
//
// A GEMM per tap with K = pixels: dW_tap[co][ci] = sum_p G[co][p] X[ci][p + tap], on the exact-fp32 matrix instruction
// v_mfma_f32_32x32x2_f32 (two pixels per instruction; the fp32 MFMA is an fmaf chain, bitwise: MI355X_MICROARCH.md) -- lane l
// supplies G[co = l % 32][pixel pair member l / 32] once per pixel pair and the nine shifted X[ci = l % 32] values, nine
// 32 x 32 accumulators (144 registers) per wave.  A workgroup (4 waves) owns a block of 32 output x 32 input channels and the
// tiles t = part, part + parts, ... of 8 x 32 pixels; wave w takes the tile rows 2w, 2w + 1.  The X halo tile (32 channels x
// 10 x 34, zero outside the image and for channels >= cin) and the G tile (32 x 8 x 32, zero outside the image) go through LDS,
// channel strides padded to odd word counts (341 / 257: the 32 lanes of a read hit 32 banks); 76.5 KB, so two workgroups
// share a CU and one stages while the other multiplies.  The four waves' accumulators are summed in wave order through LDS.
// grid (ci blocks * co blocks, parts), 256 threads.
struct GWgradP {
    const float* x; long long xbs; int cin;
    const float* g; long long gbs; int cout;
    int B, H, W, parts;
    float* partial; float* bias_partial;
};
constexpr int GW_LDS_BYTES = 32 * (GW_XCH + GW_GCH) * 4;      // 76,544
static_assert(32 * (GW_XCH + GW_GCH) >= 9 * 1024 + 256, "the final reduction reuses the tile images");

__global__ __launch_bounds__(256, 2) void gwgrad_mfma_kernel(const GWgradP P)
{
    extern __shared__ __attribute__((aligned(16))) float gw_lds[];
    float* xs = gw_lds;
    float* gs = gw_lds + 32 * GW_XCH;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, m = lane & 31, kk = lane >> 5;
    const int nci = (P.cin + 31) / 32;
    const int cib = (int)blockIdx.x % nci, cob = (int)blockIdx.x / nci, part = blockIdx.y;
    const int tilesX = (P.W + GW_COLS - 1) / GW_COLS, tilesY = (P.H + GW_ROWS - 1) / GW_ROWS;
    const int ntiles = P.B * tilesY * tilesX;
    const long long HW = (long long)P.H * P.W;
    gf32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    float bsum = 0.f;
    for (int tile = part; tile < ntiles; tile += P.parts) {
        const int tx = tile % tilesX, ty = (tile / tilesX) % tilesY, b = tile / (tilesX * tilesY);
        const int x0 = tx * GW_COLS, y0 = ty * GW_ROWS;
        __syncthreads();                               // the previous tile's reads are done
        gstage_x<4>(xs, gchan_rsrc(P.x + (long long)b * P.xbs + (long long)cib * 32 * HW, P.cin - cib * 32, HW), y0, x0, P.H, P.W, (int)HW, tid);
        gstage_g<4>(gs, GW_GCH, gchan_rsrc(P.g + (long long)b * P.gbs + (long long)cob * 32 * HW, P.cout - cob * 32, HW), y0, x0, P.H, P.W, (int)HW, tid);
        __syncthreads();
        const float* xl = xs + m * GW_XCH + kk;
        const float* gl = gs + m * GW_GCH + kk;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int y = 2 * wave + r;
#pragma unroll 4
            for (int xp = 0; xp < GW_COLS / 2; ++xp) {
                const float a = gl[y * GW_COLS + 2 * xp];
                bsum += a;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float bv = xl[(y + t / 3) * (GW_COLS + 2) + 2 * xp + t % 3];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[t], 0, 0, 0);
                }
            }
        }
    }
    // the four waves' sums in wave order (fixed: bitwise reproducible), then out: accumulator register v of lane l is
    // dW[co = 8 (v / 4) + 4 (l / 32) + v % 4][ci = l % 32]
    float* red = gw_lds;
    float* bred = gw_lds + 9 * 1024;
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int i = (t * 16 + v) * 64 + lane;
                    red[i] = (w ? red[i] : 0.f) + acc[t][v];
                }
            bred[wave * 64 + lane] = bsum;
        }
    }
    __syncthreads();
    for (int i = tid; i < 9 * 1024; i += 256) {
        const int t = i >> 10, v = (i >> 6) & 15, l = i & 63;
        const int co = cob * 32 + 8 * (v >> 2) + 4 * (l >> 5) + (v & 3), ci = cib * 32 + (l & 31);
        if (co < P.cout && ci < P.cin) P.partial[(((long long)part * P.cout + co) * P.cin + ci) * 9 + t] = red[i];
    }
    if (cib == 0 && tid < 32 && cob * 32 + tid < P.cout) {
        float sb = 0.f;
        for (int w = 0; w < 4; ++w) sb += bred[w * 64 + tid] + bred[w * 64 + 32 + tid];
        P.bias_partial[(long long)part * P.cout + cob * 32 + tid] = sb;
    }
}
// the same partial sums for convs with fewer than 16 channels on a side (the 32 x 32 blocks of the kernel above would be
// mostly padding): one workgroup per (ci, co, part), fmaf chains, fixed-order LDS tree.  grid (cin, cout, parts), part =
// a contiguous pixel range.
__global__ __launch_bounds__(256) void gwgrad_direct_kernel(const GWgradP P)
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
struct GPackDesc { long long w_off, t_off, pf_off, pt_off; int cout, cin; };   // pf / pt: block-packed copies (-1: none)
__global__ void gpack_kernel(const float* params, float* wt, const GPackDesc* descs)
{
    const GPackDesc d = descs[blockIdx.y];
    const int n = d.cout * d.cin * 9;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const int t = e % 9, ci = (e / 9) % d.cin, co = e / (9 * d.cin);
        wt[d.t_off + ((long long)ci * d.cout + co) * 9 + (8 - t)] = params[d.w_off + e];
    }
}
// block-packed weights for gconv_mfma_kernel: [co block][ci block][tap][32 ci][32 co], zero-padded; blockIdx.z = 0: the conv
// itself (OIHW), 1: its input-gradient conv (W^T with flipped taps: output channels = the conv's inputs)
__global__ void gpack_blocks_kernel(const float* params, float* wblk, const GPackDesc* descs)
{
    const GPackDesc d = descs[blockIdx.y];
    const bool tr = blockIdx.z == 1;
    const long long dst = tr ? d.pt_off : d.pf_off;
    if (dst < 0) return;
    const int co_n = tr ? d.cin : d.cout, ci_n = tr ? d.cout : d.cin;      // channel counts of the conv being packed
    const int nci = (ci_n + 31) / 32, n = ((co_n + 31) / 32) * nci * 9 * 1024;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const int k = e & 31, c = (e >> 5) & 31, t = (e >> 10) % 9, blk = e / (9 * 1024);
        const int co = (blk / nci) * 32 + k, ci = (blk % nci) * 32 + c;
        float v = 0.f;
        if (co < co_n && ci < ci_n) v = tr ? params[d.w_off + ((long long)ci * d.cin + co) * 9 + (8 - t)] : params[d.w_off + ((long long)co * d.cin + ci) * 9 + t];
        wblk[dst + e] = v;
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
    hipFree(wt); hipFree(wblk); hipFree(descs_dev); hipFree(ws); hipFree(wg_partial); hipFree(wg_bias_partial);
}

GenericNet* GenericNet::create(const xsd_config& cfg)
{
    GenericNet* n = new GenericNet();
    n->cfg = cfg;
    n->nf = cfg.num_filters; n->cin = cfg.in_channels; n->cout = cfg.out_channels; n->blocks = cfg.num_res_blocks;
    n->sr = cfg.kind == XSD_KIND_SR;
    n->nup = n->sr ? cfg.num_upsample : 0;
    long long off = 0, toff = 0, boff = 0;
    auto mk = [&](int co, int ci) {
        GConvW c; c.w = off; off += (long long)co * ci * 9; c.b = off; off += co; c.cout = co; c.cin = ci; c.t = toff; toff += (long long)co * ci * 9;
        c.pf = c.pt = -1;
        if (c.wide()) {      // block-packed copies for the MFMA kernels: the conv and its input-gradient conv
            const long long blk = (long long)((co + 31) / 32) * ((ci + 31) / 32) * 9 * 1024;
            c.pf = boff; boff += blk; c.pt = boff; boff += blk;
        }
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
    auto add = [&](const GConvW& c) { GPackDesc q; q.w_off = c.w; q.t_off = c.t; q.pf_off = c.pf; q.pt_off = c.pt; q.cout = c.cout; q.cin = c.cin; d.push_back(q); };
    add(n->first);
    for (auto& c : n->rdb) add(c);
    add(n->trunk); add(n->last);
    for (auto& c : n->up) add(c);
    if (n->sr) add(n->hr);
    n->ndesc = (int)d.size();
    int maxw = 0;
    for (auto& q : d) maxw = std::max(maxw, q.cout * q.cin * 9);
    n->max_w = maxw;
    n->wblk_floats = boff;
    if (hipMalloc((void**)&n->wt, sizeof(float) * toff) != hipSuccess ||
        (boff && hipMalloc((void**)&n->wblk, sizeof(float) * boff) != hipSuccess) ||
        hipMalloc((void**)&n->descs_dev, sizeof(GPackDesc) * d.size()) != hipSuccess ||
        hipMemcpy(n->descs_dev, d.data(), sizeof(GPackDesc) * d.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc((void**)&n->wg_partial, sizeof(float) * (size_t)GenericNet::MAX_PARTS * maxw) != hipSuccess ||
        hipMalloc((void**)&n->wg_bias_partial, sizeof(float) * ((size_t)GenericNet::MAX_PARTS * maxw / 9 + 1)) != hipSuccess) {   // parts * cout <= capacity / (9 cin)
        delete n;
        return nullptr;
    }
    return n;
}

hipError_t GenericNet::pack(const float* dev_params, hipStream_t s)
{
    params = dev_params;
    hipLaunchKernelGGL(gpack_kernel, dim3(8, ndesc), dim3(256), 0, s, dev_params, wt, reinterpret_cast<const GPackDesc*>(descs_dev));
    if (wblk) hipLaunchKernelGGL(gpack_blocks_kernel, dim3(32, ndesc, 2), dim3(256), 0, s, dev_params, wblk, reinterpret_cast<const GPackDesc*>(descs_dev));
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
    if (c.wide() && (long long)H_ * W_ * 32 * 4 < (1ll << 31)) {      // at least 16 channels on both sides: the fp32 matrix instruction (32-bit byte offsets inside a 32-channel block)
        p.w = wblk + (transposed ? c.pt : c.pf);
        GCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gconv_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, GM_LDS_BYTES));
        const int tiles = ((W_ + GW_COLS - 1) / GW_COLS) * ((H_ + GW_ROWS - 1) / GW_ROWS);
        hipLaunchKernelGGL(gconv_mfma_kernel, dim3(tiles, (p.cout + 31) / 32, B_), dim3(256), GM_LDS_BYTES, s, p);
        return hipGetLastError();
    }
    const int tiles = ((W_ + GT - 1) / GT) * ((H_ + GT - 1) / GT);
    hipLaunchKernelGGL(gconv3x3_kernel, dim3(tiles, (p.cout + GCO - 1) / GCO, B_), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t GenericNet::wgrad(hipStream_t s, const GConvW& c, const float* x, long long xbs, const float* g, long long gbs, int B_, int H_, int W_, float* grads)
{
    GWgradP p;
    p.x = x; p.xbs = xbs; p.cin = c.cin; p.g = g; p.gbs = gbs; p.cout = c.cout; p.B = B_; p.H = H_; p.W = W_;
    p.partial = wg_partial; p.bias_partial = wg_bias_partial;
    // partial-sum buffer: MAX_PARTS parts of the largest conv, so a small conv may use more parts
    const long long cap = (long long)MAX_PARTS * max_w / ((long long)c.cout * c.cin * 9);
    int parts;
    if (c.wide() && (long long)H_ * W_ * 32 * 4 < (1ll << 31)) {
        const long long tiles = (long long)B_ * ((H_ + GW_ROWS - 1) / GW_ROWS) * ((W_ + GW_COLS - 1) / GW_COLS);
        const int blocks = ((c.cin + 31) / 32) * ((c.cout + 31) / 32);
        // enough workgroups for two per CU and a few rounds of them, at least four tiles each
        parts = (int)std::max<long long>(1, std::min<long long>(std::min<long long>(cap, 2048), std::min<long long>(tiles / 4, (2048 + blocks - 1) / blocks)));
        p.parts = parts;
        GCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gwgrad_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, GW_LDS_BYTES));
        hipLaunchKernelGGL(gwgrad_mfma_kernel, dim3(blocks, parts), dim3(256), GW_LDS_BYTES, s, p);
    } else {
        const long long N = (long long)B_ * H_ * W_;
        parts = (int)std::min<long long>(std::min<long long>(cap, MAX_PARTS), std::max<long long>(1, N / 2048));
        p.parts = parts;
        hipLaunchKernelGGL(gwgrad_direct_kernel, dim3(c.cin, c.cout, parts), dim3(256), 0, s, p);
    }
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
