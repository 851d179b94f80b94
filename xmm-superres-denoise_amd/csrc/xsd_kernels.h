// xsd_kernels.h -- device-side parameter blocks and host launch prototypes of the MI355X (gfx950) engine.
//
// All activations are "feature planes": fp32 NHWC tensors [B][H][W][32] (128 B per pixel = one full L2 line per
// pixel per plane).  The reference's torch.cat over dense-block inputs (rrdb_blocks.py:49-52) never materialises:
// a conv simply walks a LIST of planes, one 32-channel chunk per K-step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <mutex>

namespace xsd {

// One-time per-DEVICE setup of a launcher (hipFuncSetAttribute of the dynamic LDS size is per device; so is the CU count):
// keyed by the current device id, serialised by a mutex, so a process that drives several GPUs, or several host threads,
// gets the attribute set and the right grid size on each of them.
struct PerDevice {
    static constexpr int MAXDEV = 64;
    std::mutex mu;
    bool done[MAXDEV] = {};
    int ncu[MAXDEV] = {};
    template <class F> hipError_t once(F&& setup, int* ncu_out)
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= MAXDEV) return hipErrorInvalidDevice;
        std::lock_guard<std::mutex> lk(mu);
        if (!done[dev]) {
            if ((e = setup()) != hipSuccess) return e;
            hipDeviceProp_t prop;
            if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
            ncu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            done[dev] = true;
        }
        if (ncu_out) *ncu_out = ncu[dev];
        return hipSuccess;
    }
};

// Workgroups of a persistent conv launch over `ntiles` tiles on `ncu` compute units (one workgroup per CU; workgroup b walks tiles
// b, b + G, b + 2G, ...).  The launch lasts as long as its longest workgroup: rounds = ceil(ntiles / ncu) tiles.  With the full grid the
// last round is run by the ntiles - (rounds - 1) ncu workgroups that have one tile more than the others; the BALANCED grid is the smallest
// one that needs no more rounds -- every workgroup gets the same number of tiles and the CUs that would have run one tile fewer stay dark:
// 338 tiles (a 416 x 416 image, the reference's tile at its default batch 1) run as 169 x 2 instead of 82 x 2 + 174 x 1.  Same critical
// path in tiles, measured faster, because off the average power bound the chip still clocks by how many CUs run matrix instructions at
// once (forced-grid stamps: in-kernel clock 1.79 GHz with 256 workgroups in flight, 1.95 with 169, 2.22 with 128): dark CUs buy the busy
// ones clock.  Scan after the one-atomic-per-workgroup fix (profiles/r06_grid_scan.txt, scan 4; DN forward, tiles/s, full -> balanced):
// 338 tiles 369 -> 388 (+5 %), 288: 377 -> 426 (+13 %), 450: 331 -> 335, 676 (3 rounds): 470 -> 483 (+2.7 %), 576: 502 -> 526 (+4.8 %), 1352 (6
// rounds: the reference's training batch of four): 493 -> 512 (+3.8 %), 1300: 514 -> 531; every grid between the balanced and the full one lies
// between the two, every grid that needs a round more is far worse (512 tiles: 256 -> 306, 192 -> 276).  From 9 rounds up the two forms are
// within +- 1 % and the full grid is kept (the bench batch, 64 rounds, never takes this path).
// (The first three scans of the round were taken while every MFMA wave published its plane maximum with a global atomic of its own --
// 2,048 same-address atomics, 22 us, at the end of every launch -- and a smaller grid also meant fewer atomics: they showed optima ABOVE the
// balanced grid and a third round paying for two full ones; both were artefacts of that and went with it: docs/LAB_NOTEBOOK.md R6.14.)
// Results are bitwise identical either way (a tile's arithmetic does not depend on which workgroup runs it).
constexpr int BALANCED_GRID_MAX_ROUNDS = 8;
inline int persistent_grid(int ntiles, int ncu)
{
    if (ntiles <= ncu) return ntiles;
    const int rounds = (ntiles + ncu - 1) / ncu;
    const int balanced = (ntiles + rounds - 1) / rounds;
#if defined(XSD_GRID_ENV)       // experiment builds only (make exp EXPFLAGS=-DXSD_GRID_ENV): the grid is min(ntiles, $XSD_EXP_GRID) -- tools/grid_scan.sh
    if (const char* g_ = getenv("XSD_EXP_GRID")) { const int v_ = atoi(g_); if (v_ > 0) return ntiles < v_ ? ntiles : (v_ < ncu ? v_ : ncu); }
#endif
#if defined(XSD_GRID_MODE)      // experiment builds (make exp EXPFLAGS=-DXSD_GRID_MODE=0 | 1 | 2): never / always balance / one round MORE than needed
    if ((XSD_GRID_MODE) == 2) return (ntiles + rounds) / (rounds + 1);
    return (XSD_GRID_MODE) ? balanced : ncu;
#endif
    return rounds <= BALANCED_GRID_MAX_ROUNDS ? balanced : ncu;
}

constexpr int TILE_H = 8;          // output rows per workgroup
constexpr int TILE_W = 32;         // output cols per workgroup (= MFMA M)
constexpr int HALO_W = TILE_W + 2; // 34
constexpr int HALO_H = TILE_H + 2; // 10
constexpr int HALO_PX = HALO_W * HALO_H;        // 340
constexpr int IN_LDS_BYTES = HALO_PX * 128;     // 43,520  (XOR-swizzled 16-B chunks, no padding)
constexpr int PANEL_FLOATS = 9 * 32 * 32;       // 9,216 floats per (K-chunk x N-chunk) weight panel
constexpr int W_LDS_BYTES = PANEL_FLOATS * 4;   // 36,864
constexpr int CONV_LDS_BYTES = IN_LDS_BYTES + W_LDS_BYTES; // 80,384 -> 2 workgroups per CU (160 KiB LDS)
// math mode 3 (bf16x6, conv3x3_s3x.hip): a weight half-panel in LDS, split into three bf16 terms: [tap][hi|mid|lo][64 lanes][8 bf16]
constexpr int S3_WH_BYTES = 9 * 3 * 1024;           // 27,648
// math mode 4 (f16x3, conv3x3_h2x.hip): two fp16 terms: [tap][hi|lo][64 lanes][8 f16]
constexpr int H2_WH_BYTES = 9 * 2 * 1024;           // 18,432

// input plane reference with general strides (pixel-shuffled reads use rs = 2*Whr*32, ps = 64)
struct PlaneIn {
    const float* p;
    long long bs; // batch stride (floats)
    int rs;       // row stride (floats)
    int ps;       // pixel stride (floats)
};

// One 32-channel output chunk and its fused epilogue:
//   v = acc(+bias); v *= a1; if(accumulate) v += dst; if(e1) v += s1*e1; v *= a2; if(e2) v += s2*e2;
//   if(e3) v += s3*e3; v = v>0 ? v : v*slope; if(mask) v = mask>0 ? v : v*mslope; dst = v
// e1/e2/e3/mask are standard planes ([B][H][W][32] at the conv's own H,W).
struct OutDesc {
    float* p;
    long long bs;
    int rs, ps;
    const float* e1;
    const float* e2;
    const float* e3;
    const float* mask;
    float a1, s1, a2, s2, s3, slope, mslope;
    int accumulate;
    // Compact lrelu' masks (split math modes, training): bit 4q + t of the 16-bit word [pixel][lane half h] says whether the
    // stored value of channel 8q + 4h + t is > 0.  A conv that produces an activation writes them (bits_out, 4 B per pixel);
    // the input-gradient conv masked by that activation reads them (bits_in) instead of the 128 B-per-pixel plane.
    unsigned short* bits_out;
    const unsigned short* bits_in;
    // math mode 4 (f16x3): slot that receives max |stored value| of this plane (atomic max of the float bits), or null
    float* amax;
};

struct ConvParams {
    int B, H, W;
    int n_in, n_out;     // (k,1): K-loop over input planes;  (1,k): one input tile, k output chunks
    int tilesX, tilesY;
    int std_rs;          // W*32
    long long std_bs;    // H*W*32
    PlaneIn in[5];
    const float* wpanel; // [n_out][n_in] panels (one of the two is 1), PANEL_FLOATS each (host side; kernels read wstep)
    const float* wstep[5]; // weight panel of each step (K-loop step or output chunk)
    const float* bias;   // [n_out*32] or nullptr
    int ablate;          // diagnostic ablation bits (0 in production; env XSD_ABLATE): 1 skip input loads, 2 skip weight loads,
                         // 4 skip the epilogue, 8 skip the MFMA loop; role-split kernel only: 16 two of three products,
                         // 32 idle sleep instead of the MFMA loop (with 8), 64 epilogue without its stores, 128 permuted
                         // fully coalesced stores, 256 fill LDS with realistic operand bits, 512 stores into an
                         // L2-resident 64 KiB window per workgroup
    int pad_;
    const void* zero;    // 256 B of zeros + 256 B of trash in HBM (epilogue operand loads / stores of lanes outside the image)
    unsigned long long* dbg; // diagnostic phase stamps (null in production): [grid][8] accumulated shader cycles
    // math mode 4 (f16x3): max |x| of each input plane and of the weight panels of this launch (device slots written by
    // the producers, read at kernel start); the kernel scales both operands by powers of two into the fp16 range
    const float* amax_in[5];
    const float* amax_w;
    OutDesc out[5];
};

// weight-gradient kernel: dW_j[tap][ci][co] += sum_px X_j[px+tap][ci] * G[px][co]
struct WgradParams {
    int B, H, W;
    int n_in;            // number of 32-channel input chunks X_j
    int n_g;             // number of 32-channel G chunks (1, or 4 for the pixel-shuffle conv)
    int tilesX, tilesY, nparts;
    PlaneIn x[5];
    PlaneIn g[5];
    float* partial;      // [nparts][n_g][n_in][9][32 ci][32 co]; pair-list launches: [nparts][npairs][9][32 ci][32 co]
    float* bias_partial; // [nparts][n_g][32]
    const void* zero;    // zero page
    int ablate;          // diagnostic (env XSD_ABLATE): 4096 = request the G tile only for the first tile of a workgroup
    int pad_;
    unsigned long long* dbg; // diagnostic phase stamps (null in production): slots [8..15] of the engine's stamp buffer
    const float* amax_x[5]; // math mode 4 (f16x3): max |x| slots of the X and G planes
    const float* amax_g[5];
    // Pair-list launch (wgrad_h2x only; npairs = 0: the full n_in x n_g rectangle): workgroup slot s of a part multiplies X plane
    // (pair_j >> 4s) & 15 with G plane (pair_n >> 4s) & 15.  All slots of a part run side by side on one XCD and walk the same
    // tiles, so every X and every G tile of the launch comes from HBM once and from that XCD's L2 for the other pairs that use it
    // (a dense block's 15 (X_j, G_n) pairs: 10 planes read instead of 20).  Grid = nparts * npairs workgroups, nparts a multiple of 8.
    int npairs, pad2_;
    unsigned long long pair_j, pair_n;
};

struct WgradReduceParams {
    const float* partial;
    const float* bias_partial;
    int nparts, n_in, n_g;
    int cin_total, cout_total;
    int shuffle;   // 1: output channel oc = 4*(32*plane + co) + n (pixel-shuffle conv, n = sub-pixel), else oc = 32*(n0 + n) + co
    int j0, n0;    // first input plane / first output chunk of this launch inside the conv (wide nets launch a conv's blocks in groups)
    int plane;     // shuffle: which 32-channel plane of the high-resolution tensor the four sub-pixel gradients belong to
    float scale;
    float* dw;     // OIHW [cout_total][cin_total][3][3]
    float* db;     // [cout_total]
    long long part_stride;   // floats between two parts' partial sums (0: n_g * n_in * 9 * 1024, the launch's own rectangle)
    int bias_stride;         // entries between two parts' bias sums (0: n_g * 32)
    int pad_;
};

struct WgradReduceBatch {   // the reductions of up to five convs as ONE launch (a dense block's pair-list weight gradient)
    static constexpr int MAXN = 5;
    int n, pad_;
    WgradReduceParams r[MAXN];
};

// pack descriptor: one conv's OIHW weights -> forward and transposed (dgrad) panels
struct PackDesc {
    long long src_w;     // offset into flat params
    long long dst_fwd;   // offset into packed fwd buffer
    long long dst_bwd;   // offset into packed dgrad buffer
    int cout, cin;       // multiples of 32
    int shuffle;
    float bwd_scale;     // folded into the transposed (input-gradient) panels: 0.2 / 0.04 for conv5 (rrdb_blocks.py:54,70)
};

// edge layers (Cin=1 or Cout=1): HBM-bound VALU kernels
struct EdgeExpandParams { // 1 -> 32 conv:  out[p][c] = bias[c] + sum_tap s[p+tap] * w[tap][c]; optional lrelu' mask
    int B, H, W;
    const float* s;      // [B][H][W]
    const float* w;      // [9][32]
    const float* bias;   // [32] or null
    float* out;          // [B][H][W][32]
    const float* mask;   // plane or null
    float mslope;
    const unsigned short* bits; // compact form of `mask` (OutDesc::bits_out layout) or null
    long long s_bs;      // floats between consecutive images of `s` (0: H*W; C*H*W when `s` is one channel of a [B][C][H][W] image)
    int accumulate;      // add to what `out` holds instead of starting from the bias (second and later image channels)
    float* amax;         // math mode 4: slot that receives max |out| (atomic max over non-negative floats as integers), or null --
                         // the consumers scale their operands by it; without it the engine sweeps the plane once more (plane_amax_kernel)
};
struct EdgeReduceParams { // 32 -> 1 conv: pre[p] = bias + sum_tap sum_c f[p+tap][c] * w[tap][c] (+ skip[p]); y = clamp(pre)
    int B, H, W;
    const float* f;      // [B][H][W][32]
    const float* w;      // [9][32]
    const float* bias;   // [1] or null
    const float* skip;   // [B][H][W] or null
    float* pre;          // pre-clamp value (may be null)
    float* y;            // output
    int clamp01;
    const float* addto;  // y = value + addto[p] (used for dx = dgrad + skip-grad), may be null
    long long y_bs;      // floats between consecutive images of y / pre / addto (0: H*W; C*H*W for one channel of a [B][C][H][W] image)
    long long skip_bs;   // the same for `skip`
};
constexpr int EDGE_BAND = 32;     // rows per workgroup of the 32 -> 1 edge conv
constexpr int EDGE_MAX_W = 4096;  // widest image row the edge kernels stage in LDS
struct EdgeWgradParams { // out[tap][c] = sum_p f[p][c] * s[p+tap]; bsum[c] = sum_p f[p][c]; ssum = sum_p s[p]
    int B, H, W;
    const float* f;
    const float* s;
    float* partial;      // [nblocks][9*32 + 32 + 1]
    int nblocks;
    long long s_bs;      // floats between consecutive images of `s` (0: H*W)
};

} // namespace xsd
