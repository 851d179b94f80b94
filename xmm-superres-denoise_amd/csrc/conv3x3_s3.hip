// conv3x3_s3.hip -- math mode 3 ("bf16x6"): the 3x3 conv (forward + input-gradient) over fp32 feature planes with
// fp32-CLASS arithmetic on the bf16 matrix cores.
// Reference layers: nn.Conv2d(32k -> 32, 3, 1, 1) of rrdb_blocks.py:27-31, generator_rrdb.py:38-44,95,101 (fp32) and
// their autograd input-gradients.
//
// Arithmetic.  Every fp32 operand is split EXACTLY into three bf16 terms, x = hi + mid + lo (round-to-nearest at each
// step; 8 + 8 + 8 significant bits cover fp32's 24), and a product w*x is evaluated as the six bf16 products
//     wh*xh + wh*xm + wm*xh + wh*xl + wl*xh + wm*xm        (dropped: wm*xl + wl*xm + wl*xl <= 2^-23 |w*x|)
// on v_mfma_f32_32x32x16_bf16.  That instruction adds its 16 exact products and the fp32 accumulator in a wide internal
// format and rounds ONCE (measured: profiles/r02_mfma_probe.txt, tools/mfma_probe.hip).  The hi*hi products go to one
// accumulator (one rounding per 16-channel k-step, 16x fewer than an fp32 fma chain), the five cross products, 2^-8 of
// it, to a second one; they meet in the epilogue.  Against float64 this is MORE accurate than exact-fp32 MFMA and than
// torch's fp32 conv on the reference networks (tests/test_hip_precision.py), at 16/6 = 2.7x the fp32 matrix rate
// (2500 / 6 = 417 TFLOP/s effective peak).  Planes, biases, residual adds and the epilogue stay fp32.
//
// Structure.  One workgroup of 512 threads (8 waves, 2 per SIMD) per CU, tile = 16 x 32 pixels, wave w owns rows 2w,
// 2w+1.  A plane step is two 16-channel HALF-steps (one MFMA k-step each):
//   * input: the 18 x 34 halo half-tile is fetched as fp32 (64 B per pixel, one float4 per lane = fully used 64-B
//     segments) into registers a half-step ahead, split by the VALU and written as three [pixel][16 x bf16] images
//     (32 B per pixel, 16-B slots XOR-swizzled by (hx >> 3) & 1 -> conflict-free ds_read_b128 for every tap shift) into
//     the OTHER of two LDS buffers while the current one is being multiplied: the five conversion rounds sit between MFMA
//     stages, each followed by the global load that refills its register for the half-step after next, and the two waves
//     of a SIMD do them at different stages so that one's VALU work lies beside the other's MFMAs;
//   * weights: fp32 half-panels in fragment order (pack_weights_s3_kernel, 18 KB instead of 27 KB of pre-split terms: the
//     weight stream is as large as the input stream at this tile size), fetched into registers at the start of the
//     half-step, split and written between the two barriers that end it (single LDS buffer of 27,648 B);
//   * plane / panel descriptors are read from a small LDS table (a runtime index into the kernel argument costs a scalar
//     memory round trip per use); the epilogue's operand combinations are straight-line variants, and its stores are
//     deferred into the next half-step's MFMA loop (see the epilogue);
//   * barriers are raw s_barrier + lgkmcnt(0): the register prefetch stays in flight across them;
//   * per half-step and wave: 9 taps x 2 rows x 6 MFMAs, fragments requested one (tap,row) stage ahead.
// K-loop over input planes (n_in x 1) and one-input/many-output (1 x n_out, re-staging the input per output chunk) run
// through the same step sequence.  LDS: 2 x 60,288 (input) + 27,648 (weights) + 640 (bias) + 128 = 148,992 B.
#include <cstdlib>
#include <type_traits>
#include "xsd_kernels.h"
#include "xsd_split.h"

namespace xsd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// Prefetch loads go through pointers rebuilt from integers (LDS descriptor table, zero page): without an explicit global
// address space hipcc emits flat_load, which returns out of order with LDS traffic and forces vmcnt(0)/lgkmcnt(0) waits.
typedef const __attribute__((address_space(1))) f32x4* gf32x4p;
typedef __attribute__((address_space(1))) f32x4* gf32x4p_w;
__device__ __forceinline__ f32x4 gload4(const float* p) { return *(gf32x4p)p; }

constexpr int T3_THREADS = 512;
constexpr int T3_ROWS = 16;                         // tile rows
constexpr int T3_PX = (T3_ROWS + 2) * HALO_W;       // 612 halo pixels
constexpr int T3_SINK = T3_PX * 32;                 // 19,584: each term image ends with a 512-B sink (64 lanes x 8 B) that takes
constexpr int T3_XT = T3_SINK + 512;                // 20,096 B per term image  the writes of staging slots beyond the half-tile
constexpr int T3_XB = 3 * T3_XT;                    // 60,288 B per input buffer
constexpr int T3_WOFF = 2 * T3_XB;                  // 120,576
constexpr int T3_BIAS = T3_WOFF + S3_WH_BYTES;      // 148,224
constexpr int T3_DESC = T3_BIAS + 5 * 32 * 4;       // 148,864: 5 x {plane base, batch stride (bytes)}, 5 x weight panel pointer
constexpr int T3_LDS_BYTES = T3_DESC + 16 * 8;      // 148,992
constexpr int T3_ROWB = HALO_W * 32;                // 1088 B per halo row of a term image
constexpr int T3_XSLOTS = T3_PX * 4;                // (pixel, channel quad) slots of one float4
constexpr int T3_XR = (T3_XSLOTS + T3_THREADS - 1) / T3_THREADS;   // 5
constexpr int T3_WSLOTS = 9 * 64 * 2;               // float4 slots of an fp32 half-panel (18,432 B)
constexpr int T3_WR = (T3_WSLOTS + T3_THREADS - 1) / T3_THREADS;   // 3

// exact 3-term split of 4 fp32 values into packed bf16 pairs
__device__ __forceinline__ void s3_split4(const f32x4& a, u32x2& hi, u32x2& mid, u32x2& lo) { split3_f32x4(a, hi, mid, lo); }   // xsd_split.h

__global__ __launch_bounds__(T3_THREADS, 2) void conv3x3_s3_kernel(const ConvParams P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* w_lds = smem + T3_WOFF;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hw = wv >> 2;            // waves w and w+4 share a SIMD: the second half staggers its staging work
    const int h = lane >> 5;
    const int l31 = lane & 31;

    const int tilesY = (P.H + T3_ROWS - 1) / T3_ROWS;
    const int ntiles = P.B * tilesY * P.tilesX;
    const int n_in = P.n_in, n_out = P.n_out;
    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
    const int items = my_tiles * n_out * n_in * 2;     // half-steps of this workgroup
    if (items <= 0) return;

    struct TileXY { int b, y0, x0; };
    auto tile_of = [&](int k) {
        int t = (int)blockIdx.x + k * G;
        TileXY r;
        const int tx = t % P.tilesX; t /= P.tilesX;
        r.x0 = tx * TILE_W; r.y0 = (t % tilesY) * T3_ROWS; r.b = t / tilesY;
        return r;
    };
    struct Cur { int j, i, s2, k; };   // output chunk, input plane, channel half, tile ordinal
    auto succ = [&](Cur c) {
        c.s2 ^= 1;
        if (c.s2 == 0 && ++c.i == n_in) { c.i = 0; if (++c.j == n_out) { c.j = 0; ++c.k; } }
        return c;
    };

    // per-lane LDS read bases (one per tap column): lane = (pixel column l31, k-half h) reads 16-B slot h of its pixel
    int abase[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int hx = l31 + dx;
        abase[dx] = (wv * 2) * T3_ROWB + hx * 32 + ((h ^ ((hx >> 3) & 1)) << 4);
    }

    // input staging slots of this thread (pixel, channel quad): LDS offsets are tile independent, global offsets per tile
    int lds_slot[T3_XR];
#pragma unroll
    for (int r = 0; r < T3_XR; ++r) {
        const int slot = r * T3_THREADS + tid;
        const int p = slot >> 2, q = slot & 3;
        const int hx = p % HALO_W;
        lds_slot[r] = slot < T3_XSLOTS ? p * 32 + (((q >> 1) ^ ((hx >> 3) & 1)) << 4) + (q & 1) * 8 : T3_SINK + lane * 8;
    }
    int goff[T3_XR];
    auto tile_offsets = [&](const TileXY& T, int rs, int ps) {
#pragma unroll
        for (int r = 0; r < T3_XR; ++r) {
            const int slot = r * T3_THREADS + tid;
            const int p = slot >> 2, q = slot & 3;
            const int hy = p / HALO_W, hx = p - hy * HALO_W;
            const int gy = T.y0 - 1 + hy, gx = T.x0 - 1 + hx;
            const bool ok = (slot < T3_XSLOTS) && (gy >= 0) && (gy < P.H) && (gx >= 0) && (gx < P.W);
            goff[r] = ok ? gy * rs + gx * ps + q * 4 : -1;
        }
    };

    // Loads are UNCONDITIONAL (padding / out-of-range slots read a page of zeros): a branch around a load makes hipcc wait
    // for it before the next one, which serialises the prefetch into dependent round trips.
    const float* zero = reinterpret_cast<const float*>(P.zero);
    f32x4 pin[T3_XR];
    f32x4 pw[T3_WR];
    // plane / panel descriptors through LDS: indexing the kernel argument with a runtime index costs a scalar-memory round
    // trip (and a stall) per use
    unsigned long long* desc = reinterpret_cast<unsigned long long*>(smem + T3_DESC);
    if (tid < 5) {
        desc[2 * tid] = reinterpret_cast<unsigned long long>(P.in[tid].p);
        desc[2 * tid + 1] = (unsigned long long)P.in[tid].bs * 4ull;
        desc[10 + tid] = reinterpret_cast<unsigned long long>(P.wstep[tid]);
    }
    auto x_base = [&](const Cur& c, const TileXY& T) {
        const unsigned long long p = desc[2 * c.i], bs = desc[2 * c.i + 1];
        return reinterpret_cast<const float*>(p + (unsigned long long)T.b * bs) + c.s2 * 16;
    };
    // `base` == nullptr: nothing left to fetch (the last two half-steps of the workgroup): every lane reads the zero page
    auto load_x_round = [&](int r, const float* base) {
        pin[r] = gload4((goff[r] >= 0 && base) ? base + goff[r] : zero);
    };
    auto store_x_round = [&](int r, char* xb) {   // unconditional: slots beyond the half-tile land in the sinks
        u32x2 hi, mid, lo;
        s3_split4(pin[r], hi, mid, lo);
        *reinterpret_cast<u32x2*>(xb + lds_slot[r]) = hi;
        *reinterpret_cast<u32x2*>(xb + T3_XT + lds_slot[r]) = mid;
        *reinterpret_cast<u32x2*>(xb + 2 * T3_XT + lds_slot[r]) = lo;
    };
    // weight half-panel: fp32 [tap][lane][8] -> float4 slot s = (tap*64 + lane)*2 + half
    auto load_w = [&](const Cur& c) {
        const float* src = reinterpret_cast<const float*>(desc[10 + c.j * n_in + c.i]) + c.s2 * (PANEL_FLOATS / 2);
#pragma unroll
        for (int r = 0; r < T3_WR; ++r) {
            const int s = r * T3_THREADS + tid;
            pw[r] = gload4(src + 4 * (s < T3_WSLOTS ? s : T3_WSLOTS - 1));
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int r = 0; r < T3_WR; ++r) {
            const int s = r * T3_THREADS + tid;
            if (s < T3_WSLOTS) {
                const int frag = s >> 7, ln = (s >> 1) & 63, sub = s & 1;   // frag = tap
                u32x2 hi, mid, lo;
                s3_split4(pw[r], hi, mid, lo);
                char* d = w_lds + frag * 3 * 1024 + ln * 16 + sub * 8;
                *reinterpret_cast<u32x2*>(d) = hi;
                *reinterpret_cast<u32x2*>(d + 1024) = mid;
                *reinterpret_cast<u32x2*>(d + 2048) = lo;
            }
        }
    };

    // bias through LDS (a global load behind the prefetch would make its consumer wait for every older VMEM op)
    float* bias_lds = reinterpret_cast<float*>(smem + T3_BIAS);
    if (tid < 160) bias_lds[tid] = (P.bias && tid < 32 * n_out) ? P.bias[tid] : 0.f;
    __syncthreads();

    // Two accumulators per tile row: acc takes the hi*hi products, accx the five cross products (see header).
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

    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // Epilogue over fp32 planes: each lane owns one pixel and 16 channels as four float4 groups -> 16-B loads/stores;
    // lanes l and l+32 cover adjacent 16-B chunks, so every store instruction writes 32 x 32 contiguous bytes.
    // * The operand combinations the plans use are compiled as straight-line variants (all loads of a row issued together):
    //   a branch around a load makes hipcc wait for that load before the next one is issued.  Lanes outside the image read
    //   the zero page and write a trash page, so there is no divergent control flow either.
    // * The STORES are deferred: a CU's store path takes about 1 KiB per 100 cycles, so eight waves storing a tile at once
    //   hold each other (and the matrix pipe) up for thousands of cycles.  The tile's results wait in 32 registers and are
    //   written one float4 group at a time between the MFMA stages of the NEXT half-step.
    f32x4 pend[8];
    float* pend_dp[2];
    float* const trash = const_cast<float*>(zero) + 64 + 4 * h;
    auto epilogue_v = [&](const OutDesc& o, const TileXY& T, auto has_acc, auto has_e1, auto has_e2, auto has_e3, auto has_mask, auto generic) {
        float* dst = o.p + (long long)T.b * o.bs;
        const long long sb = (long long)T.b * P.std_bs;
        const int x = T.x0 + l31;
        // generic variant: absent operands read zeros (scale 0 / mask slope 1 make them neutral)
        const float s1 = (decltype(generic)::value && !o.e1) ? 0.f : o.s1, s2v = (decltype(generic)::value && !o.e2) ? 0.f : o.s2;
        const float s3 = (decltype(generic)::value && !o.e3) ? 0.f : o.s3, msl = (decltype(generic)::value && !o.mask) ? 1.f : o.mslope;
        const float sacc = (decltype(generic)::value && !o.accumulate) ? 0.f : 1.f;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int y = T.y0 + wv * 2 + r;
            const bool valid = x < P.W && y < P.H;
            float* dp = valid ? dst + (long long)y * o.rs + (long long)x * o.ps + 4 * h : trash;
            const long long os = sb + (long long)y * P.std_rs + x * 32 + 4 * h;
            auto opnd = [&](const float* plane) { return (valid && plane) ? plane + os : zero; };
            const float *p1 = opnd(o.e1), *p2 = opnd(o.e2), *p3 = opnd(o.e3), *pm = opnd(o.mask);
            const float* pa = (valid && o.accumulate) ? dp : zero;
            f32x4 va[4], v1[4], v2[4], v3[4], vm[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (decltype(has_acc)::value) va[q] = gload4(pa + 8 * q);
                if constexpr (decltype(has_e1)::value) v1[q] = gload4(p1 + 8 * q);
                if constexpr (decltype(has_e2)::value) v2[q] = gload4(p2 + 8 * q);
                if constexpr (decltype(has_e3)::value) v3[q] = gload4(p3 + 8 * q);
                if constexpr (decltype(has_mask)::value) vm[q] = gload4(pm + 8 * q);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = (acc[r][4 * q + t] + accx[r][4 * q + t]) * o.a1;
                if constexpr (decltype(has_acc)::value) v += sacc * va[q];
                if constexpr (decltype(has_e1)::value) v += s1 * v1[q];
                v *= o.a2;
                if constexpr (decltype(has_e2)::value) v += s2v * v2[q];
                if constexpr (decltype(has_e3)::value) v += s3 * v3[q];
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = v[t] > 0.f ? v[t] : v[t] * o.slope;
                if constexpr (decltype(has_mask)::value) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = vm[q][t] > 0.f ? v[t] : v[t] * msl;
                }
                pend[4 * r + q] = v;
            }
            pend_dp[r] = dp;
        }
    };
    auto epilogue = [&](int j, const TileXY& T) {
        const OutDesc o = P.out[j];
        using Y = std::true_type; using N = std::false_type;
        const int kind = (o.accumulate ? 1 : 0) | (o.e1 ? 2 : 0) | (o.e2 ? 4 : 0) | (o.e3 ? 8 : 0) | (o.mask ? 16 : 0);
        switch (kind) {
        case 0: epilogue_v(o, T, N{}, N{}, N{}, N{}, N{}, N{}); break;
        case 2: epilogue_v(o, T, N{}, Y{}, N{}, N{}, N{}, N{}); break;
        case 6: epilogue_v(o, T, N{}, Y{}, Y{}, N{}, N{}, N{}); break;
        case 14: epilogue_v(o, T, N{}, Y{}, Y{}, Y{}, N{}, N{}); break;
        case 16: epilogue_v(o, T, N{}, N{}, N{}, N{}, Y{}, N{}); break;
        default: epilogue_v(o, T, Y{}, Y{}, Y{}, Y{}, Y{}, Y{}); break;   // any other combination (none in the engine's plans)
        }
    };
    auto store_pending = [&](int c) { *(gf32x4p_w)(pend_dp[c >> 2] + 8 * (c & 3)) = pend[c]; };

#ifdef XSD_DIAG   // ablation bits of the diagnostic library (timing experiments only; results are garbage when set)
    const int abl = P.ablate;
#else
    constexpr int abl = 0;
#endif
    // One half-step: 9 taps x 2 rows x 6 products from input buffer `xc`, fragments requested one stage ahead; between the
    // stages: conversion round r of the NEXT half-step's input (registers -> buffer `xn`), then the load that refills the
    // register with the half-step after next.  No sched_barrier pins here: left alone, hipcc spreads the conversion VALU
    // and the LDS traffic between the MFMAs (with pins every conversion round ran as one VALU block between MFMA groups).
    const char* wl = w_lds + lane * 16;
    auto compute = [&](auto hw_c, const char* xc, char* xn, const float* base2, bool drip) {
        constexpr int HW = decltype(hw_c)::value;
        bf16x8 bf[2][3], af[2][3];   // [set][term]: 0 = hi, 1 = mid, 2 = lo
        auto load_b = [&](int tap, bf16x8 (&b)[3]) {
#pragma unroll
            for (int t = 0; t < 3; ++t) b[t] = *reinterpret_cast<const bf16x8*>(wl + (tap * 3 + t) * 1024);
        };
        auto load_a = [&](int tap, int r, bf16x8 (&a)[3]) {
            const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
            for (int t = 0; t < 3; ++t) a[t] = *reinterpret_cast<const bf16x8*>(xc + t * T3_XT + abase[dx] + (r + dy) * T3_ROWB);
        };
        // (diagnostic knob: delaying waves 4-7, the SIMD partners of waves 0-3, by up to 2,000 cycles per half-step moves
        // time between the MFMA loop and the barrier wait but does not shorten the half-step: measured null)
#ifdef XSD_DIAG
        if (hw) for (int q = 0; q < ((abl >> 8) & 31); ++q) __builtin_amdgcn_s_sleep(1);
#endif
        load_b(0, bf[0]);
        load_a(0, 0, af[0]);
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const int tap = i >> 1, r = i & 1;
            if (i + 1 < 18 && !(abl & 32)) {
                if (((i + 1) & 1) == 0) load_b((i + 1) >> 1, bf[((i + 1) >> 1) & 1]);
                load_a((i + 1) >> 1, (i + 1) & 1, af[(i + 1) & 1]);
            }
            const bf16x8 (&w)[3] = bf[tap & 1];
            const bf16x8 (&x)[3] = af[i & 1];
            if (!(abl & 8)) {
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[2], accx[r], 0, 0, 0);   // Wh * Xl
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], x[0], accx[r], 0, 0, 0);   // Wl * Xh
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[1], accx[r], 0, 0, 0);   // Wm * Xm
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[0], acc[r], 0, 0, 0);     // Wh * Xh
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[1], accx[r], 0, 0, 0);   // Wh * Xm
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[0], accx[r], 0, 0, 0);   // Wm * Xh
            }
            // staging round rr after stage 1 + 3 rr (waves 0-3) / 2 + 3 rr (waves 4-7)
            // one deferred float4 group of the previous tile's results after every second stage
            if (drip && i >= 2 && i <= 16 && (i & 1) == 0) store_pending(i / 2 - 1);
#pragma unroll
            for (int rr = 0; rr < T3_XR; ++rr)
                if (i == 1 + 3 * rr + HW) {
                    if (!(abl & 1)) store_x_round(rr, xn);
                    if (!(abl & 16)) load_x_round(rr, base2);
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

    // ---- prologue: half-step 0 into buffer 0, half-step 1 into the staging registers
    const int rs0 = P.in[0].rs, ps0 = P.in[0].ps;
    Cur cur = {0, 0, 0, 0};
    TileXY tcur = tile_of(0);
    tile_offsets(tcur, rs0, ps0);
    {
        const float* b0 = x_base(cur, tcur);
#pragma unroll
        for (int r = 0; r < T3_XR; ++r) load_x_round(r, b0);
        load_w(cur);
#pragma unroll
        for (int r = 0; r < T3_XR; ++r) store_x_round(r, smem);
        store_w();
    }
    Cur n1 = succ(cur);                 // same tile: a tile has at least two half-steps
    TileXY t1 = tcur;
    if (items > 1) {
        const float* b1 = x_base(n1, t1);
#pragma unroll
        for (int r = 0; r < T3_XR; ++r) load_x_round(r, b1);
    }
    Cur n2 = succ(n1);
    TileXY t2 = t1;
    if (items > 2 && n2.k != n1.k) { t2 = tile_of(n2.k); tile_offsets(t2, rs0, ps0); }
    lds_barrier();
    S3_TICK(0);

    bool pending = false;     // a finished tile's results are waiting in `pend`
#pragma unroll 1
    for (int it = 0; it < items; ++it) {
        const bool more1 = (it + 1 < items), more2 = (it + 2 < items);
        if (!(abl & 2)) load_w(more1 ? n1 : cur);     // unconditional (the last half-step re-reads its own panel, unused)
        const float* base2 = more2 ? x_base(n2, t2) : nullptr;
        S3_TICK(1);
        if (cur.i == 0 && cur.s2 == 0) init_acc(cur.j);
        compute(std::integral_constant<int, 0>{}, smem + (it & 1) * T3_XB, smem + ((it + 1) & 1) * T3_XB, base2, pending);
        pending = false;
        S3_TICK(2);
        if (cur.i == n_in - 1 && cur.s2 == 1 && !(abl & 4)) {
            epilogue(cur.j, tcur);
            pending = more1;
            if (!more1) {        // last half-step of the workgroup: nothing left to hide the stores behind
#pragma unroll
                for (int c = 0; c < 8; ++c) store_pending(c);
            }
        }
        S3_TICK(3);
        if (more1) {
            lds_barrier();       // every wave is done with the weight buffer and with input buffer it&1; buffer (it+1)&1 is complete
            S3_TICK(4);
            if (!(abl & 2)) store_w();
            lds_barrier();
            S3_TICK(5);
        }
        cur = n1; tcur = t1;
        n1 = n2; t1 = t2;
        n2 = succ(n2);
        if (it + 3 < items && n2.k != n1.k) { t2 = tile_of(n2.k); tile_offsets(t2, rs0, ps0); }
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
                                           hipFuncAttributeMaxDynamicSharedMemorySize, T3_LDS_BYTES);
        if (e != hipSuccess) return e;
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        done = true;
    }
    if (p.n_in < 1 || p.n_out < 1 || p.n_in * p.n_out > 5 || !p.zero) return hipErrorInvalidValue;
    const int tilesY = (p.H + T3_ROWS - 1) / T3_ROWS;
    const int ntiles = p.B * p.tilesX * tilesY;
    if (ntiles <= 0) return hipSuccess;
    const dim3 g(ntiles < ncu ? ntiles : ncu), b(T3_THREADS);
    hipLaunchKernelGGL(conv3x3_s3_kernel, g, b, T3_LDS_BYTES, stream, p);
    return hipGetLastError();
}

} // namespace xsd
