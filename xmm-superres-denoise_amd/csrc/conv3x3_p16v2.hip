// conv3x3_p16v2.hip -- math mode 2 forward/K-loop conv with ROLE-SPLIT waves (same arithmetic, LDS layout and epilogue as
// conv3x3_p16.hip; reference layers rrdb_blocks.py:27-31, generator_rrdb.py:38-44,101 and their input-gradients).
//
// Measured on conv3x3_p16.hip (tools/stamps.py + XSD_ABLATE): MFMA time (12.1 ms) and memory time (11.9 ms) of the DN
// forward simply ADD UP (24.7 ms), independent of prefetch depth; removing the ds_reads does not help, removing the MFMAs
// does.  A wave that has LDS-DMA / VMEM instructions queued in front of its MFMAs does not start them until the memory
// pipeline has accepted the requests, and at 6-7 TB/s the pipeline is always full.  So here the waves are specialised:
//   waves 0..3 (compute, one per SIMD, 4 tile rows each): ds_read + MFMA + epilogue only;
//   waves 4..7 (loaders, one per SIMD): stream the P16 input half-tiles and weight half-panels HBM/L2 -> VGPR -> LDS with
//                ordinary 16-B loads, two register sets deep (~114 KB in flight per CU), and never execute an MFMA.
// One s_barrier per half-step joins the roles: the loader publishes half-step u+1 (ds_write) while the compute waves
// run half-step u.  No LDS-DMA here: an LDS-DMA queue is only a few instructions deep per wave, plain loads are 64 deep.
#include <cstdlib>
#include "p16.h"
#include "xsd_kernels.h"

namespace xsd {
namespace v2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int QW = 8;                        // waves
constexpr int QTH = 2 * QW;                  // 16 tile rows
constexpr int QHH = QTH + 2;                 // 18 halo rows
constexpr int QPX = QHH * HALO_W;            // 612 halo pixels
constexpr int QROWB = HALO_W * 64;           // 2176 B per halo row of a half-tile
constexpr int QIN_CHUNKS = 40;               // 1-KiB chunks per half-tile buffer (612*64 = 39,168 B used; 10 per loader)
constexpr int QIN_BYTES = QIN_CHUNKS * 1024; // 40,960
constexpr int QW_CHUNKS = 18;
constexpr int QW_BYTES = QW_CHUNKS * 1024;   // 18,432 = [9 taps][hi|lo][64 lanes][8 bf16]
constexpr int Q_W0 = 3 * QIN_BYTES, Q_W1 = Q_W0 + QW_BYTES;  // three input half-tile buffers, two half-panel buffers
constexpr int Q_BIAS = Q_W1 + QW_BYTES;      // 156,672
constexpr int Q_LDS_BYTES = Q_BIAS + 5 * 32 * 4; // 157,312


constexpr int CW = 4;                         // compute waves (one per SIMD), 4 tile rows each
constexpr int RPW = QTH / CW;                 // 4 rows per compute wave
constexpr int LW = 4;                         // loader waves (one per SIMD)
constexpr int TT = (CW + LW) * 64;            // 512 threads
constexpr int NIN = (QIN_CHUNKS + LW - 1) / LW; // 10 input chunks per loader per half-step
constexpr int NWL = (QW_CHUNKS + LW - 1) / LW;  // 5 weight chunks per loader per half-step
constexpr int NSET = NIN + NWL;               // 15 x 16 B per lane per register set

// ROWREUSE: the compute waves keep the half-panel's 18 weight fragments in registers and walk the 6 input rows of their
// 4 output rows once, applying each input fragment to every output row that uses it (54 KB of LDS reads per wave and
// half-step instead of 90 KB).
// DBG: in-kernel clocks around the MFMA loop (tools/clock_inkernel.py); a template parameter because s_memtime shares
// the LDS wait counter and would turn the counted waits of the production path into full drains.
template <bool ROWREUSE, bool DBG>
__global__ __launch_bounds__(TT, 2) void conv3x3_p16v2_kernel(const ConvParams P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wv >= CW;
    const int lw = wv - CW;
    const int h = lane >> 5;
    const int l31 = lane & 31;
    const int tilesY = (P.H + QTH - 1) / QTH;
    const int ntiles = P.B * tilesY * P.tilesX;
    const int nsteps = P.n_in;
    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
    const int nhalf = my_tiles * nsteps * 2;
    if (nhalf <= 0) return;

    struct TileXY { int b, y0, x0; };
    auto tile_of = [&](int k) {
        int t = (int)blockIdx.x + k * G;
        TileXY r;
        const int tx = t % P.tilesX; t /= P.tilesX;
        r.x0 = tx * TILE_W; r.y0 = (t % tilesY) * QTH; r.b = t / tilesY;
        return r;
    };
    auto half_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    float* bias_lds = reinterpret_cast<float*>(smem + Q_BIAS);
    if (tid < 160) bias_lds[tid] = (P.bias && tid < 32) ? P.bias[tid] : 0.f;
    if (P.ablate & 256) { // diagnostic: realistic operand bits in every LDS buffer (for MFMA-only runs: zeros draw less power)
        unsigned int* w32 = reinterpret_cast<unsigned int*>(smem);
        for (int i = tid; i < Q_BIAS / 4; i += TT) {
            unsigned int x = (unsigned int)i * 2654435761u + blockIdx.x * 40503u;
            x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
            // two bf16: sign | exponent 0x7E (0.5..1) or 0x76 for "lo" words | 7 mantissa bits
            const unsigned int e = ((i >> 3) & 1) ? 0x3B00u : 0x3F00u;
            w32[i] = ((x & 0x807Fu) | e) | ((((x >> 16) & 0x807Fu) | e) << 16);
        }
        __syncthreads();
    }

    if (loader) {
        // ================================================================================================ loader role
        // slot of chunk g, lane: pixel (g*64+lane)>>2, physical 16-B piece (g*64+lane)&3; the XOR swizzle is applied to
        // the SOURCE address so the LDS image is written linearly (dst = g*1024 + lane*16).
        // Everything below is branch-free straight-line code per half-step (invalid slots read a zero page, the request
        // cursor saturates at the last half-step): with loads under conditionals hipcc falls back to s_waitcnt vmcnt(0)
        // in front of every ds_write, which drains the younger register set and halves the bytes in flight.
        const char* zero = reinterpret_cast<const char*>(P.zero) + (lane & 3) * 16;
        int goff[NIN];
        TileXY dt = tile_of(0);
        auto tile_offsets = [&](const TileXY& T, int rs, int ps) {
#pragma unroll
            for (int k = 0; k < NIN; ++k) {
                const int slot = (lw + k * LW) * 64 + lane;
                const int px = slot >> 2, pc = slot & 3;
                const int hy = px / HALO_W, hx = px - hy * HALO_W;
                const int c = pc ^ ((hx >> 2) & 3);
                const int gy = T.y0 - 1 + hy, gx = T.x0 - 1 + hx;
                const bool ok = (px < QPX) && (gy >= 0) && (gy < P.H) && (gx >= 0) && (gx < P.W) && !(P.ablate & 1);
                goff[k] = ok ? 4 * (gy * rs + gx * ps) + (c >> 1) * 64 + (c & 1) * 16 : -1;
            }
        };
        tile_offsets(dt, P.in[0].rs, P.in[0].ps);
        int ds = 0, ds2 = 0, dk = 0, dh = 0; // request cursor: step, half, tile ordinal, half-step index
        auto issue = [&](f32x4 (&set)[NSET]) {
            const PlaneIn pl = P.in[ds];
            const char* base = reinterpret_cast<const char*>(pl.p + (long long)dt.b * pl.bs) + ds2 * 32;
            const char* wsrc = reinterpret_cast<const char*>(P.wstep[ds]) + ds2 * QW_BYTES + lane * 16;
#pragma unroll
            for (int k = 0; k < NWL; ++k) {
                const int g = lw + k * LW;
                set[NIN + k] = *reinterpret_cast<const f32x4*>(wsrc + (g < QW_CHUNKS ? g : QW_CHUNKS - 1) * 1024);
            }
#pragma unroll
            for (int k = 0; k < NIN; ++k) {
                const char* src = goff[k] >= 0 ? base + goff[k] : zero;
                set[k] = *reinterpret_cast<const f32x4*>(src);
            }
            if (dh + 1 < nhalf) { // scalar bookkeeping only
                ++dh;
                if (ds2 == 0) ds2 = 1;
                else {
                    ds2 = 0;
                    if (++ds == nsteps) { ds = 0; ++dk; dt = tile_of(dk); tile_offsets(dt, P.in[0].rs, P.in[0].ps); }
                }
            }
        };
        auto publish = [&](const f32x4 (&set)[NSET], int ibuf, int wbuf) {
            char* din = smem + ibuf * QIN_BYTES + lane * 16;
            char* dw = smem + (wbuf ? Q_W1 : Q_W0) + lane * 16;
#pragma unroll
            for (int k = 0; k < NIN; ++k) *reinterpret_cast<f32x4*>(din + (lw + k * LW) * 1024) = set[k];
#pragma unroll
            for (int k = 0; k < NWL; ++k) {
                const int g = lw + k * LW;
                if (g < QW_CHUNKS) *reinterpret_cast<f32x4*>(dw + g * 1024) = set[NIN + k];
            }
        };
        f32x4 setA[NSET], setB[NSET];
        issue(setA);                       // half-step 0
        issue(setB);                       // half-step 1 (nhalf is even and >= 2)
        publish(setA, 0, 0);
        issue(setA);                       // half-step 2 (or a harmless repeat of the last one)
        half_barrier();
        int ib1 = 1; // input buffer of the half-step being published
        auto stepB = [&]() { publish(setB, ib1, 1); issue(setB); half_barrier(); ib1 = ib1 == 2 ? 0 : ib1 + 1; };
        auto stepA = [&]() { publish(setA, ib1, 0); issue(setA); half_barrier(); ib1 = ib1 == 2 ? 0 : ib1 + 1; };
#pragma unroll 1
        for (int u = 0; u < nhalf - 2; u += 2) { // during compute(u) publish u+1 (odd -> set B), during u+1 publish u+2
            stepB();
            stepA();
        }
        stepB(); // u = nhalf - 2 publishes the last half-step
        return;
    }

    // ==================================================================================================== compute role
    int abase[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const int hx = l31 + dx, c = part * 2 + h;
            abase[dx][part] = (wv * RPW) * QROWB + hx * 64 + ((c ^ ((hx >> 2) & 3)) << 4);
        }

    f32x16 acc[RPW];
    auto init_acc = [&](int j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_lds + j * 32 + 8 * q + 4 * h);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < RPW; ++r) acc[r][4 * q + t] = bv[t];
        }
    };

    // ---- one half-step: 9 taps x 4 rows x (W_hi*X_lo + W_lo*X_hi + W_hi*X_hi)
    auto compute_taps = [&](int ibuf, int wbuf) { // software pipelined one tap ahead; inputs re-read per tap
        const char* inb = smem + ibuf * QIN_BYTES;
        const char* wl = smem + (wbuf ? Q_W1 : Q_W0) + lane * 16;
        bf16x8 bfr[2][2];    // [set][part]
        bf16x8 afr[2][RPW][2]; // [set][row][part]
        auto load_tap = [&](int tap, int set) {
            const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                bfr[set][part] = *reinterpret_cast<const bf16x8*>(wl + (tap * 2 + part) * 1024);
#pragma unroll
                for (int r = 0; r < RPW; ++r)
                    afr[set][r][part] = *reinterpret_cast<const bf16x8*>(inb + abase[dx][part] + (r + dy) * QROWB);
            }
        };
        load_tap(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int set = tap & 1;
            if (tap + 1 < 9) load_tap(tap + 1, set ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[set][0], afr[set][r][1], acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[set][1], afr[set][r][0], acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[set][0], afr[set][r][0], acc[r], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const bool skip_term = (P.ablate & 16) != 0; // diagnostic: drop the W_lo * X_hi products (wrong results, 2/3 of the MFMAs)
    auto compute_rows = [&](int ibuf, int wbuf) { // weights resident in registers, every input fragment read once
        const char* inb = smem + ibuf * QIN_BYTES;
        const char* wl = smem + (wbuf ? Q_W1 : Q_W0) + lane * 16;
        bf16x8 w[9][2];      // [tap][hi|lo]
        bf16x8 x[3][2];      // ring of input fragments, prefetched two steps ahead
        // step order: the rows shared by three output rows first, so the short steps at the end run on prefetched data
        constexpr int kRow[6] = {2, 3, 1, 4, 0, 5};
        auto load_w = [&](int tap) {
#pragma unroll
            for (int part = 0; part < 2; ++part) w[tap][part] = *reinterpret_cast<const bf16x8*>(wl + (tap * 2 + part) * 1024);
        };
        auto load_x = [&](int step, int set) {
            const int ri = kRow[step / 3], dx = step % 3;
#pragma unroll
            for (int part = 0; part < 2; ++part)
                x[set][part] = *reinterpret_cast<const bf16x8*>(inb + abase[dx][part] + ri * QROWB);
        };
        // step 0 is (row 2, dx 0) and needs taps (dy,0).  At most 15 LDS reads can be outstanding behind a counted wait
        // (lgkmcnt is 4 bits), so the rest of the panel is requested in the order of first use behind steps 0 and 1.
        load_x(0, 0); load_w(0); load_w(3); load_w(6);
        __builtin_amdgcn_sched_barrier(0);
        load_x(1, 1); load_w(1); load_w(4);
#pragma unroll
        for (int step = 0; step < 18; ++step) {
            const int set = step % 3;
            const int ri = kRow[step / 3], dx = step % 3;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int r = ri - dy;
                    if (r < 0 || r >= RPW) continue;
                    if (term == 1 && skip_term) continue;
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[dy * 3 + dx][term == 1 ? 1 : 0], x[set][term == 0 ? 1 : 0],
                                                                     acc[r], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (step == 0) { load_w(7); load_x(2, 2); load_w(2); }
            if (step == 1) { load_w(5); load_w(8); }
            if (step + 3 < 18) load_x(step + 3, set);
        }
    };
    auto compute = [&](int ibuf, int wbuf) {
        if (ROWREUSE) compute_rows(ibuf, wbuf);
        else compute_taps(ibuf, wbuf);
    };

    // ---- epilogue over P16 planes
    auto load16 = [&](const float* plane, long long px_floats, float (&v)[16]) {
        const char* px = reinterpret_cast<const char*>(plane + px_floats) + h * 32;
        const u32x4 h0 = *reinterpret_cast<const u32x4*>(px), h1 = *reinterpret_cast<const u32x4*>(px + 16);
        const u32x4 l0 = *reinterpret_cast<const u32x4*>(px + 64), l1 = *reinterpret_cast<const u32x4*>(px + 80);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            p16_join2(h0[w], l0[w], v[2 * w], v[2 * w + 1]);
            p16_join2(h1[w], l1[w], v[8 + 2 * w], v[8 + 2 * w + 1]);
        }
    };
    auto epilogue = [&](int j, const TileXY& T) {
        const OutDesc o = P.out[j];
        float* dst = o.p + (long long)T.b * o.bs;
        const long long sb = (long long)T.b * P.std_bs;
        const int x = T.x0 + l31;
        if (x >= P.W || (P.ablate & 4)) return;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int y = T.y0 + wv * RPW + r;
            if (y >= P.H) continue;
            const long long od = (long long)y * o.rs + (long long)x * o.ps;
            const long long os = sb + (long long)y * P.std_rs + x * 32;
            float v[16], e[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = acc[r][i] * o.a1;
            if (o.accumulate) {
                load16(dst, od, e);
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] += e[i];
            }
            if (o.e1) {
                load16(o.e1, os, e);
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] += o.s1 * e[i];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] *= o.a2;
            if (o.e2) {
                load16(o.e2, os, e);
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] += o.s2 * e[i];
            }
            if (o.e3) {
                load16(o.e3, os, e);
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] += o.s3 * e[i];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * o.slope;
            if (o.mask) {
                load16(o.mask, os, e);
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = e[i] > 0.f ? v[i] : v[i] * o.mslope;
            }
            unsigned int hw[8], lw[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) p16_split2(v[2 * w], v[2 * w + 1], hw[w], lw[w]);
            const u32x4 h0 = {hw[0], hw[1], hw[2], hw[3]}, h1 = {hw[4], hw[5], hw[6], hw[7]};
            const u32x4 l0 = {lw[0], lw[1], lw[2], lw[3]}, l1 = {lw[4], lw[5], lw[6], lw[7]};
            char* px = reinterpret_cast<char*>(dst + od) + h * 32;
            if (P.ablate & 64) { // diagnostic: everything but the global stores (keep the values alive)
                asm volatile("" ::"v"(h0), "v"(h1), "v"(l0), "v"(l1));
                continue;
            }
            if (P.ablate & 512) { // diagnostic: the same store instructions into a 64 KiB window per workgroup (stays in L2)
                char* rowb = reinterpret_cast<char*>(o.p) + (long long)blockIdx.x * 65536 + (wv * RPW + r) * 4096 + lane * 16;
                *reinterpret_cast<u32x4*>(rowb) = h0;
                *reinterpret_cast<u32x4*>(rowb + 1024) = h1;
                *reinterpret_cast<u32x4*>(rowb + 2048) = l0;
                *reinterpret_cast<u32x4*>(rowb + 3072) = l1;
                continue;
            }
            if (P.ablate & 128) { // diagnostic: the same bytes of the same row, permuted so every store instruction is 1 KiB contiguous
                char* rowb = reinterpret_cast<char*>(dst + (long long)y * o.rs + (long long)T.x0 * o.ps) + lane * 16;
                *reinterpret_cast<u32x4*>(rowb) = h0;
                *reinterpret_cast<u32x4*>(rowb + 1024) = h1;
                *reinterpret_cast<u32x4*>(rowb + 2048) = l0;
                *reinterpret_cast<u32x4*>(rowb + 3072) = l1;
                continue;
            }
            *reinterpret_cast<u32x4*>(px) = h0;
            *reinterpret_cast<u32x4*>(px + 16) = h1;
            *reinterpret_cast<u32x4*>(px + 64) = l0;
            *reinterpret_cast<u32x4*>(px + 80) = l1;
        }
    };


    // Nested tile / step / half loops (not one flat half-step loop): the accumulators then live in fixed registers across a
    // whole tile instead of being copied through the loop-carried merge every half-step.  One barrier per half-step, in
    // lock step with the loader role.
    int ib = 0, u = 0;
    half_barrier(); // half-step 0 published
    unsigned long long cyc_mfma = 0, wall_mfma = 0;
    const unsigned long long cyc0 = DBG ? __builtin_readcyclecounter() : 0, wall0 = DBG ? wall_clock64() : 0;
#pragma unroll 1
    for (int k = 0; k < my_tiles; ++k) {
        const TileXY cur = tile_of(k);
        init_acc(0);
#pragma unroll 1
        for (int s = 0; s < nsteps; ++s) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const unsigned long long c0 = DBG ? __builtin_readcyclecounter() : 0, w0 = DBG ? wall_clock64() : 0;
                if (!(P.ablate & 8)) compute(ib, s2);
                else if (P.ablate & 32) { __builtin_amdgcn_s_sleep(32); __builtin_amdgcn_s_sleep(32); } // ~4096 idle cycles instead of the MFMA loop
                if (DBG) { cyc_mfma += __builtin_readcyclecounter() - c0; wall_mfma += wall_clock64() - w0; }
                if (s2 == 1 && s == nsteps - 1) epilogue(0, cur);
                if (++u < nhalf) half_barrier();
                ib = ib == 2 ? 0 : ib + 1;
            }
        }
    }
    if (DBG && P.dbg && tid == 0) {
        atomicAdd(&P.dbg[0], cyc_mfma); atomicAdd(&P.dbg[1], wall_mfma);
        atomicAdd(&P.dbg[2], (unsigned long long)(__builtin_readcyclecounter() - cyc0));
        atomicAdd(&P.dbg[3], (unsigned long long)(wall_clock64() - wall0));
        atomicAdd(&P.dbg[6], (unsigned long long)nhalf);
    }
}

} // namespace v2

hipError_t launch_conv3x3_p16v2(const ConvParams& p, int rowreuse, hipStream_t stream)
{
    static bool done = false;
    static int ncu = 256;
    typedef void (*kern_t)(const ConvParams);
    static const kern_t kerns[2][2] = {{v2::conv3x3_p16v2_kernel<false, false>, v2::conv3x3_p16v2_kernel<false, true>},
                                       {v2::conv3x3_p16v2_kernel<true, false>, v2::conv3x3_p16v2_kernel<true, true>}};
    if (!done) {
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kerns[a][b]),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, v2::Q_LDS_BYTES);
                if (e != hipSuccess) return e;
            }
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        done = true;
    }
    const int tilesY = (p.H + v2::QTH - 1) / v2::QTH;
    const int ntiles = p.B * p.tilesX * tilesY;
    if (ntiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(kerns[rowreuse ? 1 : 0][p.dbg ? 1 : 0], dim3(ntiles < ncu ? ntiles : ncu), dim3(v2::TT), v2::Q_LDS_BYTES, stream, p);
    return hipGetLastError();
}

} // namespace xsd
