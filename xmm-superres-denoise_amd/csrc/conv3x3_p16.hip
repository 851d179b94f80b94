// conv3x3_p16.hip -- math mode 2: the 3x3 conv (forward + input-gradient) over P16 planes (p16.h), bf16x3 MFMA.
// Reference layers: rrdb_blocks.py:27-31, generator_rrdb.py:38-44,95,101 and their autograd input-gradients.
//
// Every byte a workgroup needs arrives by LDS-DMA (global_load_lds_dwordx4): the input tile is already stored as the
// MFMA operand (hi|lo bf16 in accumulator channel order) and the weights are pre-split by pack_weights_p16_kernel, so
// the staging path executes no VALU, no ds_write and holds no VGPRs.
//   * workgroup = 512 threads (8 waves, 2/SIMD), tile = 16 x 32 pixels, wave w owns rows 2w, 2w+1;
//   * a plane step is cut into two 16-channel HALF-steps (one MFMA k-step each): per half-step the LDS holds a
//     18x34-pixel x 64 B input half-tile (ring of 3) and an 18 KB half-panel (ring of 2), 157,312 B total; input
//     half-tiles are fetched two half-steps ahead (about 100 KB in flight per CU, what HBM latency x 25 GB/s/CU needs),
//     with ONE barrier per half-step behind a counted s_waitcnt vmcnt(N) that leaves the younger DMAs and the
//     epilogue's stores in flight;
//   * in the one-input/many-output mode both halves of the tile stay resident and only the half-panels stream;
//   * epilogue: each lane owns one pixel and the 16 channels of its accumulator half = 32 contiguous bytes of hi and of
//     lo in the P16 pixel -> four 16-B stores; residual / accumulate / lrelu'-mask operands are read the same way.
#include <cstdlib>
#include "p16.h"
#include "xsd_kernels.h"

namespace xsd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int QW = 8;                        // waves
constexpr int QT = QW * 64;                  // 512 threads
constexpr int QTH = 2 * QW;                  // 16 tile rows
constexpr int QHH = QTH + 2;                 // 18 halo rows
constexpr int QPX = QHH * HALO_W;            // 612 halo pixels
constexpr int QROWB = HALO_W * 64;           // 2176 B per halo row of a half-tile
constexpr int QIN_CHUNKS = (QPX * 4 + 63) / 64; // 39 DMA chunks (1 KiB) per half-tile
constexpr int QIN_BYTES = QIN_CHUNKS * 1024; // 39,936 (612*64 = 39,168 rounded up to whole chunks)
constexpr int QW_CHUNKS = 18;
constexpr int QW_BYTES = QW_CHUNKS * 1024;   // 18,432 = [9 taps][hi|lo][64 lanes][8 bf16]
constexpr int Q_W0 = 3 * QIN_BYTES, Q_W1 = Q_W0 + QW_BYTES;  // three input half-tile buffers, two half-panel buffers
constexpr int Q_BIAS = Q_W1 + QW_BYTES;      // 156,672
constexpr int Q_LDS_BYTES = Q_BIAS + 5 * 32 * 4; // 157,312
constexpr int QNG = (QIN_CHUNKS + QW - 1) / QW; // 5 in-tile chunks per wave

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <bool MULTI_OUT>
__global__ __launch_bounds__(QT, 2) void conv3x3_p16_kernel(const ConvParams P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef XSD_DIAG   // diagnostic library variant only (make diag; selected with XSD_LIB): ablation knobs are compiled out otherwise
    const int abl = P.ablate;
#else
    constexpr int abl = 0;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int l31 = lane & 31;

    const int tilesY = (P.H + QTH - 1) / QTH;
    const int ntiles = P.B * tilesY * P.tilesX;
    const int nsteps = MULTI_OUT ? P.n_out : P.n_in;
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

    // ---- per-lane LDS read bases: [dx][part]; lane = (pixel column l31, k-half h) reads chunk part*2 + h
    int abase[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const int hx = l31 + dx, c = part * 2 + h;
            abase[dx][part] = (wv * 2) * QROWB + hx * 64 + ((c ^ ((hx >> 2) & 3)) << 4);
        }

    // ---- LDS-DMA descriptors of this wave's input chunks for the tile being fetched: byte offset inside one image of
    // the plane (without the k-half offset), or -1 for zero padding / rows beyond the image / chunk padding.
    int goff[QNG];
    const char* zero = reinterpret_cast<const char*>(P.zero) + (lane & 3) * 16;
    auto tile_offsets = [&](const TileXY& T, int rs, int ps) {
#pragma unroll
        for (int k = 0; k < QNG; ++k) {
            const int slot = (wv + k * QW) * 64 + lane;
            const int px = slot >> 2, pc = slot & 3;
            const int hy = px / HALO_W, hx = px - hy * HALO_W;
            const int c = pc ^ ((hx >> 2) & 3);
            const int gy = T.y0 - 1 + hy, gx = T.x0 - 1 + hx;
            const bool ok = (px < QPX) && (gy >= 0) && (gy < P.H) && (gx >= 0) && (gx < P.W);
            goff[k] = ok ? 4 * (gy * rs + gx * ps) + (c >> 1) * 64 + (c & 1) * 16 : -1;
        }
    };
    auto dma_in = [&](int s, int s2, const TileXY& T, int buf) {
        const PlaneIn pl = P.in[s];
        const char* base = reinterpret_cast<const char*>(pl.p + (long long)T.b * pl.bs) + s2 * 32;
        char* dst = smem + buf * QIN_BYTES;
        if (abl & 1) return;
#pragma unroll
        for (int k = 0; k < QNG; ++k) {
            const int g = wv + k * QW;
            if (g < QIN_CHUNKS) {
                const char* src = goff[k] >= 0 ? base + goff[k] : zero;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + g * 1024), 16, 0, 0);
            }
        }
    };
    auto dma_w = [&](int s, int s2, int buf) {
        const char* src = reinterpret_cast<const char*>(P.wstep[s]) + s2 * QW_BYTES + lane * 16;
        char* dst = smem + (buf ? Q_W1 : Q_W0);
        if (abl & 2) return;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int g = wv + k * QW;
            if (g < QW_CHUNKS)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + g * 1024), (lptr_t)(dst + g * 1024), 16, 0, 0);
        }
    };

    float* bias_lds = reinterpret_cast<float*>(smem + Q_BIAS);
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

    // ---- one half-step: 9 taps x 2 rows x (W_hi*X_lo + W_lo*X_hi + W_hi*X_hi), software pipelined one tap ahead
    auto compute = [&](int ibuf, int wbuf) {
        const char* inb = smem + ibuf * QIN_BYTES;
        const char* wl = smem + (wbuf ? Q_W1 : Q_W0) + lane * 16;
        bf16x8 bfr[2][2];    // [set][part]
        bf16x8 afr[2][2][2]; // [set][row][part]
        auto load_tap = [&](int tap, int set) {
            const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                bfr[set][part] = *reinterpret_cast<const bf16x8*>(wl + (tap * 2 + part) * 1024);
#pragma unroll
                for (int r = 0; r < 2; ++r)
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
            for (int r = 0; r < 2; ++r) {
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[set][0], afr[set][r][1], acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[set][1], afr[set][r][0], acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[set][0], afr[set][r][0], acc[r], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Row-reuse form of the same half-step: the half-panel's 18 weight fragments stay in registers and the 4 input rows of
    // the wave's 2 output rows are walked once, each input fragment feeding every output row that uses it (42 instead of
    // 54 LDS fragment reads per half-step; same 54 MFMAs, same summation order per accumulator is NOT required: fp32 sums
    // of the three bf16 products commute up to rounding, covered by the parity tolerances).
    auto compute_rows = [&](int ibuf, int wbuf) {
        const char* inb = smem + ibuf * QIN_BYTES;
        const char* wl = smem + (wbuf ? Q_W1 : Q_W0) + lane * 16;
        bf16x8 w[9][2];
        bf16x8 x[3][2];
        constexpr int kRow[4] = {1, 2, 0, 3};   // rows shared by both output rows first
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
        // step 0 = (input row 1, dx 0) feeds output row 0 with tap (1,0) and output row 1 with tap (0,0)
        load_x(0, 0); load_w(0); load_w(3);
        __builtin_amdgcn_sched_barrier(0);
        load_x(1, 1); load_w(1); load_w(4); load_w(6);
#pragma unroll
        for (int step = 0; step < 12; ++step) {
            const int set = step % 3;
            const int ri = kRow[step / 3], dx = step % 3;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int r = ri - dy;
                    if (r < 0 || r >= 2) continue;
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[dy * 3 + dx][term == 1 ? 1 : 0], x[set][term == 0 ? 1 : 0],
                                                                     acc[r], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (step == 0) { load_w(7); load_x(2, 2); load_w(2); }
            if (step == 1) { load_w(5); load_w(8); }
            if (step + 3 < 12) load_x(step + 3, set);
        }
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
        if (x >= P.W || (abl & 4)) return;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int y = T.y0 + wv * 2 + r;
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
            if (o.bits_in) { // lrelu' from the compact mask written by the forward conv (2 B per lane instead of 64 B)
                const unsigned int m = o.bits_in[(os >> 5) * 2 + h];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = ((m >> i) & 1u) ? v[i] : v[i] * o.mslope;
            } else if (o.mask) {
                load16(o.mask, os, e);
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = e[i] > 0.f ? v[i] : v[i] * o.mslope;
            }
            if (o.bits_out) {
                unsigned int m = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) m |= (v[i] > 0.f ? 1u : 0u) << i;
                o.bits_out[(os >> 5) * 2 + h] = (unsigned short)m;
            }
            unsigned int hw[8], lw[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) p16_split2(v[2 * w], v[2 * w + 1], hw[w], lw[w]);
            const u32x4 h0 = {hw[0], hw[1], hw[2], hw[3]}, h1 = {hw[4], hw[5], hw[6], hw[7]};
            const u32x4 l0 = {lw[0], lw[1], lw[2], lw[3]}, l1 = {lw[4], lw[5], lw[6], lw[7]};
            char* px = reinterpret_cast<char*>(dst + od) + h * 32;
            *reinterpret_cast<u32x4*>(px) = h0;
            *reinterpret_cast<u32x4*>(px + 16) = h1;
            *reinterpret_cast<u32x4*>(px + 64) = l0;
            *reinterpret_cast<u32x4*>(px + 80) = l1;
        }
    };

    // Wait for this half-step's DMA contract, then barrier.  VMEM ops retire in order, so "everything except the
    // youngest `young` ops" = the DMAs that the NEXT half-step reads have landed, while the DMAs issued for the half-step
    // after it (input ring depth 3) and the epilogue's stores stay in flight.
    auto wait_and_barrier = [&](int young) {
        switch (young) {
#define XSD_VMCNT_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
        XSD_VMCNT_CASE(4) XSD_VMCNT_CASE(5) XSD_VMCNT_CASE(8) XSD_VMCNT_CASE(9) XSD_VMCNT_CASE(10) XSD_VMCNT_CASE(12)
        XSD_VMCNT_CASE(13) XSD_VMCNT_CASE(14) XSD_VMCNT_CASE(15)
#undef XSD_VMCNT_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    const int n_in_chunks = (wv + (QNG - 1) * QW < QIN_CHUNKS) ? QNG : QNG - 1; // input DMA instructions per wave

    // ---- prologue
    TileXY cur = tile_of(0);
    tile_offsets(cur, P.in[0].rs, P.in[0].ps);
    int s = 0, k = 0; // compute cursor: step inside the tile, tile ordinal

    if constexpr (MULTI_OUT) {
        dma_in(0, 0, cur, 0);
        dma_in(0, 1, cur, 1);
        dma_w(0, 0, 0);
        __syncthreads();
#pragma unroll 1
        for (int u = 0; u < nhalf; ++u) {
            const int s2 = u & 1;
            const bool more = (u + 1 < nhalf);
            const int s_n = s2 ? ((s + 1 == nsteps) ? 0 : s + 1) : s;
            const bool new_tile = more && s2 && (s + 1 == nsteps);
            if (more) dma_w(s_n, s2 ^ 1, s2 ^ 1);
            if (s2 == 0) init_acc(s);
            if (!(abl & 8)) compute(s2, s2);
            int nst = 0;
            if (s2 == 1) {
                epilogue(s, cur);
                const int y = cur.y0 + wv * 2;
                nst = (abl & 4) ? 0 : 4 * ((y < P.H) + (y + 1 < P.H));
            }
            if (more) {
                wait_and_barrier(nst);
                if (new_tile) { // both halves of the tile are resident in this mode: refill them now
                    cur = tile_of(++k);
                    tile_offsets(cur, P.in[0].rs, P.in[0].ps);
                    dma_in(0, 0, cur, 0);
                    dma_in(0, 1, cur, 1);
                    __syncthreads();
                }
            }
            if (s2) s = s_n;
        }
    } else {
        // DMA cursor runs two half-steps ahead of the compute cursor (input ring of 3 buffers)
        int ds = 0, ds2 = 0, dk = 0;
        TileXY dt = cur;
        auto dma_next_in = [&](int buf) {
            dma_in(ds, ds2, dt, buf);
            if (ds2 == 0) ds2 = 1;
            else {
                ds2 = 0;
                if (++ds == nsteps) { ds = 0; ++dk; if (dk < my_tiles) { dt = tile_of(dk); tile_offsets(dt, P.in[0].rs, P.in[0].ps); } }
            }
        };
        dma_w(0, 0, 0);
        dma_next_in(0);
        if (nhalf > 1) dma_next_in(1);
        __syncthreads();
        int ib = 0, db = 2;
#pragma unroll 1
        for (int u = 0; u < nhalf; ++u) {
            const int s2 = u & 1;
            const bool more = (u + 1 < nhalf);
            const int s_n = s2 ? ((s + 1 == nsteps) ? 0 : s + 1) : s;
            int young = 0;
            if (more) dma_w(s_n, s2 ^ 1, s2 ^ 1);              // read by half-step u+1: issued FIRST
            if (u + 2 < nhalf) { dma_next_in(db); young = (abl & 1) ? 0 : n_in_chunks; db = db == 2 ? 0 : db + 1; } // read by u+2
            if (s2 == 0 && s == 0) init_acc(0);
            if (!(abl & 8)) { if (abl & 16384) compute(ib, s2); else compute_rows(ib, s2); }
            if (s2 == 1 && s == nsteps - 1) {
                epilogue(0, cur);
                const int y = cur.y0 + wv * 2;
                young += (abl & 4) ? 0 : (4 + (P.out[0].bits_out != nullptr)) * ((y < P.H) + (y + 1 < P.H));
            }
            if (more) wait_and_barrier(young);
            ib = ib == 2 ? 0 : ib + 1;
            if (s2) {
                if (s + 1 == nsteps && more) cur = tile_of(++k);
                s = s_n;
            }
        }
    }
}

hipError_t launch_conv3x3_p16(const ConvParams& p, hipStream_t stream)
{
    static bool done = false;
    static int ncu = 256;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_p16_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Q_LDS_BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_p16_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, Q_LDS_BYTES);
        if (e != hipSuccess) return e;
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        done = true;
    }
    if (!p.zero) return hipErrorInvalidValue;
    const int tilesY = (p.H + QTH - 1) / QTH;
    const int ntiles = p.B * p.tilesX * tilesY;
    if (ntiles <= 0) return hipSuccess;
    const dim3 g(ntiles < ncu ? ntiles : ncu), b(QT);
    if (p.n_out > 1) hipLaunchKernelGGL((conv3x3_p16_kernel<true>), g, b, Q_LDS_BYTES, stream, p);
    else hipLaunchKernelGGL((conv3x3_p16_kernel<false>), g, b, Q_LDS_BYTES, stream, p);
    return hipGetLastError();
}

} // namespace xsd
