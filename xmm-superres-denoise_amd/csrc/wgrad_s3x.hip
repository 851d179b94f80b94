// wgrad_s3x.hip -- math mode 3 ("bf16x6"): weight gradient of the 3x3 convs over fp32 planes, ROLE-SPLIT workgroup
// (4 staging waves + 4 MFMA waves, one of each per SIMD, one workgroup per CU).  Arithmetic: the exact 3-term bf16 split
// of X and G (xsd_split.h), six products per multiply.  LDS images: per tile the X halo (6 x 34 pixels) and the G tile
// (4 x 32), each as three term images [pixel][32 x bf16]; the GEMM's K index is the pixel, so fragments are read with the
// transposing ds_read_b64_tr_b16 (pixels across lanes, four channels per lane).  Grid: 1-D, decoded so that the workgroups
// which read the same G tiles share an XCD (below).  Partial sums per workgroup, then a fixed-order two-stage reduction
// (wgrad_reduce_kernel, double accumulation): bitwise reproducible, no atomics.  Replaces autograd's conv
// weight-gradient for the reference's nn.Conv2d(32k -> 32n, 3,1,1) layers (rrdb_blocks.py:27-31; generator_rrdb.py:38-44,95,101):
//     dW[co][ci][tap] = sum_{b,y,x} G[b,y,x,co] * X[b,y+dy-1,x+dx-1,ci],   db[co] = sum G[..,co]
//
// Why roles.  In the unified kernel of rounds 1-2 (git history) a wave stages (11 global loads, 11 splits, 33 LDS writes per 4-row tile) and multiplies (108
// MFMAs) in turn; the stamps show it blocked ~3k cycles per tile just ISSUING its loads (8 waves x 11 KB in flight per CU
// fill the vector-memory queue) and ~2.3k converting, against 3.5k of matrix time, and the co-resident second workgroup
// only partly fills the holes.  Here
//   * waves 0..3 stage: tile t+2 is in flight in registers (buffer loads, hand-counted waits: conv3x3_s3x.hip explains
//     the scheme and tools/check_async_loads.py verifies it on the generated code) while tile t+1 is split and written
//     to the other of two LDS buffers;
//   * waves 4..7 multiply tile t: wave 4+r owns tile row r, nine 32x32 accumulators (one per tap) held across all tiles of
//     the workgroup; their stream is ds_read_b64_tr_b16 + v_mfma only.
// One barrier per tile.  Two waves per SIMD -> 256 registers per wave.
#include "xsd_kernels.h"
#include "xsd_split.h"

namespace xsd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#ifndef V3S_MPRIO
#define V3S_MPRIO 1    // s_setprio of the MFMA waves (0: none)
#endif
#ifndef V3S_DEPTH
#define V3S_DEPTH 1    // X fragments requested this many (dx, halo row) steps ahead
#endif
#ifndef V3_PAIR
#define V3_PAIR 1      // MFMA waves own (pixel half, row pair) instead of a tile row (round 5; -DV3_PAIR=0: the round-4 walk, for A/Bs)
#endif
constexpr int V3_TH = 4;                                              // tile rows = MFMA waves
constexpr int V3_LT = 256;                                            // staging threads (waves 0..3)
constexpr int V3_THREADS = V3_LT + 64 * V3_TH;                        // 512
constexpr int V3_HPX = (V3_TH + 2) * HALO_W;                          // 204 halo pixels
constexpr int V3_X_SLOTS = V3_HPX * 8;                                // 1632 (pixel, channel quad) slots
constexpr int V3_X_ROUNDS = (V3_X_SLOTS + V3_LT - 1) / V3_LT;         // 7
constexpr int V3_G_SLOTS = V3_TH * TILE_W * 8;                        // 1024
constexpr int V3_G_ROUNDS = V3_G_SLOTS / V3_LT;                       // 4
constexpr int V3_NL = V3_X_ROUNDS + V3_G_ROUNDS;                      // 11 loads per tile and staging thread
constexpr int V3_XT = V3_HPX * 64;                                    // 13,056 B per X term image
constexpr int V3_GT = V3_TH * TILE_W * 64;                            // 8,192 B per G term image
constexpr int V3_G_OFF = 3 * V3_XT;                                   // 39,168
constexpr int V3_BUF = V3_G_OFF + 3 * V3_GT;                          // 63,744 B per buffer
constexpr int V3_SINK = 2 * V3_BUF;                                   // writes of exhausted slots land behind the buffers
constexpr int V3_LDS_BYTES = V3_SINK + 2 * V3_XT + 2048;              // 155,648 (the sink takes the same term offsets)
static_assert(V3_LDS_BYTES <= 160 * 1024, "LDS");
static_assert(V3_TH * 4096 <= V3_BUF, "reduction slabs");

// 8 consecutive pixels (k = 8h + 0..7) of this lane's channel from a [pixel][32 x bf16] image
__device__ __forceinline__ bf16x8 v3_tr_frag(const char* lds_lane_base, int byte_off)
{
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds_lane_base + byte_off));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds_lane_base + byte_off + 4 * 64));
    s16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return __builtin_bit_cast(bf16x8, r);
}

__global__ __launch_bounds__(V3_THREADS) void wgrad_s3x_kernel(const WgradParams P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wid < 4;       // wave-uniform role
    const int h = lane >> 5;
    const int l31 = lane & 31;

    // 1-D grid decode: the n_in workgroups that read the SAME G tiles have linear ids 8 apart -> one XCD
    const int lin = blockIdx.x;
    const int xcd = lin & 7, qq = lin >> 3;
    const int parts8 = P.nparts >> 3;
    int j, n, part, slot = 0;
    if (P.npairs > 0) {                               // pair-list launch (xsd_kernels.h): slot = pair, all slots of a part side by side on one XCD
        slot = qq % P.npairs;
        part = (qq / P.npairs) * 8 + xcd;
        // nparts = 8 m + 1: the last part has no XCD of its own -- its slots are dealt over the CUs the m parts per XCD leave
        // (slot = 8 e + xcd for the e-th spare workgroup of an XCD); it shares nothing through L2, but no CU idles
        const int full = parts8 * P.npairs;
        if (qq >= full) { slot = (qq - full) * 8 + xcd; part = P.nparts - 1; if (slot >= P.npairs) return; }
        j = (int)((P.pair_j >> (4 * slot)) & 15);
        n = (int)((P.pair_n >> (4 * slot)) & 15);
    } else {
        j = qq % P.n_in;                              // input plane
        const int rest = qq / P.n_in;
        part = (rest % parts8) * 8 + xcd;
        n = rest / parts8;                            // G chunk
    }
    const int tilesY = (P.H + V3_TH - 1) / V3_TH;
    const int ntiles = P.B * tilesY * P.tilesX;
    const int my_tiles = part < ntiles ? (ntiles - part + P.nparts - 1) / P.nparts : 0;

    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

#ifdef XSD_DIAG   // phase stamps (diagnostic library only; tools/stamps_train.py): slot 16 staging rounds, 17 MFMA walk, 18 MFMA
                  // wave at the barrier, 19 staging wave at the barrier, 21 tiles (slots 0-15 belong to the conv kernels)
    unsigned long long st[2] = {0, 0};
    unsigned long long t0 = __builtin_readcyclecounter();
    const bool stamp = P.dbg != nullptr;
#define V3_TICK(i) do { if (stamp) { const unsigned long long t_ = __builtin_readcyclecounter(); st[i] += t_ - t0; t0 = t_; } } while (0)
#else
#define V3_TICK(i) do { } while (0)
#endif
#ifdef XSD_DIAG   // staging-wave stamps: branch-free (wgrad_h2x.hip)
#define V3_LTICK(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); st[i] += t_ - t0; t0 = t_; } while (0)
#else
#define V3_LTICK(i) do { } while (0)
#endif
    f32x16 acc[9];
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f}; // staging thread: its 4 channels (lt & 7) of the G tiles it stages

    if (loader) {
        // ============================ staging waves ============================
        const int lt = tid;
        const PlaneIn xp = P.x[j];
        const PlaneIn gp = P.g[n];
        constexpr int OOR = (int)0x80000000;          // lane offset that fails every range check -> the load returns 0
        // per-slot constants: X round r -> halo pixel (hy, hx), channel quad c; byte offset relative to the tile origin
        int xrel[V3_X_ROUNDS], xhx[V3_X_ROUNDS], grel[V3_G_ROUNDS], ggx[V3_G_ROUNDS];
#pragma unroll
        for (int r = 0; r < V3_X_ROUNDS; ++r) {
            const int slot = r * V3_LT + lt;
            const int p = slot >> 3, c = slot & 7;
            const int hy = p / HALO_W, hx = p - hy * HALO_W;
            xrel[r] = slot < V3_X_SLOTS ? (hy * xp.rs + hx * xp.ps + c * 4) * 4 : OOR;
            xhx[r] = slot < V3_X_SLOTS ? hx : 0x40000000;
        }
#pragma unroll
        for (int r = 0; r < V3_G_ROUNDS; ++r) {
            const int slot = r * V3_LT + lt;
            const int p = slot >> 3, c = slot & 7;
            grel[r] = ((p >> 5) * gp.rs + (p & 31) * gp.ps + c * 4) * 4;
            ggx[r] = p & 31;
        }
        const int lds0 = lt * 8;
        const bool live6 = 6 * V3_LT + lt < V3_X_SLOTS;          // last X round: 96 live threads, the others write a sink
        static_assert(V3_X_ROUNDS == 7, "sink round");

        f32x4 px[V3_X_ROUNDS] = {};
        f32x4 pg[V3_G_ROUNDS] = {};
        auto make_rsrc = [&](unsigned long long base, unsigned int bytes) {
            i32x4 d;
            d[0] = (int)(unsigned int)base; d[1] = (int)(unsigned int)((base >> 32) & 0xffffu);   // stride 0: raw buffer
            d[2] = (int)bytes; d[3] = 0x00020000;
            return d;
        };
        const unsigned int x_bytes = (unsigned int)P.H * (unsigned int)xp.rs * 4u;
        const unsigned int g_bytes = (unsigned int)P.H * (unsigned int)gp.rs * 4u;
        auto asm_load4 = [&](f32x4& dst, int off, const i32x4& rs) {
            asm volatile("buffer_load_dwordx4 %[d], %[o], %[r], 0 offen" : [d] "+v"(dst) : [o] "v"(off), [r] "s"(rs) : "memory");
        };
        auto asm_wait = [&](f32x4& v) { asm volatile("s_waitcnt vmcnt(10)" : "+v"(v) :: "memory"); };
        static_assert(V3_NL == 11, "the counted wait is vmcnt(V3_NL - 1)");

        struct TileAt { i32x4 xrs, grs; int xorg, gorg, x0; };
        auto tile_at = [&](int k) {        // descriptors and origin offsets of this workgroup's k-th tile (empty past the end)
            const bool live = k < my_tiles;
            const int t = part + k * P.nparts;
            const int tx = t % P.tilesX;
            const int t2 = t / P.tilesX;
            const int ty = t2 % tilesY;
            const int b = live ? t2 / tilesY : 0;
            TileAt a;
            a.x0 = tx * TILE_W;
            const int y0 = ty * V3_TH;
            a.xrs = make_rsrc(reinterpret_cast<unsigned long long>(xp.p + (long long)b * xp.bs), live ? x_bytes : 0u);
            a.grs = make_rsrc(reinterpret_cast<unsigned long long>(gp.p + (long long)b * gp.bs), live ? g_bytes : 0u);
            a.xorg = ((y0 - 1) * xp.rs + (a.x0 - 1) * xp.ps) * 4;    // rows above / below the image fall outside [0, bytes): zeros
            a.gorg = (y0 * gp.rs + a.x0 * gp.ps) * 4;
            return a;
        };
        // columns left / right of the image would alias the neighbouring row: those lanes get the failing offset
        auto x_off = [&](int r, const TileAt& a) { return ((unsigned)(a.x0 - 1 + xhx[r]) < (unsigned)P.W) ? a.xorg + xrel[r] : OOR; };
        auto g_off = [&](int r, const TileAt& a) { return (a.x0 + ggx[r] < P.W) ? a.gorg + grel[r] : OOR; };
#if defined(XSD_DIAG) && defined(XSD_ABL)   // timing experiments: a COMPILE-TIME constant (-DXSD_DIAG -DXSD_ABL=n builds; a run-time
                                                  // value puts the hand-counted loads and waits under branches hipcc cannot keep exact)
        constexpr int abl = XSD_ABL;     // 1: no split, 2: no LDS writes (results are garbage: timing experiments only)
#else
        constexpr int abl = 0;
#endif
        auto store_x = [&](int r, int buf) {
            u32x2 hi, mid, lo;
            if (abl & 1) { hi[0] = __float_as_uint(px[r][0]); hi[1] = __float_as_uint(px[r][1]); mid = hi; lo[0] = __float_as_uint(px[r][2]); lo[1] = __float_as_uint(px[r][3]); }
            else split3_f32x4(px[r], hi, mid, lo);
            char* d = smem + ((r == 6 && !live6) ? V3_SINK + (lt & 63) * 8 : buf + lds0 + r * (V3_LT * 8));
            if (abl & 2) { asm volatile("" :: "v"(hi), "v"(mid), "v"(lo), "v"(d)); return; }   // diag: no LDS writes
            *reinterpret_cast<u32x2*>(d) = hi;
            *reinterpret_cast<u32x2*>(d + V3_XT) = mid;
            *reinterpret_cast<u32x2*>(d + 2 * V3_XT) = lo;
        };
        auto store_g = [&](int r, int buf) {
            u32x2 hi, mid, lo;
            if (abl & 1) { hi[0] = __float_as_uint(pg[r][0]); hi[1] = __float_as_uint(pg[r][1]); mid = hi; lo[0] = __float_as_uint(pg[r][2]); lo[1] = __float_as_uint(pg[r][3]); }
            else split3_f32x4(pg[r], hi, mid, lo);
            char* d = smem + buf + V3_G_OFF + lds0 + r * (V3_LT * 8);
            if (abl & 2) { asm volatile("" :: "v"(hi), "v"(mid), "v"(lo), "v"(d)); bsum += pg[r]; return; }
            *reinterpret_cast<u32x2*>(d) = hi;
            *reinterpret_cast<u32x2*>(d + V3_GT) = mid;
            *reinterpret_cast<u32x2*>(d + 2 * V3_GT) = lo;
            bsum += pg[r];
        };

        // prologue: tile 0 into LDS buffer 0, tile 1 into the staging registers
        {
            const TileAt a = tile_at(0);
#pragma unroll
            for (int r = 0; r < V3_X_ROUNDS; ++r) asm_load4(px[r], x_off(r, a), a.xrs);
#pragma unroll
            for (int r = 0; r < V3_G_ROUNDS; ++r) asm_load4(pg[r], g_off(r, a), a.grs);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < V3_X_ROUNDS; ++r) { asm volatile("" : "+v"(px[r])); store_x(r, 0); }
#pragma unroll
            for (int r = 0; r < V3_G_ROUNDS; ++r) { asm volatile("" : "+v"(pg[r])); store_g(r, 0); }
        }
        {
            const TileAt a = tile_at(1);
#pragma unroll
            for (int r = 0; r < V3_X_ROUNDS; ++r) asm_load4(px[r], x_off(r, a), a.xrs);
#pragma unroll
            for (int r = 0; r < V3_G_ROUNDS; ++r) asm_load4(pg[r], g_off(r, a), a.grs);
        }
        lds_barrier();                                                                     // (P)
        V3_LTICK(1);
#pragma unroll 1
        for (int k = 0; k < my_tiles; ++k) {
            // tile k+1: registers -> the other buffer; each register is refilled with tile k+2 right after its split
            const TileAt a = tile_at(k + 2);
            const int nb = ((k + 1) & 1) * V3_BUF;
#pragma unroll
            for (int r = 0; r < V3_X_ROUNDS; ++r) {
                asm_wait(px[r]);
                store_x(r, nb);
                asm_load4(px[r], x_off(r, a), a.xrs);
                __builtin_amdgcn_sched_barrier(0);   // one round at a time, in order (the wait counts depend on it)
            }
#pragma unroll
            for (int r = 0; r < V3_G_ROUNDS; ++r) {
                asm_wait(pg[r]);
                store_g(r, nb);
                asm_load4(pg[r], g_off(r, a), a.grs);
                __builtin_amdgcn_sched_barrier(0);
            }
            V3_LTICK(0);
            lds_barrier();
            V3_LTICK(1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef XSD_DIAG
        if (stamp && tid == 0) { atomicAdd(&P.dbg[16], st[0]); atomicAdd(&P.dbg[19], st[1]); }
#endif
        // tiles past the end were staged as zeros (empty descriptors): bsum took 0 from them; tile 0 and 1 were counted once each
    } else {
        // ============================== MFMA waves ==============================
        const int wv = wid - 4;
        // The MFMA waves are this kernel's critical path (round-5 stamps: their walk 4.3 - 4.7k cycles per tile for 3.46k of matrix
        // work, the staging wave of the same SIMD waits 1.4 - 1.8k cycles per tile at the barrier): they issue ahead of it.
        // Same device, alternating: 9.58 -> 9.14 ms per block launch (priority 3: the same).  (In wgrad_h2x.hip the two roles
        // are co-critical and the same switch was zero-sum.)
        __builtin_amdgcn_s_setprio(V3S_MPRIO);
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
        // per-lane base of the transposing reads: lane i of a 16-lane group addresses block row q = i>>2 (pixel) and
        // columns 4p..4p+3 (p = i&3) of channel group (lane>>4)&1; the lane half h selects pixels +8.
        const int i16 = lane & 15;
        const int lane_off = (8 * h + (i16 >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (i16 & 3) * 8;
        lds_barrier();                                                                     // (P)
        V3_TICK(1);
#if V3_PAIR
        // Round 5: a wave owns a PIXEL HALF of a ROW PAIR (mf = wv & 1: pixels 16 mf .. 16 mf + 15 = the K of a 32x32x16 MFMA; rows
        // 2 rp, 2 rp + 1, rp = wv >> 1) instead of a whole tile row.  The nine accumulators are sums over pixels, so they serve both
        // rows, and the fragment of halo row h at column offset dx is the dy = h operand of the upper row AND the dy = h - 1
        // operand of the lower one: 12 (dx, h) steps fetch 36 X fragments + 6 G fragments per wave and tile instead of 54 + 6
        // (LDS fragment reads per tile and CU: 480 -> 336 ds_read_b64_tr_b16, -30 %), for the same 108 MFMAs.  The images, the
        // staging waves and the final reduction (four partial sums per tap) are unchanged.  Same device: 10.19 -> 9.95 ms per block
        // launch.  (Tried on top and removed, tools/attic/wgrad_s3x_rotated_walk.patch: the walk rotated across the barrier -- a
        // tile's last six MFMAs issued behind the next tile's first 18 reads to cover their round trip: +0.4 % per launch; with a
        // branch on "is there a next tile" the compiler waits for lgkmcnt(0) in front of the deferred MFMAs: +1.8 %.  And TWO MFMA
        // waves per SIMD -- the nine taps dealt 5 | 4 to two waves of five accumulators each, 12 waves at 134 registers, so that one
        // wave's bubble at the barrier is the other's matrix time: parity green, 9.206 -> 9.211 ms per block launch at a 1.5 % LOWER
        // clock (tools/attic/wgrad_s3x_two_mfma_waves_per_simd.patch).  Fewer stall cycles at the same milliseconds: after the
        // priority change this kernel, too, sits at the package's energy bound, and schedules have nothing left to give.)
        const int mf = wv & 1, rp = wv >> 1;
#pragma unroll 1
        for (int k = 0; k < my_tiles; ++k) {
            const char* xbase = smem + (k & 1) * V3_BUF + (2 * rp) * (HALO_W * 64) + 16 * mf * 64 + lane_off;   // + term image + (h * 34 + dx) * 64
            const char* gbase = smem + (k & 1) * V3_BUF + V3_G_OFF + (2 * rp) * (TILE_W * 64) + 16 * mf * 64 + lane_off;   // + term image + q * 32 * 64
            constexpr int DEPTH = V3S_DEPTH;          // steps the X fragments are requested ahead of their MFMAs (DEPTH + 1 register sets)
            bf16x8 g[2][3], x[DEPTH + 1][3];
            auto load_g = [&](int q, bf16x8 (&d)[3]) {
#pragma unroll
                for (int term = 0; term < 3; ++term) d[term] = v3_tr_frag(gbase, term * V3_GT + q * (TILE_W * 64));
            };
            auto load_x = [&](int st, bf16x8 (&d)[3]) {      // step st = 4 dx + h
                const int dx = st >> 2, hr = st & 3;
#pragma unroll
                for (int term = 0; term < 3; ++term) d[term] = v3_tr_frag(xbase, term * V3_XT + (hr * HALO_W + dx) * 64);
            };
            auto mac6 = [&](f32x16& a, const bf16x8 (&xx)[3], const bf16x8 (&gg)[3]) {
                // Running accumulators take every product: 12 roundings per tile row and tap, against 32 for an fp32 fma chain
                // over the same 32 pixels (single-layer error vs float64: tools/dbg_layer.py).
#ifdef V3S_NOMFMA   // energy experiment (tools/power_table_strict.sh): the whole kernel but its matrix instructions -- results are garbage
                asm volatile("" :: "v"(xx[0]), "v"(xx[1]), "v"(xx[2]), "v"(gg[0]), "v"(gg[1]), "v"(gg[2]), "v"(a));
                return;
#endif
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xx[2], gg[0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xx[0], gg[2], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xx[1], gg[1], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xx[1], gg[0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xx[0], gg[1], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xx[0], gg[0], a, 0, 0, 0);
            };
            load_g(0, g[0]);
            load_x(0, x[0]);
            load_g(1, g[1]);
#pragma unroll
            for (int d = 1; d < DEPTH; ++d) load_x(d, x[d]);
#pragma unroll
            for (int st = 0; st < 12; ++st) {
                const int dx = st >> 2, hr = st & 3;
                if (st + DEPTH < 12) load_x(st + DEPTH, x[(st + DEPTH) % (DEPTH + 1)]);
                __builtin_amdgcn_sched_barrier(0);   // the requests go out BEFORE this step's MFMAs: 192 - 384 cycles of cover per step
                if (hr <= 2) mac6(acc[3 * hr + dx], x[st % (DEPTH + 1)], g[0]);            // upper row: tap (dy = hr, dx)
                if (hr >= 1) mac6(acc[3 * (hr - 1) + dx], x[st % (DEPTH + 1)], g[1]);      // lower row: tap (dy = hr - 1, dx)
                __builtin_amdgcn_sched_barrier(0);
            }
            V3_TICK(0);
            lds_barrier();
            V3_TICK(1);
        }
#else
#pragma unroll 1
        for (int k = 0; k < my_tiles; ++k) {
            const char* xbase = smem + (k & 1) * V3_BUF + wv * (HALO_W * 64) + lane_off;   // + term image + ((dy*34 + dx + 16*mf) * 64)
            const char* gbase = smem + (k & 1) * V3_BUF + V3_G_OFF + wv * (TILE_W * 64) + lane_off;
            // Running accumulators take every product: 12 roundings per tile row and tap, against 32 for an fp32 fma chain
            // over the same 32 pixels (single-layer error vs float64: tools/dbg_layer.py).
            // software pipeline over the 18 (pixel half, tap) steps: the fragments of step s+1 are requested before the six
            // MFMAs of step s; the scheduling barrier keeps hipcc from hoisting more reads than that (it otherwise runs
            // out of registers and spills accumulators inside the loop)
            bf16x8 g[2][3], x[2][3];
            auto load_g = [&](int mf, bf16x8 (&d)[3]) {
#pragma unroll
                for (int term = 0; term < 3; ++term) d[term] = v3_tr_frag(gbase, term * V3_GT + 16 * mf * 64);
            };
            auto load_x = [&](int mf, int tap, bf16x8 (&d)[3]) {
                const int dy = tap / 3, dx = tap % 3;
                const int off = (dy * HALO_W + dx + 16 * mf) * 64;
#pragma unroll
                for (int term = 0; term < 3; ++term) d[term] = v3_tr_frag(xbase, term * V3_XT + off);
            };
            load_g(0, g[0]);
            load_x(0, 0, x[0]);
#pragma unroll
            for (int s = 0; s < 18; ++s) {
                const int mf = s / 9, tap = s % 9;
                if (s + 1 < 18) {
                    if ((s + 1) % 9 == 0) load_g(1, g[1]);
                    load_x((s + 1) / 9, (s + 1) % 9, x[(s + 1) & 1]);
                }
                __builtin_amdgcn_sched_barrier(0);   // the requests go out BEFORE this step's MFMAs: 192 cycles of cover
                const bf16x8 (&gg)[3] = g[mf];
                const bf16x8 (&xx)[3] = x[s & 1];
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xx[2], gg[0], acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xx[0], gg[2], acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xx[1], gg[1], acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xx[1], gg[0], acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xx[0], gg[1], acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xx[0], gg[0], acc[tap], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            V3_TICK(0);
            lds_barrier();
            V3_TICK(1);
        }
#endif
#ifdef XSD_DIAG
        if (stamp && tid == V3_LT) { atomicAdd(&P.dbg[17], st[0]); atomicAdd(&P.dbg[18], st[1]); atomicAdd(&P.dbg[21], (unsigned long long)my_tiles); }
#endif
    }

    // ---- cross-wave reduction through LDS, one tap at a time (one 4 KiB slab per MFMA wave), then one coalesced store per tap
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    float* outp = P.partial + (P.npairs > 0 ? ((long long)part * P.npairs + slot) * 9 : (((long long)part * P.n_g + n) * P.n_in + j) * 9) * 1024;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        if (!loader) {
            const int wv = wid - 4;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ci = (i & 3) + 8 * (i >> 2) + 4 * h;
                red[wv * 1024 + ci * 32 + l31] = acc[tap][i];
            }
        }
        __syncthreads();
        for (int e = tid; e < 1024; e += V3_THREADS) {
            float sacc = 0.f;
#pragma unroll
            for (int w = 0; w < V3_TH; ++w) sacc += red[w * 1024 + e];      // fixed order: bitwise reproducible
            outp[tap * 1024 + e] = sacc;
        }
        __syncthreads();
    }
    if (j == 0) { // bias gradient: staging thread lt staged channels 4*(lt&7)..+3 of the G tiles
        if (loader) {
#pragma unroll
            for (int i = 0; i < 4; ++i) red[tid * 4 + i] = bsum[i];
        }
        __syncthreads();
        if (tid < 32) {
            const int q = tid >> 2, i = tid & 3;
            float sacc = 0.f;
            for (int w = 0; w < V3_LT / 8; ++w) sacc += red[(w * 8 + q) * 4 + i];
            P.bias_partial[((long long)part * P.n_g + n) * 32 + tid] = sacc;
        }
    }
}

hipError_t launch_wgrad_s3x(const WgradParams& p, hipStream_t stream)
{
    static PerDevice once_;
    hipError_t e = once_.once([]() {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_s3x_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS_BYTES);
    }, nullptr);
    if (e != hipSuccess) return e;
    if ((p.nparts & 7) && !(p.npairs > 0 && (p.nparts & 7) == 1)) return hipErrorInvalidValue;
    // 32-bit byte offsets inside one batch slice of a plane (buffer loads); xsd_forward rejects such images with a message
    for (int i = 0; i < p.n_in; ++i) if ((long long)p.H * p.x[i].rs * 4 >= (1ll << 31)) return hipErrorInvalidValue;
    for (int i = 0; i < p.n_g; ++i) if ((long long)p.H * p.g[i].rs * 4 >= (1ll << 31)) return hipErrorInvalidValue;
    if (p.npairs < 0 || p.npairs > 16 || p.n_in > 5 || p.n_g > 5) return hipErrorInvalidValue;
    for (int s = 0; s < p.npairs; ++s)
        if ((int)((p.pair_j >> (4 * s)) & 15) >= p.n_in || (int)((p.pair_n >> (4 * s)) & 15) >= p.n_g) return hipErrorInvalidValue;
    // pair lists with nparts = 8 m + 1: m * npairs workgroups per XCD for the full parts + ceil(npairs / 8) per XCD for the last part
    const int pair_grid = 8 * ((p.nparts >> 3) * p.npairs + ((p.nparts & 7) ? (p.npairs + 7) / 8 : 0));
    const dim3 g(p.npairs > 0 ? pair_grid : p.nparts * p.n_in * p.n_g), b(V3_THREADS);
    hipLaunchKernelGGL(wgrad_s3x_kernel, g, b, V3_LDS_BYTES, stream, p);
    return hipGetLastError();
}

} // namespace xsd
