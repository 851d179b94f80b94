// wgrad_h2x.hip -- math mode 4 ("f16x3"): weight gradient of the 3x3 convs over fp32 planes with the two-term fp16 arithmetic
// of conv3x3_h2x.hip; role-split workgroup in the manner of wgrad_s3x.hip (here 4 staging + 8 MFMA waves, one workgroup per CU, 8 x 32-pixel
// tiles, two LDS buffers, one barrier per tile; read that file for the staging scheme).  Replaces autograd's conv
// weight-gradient for the reference's nn.Conv2d(32k -> 32n, 3,1,1) layers (rrdb_blocks.py:27-31; generator_rrdb.py:38-44,95,101):
//     dW[co][ci][tap] = sum_{b,y,x} G[b,y,x,co] * X[b,y+dy-1,x+dx-1,ci],   db[co] = sum G[..,co]
// Arithmetic: X and G are scaled by powers of two from their max |x| slots and split into h + l * 2^-11 (xsd_split.h); the
// product is Xh*Gh + (Xh*Gl + Xl*Gh) * 2^-11: three fp16 MFMA products per (pixels, tap) instead of six bf16 ones, issued as
// v_mfma_f32_16x16x32_f16 (four 16x16 tiles per 32x32 accumulator, K = the 32 pixels of a tile row).
// The three products carry different weights, so they need their own accumulators: 27 32x32 accumulators (9 taps x {hh, hl, lh})
// do not fit one wave.  The 27 single-MFMA "units" are dealt 4,4,4,3,3,3,3,3 to eight MFMA waves (7,7,7,6 per SIMD); each wave walks ALL rows of
// the tile for its units, and the weighted sum hh + 2^-11 (hl + lh), un-scaled exactly, is formed in the final fixed-order reduction.
// Round 4: the deal is by tap COLUMN.  A wave's three units are the taps (dy = 0, 1, 2; dx) of one product: at tile row r they
// multiply the halo rows r, r+1, r+2 at column offset dx, so a row's X fragment is fetched ONCE and serves three steps from a
// four-row register ring (one new fragment per step instead of three), and every wave needs only ONE term image of G.  Eight
// of the nine (product, dx) triples go to the eight waves; the ninth (lh, dx = 2) is split over the three waves that hold four
// units.  LDS fragment reads per tile row and CU: 144 -> 76 ds_read_b64_tr_b16 (the round-3 deal, units in tap order, gave
// every unit its own fragment every step; the kernel's waves spent 26 % of their cycles stalled on LDS issue).
// LDS: four images per buffer (X_h, X_l over the 10 x 34 halo, G_h, G_l), each [channel half][pixel][16 x f16]: 76,800 B, two buffers.
// Round 4, second half: a workgroup walks a contiguous chunk of the tiles DOWN 32-pixel column strips; a tile's two top halo rows are
// copied LDS -> LDS from the tile before it, the staging waves fetch and split the 8 rows below them (17 counted loads per tile and
// thread; constants and the warm-up item below).  Measured with compile-time ablations of both roles (tools/ablate_wgrad.sh): the
// staging waves alone need 4.66k cycles per tile, the MFMA waves alone 4.76k (3.58k of matrix work), together 5.6 - 6.0k -- the two
// roles share one vector-issue port per SIMD, which a 16x16x32 MFMA holds 8 cycles of every 16 -- and in MILLISECONDS the kernel
// is set by the 1400 W package power cap.
#include <type_traits>
#include "xsd_kernels.h"
#include "xsd_split.h"

namespace xsd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// (MFMA shape: v_mfma_f32_16x16x32_f16.  Round 2 A/B against the 32x32x16 form: this kernel 3.4 % slower per launch, the train
// step 0.7 % FASTER -- the chip is power-limited, the 16x16x32 shape draws less, and the conv launches between the weight-gradient
// launches run at a 2.5 % higher clock for it; the 32x32x16 variant left the tree in round 4, git history has it.)
#ifndef V3_TH_ROWS
#define V3_TH_ROWS 8     // 8-row tiles: halo 10/8 instead of 6/4 rows staged per row computed (4 rows: +6.7 % kernel time, one device)
#endif
#ifndef V3_LPRIO
#define V3_LPRIO 1       // s_setprio of the staging waves
#endif
constexpr int V3_TH = V3_TH_ROWS;                                     // tile rows
constexpr int V3_MW = 8;                                              // MFMA waves: the 27 accumulators dealt 4,4,4,3,3,3,3,3
#ifndef V3_LWAVES
#define V3_LWAVES 4     // staging waves, one per SIMD, beside two MFMA waves per SIMD (12 waves, 168 registers).  Eight staging
#endif                  // waves (16 waves, 128 registers; -DV3_LWAVES=8) measured 9 % slower on one device (with 4-row tiles): the
                        // staging waves are not the critical path (clean stamps of round 3: they wait ~1.8k cycles per tile at
                        // the barrier for the younger MFMA wave of their SIMD).
constexpr int V3_LT = 64 * V3_LWAVES;                                 // staging threads (waves 0 .. V3_LWAVES-1)
constexpr int V3_THREADS = V3_LT + 64 * V3_MW;                        // 768
constexpr int V3_HPX = (V3_TH + 2) * HALO_W;                          // 340 halo pixels
// Round 4, second half: a workgroup walks its tiles DOWN a 32-pixel column strip, so the two top halo rows of a tile are the two
// bottom halo rows of the tile before it -- already split, already in the other LDS buffer.  The staging waves fetch and split
// only the V3_TH rows below them (LDS rows 2 .. V3_TH+1) and copy the two top rows LDS -> LDS (8.7 KB per tile): 9 X rounds
// instead of 11, 17 loads per tile and staging thread instead of 19.  The first tile of a workgroup's chunk is staged whole by
// the prologue; a strip start inside the chunk is preceded by a WARM-UP item, the virtual tile ty = -1 of that strip (rows
// -7 .. 0: everything above the image fails the buffer range check and reads as zeros, row 0 is real) with an empty G
// descriptor (its products are zeros) -- one item in tilesY + 1, branch-free in the counted-load loop.
constexpr int V3_X_SLOTS = V3_TH * HALO_W * 8;                        // 2176 (pixel, channel quad) slots of the fetched rows
constexpr int V3_X_ROUNDS = (V3_X_SLOTS + V3_LT - 1) / V3_LT;         // 9
constexpr int V3_TOP_SLOTS = 2 * HALO_W * 8;                          // 544: the two top rows (prologue only)
constexpr int V3_TOP_ROUNDS = (V3_TOP_SLOTS + V3_LT - 1) / V3_LT;     // 3
constexpr int V3_G_SLOTS = V3_TH * TILE_W * 8;                        // 2048
constexpr int V3_G_ROUNDS = V3_G_SLOTS / V3_LT;                       // 8
constexpr int V3_NL = V3_X_ROUNDS + V3_G_ROUNDS;                      // 17 loads per tile and staging thread
// A term image is [channel half][pixel][16 x f16] (32-B records); the 32 lanes of a transposing read then take 256 contiguous
// bytes (8 pixels of one channel half): conflict-free at any pixel offset.  A staging write (`ds_write_b64`: 16-lane groups over
// 32 four-byte banks, MI355X_MICROARCH.md LDS table) covers 2 pixels x both channel halves x 32 B per group: the half-image
// strides are = 64 mod 128 so that the four 32-byte pieces of a group fall into four different bank octets.  (Until round 4
// the strides were = 128 mod 256 -- right for 64 banks, 2-way conflicted on 32: SQ_LDS_BANK_CONFLICT 7.2e9 of 3.1e10 LDS cycles,
// 570 cycles per tile = the 152 staging writes x 4 extra cycles; profiles/r04_pmc_sq_f16x3_dn_train_b32.txt.)
constexpr int V3_XH = V3_HPX * 32 + 64;                               // 10,944 B per X half image
constexpr int V3_XT = 2 * V3_XH;                                      // 21,888 B per X term image
constexpr int V3_GH = V3_TH * TILE_W * 32 + 64;                       // 8,256 B per G half image
constexpr int V3_GT = 2 * V3_GH;                                      // 16,512 B per G term image
static_assert(V3_XH % 128 == 64 && V3_GH % 128 == 64 && V3_XH % 16 == 0, "half-image strides");
constexpr int V3_G_OFF = 2 * V3_XT;                                   // 43,776
constexpr int V3_BUF = V3_G_OFF + 2 * V3_GT;                          // 76,800 B per buffer
constexpr int V3_SINK = 2 * V3_BUF;                                   // writes of exhausted slots land behind the buffers (hi at +0, lo at +512)
constexpr int V3_ROW2 = 2 * HALO_W * 32;                              // 2,176 B: two halo rows of one half image
constexpr int V3_CPY_UNITS = 4 * (V3_ROW2 / 16);                      // 544 16-byte units: {X_h, X_l} x {channel half} x two rows
constexpr int V3_CPY_ROUNDS = (V3_CPY_UNITS + V3_LT - 1) / V3_LT;     // 3
constexpr int V3_RED = 27 * 4096;                                     // final reduction: one 4 KiB slab per accumulator
constexpr int V3_LDS_BYTES = (V3_SINK + 1024) > V3_RED ? (V3_SINK + 1024) : V3_RED;     // 154,624
static_assert(V3_LDS_BYTES <= 160 * 1024, "LDS");

// Which of the 27 accumulators ("units", u = 9 * product + tap; product 0 = Xh*Gh, 1 = Xh*Gl, 2 = Xl*Gh; tap = 3 dy + dx) MFMA wave w
// holds in slot q: slots 0..2 = the column triple (product, dx) of the wave, taps dy = q; slot 3 (waves 0..2 only) = one tap of
// the split triple (lh, dx = 2), dy = w.  Waves w and w + 4 share a SIMD: 7, 7, 7, 6 units per SIMD.
__host__ __device__ constexpr int v3_triple_prod(int w) { return w < 3 ? 0 : (w < 6 ? 1 : 2); }
__host__ __device__ constexpr int v3_triple_dx(int w) { return w < 3 ? w : (w < 6 ? w - 3 : w - 6); }
__host__ __device__ constexpr int v3_unit(int w, int q)
{
    return q < 3 ? 9 * v3_triple_prod(w) + 3 * q + v3_triple_dx(w) : (w < 3 ? 18 + 3 * w + 2 : -1);
}
constexpr bool v3_deal_is_a_partition()
{
    bool seen[27] = {};
    for (int w = 0; w < 8; ++w)
        for (int q = 0; q < 4; ++q) {
            const int u = v3_unit(w, q);
            if (u < 0) continue;
            if (u >= 27 || seen[u]) return false;
            seen[u] = true;
        }
    for (int u = 0; u < 27; ++u) if (!seen[u]) return false;
    return true;
}
static_assert(v3_deal_is_a_partition(), "every (product, tap) accumulator is held by exactly one wave");

__global__ __launch_bounds__(V3_THREADS) void wgrad_h2x_kernel(const WgradParams P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wid < V3_LWAVES;   // wave-uniform role

    // 1-D grid decode (as in wgrad_s3x.hip): the n_in workgroups that read the SAME G tiles have linear ids 8 apart -> one XCD
    const int lin = blockIdx.x;
    const int xcd = lin & 7, qq = lin >> 3;
    const int parts8 = P.nparts >> 3;
    int j, n, part, slot = 0;
    if (P.npairs > 0) {                               // pair-list launch: slot = pair, all slots of a part side by side on one XCD
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
    // This workgroup's chunk: the tiles [t_lo, t_hi) of the order (strip = b * tilesX + tx, ty) with ty fastest, i.e. down the
    // column strips.  Its ITEMS are those tiles plus one warm-up item (ty = -1, no G) in front of every strip that starts
    // inside the chunk: item k sits at position q0 + k of the strips laid end to end with tilesY + 1 positions each
    // (position p of a strip <-> ty = p - 1).
    const int t_lo = (int)((long long)part * ntiles / P.nparts), t_hi = (int)((long long)(part + 1) * ntiles / P.nparts);
    const int SE = tilesY + 1;
    const int strip0 = t_lo / tilesY;
    const int q0 = t_lo - strip0 * tilesY + 1;
    int my_tiles = 0;                                 // items, warm-ups included
    if (t_hi > t_lo) {
        const int se = (t_hi - 1) / tilesY;
        my_tiles = (se - strip0) * SE + ((t_hi - 1) - se * tilesY + 1) - q0 + 1;
    }

    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

#ifdef XSD_DIAG   // phase stamps (diagnostic library only; tools/stamps_train.py): slot 16 staging rounds, 17 MFMA walk, 18 MFMA
                  // wave at the barrier, 19 staging wave at the barrier, 20 of [16]: inside the counted data waits, 21 tiles
                  // (slots 0-15 belong to the conv kernels)
    unsigned long long st[2] = {0, 0};
    unsigned long long wst = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    const bool stamp = P.dbg != nullptr;
#define V3_TICK(i) do { if (stamp) { const unsigned long long t_ = __builtin_readcyclecounter(); st[i] += t_ - t0; t0 = t_; } } while (0)
#else
#define V3_TICK(i) do { } while (0)
#endif
#ifdef XSD_DIAG   // staging-wave stamps: branch-free (a branch inside the staging loop splits the live ranges of its in-flight load
                  // registers and hipcc then copies them at the loop entry, before the loads have landed: ISA check, -DXSD_DIAG)
#define V3_LTICK(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); st[i] += t_ - t0; t0 = t_; } while (0)
#else
#define V3_LTICK(i) do { } while (0)
#endif
    // operand scales (powers of two) from the planes' max |x| slots; every wave computes the same values (scalar loads)
    float sx, sg, inv_sx, inv_sg;
    {
        float ax = 1.f, ag = 1.f;
#pragma unroll
        for (int i = 0; i < 5; ++i) if (i == j && P.amax_x[i]) ax = *P.amax_x[i];     // static indices into the kernel arguments
#pragma unroll
        for (int i = 0; i < 5; ++i) if (i == n && P.amax_g[i]) ag = *P.amax_g[i];
        sx = scale_for_amax(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ax))), inv_sx);
        sg = scale_for_amax(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ag))), inv_sg);
    }
    const float inv_s = inv_sx * inv_sg;

#if defined(XSD_DIAG) && defined(XSD_ABL)   // timing experiments: a COMPILE-TIME constant (-DXSD_DIAG -DXSD_ABL=n builds; a run-time
                                            // value puts the hand-counted loads and waits under branches hipcc cannot keep exact)
    constexpr int abl = XSD_ABL;     // staging waves: 1 no split, 2 no LDS writes, 4 no loads and no counted waits, 128 nothing but the barrier;
                                     // MFMA waves: 32 no MFMAs (fragment reads only), 256 no fragment reads (MFMAs only), 64 nothing but the barrier
                                     // (results are garbage: read the stamps' CYCLES, not milliseconds -- degenerate operands raise the clock)
#elif defined(V3_NOMFMA)   // energy experiment (tools/power_table.sh): the whole kernel but its matrix instructions
    constexpr int abl = 32;
#else
    constexpr int abl = 0;
#endif
    constexpr int NU = 4;              // accumulators ("units") per MFMA wave: 27 = 4 + 4 + 4 + 3 + 3 + 3 + 3 + 3
    f32x4 acc[NU][2][2];               // [unit][input-channel half][output-channel half]: four 16x16 tiles per unit
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f}; // staging thread: its 4 channels (lt & 7) of the G tiles it stages

    if (loader) {
        // ============================ staging waves ============================
        // Round 5: the staging waves are the longer role since the column ring (stamps at the bench batch: their rounds 5.2k cycles per
        // tile, the MFMA waves wait 1.5k per tile at the barrier): they issue ahead.  Same device, alternating: 6.075 -> 5.990 ms
        // per block launch (-1.4 %; priority 3 the same), step +0.5 %.  (Handing the two-row LDS -> LDS copy to the waiting MFMA
        // waves instead measured +1.1 % per launch: tools/attic/wgrad_h2x_mfma_wave_halo_copy.patch.)
        __builtin_amdgcn_s_setprio(V3_LPRIO);
        const int lt = tid;
        const PlaneIn xp = P.x[j];
        const PlaneIn gp = P.g[n];
        constexpr int OOR = (int)0x80000000;          // lane offset that fails every range check -> the load returns 0
        // per-slot constants: X round r -> pixel (hy, hx) of the FETCHED rows (LDS row hy + 2), channel quad c; byte offset
        // relative to the tile origin (= the first fetched row, image row y0 + 1, column x0 - 1)
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
        const int lds0 = (lt >> 3) * 32 + (lt & 3) * 8;        // + ((lt >> 2) & 1) * half-image stride (X and G differ) + r * V3_LT * 4
        const int xh0 = lds0 + ((lt >> 2) & 1) * V3_XH, gh0 = lds0 + ((lt >> 2) & 1) * V3_GH;
        constexpr int RL = V3_X_ROUNDS - 1;                      // last X round: only part of the threads have a slot,
        const bool live6 = RL * V3_LT + lt < V3_X_SLOTS;         // the others write a sink
        // The two top rows by LDS copy: 16-byte unit u = i * V3_LT + lt -> region u / 136 (term image, channel half), byte
        // 16 * (u % 136) of the region's two rows; source = rows V3_TH, V3_TH + 1 of the buffer being multiplied, destination =
        // rows 0, 1 of the buffer being filled.  Threads past the last unit copy 16 bytes of the sink onto themselves.
        int cpy[V3_CPY_ROUNDS];
#pragma unroll
        for (int i = 0; i < V3_CPY_ROUNDS; ++i) {
            const int u = i * V3_LT + lt;
            const int reg = u / (V3_ROW2 / 16), w16 = u - reg * (V3_ROW2 / 16);
            cpy[i] = u < V3_CPY_UNITS ? (reg >> 1) * V3_XT + (reg & 1) * V3_XH + w16 * 16 : -1;
        }

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
        auto asm_wait = [&](f32x4& v) { asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(v) : [n] "n"(V3_NL - 1) : "memory"); };
        static_assert(V3_NL - 1 <= 63, "vmcnt is a 6-bit counter");

        struct TileAt { i32x4 xrs, grs; int xorg, gorg, x0; };
        auto tile_at = [&](int k) {        // descriptors and origin offsets of this workgroup's k-th item (empty past the end)
            const bool live = k < my_tiles;
            const int q = q0 + k;
            const int sidx = q / SE;
            const int ty = q - sidx * SE - 1;                  // -1: the warm-up item of a strip (no G)
            const int strip = strip0 + sidx;
            const int b = live ? strip / P.tilesX : 0;
            const int tx = strip - (strip / P.tilesX) * P.tilesX;
            TileAt a;
            a.x0 = tx * TILE_W;
            const int y0 = ty * V3_TH;
            a.xrs = make_rsrc(reinterpret_cast<unsigned long long>(xp.p + (long long)b * xp.bs), live ? x_bytes : 0u);
            a.grs = make_rsrc(reinterpret_cast<unsigned long long>(gp.p + (long long)b * gp.bs), (live && ty >= 0) ? g_bytes : 0u);
            a.xorg = ((y0 + 1) * xp.rs + (a.x0 - 1) * xp.ps) * 4;    // rows above / below the image fall outside [0, bytes): zeros
            a.gorg = (y0 * gp.rs + a.x0 * gp.ps) * 4;                 // (the warm-up item's G offsets are negative AND its descriptor is empty)
            return a;
        };
        // columns left / right of the image would alias the neighbouring row: those lanes get the failing offset
        auto x_off = [&](int r, const TileAt& a) { return ((unsigned)(a.x0 - 1 + xhx[r]) < (unsigned)P.W) ? a.xorg + xrel[r] : OOR; };
        auto g_off = [&](int r, const TileAt& a) { return (a.x0 + ggx[r] < P.W) ? a.gorg + grel[r] : OOR; };
        auto split_to = [&](const f32x4& v, float s, char* d, int term_stride) {
            u32x2 hi, lo;
            if (abl & 1) { hi[0] = __float_as_uint(v[0]); hi[1] = __float_as_uint(v[1]); lo[0] = __float_as_uint(v[2]); lo[1] = __float_as_uint(v[3]); }
            else split2_f16x4(v, s, hi, lo);
            if (abl & 2) { asm volatile("" :: "v"(hi), "v"(lo), "v"(d)); return; }   // diag: no LDS writes
            *reinterpret_cast<u32x2*>(d) = hi;
            *reinterpret_cast<u32x2*>(d + term_stride) = lo;
        };
        auto store_x = [&](int r, int buf) {            // fetched rows: LDS rows 2 ..
            const bool sink = (r == RL && !live6);
            split_to(px[r], sx, smem + (sink ? V3_SINK + (lt & 63) * 8 : buf + V3_ROW2 + xh0 + r * (V3_LT * 4)), sink ? 512 : V3_XT);
        };
        auto store_g = [&](int r, int buf) {
            split_to(pg[r], sg, smem + buf + V3_G_OFF + gh0 + r * (V3_LT * 4), V3_GT);
            bsum += pg[r];
        };

        // prologue: item 0 (the chunk's first tile, never a warm-up) WHOLE into LDS buffer 0 -- its two top rows through three
        // extra rounds of loads --, item 1 into the staging registers
        {
            const TileAt a = tile_at(0);
            f32x4 ptop[V3_TOP_ROUNDS] = {};
#pragma unroll
            for (int r = 0; r < V3_TOP_ROUNDS; ++r) {
                const int slot = r * V3_LT + lt;
                const int p = slot >> 3, c = slot & 7;
                const int hy = p / HALO_W, hx = p - hy * HALO_W;
                const bool ok = slot < V3_TOP_SLOTS && ((unsigned)(a.x0 - 1 + hx) < (unsigned)P.W);
                asm_load4(ptop[r], ok ? a.xorg + ((hy - 2) * xp.rs + hx * xp.ps + c * 4) * 4 : OOR, a.xrs);
            }
#pragma unroll
            for (int r = 0; r < V3_X_ROUNDS; ++r) asm_load4(px[r], x_off(r, a), a.xrs);
#pragma unroll
            for (int r = 0; r < V3_G_ROUNDS; ++r) asm_load4(pg[r], g_off(r, a), a.grs);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < V3_TOP_ROUNDS; ++r) {
                asm volatile("" : "+v"(ptop[r]));
                const bool sink = r * V3_LT + lt >= V3_TOP_SLOTS;
                split_to(ptop[r], sx, smem + (sink ? V3_SINK + (lt & 63) * 8 : xh0 + r * (V3_LT * 4)), sink ? 512 : V3_XT);
            }
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
        // round i of a pass: X round i for i < V3_X_ROUNDS, then the G rounds
        auto rreg = [&](int i) -> f32x4& { return i < V3_X_ROUNDS ? px[i] : pg[i - V3_X_ROUNDS]; };
        auto rdst = [&](int i, int buf) -> char* {
            if (i < V3_X_ROUNDS) return smem + ((i == RL && !live6) ? V3_SINK + (lt & 63) * 8 : buf + V3_ROW2 + xh0 + i * (V3_LT * 4));
            return smem + buf + V3_G_OFF + gh0 + (i - V3_X_ROUNDS) * (V3_LT * 4);
        };
        auto rstride = [&](int i) { return i < V3_X_ROUNDS ? ((i == RL && !live6) ? 512 : V3_XT) : V3_GT; };
        // The cursor of the item whose loads a pass issues (item k + 2), advanced by compare-and-select instead of decoded by two
        // integer divisions per pass; the lanes' column validity is a property of the STRIP, so the offsets with the failing
        // value folded in are rebuilt only when the cursor enters a strip (a wave-uniform branch that touches no load register).
        struct Cur3 { int ty, tx, b; };
        Cur3 c2;
        {
            const int q = q0 + 2, sidx = q / SE, strip = strip0 + sidx;
            c2.ty = q - sidx * SE - 1; c2.b = strip / P.tilesX; c2.tx = strip - c2.b * P.tilesX;
        }
        int xcol[V3_X_ROUNDS], gcol[V3_G_ROUNDS];
        auto strip_offsets = [&](int tx) {
            const int x0 = tx * TILE_W;
#pragma unroll
            for (int r = 0; r < V3_X_ROUNDS; ++r) xcol[r] = ((unsigned)(x0 - 1 + xhx[r]) < (unsigned)P.W) ? xrel[r] : OOR;
#pragma unroll
            for (int r = 0; r < V3_G_ROUNDS; ++r) gcol[r] = (x0 + ggx[r] < P.W) ? grel[r] : OOR;
        };
        strip_offsets(c2.tx);
#pragma unroll 1
        for (int k = 0; k < my_tiles; ++k) {
            if (abl & 128) { lds_barrier(); continue; }
            // item k+1: registers -> the other buffer; each register is refilled with item k+2 right after its split
            TileAt a;
            {
                const bool live = k + 2 < my_tiles;
                if (c2.ty == -1) strip_offsets(c2.tx);
                a.x0 = c2.tx * TILE_W;
                const int y0 = c2.ty * V3_TH;
                a.xrs = make_rsrc(reinterpret_cast<unsigned long long>(xp.p + (long long)c2.b * xp.bs), live ? x_bytes : 0u);
                a.grs = make_rsrc(reinterpret_cast<unsigned long long>(gp.p + (long long)c2.b * gp.bs), (live && c2.ty >= 0) ? g_bytes : 0u);
                a.xorg = ((y0 + 1) * xp.rs + (a.x0 - 1) * xp.ps) * 4;
                a.gorg = (y0 * gp.rs + a.x0 * gp.ps) * 4;
                if (++c2.ty == tilesY) { c2.ty = -1; if (++c2.tx == P.tilesX) { c2.tx = 0; ++c2.b; } }
            }
            auto roff = [&](int i) { return i < V3_X_ROUNDS ? a.xorg + xcol[i] : a.gorg + gcol[i - V3_X_ROUNDS]; };   // (0x80000000 + a small origin still fails the range check)
            auto rrs = [&](int i) -> const i32x4& { return i < V3_X_ROUNDS ? a.xrs : a.grs; };
            const int cb = (k & 1) * V3_BUF, nb = ((k + 1) & 1) * V3_BUF;
            // item k+1's two top rows = item k's two bottom rows (complete since the barrier that ended the previous pass)
            f32x4 ctmp[V3_CPY_ROUNDS];
#pragma unroll
            for (int i = 0; i < V3_CPY_ROUNDS; ++i)
                ctmp[i] = *reinterpret_cast<const f32x4*>(smem + (cpy[i] >= 0 ? cb + V3_TH * HALO_W * 32 + cpy[i] : V3_SINK + (lt & 63) * 16));
#pragma unroll
            for (int i = 0; i < V3_NL; ++i) {
#ifdef XSD_DIAG
                const unsigned long long w0_ = __builtin_readcyclecounter();
#endif
                if (!(abl & 4)) asm_wait(rreg(i));
#ifdef XSD_DIAG
                wst += __builtin_readcyclecounter() - w0_;
#endif
                split_to(rreg(i), i < V3_X_ROUNDS ? sx : sg, rdst(i, nb), rstride(i));
                if (i >= V3_X_ROUNDS) bsum += rreg(i);
                if (!(abl & 4)) asm_load4(rreg(i), roff(i), rrs(i));
                if (i == V3_X_ROUNDS - 1) {
#pragma unroll
                    for (int c = 0; c < V3_CPY_ROUNDS; ++c)
                        *reinterpret_cast<f32x4*>(smem + (cpy[c] >= 0 ? nb + cpy[c] : V3_SINK + (lt & 63) * 16)) = ctmp[c];
                }
                __builtin_amdgcn_sched_barrier(0);   // one round at a time, in order (the wait counts depend on it)
            }
            V3_LTICK(0);
            lds_barrier();
            V3_LTICK(1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef XSD_DIAG
        if (stamp && tid == 0) { atomicAdd(&P.dbg[16], st[0]); atomicAdd(&P.dbg[19], st[1]); atomicAdd(&P.dbg[20], wst); }
#endif
        // tiles past the end were staged as zeros (empty descriptors): bsum took 0 from them; tile 0 and 1 were counted once each
    } else {
        // ============================== MFMA waves ==============================
#ifdef V3_MPRIO
        __builtin_amdgcn_s_setprio(V3_MPRIO);
#endif
        const int wv = wid - V3_LWAVES;        // MFMA wave index: its units are v3_unit(wv, 0..3)
#pragma unroll
        for (int k = 0; k < NU; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[k][i >> 3][(i >> 2) & 1][i & 3] = 0.f;
        // per-lane base of the transposing reads: lane i of a 16-lane group addresses block row q = i>>2 (pixel) and
        // columns 4p..4p+3 (p = i&3) of channel group (lane>>4)&1; the lane half h selects pixels +8.
        const int i16 = lane & 15;
        // 16x16x32: lane group g = lane >> 4 supplies k = 8g .. 8g+7, which this kernel maps to the pixels 4g .. 4g+3 (first
        // read) and 16 + 4g .. 16 + 4g+3 (second read) of the 32-pixel row -- any k <-> pixel map serves, X and G use the same
        const int lane_off = (4 * (lane >> 4) + (i16 >> 2)) * 32 + (i16 & 3) * 8;
        lds_barrier();                                                                     // (P)
        V3_TICK(1);
        // one instantiation per wave (the unit table is a compile-time function of the wave index)
        auto walk = [&](auto WV) {
            constexpr int w = decltype(WV)::value;
            constexpr int prodT = v3_triple_prod(w), dxT = v3_triple_dx(w);
            constexpr bool single = w < 3;                                   // fourth unit: (lh, dy = w, dx = 2)
            constexpr int x_img = prodT == 2 ? V3_XT : 0;                    // the triple's X term image: low term for lh
            constexpr int g_img = prodT == 1 ? V3_GT : 0;                    // the wave's G term image: low term for hl (the single is lh: G high, like hh)
            static_assert(!single || prodT == 0, "the waves with a fourth unit hold hh triples: one G image serves both");
#pragma unroll 1
            for (int k = 0; k < my_tiles; ++k) {
                const char* xb = smem + (k & 1) * V3_BUF + lane_off;                       // + term image + half image + (halo row * 34 + dx) * 32
                const char* gb = smem + (k & 1) * V3_BUF + V3_G_OFF + lane_off + g_img;    // + half image + row * 32 * 32
                // 2 * V3_TH steps (tile row r = st >> 1, input-channel half a = st & 1) of 32 pixels; per unit and step two MFMAs (the
                // two output-channel halves).  Halo row h's fragment of the wave's column lives in xr[h & 3][a] from the step
                // that requests it (two rows ahead of its first use) until tile row h has used it as its dy = 0 operand; the G
                // fragments of a row are fetched with its a = 0 step and kept for a = 1; the single unit's fragment is
                // requested one step ahead.
                f16x8 xr[4][2], xs[2], gf[2][2];      // gf[row parity][output-channel half]
                auto frag16 = [&](const char* base, int off) {
                    typedef __attribute__((address_space(3))) s16x4* lds_p;
                    if (abl & 256) { f16x8 z; asm volatile("" : "=v"(z)); return z; }
                    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + off));
                    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + off + 16 * 32));
                    s16x8 v;
                    v[0] = lo4[0]; v[1] = lo4[1]; v[2] = lo4[2]; v[3] = lo4[3];
                    v[4] = hi4[0]; v[5] = hi4[1]; v[6] = hi4[2]; v[7] = hi4[3];
                    return __builtin_bit_cast(f16x8, v);
                };
                auto load_row = [&](int hrow, int a) { xr[hrow & 3][a] = frag16(xb, x_img + a * V3_XH + (hrow * HALO_W + dxT) * 32); };
                auto load_g = [&](int r) {
#pragma unroll
                    for (int b = 0; b < 2; ++b) gf[r & 1][b] = frag16(gb, b * V3_GH + r * TILE_W * 32);
                };
                auto load_single = [&](int st) { xs[st & 1] = frag16(xb, V3_XT + (st & 1) * V3_XH + (((st >> 1) + w) * HALO_W + 2) * 32); };
                // what step 0 and 1 need: halo rows 0, 1, 2 in both halves, G row 0, the single's first fragment
                if (abl & 64) { V3_TICK(0); lds_barrier(); V3_TICK(1); continue; }
                load_g(0);
#pragma unroll
                for (int a = 0; a < 2; ++a) { load_row(0, a); load_row(1, a); load_row(2, a); }
                if (single) load_single(0);
#pragma unroll
                for (int st = 0; st < 2 * V3_TH; ++st) {
                    const int r = st >> 1, a = st & 1;
                    // requests for later steps go out BEFORE this step's MFMAs: halo row r + 3 (first used by tile row r + 1; its
                    // slot was last read by tile row r - 1), the next row's G with the a = 1 step, the single's next fragment
                    if (r + 3 < V3_TH + 2) load_row(r + 3, a);
                    if (a == 1 && r + 1 < V3_TH) load_g(r + 1);
                    if (single && st + 1 < 2 * V3_TH) load_single(st + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 3; ++q)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            if (abl & 32) asm volatile("" :: "v"(xr[(r + q) & 3][a]), "v"(gf[r & 1][b]));
                            else acc[q][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xr[(r + q) & 3][a], gf[r & 1][b], acc[q][a][b], 0, 0, 0);
                    if (single) {
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            if (abl & 32) asm volatile("" :: "v"(xs[st & 1]), "v"(gf[r & 1][b]));
                            else acc[3][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xs[st & 1], gf[r & 1][b], acc[3][a][b], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                V3_TICK(0);
                lds_barrier();
                V3_TICK(1);
            }
        };
        switch (wv) {
        case 0: walk(std::integral_constant<int, 0>{}); break;
        case 1: walk(std::integral_constant<int, 1>{}); break;
        case 2: walk(std::integral_constant<int, 2>{}); break;
        case 3: walk(std::integral_constant<int, 3>{}); break;
        case 4: walk(std::integral_constant<int, 4>{}); break;
        case 5: walk(std::integral_constant<int, 5>{}); break;
        case 6: walk(std::integral_constant<int, 6>{}); break;
        default: walk(std::integral_constant<int, 7>{}); break;
        }
#ifdef XSD_DIAG
        if (stamp && tid == V3_LT) { atomicAdd(&P.dbg[17], st[0]); atomicAdd(&P.dbg[18], st[1]); atomicAdd(&P.dbg[21], (unsigned long long)my_tiles); }
#endif
    }

    // ---- final reduction through LDS: every accumulator into its own 4 KiB slab (slab = unit number), then per tap the
    // weighted sum hh + 2^-11 (hl + lh), un-scaled (a power of two: exact), in a fixed order (bitwise reproducible)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    float* outp = P.partial + (P.npairs > 0 ? ((long long)part * P.npairs + slot) * 9 : (((long long)part * P.n_g + n) * P.n_in + j) * 9) * 1024;
    if (!loader) {
        const int wv = wid - V3_LWAVES;
#pragma unroll
        for (int q = 0; q < NU; ++q) {
            const int u = v3_unit(wv, q);            // slab = unit number 9 * product + tap (-1: this wave has no fourth unit)
            if (u >= 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {      // tile (a, b), register t: input channel 16a + 4 (lane >> 4) + t, output channel 16b + (lane & 15)
                    const int a = i >> 3, b = (i >> 2) & 1, t = i & 3;
                    red[u * 1024 + (16 * a + 4 * (lane >> 4) + t) * 32 + 16 * b + (lane & 15)] = acc[q][a][b][t];
                }
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < 9 * 1024; e += V3_THREADS) {
        const int tap = e >> 10, el = e & 1023;
        const float hh = red[tap * 1024 + el], hl = red[(9 + tap) * 1024 + el], lh = red[(18 + tap) * 1024 + el];
        outp[e] = (hh + (hl + lh) * 0x1p-11f) * inv_s;
    }
    __syncthreads();
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

hipError_t launch_wgrad_h2x(const WgradParams& p, hipStream_t stream)
{
    static PerDevice once_;
    hipError_t e = once_.once([]() {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_h2x_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS_BYTES);
    }, nullptr);
    if (e != hipSuccess) return e;
    if ((p.nparts & 7) && !(p.npairs > 0 && (p.nparts & 7) == 1)) return hipErrorInvalidValue;
    if (p.npairs < 0 || p.npairs > 16 || p.n_in > 5 || p.n_g > 5) return hipErrorInvalidValue;
    for (int s = 0; s < p.npairs; ++s)
        if ((int)((p.pair_j >> (4 * s)) & 15) >= p.n_in || (int)((p.pair_n >> (4 * s)) & 15) >= p.n_g) return hipErrorInvalidValue;
    for (int i = 0; i < p.n_in; ++i) if (!p.amax_x[i]) return hipErrorInvalidValue;   // operand scales: never guessed (fp16 range)
    for (int i = 0; i < p.n_g; ++i) if (!p.amax_g[i]) return hipErrorInvalidValue;
    // 32-bit byte offsets inside one batch slice of a plane (buffer loads); xsd_forward rejects such images with a message
    for (int i = 0; i < p.n_in; ++i) if ((long long)p.H * p.x[i].rs * 4 >= (1ll << 31)) return hipErrorInvalidValue;
    for (int i = 0; i < p.n_g; ++i) if ((long long)p.H * p.g[i].rs * 4 >= (1ll << 31)) return hipErrorInvalidValue;
    // pair lists with nparts = 8 m + 1: m * npairs workgroups per XCD for the full parts + ceil(npairs / 8) per XCD for the last part
    const int pair_grid = 8 * ((p.nparts >> 3) * p.npairs + ((p.nparts & 7) ? (p.npairs + 7) / 8 : 0));
    const dim3 g(p.npairs > 0 ? pair_grid : p.nparts * p.n_in * p.n_g), b(V3_THREADS);
    hipLaunchKernelGGL(wgrad_h2x_kernel, g, b, V3_LDS_BYTES, stream, p);
    return hipGetLastError();
}

} // namespace xsd
