// conv3x3_s3x.hip -- math mode 3 ("bf16x6"): the 3x3 conv (forward + input-gradient) over fp32 feature planes with
// fp32-CLASS arithmetic on the bf16 matrix cores; ROLE-SPLIT workgroup: 8 MFMA waves + 4 staging waves.
// Reference layers: nn.Conv2d(32k -> 32, 3, 1, 1) of rrdb_blocks.py:27-31, generator_rrdb.py:38-44,95,101 (fp32) and
// their autograd input-gradients.
//
// Arithmetic: exact 3-term bf16 split of both operands (xsd_split.h: x = hi + mid + lo, each term the round-to-nearest bf16
// of what the previous ones left), six bf16 MFMA products per fp32 product (hh, hm, mh, hl, lh, mm; dropped <= 2^-23
// relative), hi*hi in one accumulator and the five cross products in a second one, MFMA single-rounding accumulation.
//
// Why roles.  In a kernel whose waves all do everything in order (rounds 1-2 had one; git history) a wave that is blocked issuing a global load or a
// store, or that runs its share of the fp32 -> 3 x bf16 conversion, issues no MFMA meanwhile.  Here the two kinds of work
// live in different waves of the same SIMD, where the hardware overlaps them (MFMA and VALU/VMEM pipes are separate):
//   * waves 0..3 (one per SIMD) stage: buffer_load the next-but-one input half-tile and weight half-panel as fp32 into
//     registers (15 loads in flight per wave, hand-counted waits), split what arrived a half-step earlier and write it
//     to LDS: the input into the other of two buffers while the MFMA waves multiply the current one, the weights (kept
//     split in registers) between the two barriers that end a half-step (single 27,648-B buffer), which leaves only LDS
//     writes in that window;
//   * waves 4..11 (two per SIMD) multiply: tile rows 2w, 2w+1 of the 16 x 32 tile; their stream is ds_read_b128 + v_mfma
//     (63 + 108 per half-step, fragments reused across the two rows), plus the epilogue at the end of a tile whose
//     stores are dripped into the next half-step's walk.
// Three waves per SIMD -> 168 registers per wave.
// LDS images: a staged half-tile is the 18 x 34 halo of 16 channels, stored once per term as [halo pixel][16 x bf16] = 32 B
// per pixel in two 16-B slots (channels 0-7 | 8-15); slot s of halo column hx lies at ((s ^ ((hx >> 3) & 1)) << 4), so that
// the 32 pixels x 2 halves a ds_read_b128 fragment request touches spread over all banks for every tap offset dx.  A weight
// half-panel is [tap][term][64 lanes][8 x bf16] in MFMA fragment order (S3_WH_BYTES, xsd_kernels.h).  Global accesses are
// raw-buffer accesses against per-plane descriptors (base = the plane's batch slice, range = its bytes): lanes outside the
// image carry an offset that fails the range check and read zeros / store nothing.
// What made it pay (each measured with the in-kernel stamps of the diagnostic build): no packed-f32
// VALU beside the MFMAs (csrc/Makefile), an 22-instruction split, staging rounds without address arithmetic or
// branches, LDS writes that are ds_write and not flat stores, counted waits instead of hipcc's vmcnt(0), deferred stores.
#include <cstdlib>
#include <type_traits>
#include "xsd_kernels.h"
#include "xsd_split.h"

namespace xsd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) f32x4* x3_gload_p;
typedef __attribute__((address_space(1))) f32x4* x3_gstore_p;
__device__ __forceinline__ f32x4 x3_gload4(const float* p) { return *(x3_gload_p)p; }

constexpr int X3_MWAVES = 8;                        // MFMA waves
constexpr int X3_LWAVES = 4;                        // staging waves
constexpr int X3_THREADS = 64 * (X3_MWAVES + X3_LWAVES);   // 768
constexpr int X3_LT = 64 * X3_LWAVES;               // 256 staging threads
constexpr int X3_ROWS = 16;
constexpr int X3_PX = (X3_ROWS + 2) * HALO_W;       // 612 halo pixels
constexpr int X3_SINK = X3_PX * 32;                 // 19,584: each term image ends with a 512-B sink
constexpr int X3_XT = X3_SINK + 512;                // 20,096 B per term image
constexpr int X3_XB = 3 * X3_XT;                    // 60,288 B per input buffer
constexpr int X3_WOFF = 2 * X3_XB;                  // 120,576
constexpr int X3_WSINK = S3_WH_BYTES;               // sink behind the weight buffer (2048 + 512 B)
constexpr int X3_BIAS = X3_WOFF + S3_WH_BYTES + 3072; // 151,296
constexpr int X3_DESC = X3_BIAS + 5 * 32 * 4;       // 151,936
constexpr int X3_LDS_BYTES = X3_DESC + 16 * 8;      // 152,064
constexpr int X3_ROWB = HALO_W * 32;                // 1088
constexpr int X3_XSLOTS = X3_PX * 4;                // 2448 float4 slots of an input half-tile
constexpr int X3_XR = (X3_XSLOTS + X3_LT - 1) / X3_LT;     // 10
constexpr int X3_WSLOTS = 9 * 64 * 2;               // 1152 float4 slots of an fp32 half-panel
constexpr int X3_WR = (X3_WSLOTS + X3_LT - 1) / X3_LT;     // 5

__device__ __forceinline__ void x3_split4(const f32x4& a, u32x2& hi, u32x2& mid, u32x2& lo) { split3_f32x4(a, hi, mid, lo); }   // xsd_split.h

__global__ __launch_bounds__(X3_THREADS) void conv3x3_s3x_kernel(const ConvParams P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* w_lds = smem + X3_WOFF;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // Roles by wave id: the staging waves are waves 0..3, i.e. the OLDEST wave of each SIMD.  Vector issue on a SIMD is
    // arbitrated by age, and a pending MFMA of an older wave shuts the younger waves' VALU out: with the staging waves
    // youngest their 340 VALU instructions per half-step took ~9k cycles (measured with the phase stamps), i.e. they only
    // ran while the MFMA waves sat at a barrier.  The MFMA waves need 8 issue cycles of every 32.
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wid < X3_LWAVES;               // wave-uniform role
    const int lt = tid;                                // staging thread index 0..255 (staging waves only)
    const int wv = wid - X3_LWAVES;                    // MFMA wave index 0..7 -> tile rows 2wv, 2wv+1

    const int tilesY = (P.H + X3_ROWS - 1) / X3_ROWS;
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
        r.x0 = tx * TILE_W; r.y0 = (t % tilesY) * X3_ROWS; r.b = t / tilesY;
        return r;
    };
    struct Cur { int j, i, s2, k; };   // output chunk, input plane, channel half, tile ordinal
    auto succ = [&](Cur c) {
        c.s2 ^= 1;
        if (c.s2 == 0 && ++c.i == n_in) { c.i = 0; if (++c.j == n_out) { c.j = 0; ++c.k; } }
        return c;
    };

    // plane / panel descriptors and the bias through LDS
    unsigned long long* desc = reinterpret_cast<unsigned long long*>(smem + X3_DESC);
    if (tid < 5) {
        desc[2 * tid] = reinterpret_cast<unsigned long long>(P.in[tid].p);
        desc[2 * tid + 1] = (unsigned long long)P.in[tid].bs * 4ull;
        desc[10 + tid] = reinterpret_cast<unsigned long long>(P.wstep[tid]);
    }
    float* bias_lds = reinterpret_cast<float*>(smem + X3_BIAS);
    if (tid < 160) bias_lds[tid] = (P.bias && tid < 32 * n_out) ? P.bias[tid] : 0.f;
    __syncthreads();

    const float* zero = reinterpret_cast<const float*>(P.zero);
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    const int rs0 = P.in[0].rs, ps0 = P.in[0].ps;

    if (loader) {
        // ============================ staging waves ============================
        // Staging slots of this thread: slot(r) = r*256 + lt -> halo pixel p = r*64 + (lt >> 2), channel quad q = lt & 3.
        // Everything per-lane about a slot is computed ahead of the loop and kept in registers: its LDS byte offset (fixed)
        // and its byte offset inside the input plane of the tile being prefetched (rebuilt when the tile changes).  A round
        // of the loop is then: counted wait, 18-instruction split, three ds_write_b64, one buffer_load.
        //
        // Loads are BUFFER loads (128-bit descriptor in SGPRs + 32-bit lane offset): padding pixels, exhausted slots and the
        // rounds after the last half-step carry an offset / a descriptor length that fails the hardware range check, which
        // returns zeros without touching memory -- no zero page, no select, no 64-bit address arithmetic.
        // They are issued and waited for BY HAND (inline asm): hipcc's own bookkeeping puts one s_waitcnt vmcnt(0) in front
        // of the first conversion of a half-step, which halves the prefetch distance.  These waves execute no other
        // vector-memory instruction, so the count is exact: every load is issued right after the conversion of the round
        // whose register it refills, in the fixed order X0..X9, W0..W4; rounds are taken in pairs (four independent split
        // chains for the in-order wave), and when a pair's data is needed exactly 13 younger loads exist (14 for the
        // single last round) -- `s_waitcnt vmcnt(13)` keeps a whole half-step of loads in flight.  hipcc does not know these
        // registers are written asynchronously: the load takes its destination as an in/out operand (the new value lands
        // where the consumed one was), the wait statement too, and tools/check_async_loads.py (run by tests/test_isa.py)
        // verifies in the generated code that nothing else in the loop touches them and that no other vmem op exists.
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        constexpr int OOR = (int)0x80000000;          // lane offset that fails every range check
        int xlds[X3_XR], xoff[X3_XR];
        short hy_[X3_XR], hx_[X3_XR];
#pragma unroll
        for (int r = 0; r < X3_XR; ++r) {
            const int pp = r * 64 + (lt >> 2);
            const int hy = pp / HALO_W, hx = pp - hy * HALO_W;
            hy_[r] = (short)hy; hx_[r] = (short)hx;
            const int off = pp * 32 + ((((lt >> 1) & 1) ^ ((hx >> 3) & 1)) << 4) + (lt & 1) * 8;
            xlds[r] = (r * X3_LT + lt < X3_XSLOTS) ? off : X3_SINK + (lt & 63) * 8;
            xoff[r] = OOR;
        }
        auto tile_offsets = [&](const TileXY& T) {
#pragma unroll
            for (int r = 0; r < X3_XR; ++r) {
                const int gy = T.y0 - 1 + hy_[r], gx = T.x0 - 1 + hx_[r];
                const bool ok = (r * X3_LT + lt < X3_XSLOTS) && (gy >= 0) && (gy < P.H) && (gx >= 0) && (gx < P.W);
                xoff[r] = ok ? (gy * rs0 + gx * ps0 + (lt & 3) * 4) * 4 : OOR;
            }
        };
        const int woff = lt * 16;
        f32x4 pin[X3_XR] = {};
        f32x4 pw[X3_WR] = {};
        u32x2 wsh[X3_WR] = {}, wsm[X3_WR] = {}, wsl[X3_WR] = {};
        // descriptors are read from LDS (per lane) and made scalar again: uniform values belong in SGPRs
        auto uniform64 = [&](unsigned long long v) {
            const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v), hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
            return ((unsigned long long)hi << 32) | lo;
        };
        auto make_rsrc = [&](unsigned long long base, unsigned int bytes) {
            i32x4 d;
            d[0] = (int)(unsigned int)base; d[1] = (int)(unsigned int)((base >> 32) & 0xffffu);   // stride 0: raw buffer
            d[2] = (int)bytes; d[3] = 0x00020000;
            return d;
        };
#if defined(XSD_DIAG) && defined(XSD_ABL)   // timing experiments: a COMPILE-TIME constant (-DXSD_DIAG -DXSD_ABL=n builds; a run-time
                                                  // value puts the hand-counted loads and waits under branches hipcc cannot keep exact)
        constexpr int abl = XSD_ABL;     // 16: empty input descriptors (no input traffic); 1: no split (raw registers written);
                                      // 2: no input LDS writes; 4: no input loads and no counted waits at all
#else
        constexpr int abl = 0;
#endif
        const unsigned int plane_bytes = (unsigned int)P.H * (unsigned int)rs0 * 4u;
        auto x_rsrc = [&](const Cur& c, const TileXY& T, bool live) {
            const unsigned long long p = uniform64(desc[2 * c.i]), bs = uniform64(desc[2 * c.i + 1]);
            return make_rsrc(p + (unsigned long long)T.b * bs + c.s2 * 64, (live && !(abl & 16)) ? plane_bytes - c.s2 * 64 : 0u);
        };
        auto w_rsrc = [&](const Cur& c, bool live) {
            return make_rsrc(uniform64(desc[10 + c.j * n_in + c.i]) + c.s2 * (PANEL_FLOATS / 2) * 4, live ? X3_WSLOTS * 16u : 0u);
        };
        auto asm_load4 = [&](f32x4& dst, int off, const i32x4& rs) {
            asm volatile("buffer_load_dwordx4 %[d], %[o], %[r], 0 offen" : [d] "+v"(dst) : [o] "v"(off), [r] "s"(rs) : "memory");
        };
        // same, and the split of the consumed value is finished first (its results pass through the statement)
        auto asm_load4_after = [&](f32x4& dst, int off, const i32x4& rs, u32x2& a, u32x2& b, u32x2& c) {
            asm volatile("buffer_load_dwordx4 %[d], %[o], %[r], 0 offen" : [d] "+v"(dst), "+v"(a), "+v"(b), "+v"(c) : [o] "v"(off), [r] "s"(rs) : "memory");
        };
        auto asm_wait14 = [&](f32x4& v) { asm volatile("s_waitcnt vmcnt(14)" : "+v"(v) :: "memory"); };
        auto asm_wait13 = [&](f32x4& u, f32x4& v) { asm volatile("s_waitcnt vmcnt(13)" : "+v"(u), "+v"(v) :: "memory"); };
        auto load_x_round = [&](int r, const i32x4& rs) { asm_load4(pin[r], xoff[r], rs); };
        auto store_x_round = [&](int r, int xb) {   // xb: byte offset of the input buffer in LDS
            u32x2 hi, mid, lo;
            if (abl & 1) { hi[0] = __float_as_uint(pin[r][0]); hi[1] = __float_as_uint(pin[r][1]); mid = hi; lo[0] = __float_as_uint(pin[r][2]); lo[1] = __float_as_uint(pin[r][3]); }
            else x3_split4(pin[r], hi, mid, lo);
            char* d = smem + xb + xlds[r];
            if (abl & 2) { asm volatile("" :: "v"(hi), "v"(mid), "v"(lo), "v"(d)); return; }
            *reinterpret_cast<u32x2*>(d) = hi;
            *reinterpret_cast<u32x2*>(d + X3_XT) = mid;
            *reinterpret_cast<u32x2*>(d + 2 * X3_XT) = lo;
        };
        auto load_w_round = [&](int r, const i32x4& rs) { asm_load4_after(pw[r], woff + r * (X3_LT * 16), rs, wsh[r], wsm[r], wsl[r]); };
        auto store_w = [&]() {
#pragma unroll
            for (int r = 0; r < X3_WR; ++r) {
                const int s = r * X3_LT + lt;
                const int frag = s >> 7, ln = (s >> 1) & 63, sub = s & 1;   // frag = tap
                char* d = w_lds + (s < X3_WSLOTS ? frag * 3 * 1024 + ln * 16 + sub * 8 : X3_WSINK + lane * 8);
                *reinterpret_cast<u32x2*>(d) = wsh[r];
                *reinterpret_cast<u32x2*>(d + 1024) = wsm[r];
                *reinterpret_cast<u32x2*>(d + 2048) = wsl[r];
            }
        };

        // prologue: half-step 0 into LDS, half-step 1 into the staging registers
        Cur cur = {0, 0, 0, 0};
        TileXY tcur = tile_of(0);
        tile_offsets(tcur);
        {
            const i32x4 b0 = x_rsrc(cur, tcur, true), w0 = w_rsrc(cur, true);
#pragma unroll
            for (int r = 0; r < X3_XR; ++r) load_x_round(r, b0);
#pragma unroll
            for (int r = 0; r < X3_WR; ++r) load_w_round(r, w0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < X3_XR; ++r) { asm volatile("" : "+v"(pin[r])); store_x_round(r, 0); }
#pragma unroll
            for (int r = 0; r < X3_WR; ++r) { asm volatile("" : "+v"(pw[r])); x3_split4(pw[r], wsh[r], wsm[r], wsl[r]); }
            store_w();
        }
        Cur n1 = succ(cur);                 // same tile: a tile has at least two half-steps
        TileXY t1 = tcur;
        {
            const i32x4 b1 = x_rsrc(n1, t1, items > 1), w1 = w_rsrc(n1, items > 1);
#pragma unroll
            for (int r = 0; r < X3_XR; ++r) load_x_round(r, b1);
#pragma unroll
            for (int r = 0; r < X3_WR; ++r) load_w_round(r, w1);
        }
        Cur n2 = succ(n1);
        TileXY t2 = t1;
        if (items > 2 && n2.k != n1.k) { t2 = tile_of(n2.k); tile_offsets(t2); }
        lds_barrier();                                                                     // (P)
#ifdef XSD_DIAG   // staging-wave phase stamps, slots 8..12 (X rounds, W rounds, wait for barrier A, weight write, barrier B)
        unsigned long long lst[5] = {0, 0, 0, 0, 0};
        unsigned long long lt0 = __builtin_readcyclecounter();
        const bool lstamp = P.dbg != nullptr;
#define X3_LTICK(i) do { if (lstamp) { const unsigned long long t_ = __builtin_readcyclecounter(); lst[i] += t_ - lt0; lt0 = t_; } } while (0)
#else
#define X3_LTICK(i) do { } while (0)
#endif

#pragma unroll 1
        for (int it = 0; it < items; ++it) {
            const bool more1 = (it + 1 < items), more2 = (it + 2 < items);
            X3_LTICK(4);
            const i32x4 xrs = x_rsrc(n2, t2, more2), wrs = w_rsrc(n2, more2);
            const int xn = ((it + 1) & 1) * X3_XB;
            // input of half-step it+1: registers -> the other buffer; then refill each register with half-step it+2
#pragma unroll
            for (int r = 0; r < X3_XR; r += 2) {     // two rounds at a time: four independent split chains in flight
                if (!(abl & 4)) asm_wait13(pin[r], pin[r + 1]);
                store_x_round(r, xn);
                store_x_round(r + 1, xn);
                if (!(abl & 4)) { load_x_round(r, xrs); load_x_round(r + 1, xrs); }
                __builtin_amdgcn_sched_barrier(0);   // a pair at a time, in order (the wait counts depend on it)
            }
            X3_LTICK(0);
            // weights of half-step it+1: split in registers now, written when the MFMA waves are done with the buffer
#pragma unroll
            for (int r = 0; r + 1 < X3_WR; r += 2) {
                if (!(abl & 4)) asm_wait13(pw[r], pw[r + 1]);
                x3_split4(pw[r], wsh[r], wsm[r], wsl[r]);
                x3_split4(pw[r + 1], wsh[r + 1], wsm[r + 1], wsl[r + 1]);
                if (!(abl & 4)) { load_w_round(r, wrs); load_w_round(r + 1, wrs); }
                __builtin_amdgcn_sched_barrier(0);
            }
            static_assert(X3_XR % 2 == 0 && X3_WR % 2 == 1, "pairing of the staging rounds");
            if (!(abl & 4)) asm_wait14(pw[X3_WR - 1]);
            x3_split4(pw[X3_WR - 1], wsh[X3_WR - 1], wsm[X3_WR - 1], wsl[X3_WR - 1]);
            if (!(abl & 4)) load_w_round(X3_WR - 1, wrs);
            __builtin_amdgcn_sched_barrier(0);
            X3_LTICK(1);
            if (more1) {
                lds_barrier();                                                             // (A)
                X3_LTICK(2);
                store_w();
                X3_LTICK(3);
                lds_barrier();                                                             // (B)
            }
            n1 = n2; t1 = t2;
            n2 = succ(n2);
            if (it + 3 < items && n2.k != n1.k) { t2 = tile_of(n2.k); tile_offsets(t2); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef XSD_DIAG
        if (lstamp && lt == 0) {
#pragma unroll
            for (int q = 0; q < 5; ++q) atomicAdd(&P.dbg[8 + q], lst[q]);
        }
#endif
        return;
    }

    // ============================== MFMA waves ==============================
#ifndef X3S_MPRIO
#define X3S_MPRIO 1
#endif
#if X3S_MPRIO
    __builtin_amdgcn_s_setprio(X3S_MPRIO);   // the MFMA waves are the critical path (the staging wave waits ~5k of a half-step's ~9.9k cycles at
#endif                                       // the barrier); same device, alternating: 2.255 -> 2.241 ms per launch (-0.6 %), priority 2 the same
    f32x16 acc[2], accx[2];
    auto init_acc = [&](int j) {
        int hh = lane;                    // rebuilt from the lane id: held across the loop it is spilled
        asm volatile("" : "+v"(hh));
        hh >>= 5;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_lds + j * 32 + 8 * q + 4 * hh);
#pragma unroll
            for (int t = 0; t < 4; ++t) { acc[0][4 * q + t] = bv[t]; acc[1][4 * q + t] = bv[t]; accx[0][4 * q + t] = 0.f; accx[1][4 * q + t] = 0.f; }
        }
    };
    const char* wl = w_lds + lane * 16;
    // One half-step of a wave: 2 output rows x 32 pixels x 32 output channels, 9 taps x 16 input channels, 108 MFMAs.
    // The LDS read path is the co-limiter of this kernel (8 waves x 81 KB per half-step is the CU's whole 128 B/clk for
    // 5k cycles, measured with the MFMAs ablated), so fragments are reused in registers: the walk goes over the four INPUT
    // rows the two output rows touch; an input row's fragment (ir, dx) serves output row 0 with tap dy = ir and output row 1
    // with tap dy = ir - 1, and a tap's weight fragment is kept for the next input row.  63 ds_read_b128 instead of 81.
    // A finished tile's results wait in `pend` and are stored one float4 per lane at a time during the next half-step's
    // MFMA walk: eight waves storing their 8 KB at once is a 64-KB burst into a store path that takes ~10 B/clk per CU,
    // i.e. ~6k cycles with the matrix pipe idle.
    f32x4 pend[8];
    float* pend_dp[2];
    auto store_pending = [&](int c) { *(x3_gstore_p)(pend_dp[c >> 2] + 8 * (c & 3)) = pend[c]; };
    auto compute = [&](const char* xc, bool drip) {
        bf16x8 xf[2][3], wf[2][3];   // [slot][term]: 0 = hi, 1 = mid, 2 = lo; weight slot = dy & 1
        // fragment base offsets rebuilt per half-step from the lane id (a few VALU): held across the loop they get spilled,
        // and a scratch reload in front of the MFMAs is a vector-memory round trip
        int abase[3];
        {
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int hx = (ln & 31) + dx;
                abase[dx] = (wv * 2) * X3_ROWB + hx * 32 + (((ln >> 5) ^ ((hx >> 3) & 1)) << 4);
            }
        }
        auto load_w = [&](int tap, bf16x8 (&b)[3]) {
#ifdef XSD_EXP_NOREAD    // timing experiment: MFMAs on whatever the registers hold
            asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]));
#else
#pragma unroll
            for (int t = 0; t < 3; ++t) b[t] = *reinterpret_cast<const bf16x8*>(wl + (tap * 3 + t) * 1024);
#endif
        };
        auto load_x = [&](int ir, int dx, bf16x8 (&a)[3]) {
#ifdef XSD_EXP_NOREAD
            asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]));
#else
#pragma unroll
            for (int t = 0; t < 3; ++t) a[t] = *reinterpret_cast<const bf16x8*>(xc + t * X3_XT + abase[dx] + ir * X3_ROWB);
#endif
        };
        auto mac = [&](int r, const bf16x8 (&w)[3], const bf16x8 (&x)[3]) {
#ifdef XSD_EXP_NOMFMA    // timing experiment: fragment reads only
            asm volatile("" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(x[0]), "v"(x[1]), "v"(x[2]));
            return;
#endif
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[2], accx[r], 0, 0, 0);   // Wh * Xl
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], x[0], accx[r], 0, 0, 0);   // Wl * Xh
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[1], accx[r], 0, 0, 0);   // Wm * Xm
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[0], acc[r], 0, 0, 0);     // Wh * Xh
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[1], accx[r], 0, 0, 0);   // Wh * Xm
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[0], accx[r], 0, 0, 0);   // Wm * Xh
        };
        load_x(0, 0, xf[0]);
        load_w(0, wf[0]);
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            const int ir = s & 3;                                 // column dx = s >> 2
            const int irn = (s + 1) & 3, dxn = (s + 1) >> 2;
            if (s + 1 < 12) load_x(irn, dxn, xf[(s + 1) & 1]);
            if (ir >= 1) mac(1, wf[(ir - 1) & 1], xf[s & 1]);      // output row 1, tap (dy = ir - 1, dx)
            // two weight slots (dy & 1): the slot row 1 has just finished with takes the fragment the next step needs
            if (s + 1 < 12 && irn <= 2) load_w(irn * 3 + dxn, wf[irn & 1]);
            if (ir <= 2) mac(0, wf[ir & 1], xf[s & 1]);            // output row 0, tap (dy = ir, dx)
            if (drip && s >= 1 && s <= 8) store_pending(s - 1);
        }
    };

    // Epilogue over fp32 planes (straight-line operand variants; lanes outside the image read the zero page and write a
    // trash page): each lane owns one pixel and 16 channels as four float4 groups.
    auto epilogue_v = [&](const OutDesc& o, const TileXY& T, auto has_acc, auto has_e1, auto has_e2, auto has_e3, auto has_mask, auto generic, auto has_bits, auto has_bout) {
        // has_bits / has_bout: compact lrelu' masks, one 16-bit word per lane and row (conv3x3_h2x.hip)
        float* dst = o.p + (long long)T.b * o.bs;
        const long long sb = (long long)T.b * P.std_bs;
        // lane coordinates rebuilt from the lane id here: held across the K-loop they are spilled, and a scratch reload in the
        // epilogue is a full memory round trip with the matrix pipe idle (conv3x3_h2x.hip)
        unsigned int all = ~0u;
        asm volatile("" : "+s"(all));
        const int ln = (int)__builtin_amdgcn_mbcnt_hi(all, __builtin_amdgcn_mbcnt_lo(all, 0u));
        const int h = ln >> 5, l31 = ln & 31;
        float* const trash = const_cast<float*>(zero) + 64 + 4 * h;
        const int x = T.x0 + l31;
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
            const int widx = (((int)T.b * P.H + y) * P.W + x) * 2 + h;     // (pixel, lane half) -> 16-bit word (B*H*W*2 < 2^31)
            unsigned int mbits = 0, obits = 0;
            if constexpr (decltype(has_bits)::value) mbits = o.bits_in[valid ? widx : 0];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (decltype(has_acc)::value) va[q] = x3_gload4(pa + 8 * q);
                if constexpr (decltype(has_e1)::value) v1[q] = x3_gload4(p1 + 8 * q);
                if constexpr (decltype(has_e2)::value) v2[q] = x3_gload4(p2 + 8 * q);
                if constexpr (decltype(has_e3)::value) v3[q] = x3_gload4(p3 + 8 * q);
                if constexpr (decltype(has_mask)::value && !decltype(has_bits)::value) vm[q] = x3_gload4(pm + 8 * q);
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
                if constexpr (decltype(has_bits)::value) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = ((mbits >> (4 * q + t)) & 1u) ? v[t] : v[t] * msl;
                } else if constexpr (decltype(has_mask)::value) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = vm[q][t] > 0.f ? v[t] : v[t] * msl;
                }
                if constexpr (decltype(has_bout)::value) {      // this plane's own lrelu' mask for the backward pass
#pragma unroll
                    for (int t = 0; t < 4; ++t) obits |= (v[t] > 0.f ? 1u : 0u) << (4 * q + t);
                }
                pend[4 * r + q] = v;
            }
            pend_dp[r] = dp;
            if constexpr (decltype(has_bout)::value) { if (valid) o.bits_out[widx] = (unsigned short)obits; }
        }
    };
    auto epilogue = [&](int j, const TileXY& T) {
        const OutDesc o = P.out[j];
        using Y = std::true_type; using N = std::false_type;
        const int kind = (o.accumulate ? 1 : 0) | (o.e1 ? 2 : 0) | (o.e2 ? 4 : 0) | (o.e3 ? 8 : 0) | (o.mask ? 16 : 0);
        switch (kind | ((o.mask && o.bits_in) ? 32 : 0) | (o.bits_out ? 64 : 0)) {
        case 0: epilogue_v(o, T, N{}, N{}, N{}, N{}, N{}, N{}, N{}, N{}); break;
        case 64: epilogue_v(o, T, N{}, N{}, N{}, N{}, N{}, N{}, N{}, Y{}); break;
        case 2: epilogue_v(o, T, N{}, Y{}, N{}, N{}, N{}, N{}, N{}, N{}); break;
        case 6: epilogue_v(o, T, N{}, Y{}, Y{}, N{}, N{}, N{}, N{}, N{}); break;
        case 14: epilogue_v(o, T, N{}, Y{}, Y{}, Y{}, N{}, N{}, N{}, N{}); break;
        case 16: epilogue_v(o, T, N{}, N{}, N{}, N{}, Y{}, N{}, N{}, N{}); break;
        case 48: epilogue_v(o, T, N{}, N{}, N{}, N{}, Y{}, N{}, Y{}, N{}); break;
        default: epilogue_v(o, T, Y{}, Y{}, Y{}, Y{}, Y{}, Y{}, N{}, N{}); break;   // any other combination (none in the engine's plans)
        }
        // every load of the epilogue (operands, register reloads) has landed before the MFMA walk starts: its deferred
        // stores then need no vector-memory waits (hipcc would otherwise put `vmcnt(1)` in front of each, i.e. wait for
        // the store before the previous one)
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), gfx9 encoding
    };

#ifdef XSD_DIAG   // phase stamps (diagnostic library variant only; tools/stamps.py): accumulated shader cycles per phase
    unsigned long long st[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz: in-kernel clock = stamped cycles / this
    unsigned long long t0 = __builtin_readcyclecounter();
    const bool stamp = P.dbg != nullptr;
#define X3_TICK(i) do { if (stamp) { const unsigned long long t_ = __builtin_readcyclecounter(); st[i] += t_ - t0; t0 = t_; } } while (0)
#else
#define X3_TICK(i) do { } while (0)
#endif

    Cur cur = {0, 0, 0, 0};
    TileXY tcur = tile_of(0);
    bool pending = false;     // a finished tile's results are waiting in `pend`
    lds_barrier();                                                                         // (P)
    X3_TICK(0);
#pragma unroll 1
    for (int it = 0; it < items; ++it) {
        const bool more1 = (it + 1 < items);
        X3_TICK(1);
        if (cur.i == 0 && cur.s2 == 0) init_acc(cur.j);
        compute(smem + (it & 1) * X3_XB, pending);
        pending = false;
        X3_TICK(2);
        if (cur.i == n_in - 1 && cur.s2 == 1) {
            epilogue(cur.j, tcur);
            pending = more1;
            if (!more1) {
#pragma unroll
                for (int c = 0; c < 8; ++c) store_pending(c);
            }
        }
        X3_TICK(3);
        if (more1) {
            lds_barrier();                                                                 // (A)
            X3_TICK(4);
            lds_barrier();                                                                 // (B)
            X3_TICK(5);
        }
        const Cur nx = succ(cur);
        if (nx.k != cur.k && more1) tcur = tile_of(nx.k);
        cur = nx;
    }
#ifdef XSD_DIAG
    if (stamp && tid == 64 * X3_LWAVES) {
#pragma unroll
        for (int q = 0; q < 6; ++q) atomicAdd(&P.dbg[q], st[q]);
        atomicAdd(&P.dbg[6], (unsigned long long)items);
        atomicAdd(&P.dbg[7], __builtin_amdgcn_s_memrealtime() - rt0);
    }
    if (stamp && tid == 64 * (X3_LWAVES + X3_MWAVES - 1)) {   // the youngest MFMA wave: slots 13..15 = MFMA walk, epilogue, barrier A
        atomicAdd(&P.dbg[13], st[2]); atomicAdd(&P.dbg[14], st[3]); atomicAdd(&P.dbg[15], st[4]);
    }
#endif
}

static PerDevice g_once;

hipError_t launch_conv3x3_s3x(const ConvParams& p, hipStream_t stream)
{
    int ncu = 256;
    hipError_t e = g_once.once([]() {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_s3x_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS_BYTES);
    }, &ncu);
    if (e != hipSuccess) return e;
    if (p.n_in < 1 || p.n_out < 1 || p.n_in * p.n_out > 5 || !p.zero) return hipErrorInvalidValue;
    for (int j = 0; j < p.n_out; ++j)       // deferred stores: 32-bit BYTE offsets against a descriptor of H * rs * 4 bytes
        if ((long long)p.H * p.out[j].rs * 4 >= (1ll << 31)) return hipErrorInvalidValue;
    for (int i = 0; i < p.n_in; ++i)        // the staging waves take every input's lane offsets and range from in[0]'s strides
        if (p.in[i].rs != p.in[0].rs || p.in[i].ps != p.in[0].ps) return hipErrorInvalidValue;
    if ((long long)p.H * p.in[0].rs * 4 >= (1ll << 31)) return hipErrorInvalidValue;
    const int tilesY = (p.H + X3_ROWS - 1) / X3_ROWS;
    const int ntiles = p.B * p.tilesX * tilesY;
    if (ntiles <= 0) return hipSuccess;
    const dim3 g(persistent_grid(ntiles, ncu)), b(X3_THREADS);
    hipLaunchKernelGGL(conv3x3_s3x_kernel, g, b, X3_LDS_BYTES, stream, p);
    return hipGetLastError();
}

} // namespace xsd
