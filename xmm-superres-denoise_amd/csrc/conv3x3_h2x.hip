// conv3x3_h2x.hip -- math mode 4 ("f16x3"): the 3x3 conv (forward + input-gradient) over fp32 feature planes with
// fp32-CLASS arithmetic on the fp16 matrix cores at HALF the matrix work of mode 3; role-split workgroup as in
// conv3x3_s3x.hip (4 staging + 8 MFMA waves; read that file for the structure, the counted prefetch and the LDS images).
// Reference layers: nn.Conv2d(32k -> 32, 3, 1, 1) of rrdb_blocks.py:27-31, generator_rrdb.py:38-44,95,101 (fp32) and
// their autograd input-gradients.
//
// Arithmetic.  Each operand tensor is scaled by a power of two chosen from its max |x| (written by the kernel that produced
// it, OutDesc::amax / plane_amax_kernel; the weights' by the pack kernel) so that it fits the fp16 range with full
// precision, and split into two fp16 terms x*s = h + l*2^-11 (22-23 significant bits, xsd_split.h).  A product is
// h*h (first accumulator) + (h*l + l*h) (second accumulator, weighted 2^-11 in the epilogue): three
// v_mfma_f32_32x32x16_f16 instead of the six bf16 ones of mode 3; the MFMA sums its 16 products and the fp32 accumulator
// exactly and rounds once (tools/mfma_probe_f16.hip).  The epilogue undoes the scales (exact) and reports max |result| of
// the plane it writes.  Dropped: l*l (2^-22 relative) and the operands' last 1-2 bits; measured against float64 in
// tests/test_hip_precision.py next to mode 3, the exact-fp32 mode and torch's fp32.
#include <cstdlib>
#include <type_traits>
#include "xsd_kernels.h"
#include "xsd_split.h"

namespace xsd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
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
constexpr int X3_XT = 20 * 1024;                    // 20,480 B per term image (sink included), 1-KiB aligned
static_assert(X3_XT >= X3_SINK + 512, "sink");
constexpr int X3_XB = 2 * X3_XT;                    // 40,960 B per input buffer (two term images)
constexpr int X3_WSINK = H2_WH_BYTES;               // sink behind each weight buffer (2 KiB: the two LDS-DMA pieces of the last
                                                    // round that have no slots)
// LDS layout: two input buffers, two weight buffers of 18,432 + 2,048 B (a half-step ends with ONE barrier; the three-term
// images of mode 3 leave no room for the second), bias, descriptors.
struct X3L {
    static constexpr int NXB = 2;
    static constexpr int WB = H2_WH_BYTES + 2048;
    static constexpr int WOFF = NXB * X3_XB;            // 81,920
    static constexpr int BIAS = WOFF + 2 * WB;          // 122,880
    static constexpr int DESC = BIAS + 5 * 32 * 4;
    static constexpr int BYTES = DESC + 16 * 8;         // 123,648
};
static_assert(X3L::BYTES <= 160 * 1024, "LDS");
constexpr int X3_ROWB = HALO_W * 32;                // 1088
constexpr int X3_XSLOTS = X3_PX * 4;                // 2448 float4 slots of an input half-tile
constexpr int X3_XR = (X3_XSLOTS + X3_LT - 1) / X3_LT;     // 10
constexpr int X3_WSLOTS = 9 * 64 * 2;               // 1152 float4 slots of an fp32 half-panel
constexpr int X3_WR = (X3_WSLOTS + X3_LT - 1) / X3_LT;     // 5 LDS-DMA pieces per staging wave and half-step

__device__ __forceinline__ void x3_split4(const f32x4& a, float s, u32x2& hi, u32x2& lo) { split2_f16x4(a, s, hi, lo); }   // xsd_split.h

// Epilogue kind of a launch: bit 0 accumulate, 1 e1, 2 e2, 3 e3, 4 mask plane, 5 mask as compact bits, 6 compact-mask output (the
// case labels of `epilogue` below).  KIND < 0: the catch-all instance that decides per output chunk at run time
// (conv3x3_h2x_kernel); KIND >= 0: an instance that holds ONE epilogue variant and, for the kinds whose launches never have a
// LeakyReLU of their own (operand and masked kinds: conv5, trunk, every input-gradient launch), no LeakyReLU instructions
// either -- its own register allocation, chosen by the launcher when every output chunk of the launch is of that kind.
constexpr bool x3_kind_has_lrelu(int kind) { return kind < 0 || kind == 0 || kind == 64; }

template <int KIND>
__device__ __forceinline__ void conv3x3_h2x_body(const ConvParams& P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef X3L L;
    char* w_lds = smem + L::WOFF;

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

    // Prologue (round 6).  At the reference's own batch sizes (1 / 4 / 8: a 416 x 416 image is 338 tiles on 256 workgroups) a launch
    // is 4 - 20 half-steps long and what precedes the first MFMA is a tenth of it.  Until round 6 that was FOUR dependent memory
    // round trips behind each other -- descriptors + bias into LDS and a barrier; the max-|x| slots one by one (a pointer test, a
    // scalar load and a wait each: the ISA showed up to six in a chain); then the first input tile and weight half-panel -- 6.5 us
    // from kernel entry to the first barrier (tools/stamps_small.py).  Now everything that needs only the kernel arguments is
    // requested at once: the slots as six independent scalar loads (absent inputs read the zero page), and the staging waves
    // issue half-step 0's input loads and weight DMA from P.in[0] / P.wstep[0] directly BEFORE they fill the descriptor table.
    const float* zero = reinterpret_cast<const float*>(P.zero);

    // plane / panel descriptors and the bias through LDS (written by threads of the staging waves, read after barrier A)
    unsigned long long* desc = reinterpret_cast<unsigned long long*>(smem + L::DESC);
    float* bias_lds = reinterpret_cast<float*>(smem + L::BIAS);
    auto fill_tables = [&]() {
        if (tid < 5) {
            desc[2 * tid] = reinterpret_cast<unsigned long long>(P.in[tid].p);
            desc[2 * tid + 1] = (unsigned long long)P.in[tid].bs * 4ull;
            desc[10 + tid] = reinterpret_cast<unsigned long long>(P.wstep[tid]);
        }
        if (tid < 160) bias_lds[tid] = (P.bias && tid < 32 * n_out) ? P.bias[tid] : 0.f;
        if (tid == 0) desc[15] = 0ull;      // the workgroup's max-|x| combiner: [31:0] running maximum (float bits), [63:32] MFMA waves arrived
    };
    static_assert(160 <= X3_LT, "the table-filling threads belong to the staging waves");

    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    const int rs0 = P.in[0].rs, ps0 = P.in[0].ps;

    // operand scales (powers of two) from the max |x| of the input planes and of the weight panels; every wave computes the
    // same values from the same slots: six INDEPENDENT scalar loads (absent inputs read the zero page; the launcher rejects a
    // missing slot of a live input), one wait.  The staging waves call this with half-step 0's vector loads already in flight.
    // (Loading the slots at kernel entry and reducing here kept six more SGPRs live across the staging set-up: the catch-all
    // instance spilled a register to scratch -- a scratch access is a vector-memory operation the counted waits do not know.)
    float sx, sw, inv_sx, inv_sw, inv_s;
    auto operand_scales = [&]() {
        float a_in[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {           // static indices: a runtime index into the kernel arguments costs a scratch copy
            const float* ap = (i < n_in && P.amax_in[i]) ? P.amax_in[i] : zero;
            a_in[i] = *ap;
        }
        const float a_w = *(P.amax_w ? P.amax_w : zero);
        float ax = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i) ax = a_in[i] > ax ? a_in[i] : ax;
        ax = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ax)));
        sx = scale_for_amax(ax, inv_sx);
        sw = scale_for_amax(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a_w))), inv_sw);
        inv_s = inv_sx * inv_sw;      // undoes both scales in the epilogue (exact)
    };

    if (loader) {
        // ============================ staging waves ============================
#ifdef X3_LPRIO
        __builtin_amdgcn_s_setprio(X3_LPRIO);
#endif
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        constexpr int OOR = (int)0x80000000;          // lane offset that fails every range check
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
        auto w_rsrc = [&](const Cur& c, bool live) {
            return make_rsrc(uniform64(desc[10 + c.j * n_in + c.i]) + c.s2 * (PANEL_FLOATS / 2) * 4, live ? X3_WSLOTS * 16u : 0u);
        };
        // Weights by LDS-DMA: the pre-split half-panel is a lane-linear image, so `buffer_load_dwordx4 ... lds` copies it
        // global -> LDS with no register, no ds_write and no address arithmetic (lane offset 16 * lane, the round and the
        // wave in the scalar offset, the wave's 1-KiB destination in M0 -- written in the same statement, hipcc owns M0).
        // In the last round only waves 0 and 1 have slots; the other two land in the sink behind the buffer.
        const unsigned int lds0 = __builtin_amdgcn_readfirstlane((unsigned int)(unsigned long long)(__attribute__((address_space(3))) char*)smem);
        auto dma_w_round = [&](int r, const i32x4& rs, int wb, bool on = true) {
            unsigned int all = ~0u;
            asm volatile("" : "+s"(all));
            const int voff = 16 * (int)__builtin_amdgcn_mbcnt_hi(all, __builtin_amdgcn_mbcnt_lo(all, 0u));
            const int soff = r * (X3_LT * 16) + wid * 1024;
            const bool live = on && ((r + 1) * X3_LT <= X3_WSLOTS || r * X3_LT + wid * 64 < X3_WSLOTS);     // wave-uniform
            const unsigned int dst = lds0 + L::WOFF + wb + (live ? soff : X3_WSINK + (wid & 1) * 1024);
            asm volatile("s_mov_b32 m0, %[l]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[o], %[r], %[s] offen lds"
                         :: [o] "v"(voff), [r] "s"(rs), [s] "s"(soff), [l] "s"(dst) : "memory");
        };
        // Staging slots of this thread: slot(r) = r*256 + lt -> halo pixel p = r*64 + (lt >> 2), channel quad q = lt & 3.
        // Everything per-lane about a slot is computed ahead of the loop and kept in registers: its LDS byte offset (fixed)
        // and its byte offset inside the input plane of the tile being prefetched (rebuilt when the tile changes).  A round
        // of the loop is then: counted wait, 12-instruction split, two ds_write_b64, one buffer_load.
        //
        // Loads are BUFFER loads (128-bit descriptor in SGPRs + 32-bit lane offset): padding pixels, exhausted slots and the
        // rounds after the last half-step carry an offset / a descriptor length that fails the hardware range check, which
        // returns zeros without touching memory -- no zero page, no select, no 64-bit address arithmetic.
        // They are issued and waited for BY HAND (inline asm): hipcc's own bookkeeping puts one s_waitcnt vmcnt(0) in front
        // of the first conversion of a half-step, which halves the prefetch distance.  These waves execute no other
        // vector-memory instruction, so the count is exact: a pass issues the five LDS-DMA pieces of the next half-step's
        // weights (below), then every input load right after the conversion of the round whose register it refills, in the
        // fixed order X0..X9; rounds are taken in pairs (four independent split chains for the in-order wave), and when a
        // pair's data is needed exactly 13 younger operations exist -- `s_waitcnt vmcnt(13)` keeps a whole half-step of
        // input loads in flight.  hipcc does not know these
        // registers are written asynchronously: the load takes its destination as an in/out operand (the new value lands
        // where the consumed one was), the wait statement too, and tools/check_async_loads.py (run by tests/test_isa.py)
        // verifies in the generated code that nothing else in the loop touches them and that no other vmem op exists.
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
        f32x4 pin[X3_XR] = {};
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
        auto asm_load4 = [&](f32x4& dst, int off, const i32x4& rs) {
            asm volatile("buffer_load_dwordx4 %[d], %[o], %[r], 0 offen" : [d] "+v"(dst) : [o] "v"(off), [r] "s"(rs) : "memory");
        };
        auto asm_wait13 = [&](f32x4& u, f32x4& v) { asm volatile("s_waitcnt vmcnt(13)" : "+v"(u), "+v"(v) :: "memory"); };
        auto load_x_round = [&](int r, const i32x4& rs) { asm_load4(pin[r], xoff[r], rs); };
        auto store_x_round = [&](int r, int xb) {   // xb: byte offset of the input buffer in LDS
            u32x2 hi, lo;
            if (abl & 1) { hi[0] = __float_as_uint(pin[r][0]); hi[1] = __float_as_uint(pin[r][1]); lo[0] = __float_as_uint(pin[r][2]); lo[1] = __float_as_uint(pin[r][3]); }
            else x3_split4(pin[r], sx, hi, lo);
            char* d = smem + xb + xlds[r];
            if (abl & 2) { asm volatile("" :: "v"(hi), "v"(lo), "v"(d)); return; }
            *reinterpret_cast<u32x2*>(d) = hi;
            *reinterpret_cast<u32x2*>(d + X3_XT) = lo;
        };
        // prologue: half-step 0 into LDS, half-step 1 into the staging registers.  Half-step 0 is (chunk 0, plane 0, half 0): its
        // descriptors come straight from the kernel arguments (static indices), so its loads are in flight before the tables exist
        Cur cur = {0, 0, 0, 0};
        TileXY tcur = tile_of(0);
        tile_offsets(tcur);
        {
            const i32x4 b0 = make_rsrc(uniform64(reinterpret_cast<unsigned long long>(P.in[0].p) + (unsigned long long)tcur.b * ((unsigned long long)P.in[0].bs * 4ull)),
                                       (abl & 16) ? 0u : plane_bytes);
            const i32x4 w0 = make_rsrc(uniform64(reinterpret_cast<unsigned long long>(P.wstep[0])), X3_WSLOTS * 16u);
#pragma unroll
            for (int r = 0; r < X3_XR; ++r) load_x_round(r, b0);
#pragma unroll
            for (int r = 0; r < X3_WR; ++r) dma_w_round(r, w0, 0);
            operand_scales();
            fill_tables();
            lds_barrier();                                                                 // (A) tables visible
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < X3_XR; ++r) { asm volatile("" : "+v"(pin[r])); store_x_round(r, 0); }
        }
        Cur n1 = succ(cur);                 // same tile: a tile has at least two half-steps
        TileXY t1 = tcur;
        {
            const i32x4 b1 = x_rsrc(n1, t1, items > 1);
#pragma unroll
            for (int r = 0; r < X3_XR; ++r) load_x_round(r, b1);
        }
        Cur n2 = succ(n1);
        TileXY t2 = t1;
        if (items > 2 && n2.k != n1.k) { t2 = tile_of(n2.k); tile_offsets(t2); }
        lds_barrier();                                                                     // (P)
#ifdef XSD_DIAG   // staging-wave phase stamps, slots 8..12: [8] descriptors + weight DMA issue + X rounds, [9] wait for the DMA pieces, [10] barrier, [11] of [8]: inside the counted data waits, [12] cursor / tile offsets of the next half-steps (between the barrier and the loop top)
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
            const int xn = ((it + 1) & 1) * X3_XB;
            const int wn = ((it + 1) & 1) * L::WB;
            // weights of half-step it+1 straight into the other weight buffer (free since the barrier that ended it-1): five
            // DMA pieces, issued first, so that they are OLDER than this pass's ten input loads -- `vmcnt(10)` before the
            // barrier retires them and leaves the input loads of half-step it+2 in flight; the pair waits below see
            // 8 - 2k older input loads + 5 pieces + 2k refills = 13 younger operations, as in the register scheme
            {
#ifdef X3_WRES
                // Launches with ONE weight panel (one input plane, one output chunk: conv1 of every dense block, the trunk conv and
                // their input-gradient counterparts): the two half-panels of a tile are the two half-panels of every tile, and
                // half-step `it` reads buffer it & 1 = its channel half -- so after the first two half-steps both buffers hold what
                // every later half-step needs, and the pieces go out with an empty descriptor into the sink (same instruction
                // stream, same counted waits, no traffic: 18 KB of the 58 KB a CU takes in per half-step)
                const bool wlive = more1 && !(n_in * n_out == 1 && it >= 1);
#else
                const bool wlive = more1;
#endif
                const i32x4 wrs1 = w_rsrc(n1, wlive);
                if (!(abl & 4)) {
#pragma unroll
                    for (int r = 0; r < X3_WR; ++r) dma_w_round(r, wrs1, wn, wlive);
                }
            }
            const i32x4 xrs = x_rsrc(n2, t2, more2);
            // input of half-step it+1: registers -> the other buffer; then refill each register with half-step it+2
#pragma unroll
            for (int r = 0; r < X3_XR; r += 2) {     // two rounds at a time: four independent split chains in flight
#ifdef XSD_DIAG   // slot 11: cycles the staging wave spends in its counted waits, i.e. waiting for load data
                unsigned long long w0_ = 0;
                if (lstamp) w0_ = __builtin_readcyclecounter();
#endif
                if (!(abl & 4)) asm_wait13(pin[r], pin[r + 1]);
#ifdef XSD_DIAG
                if (lstamp) lst[3] += __builtin_readcyclecounter() - w0_;
#endif
                store_x_round(r, xn);
                store_x_round(r + 1, xn);
                if (!(abl & 4)) { load_x_round(r, xrs); load_x_round(r + 1, xrs); }
                __builtin_amdgcn_sched_barrier(0);   // a pair at a time, in order (the wait counts depend on it)
            }
            X3_LTICK(0);
            if (!(abl & 4)) asm volatile("s_waitcnt vmcnt(%[n])" :: [n] "n"(X3_XR) : "memory");   // the DMA pieces have landed
            X3_LTICK(1);
            lds_barrier();                                                                 // end of half-step `it`
            X3_LTICK(2);
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
#ifdef XSD_DIAG   // the stamps' origin is kernel entry (slot 0 = entry -> barrier P, both prologue barriers inside)
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz: in-kernel clock = stamped cycles / this
    unsigned long long t0 = __builtin_readcyclecounter();
#endif
    operand_scales();
    lds_barrier();                                                                         // (A) tables visible
#ifndef X3_MPRIO
#define X3_MPRIO 2
#endif
    __builtin_amdgcn_s_setprio(X3_MPRIO);   // the MFMA waves are the critical path, the staging waves have ~30 % slack (A/B on one device: -0.9 %)
    f32x16 acc[2], accx[2];
    auto init_acc = [&](int j) {
        int hh = lane;                    // rebuilt from the lane id: held across the loop it is spilled
        asm volatile("" : "+v"(hh));
        hh >>= 5;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_lds + j * 32 + 8 * q + 4 * hh);
#pragma unroll
            for (int t = 0; t < 4; ++t) {   // the accumulators are in scaled units: sum (sx x)(sw w) = sx sw * (true sum)
                const float b = bv[t] * (sx * sw);
                acc[0][4 * q + t] = b; acc[1][4 * q + t] = b; accx[0][4 * q + t] = 0.f; accx[1][4 * q + t] = 0.f;
            }
        }
    };
    // One half-step of a wave: 2 output rows x 32 pixels x 32 output channels, 9 taps x 16 input channels, 54 MFMAs.
    // The walk goes over the four INPUT rows the two output rows touch: an input row's fragment (ir, dx) serves output row 0
    // with tap dy = ir and output row 1 with tap dy = ir - 1, and a tap's weight fragment is kept for the next input row
    // (42 ds_read_b128 per half-step).
    // A finished tile's results wait in `pend` and are stored one float4 per lane at a time during the next half-step's
    // MFMA walk: eight waves storing their 8 KB at once is a 64-KB burst into a store path that takes ~10 B/clk per CU,
    // i.e. ~6k cycles with the matrix pipe idle.
    // (buffer stores: a 32-bit lane offset per row against a wave-uniform descriptor of the output plane's batch slice; lanes
    // outside the image carry an offset that fails the range check and are dropped by the hardware -- two registers instead
    // of two 64-bit pointers that hipcc spills, and no trash page)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    float run_max = 0.f;      // max |stored in-image value| over everything this lane has produced in this launch
    f32x4 pend[8];
    int pend_off[2];
    unsigned int pend_lo = 0, pend_hi = 0;     // base address and byte size of the output plane's batch slice
    int pend_nb = 0;
    auto store_pending = [&](int c) {
        // descriptor rebuilt from readfirstlane'd words at the use: carried across the loop as a descriptor, hipcc cannot
        // prove it wave-uniform and wraps every store in a waterfall loop
        const unsigned long long d = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane(pend_hi) << 32) |
                                     (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane(pend_lo);   // (the builtin returns int: no sign extension)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(d), 0, __builtin_amdgcn_readfirstlane(pend_nb), 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pend[c]), rs, pend_off[c >> 2] + 32 * (c & 3), 0, 0);
    };
    auto compute = [&](const char* xc, int wpar, bool drip) {
        f16x8 xf[3][2], wf[3][2];   // [slot][term]: 0 = hi, 1 = lo (scaled 2^11)
        // fragment base offsets rebuilt per half-step from the lane id (a few VALU): held across the loop they get spilled,
        // and a scratch reload in front of the MFMAs is a vector-memory round trip
        int abase[3];
        // the lane id is recomputed (two mbcnt on a mask hipcc cannot see through) rather than kept in a register across the
        // loop, where it gets spilled
        unsigned int all = ~0u;
        asm volatile("" : "+s"(all));
        const int ln = (int)__builtin_amdgcn_mbcnt_hi(all, __builtin_amdgcn_mbcnt_lo(all, 0u));
        const char* wl = w_lds + wpar * L::WB + ln * 16;
        {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int hx = (ln & 31) + dx;
                abase[dx] = (wv * 2) * X3_ROWB + hx * 32 + (((ln >> 5) ^ ((hx >> 3) & 1)) << 4);
            }
        }
        auto load_w = [&](int tap, f16x8 (&b)[2]) {
#pragma unroll
            for (int t = 0; t < 2; ++t) b[t] = *reinterpret_cast<const f16x8*>(wl + (tap * 2 + t) * 1024);
        };
        auto load_x = [&](int ir, int dx, f16x8 (&a)[2]) {
#pragma unroll
            for (int t = 0; t < 2; ++t) a[t] = *reinterpret_cast<const f16x8*>(xc + t * X3_XT + abase[dx] + ir * X3_ROWB);
        };
        auto mac = [&](int r, const f16x8 (&w)[2], const f16x8 (&x)[2]) {
#ifdef X3_NOMFMA   // energy experiment (tools/power_table.sh): the whole kernel but its matrix instructions -- results are garbage
            asm volatile("" :: "v"(w[0]), "v"(w[1]), "v"(x[0]), "v"(x[1]));
            return;
#endif
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0], x[1], accx[r], 0, 0, 0);   // Wh * Xl   } weighted 2^-11
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[1], x[0], accx[r], 0, 0, 0);   // Wl * Xh   } in the epilogue
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0], x[0], acc[r], 0, 0, 0);     // Wh * Xh
        };
        // Software pipeline over the 12 (column, input row) steps, TWO steps deep: the fragments of step s+2 are requested
        // before the MFMAs of step s (scheduling barriers keep hipcc from sinking the reads next to their first use).  A
        // step has only 3 or 6 MFMAs (96 / 192 cycles) here, one step of cover would expose the LDS latency every step.
        // Three slots each: an input fragment lives 3 steps; a weight fragment (dy, dx), numbered n = 3 dx + dy in the order
        // of first use, is requested two steps before its step 4 dx + dy and used there (row 0) and in the next step
        // (row 1) -- any four consecutive n straddle a column change, i.e. an extra step, so n % 3 never collides.
        // (Round 3 A/B, one device: the reads of step s+2 interleaved one behind each MFMA of step s by sched_group_barrier
        // instead of issued as a block: conv -0.5 % forward / -0.9 % in the train step, inside the run-to-run spread of 0.8 %;
        // the four reads step 0 needs forced in front of the other four: no change.  Two half workgroups per CU (2 staging + 4
        // MFMA waves, 8 x 32 tiles, 81 KB of LDS each: independent barrier domains; with and without a start stagger): conv
        // +15 % time (1.394 -> 1.603 ms per launch), the weight gradient beside it -7 % (power).  ONE MFMA wave per SIMD with
        // three tile rows (12 x 32 tile, 4 + 4 waves, 256 registers, no spills) and its reads interleaved one behind each MFMA --
        // the configuration a flag-synchronised ring would need: +9 % time (1.415 -> 1.546 ms; round 2 measured +8.6 % for four
        // rows without the interleave): a lone in-order wave reaches 48 cycles per MFMA here, not 32.  The weight pieces issued by
        // the OLDER MFMA wave of each SIMD (it idles ~2.5k cycles at the barrier) two half-steps ahead into a third weight buffer,
        // the staging waves' queue holding input loads only: +2.1 ... +3.3 % time -- the pieces' issue is hidden where it is,
        // and the younger wave's walk grows beside a partner that issues vector-memory instructions.  The epilogue's trailing
        // vmcnt(0) removed: no change (the compact-mask stores it waits for are not exposed).  None adopted.)
        load_x(0, 0, xf[0]);
        load_w(0, wf[0]);
        load_x(1, 0, xf[1]);
        load_w(1 * 3 + 0, wf[1]);
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            const int dx = s >> 2, ir = s & 3;
            if (s + 2 < 12) {
                const int dx2 = (s + 2) >> 2, ir2 = (s + 2) & 3;
                load_x(ir2, dx2, xf[(s + 2) % 3]);
                if (ir2 <= 2) load_w(ir2 * 3 + dx2, wf[(3 * dx2 + ir2) % 3]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ir >= 1) mac(1, wf[(3 * dx + ir - 1) % 3], xf[s % 3]);      // output row 1, tap (dy = ir - 1, dx)
            if (ir <= 2) mac(0, wf[(3 * dx + ir) % 3], xf[s % 3]);          // output row 0, tap (dy = ir, dx)
            if (drip && s >= 1 && s <= 8) store_pending(s - 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Epilogue over fp32 planes (straight-line operand variants; lanes outside the image read the zero page and write a
    // trash page): each lane owns one pixel and 16 channels as four float4 groups.
#ifdef XSD_DIAG   // epilogue split (slots 22..26 of the stamp buffer, oldest MFMA wave): [22] entry -> first row's operand / mask requests issued,
                  // [23] first row's arithmetic, [24] second row (requests + arithmetic), [25] the trailing vmcnt(0), [26] epilogues
    unsigned long long est[5] = {0, 0, 0, 0, 0};
    unsigned long long et0 = 0;
#define X3_ETICK(i) do { if (P.dbg) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); est[i] += t_ - et0; et0 = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define X3_ETICK(i) do { } while (0)
#endif
    auto epilogue_v = [&](const OutDesc& o, const TileXY& T, auto has_acc, auto has_e1, auto has_e2, auto has_e3, auto has_mask, auto generic, auto has_bits, auto has_bout) {
        // has_bits: the lrelu' mask comes as one 16-bit word per lane and row (bits_in, written by the forward conv's epilogue
        // below: bit 4q + t <-> this lane's channel 8q + 4h + t) instead of the 128-byte-per-pixel activation plane
        float* dst = o.p + (long long)T.b * o.bs;
        const long long sb = (long long)T.b * P.std_bs;
        // lane coordinates rebuilt from the lane id here (held across the K-loop they are spilled, and a scratch reload in the
        // epilogue is a full memory round trip -- ~4 us under this kernel's load -- with the matrix pipe idle: -1.8 %)
        unsigned int all = ~0u;
        asm volatile("" : "+s"(all));
        const int ln = (int)__builtin_amdgcn_mbcnt_hi(all, __builtin_amdgcn_mbcnt_lo(all, 0u));
        const int h = ln >> 5, l31 = ln & 31;
        float* const trash = const_cast<float*>(zero) + 64 + 4 * h;
        const int x = T.x0 + l31;
        const float s1 = (decltype(generic)::value && !o.e1) ? 0.f : o.s1, s2v = (decltype(generic)::value && !o.e2) ? 0.f : o.s2;
        const float s3 = (decltype(generic)::value && !o.e3) ? 0.f : o.s3, msl = (decltype(generic)::value && !o.mask) ? 1.f : o.mslope;
        const float sacc = (decltype(generic)::value && !o.accumulate) ? 0.f : 1.f;
        // a1e undoes the operand scales (a power of two: exact).  Without an accumulate / e1 operand between the two factors
        // a2 multiplies the same value and is folded in (every such launch of the engine has a2 = 1: exact there)
        constexpr bool fold_a2 = !decltype(has_acc)::value && !decltype(has_e1)::value;
        const float a1e = fold_a2 ? o.a1 * inv_s * o.a2 : o.a1 * inv_s;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int y = T.y0 + wv * 2 + r;
            const bool valid = x < P.W && y < P.H;
            const int doff = valid ? (y * o.rs + x * o.ps + 4 * h) * 4 : (int)0x80000000;
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
            if (r == 0) X3_ETICK(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = (acc[r][4 * q + t] + accx[r][4 * q + t] * 0x1p-11f) * a1e;
                if constexpr (decltype(has_acc)::value) v += sacc * va[q];
                if constexpr (decltype(has_e1)::value) v += s1 * v1[q];
                if constexpr (!fold_a2) v *= o.a2;
                if constexpr (decltype(has_e2)::value) v += s2v * v2[q];
                if constexpr (decltype(has_e3)::value) v += s3 * v3[q];
                if constexpr (x3_kind_has_lrelu(KIND)) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = __builtin_fmaxf(v[t], v[t] * o.slope);   // = v > 0 ? v : v * slope for 0 <= slope <= 1 (0.2, 0.01, 1)
                }
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
#pragma unroll
                for (int t = 0; t < 4; ++t) run_max = __builtin_fmaxf(run_max, valid ? __builtin_fabsf(v[t]) : 0.f);   // select, no branch
            }
            pend_off[r] = doff;
            if constexpr (decltype(has_bout)::value) { if (valid) o.bits_out[widx] = (unsigned short)obits; }
            X3_ETICK(1 + r);
        }
        {
            const unsigned long long d = reinterpret_cast<unsigned long long>(dst);
            pend_lo = (unsigned int)d; pend_hi = (unsigned int)(d >> 32);
            pend_nb = (int)((unsigned int)P.H * (unsigned int)o.rs * 4u);
        }
    };
    auto epilogue = [&](int j, const TileXY& T) {
#ifdef XSD_DIAG
        if (P.dbg) { et0 = __builtin_readcyclecounter(); est[4] += 1; }
#endif
        const OutDesc o = P.out[j];
        using Y = std::true_type; using N = std::false_type;
        const int kind = (o.accumulate ? 1 : 0) | (o.e1 ? 2 : 0) | (o.e2 ? 4 : 0) | (o.e3 ? 8 : 0) | (o.mask ? 16 : 0);
        // (bits_out with operands other than none, bits_in without mask: not in the engine's plans -> rejected by the launcher)
        if constexpr (KIND == 0) epilogue_v(o, T, N{}, N{}, N{}, N{}, N{}, N{}, N{}, N{});
        else if constexpr (KIND == 64) epilogue_v(o, T, N{}, N{}, N{}, N{}, N{}, N{}, N{}, Y{});
        else if constexpr (KIND == 2) epilogue_v(o, T, N{}, Y{}, N{}, N{}, N{}, N{}, N{}, N{});
        else if constexpr (KIND == 6) epilogue_v(o, T, N{}, Y{}, Y{}, N{}, N{}, N{}, N{}, N{});
        else if constexpr (KIND == 14) epilogue_v(o, T, N{}, Y{}, Y{}, Y{}, N{}, N{}, N{}, N{});
        else if constexpr (KIND == 16) epilogue_v(o, T, N{}, N{}, N{}, N{}, Y{}, N{}, N{}, N{});
        else if constexpr (KIND == 48) epilogue_v(o, T, N{}, N{}, N{}, N{}, Y{}, N{}, Y{}, N{});
        else {
            static_assert(KIND < 0, "no epilogue variant for this kind");
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
        }
        // every load of the epilogue (operands, register reloads) has landed before the MFMA walk starts: its deferred
        // stores then need no vector-memory waits (hipcc would otherwise put `vmcnt(1)` in front of each, i.e. wait for
        // the store before the previous one)
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), gfx9 encoding
        X3_ETICK(3);
    };

#ifdef XSD_DIAG   // phase stamps (diagnostic library variant only; tools/stamps.py): accumulated shader cycles per phase
    unsigned long long st[6] = {0, 0, 0, 0, 0, 0};
    const bool stamp = P.dbg != nullptr;
#define X3_TICK(i) do { if (stamp) { const unsigned long long t_ = __builtin_readcyclecounter(); st[i] += t_ - t0; t0 = t_; } } while (0)
#else
#define X3_TICK(i) do { } while (0)
#endif

    Cur cur = {0, 0, 0, 0};
    TileXY tcur = tile_of(0);
    bool pending = false;     // a finished tile's results are waiting in `pend`
    int xpar = 0;             // input buffer of the current half-step
    lds_barrier();                                                                         // (P)
    X3_TICK(0);
#pragma unroll 1
    for (int it = 0; it < items; ++it) {
        const bool more1 = (it + 1 < items);
        X3_TICK(1);
        if (cur.i == 0 && cur.s2 == 0) init_acc(cur.j);
        compute(smem + xpar * X3_XB, it & 1, pending);
        xpar = (xpar + 1 == L::NXB) ? 0 : xpar + 1;
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
        lds_barrier();                                                                     // end of half-step `it`
        X3_TICK(4);
        const Cur nx = succ(cur);
        if (nx.k != cur.k && more1) tcur = tile_of(nx.k);
        cur = nx;
    }
    // The planes' max |x| for the kernels that consume them (they scale their operands into the fp16 range with it): one
    // value per launch -- for a launch with several output planes the maximum over all of them, a valid (if not the
    // tightest) bound for each -- reduced over the wave by DPP (row shifts, then the two row broadcasts: lane 63 ends up with
    // the maximum; 0 is the identity for non-negative values) and published with one atomic per wave and plane.  Done
    // once at the end of the kernel: an atomic per tile would sit in front of the epilogue's vmcnt(0) for ~3k cycles.
    {
        auto dpp_max = [&](float x, auto ctrl, auto rowmask) {
            const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, decltype(rowmask)::value, 0xf, true);
            return __builtin_fmaxf(x, __builtin_bit_cast(float, t));
        };
        float m = run_max;
        m = dpp_max(m, std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});   // row_shr:1
        m = dpp_max(m, std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});   // row_shr:2
        m = dpp_max(m, std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});   // row_shr:4
        m = dpp_max(m, std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});   // row_shr:8
        m = dpp_max(m, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});   // row_bcast:15 -> rows 1, 3
        m = dpp_max(m, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});   // row_bcast:31 -> rows 2, 3
        // ONE global atomic per workgroup and plane (round 6).  Eight per workgroup -- 2,048 atomics on one address at the end of every
        // launch -- cost 22 us per launch on this device whatever the launch computed (tools/launch_floor_probe.hip: an empty
        // conv-shaped launch 2.6 us, with 8 atomicMax per workgroup to one address 24.5 us, with one per workgroup 4.2 us): same-address
        // atomics are served one after the other at the memory side.  The MFMA waves combine through an LDS word (ds_max), the last
        // one to arrive (an LDS counter; LDS operations of a wave complete in order, so its read sees every maximum) publishes.
        if (lane == 63) {
            unsigned int* comb = reinterpret_cast<unsigned int*>(desc + 15);
            atomicMax(&comb[0], __float_as_uint(m));                     // non-negative floats order as integers
            if (atomicAdd(&comb[1], 1u) == (unsigned int)(X3_MWAVES - 1)) {
                const unsigned int all = atomicMax(&comb[0], 0u);
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    if (j < n_out && P.out[j].amax) atomicMax(reinterpret_cast<unsigned int*>(P.out[j].amax), all);
            }
        }
    }
#ifdef XSD_DIAG
    if (stamp && tid == 64 * X3_LWAVES) {
#pragma unroll
        for (int q = 0; q < 6; ++q) atomicAdd(&P.dbg[q], st[q]);
        atomicAdd(&P.dbg[6], (unsigned long long)items);
#pragma unroll
        for (int q = 0; q < 5; ++q) atomicAdd(&P.dbg[22 + q], est[q]);
        atomicAdd(&P.dbg[7], __builtin_amdgcn_s_memrealtime() - rt0);
    }
    if (stamp && tid == 64 * (X3_LWAVES + X3_MWAVES - 1)) {   // the youngest MFMA wave: slots 13..15 = MFMA walk, epilogue, barrier A
        atomicAdd(&P.dbg[13], st[2]); atomicAdd(&P.dbg[14], st[3]); atomicAdd(&P.dbg[15], st[4]);
    }
#endif
}

__global__ __launch_bounds__(X3_THREADS) void conv3x3_h2x_kernel(const ConvParams P) { conv3x3_h2x_body<-1>(P); }
template <int KIND> __global__ __launch_bounds__(X3_THREADS) void conv3x3_h2x_kind_kernel(const ConvParams P) { conv3x3_h2x_body<KIND>(P); }

// Which kinds run on their own instance (bit i <-> X3_KIND_LIST[i]); the others, and every launch whose output chunks differ in
// kind or carry a LeakyReLU an instance has dropped, run on the catch-all kernel.  Set from same-device per-kind measurements
// (profiles/r04_ab_conv_kind_instances.txt): an instance is only worth its code if its launches get faster.
#ifndef X3_KINDS
#define X3_KINDS 68      // kinds 2 (one operand plane) and 48 (compact-mask read)
#endif
constexpr int X3_KIND_LIST[7] = {0, 64, 2, 6, 14, 16, 48};

static PerDevice g_once;

static int x3_kind_of(const OutDesc& o)
{
    return (o.accumulate ? 1 : 0) | (o.e1 ? 2 : 0) | (o.e2 ? 4 : 0) | (o.e3 ? 8 : 0) | (o.mask ? 16 : 0) | ((o.mask && o.bits_in) ? 32 : 0) | (o.bits_out ? 64 : 0);
}

hipError_t launch_conv3x3_h2x(const ConvParams& p, hipStream_t stream)
{
    int ncu = 256;
    hipError_t e = g_once.once([]() {
        hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_h2x_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, X3L::BYTES);
        // only the adopted instances exist in the library: `if constexpr` on the X3_KINDS bit keeps the others uninstantiated
#define X3_ATTR(I, K) if constexpr ((X3_KINDS >> I) & 1) { static_assert(X3_KIND_LIST[I] == K, "kind list"); if (r == hipSuccess) r = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_h2x_kind_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, X3L::BYTES); }
        X3_ATTR(0, 0) X3_ATTR(1, 64) X3_ATTR(2, 2) X3_ATTR(3, 6) X3_ATTR(4, 14) X3_ATTR(5, 16) X3_ATTR(6, 48)
#undef X3_ATTR
        return r;
    }, &ncu);
    if (e != hipSuccess) return e;
    if (p.n_in < 1 || p.n_out < 1 || p.n_in * p.n_out > 5 || !p.zero) return hipErrorInvalidValue;
    // the operand scales come from these slots: an unknown maximum must not silently become "1.0" (fp16 would overflow)
    if (!p.amax_w) return hipErrorInvalidValue;
    for (int j = 0; j < p.n_out; ++j) {      // compact masks: only the combinations the epilogue has variants for
        const OutDesc& o = p.out[j];
        if (o.bits_out && (o.accumulate || o.e1 || o.e2 || o.e3 || o.mask)) return hipErrorInvalidValue;
        // the deferred stores address the output plane with 32-bit BYTE offsets against a descriptor of H * rs * 4 bytes
        if ((long long)p.H * o.rs * 4 >= (1ll << 31)) return hipErrorInvalidValue;
    }
    for (int i = 0; i < p.n_in; ++i) {
        if (!p.amax_in[i]) return hipErrorInvalidValue;
        // the staging waves compute every input's lane offsets and buffer range from in[0]'s strides (32-bit byte offsets)
        if (p.in[i].rs != p.in[0].rs || p.in[i].ps != p.in[0].ps) return hipErrorInvalidValue;
    }
    if ((long long)p.H * p.in[0].rs * 4 >= (1ll << 31)) return hipErrorInvalidValue;
    const int tilesY = (p.H + X3_ROWS - 1) / X3_ROWS;
    const int ntiles = p.B * p.tilesX * tilesY;
    if (ntiles <= 0) return hipSuccess;
    const dim3 g(persistent_grid(ntiles, ncu)), b(X3_THREADS);
    // the launch's kind: the same for every output chunk, and no LeakyReLU where the instance has none
    int kind = x3_kind_of(p.out[0]);
    for (int j = 1; j < p.n_out; ++j) if (x3_kind_of(p.out[j]) != kind) kind = -1;
    if (kind >= 0 && !x3_kind_has_lrelu(kind))
        for (int j = 0; j < p.n_out; ++j) if (p.out[j].slope != 1.f) kind = -1;
    int idx = -1;
    for (int i = 0; i < 7; ++i) if (X3_KIND_LIST[i] == kind && ((X3_KINDS >> i) & 1)) idx = i;
    bool launched = false;
#define X3_LAUNCH(I, K) if constexpr ((X3_KINDS >> I) & 1) { if (idx == I) { hipLaunchKernelGGL(conv3x3_h2x_kind_kernel<K>, g, b, X3L::BYTES, stream, p); launched = true; } }
    X3_LAUNCH(0, 0) X3_LAUNCH(1, 64) X3_LAUNCH(2, 2) X3_LAUNCH(3, 6) X3_LAUNCH(4, 14) X3_LAUNCH(5, 16) X3_LAUNCH(6, 48)
#undef X3_LAUNCH
    if (!launched) hipLaunchKernelGGL(conv3x3_h2x_kernel, g, b, X3L::BYTES, stream, p);
    return hipGetLastError();
}

} // namespace xsd
