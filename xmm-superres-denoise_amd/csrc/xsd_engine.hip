// xsd_engine.hip -- host side of the engine: network plan (forward / backward launch schedules over feature planes),
// workspace, weight packing, and the C ABI declared in include/xsd.h.
//
// The schedule restates, layer by layer, the reference graph
//   _GeneratorRRDB.forward / GeneratorRRDB_SR.forward / GeneratorRRDB_DN.forward (generator_rrdb.py:66-69,103-110,130-137)
//   RRDB.forward / ResidualDenseBlock_5C.forward                                   (rrdb_blocks.py:37-54,66-70)
//   Model.forward's second clamp                                                   (models/model.py:48-49)
// and its reverse-mode derivative (what torch autograd would run), with every elementwise op fused into the epilogue
// of the conv that produces its operand.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <map>
#include <tuple>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <new>
#include <vector>

#include "../../include/xsd.h"
#include "xsd_aux.h"
#include "generic_net.h"
#include "xsd_loss.h"

static constexpr int EDGE_WGRAD_BLOCKS = 2048; // 8 workgroups of 256 threads per CU
#include "xsd_kernels.h"

using namespace xsd;

namespace xsd {
int debug_conv_occupancy(int lds_bytes);
float debug_residency_ms(int grid, int threads, int lds_bytes, int us);
hipError_t probe_mfma_stream(int fmt, double seconds, double* mfma_tflops, double* sclk_ghz, hipStream_t stream);
hipError_t launch_pack_shuffle_bias(const float* b, float* out, int planes, hipStream_t s);
}

static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) return fail(XSD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

struct ConvW {       // one MFMA conv's weights
    long long w_off, b_off; // flat-param offsets
    int cout, cin, shuffle;
    long long fwd_off, bwd_off; // packed offsets (floats)
    long long sbias_off;        // shuffled-bias offset (shuffle convs) or -1
    float bwd_scale = 1.f;      // folded into the input-gradient panels
};

typedef std::function<hipError_t(hipStream_t)> Launch;

struct ProfRec { hipEvent_t a, b; double flop, bytes; int klass; };

struct xsd_engine {
    xsd_config cfg;
    GenericNet* generic = nullptr;   // what the plane kernels do not take (generic_net.hip): more than 256 filters, more than 8 image channels
    // Widths that are no multiple of 32 run zero-padded to the next one: the engine is built for `nf_pad` filters over an internal,
    // padded flat parameter vector (pad weights and biases 0 -> pad channels are exactly 0 everywhere, forward and backward); the
    // caller's vectors keep the reference's layout for `cfg.num_filters` (pad_params_kernel expands, pad_grads_kernel gathers)
    int nf_pad = 0;                  // 0: no padding
    long long nparams_pad = 0;
    float* params_pad = nullptr;
    float* grads_pad = nullptr;
    void* pad_descs = nullptr;       // PadDesc[npad]: first | dense blocks | trunk, last, up.., hr  (= backward stages last | 1..blocks | 0)
    int npad = 0;
    std::vector<long long> rrdb_begin_real;
    long long nparams_real = 0;
    int planes = 1;                  // num_filters / 32: 32-channel planes per feature tensor (1 = the shipped configuration; 2..8: Builder::build_multi)
    long long nparams = 0;
    // flat-param offsets
    long long first_w = 0, first_b = 0, last_w = 0, last_b = 0;
    std::vector<ConvW> rdb; // blocks*15
    ConvW trunk, hr;
    std::vector<ConvW> up;
    std::vector<long long> rrdb_begin; // flat offset where rrdb.i starts; [blocks] = trunk offset
    // packed weights
    float* pk_fwd = nullptr;
    float* pk_bwd = nullptr;
    unsigned short* pk_fwd_s = nullptr; // split-mode panels (mode 3: fp32 in bf16-MFMA fragment order; mode 4: two-term fp16 images),
    unsigned short* pk_bwd_s = nullptr; // same byte size / offsets as pk_fwd / pk_bwd
    int chunk = 0;             // diagnostic library only (env XSD_CHUNK): images per dense-block sweep (0 = whole batch)
    int ablate = 0;            // diagnostic library only (env XSD_ABLATE): ablation knobs of the kernels
    int math = 4;              // include/xsd.h: xsd_set_math (default: f16x3, the faster of the two fp32-class split modes)
    float* pk_edge = nullptr; // first_fwd, first_bwd, last_fwd, last_bwd (288 each)
    float* pk_sbias = nullptr;
    PackDesc* descs_dev = nullptr;
    int ndesc = 0;
    long long pk_floats = 0;
    // math mode 4 (f16x3): max |x| slots; [0] packed forward panels, [1] packed input-gradient panels, [2..] planes of the plan
    float* amax = nullptr;
    int amax_used = 2;
    int amax_bwd_first = 2;    // first slot the plan hands out while it builds the BACKWARD stages (re-zeroed by stage 0 of every backward)
    int ncu = 256;             // compute units of the device the engine was created on (the weight gradient's block launch is dealt from it)
    int amax_cap = 65536;      // floats allocated behind `amax`; ensure_plan grows it to what the plan's sizing pass counted (+ AMAX_TAIL)
    static constexpr int AMAX_TAIL = 64;   // the last slots belong to the single-conv entry points (pack_single, the weight-gradient hook)
    const float* params = nullptr; // borrowed (bias reads)
    bool packed = false;
    // workspace
    char* ws = nullptr;
    size_t ws_bytes = 0;
    float* wg_partial = nullptr;
    float* wg_bias_partial = nullptr;
    float* edge_partial = nullptr;
    double* loss_partial = nullptr;
    int nparts = 256;
    // plan
    int pB = 0, pH = 0, pW = 0, ptrain = -1;
    std::vector<Launch> fwd_ops;
    std::vector<std::vector<Launch>> bwd_stages;
    bool fwd_saved = false;
    // late-bound call pointers
    const float* b_x = nullptr;
    float* b_y = nullptr;
    const float* b_dy = nullptr;
    float* b_dx = nullptr;
    float* b_grads = nullptr;
    void* zero_page = nullptr;         // 256 B of zeros (padding source of the DMA / unconditional loads) + 256 B of trash
    unsigned long long* dbg = nullptr; // conv phase stamps (diagnostic)
    // profiling
    bool prof = false;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
};

// ------------------------------------------------------------------------------------------------------------
static void take_conv(long long& off, int cout, int cin, long long& w, long long& b)
{
    w = off; off += (long long)cout * cin * 9;
    b = off; off += cout;
}

// weight gradient of the split modes: mode 3 (bf16x6) and mode 4 (f16x3) have their own role-split kernels (one workgroup
// per CU, `nparts` partial sums); diagnostic ablate bit 22 runs mode 3's kernel in mode 4
static hipError_t launch_wgrad_split(int math, int ablate, const WgradParams& p, hipStream_t s)
{
    return (math == 4 && !(ablate & (1 << 22))) ? launch_wgrad_h2x(p, s) : launch_wgrad_s3x(p, s);
}

static hipError_t prof_launch(xsd_engine* e, int klass, double flop, double bytes, hipStream_t s, const std::function<hipError_t()>& f)
{
    if (!e->prof) return f();
    auto get_ev = [&]() -> hipEvent_t {
        if (e->ev_used == e->ev_pool.size()) {
            hipEvent_t ev;
            if (hipEventCreate(&ev) != hipSuccess) return nullptr;
            e->ev_pool.push_back(ev);
        }
        return e->ev_pool[e->ev_used++];
    };
    ProfRec r{get_ev(), get_ev(), flop, bytes, klass};
    if (!r.a || !r.b) return hipErrorOutOfMemory;
    hipEventRecord(r.a, s);
    hipError_t err = f();
    hipEventRecord(r.b, s);
    e->recs.push_back(r);
    return err;
}

// conv launch of the engine's math mode (mode 4 also needs the weight buffers' max |w| slot: forward or input-gradient panels)
static hipError_t launch_conv_any(xsd_engine* eng, ConvParams& p, hipStream_t s)
{
    if (eng->math == 4) {
        if (!p.amax_w) {
            const float* lo = reinterpret_cast<const float*>(eng->pk_fwd_s);
            p.amax_w = (p.wstep[0] >= lo && p.wstep[0] < lo + eng->pk_floats) ? eng->amax + 0 : eng->amax + 1;
        }
        return launch_conv3x3_h2x(p, s);
    }
    return eng->math == 3 ? launch_conv3x3_s3x(p, s) : launch_conv3x3_mfma(p, s);
}

// Profile classes of xsd_profile_read (include/xsd.h): 0 conv (forward + input-gradient), 1 weight gradient; the HBM-bound kernels:
// 2 edge_expand (1 -> 32), 3 edge_reduce (32 -> 1), 4 edge_wgrad, 5 L1 loss, 6 Adam, 7 clamp backward, 8 plane max |x| sweeps.
// `bytes` is ALGORITHMIC traffic (SURVEY 8d rule: every operand once), `flop` 2 x MAC.
enum { PK_CONV = 0, PK_WGRAD = 1, PK_EDGE_EXPAND = 2, PK_EDGE_REDUCE = 3, PK_EDGE_WGRAD = 4, PK_LOSS = 5, PK_ADAM = 6, PK_CLAMP_BWD = 7, PK_PLANE_AMAX = 8, PK_COUNT = 9 };
static hipError_t run_edge_expand(xsd_engine* e, const EdgeExpandParams& p, hipStream_t s)
{
    const double px = (double)p.B * p.H * p.W;      // reads the image (4 B/px), writes the plane (128); + the plane it adds to; + the mask plane or its compact words
    const double bytes = px * (4.0 + 128.0 + (p.accumulate ? 128.0 : 0.0) + (p.bits ? 4.0 : (p.mask ? 128.0 : 0.0)));
    return prof_launch(e, PK_EDGE_EXPAND, 2.0 * 9 * 32 * px, bytes, s, [&]() { return launch_edge_expand(p, s); });
}
static hipError_t run_edge_reduce(xsd_engine* e, const EdgeReduceParams& p, hipStream_t s)
{
    const double px = (double)p.B * p.H * p.W;      // reads the plane (128 B/px) (+ skip, + addto), writes y (+ pre)
    const double bytes = px * (128.0 + 4.0 + (p.skip ? 4.0 : 0.0) + (p.addto ? 4.0 : 0.0) + (p.pre ? 4.0 : 0.0));
    return prof_launch(e, PK_EDGE_REDUCE, 2.0 * 9 * 32 * px, bytes, s, [&]() { return launch_edge_reduce(p, s); });
}
static hipError_t run_edge_wgrad(xsd_engine* e, const EdgeWgradParams& p, int which, float* dw, float* db, hipStream_t s, int cstride = 9)
{
    const double px = (double)p.B * p.H * p.W;      // reads the plane and the image once
    return prof_launch(e, PK_EDGE_WGRAD, 2.0 * 9 * 32 * px, px * (128.0 + 4.0), s, [&]() { return launch_edge_wgrad(p, which, dw, db, s, cstride); });
}

// ------------------------------------------------------------------------------------------------------------
// Plan builder
// ------------------------------------------------------------------------------------------------------------
struct Builder {
    xsd_engine* e;
    int B, H, W;
    bool train;
    uintptr_t base;      // 0 in the sizing pass
    size_t top = 0, peak = 0;
    std::vector<std::vector<size_t>> freelist; // per level
    // math mode 4 (f16x3): which plane views have a slot holding their max |x| (set by the producing conv's epilogue or by
    // a plane_amax launch in front of the first consumer); a view dies when its buffer is handed out again
    typedef std::tuple<uintptr_t, int, int> ViewKey;
    std::map<ViewKey, float*> amax_valid;
    Builder(xsd_engine* e_, int B_, int H_, int W_, bool train_, uintptr_t base_)
        : e(e_), B(B_), H(H_), W(W_), train(train_), base(base_), freelist(8) {}

    float* new_slot()
    {
        // one slot per plane view a launch writes or reads first.  The sizing pass of ensure_plan only COUNTS (a deep, wide net
        // needs more than the initial allocation: 256 filters x 64 blocks is ~80k); the real pass runs after the array has
        // been grown to the count, so no two planes ever share a slot
        const int i = e->amax_used++;
        return (e->amax && i < e->amax_cap - xsd_engine::AMAX_TAIL) ? e->amax + i : nullptr;
    }
    void invalidate_range(uintptr_t lo, size_t bytes)
    {
        for (auto it = amax_valid.begin(); it != amax_valid.end();) {
            const uintptr_t k = std::get<0>(it->first);
            if (k >= lo && k < lo + bytes) it = amax_valid.erase(it); else ++it;
        }
    }
    // a producer outside the conv kernels (edge_expand) reports the max |x| of the standard plane it writes
    float* report_slot(float* plane, int level)
    {
        if (e->math != 4) return nullptr;
        const PlaneIn v = std_in(plane, level);
        invalidate_range(reinterpret_cast<uintptr_t>(plane), plane_bytes(level));
        float* slot = new_slot();
        amax_valid[ViewKey(reinterpret_cast<uintptr_t>(v.p), v.rs, v.ps)] = slot;
        return slot;
    }
    // slot of an input view; if nobody has reported it yet, `pre` gets the reduction launch
    float* slot_of(const PlaneIn& v, int Hv, int Wv, std::vector<Launch>& pre)
    {
        const ViewKey key(reinterpret_cast<uintptr_t>(v.p), v.rs, v.ps);
        auto it = amax_valid.find(key);
        if (it != amax_valid.end()) return it->second;
        float* slot = new_slot();
        amax_valid[key] = slot;
        const int Bv = B;
        xsd_engine* eng = e;
        pre.push_back([eng, v, Bv, Hv, Wv, slot](hipStream_t s) { return prof_launch(eng, PK_PLANE_AMAX, 0.0, 128.0 * Bv * Hv * Wv, s, [&]() { return launch_plane_amax(v, Bv, Hv, Wv, slot, s); }); });
        return slot;
    }

    // The ONE launch that wrote the four pixel-shuffled views of a high-resolution plane published ONE maximum -- over all of its
    // output chunks (conv3x3_h2x.hip: run_max) -- into each view's slot: that number IS the plane's max |x|, so the standard view
    // of the plane (what HRconv and the weight gradient read) takes view 0's slot instead of a plane_amax sweep over 4 x the
    // low-resolution pixels (round 6: 2.7 % of a batch-1 SR forward, 0.8 % at batch 16; results identical by construction)
    void alias_shuffled(float* hr, int level)
    {
        if (e->math != 4) return;
        OutDesc o; memset(&o, 0, sizeof(o));
        shuf_out(o, hr, level, 0);
        auto it = amax_valid.find(ViewKey(reinterpret_cast<uintptr_t>(o.p), o.rs, o.ps));
        if (it == amax_valid.end()) return;
        const PlaneIn v = std_in(hr, level + 1);
        amax_valid[ViewKey(reinterpret_cast<uintptr_t>(v.p), v.rs, v.ps)] = it->second;
    }

    size_t plane_bytes(int level, int ch = 32) const
    {
        return ((size_t)B * (H << level) * (W << level) * ch * sizeof(float) + 255) & ~(size_t)255;
    }
    float* alloc(int level)
    {
        size_t off;
        if (!freelist[level].empty()) { off = freelist[level].back(); freelist[level].pop_back(); }
        else { off = top; top += plane_bytes(level); if (top > peak) peak = top; }
        if (e->math == 4) invalidate_range(base + off, plane_bytes(level));
        return reinterpret_cast<float*>(base + off);
    }
    float* alloc1(int level) // 1-channel image
    {
        size_t off = top; top += plane_bytes(level, 1); if (top > peak) peak = top;
        return reinterpret_cast<float*>(base + off);
    }
    float* alloc_img(int level, int ch) // ch-channel image [B][ch][H][W]
    {
        size_t off = top; top += plane_bytes(level, ch); if (top > peak) peak = top;
        return reinterpret_cast<float*>(base + off);
    }
    void release(float* p, int level, bool force = false)
    {
        if (train && !force) return; // saved for backward
        freelist[level].push_back(reinterpret_cast<uintptr_t>(p) - base);
    }

    // packed-panel pointers (every mode's panels are PANEL_FLOATS * 4 bytes)
    const float* fwdp(long long off) const
    {
        return e->math ? reinterpret_cast<const float*>(e->pk_fwd_s) + off : e->pk_fwd + off;
    }
    const float* bwdp(long long off) const
    {
        return e->math ? reinterpret_cast<const float*>(e->pk_bwd_s) + off : e->pk_bwd + off;
    }
    ConvParams conv_base(int level) const
    {
        ConvParams p;
        memset(&p, 0, sizeof(p));
        p.B = B; p.H = H << level; p.W = W << level;
        p.tilesX = (p.W + TILE_W - 1) / TILE_W;
        p.tilesY = (p.H + TILE_H - 1) / TILE_H;
        p.std_rs = p.W * 32;
        p.std_bs = (long long)p.H * p.W * 32;
        for (int j = 0; j < 5; ++j) { p.out[j].a1 = 1.f; p.out[j].a2 = 1.f; p.out[j].slope = 1.f; p.out[j].mslope = 1.f; }
        return p;
    }
    PlaneIn std_in(const float* p, int level) const
    {
        PlaneIn r; r.p = p; r.ps = 32; r.rs = (W << level) * 32; r.bs = (long long)(H << level) * (W << level) * 32; return r;
    }
    // sub-pixel (i,j) view at `level` of a plane stored at level+1 (PixelShuffle(2), generator_rrdb.py:97)
    PlaneIn shuf_in(const float* hr, int level, int n) const
    {
        const int Wl = W << level, Hl = H << level;
        PlaneIn r; r.p = hr + ((long long)(n >> 1) * 2 * Wl + (n & 1)) * 32; r.ps = 64; r.rs = 4 * Wl * 32;
        r.bs = (long long)4 * Hl * Wl * 32; return r;
    }
    void std_out(OutDesc& o, float* p, int level) const
    {
        o.p = p; o.ps = 32; o.rs = (W << level) * 32; o.bs = (long long)(H << level) * (W << level) * 32;
    }
    void shuf_out(OutDesc& o, float* hr, int level, int n) const
    {
        PlaneIn r = shuf_in(hr, level, n);
        o.p = const_cast<float*>(r.p); o.ps = r.ps; o.rs = r.rs; o.bs = r.bs;
    }

    // the same conv restricted to images [b0, b0 + nb)
    static ConvParams slice(const ConvParams& q, int b0, int nb)
    {
        ConvParams p = q;
        p.B = nb;
        for (int i = 0; i < 5; ++i) if (p.in[i].p) p.in[i].p += (long long)b0 * p.in[i].bs;
        for (int j = 0; j < 5; ++j) {
            OutDesc& o = p.out[j];
            if (o.p) o.p += (long long)b0 * o.bs;
            if (o.e1) o.e1 += (long long)b0 * p.std_bs;
            if (o.e2) o.e2 += (long long)b0 * p.std_bs;
            if (o.e3) o.e3 += (long long)b0 * p.std_bs;
            if (o.mask) o.mask += (long long)b0 * p.std_bs;
            if (o.bits_out) o.bits_out += (long long)b0 * (p.std_bs / 16);   // 2 half-words per pixel of 32 floats
            if (o.bits_in) o.bits_in += (long long)b0 * (p.std_bs / 16);
        }
        return p;
    }
    Launch conv_launch(const ConvParams& p_in, bool bias_from_params, long long bias_off)
    {
        ConvParams p = p_in;
        {
            const int nst = p.n_out > 1 ? p.n_out : p.n_in;
            for (int i = 0; i < nst; ++i) if (!p.wstep[i]) p.wstep[i] = p.wpanel + (long long)i * PANEL_FLOATS;
        }
        xsd_engine* eng = e;
        const double px = (double)p.B * p.H * p.W;
        const double flop = 2.0 * 9 * 32 * 32 * p.n_in * p.n_out * px;
        const double bytes = 128.0 * (p.n_in + p.n_out) * px;
        std::vector<Launch> pre;     // math mode 4: reductions for input planes nobody has reported yet
        if (e->math == 4) {
            for (int i = 0; i < p.n_in; ++i) p.amax_in[i] = slot_of(p.in[i], p.H, p.W, pre);
            for (int j = 0; j < p.n_out; ++j) {   // this launch reports its own outputs
                OutDesc& o = p.out[j];
                // every view of the plane this launch writes loses its slot: a shuffled view (shuf_out) is one of four that
                // interleave over the whole high-resolution plane, so the range is the view's full extent, not its first byte
                invalidate_range(reinterpret_cast<uintptr_t>(o.p), (size_t)p.B * (size_t)o.bs * sizeof(float));
                o.amax = new_slot();
                amax_valid[ViewKey(reinterpret_cast<uintptr_t>(o.p), o.rs, o.ps)] = o.amax;
            }
        }
        return [eng, p, pre, bias_from_params, bias_off, flop, bytes](hipStream_t s) mutable {
            if (bias_from_params) p.bias = eng->params + bias_off;
            p.dbg = eng->dbg;
            p.ablate = eng->ablate;
            p.zero = eng->zero_page;
            for (auto& f : pre) { hipError_t err = f(s); if (err != hipSuccess) return err; }
            return prof_launch(eng, 0, flop, bytes, s, [&]() { return launch_conv_any(eng, p, s); });
        };
    }
    // wgrad + fixed-order reduce into the flat gradient vector
    void wgrad_launch(std::vector<Launch>& ops, int level, const std::vector<PlaneIn>& xs, const std::vector<PlaneIn>& gs,
                      const ConvW& cw, float scale, int j0 = 0, int n0 = 0, int plane = 0)
    {
        xsd_engine* eng = e;
        WgradParams wp;
        memset(&wp, 0, sizeof(wp));
        wp.B = B; wp.H = H << level; wp.W = W << level;
        wp.tilesX = (wp.W + TILE_W - 1) / TILE_W; wp.tilesY = (wp.H + TILE_H - 1) / TILE_H;
        wp.n_in = (int)xs.size(); wp.n_g = (int)gs.size(); wp.nparts = e->nparts;
#ifndef XSD_WGRAD_ROUNDS
        // The grid is nparts x n_in x n_g workgroups of one per CU: with nparts = the CU count a launch over k (X, G) pairs runs k
        // rounds of workgroups, and every round pays a workgroup's fixed cost again (first two tiles' round trips, the final
        // LDS reduction and partial-sum store, ~30 us of a ~450 us round: kernel trace, 0.476 / 0.911 / 1.359 / 1.731 / 2.250 ms
        // for 1..5 pairs).  Where the pairs divide the CUs into a multiple of eight parts (2 and 4 pairs: 128 / 64 parts) the whole
        // launch is ONE round of k-times longer workgroups; the workgroups that share G tiles still run side by side on one XCD.
        // 3 and 5 pairs would leave 16 CUs idle (240 workgroups): they keep the k rounds.
        {
            const int pairs = wp.n_in * wp.n_g;
            if (e->math >= 3 && pairs > 1 && e->nparts % pairs == 0 && ((e->nparts / pairs) & 7) == 0) wp.nparts = e->nparts / pairs;   // (the split modes' kernels: one workgroup per CU; the exact-fp32 kernel runs two)
        }
#endif
        for (size_t i = 0; i < xs.size(); ++i) wp.x[i] = xs[i];
        for (size_t i = 0; i < gs.size(); ++i) wp.g[i] = gs[i];
        std::vector<Launch> pre;     // math mode 4: reductions for planes nobody has reported yet
        if (e->math == 4) {
            for (size_t i = 0; i < xs.size(); ++i) wp.amax_x[i] = slot_of(xs[i], wp.H, wp.W, pre);
            for (size_t i = 0; i < gs.size(); ++i) wp.amax_g[i] = slot_of(gs[i], wp.H, wp.W, pre);
        }
        WgradReduceParams rp;
        memset(&rp, 0, sizeof(rp));
        rp.nparts = wp.nparts; rp.n_in = wp.n_in; rp.n_g = wp.n_g; rp.cin_total = cw.cin; rp.cout_total = cw.cout;
        rp.shuffle = cw.shuffle; rp.scale = scale; rp.j0 = j0; rp.n0 = n0; rp.plane = plane;
        const long long w_off = cw.w_off, b_off = cw.b_off;
        const double px = (double)wp.B * wp.H * wp.W;
        const double flop = 2.0 * 9 * 32 * 32 * wp.n_in * wp.n_g * px;
        const double bytes = 128.0 * (wp.n_in + wp.n_g) * px; // SURVEY 8(d) rule: every X plane and every G plane once (the kernel itself re-reads G once per X plane)
        ops.push_back([eng, wp, rp, pre, w_off, b_off, flop, bytes](hipStream_t s) mutable {
            for (auto& f : pre) { hipError_t perr = f(s); if (perr != hipSuccess) return perr; }
            wp.partial = eng->wg_partial; wp.bias_partial = eng->wg_bias_partial;
            rp.partial = eng->wg_partial; rp.bias_partial = eng->wg_bias_partial;
            rp.dw = eng->b_grads + w_off; rp.db = eng->b_grads + b_off;
            wp.zero = eng->zero_page; wp.ablate = eng->ablate; wp.dbg = eng->dbg;
            hipError_t err = prof_launch(eng, 1, flop, bytes, s, [&]() { return eng->math >= 3 ? launch_wgrad_split(eng->math, eng->ablate, wp, s) : launch_wgrad_mfma(wp, s); });
            if (err != hipSuccess) return err;
            return launch_wgrad_reduce(rp, s);
        });
    }

    // A dense block's FIFTEEN (X_j, G_n) pairs (conv n + 1 reads x_0 .. x_n: j <= n) as ONE pair-list launch of the f16x3 kernel:
    // 16 parts x 15 slots = 240 workgroups (+ a 17th part on the 16 spare CUs), the 15 of a part side by side on one XCD walking the same tiles, so that every X and
    // every G tile comes from HBM once and from that XCD's L2 for the other pairs (10 planes read instead of the 20 of five
    // launches per G).  The step runs at the package power cap and HBM traffic is a third of a launch's dynamic energy
    // (DESIGN.md 6.5): bytes are what there is to save; the 16 CUs the launch leaves idle draw next to nothing.
    // Slots are conv-major (conv 5 first), so conv n's blocks are contiguous and each conv gets its own fixed-order reduce.
    static int block_parts_per_xcd(int ncu) { const int m = ncu / 120; return m > 10 ? 10 : m; }     // ((8 m + 1) x 15 partial panels must fit wg_partial's 256 x 5)
    void wgrad_block_launch(std::vector<Launch>& ops, const float* const xpl[5], const float* const Gp[6], const ConvW* cw, float gscale)
    {
        xsd_engine* eng = e;
        WgradParams wp;
        memset(&wp, 0, sizeof(wp));
        wp.B = B; wp.H = H; wp.W = W;
        wp.tilesX = (wp.W + TILE_W - 1) / TILE_W; wp.tilesY = (wp.H + TILE_H - 1) / TILE_H;
        // m parts per XCD, each 15 workgroups side by side: m = CUs / (8 x 15) (MI355X: 256 CUs -> 16 parts = 240 workgroups; the
        // kernel's decode takes any nparts = 8 m or 8 m + 1; the caller falls back to one launch per G when m = 0)
        const int m = block_parts_per_xcd(e->ncu);
        wp.n_in = 5; wp.n_g = 5; wp.nparts = 8 * m; wp.npairs = 15;
        // one more part on the CUs the 8 m x 15 workgroups leave, when those can hold its 15 slots two per XCD (>= 16 spare CUs;
        // MI355X: exactly 16): no L2 sharing for it, no idle CU
        // (XSD_WGRAD_TAIL=0 (include/xsd.h): 8 m parts; same device 124.7 -> 125.2 tiles/s, the kernel -1.9 %)
#ifdef XSD_TEST_HOOKS      // only the hooks variant of the library (make hooks; include/xsd.h) reads the environment here
        static const int tail = getenv("XSD_WGRAD_TAIL") ? atoi(getenv("XSD_WGRAD_TAIL")) : 1;
#else
        constexpr int tail = 1;
#endif
        if (tail && e->ncu - 120 * m >= 16) wp.nparts = 8 * m + 1;
        std::vector<Launch> pre;
        for (int i = 0; i < 5; ++i) {
            wp.x[i] = std_in(xpl[i], 0); wp.g[i] = std_in(Gp[i + 1], 0);      // g[n] = G_{n+1}, the gradient at conv n+1's output
            if (e->math == 4) {
                wp.amax_x[i] = slot_of(wp.x[i], wp.H, wp.W, pre);
                wp.amax_g[i] = slot_of(wp.g[i], wp.H, wp.W, pre);
            }
        }
        int first[5], slot = 0;
        for (int n = 4; n >= 0; --n) {
            first[n] = slot;
            for (int j = 0; j <= n; ++j, ++slot) { wp.pair_j |= (unsigned long long)j << (4 * slot); wp.pair_n |= (unsigned long long)n << (4 * slot); }
        }
        WgradReduceParams rp[5];
        long long w_off[5], b_off[5];
        for (int n = 0; n < 5; ++n) {
            memset(&rp[n], 0, sizeof(rp[n]));
            rp[n].nparts = wp.nparts; rp[n].n_in = n + 1; rp[n].n_g = 1; rp[n].cin_total = cw[n].cin; rp[n].cout_total = cw[n].cout;
            rp[n].shuffle = 0; rp[n].scale = n == 4 ? gscale : 1.f;
            rp[n].part_stride = (long long)wp.npairs * PANEL_FLOATS; rp[n].bias_stride = wp.n_g * 32;
            w_off[n] = cw[n].w_off; b_off[n] = cw[n].b_off;
        }
        const double px = (double)wp.B * wp.H * wp.W;
        const double flop = 2.0 * 9 * 32 * 32 * 15 * px;
        const double bytes = 128.0 * 10 * px;      // SURVEY 8(d) rule: every X plane and every G plane once
        std::vector<long long> wo(w_off, w_off + 5), bo(b_off, b_off + 5);
        std::vector<WgradReduceParams> rps(rp, rp + 5);
        std::vector<int> fst(first, first + 5);
        ops.push_back([eng, wp, rps, fst, pre, wo, bo, flop, bytes](hipStream_t s) mutable {
            for (auto& f : pre) { hipError_t perr = f(s); if (perr != hipSuccess) return perr; }
            wp.partial = eng->wg_partial; wp.bias_partial = eng->wg_bias_partial;
            wp.zero = eng->zero_page; wp.ablate = eng->ablate; wp.dbg = eng->dbg;
            hipError_t err = prof_launch(eng, 1, flop, bytes, s, [&]() { return launch_wgrad_split(eng->math, eng->ablate, wp, s); });
            if (err != hipSuccess) return err;
            WgradReduceBatch rb;      // the five convs' fixed-order reductions as one launch (bitwise what five launches gave)
            memset(&rb, 0, sizeof(rb));
            rb.n = 5;
            for (int n = 0; n < 5; ++n) {
                WgradReduceParams r = rps[n];
                r.partial = eng->wg_partial + (long long)fst[n] * PANEL_FLOATS; r.bias_partial = eng->wg_bias_partial + n * 32;
                r.dw = eng->b_grads + wo[n]; r.db = eng->b_grads + bo[n];
                rb.r[n] = r;
            }
            return launch_wgrad_reduce_multi(rb, s);
        });
    }

    // ---------------------------------------------------------------------------------------------------------
    void build()
    {
        if (e->planes > 1 || e->cfg.in_channels != 1 || e->cfg.out_channels != 1) { build_multi(); return; }
        xsd_engine* eng = e;
        const int blocks = e->cfg.num_res_blocks;
        const bool sr = e->cfg.kind == XSD_KIND_SR;
        const int nup = sr ? e->cfg.num_upsample : 0;
        std::vector<Launch>& F = e->fwd_ops;
        F.clear();
        e->bwd_stages.assign(blocks + 2, {});

        struct RdbAct { float* xin; float* xs[4]; float* out; unsigned short* xb[4]; };
        const bool rdb_bits = train && e->math >= 3;   // compact lrelu' masks of the dense blocks' activations (role-split conv epilogues)
        std::vector<RdbAct> acts(blocks * 3);
        std::vector<float*> rin(blocks + 1);

        // ---- forward ------------------------------------------------------------------------------------------
        float* fea = alloc(0);
        { // conv_first (generator_rrdb.py:67)
            EdgeExpandParams p; memset(&p, 0, sizeof(p));
            p.B = B; p.H = H; p.W = W; p.out = fea; p.w = e->pk_edge + 0; p.mslope = 1.f;
            p.amax = report_slot(fea, 0);
            const long long boff = e->first_b;
            F.push_back([eng, p, boff](hipStream_t s) mutable { p.s = eng->b_x; p.bias = eng->params + boff; return run_edge_expand(eng, p, s); });
        }
        float* cur = fea;
        for (int i = 0; i < blocks; ++i) {
            rin[i] = cur;
            for (int r = 0; r < 3; ++r) {
                RdbAct& a = acts[i * 3 + r];
                a.xin = cur;
                ConvParams rp[5];
                for (int c = 0; c < 5; ++c) {
                    const ConvW& cw = e->rdb[(i * 3 + r) * 5 + c];
                    ConvParams p = conv_base(0);
                    p.n_in = c + 1; p.n_out = 1;
                    p.in[0] = std_in(a.xin, 0);
                    for (int k = 0; k < c; ++k) p.in[k + 1] = std_in(a.xs[k], 0);
                    p.wpanel = fwdp(cw.fwd_off);
                    float* o = alloc(0);
                    std_out(p.out[0], o, 0);
                    if (c < 4) {                                            // rrdb_blocks.py:38-52
                        p.out[0].slope = 0.2f; a.xs[c] = o;
                        a.xb[c] = rdb_bits ? reinterpret_cast<unsigned short*>(alloc1(0)) : nullptr;
                        p.out[0].bits_out = a.xb[c];
                    }
                    else {
                        p.out[0].a1 = 0.2f; p.out[0].e1 = a.xin; p.out[0].s1 = 1.f;           // x5*0.2 + x   (:54)
                        if (r == 2) { p.out[0].a2 = 0.2f; p.out[0].e2 = rin[i]; p.out[0].s2 = 1.f; } // out*0.2 + x (:70)
                        a.out = o;
                    }
                    rp[c] = p;
                }
                // Sweep the dense block in batch chunks: with <= 1-2 images per sweep the block's six planes
                // (6 x 33.5 MB per 512^2 image) stay in the Infinity Cache between conv_k and conv_{k+1..5}.
                const int cb = (e->chunk > 0 && e->chunk < B) ? e->chunk : B;
                for (int b0 = 0; b0 < B; b0 += cb)
                    for (int c = 0; c < 5; ++c)
                        F.push_back(conv_launch(slice(rp[c], b0, std::min(cb, B - b0)), true, e->rdb[(i * 3 + r) * 5 + c].b_off));
                for (int k = 0; k < 4; ++k) release(a.xs[k], 0);
                if (r > 0) release(a.xin, 0);
                cur = a.out;
            }
            if (i > 0) release(rin[i], 0);
        }
        rin[blocks] = cur;
        float* T = alloc(0);
        { // fea + trunk_conv(rrdb(fea)) (generator_rrdb.py:68-69)
            ConvParams p = conv_base(0);
            p.n_in = 1; p.n_out = 1; p.in[0] = std_in(cur, 0); p.wpanel = fwdp(e->trunk.fwd_off);
            std_out(p.out[0], T, 0);
            p.out[0].e1 = fea; p.out[0].s1 = 1.f;
            F.push_back(conv_launch(p, true, e->trunk.b_off));
        }
        if (blocks > 0) release(cur, 0);
        release(fea, 0);

        std::vector<float*> U(nup, nullptr);
        float* H1 = nullptr;
        float* pre = nullptr;
        const int lo = nup; // output level
        if (sr) {
            const float* feat = T;
            for (int u = 0; u < nup; ++u) { // upsampling: conv 32->128, LeakyReLU(0.01), PixelShuffle(2) (generator_rrdb.py:93-99)
                U[u] = alloc(u + 1);
                ConvParams p = conv_base(u);
                p.n_in = 1; p.n_out = 4; p.in[0] = std_in(feat, u); p.wpanel = fwdp(e->up[u].fwd_off);
                for (int n = 0; n < 4; ++n) { shuf_out(p.out[n], U[u], u, n); p.out[n].slope = 0.01f; }
                p.bias = e->pk_sbias + e->up[u].sbias_off;
                F.push_back(conv_launch(p, false, 0));
                alias_shuffled(U[u], u);
                release(const_cast<float*>(feat), u);
                feat = U[u];
            }
            H1 = alloc(lo);
            { // lrelu(HRconv(fea)) (generator_rrdb.py:107)
                ConvParams p = conv_base(lo);
                p.n_in = 1; p.n_out = 1; p.in[0] = std_in(feat, lo); p.wpanel = fwdp(e->hr.fwd_off);
                std_out(p.out[0], H1, lo); p.out[0].slope = 0.2f;
                F.push_back(conv_launch(p, true, e->hr.b_off));
            }
            if (nup > 0) release(U[nup - 1], lo); else release(T, 0);
        }
        pre = train ? alloc1(lo) : nullptr;
        { // conv_last (+x for DN) + clamp, clamp (generator_rrdb.py:107-108,132-135; model.py:49)
            EdgeReduceParams p; memset(&p, 0, sizeof(p));
            p.B = B; p.H = H << lo; p.W = W << lo; p.f = sr ? H1 : T; p.w = e->pk_edge + 2 * 288; p.pre = pre; p.clamp01 = 1;
            const long long boff = e->last_b;
            F.push_back([eng, p, boff, sr](hipStream_t s) mutable {
                p.bias = eng->params + boff; p.skip = sr ? nullptr : eng->b_x; p.y = eng->b_y; return run_edge_reduce(eng, p, s);
            });
        }
        if (!train) return;

        // ---- backward -----------------------------------------------------------------------------------------
        e->amax_bwd_first = e->amax_used;      // every slot from here on is first written by a backward launch
        float* dpre = alloc1(lo);
        float* dT = alloc(0);
        { // stage 0: output head
            std::vector<Launch>& S = e->bwd_stages[0];
            const long long npx = (long long)B * (H << lo) * (W << lo);
            S.push_back([eng, pre, dpre, npx](hipStream_t s) { return prof_launch(eng, PK_CLAMP_BWD, 0.0, 12.0 * npx, s, [&]() { return launch_clamp_bwd(pre, eng->b_dy, dpre, npx, s); }); });
            { // conv_last weight grad
                EdgeWgradParams p; memset(&p, 0, sizeof(p));
                p.B = B; p.H = H << lo; p.W = W << lo; p.f = sr ? H1 : T; p.s = dpre; p.nblocks = EDGE_WGRAD_BLOCKS;
                const long long wo = e->last_w, bo = e->last_b;
                S.push_back([eng, p, wo, bo](hipStream_t s) mutable {
                    p.partial = eng->edge_partial; return run_edge_wgrad(eng, p, 1, eng->b_grads + wo, eng->b_grads + bo, s);
                });
            }
            if (!sr) {
                EdgeExpandParams p; memset(&p, 0, sizeof(p));
                p.B = B; p.H = H; p.W = W; p.s = dpre; p.w = e->pk_edge + 3 * 288; p.out = dT; p.mslope = 1.f;
                p.amax = report_slot(dT, 0);
                S.push_back([eng, p](hipStream_t s) { return run_edge_expand(eng, p, s); });
            } else {
                float* GH = alloc(lo);
                { // d(H1) masked by lrelu'(0.2)
                    EdgeExpandParams p; memset(&p, 0, sizeof(p));
                    p.B = B; p.H = H << lo; p.W = W << lo; p.s = dpre; p.w = e->pk_edge + 3 * 288; p.out = GH; p.mask = H1; p.mslope = 0.2f;
                    p.amax = report_slot(GH, lo);
                    S.push_back([eng, p](hipStream_t s) { return run_edge_expand(eng, p, s); });
                }
                const float* hr_in = nup > 0 ? U[nup - 1] : T;
                wgrad_launch(S, lo, {std_in(hr_in, lo)}, {std_in(GH, lo)}, e->hr, 1.f);
                float* G = nup > 0 ? alloc(lo) : dT;
                { // HRconv input gradient (masked by the upsample LeakyReLU(0.01) when it feeds a pixel-shuffle)
                    ConvParams p = conv_base(lo);
                    p.n_in = 1; p.n_out = 1; p.in[0] = std_in(GH, lo); p.wpanel = bwdp(e->hr.bwd_off);
                    std_out(p.out[0], G, lo);
                    if (nup > 0) { p.out[0].mask = U[nup - 1]; p.out[0].mslope = 0.01f; }
                    S.push_back(conv_launch(p, false, 0));
                }
                for (int u = nup - 1; u >= 0; --u) {
                    const float* xin = u > 0 ? U[u - 1] : T;
                    std::vector<PlaneIn> gs;
                    for (int n = 0; n < 4; ++n) gs.push_back(shuf_in(G, u, n));
                    wgrad_launch(S, u, {std_in(xin, u)}, gs, e->up[u], 1.f);
                    float* Gn = u > 0 ? alloc(u) : dT;
                    ConvParams p = conv_base(u);
                    p.n_in = 4; p.n_out = 1;
                    for (int n = 0; n < 4; ++n) p.in[n] = gs[n];
                    p.wpanel = bwdp(e->up[u].bwd_off);
                    std_out(p.out[0], Gn, u);
                    if (u > 0) { p.out[0].mask = U[u - 1]; p.out[0].mslope = 0.01f; }
                    S.push_back(conv_launch(p, false, 0));
                    G = Gn;
                }
            }
            // trunk_conv
            wgrad_launch(S, 0, {std_in(rin[blocks], 0)}, {std_in(dT, 0)}, e->trunk, 1.f);
        }
        float* dR = alloc(0);
        {
            ConvParams p = conv_base(0);
            p.n_in = 1; p.n_out = 1; p.in[0] = std_in(dT, 0); p.wpanel = bwdp(e->trunk.bwd_off);
            std_out(p.out[0], dR, 0);
            e->bwd_stages[0].push_back(conv_launch(p, false, 0));
        }
        float* dS[5] = {nullptr, alloc(0), alloc(0), alloc(0), alloc(0)};
        for (int i = blocks - 1; i >= 0; --i) {
            std::vector<Launch>& S = e->bwd_stages[blocks - i];
            float* dOut = dR;
            for (int r = 2; r >= 0; --r) {
                const RdbAct& a = acts[i * 3 + r];
                const float gscale = r == 2 ? 0.04f : 0.2f;
                const ConvW* cw = &e->rdb[(i * 3 + r) * 5];
                dS[0] = alloc(0);
                const float* xpl[5] = {a.xin, a.xs[0], a.xs[1], a.xs[2], a.xs[3]};
                // Input-gradients as K-loops (no read-modify-write): dS_j = sum_{c>j} conv^T_c[j](G_c) in ONE launch over
                // the planes G_5 = dOut, G_4..G_{j+1}; its epilogue adds the residual-path terms (j = 0) or applies
                // lrelu'(x_j) (j >= 1), which makes dS_j the G_j of conv_j.  conv5's 0.2 / 0.04 factor lives in its
                // transposed panels (PackDesc.bwd_scale); its weight gradient uses the unscaled dOut and scales in the reduce.
                const float* Gp[6] = {nullptr, dS[1], dS[2], dS[3], dS[4], dOut}; // G_c, c = 1..5
                // one pair-list weight-gradient launch per dense block (f16x3 kernel) once every G exists, i.e. in front of dS_0
                // (XSD_WGRAD_BLOCK=0 restores one launch per G for same-library A/Bs: 122.6 -> 125.6 tiles/s on one device, profiles/r04_ab_wgrad_block_launch.txt)
#ifdef XSD_TEST_HOOKS
                static const bool block_wgrad = getenv("XSD_WGRAD_BLOCK") ? atoi(getenv("XSD_WGRAD_BLOCK")) != 0 : true;
#else
                constexpr bool block_wgrad = true;
#endif
                const bool mega = block_wgrad && e->math >= 3 && block_parts_per_xcd(e->ncu) >= 1;      // both role-split weight-gradient kernels take pair lists; fewer than 120 CUs: one launch per G
                for (int c = 4; c >= 0; --c) { // conv index c (0-based) = conv_{c+1}
                    std::vector<PlaneIn> xs;
                    for (int kk = 0; kk <= c; ++kk) xs.push_back(std_in(xpl[kk], 0));
                    if (!mega) wgrad_launch(S, 0, xs, {std_in(Gp[c + 1], 0)}, cw[c], c == 4 ? gscale : 1.f);
                    else if (c == 0) wgrad_block_launch(S, xpl, Gp, cw, gscale);
                    // now every G needed by dS_c exists: G_5 .. G_{c+1}
                    const int j = c;
                    ConvParams p = conv_base(0);
                    p.n_in = 5 - j; p.n_out = 1;
                    for (int i = 0; i < 5 - j; ++i) { // step i reads G_{5-i}
                        const int cc = 5 - i;          // 1-based conv whose gradient plane this is
                        p.in[i] = std_in(Gp[cc], 0);
                        p.wstep[i] = bwdp(cw[cc - 1].bwd_off + (long long)j * PANEL_FLOATS);
                    }
                    p.wpanel = p.wstep[0];
                    OutDesc& o = p.out[0];
                    std_out(o, dS[j], 0);
                    if (j == 0) {
                        o.e1 = dOut; o.s1 = r == 2 ? 0.2f : 1.f;
                        if (r == 0) { o.e2 = dR; o.s2 = 1.f; if (i == 0) { o.e3 = dT; o.s3 = 1.f; } }
                    } else { o.mask = xpl[j]; o.mslope = 0.2f; o.bits_in = a.xb[j - 1]; }
                    S.push_back(conv_launch(p, false, 0));
                }
                if (dOut != dR) release(dOut, 0, true);
                dOut = dS[0];
            }
            release(dR, 0, true);
            dR = dOut;
        }
        { // last stage: conv_first weight grad and (optionally) dx
            std::vector<Launch>& S = e->bwd_stages[blocks + 1];
            float* dFea = dR; // when blocks == 0 this is d(rrdb out) and needs + dT; blocks >= 1 is enforced at create
            EdgeWgradParams p; memset(&p, 0, sizeof(p));
            p.B = B; p.H = H; p.W = W; p.f = dFea; p.nblocks = EDGE_WGRAD_BLOCKS;
            const long long wo = e->first_w, bo = e->first_b;
            S.push_back([eng, p, wo, bo](hipStream_t s) mutable {
                p.s = eng->b_x; p.partial = eng->edge_partial; return run_edge_wgrad(eng, p, 0, eng->b_grads + wo, eng->b_grads + bo, s);
            });
            EdgeReduceParams q; memset(&q, 0, sizeof(q));
            q.B = B; q.H = H; q.W = W; q.f = dFea; q.w = e->pk_edge + 1 * 288; q.clamp01 = 0;
            const float* skipg = sr ? nullptr : dpre;
            S.push_back([eng, q, skipg](hipStream_t s) mutable {
                if (!eng->b_dx) return hipSuccess;
                q.addto = skipg; q.y = eng->b_dx; return run_edge_reduce(eng, q, s);
            });
        }
    }

    // ---------------------------------------------------------------------------------------------------------
    // Wide nets and image channels: num_filters = 32 P (P = 1..8), 1..8 image channels -- the dense block's own default width is 64
    // (rrdb_blocks.py:23).  A feature tensor is P planes of 32 channels in torch.cat's channel order; a conv with 32 a
    // inputs and 32 b outputs is b output chunks, each ONE K-loop over the a input planes, cut into launches of <= 5 planes: the
    // first launch carries the bias, the later ones add to the plane it wrote (`accumulate` epilogue), the last one carries the
    // layer's epilogue (residuals, LeakyReLU, masks).  LeakyReLU' masks are read from the activation planes (the compact mask
    // words have no epilogue variant with `accumulate`).  Same kernels, same packed panels ([chunk][plane] / [plane][chunk]
    // order, pack_weights*_kernel), same backward stages and flat gradient layout as build().
    typedef std::vector<float*> Tensor;
    void build_multi()
    {
        xsd_engine* eng = e;
        const int P = e->planes;
        const int blocks = e->cfg.num_res_blocks;
        const bool sr = e->cfg.kind == XSD_KIND_SR;
        const int nup = sr ? e->cfg.num_upsample : 0;
        std::vector<Launch>& F = e->fwd_ops;
        F.clear();
        e->bwd_stages.assign(blocks + 2, {});
        const int CI = e->cfg.in_channels, CO = e->cfg.out_channels;    // image channels: the image-side layers run one (image channel, plane) at a time
        const float* edge = e->pk_edge;      // [first_fwd | first_bwd][in channel][plane][288], [last_fwd | last_bwd][out channel][plane][288]
        auto first_fwd = [&](int ch, int q) { return edge + 288 * (ch * P + q); };
        auto first_bwd = [&](int ch, int q) { return edge + 288 * (P * CI + ch * P + q); };
        auto last_fwd = [&](int co, int q) { return edge + 288 * (2 * P * CI + co * P + q); };
        auto last_bwd = [&](int co, int q) { return edge + 288 * (2 * P * CI + P * CO + co * P + q); };
        const long long HW = (long long)H * W;

        auto alloc_t = [&](int level) { Tensor t(P); for (auto& p : t) p = alloc(level); return t; };
        auto release_t = [&](const Tensor& t, int level, bool force = false) { for (float* p : t) release(p, level, force); };
        auto planes_of = [&](const Tensor& t, int level) { std::vector<PlaneIn> v; for (float* p : t) v.push_back(std_in(p, level)); return v; };
        auto out_desc = [&](float* p, int level) { OutDesc o = conv_base(level).out[0]; std_out(o, p, level); return o; };
        // one output plane: the K-loop over `ins` with the panels `pan` (one per input plane), <= 5 planes per launch
        auto kloop = [&](std::vector<Launch>& ops, int level, const std::vector<PlaneIn>& ins, const std::vector<const float*>& pan,
                         const OutDesc& fin, bool bias_from_params, long long bias_off, const float* bias_ptr) {
            const int n = (int)ins.size();
            for (int base = 0; base < n; base += 5) {
                const int cnt = std::min(5, n - base);
                const bool first = base == 0, last = base + cnt >= n;
                ConvParams p = conv_base(level);
                p.n_in = cnt; p.n_out = 1;
                for (int k = 0; k < cnt; ++k) { p.in[k] = ins[base + k]; p.wstep[k] = pan[base + k]; }
                p.wpanel = p.wstep[0];
                OutDesc& o = p.out[0];
                if (last) o = fin;
                else { o.p = fin.p; o.ps = fin.ps; o.rs = fin.rs; o.bs = fin.bs; o.a1 = fin.a1; }   // partial sums carry the final scale
                o.accumulate = first ? 0 : 1;
                if (first) p.bias = bias_ptr;
                ops.push_back(conv_launch(p, first && bias_from_params, bias_off));
            }
        };
        // weight gradient of one conv: per output chunk, the input planes in groups of <= 5
        auto wgrad_all = [&](std::vector<Launch>& ops, int level, const std::vector<PlaneIn>& xs, const Tensor& g, const ConvW& cw, float scale) {
            for (int q = 0; q < P; ++q)
                for (int base = 0; base < (int)xs.size(); base += 5) {
                    std::vector<PlaneIn> grp(xs.begin() + base, xs.begin() + std::min<size_t>(xs.size(), base + 5));
                    wgrad_launch(ops, level, grp, {std_in(g[q], level)}, cw, scale, base, q, 0);
                }
        };
        struct RdbActM { Tensor xin, xs[4], out; };
        std::vector<RdbActM> acts(blocks * 3);
        std::vector<Tensor> rin(blocks + 1);

        // ---- forward ------------------------------------------------------------------------------------------
        Tensor fea = alloc_t(0);
        for (int q = 0; q < P; ++q)   // conv_first (generator_rrdb.py:67), 32 output channels of one image channel per launch
            for (int ch = 0; ch < CI; ++ch) {
                EdgeExpandParams p; memset(&p, 0, sizeof(p));
                p.B = B; p.H = H; p.W = W; p.out = fea[q]; p.w = first_fwd(ch, q); p.mslope = 1.f; p.s_bs = CI * HW; p.accumulate = ch > 0;
                const long long boff = e->first_b + 32 * q, xo = ch * HW;
                F.push_back([eng, p, boff, xo, ch](hipStream_t s) mutable {
                    p.s = eng->b_x + xo; p.bias = ch == 0 ? eng->params + boff : nullptr; return run_edge_expand(eng, p, s); });
            }
        Tensor cur = fea;
        for (int i = 0; i < blocks; ++i) {
            rin[i] = cur;
            for (int r = 0; r < 3; ++r) {
                RdbActM& a = acts[i * 3 + r];
                a.xin = cur;
                for (int c = 0; c < 5; ++c) {
                    const ConvW& cw = e->rdb[(i * 3 + r) * 5 + c];
                    const int ns = (c + 1) * P;
                    std::vector<PlaneIn> ins;
                    for (int t = 0; t <= c; ++t) for (float* pl : (t == 0 ? a.xin : a.xs[t - 1])) ins.push_back(std_in(pl, 0));
                    Tensor o = alloc_t(0);
                    for (int q = 0; q < P; ++q) {
                        std::vector<const float*> pan;
                        for (int k = 0; k < ns; ++k) pan.push_back(fwdp(cw.fwd_off + ((long long)q * ns + k) * PANEL_FLOATS));
                        OutDesc fin = out_desc(o[q], 0);
                        if (c < 4) fin.slope = 0.2f;                                                   // rrdb_blocks.py:38-52
                        else {
                            fin.a1 = 0.2f; fin.e1 = a.xin[q]; fin.s1 = 1.f;                            // x5*0.2 + x   (:54)
                            if (r == 2) { fin.a2 = 0.2f; fin.e2 = rin[i][q]; fin.s2 = 1.f; }           // out*0.2 + x (:70)
                        }
                        kloop(F, 0, ins, pan, fin, true, cw.b_off + 32 * q, nullptr);
                    }
                    if (c < 4) a.xs[c] = o; else a.out = o;
                }
                for (int k = 0; k < 4; ++k) release_t(a.xs[k], 0);
                if (r > 0) release_t(a.xin, 0);
                cur = a.out;
            }
            if (i > 0) release_t(rin[i], 0);
        }
        rin[blocks] = cur;
        Tensor T = alloc_t(0);
        for (int q = 0; q < P; ++q) { // fea + trunk_conv(rrdb(fea)) (generator_rrdb.py:68-69)
            std::vector<const float*> pan;
            for (int k = 0; k < P; ++k) pan.push_back(fwdp(e->trunk.fwd_off + ((long long)q * P + k) * PANEL_FLOATS));
            OutDesc fin = out_desc(T[q], 0);
            fin.e1 = fea[q]; fin.s1 = 1.f;
            kloop(F, 0, planes_of(cur, 0), pan, fin, true, e->trunk.b_off + 32 * q, nullptr);
        }
        release_t(cur, 0);
        release_t(fea, 0);

        std::vector<Tensor> U(nup);
        Tensor H1;
        const int lo = nup; // output level
        if (sr) {
            Tensor feat = T;
            for (int u = 0; u < nup; ++u) { // upsampling: conv 32P -> 128P, LeakyReLU(0.01), PixelShuffle(2) (generator_rrdb.py:93-99)
                U[u] = alloc_t(u + 1);
                for (int q = 0; q < P; ++q)
                    for (int sub = 0; sub < 4; ++sub) {           // chunk n = sub * P + q holds the channels 4 (32 q + c) + sub
                        const int n = sub * P + q;
                        std::vector<const float*> pan;
                        for (int k = 0; k < P; ++k) pan.push_back(fwdp(e->up[u].fwd_off + ((long long)n * P + k) * PANEL_FLOATS));
                        OutDesc fin = conv_base(u).out[0];
                        shuf_out(fin, U[u][q], u, sub);
                        fin.slope = 0.01f;
                        kloop(F, u, planes_of(feat, u), pan, fin, false, 0, e->pk_sbias + e->up[u].sbias_off + 32 * n);
                    }
                release_t(feat, u);
                feat = U[u];
            }
            H1 = alloc_t(lo);
            for (int q = 0; q < P; ++q) { // lrelu(HRconv(fea)) (generator_rrdb.py:107)
                std::vector<const float*> pan;
                for (int k = 0; k < P; ++k) pan.push_back(fwdp(e->hr.fwd_off + ((long long)q * P + k) * PANEL_FLOATS));
                OutDesc fin = out_desc(H1[q], lo);
                fin.slope = 0.2f;
                kloop(F, lo, planes_of(feat, lo), pan, fin, true, e->hr.b_off + 32 * q, nullptr);
            }
            release_t(feat, lo);
        }
        const long long HWo = (long long)(H << lo) * (W << lo);
        float* pre = train ? alloc_img(lo, CO) : nullptr;
        const Tensor& headf = sr ? H1 : T;
        // conv_last (+x for DN) + clamp, clamp (generator_rrdb.py:107-108,132-135; model.py:49): per output channel the planes' partial
        // sums chained in place through y, the last launch adds the skip and clamps
        for (int co = 0; co < CO; ++co)
            for (int q = 0; q < P; ++q) {
                EdgeReduceParams p; memset(&p, 0, sizeof(p));
                p.B = B; p.H = H << lo; p.W = W << lo; p.f = headf[q]; p.w = last_fwd(co, q); p.y_bs = CO * HWo; p.skip_bs = CI * HW;
                const bool first = q == 0, last = q == P - 1;
                const long long boff = e->last_b + co, yo = co * HWo, xo = (CI == 1 ? 0 : co) * HW;   // DN: x broadcasts when it has one channel
                F.push_back([eng, p, first, last, boff, yo, xo, pre, sr](hipStream_t s) mutable {
                    p.y = eng->b_y + yo;
                    if (first) p.bias = eng->params + boff; else p.addto = p.y;
                    if (last) { p.skip = sr ? nullptr : eng->b_x + xo; p.pre = pre ? pre + yo : nullptr; p.clamp01 = 1; }
                    return run_edge_reduce(eng, p, s);
                });
            }
        if (!train) return;

        // ---- backward -----------------------------------------------------------------------------------------
        e->amax_bwd_first = e->amax_used;      // every slot from here on is first written by a backward launch
        float* dpre = alloc_img(lo, CO);
        Tensor dT = alloc_t(0);
        { // stage 0: output head
            std::vector<Launch>& S = e->bwd_stages[0];
            const long long npx = (long long)B * CO * HWo;
            S.push_back([eng, pre, dpre, npx](hipStream_t s) { return prof_launch(eng, PK_CLAMP_BWD, 0.0, 12.0 * npx, s, [&]() { return launch_clamp_bwd(pre, eng->b_dy, dpre, npx, s); }); });
            for (int co = 0; co < CO; ++co)
            for (int q = 0; q < P; ++q) { // conv_last weight grad, 32 input channels of one output channel per launch (the bias gradient is the same sum every time)
                EdgeWgradParams p; memset(&p, 0, sizeof(p));
                p.B = B; p.H = H << lo; p.W = W << lo; p.f = headf[q]; p.s = dpre + co * HWo; p.s_bs = CO * HWo; p.nblocks = EDGE_WGRAD_BLOCKS;
                const long long wo = e->last_w + ((long long)co * 32 * P + 32 * q) * 9, bo = e->last_b + co;
                S.push_back([eng, p, wo, bo](hipStream_t s) mutable {
                    p.partial = eng->edge_partial; return run_edge_wgrad(eng, p, 1, eng->b_grads + wo, eng->b_grads + bo, s);
                });
            }
            if (!sr) {
                for (int q = 0; q < P; ++q)
                    for (int co = 0; co < CO; ++co) {
                        EdgeExpandParams p; memset(&p, 0, sizeof(p));
                        p.B = B; p.H = H; p.W = W; p.s = dpre + co * HWo; p.s_bs = CO * HWo; p.w = last_bwd(co, q); p.out = dT[q]; p.mslope = 1.f;
                        p.accumulate = co > 0;
                        S.push_back([eng, p](hipStream_t s) { return run_edge_expand(eng, p, s); });
                    }
            } else {
                Tensor GH = alloc_t(lo);
                for (int q = 0; q < P; ++q)   // d(H1) masked by lrelu'(0.2): the output channels' contributions summed, the mask on the last
                    for (int co = 0; co < CO; ++co) {
                        EdgeExpandParams p; memset(&p, 0, sizeof(p));
                        p.B = B; p.H = H << lo; p.W = W << lo; p.s = dpre + co * HWo; p.s_bs = CO * HWo; p.w = last_bwd(co, q); p.out = GH[q];
                        p.accumulate = co > 0; p.mslope = 1.f;
                        if (co == CO - 1) { p.mask = H1[q]; p.mslope = 0.2f; }
                        S.push_back([eng, p](hipStream_t s) { return run_edge_expand(eng, p, s); });
                    }
                const Tensor& hr_in = nup > 0 ? U[nup - 1] : T;
                wgrad_all(S, lo, planes_of(hr_in, lo), GH, e->hr, 1.f);
                Tensor G = nup > 0 ? alloc_t(lo) : dT;
                for (int pl = 0; pl < P; ++pl) { // HRconv input gradient (masked by the upsample LeakyReLU(0.01) when it feeds a pixel-shuffle)
                    std::vector<const float*> pan;
                    for (int k = 0; k < P; ++k) pan.push_back(bwdp(e->hr.bwd_off + ((long long)pl * P + k) * PANEL_FLOATS));
                    OutDesc fin = out_desc(G[pl], lo);
                    if (nup > 0) { fin.mask = U[nup - 1][pl]; fin.mslope = 0.01f; }
                    kloop(S, lo, planes_of(GH, lo), pan, fin, false, 0, nullptr);
                }
                for (int u = nup - 1; u >= 0; --u) {
                    const Tensor& xin = u > 0 ? U[u - 1] : T;
                    for (int q = 0; q < P; ++q) {                 // dW of the shuffle conv: the four sub-pixel views of gradient plane q against every input plane
                        std::vector<PlaneIn> gs;
                        for (int sub = 0; sub < 4; ++sub) gs.push_back(shuf_in(G[q], u, sub));
                        for (int j = 0; j < P; ++j) wgrad_launch(S, u, {std_in(xin[j], u)}, gs, e->up[u], 1.f, j, 0, q);
                    }
                    Tensor Gn = u > 0 ? alloc_t(u) : dT;
                    for (int pl = 0; pl < P; ++pl) {
                        std::vector<PlaneIn> ins;
                        std::vector<const float*> pan;
                        for (int sub = 0; sub < 4; ++sub)
                            for (int q = 0; q < P; ++q) {
                                ins.push_back(shuf_in(G[q], u, sub));
                                pan.push_back(bwdp(e->up[u].bwd_off + ((long long)pl * (4 * P) + (sub * P + q)) * PANEL_FLOATS));
                            }
                        OutDesc fin = out_desc(Gn[pl], u);
                        if (u > 0) { fin.mask = U[u - 1][pl]; fin.mslope = 0.01f; }
                        kloop(S, u, ins, pan, fin, false, 0, nullptr);
                    }
                    G = Gn;
                }
            }
            // trunk_conv
            wgrad_all(S, 0, planes_of(rin[blocks], 0), dT, e->trunk, 1.f);
        }
        Tensor dR = alloc_t(0);
        for (int pl = 0; pl < P; ++pl) {
            std::vector<const float*> pan;
            for (int k = 0; k < P; ++k) pan.push_back(bwdp(e->trunk.bwd_off + ((long long)pl * P + k) * PANEL_FLOATS));
            kloop(e->bwd_stages[0], 0, planes_of(dT, 0), pan, out_desc(dR[pl], 0), false, 0, nullptr);
        }
        Tensor dS[5] = {Tensor(), alloc_t(0), alloc_t(0), alloc_t(0), alloc_t(0)};
        for (int i = blocks - 1; i >= 0; --i) {
            std::vector<Launch>& S = e->bwd_stages[blocks - i];
            Tensor dOut = dR;
            for (int r = 2; r >= 0; --r) {
                const RdbActM& a = acts[i * 3 + r];
                const float gscale = r == 2 ? 0.04f : 0.2f;
                const ConvW* cw = &e->rdb[(i * 3 + r) * 5];
                dS[0] = alloc_t(0);
                const Tensor* xpl[5] = {&a.xin, &a.xs[0], &a.xs[1], &a.xs[2], &a.xs[3]};
                // as in build(): dS_j = sum_{c>j} conv^T_c[j](G_c) as ONE K-loop over the planes of G_5 = dOut, G_4 .. G_{j+1}
                const Tensor* Gp[6] = {nullptr, &dS[1], &dS[2], &dS[3], &dS[4], &dOut}; // G_c, c = 1..5
                for (int c = 4; c >= 0; --c) { // conv index c (0-based) = conv_{c+1}
                    std::vector<PlaneIn> xs;
                    for (int t = 0; t <= c; ++t) for (float* pl : *xpl[t]) xs.push_back(std_in(pl, 0));
                    wgrad_all(S, 0, xs, *Gp[c + 1], cw[c], c == 4 ? gscale : 1.f);
                    const int j = c;
                    for (int pl = 0; pl < P; ++pl) {
                        std::vector<PlaneIn> ins;
                        std::vector<const float*> pan;
                        for (int ii = 0; ii < 5 - j; ++ii) {
                            const int cc = 5 - ii;          // 1-based conv whose gradient planes these are; its input plane index of (tensor j, plane pl) is j P + pl
                            for (int q = 0; q < P; ++q) {
                                ins.push_back(std_in((*Gp[cc])[q], 0));
                                pan.push_back(bwdp(cw[cc - 1].bwd_off + (((long long)j * P + pl) * P + q) * PANEL_FLOATS));
                            }
                        }
                        OutDesc fin = out_desc(dS[j][pl], 0);
                        if (j == 0) {
                            fin.e1 = dOut[pl]; fin.s1 = r == 2 ? 0.2f : 1.f;
                            if (r == 0) { fin.e2 = dR[pl]; fin.s2 = 1.f; if (i == 0) { fin.e3 = dT[pl]; fin.s3 = 1.f; } }
                        } else { fin.mask = (*xpl[j])[pl]; fin.mslope = 0.2f; }
                        kloop(S, 0, ins, pan, fin, false, 0, nullptr);
                    }
                }
                if (dOut[0] != dR[0]) release_t(dOut, 0, true);
                dOut = dS[0];
            }
            release_t(dR, 0, true);
            dR = dOut;
        }
        { // last stage: conv_first weight grad and (optionally) dx
            std::vector<Launch>& S = e->bwd_stages[blocks + 1];
            const Tensor dFea = dR;
            for (int ch = 0; ch < CI; ++ch)
                for (int q = 0; q < P; ++q) {
                    EdgeWgradParams p; memset(&p, 0, sizeof(p));
                    p.B = B; p.H = H; p.W = W; p.f = dFea[q]; p.nblocks = EDGE_WGRAD_BLOCKS; p.s_bs = CI * HW;
                    const long long wo = e->first_w + ((long long)32 * q * CI + ch) * 9, bo = e->first_b + 32 * q, xo = ch * HW;
                    const int cstride = 9 * CI;
                    S.push_back([eng, p, wo, bo, xo, cstride](hipStream_t s) mutable {
                        p.s = eng->b_x + xo; p.partial = eng->edge_partial;
                        return run_edge_wgrad(eng, p, 0, eng->b_grads + wo, eng->b_grads + bo, s, cstride);
                    });
                }
            const bool bcast = !sr && CI == 1 && CO > 1;  // DN with a one-channel x added to every output channel: dx += sum over the channels of dpre
            const float* skipg = (sr || bcast) ? nullptr : dpre;     // DN: d(out + x)/dx per channel
            // dx per image channel: the planes' partial sums chained in place (all launches skipped when the caller wants no dx)
            for (int ch = 0; ch < CI; ++ch)
                for (int q = 0; q < P; ++q) {
                    EdgeReduceParams p; memset(&p, 0, sizeof(p));
                    p.B = B; p.H = H; p.W = W; p.f = dFea[q]; p.w = first_bwd(ch, q); p.y_bs = CI * HW; p.skip_bs = CO * HW;
                    const bool first = q == 0, last = q == P - 1;
                    const long long xo = ch * HW;
                    S.push_back([eng, p, first, last, xo, skipg](hipStream_t s) mutable {
                        if (!eng->b_dx) return hipSuccess;
                        p.y = eng->b_dx + xo;
                        if (!first) p.addto = p.y;
                        if (last && skipg) p.skip = skipg + xo;
                        return run_edge_reduce(eng, p, s);
                    });
                }
            if (bcast) {
                const int nb = B, nc = CO;
                S.push_back([eng, dpre, nb, nc, HW](hipStream_t s) { return eng->b_dx ? launch_add_channels(eng->b_dx, dpre, nc, HW, nb, s) : hipSuccess; });
            }
        }
    }
};

static int ensure_plan(xsd_engine* e, int B, int H, int W, bool train)
{
    if (e->pB == B && e->pH == H && e->pW == W && e->ptrain == (int)train) return XSD_OK;
    e->pB = e->pH = e->pW = 0; e->ptrain = -1; e->fwd_saved = false;
    Builder sizing(e, B, H, W, train, 0);
    e->amax_used = 2;
    sizing.build();
    const size_t need = sizing.peak + 256;
    if (need > e->ws_bytes) {
        // A batch that cannot fit is refused BEFORE the workspace the engine holds is given up: what the device could offer at best is
        // its free memory plus that workspace.  The engine stays usable (same plan cache state as after any other shape change), and
        // the message names the way out the reference has for this (rrdb_blocks.py:39-47: recompute instead of keeping activations).
        const double gb = 1.0 / (1024.0 * 1024.0 * 1024.0);
        const char* hint = train ? "; the saved activations of a training step are ~7.9 KB per low-resolution pixel -- construct the generator with "
                                   "memory_efficient=True (chunked recompute, XSD_ME_CHUNK tiles at a time: the reference's rrdb_blocks.py:39-47 policy) or use a smaller batch"
                                 : "; use a smaller batch per call";
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > free_b + e->ws_bytes)
            return fail(XSD_ERR_NOMEM, "a workspace of %.1f GiB for %d x %d x %d tiles does not fit this device (%.1f GiB free + %.1f GiB held by this engine of %.1f GiB)%s",
                        need * gb, B, H, W, free_b * gb, e->ws_bytes * gb, total_b * gb, hint);
        if (e->ws) { hipDeviceSynchronize(); hipFree(e->ws); e->ws = nullptr; e->ws_bytes = 0; }
        hipError_t err = hipMalloc((void**)&e->ws, need);
        if (err != hipSuccess) {
            (void)hipGetLastError();      // the failed allocation must not surface again as the next launch's error
            return fail(XSD_ERR_NOMEM, "workspace hipMalloc(%.1f GiB for %d x %d x %d tiles) failed: %s%s", need * gb, B, H, W, hipGetErrorString(err), hint);
        }
        e->ws_bytes = need;
    }
    if (e->amax_used + xsd_engine::AMAX_TAIL > e->amax_cap) {     // the sizing pass counted more max-|x| slots than are allocated
        const int cap = ((e->amax_used + xsd_engine::AMAX_TAIL + 4095) / 4096) * 4096;
        hipDeviceSynchronize();
        float* grown = nullptr;
        hipError_t err = hipMalloc((void**)&grown, sizeof(float) * (size_t)cap);
        if (err == hipSuccess) err = hipMemset(grown, 0, sizeof(float) * (size_t)cap);
        if (err != hipSuccess) { if (grown) hipFree(grown); return fail(XSD_ERR_NOMEM, "max-|x| slot array hipMalloc(%d floats) failed: %s", cap, hipGetErrorString(err)); }
        if (e->amax) {      // slots 0 / 1 hold the packed panels' maxima (written by the last xsd_pack_weights): they move along
            err = hipMemcpy(grown, e->amax, 2 * sizeof(float), hipMemcpyDeviceToDevice);
            if (err != hipSuccess) { hipFree(grown); return fail(XSD_ERR_HIP, "max-|x| slot copy failed: %s", hipGetErrorString(err)); }
            hipFree(e->amax);
        }
        e->amax = grown; e->amax_cap = cap;
    }
    e->amax_used = 2;
    e->amax_bwd_first = 1 << 30;      // a forward-only plan has no backward slots; build() sets it where the backward stages begin
    Builder real(e, B, H, W, train, reinterpret_cast<uintptr_t>(e->ws));
    real.build();
    e->pB = B; e->pH = H; e->pW = W; e->ptrain = (int)train;
    return XSD_OK;
}

// ------------------------------------------------------------------------------------------------------------
// zero-padded widths: expansion of the caller's parameters / gathering of the gradients
// ------------------------------------------------------------------------------------------------------------
struct PadDesc {
    long long w_r, b_r, w_p, b_p;    // weight / bias offsets in the caller's (real) and in the padded flat vector
    int cout_r, cin_r, cin_p;        // OIHW extents (cout_p never matters: output channel oc keeps its index)
    int tin_r, tin_p;                // width of the tensors the input is concatenated from: input channel t*tin_r + k <-> t*tin_p + k
};
__global__ void pad_params_kernel(const float* real, float* padded, const PadDesc* descs, int to_real /* 0: expand params, 1: gather grads */)
{
    const PadDesc d = descs[blockIdx.y];
    const long long nw = (long long)d.cout_r * d.cin_r * 9;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < nw + d.cout_r; e += (long long)gridDim.x * blockDim.x) {
        long long r, q;
        if (e < nw) {
            const int tap = (int)(e % 9), ic = (int)((e / 9) % d.cin_r), oc = (int)(e / (9ll * d.cin_r));
            const int icp = (ic / d.tin_r) * d.tin_p + ic % d.tin_r;
            r = d.w_r + e; q = d.w_p + ((long long)oc * d.cin_p + icp) * 9 + tap;
        } else { r = d.b_r + (e - nw); q = d.b_p + (e - nw); }
        if (to_real) const_cast<float*>(real)[r] = padded[q]; else padded[q] = real[r];
    }
}

// ------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------
extern "C" {

const char* xsd_last_error(void) { return g_err.c_str(); }
#if defined(XSD_DIAG)
const char* xsd_version(void) { return "xsd-hip gfx950 r6 (diagnostic variant)"; }
#elif defined(XSD_TEST_HOOKS)
const char* xsd_version(void) { return "xsd-hip gfx950 r6 (test-hooks variant)"; }
#else
const char* xsd_version(void) { return "xsd-hip gfx950 r6"; }
#endif

int xsd_create(const xsd_config* cfg, xsd_engine** out)
{
    if (!cfg || !out) return fail(XSD_ERR_ARG, "null argument");
    if (cfg->kind != XSD_KIND_DN && cfg->kind != XSD_KIND_SR) return fail(XSD_ERR_ARG, "kind must be 0 (DN) or 1 (SR)");
    if (cfg->in_channels < 1 || cfg->out_channels < 1 || cfg->num_filters < 1 || cfg->in_channels > 1024 || cfg->out_channels > 1024 || cfg->num_filters > 1024)
        return fail(XSD_ERR_ARG, "in_channels, out_channels and num_filters must be in [1,1024] (got %d,%d,%d)", cfg->in_channels, cfg->out_channels, cfg->num_filters);
    // GeneratorRRDB_DN adds its input to the conv_last output (generator_rrdb.py:134): the shapes must broadcast
    if (cfg->kind == XSD_KIND_DN && cfg->in_channels != cfg->out_channels && cfg->in_channels != 1)
        return fail(XSD_ERR_ARG, "DN: `out + x` needs in_channels == out_channels or in_channels == 1 (got %d,%d)", cfg->in_channels, cfg->out_channels);
    if (cfg->num_res_blocks < 1 || cfg->num_res_blocks > 64) return fail(XSD_ERR_ARG, "num_res_blocks must be in [1,64]");
    if (cfg->kind == XSD_KIND_SR && (cfg->num_upsample < 1 || cfg->num_upsample > 2)) return fail(XSD_ERR_ARG, "num_upsample must be 1 or 2");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(XSD_ERR_HIP, "no HIP device available");
    xsd_engine* e = new xsd_engine();
    e->cfg = *cfg;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) e->ncu = prop.multiProcessorCount;
#ifdef XSD_TEST_HOOKS
        // test hooks (include/xsd.h; hooks variant only): plan as if the device had this many CUs; start the max-|x| slot array this small
        if (const char* m = getenv("XSD_TEST_NCU")) { const int v = atoi(m); if (v >= 8 && v <= 4096) e->ncu = v; }
        if (const char* m = getenv("XSD_TEST_AMAX_CAP")) { const int v = atoi(m); if (v >= 2 * xsd_engine::AMAX_TAIL && v <= e->amax_cap) e->amax_cap = v; }
#endif
    }
    // 32-channel planes (widths that are no multiple of 32 zero-padded to the next one), a few image channels
    const bool plane_path = cfg->num_filters <= 256 && cfg->in_channels <= 8 && cfg->out_channels <= 8 &&
                            (cfg->kind == XSD_KIND_SR || cfg->in_channels == cfg->out_channels || cfg->in_channels == 1);
    if (!plane_path) {
        // more than 256 filters or more than 8 image channels: the generic-width path (exact fp32)
        e->generic = GenericNet::create(*cfg);
        if (!e->generic) { delete e; return fail(XSD_ERR_NOMEM, "generic-width engine: device allocation failed"); }
        e->nparams = e->generic->nparams;
        if (hipMalloc((void**)&e->loss_partial, sizeof(double) * 1024) != hipSuccess) { xsd_destroy(e); return fail(XSD_ERR_NOMEM, "device allocation failed"); }
        *out = e;
        return XSD_OK;
    }
#ifdef XSD_DIAG   // the diagnostic library variant (make diag, selected with XSD_LIB) is the only build that reads these
    if (const char* m = getenv("XSD_ABLATE")) e->ablate = atoi(m);
    if (const char* m = getenv("XSD_CHUNK")) e->chunk = atoi(m);
#endif
    if (const char* m = getenv("XSD_MATH")) {
        if (strcmp(m, "f16x3") == 0 || strcmp(m, "4") == 0) e->math = 4;
        else if (strcmp(m, "bf16x6") == 0 || strcmp(m, "3") == 0) e->math = 3;
        else if (strcmp(m, "fp32") == 0 || strcmp(m, "0") == 0) e->math = 0;
        else { delete e; return fail(XSD_ERR_ARG, "XSD_MATH=%s: the math modes are fp32, bf16x6 and f16x3", m); }
    }
    const int blocks = cfg->num_res_blocks, nup = cfg->kind == XSD_KIND_SR ? cfg->num_upsample : 0;
    const int nf_r = cfg->num_filters, nf = (nf_r + 31) / 32 * 32;      // real / padded width
    const int CIc = cfg->in_channels, COc = cfg->out_channels;
    e->planes = nf / 32;
    if (nf != nf_r) e->nf_pad = nf;
    long long off = 0, pk = 0, sb = 0, off_r = 0;
    std::vector<PadDesc> pads;      // every conv in parameter order, with its place in the caller's (real-width) flat vector
    auto pad_add = [&](long long w_p, long long b_p, int cout_r, int cin_r, int cin_p, int tin_r, int tin_p) {
        PadDesc d; d.w_p = w_p; d.b_p = b_p; d.w_r = off_r; off_r += (long long)cout_r * cin_r * 9; d.b_r = off_r; off_r += cout_r;
        d.cout_r = cout_r; d.cin_r = cin_r; d.cin_p = cin_p; d.tin_r = tin_r; d.tin_p = tin_p;
        pads.push_back(d);
    };
    take_conv(off, nf, CIc, e->first_w, e->first_b);
    pad_add(e->first_w, e->first_b, nf_r, CIc, CIc, CIc, CIc);
    auto mk = [&](int cout, int cin, int shuffle) {
        ConvW c; take_conv(off, cout, cin, c.w_off, c.b_off);
        c.cout = cout; c.cin = cin; c.shuffle = shuffle;
        c.fwd_off = pk; c.bwd_off = pk; pk += (long long)(cout / 32) * (cin / 32) * PANEL_FLOATS;
        c.sbias_off = -1;
        if (shuffle) { c.sbias_off = sb; sb += cout; }
        pad_add(c.w_off, c.b_off, cout / nf * nf_r, cin / nf * nf_r, cin, nf_r, nf);      // cout, cin are multiples of the (padded) width
        return c;
    };
    for (int i = 0; i < blocks; ++i) {
        e->rrdb_begin.push_back(off);
        e->rrdb_begin_real.push_back(off_r);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 5; ++c) {
                e->rdb.push_back(mk(nf, nf * (c + 1), 0));
                // conv5's gradient arrives as 0.2*dOut (x5*0.2 + x) and, for RDB3, 0.2*0.2*dOut (out*0.2 + x): fold it
                if (c == 4) e->rdb.back().bwd_scale = r == 2 ? 0.04f : 0.2f;
            }
    }
    e->rrdb_begin.push_back(off);
    e->rrdb_begin_real.push_back(off_r);
    e->trunk = mk(nf, nf, 0);
    take_conv(off, COc, nf, e->last_w, e->last_b);
    pad_add(e->last_w, e->last_b, COc, nf_r, nf, nf_r, nf);
    for (int u = 0; u < nup; ++u) e->up.push_back(mk(4 * nf, nf, 1));
    if (cfg->kind == XSD_KIND_SR) e->hr = mk(nf, nf, 0);
    e->nparams_pad = off;
    e->nparams_real = off_r;
    e->npad = (int)pads.size();
    e->nparams = e->nf_pad ? off_r : off;       // what the caller's flat vectors hold
    e->pk_floats = pk;

    std::vector<PackDesc> descs;
    auto add = [&](const ConvW& c) { PackDesc d; d.src_w = c.w_off; d.dst_fwd = c.fwd_off; d.dst_bwd = c.bwd_off; d.cout = c.cout; d.cin = c.cin; d.shuffle = c.shuffle; d.bwd_scale = c.bwd_scale; descs.push_back(d); };
    for (auto& c : e->rdb) add(c);
    add(e->trunk);
    for (auto& c : e->up) add(c);
    if (cfg->kind == XSD_KIND_SR) add(e->hr);
    e->ndesc = (int)descs.size();
#define CK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { int rc = fail(XSD_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); xsd_destroy(e); return rc; } } while (0)
    CK(hipMalloc((void**)&e->pk_fwd, sizeof(float) * pk));
    CK(hipMalloc((void**)&e->pk_bwd, sizeof(float) * pk));
    CK(hipMalloc((void**)&e->pk_fwd_s, sizeof(float) * pk));
    CK(hipMalloc((void**)&e->pk_bwd_s, sizeof(float) * pk));
    if (e->nf_pad) {
        CK(hipMalloc((void**)&e->params_pad, sizeof(float) * e->nparams_pad));
        CK(hipMalloc((void**)&e->grads_pad, sizeof(float) * e->nparams_pad));
        CK(hipMalloc((void**)&e->pad_descs, sizeof(PadDesc) * pads.size()));
        CK(hipMemcpy(e->pad_descs, pads.data(), sizeof(PadDesc) * pads.size(), hipMemcpyHostToDevice));
    }
    CK(hipMalloc((void**)&e->amax, sizeof(float) * e->amax_cap));
    CK(hipMemset(e->amax, 0, sizeof(float) * e->amax_cap));
    CK(hipMalloc((void**)&e->zero_page, 512));   // [0,256): zeros (padding source); [256,512): trash (stores of lanes outside the image)
    CK(hipMemset(e->zero_page, 0, 512));
    CK(hipMalloc((void**)&e->pk_edge, sizeof(float) * 2 * 288 * e->planes * (cfg->in_channels + cfg->out_channels)));   // [first_fwd | first_bwd][in channel][plane][288], [last_fwd | last_bwd][out channel][plane][288]
    CK(hipMalloc((void**)&e->pk_sbias, sizeof(float) * (sb ? sb : 1)));
    CK(hipMalloc((void**)&e->descs_dev, sizeof(PackDesc) * descs.size()));
    CK(hipMemcpy(e->descs_dev, descs.data(), sizeof(PackDesc) * descs.size(), hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&e->wg_partial, sizeof(float) * (size_t)e->nparts * 5 * PANEL_FLOATS));
    CK(hipMalloc((void**)&e->wg_bias_partial, sizeof(float) * (size_t)e->nparts * 4 * 32));
    CK(hipMalloc((void**)&e->edge_partial, sizeof(float) * EDGE_WGRAD_BLOCKS * 321));
    CK(hipMalloc((void**)&e->loss_partial, sizeof(double) * 1024));
#undef CK
    *out = e;
    return XSD_OK;
}

void xsd_destroy(xsd_engine* e)
{
    if (!e) return;
    hipDeviceSynchronize();
    delete e->generic;
    for (auto ev : e->ev_pool) hipEventDestroy(ev);
    hipFree(e->pk_fwd); hipFree(e->pk_bwd); hipFree(e->pk_fwd_s); hipFree(e->pk_bwd_s); hipFree(e->zero_page); hipFree(e->amax); hipFree(e->pk_edge); hipFree(e->pk_sbias); hipFree(e->descs_dev);
    hipFree(e->wg_partial); hipFree(e->wg_bias_partial); hipFree(e->edge_partial); hipFree(e->loss_partial);
    hipFree(e->ws); hipFree(e->params_pad); hipFree(e->grads_pad); hipFree(e->pad_descs);
    delete e;
}

int64_t xsd_param_count(const xsd_engine* e) { return e ? e->nparams : 0; }

int xsd_set_math(xsd_engine* e, int mode)
{
    if (!e || (mode != 0 && mode != 3 && mode != 4)) return fail(XSD_ERR_ARG, "math mode must be 0 (fp32, exact), 3 (bf16x6) or 4 (f16x3); modes 1 and 2 (16-bit significands) were removed in round 3");
    if (e->generic) { e->math = mode; return XSD_OK; }   // the generic-width kernels are exact fp32 whatever the mode says
    if (mode != e->math) { e->math = mode; e->packed = false; e->pB = 0; e->ptrain = -1; e->fwd_saved = false; }
    return XSD_OK;
}
// what the kernels actually compute in: the exact-fp32 kernels of generic_net.hip serve every mode setting
int xsd_get_math(const xsd_engine* e) { return e ? (e->generic ? 0 : e->math) : -1; }

int xsd_pack_weights(xsd_engine* e, const float* dev_params, void* stream)
{
    if (!e || !dev_params) return fail(XSD_ERR_ARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    e->params = dev_params;
    if (e->generic) { HIPCHK(e->generic->pack(dev_params, s)); e->packed = true; return XSD_OK; }
    if (e->nf_pad) {     // the engine runs on the zero-padded copy
        HIPCHK(hipMemsetAsync(e->params_pad, 0, sizeof(float) * e->nparams_pad, s));
        hipLaunchKernelGGL(pad_params_kernel, dim3(64, e->npad), dim3(256), 0, s, dev_params, e->params_pad, static_cast<const PadDesc*>(e->pad_descs), 0);
        HIPCHK(hipGetLastError());
        dev_params = e->params_pad;
        e->params = dev_params;
    }
    if (e->math == 4) {
        // fp32 fragment-order panels into the (otherwise unused) mode-0 buffers, max |w| of the forward and of the
        // input-gradient panels (one power-of-two scale each), then the two-term fp16 images the conv kernel copies to LDS
        HIPCHK(launch_pack_weights_s3(dev_params, e->descs_dev, e->ndesc, e->pk_fwd, e->pk_bwd, s));
        HIPCHK(hipMemsetAsync(e->amax, 0, 2 * sizeof(float), s));
        HIPCHK(launch_buffer_amax(e->pk_fwd, e->pk_floats, e->amax + 0, s));
        HIPCHK(launch_buffer_amax(e->pk_bwd, e->pk_floats, e->amax + 1, s));
        HIPCHK(launch_split_panels_f16(e->pk_fwd, e->pk_fwd_s, e->pk_floats, e->amax + 0, s));
        HIPCHK(launch_split_panels_f16(e->pk_bwd, e->pk_bwd_s, e->pk_floats, e->amax + 1, s));
    } else if (e->math == 3) {
        HIPCHK(launch_pack_weights_s3(dev_params, e->descs_dev, e->ndesc, reinterpret_cast<float*>(e->pk_fwd_s), reinterpret_cast<float*>(e->pk_bwd_s), s));
    }
    else
        HIPCHK(launch_pack_weights(dev_params, e->descs_dev, e->ndesc, e->pk_fwd, e->pk_bwd, s));
    {   // conv_first W[c][ch][tap] and conv_last W[co][c][tap], 32 feature channels (one plane) of one image channel at a time
        const int P = e->planes, ci = e->cfg.in_channels, co = e->cfg.out_channels, nf = 32 * P;
        float* ff = e->pk_edge; float* fb = ff + 288 * P * ci; float* lf = fb + 288 * P * ci; float* lb = lf + 288 * P * co;
        for (int ch = 0; ch < ci; ++ch)
            for (int q = 0; q < P; ++q)
                HIPCHK(launch_pack_edge(dev_params + e->first_w + ((long long)32 * q * ci + ch) * 9, nullptr, ff + 288 * (ch * P + q), fb + 288 * (ch * P + q),
                                        nullptr, nullptr, s, 9 * ci));
        for (int o = 0; o < co; ++o)
            for (int q = 0; q < P; ++q)
                HIPCHK(launch_pack_edge(nullptr, dev_params + e->last_w + ((long long)o * nf + 32 * q) * 9, nullptr, nullptr, lf + 288 * (o * P + q),
                                        lb + 288 * (o * P + q), s));
    }
    for (auto& c : e->up) HIPCHK(launch_pack_shuffle_bias(dev_params + c.b_off, e->pk_sbias + c.sbias_off, e->planes, s));
    e->packed = true;
    return XSD_OK;
}

int xsd_forward(xsd_engine* e, const float* dev_x, float* dev_y, int B, int H, int W, int save_for_backward, void* stream)
{
    if (!e || !dev_x || !dev_y) return fail(XSD_ERR_ARG, "null argument");
    if (B < 1 || H < 1 || W < 1) return fail(XSD_ERR_ARG, "bad shape %dx%dx%d", B, H, W);
    if (!e->packed) return fail(XSD_ERR_STATE, "xsd_pack_weights must be called before xsd_forward");
    const int lo = e->cfg.kind == XSD_KIND_SR ? e->cfg.num_upsample : 0;
    if (e->generic) {
        const long long big = (long long)B * (H << lo) * (W << lo) * std::max(5 * e->cfg.num_filters, 4 * e->cfg.num_filters);
        if (big >= (1ll << 40)) return fail(XSD_ERR_ARG, "batch x image too large");
        hipError_t err = e->generic->forward(dev_x, dev_y, B, H, W, save_for_backward != 0, (hipStream_t)stream);
        if (err != hipSuccess) return fail(err == hipErrorOutOfMemory ? XSD_ERR_NOMEM : XSD_ERR_HIP, "generic-width forward: %s", hipGetErrorString(err));
        e->fwd_saved = save_for_backward != 0;
        return XSD_OK;
    }
    if ((long long)(H << lo) * (W << lo) * 32 >= (1ll << 31)) return fail(XSD_ERR_ARG, "image too large for 32-bit in-image offsets");
    // the split-mode kernels address a plane's batch slice with 32-bit BYTE offsets (buffer loads / stores): 128 B per pixel
    if (e->math >= 3 && (long long)(H << lo) * (W << lo) * 128 >= (1ll << 31))
        return fail(XSD_ERR_ARG, "image of %d x %d output pixels is too large for math modes bf16x6 / f16x3 (32-bit byte offsets: fewer than 2^24 pixels per image); use math mode fp32", H << lo, W << lo);
    if ((W << lo) > EDGE_MAX_W) return fail(XSD_ERR_ARG, "output rows wider than %d pixels are not supported", EDGE_MAX_W);
    int rc = ensure_plan(e, B, H, W, save_for_backward != 0);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    e->b_x = dev_x; e->b_y = dev_y;
    if (e->math == 4 && e->amax_used > 2) HIPCHK(hipMemsetAsync(e->amax + 2, 0, sizeof(float) * (e->amax_used - 2), s));   // every plane slot of the plan (forward and backward)
    for (auto& op : e->fwd_ops) HIPCHK(op(s));
    e->fwd_saved = save_for_backward != 0;
    return XSD_OK;
}

int xsd_backward_num_stages(const xsd_engine* e) { return e ? e->cfg.num_res_blocks + 2 : 0; }

int xsd_backward_stage(xsd_engine* e, int stage, const float* dev_dy, float* dev_dx_or_null, float* dev_grads, void* stream)
{
    if (!e || !dev_dy || !dev_grads) return fail(XSD_ERR_ARG, "null argument");
    if (!e->fwd_saved) return fail(XSD_ERR_STATE, "xsd_backward needs a preceding xsd_forward(save_for_backward=1)");
    if (e->generic) {
        if (stage < 0 || stage >= e->generic->num_stages()) return fail(XSD_ERR_ARG, "stage %d out of range", stage);
        hipError_t err = e->generic->backward_stage(stage, dev_dy, dev_dx_or_null, dev_grads, (hipStream_t)stream);
        if (err != hipSuccess) return fail(XSD_ERR_HIP, "generic-width backward stage %d: %s", stage, hipGetErrorString(err));
        return XSD_OK;
    }
    if (stage < 0 || stage >= (int)e->bwd_stages.size()) return fail(XSD_ERR_ARG, "stage %d out of range", stage);
    hipStream_t s = (hipStream_t)stream;
    e->b_dy = dev_dy; e->b_dx = dev_dx_or_null; e->b_grads = e->nf_pad ? e->grads_pad : dev_grads;
    // f16x3: the slots the backward launches report their planes' max |x| into (atomic max) start every backward at zero -- a
    // second backward after the same forward (another dy) must not inherit the first one's maxima (they would only over-
    // estimate, safe against fp16 overflow but costing operand bits)
    if (stage == 0 && e->math == 4 && e->amax_used > e->amax_bwd_first)
        HIPCHK(hipMemsetAsync(e->amax + e->amax_bwd_first, 0, sizeof(float) * (e->amax_used - e->amax_bwd_first), s));
    for (auto& op : e->bwd_stages[stage]) HIPCHK(op(s));
    if (e->nf_pad) {     // this stage's convs from the padded gradient into the caller's vector
        const int blocks = e->cfg.num_res_blocks;
        int d0, d1;
        if (stage == blocks + 1) { d0 = 0; d1 = 1; }                                   // conv_first
        else if (stage >= 1) { const int i = blocks - stage; d0 = 1 + 15 * i; d1 = d0 + 15; }   // dense blocks of rrdb.i
        else { d0 = 1 + 15 * blocks; d1 = e->npad; }                                  // trunk, conv_last, upsampling, HRconv
        hipLaunchKernelGGL(pad_params_kernel, dim3(64, d1 - d0), dim3(256), 0, s, dev_grads, e->grads_pad, static_cast<const PadDesc*>(e->pad_descs) + d0, 1);
        HIPCHK(hipGetLastError());
    }
    return XSD_OK;
}

int xsd_backward(xsd_engine* e, const float* dev_dy, float* dev_dx_or_null, float* dev_grads, void* stream)
{
    const int n = xsd_backward_num_stages(e);
    for (int st = 0; st < n; ++st) {
        int rc = xsd_backward_stage(e, st, dev_dy, dev_dx_or_null, dev_grads, stream);
        if (rc) return rc;
    }
    return XSD_OK;
}

int xsd_grad_range(const xsd_engine* e, int stage, int range_idx, int64_t* offset, int64_t* count)
{
    if (!e) return 0;
    const int blocks = e->cfg.num_res_blocks;
    if (range_idx != 0 || stage < 0 || stage > blocks + 1) return 1;
    long long a, b;
    if (e->generic) {
        e->generic->grad_range(stage, &a, &b);
        if (offset) *offset = a;
        if (count) *count = b;
        return 1;
    }
    const std::vector<long long>& rb = e->nf_pad ? e->rrdb_begin_real : e->rrdb_begin;   // offsets in the caller's vector
    if (stage == 0) { a = rb[blocks]; b = e->nparams; }
    else if (stage <= blocks) { const int i = blocks - stage; a = rb[i]; b = rb[i + 1]; }
    else { a = 0; b = rb[0]; }
    if (offset) *offset = a;
    if (count) *count = b - a;
    return 1;
}

int xsd_l1_loss(xsd_engine* e, const float* dev_y, const float* dev_target, float* dev_dy_or_null, float* dev_loss, int64_t n, void* stream)
{
    if (!e || !dev_y || !dev_target || !dev_loss || n <= 0) return fail(XSD_ERR_ARG, "bad argument");
    HIPCHK(prof_launch(e, PK_LOSS, 0.0, (dev_dy_or_null ? 12.0 : 8.0) * n, (hipStream_t)stream, [&]() { return launch_l1_loss(dev_y, dev_target, dev_dy_or_null, e->loss_partial, 1024, dev_loss, n, (hipStream_t)stream); }));
    return XSD_OK;
}

struct xsd_loss_fn {
    LossWeights w;
    int channels = 1;        // images per sample (xsd_loss_set_channels)
    void* ws = nullptr;      // workspace, grown on demand
    size_t ws_bytes = 0;
};

int xsd_loss_create(const xsd_loss_config* cfg, xsd_loss_fn** out)
{
    if (!cfg || !out) return fail(XSD_ERR_ARG, "bad argument");
    LossWeights w;
    w.w[0] = cfg->w_l1; w.w[1] = cfg->w_poisson; w.w[2] = cfg->w_psnr; w.w[3] = cfg->w_ssim; w.w[4] = cfg->w_ms_ssim;
    w.correction = cfg->correction; w.sigma = cfg->sigma; w.k1 = cfg->k1; w.k2 = cfg->k2; w.kernel_size = cfg->kernel_size;
    bool any = false;
    for (int i = 0; i < 5; ++i) any = any || w.w[i] != 0.f;
    if (!any) return fail(XSD_ERR_ARG, "loss: no term has a non-zero weight");   // `assert metrics` (loss_functions.py:38)
    if ((w.w[3] != 0.f || w.w[4] != 0.f) && !(w.sigma > 0.f)) return fail(XSD_ERR_ARG, "loss: sigma must be > 0");
    xsd_loss_fn* f = new (std::nothrow) xsd_loss_fn();
    if (!f) return fail(XSD_ERR_NOMEM, "out of host memory");
    f->w = w;
    *out = f;
    return XSD_OK;
}

void xsd_loss_destroy(xsd_loss_fn* f)
{
    if (!f) return;
    hipFree(f->ws);
    delete f;
}

int xsd_loss_set_channels(xsd_loss_fn* f, int channels)
{
    if (!f || channels < 1 || channels > 1024) return fail(XSD_ERR_ARG, "xsd_loss_set_channels: channels must be 1..1024");
    f->channels = channels;
    return XSD_OK;
}

int xsd_loss_eval(xsd_loss_fn* f, const float* dev_y, const float* dev_target, float* dev_dy_or_null, float* dev_out8, int B, int H,
                  int W, void* stream)
{
    if (!f || !dev_y || !dev_target || !dev_out8) return fail(XSD_ERR_ARG, "bad argument");
    if (B < 1 || B % f->channels) return fail(XSD_ERR_ARG, "xsd_loss_eval: %d images are no whole number of %d-channel samples (xsd_loss_set_channels)", B, f->channels);
    const char* why = nullptr;
    if (loss_check(f->w, B, H, W, &why)) return fail(XSD_ERR_ARG, why);
    const size_t need = loss_workspace_bytes(B, H, W);
    if (need > f->ws_bytes) {
        HIPCHK(hipStreamSynchronize((hipStream_t)stream));
        if (f->ws) HIPCHK(hipFree(f->ws));
        f->ws = nullptr; f->ws_bytes = 0;
        if (hipMalloc(&f->ws, need) != hipSuccess) return fail(XSD_ERR_NOMEM, "loss workspace allocation failed");
        f->ws_bytes = need;
    }
    HIPCHK(launch_loss(f->w, dev_y, dev_target, dev_dy_or_null, dev_out8, B, f->channels, H, W, f->ws, (hipStream_t)stream));
    return XSD_OK;
}

int xsd_adam_step(xsd_engine* e, float* dev_params, const float* dev_grads, float* dev_m, float* dev_v, int64_t n, int step,
                  float lr, float beta1, float beta2, float eps, float grad_scale, void* stream)
{
    if (!dev_params || !dev_grads || !dev_m || !dev_v || n <= 0 || step < 1) return fail(XSD_ERR_ARG, "bad argument");
    auto go = [&]() { return launch_adam(dev_params, dev_grads, dev_m, dev_v, n, step, lr, beta1, beta2, eps, grad_scale, (hipStream_t)stream); };
    if (e) HIPCHK(prof_launch(e, PK_ADAM, 0.0, 28.0 * n, (hipStream_t)stream, go));      // reads p, g, m, v; writes p, m, v
    else HIPCHK(go());
    return XSD_OK;
}

int xsd_mask_pad_normalize(const void* dev_counts, int counts_is_int32, const uint8_t* dev_mask_or_null, float* dev_out, int B,
                           int Hin, int Win, int res, int do_normalize, float max_val, int stretch, void* stream)
{
    if (!dev_counts || !dev_out || B < 1 || Hin < 1 || Win < 1 || res < 1) return fail(XSD_ERR_ARG, "bad argument");
    if (stretch < 0 || stretch > 3) return fail(XSD_ERR_ARG, "stretch must be 0..3");
    if (do_normalize && !(max_val > 0.f)) return fail(XSD_ERR_ARG, "max_val must be > 0 on the fused path");
    MaskPadParams p; memset(&p, 0, sizeof(p));
    if (counts_is_int32) p.counts_i32 = (const int32_t*)dev_counts; else p.counts_f32 = (const float*)dev_counts;
    p.mask = dev_mask_or_null; p.out = dev_out; p.B = B; p.Hin = Hin; p.Win = Win; p.res = res;
    p.y_top = (int)std::floor((res - Hin) / 2.0); p.x_left = (int)std::floor((res - Win) / 2.0);
    p.do_norm = do_normalize; p.mode = stretch; p.max_val = max_val; p.upsample = 1;
    HIPCHK(launch_mask_pad_normalize(p, (hipStream_t)stream));
    return XSD_OK;
}

int xsd_compose_input(const void* dev_img, const void* dev_agn_or_null, const void* dev_bkg_or_null, int is_int32, int big_endian,
                      const uint8_t* dev_mask_or_null, float* dev_out, int B, int Hin, int Win, int upsample, int res,
                      int do_normalize, float max_val, int stretch, void* stream)
{
    if (!dev_img || !dev_out || B < 1 || Hin < 1 || Win < 1 || res < 1 || upsample < 1) return fail(XSD_ERR_ARG, "bad argument");
    if (stretch < 0 || stretch > 3) return fail(XSD_ERR_ARG, "stretch must be 0..3");
    if (do_normalize && !(max_val > 0.f)) return fail(XSD_ERR_ARG, "max_val must be > 0 on the fused path");
    MaskPadParams p; memset(&p, 0, sizeof(p));
    if (is_int32) p.counts_i32 = (const int32_t*)dev_img; else p.counts_f32 = (const float*)dev_img;
    p.extra1 = dev_agn_or_null; p.extra2 = dev_bkg_or_null; p.big_endian = big_endian; p.upsample = upsample;
    p.mask = dev_mask_or_null; p.out = dev_out; p.B = B; p.Hin = Hin; p.Win = Win; p.res = res;
    p.y_top = (int)std::floor((res - Hin * upsample) / 2.0); p.x_left = (int)std::floor((res - Win * upsample) / 2.0);
    p.do_norm = do_normalize; p.mode = stretch; p.max_val = max_val;
    HIPCHK(launch_mask_pad_normalize(p, (hipStream_t)stream));
    return XSD_OK;
}

int xsd_normalize(const float* dev_in, float* dev_out, int64_t n, float max_val, int stretch, int inverse, void* stream)
{
    if (!dev_in || !dev_out || n <= 0 || !(max_val > 0.f) || stretch < 0 || stretch > 3) return fail(XSD_ERR_ARG, "bad argument");
    HIPCHK(launch_normalize(dev_in, dev_out, n, max_val, stretch, inverse, (hipStream_t)stream));
    return XSD_OK;
}

int xsd_image_upsample(const float* dev_in, float* dev_out, int N, int H, int W, int scale, void* stream)
{
    if (!dev_in || !dev_out || N < 1 || H < 1 || W < 1 || scale < 1) return fail(XSD_ERR_ARG, "bad argument");
    HIPCHK(launch_upsample_nearest(dev_in, dev_out, N, H, W, scale, (hipStream_t)stream));
    return XSD_OK;
}

int xsd_debug_persistent_grid(int ntiles, int ncu) { return (ntiles < 0 || ncu < 1) ? -1 : xsd::persistent_grid(ntiles, ncu); }
int xsd_debug_occupancy(int lds_bytes) { return xsd::debug_conv_occupancy(lds_bytes); }
float xsd_debug_residency_ms(int grid, int threads, int lds_bytes, int us) { return xsd::debug_residency_ms(grid, threads, lds_bytes, us); }

int xsd_probe_mfma_stream(int fmt, double seconds, double* mfma_tflops, double* sclk_ghz, void* stream)
{
    if ((fmt != 0 && fmt != 1) || !(seconds > 0.0) || seconds > 30.0 || !mfma_tflops)
        return fail(XSD_ERR_ARG, "xsd_probe_mfma_stream: fmt must be 0 (f16) or 1 (bf16), 0 < seconds <= 30, mfma_tflops non-NULL");
    HIPCHK(xsd::probe_mfma_stream(fmt, seconds, mfma_tflops, sclk_ghz, (hipStream_t)stream));
    return XSD_OK;
}

// diagnostic: accumulate shader-cycle stamps of the conv kernel's phases (enable != 0 allocates/zeroes; read copies out)
int xsd_debug_stamps(xsd_engine* e, int enable, unsigned long long* out16)
{
    if (!e) return fail(XSD_ERR_ARG, "null engine");
    HIPCHK(hipDeviceSynchronize());
    if (out16 && e->dbg) HIPCHK(hipMemcpy(out16, e->dbg, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (enable) {
        if (!e->dbg) HIPCHK(hipMalloc((void**)&e->dbg, 32 * sizeof(unsigned long long)));
        HIPCHK(hipMemset(e->dbg, 0, 32 * sizeof(unsigned long long)));
    } else if (e->dbg) { hipFree(e->dbg); e->dbg = nullptr; }
    return XSD_OK;
}

int xsd_profile_enable(xsd_engine* e, int enable)
{
    if (!e) return fail(XSD_ERR_ARG, "null engine");
    hipDeviceSynchronize();
    e->recs.clear(); e->ev_used = 0; e->prof = enable != 0;
    return XSD_OK;
}

int xsd_profile_read(xsd_engine* e, int klass, double* total_ms, int64_t* launches, double* total_flop, double* total_bytes)
{
    if (!e) return fail(XSD_ERR_ARG, "null engine");
    HIPCHK(hipDeviceSynchronize());
    double ms = 0, fl = 0, by = 0; int64_t n = 0;
    for (auto& r : e->recs) {
        if (r.klass != klass) continue;
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, r.a, r.b));
        ms += t; fl += r.flop; by += r.bytes; ++n;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    if (total_flop) *total_flop = fl;
    if (total_bytes) *total_bytes = by;
    return XSD_OK;
}

// ---- single-layer test hooks ---------------------------------------------------------------------------------
static int pack_single(const float* dev_w, int cout, int cin, float** fwd, float** bwd, int math, hipStream_t s, float* amax_slots = nullptr)
{
    const long long n = (long long)(cout / 32) * (cin / 32) * PANEL_FLOATS;
    PackDesc d; d.src_w = 0; d.dst_fwd = 0; d.dst_bwd = 0; d.cout = cout; d.cin = cin; d.shuffle = 0; d.bwd_scale = 1.f;
    PackDesc* dd = nullptr;
    HIPCHK(hipMalloc((void**)fwd, sizeof(float) * n));
    HIPCHK(hipMalloc((void**)bwd, sizeof(float) * n));
    HIPCHK(hipMalloc((void**)&dd, sizeof(PackDesc)));
    HIPCHK(hipMemcpy(dd, &d, sizeof(d), hipMemcpyHostToDevice));
    if (math == 4) {   // as xsd_pack_weights: fp32 panels -> max |w| (slots CAP-4 / CAP-3 of `amax`) -> two-term fp16 images
        float *tf = nullptr, *tb = nullptr;
        HIPCHK(hipMalloc((void**)&tf, sizeof(float) * n));
        HIPCHK(hipMalloc((void**)&tb, sizeof(float) * n));
        HIPCHK(launch_pack_weights_s3(dev_w, dd, 1, tf, tb, s));
        HIPCHK(hipMemsetAsync(amax_slots, 0, 2 * sizeof(float), s));
        HIPCHK(launch_buffer_amax(tf, n, amax_slots + 0, s));
        HIPCHK(launch_buffer_amax(tb, n, amax_slots + 1, s));
        HIPCHK(launch_split_panels_f16(tf, *fwd, n, amax_slots + 0, s));
        HIPCHK(launch_split_panels_f16(tb, *bwd, n, amax_slots + 1, s));
        HIPCHK(hipStreamSynchronize(s));
        hipFree(tf); hipFree(tb);
    } else if (math == 3) HIPCHK(launch_pack_weights_s3(dev_w, dd, 1, *fwd, *bwd, s));
    else HIPCHK(launch_pack_weights(dev_w, dd, 1, *fwd, *bwd, s));
    HIPCHK(hipStreamSynchronize(s));
    hipFree(dd);
    return XSD_OK;
}

static hipError_t run_conv(xsd_engine* e, ConvParams& p, hipStream_t s)
{
    p.zero = e->zero_page;
    p.ablate = e->ablate;      // (0 outside the diagnostic library)
    for (int i = 0; i < (p.n_out > 1 ? p.n_out : p.n_in); ++i) p.wstep[i] = p.wpanel + (long long)i * PANEL_FLOATS;
    if (e->math == 4) {   // test hook: nobody has reported the operands' max |x| -> reduce them here (slots at the end of the array)
        float* t = e->amax + e->amax_cap - 32;    // (CAP-16.. belong to the weight-gradient hook, CAP-4 / CAP-3 to pack_single)
        hipError_t err = hipMemsetAsync(t, 0, 8 * sizeof(float), s);
        if (err != hipSuccess) return err;
        for (int i = 0; i < p.n_in; ++i) {
            err = launch_plane_amax(p.in[i], p.B, p.H, p.W, t + i, s);
            if (err != hipSuccess) return err;
            p.amax_in[i] = t + i;
        }
        if (!p.amax_w) return hipErrorInvalidValue;   // set by the hook from pack_single's slots (forward or input-gradient panels)
        return launch_conv3x3_h2x(p, s);
    }
    if (e->math == 3) return launch_conv3x3_s3x(p, s);
    return launch_conv3x3_mfma(p, s);
}

int xsd_test_conv3x3(xsd_engine* e, const float* const* in_planes, int n_in, const float* dev_w_oihw, const float* dev_bias,
                     float* const* out_planes, int n_out, float slope, int B, int H, int W, void* stream)
{
    if (!e || n_in < 1 || n_in > 5 || n_out < 1 || n_out > 5 || (n_in > 1 && n_out > 1)) return fail(XSD_ERR_ARG, "bad n_in/n_out");
    if (e->generic) return fail(XSD_ERR_ARG, "single-layer test hooks exist for the 32-filter MFMA path only");
    hipStream_t s = (hipStream_t)stream;
    float *fwd = nullptr, *bwd = nullptr;
    int rc = pack_single(dev_w_oihw, 32 * n_out, 32 * n_in, &fwd, &bwd, e->math, s, e->amax + e->amax_cap - 4);
    if (rc) return rc;
    Builder b(e, B, H, W, false, 0);
    ConvParams p = b.conv_base(0);
    p.n_in = n_in; p.n_out = n_out; p.wpanel = fwd; p.bias = dev_bias;
    p.amax_w = e->amax + e->amax_cap - 4;     // mode 4: max |w| of the forward panels (pack_single)
    for (int i = 0; i < n_in; ++i) p.in[i] = b.std_in(in_planes[i], 0);
    for (int j = 0; j < n_out; ++j) { b.std_out(p.out[j], out_planes[j], 0); p.out[j].slope = slope; }
    hipError_t err = run_conv(e, p, s);
    hipStreamSynchronize(s);
    hipFree(fwd); hipFree(bwd);
    if (err != hipSuccess) return fail(XSD_ERR_HIP, "conv launch: %s", hipGetErrorString(err));
    return XSD_OK;
}

int xsd_test_conv3x3_bwd(xsd_engine* e, const float* const* in_planes, int n_in, const float* dev_w_oihw, const float* dev_g_plane,
                         float* const* dx_planes, float* dev_dw_oihw, float* dev_db, int B, int H, int W, void* stream)
{
    if (!e || n_in < 1 || n_in > 5) return fail(XSD_ERR_ARG, "bad n_in");
    if (e->generic) return fail(XSD_ERR_ARG, "single-layer test hooks exist for the 32-filter MFMA path only");
    hipStream_t s = (hipStream_t)stream;
    float *fwd = nullptr, *bwd = nullptr;
    int rc = pack_single(dev_w_oihw, 32, 32 * n_in, &fwd, &bwd, e->math, s, e->amax + e->amax_cap - 4);
    if (rc) return rc;
    const float* g = dev_g_plane;
    Builder b(e, B, H, W, false, 0);
    ConvParams p = b.conv_base(0);
    p.n_in = 1; p.n_out = n_in; p.wpanel = bwd; p.in[0] = b.std_in(g, 0);
    p.amax_w = e->amax + e->amax_cap - 3;     // mode 4: max |w| of the input-gradient panels
    for (int j = 0; j < n_in; ++j) b.std_out(p.out[j], dx_planes[j], 0);
    hipError_t err = run_conv(e, p, s);
    if (err == hipSuccess) {
        WgradParams wp; memset(&wp, 0, sizeof(wp));
        wp.B = B; wp.H = H; wp.W = W; wp.tilesX = p.tilesX; wp.tilesY = p.tilesY; wp.n_in = n_in; wp.n_g = 1; wp.nparts = e->nparts;
        for (int i = 0; i < n_in; ++i) wp.x[i] = b.std_in(in_planes[i], 0);
        wp.g[0] = b.std_in(g, 0);
        wp.partial = e->wg_partial; wp.bias_partial = e->wg_bias_partial; wp.zero = e->zero_page; wp.ablate = e->ablate;
        if (e->math == 4) {   // test hook: reduce the operands' max |x| here (slots at the end of the array, after run_conv's)
            float* t = e->amax + e->amax_cap - 16;
            err = hipMemsetAsync(t, 0, 8 * sizeof(float), s);
            for (int i = 0; i < n_in && err == hipSuccess; ++i) { err = launch_plane_amax(wp.x[i], B, H, W, t + i, s); wp.amax_x[i] = t + i; }
            if (err == hipSuccess) { err = launch_plane_amax(wp.g[0], B, H, W, t + 5, s); wp.amax_g[0] = t + 5; }
        }
        if (err == hipSuccess)
            err = e->math >= 3 ? launch_wgrad_split(e->math, e->ablate, wp, s) : launch_wgrad_mfma(wp, s);
        if (err == hipSuccess) {
            WgradReduceParams rp; memset(&rp, 0, sizeof(rp));
            rp.partial = e->wg_partial; rp.bias_partial = e->wg_bias_partial; rp.nparts = wp.nparts; rp.n_in = n_in; rp.n_g = 1;
            rp.cin_total = 32 * n_in; rp.cout_total = 32; rp.shuffle = 0; rp.scale = 1.f; rp.dw = dev_dw_oihw; rp.db = dev_db;
            err = launch_wgrad_reduce(rp, s);
        }
    }
    hipStreamSynchronize(s);
    hipFree(fwd); hipFree(bwd);
    if (err != hipSuccess) return fail(XSD_ERR_HIP, "bwd launch: %s", hipGetErrorString(err));
    return XSD_OK;
}

} // extern "C"
