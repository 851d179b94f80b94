// mfma_stream_probe.hip -- the rate the matrix pipes of THIS device sustain on the conv kernels' own MFMA stream.
//
// A measurement entry of the C ABI (include/xsd.h: xsd_probe_mfma_stream), not a kernel of the hot path.  bench.py prices the
// conv kernel against the nominal dense 16-bit MFMA peak (2.5 PFLOP/s at 2.4 GHz); at the 1400 W package cap the chip does not
// hold 2.4 GHz under a dense MFMA stream, so the line also carries what the bare stream reaches on the device the bench ran on
// (`roofline.sustained_peak`), measured in the same process.
//
// The stream is that of one MFMA wave of conv3x3_h2x.hip / conv3x3_s3x.hip for 2 output rows x 32 pixels x 32 output channels
// x 9 taps x 16 input channels: eight waves per workgroup (two per SIMD), one workgroup per CU, an LDS image filled once with
// realistic split operands (N(0,1) variates scaled into the format's range; the low terms are the next significand bits), per
// half-step 18 (weight fragment, input fragment) pairs x NPROD products of v_mfma_f32_32x32x16_{f16,bf16}, their fragments
// read from LDS two steps ahead with ds_read_b128, one barrier per half-step; no staging waves, no global loads, no epilogue.
//   fmt 0: f16, two-term fragments, 3 products per pair (54 MFMAs, 42 ds_read_b128 per wave and half-step)
//   fmt 1: bf16, three-term fragments, 6 products per pair (108 MFMAs, 63 ds_read_b128)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <vector>

#include <algorithm>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int HALO_W = 34, ROWB = HALO_W * 32, XT = 20 * 1024;       // one term image of the input tile: 20 KiB
template <int NT> struct Lay {
    static constexpr int WOFF = NT * XT, WB = 10 * NT * 1024, BYTES = WOFF + WB;
};

template <bool BF> struct Frag;
template <> struct Frag<false> { typedef f16x8 T; static constexpr int NT = 2; };
template <> struct Frag<true> { typedef bf16x8 T; static constexpr int NT = 3; };

template <bool BF>
__global__ __launch_bounds__(512) void mfma_stream_kernel(const unsigned int* __restrict__ src, float* __restrict__ out,
                                                          unsigned long long* __restrict__ clk, int iters)
{
    typedef typename Frag<BF>::T V;
    constexpr int NT = Frag<BF>::NT;
    typedef Lay<NT> L;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < L::BYTES / 4; i += 512) reinterpret_cast<unsigned int*>(smem)[i] = src[i];
    __syncthreads();
    const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const char* xc = smem;
    const char* wlane = smem + L::WOFF + ln * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc[2], accx[2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 16; ++k) { acc[r][k] = 0.f; accx[r][k] = 0.f; }
    int abase[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int hx = (ln & 31) + dx;
        abase[dx] = (wv * 2) * ROWB + hx * 32 + (((ln >> 5) ^ ((hx >> 3) & 1)) << 4);
    }
    auto load_w = [&](int tap, V (&b)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = *reinterpret_cast<const V*>(wlane + (tap * NT + t) * 1024);
    };
    auto load_x = [&](int ir, int dx, V (&a)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) a[t] = *reinterpret_cast<const V*>(xc + t * XT + abase[dx] + ir * ROWB);
    };
    auto mac = [&](int r, const V (&w)[NT], const V (&x)[NT]) {
        if constexpr (BF) {
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[2], accx[r], 0, 0, 0);
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], x[0], accx[r], 0, 0, 0);
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[1], accx[r], 0, 0, 0);
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[0], acc[r], 0, 0, 0);
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[1], accx[r], 0, 0, 0);
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[0], accx[r], 0, 0, 0);
        } else {
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0], x[1], accx[r], 0, 0, 0);
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[1], x[0], accx[r], 0, 0, 0);
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0], x[0], acc[r], 0, 0, 0);
        }
    };
    V xf[3][NT], wf[3][NT];
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        load_x(0, 0, xf[0]);
        load_w(0, wf[0]);
        load_x(1, 0, xf[1]);
        load_w(3, wf[1]);
#pragma unroll
        for (int s = 0; s < 12; ++s) {        // (column offset dx, input row ir): input row ir feeds output row 0 (tap row ir) and row 1 (tap row ir - 1)
            const int dx = s >> 2, ir = s & 3;
            if (s + 2 < 12) {
                const int dx2 = (s + 2) >> 2, ir2 = (s + 2) & 3;
                load_x(ir2, dx2, xf[(s + 2) % 3]);
                if (ir2 <= 2) load_w(ir2 * 3 + dx2, wf[(3 * dx2 + ir2) % 3]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ir >= 1) mac(1, wf[(3 * dx + ir - 1) % 3], xf[s % 3]);
            if (ir <= 2) mac(0, wf[(3 * dx + ir) % 3], xf[s % 3]);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    float sink = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 16; ++k) sink += acc[r][k] + accx[r][k];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = sink;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// deterministic N(0,1)-like variates (sum of 12 uniforms - 6): operand bits as a trained layer's, which is what the matrix
// pipes' power -- and through the package cap their clock -- depends on
struct Lcg {
    uint64_t s;
    float uni() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 40) & 0xffffff) * (1.0f / 16777216.0f); }
    float normal() { float a = -6.f; for (int i = 0; i < 12; ++i) a += uni(); return a; }
};

unsigned short f16_bits(float f) { _Float16 h = (_Float16)f; unsigned short u; __builtin_memcpy(&u, &h, 2); return u; }
unsigned short bf16_trunc(float f) { unsigned int u; __builtin_memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
float bf16_val(unsigned short b) { unsigned int u = (unsigned int)b << 16; float f; __builtin_memcpy(&f, &u, 4); return f; }

void fill_image(std::vector<unsigned short>& img, bool bf, int nt)
{
    Lcg g{12345};
    const int woff = nt * XT;
    // element i of term t: input image at t * XT + 2 i; weight fragment of tap k at woff + (k * nt + t) * 1024 + 2 i
    auto put = [&](size_t base_bytes, size_t term_stride_bytes, size_t n) {
        for (size_t i = 0; i < n; ++i) {
            if (!bf) {
                const float x = g.normal() * 2048.f;
                const _Float16 h = (_Float16)x;
                img[(base_bytes) / 2 + i] = f16_bits((float)h);
                img[(base_bytes + term_stride_bytes) / 2 + i] = f16_bits((x - (float)h) * 2048.f);
            } else {
                float x = g.normal();
                for (int t = 0; t < 3; ++t) {
                    const unsigned short b = bf16_trunc(x);
                    img[(base_bytes + t * term_stride_bytes) / 2 + i] = b;
                    x -= bf16_val(b);
                }
            }
        }
    };
    put(0, XT, XT / 2);
    for (int tap = 0; tap < 10; ++tap) put(woff + (size_t)tap * nt * 1024, 1024, 512);
}

template <bool BF>
hipError_t run_probe(double seconds, double* tflops, double* ghz, hipStream_t st)
{
    constexpr int NT = Frag<BF>::NT;
    typedef Lay<NT> L;
    int dev = 0;
    hipDeviceProp_t prop;
    hipError_t e;
    if ((e = hipGetDevice(&dev)) != hipSuccess || (e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
    const int G = prop.multiProcessorCount;
    std::vector<unsigned short> img(L::BYTES / 2);
    fill_image(img, BF, NT);
    unsigned int* d_src = nullptr; float* d_out = nullptr; unsigned long long* d_clk = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t rc = hipSuccess;
    auto fail = [&](const char*, hipError_t err) { rc = err; };
    if ((e = hipMalloc(&d_src, L::BYTES)) != hipSuccess || (e = hipMalloc(&d_out, sizeof(float) * 512 * G)) != hipSuccess ||
        (e = hipMalloc(&d_clk, sizeof(unsigned long long) * 2 * G)) != hipSuccess) fail("hipMalloc", e);
    if (rc == hipSuccess && (e = hipMemcpyAsync(d_src, img.data(), L::BYTES, hipMemcpyHostToDevice, st)) != hipSuccess) fail("hipMemcpy", e);
    if (rc == hipSuccess && (e = hipStreamSynchronize(st)) != hipSuccess) fail("sync", e);
    if (rc == hipSuccess && (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_stream_kernel<BF>), hipFuncAttributeMaxDynamicSharedMemorySize, L::BYTES)) != hipSuccess) fail("LDS size", e);
    if (rc == hipSuccess && ((e = hipEventCreate(&e0)) != hipSuccess || (e = hipEventCreate(&e1)) != hipSuccess)) fail("hipEventCreate", e);
    if (rc == hipSuccess) {
        // launches of ~25 ms (8192 half-steps) until `seconds` have passed: the package-power governor settles within the first
        // few hundred ms; the SECOND half of the launches is what is reported
        const int iters = 8192;
        const int nmfma = BF ? 108 : 54;
        std::vector<float> ms;
        std::vector<double> gz;
        std::vector<unsigned long long> clk(2 * G);
        double total = 0.0;
        while (total < seconds * 1e3 && ms.size() < 4096) {
            if ((e = hipEventRecord(e0, st)) != hipSuccess) { fail("hipEventRecord", e); break; }
            hipLaunchKernelGGL(mfma_stream_kernel<BF>, dim3(G), dim3(512), L::BYTES, st, d_src, d_out, d_clk, iters);
            // a rejected launch (the 60 / 90 KiB LDS request, 512 threads) must not come back as "0 ms": inf TFLOP/s with rc = success
            if ((e = hipGetLastError()) != hipSuccess) { fail("launch", e); break; }
            if ((e = hipEventRecord(e1, st)) != hipSuccess) { fail("hipEventRecord", e); break; }
            if ((e = hipEventSynchronize(e1)) != hipSuccess) { fail("kernel", e); break; }
            float m = 0.f;
            if ((e = hipEventElapsedTime(&m, e0, e1)) != hipSuccess) { fail("hipEventElapsedTime", e); break; }
            if (!(m > 0.f)) { fail("no elapsed time", hipErrorUnknown); break; }
            if ((e = hipMemcpy(clk.data(), d_clk, sizeof(unsigned long long) * 2 * G, hipMemcpyDeviceToHost)) != hipSuccess) { fail("hipMemcpy", e); break; }
            std::vector<double> g(G);
            for (int i = 0; i < G; ++i) g[i] = clk[2 * i + 1] ? (double)clk[2 * i] / ((double)clk[2 * i + 1] * 10.0) : 0.0;    // s_memrealtime: 100 MHz
            std::nth_element(g.begin(), g.begin() + G / 2, g.end());
            ms.push_back(m); gz.push_back(g[G / 2]);
            total += m;
        }
        if (rc == hipSuccess && !ms.empty()) {
            double tm = 0.0, tg = 0.0;
            const size_t from = ms.size() / 2;
            for (size_t i = from; i < ms.size(); ++i) { tm += ms[i]; tg += gz[i]; }
            const double n = (double)(ms.size() - from);
            const double flop = (double)G * 8.0 * nmfma * 32768.0 * iters;      // 32x32x16 MFMA = 2 * 32 * 32 * 16 FLOP
            if (tflops) *tflops = flop / (tm / n * 1e-3) / 1e12;
            if (ghz) *ghz = tg / n;
        }
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    hipFree(d_src); hipFree(d_out); hipFree(d_clk);
    return rc;
}

}  // namespace

namespace xsd {
// fmt 0: f16 stream (f16x3's), 1: bf16 stream (bf16x6's); dense 16-bit MFMA TFLOP/s and the in-kernel shader clock over the
// second half of `seconds` of back-to-back launches (the C-ABI wrapper with the argument checks: xsd_engine.hip)
hipError_t probe_mfma_stream(int fmt, double seconds, double* mfma_tflops, double* sclk_ghz, hipStream_t stream)
{
    return fmt == 0 ? run_probe<false>(seconds, mfma_tflops, sclk_ghz, stream) : run_probe<true>(seconds, mfma_tflops, sclk_ghz, stream);
}
}  // namespace xsd
