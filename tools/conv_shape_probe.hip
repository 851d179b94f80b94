// conv_shape_probe.hip -- what would the f16x3 conv's MFMA waves do with v_mfma_f32_16x16x32_f16 instead of 32x32x16?
//
// A probe, not a kernel of the product: eight waves per workgroup (two per SIMD, as the conv's MFMA waves), one workgroup per
// CU, an LDS filled ONCE with realistic two-term fp16 operand images (h ~ fp16 of N(0,1) scaled to the top of the range, l =
// the next 11 bits scaled 2^11: xsd_split.h), and per "half-step" exactly the instruction mix of one MFMA wave of
// csrc/conv3x3_h2x.hip for 2 output rows x 32 pixels x 32 output channels x 9 taps x 16 input channels x 3 products:
//   shape A (shipped): 54 x v_mfma_f32_32x32x16_f16, 42 x ds_read_b128, fragments two steps ahead;
//   shape B          : K = 32 = two taps x 16 channels (lanes 0-31 tap a, lanes 32-63 tap b), nine taps = four pairs + one
//                      half-empty step: 120 x v_mfma_f32_16x16x32_f16 (2 rows x 2 pixel halves x 2 channel halves x 5 K-steps
//                      x 3 products), 60 x ds_read_b128 (an input fragment can no longer serve both output rows), fragments
//                      one step ahead (two would need 96 fragment registers beside 64 accumulators and the kernel's 32
//                      deferred-store registers: over the 168 of three waves per SIMD).
// One barrier per half-step as in the kernel; no staging waves, no stores, no epilogue.  Reports cycles per half-step
// (s_memtime), the in-kernel clock (s_memtime / s_memrealtime) and the wall time per half-step of both shapes, alternating.
// Build + run:  hipcc -O3 --offload-arch=gfx950 tools/conv_shape_probe.hip -o tools/bin/conv_shape_probe && tools/bin/conv_shape_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int HALO_W = 34, ROWB = HALO_W * 32, XT = 20 * 1024, WOFF = 2 * XT, WB = 10 * 2 * 1024, LDS_BYTES = WOFF + WB;

template <int SHAPE>
__global__ __launch_bounds__(512) void probe_kernel(const unsigned int* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ clk, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < LDS_BYTES / 4; i += 512) reinterpret_cast<unsigned int*>(smem)[i] = src[i];
    __syncthreads();
    const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const char* xc = smem;
    const char* wl = smem + WOFF;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sink = 0.f;
    if constexpr (SHAPE == 0 || SHAPE == 3) {
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
        const char* wlane = wl + ln * 16;
        auto load_w = [&](int tap, f16x8 (&b)[2]) {
#pragma unroll
            for (int t = 0; t < 2; ++t) b[t] = *reinterpret_cast<const f16x8*>(wlane + (tap * 2 + t) * 1024);
        };
        auto load_x = [&](int ir, int dx, f16x8 (&a)[2]) {
#pragma unroll
            for (int t = 0; t < 2; ++t) a[t] = *reinterpret_cast<const f16x8*>(xc + t * XT + abase[dx] + ir * ROWB);
        };
        auto mac = [&](int r, const f16x8 (&w)[2], const f16x8 (&x)[2]) {
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0], x[1], accx[r], 0, 0, 0);
            accx[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[1], x[0], accx[r], 0, 0, 0);
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0], x[0], acc[r], 0, 0, 0);
        };
        f16x8 xf[3][2], wf[3][2];
        if (SHAPE == 3) {      // fragments resident: the MFMA sequence of shape A without its 42 LDS reads per half-step (what do they cost?)
#pragma unroll
            for (int q = 0; q < 3; ++q) { load_x(q, q, xf[q]); load_w(q, wf[q]); }
        }
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
            if (SHAPE == 0) {
            load_x(0, 0, xf[0]);
            load_w(0, wf[0]);
            load_x(1, 0, xf[1]);
            load_w(3, wf[1]);
            } else {
#pragma unroll
                for (int q = 0; q < 3; ++q) { asm volatile("" : "+v"(xf[q][0]), "+v"(xf[q][1]), "+v"(wf[q][0]), "+v"(wf[q][1])); }
            }
#pragma unroll
            for (int s = 0; s < 12; ++s) {
                const int dx = s >> 2, ir = s & 3;
                if (SHAPE == 0 && s + 2 < 12) {
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
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < 16; ++k) sink += acc[r][k] + accx[r][k];
    } else {
        // output tiles [row r][pixel half p][channel half c], 16 x 16 each
        f32x4 acc[2][2][2], accx[2][2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int c = 0; c < 2; ++c) { acc[r][p][c] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[r][p][c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        // K-step j pairs taps (2j, 2j + 1) (j = 4: tap 8 and an empty half): lanes 0-31 supply tap 2j, lanes 32-63 tap 2j + 1;
        // within a half, lanes 0-15 / 16-31 the channel octets 0 / 1.  Per-lane byte offsets of the input fragment of (j, p):
        // The lane -> (pixel, octet, tap) map of such a kernel would come with its own conflict-free LDS swizzle; the probe does not
        // design one: every input fragment read takes one lane-linear 1-KiB chunk of the image (twenty chunks: (j, pixel half,
        // row)), i.e. the conflict-free ideal.  The weight fragments use the real image of the shipped kernel.
        const int kb = ln >> 4, px = ln & 15, oct = kb & 1, sel = kb >> 1;
        int xoff[5][2];
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p) xoff[j][p] = ((j * 2 + p) * 2) * 1024 + ln * 16;      // + r * 1024
        const int wbase = sel * 2048 + (oct * 32 + px) * 16;      // + j * 4096 + term * 1024 + channel half * 256
        auto load_w = [&](int j, f16x8 (&b)[2][2]) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int t = 0; t < 2; ++t) b[c][t] = *reinterpret_cast<const f16x8*>(wl + wbase + j * 4096 + t * 1024 + c * 256);
        };
        auto load_x = [&](int j, int r, f16x8 (&a)[2][2]) {
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int t = 0; t < 2; ++t) a[p][t] = *reinterpret_cast<const f16x8*>(xc + t * XT + xoff[j][p] + r * 1024);
        };
        auto half_step = [&](auto NSV) {
            constexpr int NS = decltype(NSV)::value;       // 10: five K-steps (tap pairs + the half-empty ninth); 8: four
            f16x8 xf[2][2][2], wf[2][2][2];
            load_w(0, wf[0]);
            load_x(0, 0, xf[0]);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int j = s >> 1, r = s & 1;
                if (s + 1 < NS) {
                    const int j2 = (s + 1) >> 1, r2 = (s + 1) & 1;
                    load_x(j2, r2, xf[(s + 1) & 1]);
                    if (r2 == 0) load_w(j2, wf[j2 & 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        accx[r][p][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j & 1][c][0], xf[s & 1][p][1], accx[r][p][c], 0, 0, 0);
                        accx[r][p][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j & 1][c][1], xf[s & 1][p][0], accx[r][p][c], 0, 0, 0);
                        acc[r][p][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j & 1][c][0], xf[s & 1][p][0], acc[r][p][c], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
            // SHAPE 2: the zero-waste form -- the ninth tap's two channel halves as ONE K = 32 step in every other half-step (a third
            // input buffer would keep the first half's image): four K-steps, then five = 108 MFMAs + 54 reads per half-step on average
            if (SHAPE == 2 && !(it & 1)) half_step(std::integral_constant<int, 8>{});
            else half_step(std::integral_constant<int, 10>{});
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int k = 0; k < 4; ++k) sink += acc[r][p][c][k] + accx[r][p][c][k];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = sink;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

static unsigned short f2h(float f)
{
    _Float16 h = (_Float16)f;
    unsigned short u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int only = argc > 2 ? atoi(argv[2]) : -1;      // 0 / 1: that shape only (for a power reading: tools/power_of.py)
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int G = prop.multiProcessorCount;
    // operand images: term 0 = h (11 bits of an N(0,1) value scaled to |max| ~ 2^13), term 1 = l (the next bits x 2^11, |l| <= 2^10 |ulp|)
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<unsigned short> img(LDS_BYTES / 2);
    auto fill = [&](size_t byte0, size_t bytes, bool lo) {
        for (size_t i = 0; i < bytes / 2; ++i) {
            const float x = nd(rng) * 2048.f;
            const _Float16 h = (_Float16)x;
            const float r = (x - (float)h) * 2048.f;
            img[byte0 / 2 + i] = lo ? f2h(r) : f2h((float)h);
        }
    };
    fill(0, XT, false); fill(XT, XT, true);
    for (int tap = 0; tap < 10; ++tap) { fill(WOFF + tap * 2048, 1024, false); fill(WOFF + tap * 2048 + 1024, 1024, true); }
    unsigned int* d_src; float* d_out; unsigned long long* d_clk;
    hipMalloc(&d_src, LDS_BYTES); hipMalloc(&d_out, sizeof(float) * 512 * G); hipMalloc(&d_clk, sizeof(unsigned long long) * 2 * G);
    hipMemcpy(d_src, img.data(), LDS_BYTES, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<unsigned long long> clk(2 * G);
    printf("conv_shape_probe: %d CUs, %d half-steps per launch; per wave and half-step: A = 54 x 32x32x16 + 42 ds_read_b128, B = 120 x 16x16x32 + 60 ds_read_b128\n", G, iters);
    for (int rep = 0; rep < 3; ++rep)
        for (int shape = 0; shape < 4; ++shape) {
            if (only >= 0 && shape != only) continue;
            for (int pass = 0; pass < 2; ++pass) {          // pass 0 warms the clock governor up, pass 1 is reported
                hipEventRecord(e0, 0);
                if (shape == 0) hipLaunchKernelGGL(probe_kernel<0>, dim3(G), dim3(512), LDS_BYTES, 0, d_src, d_out, d_clk, iters);
                else if (shape == 1) hipLaunchKernelGGL(probe_kernel<1>, dim3(G), dim3(512), LDS_BYTES, 0, d_src, d_out, d_clk, iters);
                else if (shape == 2) hipLaunchKernelGGL(probe_kernel<2>, dim3(G), dim3(512), LDS_BYTES, 0, d_src, d_out, d_clk, iters);
                else hipLaunchKernelGGL(probe_kernel<3>, dim3(G), dim3(512), LDS_BYTES, 0, d_src, d_out, d_clk, iters);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
            }
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(clk.data(), d_clk, sizeof(unsigned long long) * 2 * G, hipMemcpyDeviceToHost);
            std::vector<double> cyc(G), ghz(G);
            for (int g = 0; g < G; ++g) { cyc[g] = (double)clk[2 * g] / iters; ghz[g] = (double)clk[2 * g] / ((double)clk[2 * g + 1] * 10.0); }   // s_memrealtime: 100 MHz
            std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
            const double pipe = (shape == 0 || shape == 3) ? 2 * 54 * 32.0 : (shape == 1 ? 2 * 120 * 16.0 : 2 * 108 * 16.0);      // matrix-pipe cycles per half-step and SIMD (two waves)
            printf("  shape %c: %8.1f cycles per half-step (matrix pipe needs %.0f: %.0f %% busy), in-kernel clock %.3f GHz, %7.3f us per half-step wall\n",
                   "ABCD"[shape], cyc[G / 2], pipe, 100.0 * pipe / cyc[G / 2], ghz[G / 2], 1e3 * ms / iters);
        }
    return 0;
}
