// mfma_probe.hip -- how does v_mfma_f32_32x32x16_f16 round (f16 twin of mfma_probe.hip)?  (diagnostic; decides whether a 3-term bf16 split with fp32
// accumulation in the MFMA is "fp32-class".)  Every row of A is (a_0..a_15), every column of B is (b_0..b_15), so each of the
// 1024 outputs is  C + sum_k a_k*b_k  evaluated by the matrix core.  Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void probe(const float* a, const float* b, const float* c, float* d, int chain)
{
    const int h = threadIdx.x >> 5;
    bf16x8 av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = (_Float16)a[8 * h + j]; bv[j] = (_Float16)b[8 * h + j]; }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = c[0];
    for (int r = 0; r < chain; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
    if (threadIdx.x == 0) d[0] = acc[0];
    if (threadIdx.x == 37) d[1] = acc[5];
}

static float run(const std::vector<float>& a, const std::vector<float>& b, float c, int chain = 1)
{
    static float *da = nullptr, *db, *dc, *dd;
    if (!da) { hipMalloc(&da, 64); hipMalloc(&db, 64); hipMalloc(&dc, 4); hipMalloc(&dd, 8); }
    hipMemcpy(da, a.data(), 64, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), 64, hipMemcpyHostToDevice);
    hipMemcpy(dc, &c, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dc, dd, chain);
    float out[2];
    hipMemcpy(out, dd, 8, hipMemcpyDeviceToHost);
    if (out[0] != out[1]) printf("  (outputs differ across lanes: %a vs %a)\n", out[0], out[1]);
    return out[0];
}

int main()
{
    std::vector<float> a(16), b(16);
    auto fill = [&](float av, float bv) { for (int k = 0; k < 16; ++k) { a[k] = av; b[k] = bv; } };
    // T1: 16 products of 2^-25 beside C = 1
    fill(ldexpf(1, -12), ldexpf(1, -13));
    printf("T1 C=1 + 16 x 2^-25 : %a   (exact 1+2^-21 = %a; sequential fp32 RN = 1)\n", run(a, b, 1.f), 1.0 + ldexp(1, -21));
    // T2: one product at / just above half an ulp of C
    fill(0.f, 0.f); a[3] = ldexpf(1, -12); b[3] = ldexpf(1, -12);
    printf("T2a C=1 + 2^-24 (tie) : %a   (RNE 1; away 1+2^-23)\n", run(a, b, 1.f));
    a[3] = ldexpf(1.0078125f, -12);
    printf("T2b C=1 + 2^-24(1+2^-7) : %a   (RN 1+2^-23 = %a; truncation 1)\n", run(a, b, 1.f), 1.0 + ldexp(1, -23));
    // T3: small negative product
    a[3] = -ldexpf(1, -12); b[3] = ldexpf(1, -13);
    printf("T3 C=1 - 2^-25 : %a   (RN 1; toward zero 1-2^-24 = %a)\n", run(a, b, 1.f), 1.0 - ldexp(1, -24));
    // T4: many half-ulp products beside a big C
    fill(0.5f, 1.f);
    printf("T4 C=2^24 + 16 x 0.5 : %.1f   (exact %.1f; sequential RNE %.1f)\n", run(a, b, 16777216.f), 16777216.0 + 8, 16777216.0);
    // T5/T6: alignment window: products {1, -1, 2^-k}; C = 0
    printf("T6 C=0, products {1, -1, 2^-k}: k ->");
    for (int k = 20; k <= 28; k += 2) {
        fill(0.f, 0.f); a[0] = 1.f; b[0] = 1.f; a[1] = -1.f; b[1] = 1.f; a[9] = ldexpf(1, -(k / 2)); b[9] = ldexpf(1, -(k - k / 2));
        const float r = run(a, b, 0.f);
        printf(" %d:%s", k, r == ldexpf(1, -k) ? "ok" : (r == 0.f ? "LOST" : "part"));
    }
    printf("\n");
    printf("T7 C=1, products {-1, 2^-k}: k ->");
    for (int k = 20; k <= 28; k += 2) {
        fill(0.f, 0.f); a[1] = -1.f; b[1] = 1.f; a[9] = ldexpf(1, -(k / 2)); b[9] = ldexpf(1, -(k - k / 2));
        const float r = run(a, b, 1.f);
        printf(" %d:%s", k, r == ldexpf(1, -k) ? "ok" : (r == 0.f ? "LOST" : "part"));
    }
    printf("\n");
    printf("T8 C=2^30, products {-2^15*2^15, 2^-k}: k ->");
    for (int k = 0; k <= 28; k += 2) {
        fill(0.f, 0.f); a[1] = -32768.f; b[1] = 32768.f; a[9] = ldexpf(1, -(k / 2)); b[9] = ldexpf(1, -(k - k / 2));
        const float r = run(a, b, 1073741824.f);
        printf(" %d:%s", k, r == ldexpf(1, -k) ? "ok" : (r == 0.f ? "LOST" : "part"));
    }
    printf("\n");
    // T9: statistics: random bf16 data, chain of 90 MFMAs (K = 1440) vs float64, vs a sequential fp32 fmaf chain on the host
    {
        unsigned s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
        auto tobf = [](float x) { return (float)(_Float16)x; };
        double e_mfma = 0, e_seq = 0, ref2 = 0;
        for (int trial = 0; trial < 200; ++trial) {
            for (int k = 0; k < 16; ++k) { a[k] = tobf(rnd()); b[k] = tobf(rnd()); }
            const float got = run(a, b, 0.f, 90);
            double ex = 0; float sq = 0.f;
            for (int r = 0; r < 90; ++r) for (int k = 0; k < 16; ++k) { ex += (double)a[k] * b[k]; sq = fmaf(a[k], b[k], sq); }
            e_mfma += (got - ex) * (got - ex); e_seq += (sq - ex) * (sq - ex); ref2 += ex * ex;
        }
        printf("T9 K=1440 same-sign-ish chains: rel rms err  mfma %.3e   sequential fmaf %.3e\n", sqrt(e_mfma / ref2), sqrt(e_seq / ref2));
    }
    return 0;
}
