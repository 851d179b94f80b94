// conv_core.h -- the MFMA inner loop of conv3x3_mfma.hip (exact-fp32 mode, 8x32 tiles): 9 taps x 2 tile rows,
// software-pipelined by hand at (tap,row) granularity: the fragments of stage i+1 are requested BEFORE the MFMAs of
// stage i (two named fragment sets; after full unrolling every index is a compile-time constant, so the compiler keeps
// both sets in registers and emits counted lgkmcnt waits).
#pragma once
#include <hip/hip_runtime.h>

namespace xsd {

typedef float cf32x16 __attribute__((ext_vector_type(16)));
typedef float cf32x4 __attribute__((ext_vector_type(4)));

// in_lds: input halo tile; wl: this lane's base into the weight panel (panel + lane*16);
// abase[dx][k]: per-lane byte offsets of the 4 input-fragment chunks for tap column dx (tile row 0 of this wave);
// ROWB: bytes per halo row.  acc[r]: D = W x X accumulators of the wave's two tile rows.
template <int ROWB>
__device__ __forceinline__ void conv_compute(const char* in_lds, const char* wl, const int (&abase)[3][4], cf32x16 (&acc)[2])
{
    cf32x4 bf[2][4], af[2][4];
    auto load_b = [&](int tap, cf32x4 (&b)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const cf32x4*>(wl + (tap * 4 + j) * 1024);
    };
    auto load_a = [&](int tap, int r, cf32x4 (&a)[4]) {
        const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const cf32x4*>(in_lds + abase[dx][j] + (r + dy) * ROWB);
    };
    load_b(0, bf[0]);
    load_a(0, 0, af[0]);
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        const int tap = i >> 1, r = i & 1;
        if (i + 1 < 18) {
            if (((i + 1) & 1) == 0) load_b((i + 1) >> 1, bf[((i + 1) >> 1) & 1]);
            load_a((i + 1) >> 1, (i + 1) & 1, af[(i + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0); // keep the next stage's ds_reads in front of this stage's MFMAs
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[tap & 1][j][q], af[i & 1][j][q], acc[r], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

} // namespace xsd
