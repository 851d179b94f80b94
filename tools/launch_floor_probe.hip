// launch_floor_probe.hip -- what ONE dependent launch of a conv-shaped grid costs on this device before it computes anything
// (round 6: the batch-1 operating points of the reference are launch-bound -- 61 conv launches of ~40-75 us per 416 x 416 image).
// N back-to-back launches on one stream (each waits for the one before it, as the layers of the net do) of a kernel with the conv
// kernel's launch shape: 256 workgroups x 768 threads, 123,648 B of dynamic LDS, an 888-byte argument block.  Variants:
//   empty            nothing but s_endpgm                                   -> dispatch + completion + barrier-to-next-dispatch
//   small            the same with 64 threads and no LDS                    -> what the big workgroup shape adds
//   write            every workgroup stores 84.5 KiB (21.6 MB per launch = a 416 x 416 plane) and reads it back in the next launch
//                                                                           -> + end-of-kernel L2 write-back / invalidate across the 8 XCDs
//   chain            like `write`, and the workgroup first does a dependent chain of 3 global round trips (argument -> slot -> data)
//                                                                           -> + the prologue's serialised latencies
// build: hipcc -O3 --offload-arch=gfx950 tools/launch_floor_probe.hip -o tools/bin/launch_floor_probe (tools/bin/ is git-ignored but travels with gpurun); run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Args { float* buf; const float* const* slots; int mode; int words_per_wg; char pad[888 - 24]; };

__global__ __launch_bounds__(768) void shaped_kernel(const Args a)
{
    extern __shared__ char lds[];
    if (a.mode == 0) return;
    if (a.mode >= 10) {                    // same-address atomics at the end of a launch: mode 10 = one per wave of 8 (the conv's MFMA waves), 11 = one per workgroup
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane == 63 && w >= 4 && (a.mode == 10 || w == 4)) atomicMax(reinterpret_cast<unsigned int*>(a.buf), (unsigned int)(blockIdx.x * 16 + w));
        return;
    }
    float s = 1.f;
    if (a.mode == 2) {                     // dependent chain: slot pointer -> value -> index
        const float* p = a.slots[blockIdx.x & 7];
        const float v = p[0];
        s = a.buf[(int)v + threadIdx.x];
    }
    float* mine = a.buf + (size_t)blockIdx.x * a.words_per_wg;
    for (int i = threadIdx.x * 4; i < a.words_per_wg; i += 768 * 4) {
        float4 v = *reinterpret_cast<float4*>(mine + i);
        v.x = v.x * 0.5f + s; v.y += 1.f; v.z *= 0.25f; v.w += s;
        *reinterpret_cast<float4*>(mine + i) = v;
    }
    if (a.mode < 0) lds[threadIdx.x] = 1;
}
__global__ __launch_bounds__(64) void small_kernel(const Args a) { if (a.mode < 0) a.buf[0] = 1.f; }

static double run(const char* name, int n, int threads, int lds, Args a, bool small)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < n; ++i) {
            if (small) hipLaunchKernelGGL(small_kernel, dim3(256), dim3(64), 0, 0, a);
            else hipLaunchKernelGGL(shaped_kernel, dim3(256), dim3(threads), lds, 0, a);
        }
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("  %-46s %8.2f us per launch (%d launches)\n", name, 1e3 * ms / n, n);
    return 1e3 * ms / n;
}

int main()
{
    const int lds = 123648, n = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&shaped_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int words = 21632;                // 84.5 KiB per workgroup: 256 of them = one 416 x 416 x 32-channel fp32 plane
    float* buf = nullptr;
    hipMalloc(&buf, sizeof(float) * (size_t)words * 256 + 4096);
    hipMemset(buf, 0, sizeof(float) * (size_t)words * 256 + 4096);
    float* slotv = nullptr;
    hipMalloc(&slotv, 64);
    hipMemset(slotv, 0, 64);
    const float** slots = nullptr;
    hipMalloc(&slots, 8 * sizeof(float*));
    std::vector<const float*> h(8, slotv);
    hipMemcpy(slots, h.data(), 8 * sizeof(float*), hipMemcpyHostToDevice);
    Args a{};
    a.buf = buf; a.slots = slots; a.words_per_wg = words;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("launch floor on %s (%d CUs), dependent launches on one stream\n", prop.name, prop.multiProcessorCount);
    a.mode = 0; run("small: 256 x 64 threads, no LDS, empty", n, 64, 0, a, true);
    a.mode = 0; run("empty: 256 x 768 threads, 123,648 B LDS", n, 768, lds, a, false);
    a.mode = 1; run("write: + 84.5 KiB read-modify-write per WG", n, 768, lds, a, false);
    a.mode = 2; run("chain: + 3 dependent global round trips first", n, 768, lds, a, false);
    a.mode = 1; a.words_per_wg = 4096; run("write 16 KiB per WG (4 MB per launch: L2-sized)", n, 768, lds, a, false);
    a.mode = 10; run("empty + 8 atomicMax per WG to ONE address (2048)", n, 768, lds, a, false);
    a.mode = 11; run("empty + 1 atomicMax per WG to ONE address (256)", n, 768, lds, a, false);
    return 0;
}
