// p16.h -- the "P16" feature-plane format used by math mode 2 (bf16x3 with pre-split activations).
//
// A P16 plane has the same geometry and byte size as an fp32 plane ([B][H][W] pixels x 128 B), but a pixel holds its
// 32 channels as two bf16 terms, value = hi + lo (16 significant bits):
//     bytes [0,64)   : hi[pos], pos = 0..31
//     bytes [64,128) : lo[pos]
// and channels are stored in "accumulator order": channel ch = 8q + 4h + t (q<4, h<2, t<4) sits at position
//     pos(ch) = 16h + 4q + t,
// which is exactly the order in which one MFMA lane-half holds its 16 output channels, so a conv epilogue stores
// 32 contiguous bytes of hi and 32 of lo per lane, and a 16-channel MFMA k-step is one contiguous 32-B half of each.
// The consumer convs split every activation into hi+lo anyway (bf16x3), so storing the split loses nothing for them;
// residual adds see 16-bit-mantissa values (whole-net error ~5e-6 of max, simulated and measured, tolerance 1e-3).
// Because the stored format IS the MFMA operand format, tiles go HBM -> LDS by LDS-DMA with no VALU work at all.
#pragma once
#include <hip/hip_runtime.h>

namespace xsd {

__host__ __device__ __forceinline__ int p16_pos(int ch) { return 16 * ((ch >> 2) & 1) + 4 * (ch >> 3) + (ch & 3); }
__host__ __device__ __forceinline__ int p16_ch(int pos) { return 8 * ((pos & 15) >> 2) + 4 * (pos >> 4) + (pos & 3); }

typedef unsigned int p16u4 __attribute__((ext_vector_type(4)));
typedef unsigned int p16u2 __attribute__((ext_vector_type(2)));
typedef float p16f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned int p16_bf(float x) { return (unsigned int)__builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float p16_f(unsigned int bits16) { return __builtin_bit_cast(float, bits16 << 16); }

// split two floats into packed (hi0 | hi1<<16), (lo0 | lo1<<16)
__device__ __forceinline__ void p16_split2(float a, float b, unsigned int& hi, unsigned int& lo)
{
    const unsigned int ha = p16_bf(a), hb = p16_bf(b);
    const float fa = p16_f(ha), fb = p16_f(hb);
    hi = ha | (hb << 16);
    lo = p16_bf(a - fa) | (p16_bf(b - fb) << 16);
}
// value pair from packed hi/lo words
__device__ __forceinline__ void p16_join2(unsigned int hi, unsigned int lo, float& a, float& b)
{
    a = __builtin_bit_cast(float, hi << 16) + __builtin_bit_cast(float, lo << 16);
    b = __builtin_bit_cast(float, hi & 0xffff0000u) + __builtin_bit_cast(float, lo & 0xffff0000u);
}
// 4 consecutive positions (8 B of hi, 8 B of lo)
__device__ __forceinline__ p16f4 p16_load4(const char* px, int pos)
{
    const p16u2 h = *reinterpret_cast<const p16u2*>(px + pos * 2);
    const p16u2 l = *reinterpret_cast<const p16u2*>(px + 64 + pos * 2);
    float a, b, c, d;
    p16_join2(h[0], l[0], a, b);
    p16_join2(h[1], l[1], c, d);
    p16f4 v = {a, b, c, d};
    return v;
}
__device__ __forceinline__ void p16_store4(char* px, int pos, const p16f4& v)
{
    unsigned int h0, l0, h1, l1;
    p16_split2(v[0], v[1], h0, l0);
    p16_split2(v[2], v[3], h1, l1);
    const p16u2 h = {h0, h1}, l = {l0, l1};
    *reinterpret_cast<p16u2*>(px + pos * 2) = h;
    *reinterpret_cast<p16u2*>(px + 64 + pos * 2) = l;
}

} // namespace xsd
