// xsd_split.h -- the exact 3-term bf16 split of math mode 3 ("bf16x6"), shared by conv3x3_s3x.hip and wgrad_s3x.hip.
// x = hi + mid + lo exactly (each term the round-to-nearest-even bf16 of what the previous terms left), four fp32 values
// -> three word pairs of packed bf16.  Written over the natural pairs so that hipcc emits 22 VALU instructions per float4
// (v_cvt_pk_bf16_f32, v_lshlrev/v_and to widen, v_sub_f32 for the residuals); the element-wise form it replaced compiled
// to 32 (cross-paired converts, v_mov copies and SDWA merges).  The residual subtractions are single v_sub_f32 through
// inline asm ON PURPOSE: left to itself hipcc packs them into v_pk_add_f32, and packed-f32 VALU beside running MFMAs
// costs ~13 extra cycles per instruction on this chip (MI355X_MICROARCH.md, "price of one filler beside MFMAs").
#pragma once

namespace xsd {

typedef float split_f32x4 __attribute__((ext_vector_type(4)));
typedef float split_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 split_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int split_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float split_sub(float a, float b)
{
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ void split3_pair(split_f32x2 x, unsigned int& h, unsigned int& m, unsigned int& l)
{
    h = __builtin_bit_cast(unsigned int, __builtin_convertvector(x, split_bf16x2));
    split_f32x2 hf;
    hf[0] = __builtin_bit_cast(float, h << 16);
    hf[1] = __builtin_bit_cast(float, h & 0xffff0000u);
    split_f32x2 r;
    r[0] = split_sub(x[0], hf[0]);
    r[1] = split_sub(x[1], hf[1]);
    m = __builtin_bit_cast(unsigned int, __builtin_convertvector(r, split_bf16x2));
    split_f32x2 mf;
    mf[0] = __builtin_bit_cast(float, m << 16);
    mf[1] = __builtin_bit_cast(float, m & 0xffff0000u);
    split_f32x2 q;
    q[0] = split_sub(r[0], mf[0]);
    q[1] = split_sub(r[1], mf[1]);
    l = __builtin_bit_cast(unsigned int, __builtin_convertvector(q, split_bf16x2));
}

__device__ __forceinline__ void split3_f32x4(const split_f32x4& a, split_u32x2& hi, split_u32x2& mid, split_u32x2& lo)
{
    unsigned int h0, m0, l0, h1, m1, l1;
    split3_pair(a.xy, h0, m0, l0);
    split3_pair(a.zw, h1, m1, l1);
    hi[0] = h0; hi[1] = h1;
    mid[0] = m0; mid[1] = m1;
    lo[0] = l0; lo[1] = l1;
}

// ---- math mode 4 ("f16x3"): two-term fp16 split of a SCALED fp32 value --------------------------------------------------
// xs = x * s (s a power of two chosen from the tensor's max |x| so that |xs| < 2^14: scale_for_amax below),
// h = fp16(xs) (RNE, 11 significant bits), l = fp16((xs - h) * 2^11) (the next 11 bits, kept at h's magnitude so that it never
// falls into the fp16 subnormals before h does).  xs = h + l * 2^-11 up to 2^-22 |xs| (half an ulp of l), i.e. 22-23 of
// fp32's 24 significant bits; the product of two such operands is taken as h*h + (h*l + l*h) * 2^-11 (three fp16 MFMAs,
// the two cross products into a second accumulator).  Network-level error against float64: tools/sim_split3.py and
// tests/test_hip_precision.py.  12 VALU per float4.
typedef _Float16 split_f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2_f16_pair(float x0, float x1, float s, unsigned int& h, unsigned int& l)
{
    // Mixed-precision fma instructions do scale, convert and subtract in one step each (12 VALU per float4 instead of the 20
    // of v_mul + v_cvt_pk_f16_f32 + v_cvt_f32_f16 + v_sub + v_mul + v_cvt_pk): v_fma_mixlo/hi_f16 write f16(a * b + c) (one
    // RNE rounding of the exact fp32 product) into the low / high half of a register, v_fma_mix_f32 takes the f16 half of h
    // as its addend directly.  Same values as the long form, bit for bit.
    // (one asm statement: between separate statements hipcc pads every output with an s_nop)
    unsigned int hh, ll;
    float r0, r1;
    const float k = 2048.f;
    asm("v_fma_mixlo_f16 %[h], %[x0], %[s], 0\n\t"
        "v_fma_mixhi_f16 %[h], %[x1], %[s], 0\n\t"
        "v_fma_mix_f32 %[r0], %[x0], %[s], -%[h] op_sel_hi:[0,0,1]\n\t"                  // x0*s - h.lo (exact)
        "v_fma_mix_f32 %[r1], %[x1], %[s], -%[h] op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"   // x1*s - h.hi
        "v_fma_mixlo_f16 %[l], %[r0], %[k], 0\n\t"
        "v_fma_mixhi_f16 %[l], %[r1], %[k], 0"
        : [h] "=&v"(hh), [l] "=&v"(ll), [r0] "=&v"(r0), [r1] "=&v"(r1)
        : [x0] "v"(x0), [x1] "v"(x1), [s] "s"(s), [k] "s"(k));
    h = hh;
    l = ll;
}

__device__ __forceinline__ void split2_f16x4(const split_f32x4& a, float s, split_u32x2& hi, split_u32x2& lo)
{
    // both pairs in one statement, interleaved: two independent dependency chains for the in-order staging wave
    unsigned int h0, h1, l0, l1;
    float r0, r1, r2, r3;
    const float k = 2048.f;
    asm("v_fma_mixlo_f16 %[h0], %[x0], %[s], 0\n\t"
        "v_fma_mixlo_f16 %[h1], %[x2], %[s], 0\n\t"
        "v_fma_mixhi_f16 %[h0], %[x1], %[s], 0\n\t"
        "v_fma_mixhi_f16 %[h1], %[x3], %[s], 0\n\t"
        "v_fma_mix_f32 %[r0], %[x0], %[s], -%[h0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %[r2], %[x2], %[s], -%[h1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %[r1], %[x1], %[s], -%[h0] op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %[r3], %[x3], %[s], -%[h1] op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %[l0], %[r0], %[k], 0\n\t"
        "v_fma_mixlo_f16 %[l1], %[r2], %[k], 0\n\t"
        "v_fma_mixhi_f16 %[l0], %[r1], %[k], 0\n\t"
        "v_fma_mixhi_f16 %[l1], %[r3], %[k], 0"
        : [h0] "=&v"(h0), [h1] "=&v"(h1), [l0] "=&v"(l0), [l1] "=&v"(l1), [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3)
        : [x0] "v"(a[0]), [x1] "v"(a[1]), [x2] "v"(a[2]), [x3] "v"(a[3]), [s] "s"(s), [k] "s"(k));
    hi[0] = h0; hi[1] = h1;
    lo[0] = l0; lo[1] = l1;
}

// power of two s with amax * s in [2^13, 2^14) (1 for amax = 0, subnormal, inf or NaN); exponent clamped to +-60 so that
// products of two scales and their reciprocals stay finite.  `inv` receives 1 / s.
__device__ __forceinline__ float scale_for_amax(float amax, float& inv)
{
    const unsigned int u = __builtin_bit_cast(unsigned int, amax);
    const int ex = (int)((u >> 23) & 0xff);
    int k = 140 - ex;                       // amax in [2^(ex-127), 2^(ex-126))  ->  s = 2^(14 - (ex - 126))
    if (ex == 0 || ex == 255) k = 0;
    k = k > 60 ? 60 : (k < -60 ? -60 : k);
    inv = __builtin_bit_cast(float, (unsigned int)(127 - k) << 23);
    return __builtin_bit_cast(float, (unsigned int)(127 + k) << 23);
}

} // namespace xsd
