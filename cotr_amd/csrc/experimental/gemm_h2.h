// Research (round-3 verdict item 8): fp32 products from SPLIT-f16 MFMAs - libcotr_hip_exp.so only, never on a product path.
//
// An fp32 number a is carried as ONE packed dword (h | l << 16): h = f16(a), l = f16((a - float(h)) * 2^11) - 22 significand bits, the
// same 4 bytes per element as fp32, so every tile / LDS-DMA / swizzle decision of the large-tile kernel applies unchanged (and a convolution
// gathers packed pixels exactly like fp32 ones).  A product a*b is then  h_a*h_b + 2^-11 (h_a*l_b + l_a*h_b)  [l_a*l_b, 2^-22, dropped]:
// three v_mfma_f32_32x32x16_f16 (8 passes each) per 32x32x16 block against eight v_mfma_f32_32x32x2_f32 (16 passes each) - 5.3x fewer matrix
// pipe cycles.  The cross terms accumulate in their own fp32 accumulator (scaled by 2^11 so that l stays in f16's normal range) and join the
// main one once, in the epilogue.  Error model and measurements: tools/split_mfma_numerics.py, profiles/r4_split_f16_*.
// Limits (why this is research): |a| must stay below 65504 (f16 overflow -> inf / NaN where fp32 is finite) and operands below ~1e-4 lose
// the l term to f16 subnormals; results are NOT bit-identical to the fp32-MFMA path (they are as close to the fp64 truth).
#pragma once
#include <hip/hip_fp16.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned int h2_pack(const float a) {
  const __half h = __float2half_rn(a);
  const __half l = __float2half_rn((a - __half2float(h)) * 2048.f);
  return (unsigned int)__half_as_ushort(h) | ((unsigned int)__half_as_ushort(l) << 16);
}

// ... and with the RANGE CHECK of round 5: f16(a) overflows from |a| >= 65504 on (inf where fp32 is finite).  Every place that packs
// ACTIVATIONS raises a device flag there (h2_overflow_flag(), gemm_h2.hip); the ABI entry points of the research library read it
// after a split-f16 pass and re-run the pass on the fp32-MFMA kernels when it is set (experimental/api.hip: h2_guarded).
__device__ __forceinline__ unsigned int h2_pack_chk(const float a, int* ovf) {
  if (!(fabsf(a) < 65504.f)) *ovf = 1;                  // (also NaN / inf inputs: the fp32 path decides what they mean)
  return h2_pack(a);
}

__device__ __forceinline__ float h2_unpack(const unsigned int d) {
  return fmaf(__half2float(__ushort_as_half((unsigned short)(d >> 16))), 0x1p-11f, __half2float(__ushort_as_half((unsigned short)(d & 0xFFFFu))));
}

// 8 packed dwords (k = 0..7 of one row) -> the 8 hi halves / the 8 lo halves, each as an MFMA operand
__device__ __forceinline__ void h2_unzip(const u32x4 d0, const u32x4 d1, f16x8& hi, f16x8& lo) {
  u32x4 h, l;
  h[0] = __builtin_amdgcn_perm(d0[1], d0[0], 0x05040100u);
  h[1] = __builtin_amdgcn_perm(d0[3], d0[2], 0x05040100u);
  h[2] = __builtin_amdgcn_perm(d1[1], d1[0], 0x05040100u);
  h[3] = __builtin_amdgcn_perm(d1[3], d1[2], 0x05040100u);
  l[0] = __builtin_amdgcn_perm(d0[1], d0[0], 0x07060302u);
  l[1] = __builtin_amdgcn_perm(d0[3], d0[2], 0x07060302u);
  l[2] = __builtin_amdgcn_perm(d1[1], d1[0], 0x07060302u);
  l[3] = __builtin_amdgcn_perm(d1[3], d1[2], 0x07060302u);
  hi = __builtin_bit_cast(f16x8, h);
  lo = __builtin_bit_cast(f16x8, l);
}

// one 32-deep K step of a wavefront's 64 x (32 TN) sub-tile: two k16 slices, 3 MFMAs per (A block, B block, slice).  As / Ws: this lane's
// fragment row (32 dwords, 16-byte chunks XOR-swizzled with sw) of the first A / W block; the blocks are 32 rows = 32 * 32 dwords apart.
template <int TN>
__device__ __forceinline__ void h2_kstep(const float* As, const float* Ws, const int hh, const int sw, f32x16 (&acc)[2][TN],
                                         f32x16 (&accx)[2][TN]) {
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const int ch0 = ((s2 * 4 + hh) ^ sw) * 4, ch1 = ((s2 * 4 + 2 + hh) ^ sw) * 4;      // chunks j = 2 s2, 2 s2 + 1 of gemm_big_body
    f16x8 ah[2], al[2], bh[TN], bl[TN];
#pragma unroll
    for (int a = 0; a < 2; ++a)
      h2_unzip(*reinterpret_cast<const u32x4*>(As + a * 32 * 32 + ch0), *reinterpret_cast<const u32x4*>(As + a * 32 * 32 + ch1), ah[a], al[a]);
#pragma unroll
    for (int b = 0; b < TN; ++b)
      h2_unzip(*reinterpret_cast<const u32x4*>(Ws + b * 32 * 32 + ch0), *reinterpret_cast<const u32x4*>(Ws + b * 32 * 32 + ch1), bh[b], bl[b]);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
        accx[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bl[b], accx[a][b], 0, 0, 0);
        accx[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bh[b], accx[a][b], 0, 0, 0);
      }
  }
}
