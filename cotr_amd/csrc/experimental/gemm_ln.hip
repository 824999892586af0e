// y = LayerNorm(x . W^T + bias + residual) for the 256-wide projections that are followed by a LayerNorm - the attention
// out-projection (+ residual, norm1 / norm2) and linear2 of the feed-forward block (+ residual, norm2 / norm3),
// COTR/models/transformer.py:154-158, 196-201 - in ONE launch for the many-row regime (>= 24576 rows: 64-pair encoder passes, the
// decoder at 32 pairs x 1000 queries, the dense pass), gfx950.  MEASURED (profiles/r3_ab_gemm_plus_layernorm_one_launch.txt): 5-8 % faster
// than the large-tile GEMM + layernorm_kernel in isolation, no gain inside the forward - off by default (cotr_set_gemm_ln_min_rows).
//
// A workgroup owns 128 COMPLETE rows: tile 128 x 256 (4 wavefronts as 2 x 2, each 64 x 128 = 2 x 4 MFMA blocks of
// v_mfma_f32_32x32x2_f32), so the row statistics need no second launch and the [rows, 256] pre-norm tensor is never written to or
// read back from memory (2 x 33 MB per use at 32768 rows, and a launch).  The operand path is gemm_big.hip's (global -> LDS by
// LDS-DMA, unpadded tiles with XOR-swizzled 16-byte chunks, ring of three stages with counted vmcnt and one raw s_barrier per
// 32-deep K step): the same MFMA sequence per output element as the large-tile GEMM configurations, so the accumulators are
// bit-identical to theirs.  Epilogue: the accumulators go to a [128][260] fp32 tile in LDS (it takes the place of the operand
// stages), then every wavefront normalises 32 rows the way layernorm_kernel (pointwise.hip) does - a float4 per lane per row, bias and
// residual added in that order, two-pass statistics by wave shuffles (8 rows at a time) - so the output is bit-identical to
// gemm + layernorm_kernel (tests/test_ops_gpu.py).
#include "../common.h"

#define BK 32

namespace {

constexpr int LN_BM = 128, LN_BN = 256, LN_TN = 4;
constexpr int LN_STAGE = (LN_BM + LN_BN) * BK;          // floats per stage
constexpr int LN_XS = 260;                              // padded row of the pre-norm tile
constexpr size_t LN_SMEM = (size_t)3 * LN_STAGE * sizeof(float);
static_assert((size_t)LN_BM * LN_XS * sizeof(float) <= LN_SMEM, "the pre-norm tile fits in the operand stages");

__device__ __forceinline__ float ln_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__global__ __launch_bounds__(256) void gemm_ln_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                      const float* __restrict__ bias, const float* __restrict__ residual, int ldr,
                                                      const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                      float* __restrict__ Y, int M, int K, const float* __restrict__ zeros) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int m0 = blockIdx.x * LN_BM;
  const int KT = K / BK;

  // ---- LDS-DMA bookkeeping (gemm_big.hip): lane -> (row lane>>3 of the instruction's 8 rows, physical 16-B chunk lane&7) ----
  const int drow = lane >> 3, pch = lane & 7;
  const float* a_ptr[4];
  bool a_ok[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = wave * 32 + q * 8 + drow;
    const int lch = pch ^ ((row >> 1) & 7);
    const int m = m0 + row;
    a_ok[q] = m < M;
    a_ptr[q] = A + (size_t)(a_ok[q] ? m : 0) * lda + lch * 4;
  }
  const float* w_ptr[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int row = wave * 64 + q * 8 + drow;           // W row = output column
    const int lch = pch ^ ((row >> 1) & 7);
    w_ptr[q] = W + (size_t)row * K + lch * 4;
  }
  auto dma_tile = [&](int kt, int buf) {
    float* As = smem + buf * LN_STAGE;
    float* Ws = As + LN_BM * BK;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float* src = a_ok[q] ? a_ptr[q] + kt * BK : zeros;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(As + (wave * 32 + q * 8) * BK), 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_ptr[q] + kt * BK),
                                       (__attribute__((address_space(3))) void*)(Ws + (wave * 64 + q * 8) * BK), 16, 0, 0);
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hh = lane >> 5;
  const int sw = (l31 >> 1) & 7;
  f32x16 acc[2][LN_TN];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < LN_TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  dma_tile(0, 0);
  if (KT > 1) dma_tile(1, 1);
  int st = 0;
  for (int kt = 0; kt < KT; ++kt) {
    // in flight, oldest first: [tile kt] [tile kt+1]; one tile = 12 DMA instructions per wavefront
    if (kt + 1 < KT) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // everybody's share of tile kt is in LDS; stage (kt+2)%3 is free
    asm volatile("" ::: "memory");
    if (kt + 2 < KT) dma_tile(kt + 2, st == 0 ? 2 : st - 1);
    const float* As = smem + st * LN_STAGE + (wm * 64 + l31) * BK;
    const float* Ws = smem + st * LN_STAGE + LN_BM * BK + (wn * 128 + l31) * BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch = ((j * 2 + hh) ^ sw) * 4;
      f32x4 af[2], bf[LN_TN];
#pragma unroll
      for (int a = 0; a < 2; ++a) af[a] = *reinterpret_cast<const f32x4*>(As + a * 32 * BK + ch);
#pragma unroll
      for (int b = 0; b < LN_TN; ++b) bf[b] = *reinterpret_cast<const f32x4*>(Ws + b * 32 * BK + ch);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < LN_TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][e], bf[b][e], acc[a][b], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this step's fragment reads are retired before the next barrier
    st = st == 2 ? 0 : st + 1;
  }
  __syncthreads();                                      // every wavefront is done reading the operand stages

  // ---- the accumulators as a row-major [128][256] tile in LDS (D layout: rows (r&3) + 8 (r>>2) + 4 hh, column = lane & 31) ----
  float* Xs = smem;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < LN_TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Xs[(wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * LN_XS + wn * 128 + b * 32 + l31] = acc[a][b][r];
  __syncthreads();

  // ---- bias, residual, LayerNorm: layernorm_kernel's arithmetic, a row per pass of a wavefront --------------------------------
  const f32x4 bi = bias ? *reinterpret_cast<const f32x4*>(bias + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 ww = *reinterpret_cast<const f32x4*>(ln_w + lane * 4);
  const f32x4 bb = *reinterpret_cast<const f32x4*>(ln_b + lane * 4);
  // 8 rows per pass: their loads go out together and their 2 x 6 shuffle steps are 8 independent chains (one row at a time is a
  // dependent chain of ~12 cross-lane operations with nobody else on the SIMD to hide it: 15 us per workgroup)
  constexpr int RB = 8;
#pragma unroll 1
  for (int i0 = 0; i0 < 32; i0 += RB) {
    const int row0 = wave * 32 + i0;
    if (m0 + row0 >= M) break;                          // (wave-uniform)
    f32x4 v[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int m = m0 + row0 + i < M ? m0 + row0 + i : M - 1;     // rows past M: clamped loads, no store
      v[i] = *reinterpret_cast<const f32x4*>(Xs + (row0 + i) * LN_XS + lane * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] = v[i][e] + bi[e];     // the GEMM epilogue's order: + bias, then + residual
      if (residual) {
        const f32x4 rr = *reinterpret_cast<const f32x4*>(residual + (size_t)m * ldr + lane * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] += rr[e];
      }
    }
    float sum[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) sum[i] = v[i][0] + v[i][1] + v[i][2] + v[i][3];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int i = 0; i < RB; ++i) sum[i] += __shfl_xor(sum[i], off);
    f32x4 d[RB];
    float sq[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const float mean = sum[i] * (1.f / 256.f);
      d[i] = f32x4{v[i][0] - mean, v[i][1] - mean, v[i][2] - mean, v[i][3] - mean};
      sq[i] = d[i][0] * d[i][0] + d[i][1] * d[i][1] + d[i][2] * d[i][2] + d[i][3] * d[i][3];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int i = 0; i < RB; ++i) sq[i] += __shfl_xor(sq[i], off);
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const float var = sq[i] * (1.f / 256.f);
      const float rstd = 1.f / sqrtf(var + 1e-5f);
      f32x4 out;
#pragma unroll
      for (int e = 0; e < 4; ++e) out[e] = d[i][e] * rstd * ww[e] + bb[e];
      if (m0 + row0 + i < M) *reinterpret_cast<f32x4*>(Y + (size_t)(m0 + row0 + i) * 256 + lane * 4) = out;
    }
  }
}

}  // namespace

// y [M][256] = LayerNorm(x [M][K] . w [256][K]^T + bias + residual [M][ldr]) * ln_w + ln_b;  K a multiple of 32
int launch_gemm_ln(const float* x, int lda, const float* w, const float* bias, const float* residual, int ldr, const float* ln_w,
                   const float* ln_b, float* y, int M, int K, hipStream_t s) {
  if (M <= 0) return 0;
  if (K % BK != 0 || K <= 0 || lda % 4 != 0 || (residual && ldr % 4 != 0) || x == nullptr || w == nullptr || ln_w == nullptr ||
      ln_b == nullptr || y == nullptr)
    return -1;
  if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15) || ((uintptr_t)residual & 15) || ((uintptr_t)bias & 15)) return -1;
  const float* zeros = gemm_zero_buffer();
  if (zeros == nullptr) return -2;
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ln_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LN_SMEM) !=
        hipSuccess)
      return -2;
    attr_set.set();
  }
  hipLaunchKernelGGL(gemm_ln_kernel, dim3((M + LN_BM - 1) / LN_BM), dim3(256), LN_SMEM, s, x, lda, w, bias, residual, ldr, ln_w, ln_b, y, M,
                     K, zeros);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
