// RESEARCH (knob split_f16, libcotr_hip_exp.so only): the K = 256 projections on packed split-f16 operands with the A tile RESIDENT IN
// REGISTERS - configuration 50.
//
// Why (docs/LABNOTES.md 3e): configurations 46 - 49 are not bound by the matrix pipe but by what a CU ingests: a 128 x 128 tile pulls 32 KB per
// 32-deep K step (40 GB/s per CU, 0.79 us per step and tile slot at 4096^3) against 0.32 us of MFMAs, and at K = 256 a tile lives for 8
// steps only, so the 2.4-9 us it costs around its K loop (docs/LABNOTES.md 3d) dominate.  Here a workgroup (4 wavefronts, ONE per SIMD: 512
// registers each) owns 128 rows for ALL its column tiles: each wavefront loads the hi / lo halves of its 32 rows x 256 k ONCE, already
// separated, into 128 registers (it computes 32 rows x 128 columns of a tile: with 64 x 64 per wavefront the 256 A registers + two accumulator
// sets spill) (the MFMA A operands of every step come from there: no LDS read, no v_perm for A), and only W streams -
// 16 KB per step instead of 32 - through an 8-stage LDS ring of 128 KB that runs 7 steps ahead and straight across column tiles.
// One s_barrier per step; one drained wait (vmcnt(0)) per column tile, because the epilogue's stores share the counter with the ring's loads.
// Epilogue straight from the accumulators (bias, column scale, fp32 / packed residual, ReLU, fp32 or packed output).
// MEASURED (profiles/r4_split_f16_gemm_configs.txt): correct, same bits as 46 - 49 - and 138 us against 74 (configuration 47) at 32000 x 1024 x 256:
// one wavefront per SIMD leaves nobody to issue while it waits for its fragment reads; as hipcc schedules the loop a K step takes 2.2 us.
// Never picked by h2_config; kept so that the measurement can be repeated.
#include "../common.h"
#include "gemm_h2.h"

namespace {
constexpr int RK = 256, RKT = RK / 32, RSTG = 8;
constexpr size_t H2R_SMEM = (size_t)RSTG * 128 * 32 * sizeof(float);

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_h2r_kernel(const GemmParams p, const int nsplit) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int rt = blockIdx.x / nsplit, part = blockIdx.x - rt * nsplit;
  const int m0 = rt * 128;
  const int ntiles = p.N / 128;
  const int nt0 = (int)((long)ntiles * part / nsplit), nt1 = (int)((long)ntiles * (part + 1) / nsplit);
  const int F = (nt1 - nt0) * RKT;                       // W tiles this workgroup consumes, in order
  if (F <= 0) return;

  // ---- the W ring: one DMA instruction = 8 rows x 128 B into 1 KB of LDS; chunks XOR-swizzled with (row >> 1) & 7 (gemm_big.hip) ----
  const int drow = lane >> 3, pch = lane & 7;
  int w_off[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = wave * 32 + q * 8 + drow;
    w_off[q] = row * RK + (pch ^ ((row >> 1) & 7)) * 4;
  }
  auto dma_w = [&](int f) {
    const int nt = nt0 + f / RKT, kt = f - (f / RKT) * RKT;
    const float* wbase = p.W + (size_t)nt * 128 * RK + kt * 32;
    float* Ws = smem + (f & (RSTG - 1)) * (128 * 32);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase + w_off[q]),
                                       (__attribute__((address_space(3))) void*)(Ws + (wave * 32 + q * 8) * 32), 16, 0, 0);
  };
#pragma unroll
  for (int f = 0; f < RSTG - 1; ++f)
    if (f < F) dma_w(f);

  // ---- A: this wavefront's 64 rows x 256 k, hi / lo separated, in registers for the whole kernel ----
  f16x8 ah[RKT][2], al[RKT][2];
  {
    const int m = m0 + wave * 32 + l31;
    const float* arow = p.A + (size_t)(m < p.M ? m : 0) * p.lda + hh * 4;
#pragma unroll
    for (int kt = 0; kt < RKT; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        // the k slots of gemm_big_body's fragments: chunks j = 2 s2, 2 s2 + 1 of the 32-deep step, 4 dwords at (2 j + hh) * 4
        u32x4 d0 = *reinterpret_cast<const u32x4*>(arow + kt * 32 + (4 * s2) * 4);
        u32x4 d1 = *reinterpret_cast<const u32x4*>(arow + kt * 32 + (4 * s2 + 2) * 4);
        if (m >= p.M) d0 = d1 = u32x4{0u, 0u, 0u, 0u};
        h2_unzip(d0, d1, ah[kt][s2], al[kt][s2]);
      }
  }

  const int sw = (l31 >> 1) & 7;
  for (int nt = nt0; nt < nt1; ++nt) {
    f32x16 acc[4], accx[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = accx[b][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < RKT; ++kt) {
      const int f = (nt - nt0) * RKT + kt;
      // tile f has landed: in flight behind it are at most RSTG - 2 tiles (4 DMA instructions per wavefront each); after a column
      // tile's epilogue the stores share the counter, so its first step - and the tail of the ring - drain it
      if (kt == 0 || f + RSTG - 2 >= F) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (f + RSTG - 1 < F) dma_w(f + RSTG - 1);         // into the stage whose readers all passed this barrier
      const float* Ws = smem + (f & (RSTG - 1)) * (128 * 32) + l31 * 32;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int ch0 = ((s2 * 4 + hh) ^ sw) * 4, ch1 = ((s2 * 4 + 2 + hh) ^ sw) * 4;
        f16x8 bh[4], bl[4];
#pragma unroll
        for (int b = 0; b < 4; ++b)
          h2_unzip(*reinterpret_cast<const u32x4*>(Ws + b * 32 * 32 + ch0), *reinterpret_cast<const u32x4*>(Ws + b * 32 * 32 + ch1), bh[b], bl[b]);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kt][s2], bh[b], acc[b], 0, 0, 0);
          accx[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kt][s2], bl[b], accx[b], 0, 0, 0);
          accx[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kt][s2], bh[b], accx[b], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this step's fragment reads are retired before the next barrier
    }
    // ---- epilogue of column tile nt, straight from the accumulators ----
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int col = nt * 128 + b * 32 + l31;
        const float biv = (p.bias ? p.bias[col] : 0.f), csv = col < p.colscale_n ? p.colscale : 1.f;
        float* crow = p.C + (size_t)(m0 + wave * 32 + 4 * hh) * p.ldc + col;
        const float* rrow = p.residual ? p.residual + (size_t)(m0 + wave * 32 + 4 * hh) * p.ldr + col : nullptr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          if (m0 + wave * 32 + 4 * hh + dr >= p.M) continue;
          float x = (fmaf(accx[b][r], 0x1p-11f, acc[b][r]) + biv) * csv;
          if (rrow) {
            const float rv = rrow[(size_t)dr * p.ldr];
            x += (p.h2_flags & 2) ? h2_unpack(__float_as_uint(rv)) : rv;
          }
          if (p.relu) x = (x < 0.f) ? 0.f : x;
          crow[(size_t)dr * p.ldc] = (p.h2_flags & 1) ? __uint_as_float(h2_pack_chk(x, p.h2_ovf)) : x;
        }
      }
  }
}
}  // namespace

// dense only, K = 256, N a multiple of 128, packed A / W (16-byte aligned rows: lda % 4 == 0)
int launch_gemm_h2r(const GemmParams& p0, hipStream_t s) {
  GemmParams p = p0;
  if (p.K != RK || p.N % 128 != 0 || p.M <= 0 || p.A2 != nullptr || p.lda % 4 != 0 || p.scale != nullptr || p.res_row_mod > 0) return -1;
  if (!gemm_fill_divs(p, GEMM_DENSE, 128, 128)) return -1;
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h2r_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)H2R_SMEM) != hipSuccess)
      return -2;
    attr_set.set();
  }
  const int row_tiles = (p.M + 127) / 128, ntiles = p.N / 128;
  int nsplit = 1;
  while (row_tiles * nsplit < 224 && nsplit * 2 <= ntiles) nsplit *= 2;   // every CU a workgroup before column tiles are shared out
  p.h2_ovf = h2_overflow_flag();
  if (p.h2_ovf == nullptr) return -2;
  hipLaunchKernelGGL(gemm_h2r_kernel, dim3(row_tiles * nsplit), dim3(256), H2R_SMEM, s, p, nsplit);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
