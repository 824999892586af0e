// conv2 (3x3) -> conv3 (1x1 expansion) of a LAYER2 bottleneck in ONE launch for many pairs (fp32 MFMA, gfx950) - conv23.hip's
// construction at 128 channels (torchvision Bottleneck.forward with the reference's FrozenBatchNorm2d, COTR/models/backbone.py:46-56):
//     t2 = relu(bn2(conv3x3(t1)))            128 -> 128, stride S (2 in block 0, 1 in blocks 1-3), padding 1, each half padded on its own
//     y  = relu(bn3(conv1x1(t2)) + identity) 128 -> 512
// At 32 pairs the two launches cost 163.5 us (0.75 of the peak) + 104 us (0.53: 8.6 GFLOP against 301 MB on 128 x 64 tiles that are all
// prologue and epilogue).  Here the 128 x 128 tile of t2 a workgroup has just computed is contracted with W3 on the spot.
//
// Work decomposition: workgroup = 4 wavefronts = 128 output pixels (TWO image rows of a pair: 2 x (32 + 32)) x all 128 conv2 channels;
// a wavefront owns 32 pixels (one half of one row).  Two workgroups per CU (64 KB of LDS), started out of step (conv23.hip).
//   phase 1  conv2 as an implicit GEMM, K = 1152 in 36 steps of 32 (step = tap, quarter of the channels), operands global -> LDS by
//            LDS-DMA (two 32 KB stages, one barrier per step).  The product is formed TRANSPOSED - A operand = W2 rows, B operand =
//            pixels, four accumulators per wavefront - so that a lane ends up holding, for ITS pixel, the channels 32 cb + (r&3) +
//            8(r>>2) + 4*half: what the A operand of the next product wants from that lane.  bn2 + ReLU in those registers.
//   phase 2  conv3: [32 pixels x 128] . W3[512 x 128]^T per wavefront, 32 output channels at a time.  W3 in sixteen 16 KB pieces through
//            two LDS slots laid over the dead operand stages, global -> registers -> LDS one piece ahead (plain loads: an LDS-DMA in
//            flight would turn every wait the compiler inserts for the identity registers into vmcnt(0) - expand.hip), one s_barrier
//            per piece; epilogue straight from the accumulator (lane = channel) with the next block's identity already requested.
// The k order of both products is the large-tile GEMM's: results are bit-identical to the two launches.
#include "common.h"

struct Conv23mParams {
  const float* t1;        // [B][32 S][64 S][128]
  const float* w2;        // [128][3][3][128]
  const float* s2;
  const float* b2;
  const float* w3;        // [512][128]
  const float* s3;
  const float* b3;
  const float* residual;  // [B][32][64][512]
  float* y;               // [B][32][64][512]
  const float* zeros;
  int tiles;              // B * 16 pairs of image rows
  int stagger;
};

#define C23M_BK 32
#define C23M_KT 36                               // 9 taps x 4 channel quarters
#define C23M_STAGE ((128 + 128) * C23M_BK)       // floats per operand stage
#define C23M_SLOT 4096                           // floats per W3 piece: [4 k tiles][32 output channels][32]
#define C23M_SMEM (2 * C23M_STAGE * 4)

struct C23mLane {
  int lane, l31, hh, sw, drow, pch;
};

__device__ __forceinline__ void c23m_barrier() {   // (not __syncthreads(): its fence waits for every outstanding store)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// W3 piece pc (output channels 32 pc .. +31) -> registers / registers -> LDS slot, 16-byte chunks XOR-swizzled with (row >> 1) & 7
__device__ __forceinline__ void c23m_w3_load(const Conv23mParams& p, const int t, const int pc, f32x4 (&wr)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + 256 * i;                 // 32 rows x 32 chunks
    wr[i] = *reinterpret_cast<const f32x4*>(p.w3 + (size_t)(pc * 32 + (idx >> 5)) * 128 + (idx & 31) * 4);
  }
}
__device__ __forceinline__ void c23m_w3_store(float* slot, const int t, const f32x4 (&wr)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + 256 * i;
    const int row = idx >> 5, c4 = idx & 31;
    *reinterpret_cast<f32x4*>(slot + (c4 >> 3) * 1024 + row * 32 + (((c4 & 7) ^ ((row >> 1) & 7)) << 2)) = wr[i];
  }
}
// identity of output block j: reg r = pixel (r&3) + 8(r>>2) + 4 half of the wavefront's 32, lane&31 = channel
__device__ __forceinline__ void c23m_res_load(const Conv23mParams& p, const float* __restrict__ rbase, const C23mLane& L, const int j,
                                              float (&res)[16], float& sc, float& bi) {
  sc = p.s3[j * 32 + L.l31];
  bi = p.b3[j * 32 + L.l31];
  const unsigned lo = L.hh * 2048u + L.l31 + j * 32;
#pragma unroll
  for (int r = 0; r < 16; ++r) res[r] = rbase[lo + ((r & 3) + 8 * (r >> 2)) * 512u];
}
__device__ __forceinline__ void c23m_block(const float* slot, float* __restrict__ ybase, const C23mLane& L, const int j, const f32x16 (&t2)[4],
                                           const float (&res)[16], const float sc, const float bi) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(slot + cb * 1024 + L.l31 * 32 + (((g * 2 + L.hh) ^ L.sw) << 2));
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(t2[cb][g * 4 + e], w[e], acc, 0, 0, 0);
    }
  const unsigned lo = L.hh * 2048u + L.l31 + j * 32;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = fmaf(acc[r], sc, bi) + res[r];
    v = (v < 0.f) ? 0.f : v;
    ybase[lo + ((r & 3) + 8 * (r >> 2)) * 512u] = v;
  }
}

template <int S>
__global__ __launch_bounds__(256, 2) void conv23m_kernel(const Conv23mParams p) {
  constexpr int HIN = 32 * S, WIN = 32 * S;       // one half of the input
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  C23mLane L;
  L.lane = t & 63;
  L.l31 = L.lane & 31;
  L.hh = L.lane >> 5;
  L.sw = (L.l31 >> 1) & 7;
  L.drow = L.lane >> 3;
  L.pch = L.lane & 7;
  // every XCD a contiguous range of image rows (the input rows a tile reads are then in ITS L2, fetched by its neighbours)
  const int bid = blockIdx.x;
  const int tile = (bid & 7) * (p.tiles >> 3) + (bid >> 3);
  const int b = tile >> 4;
  const int ho = (tile & 15) * 2 + (wave >> 1);   // this wavefront's output row; its pixels: half wave & 1, columns 0 .. 31
  const size_t m0 = (size_t)tile * 128;
  if (bid < 512 && p.stagger > 0) {               // conv23.hip: the first round's two workgroups of a CU out of lockstep
    const int n = (bid >> 8) * p.stagger;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
  }

  // ---- LDS-DMA bookkeeping of phase 1: lane -> (row lane>>3 of the instruction's 8 rows, physical 16-B chunk lane&7) ----
  const float* a_ptr[4];
  int wi0[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = wave * 32 + q * 8 + L.drow;   // tile-local pixel
    const int lch = L.pch ^ ((row >> 1) & 7);
    wi0[q] = (row & 31) * S - 1;                  // input column of tap (., 0)
    // input pixel of tap (0, 0): may lie in front of the tensor, only dereferenced in range
    a_ptr[q] = p.t1 + ((long)((b * HIN + ho * S - 1) * (2 * WIN) + (wave & 1) * WIN + wi0[q])) * 128 + lch * 4;
  }
  const float* w_ptr[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = wave * 32 + q * 8 + L.drow;
    const int lch = L.pch ^ ((row >> 1) & 7);
    w_ptr[q] = p.w2 + (size_t)row * 1152 + lch * 4;
  }
  auto dma_tile = [&](int kt, int buf) {
    float* As = smem + buf * C23M_STAGE;
    float* Ws = As + 128 * C23M_BK;
    const int tap = kt >> 2;
    const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
    const int tapoff = (ky * (2 * WIN) + kx) * 128 + (kt & 3) * 32;
    const bool row_ok = (unsigned)(ho * S - 1 + ky) < (unsigned)HIN;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = row_ok && (unsigned)(wi0[q] + kx) < (unsigned)WIN;
      const float* src = ok ? a_ptr[q] + tapoff : p.zeros;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(As + (wave * 32 + q * 8) * C23M_BK), 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_ptr[q] + kt * C23M_BK),
                                       (__attribute__((address_space(3))) void*)(Ws + (wave * 32 + q * 8) * C23M_BK), 16, 0, 0);
  };

  dma_tile(0, 0);
  dma_tile(1, 1);

  // ---- phase 1: t2^T[128 channels][32 pixels of this wavefront] ----
  f32x16 acc[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
  for (int kt = 0; kt < C23M_KT; ++kt) {
    LDS_DMA_WAIT_ALL();
    __syncthreads();
    if (kt >= 1 && kt + 1 < C23M_KT) dma_tile(kt + 1, (kt + 1) & 1);
    const float* As = smem + (kt & 1) * C23M_STAGE + (wave * 32 + L.l31) * C23M_BK;
    const float* Ws = smem + (kt & 1) * C23M_STAGE + 128 * C23M_BK + L.l31 * C23M_BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch = ((j * 2 + L.hh) ^ L.sw) * 4;
      const f32x4 af = *reinterpret_cast<const f32x4*>(As + ch);
      f32x4 w[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) w[cb] = *reinterpret_cast<const f32x4*>(Ws + cb * 32 * C23M_BK + ch);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[cb][e], af[e], acc[cb], 0, 0, 0);
    }
  }
  __syncthreads();                                // every wavefront is done reading the operand stages

  // ---- W3 pieces 0 (-> slot 0) and 1 (-> registers), the identity of block 0, bn2 + ReLU in the accumulator registers ----
  const float* __restrict__ rbase = p.residual + (m0 + wave * 32) * 512;   // wave-uniform
  float* __restrict__ ybase = p.y + (m0 + wave * 32) * 512;
  f32x4 wr[4];
  c23m_w3_load(p, t, 0, wr);
  float res_a[16], res_b[16], sc_a, bi_a, sc_b, bi_b;
  c23m_res_load(p, rbase, L, 0, res_a, sc_a, bi_a);
  f32x16 t2[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 sc = *reinterpret_cast<const f32x4*>(p.s2 + cb * 32 + g * 8 + L.hh * 4);
      const f32x4 bi = *reinterpret_cast<const f32x4*>(p.b2 + cb * 32 + g * 8 + L.hh * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = fmaf(acc[cb][g * 4 + e], sc[e], bi[e]);
        t2[cb][g * 4 + e] = (v < 0.f) ? 0.f : v;
      }
    }
  c23m_w3_store(smem, t, wr);
  c23m_w3_load(p, t, 1, wr);
  c23m_barrier();

  // ---- phase 2: 16 blocks of 32 output channels, two per iteration (the identity registers alternate) ----
  for (int j = 0; j < 16; j += 2) {
    // block j: piece j + 1 goes from the registers to the other slot (its readers passed the barrier that closed block j - 1)
    c23m_w3_store(smem + C23M_SLOT, t, wr);
    c23m_w3_load(p, t, j + 2 < 16 ? j + 2 : 15, wr);
    c23m_res_load(p, rbase, L, j + 1, res_b, sc_b, bi_b);
    c23m_block(smem, ybase, L, j, t2, res_a, sc_a, bi_a);
    c23m_barrier();
    // block j + 1
    if (j + 2 < 16) {
      c23m_w3_store(smem, t, wr);
      c23m_w3_load(p, t, j + 3 < 16 ? j + 3 : 15, wr);
      c23m_res_load(p, rbase, L, j + 2, res_a, sc_a, bi_a);
    }
    c23m_block(smem + C23M_SLOT, ybase, L, j + 1, t2, res_b, sc_b, bi_b);
    c23m_barrier();
  }
}

// t1 [B][32 S][64 S][128] -> y [B][32][64][512]; residual [B][32][64][512]; stride S = 1 or 2
int launch_conv23m(const float* t1, const float* w2, const float* s2, const float* b2, const float* w3, const float* s3, const float* b3,
                   const float* residual, float* y, int B, int stride, hipStream_t s) {
  if (B <= 0 || (stride != 1 && stride != 2) || !t1 || !w2 || !s2 || !b2 || !w3 || !s3 || !b3 || !residual || !y) return -1;
  if (((uintptr_t)t1 | (uintptr_t)w2 | (uintptr_t)w3 | (uintptr_t)s2 | (uintptr_t)b2) & 15) return -1;
  Conv23mParams p;
  p.t1 = t1; p.w2 = w2; p.s2 = s2; p.b2 = b2; p.w3 = w3; p.s3 = s3; p.b3 = b3; p.residual = residual; p.y = y;
  p.zeros = gemm_zero_buffer();
  if (p.zeros == nullptr) return -2;
  p.tiles = B * 16;
  p.stagger = cotr_num_cus() == 256 ? 5 : 0;
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv23m_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, C23M_SMEM) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv23m_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, C23M_SMEM) != hipSuccess)
      return -2;
    attr_set.set();
  }
  if (stride == 1) hipLaunchKernelGGL(conv23m_kernel<1>, dim3(p.tiles), dim3(256), C23M_SMEM, s, p);
  else hipLaunchKernelGGL(conv23m_kernel<2>, dim3(p.tiles), dim3(256), C23M_SMEM, s, p);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
