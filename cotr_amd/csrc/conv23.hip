// conv2 (3x3) -> conv3 (1x1 expansion) of a layer1 bottleneck in ONE launch for many pairs (fp32 MFMA, gfx950) - torchvision
// Bottleneck.forward with the reference's FrozenBatchNorm2d (COTR/models/backbone.py:46-56) on the NHWC "side-by-side" layout:
//     t2 = relu(bn2(conv3x3(t1)))            64 -> 64, padding 1, each 64-wide half padded on its own
//     y  = relu(bn3(conv1x1(t2)) + identity) 64 -> 256
//
// Why: at 32 pairs the two launches cost 175 + 150 us per block (profiles/r5_final_kernel_times_hip_events_b32_q1000.txt).  The
// expansion is bound by HBM - 8.6 GFLOP (55 us of fp32 MFMA) against 67 MB of t2 in, 268 MB of identity in and 268 MB of y out
// (4.0 TB/s) - while conv2 is bound by the matrix pipe and moves almost nothing.  In one kernel the 128 x 64 tile of t2 that a
// workgroup has just computed is contracted with W3 on the spot: t2 never goes to HBM and the identity / y traffic of one
// workgroup runs under the conv2 K loops of the two others that share its CU.
//
// Work decomposition: workgroup = 4 wavefronts = 128 consecutive pixels (ONE image row of a pair: 64 pixels of each half) x all
// 64 conv2 channels; a wavefront owns 32 pixels.  Three workgroups per CU (51.7 KB of LDS, <= 168 registers).
//   phase 1  conv2 as an implicit GEMM, K = 576 in 18 steps of 32 (step = tap, half of the channels), operands global -> LDS by
//            LDS-DMA (two stages, one barrier per step; the tile geometry and chunk swizzle of gemm_big.hip).  The product is
//            formed TRANSPOSED - A operand = W2 rows, B operand = pixels - so that a lane ends up holding, for ITS pixel, the
//            channels (r&3) + 8(r>>2) + 4*half: exactly what the A operand of the next product wants from that lane
//            (v_mfma_f32_32x32x2_f32: lane (row l&31, half l>>5) supplies k = 2*step + half; the k ORDER is free as long as
//            both operands use the same one).  bn2 + ReLU happen in those registers; t2 is never written anywhere.
//   phase 2  conv3: [32 pixels x 64] . W3[256 x 64]^T per wavefront, 64 output channels (two accumulators, alternating) at a
//            time; W3 arrives in four 16 KB pieces through a ring of three LDS slots laid over the dead operand stages (piece 0
//            is requested during the last K step, 1 and 2 right behind the loop, 3 into slot 0 when everybody is done with piece
//            0).  No request is ever waited for behind a store (vmcnt counts loads and stores in ONE in-order queue on gfx9, and
//            a store's acknowledgement takes microseconds): every W3 request is issued before the stores of the piece in hand.
//            Epilogue straight from the accumulators: a lane = one output channel, 32 lanes = 128 contiguous bytes of a pixel
//            row, for the identity read and the y write alike.
// Same arithmetic per output as the two-launch path, and the same ORDER of the partial sums as the large-tile GEMM configurations
// (26 / 27) those launches run from 5 pairs on: phase 2's k order 8g + 4 * half + e is theirs, so the results are their bits
// (tests/test_parity_gpu.py::test_backbone_fusions_are_bit_identical_to_the_launches_they_replace; op level from 16 pairs on:
// tests/test_ops_gpu.py::test_conv23_one_launch); against the small-tile configurations of fewer pairs: fp32 rounding.
#include "common.h"

struct Conv23Params {
  const float* t1;        // [B][64][128][64]   relu(bn1(conv1(x)))
  const float* w2;        // [64][3][3][64]
  const float* s2;
  const float* b2;
  const float* w3;        // [256][64]
  const float* s3;
  const float* b3;
  const float* residual;  // [B][64][128][256]  identity
  float* y;               // [B][64][128][256]
  const float* zeros;
  int tiles;              // B * 64 image rows
  int stagger;            // first-round workgroups of CU slot k = bid >> 8 start k * stagger * 8128 cycles late
};

#define C23_BK 32
#define C23_KT 18                              // 9 taps x 2 channel halves
#define C23_STAGE ((128 + 64) * C23_BK)        // floats per operand stage: 128 pixels + 64 conv2 channels, 32 deep
#define C23_SLOT (2 * 64 * C23_BK)             // floats per W3 piece: [2 k tiles][64 output channels][32]
#define C23_PAR (3 * C23_SLOT)                 // float offset of the FrozenBN parameters: s2 64, b2 64, s3 256, b3 256
#define C23_SMEM ((C23_PAR + 640) * 4)
static_assert(2 * C23_STAGE <= C23_PAR, "the operand stages lie under the W3 ring");
static_assert(C23_STAGE >= C23_SLOT, "W3 piece 0 fits the stage that is free during the last K step");

struct C23Lane {
  int lane, l31, hh, sw, drow, pch;
};

// W3 piece `pc` (output channels 64 pc .. +63) -> ring slot `slot`: 4 LDS-DMA instructions per wavefront (16 rows x 2 k tiles)
__device__ __forceinline__ void c23_dma_w3(const Conv23Params& p, float* smem, const C23Lane& L, const int wave, const int pc, const int slot) {
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int row = wave * 16 + q * 8 + L.drow;
      const int lch = L.pch ^ ((row >> 1) & 7);
      const float* src = p.w3 + (size_t)(pc * 64 + row) * 64 + kt * 32 + lch * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(smem + slot * C23_SLOT + kt * 2048 + (wave * 16 + q * 8) * C23_BK),
                                       16, 0, 0);
    }
}

// s_barrier alone: __syncthreads() is a fence too, and the fence waits for EVERY outstanding memory operation (vmcnt(0)) - the stores of
// the epilogues included; phase 2 states its waits itself
__device__ __forceinline__ void c23_barrier() {
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// block J of phase 2 = output channels 32 J .. +31 (half of W3 piece J/2, ring slot (J/2) % 3): 32 matrix instructions into one accumulator
template <int J>
__device__ __forceinline__ void c23_block_mfma(const float* smem, const C23Lane& L, const f32x16 (&t2)[2], f32x16& acc) {
  constexpr int SLOT = (J >> 1) % 3;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ch = ((g * 2 + L.hh) ^ L.sw) * 4;
      const f32x4 w = *reinterpret_cast<const f32x4*>(smem + SLOT * C23_SLOT + cb * 2048 + ((J & 1) * 32 + L.l31) * C23_BK + ch);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(t2[cb][g * 4 + e], w[e], acc, 0, 0, 0);
    }
}

// identity read / y write of one block: wave-uniform base (the wavefront's first pixel row) + a 32-bit lane offset + an immediate:
// reg r = pixel (r&3) + 8(r>>2) + 4 half of the wavefront's 32, lane&31 = channel
template <int J>
__device__ __forceinline__ void c23_res_load(const float* __restrict__ rbase, const C23Lane& L, float (&res)[16]) {
  const unsigned lo = L.hh * 1024u + L.l31;
#pragma unroll
  for (int r = 0; r < 16; ++r) res[r] = rbase[lo + (r >> 2) * 2048u + ((r & 3) * 256 + J * 32)];
}

template <int J>
__device__ __forceinline__ void c23_block_store(float* __restrict__ ybase, const float* pars, const C23Lane& L, const f32x16& acc,
                                                const float (&res)[16]) {
  const unsigned lo = L.hh * 1024u + L.l31;
  const float sc = pars[128 + J * 32 + L.l31], bi = pars[384 + J * 32 + L.l31];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = fmaf(acc[r], sc, bi) + res[r];
    v = (v < 0.f) ? 0.f : v;
    ybase[lo + (r >> 2) * 2048u + ((r & 3) * 256 + J * 32)] = v;
  }
}

// ABL (tools/micro/conv23_probe.hip only; the product is <0>): 1 = no identity reads, y written only by a branch never taken; 2 = no
// phase 2 at all (t2 summed into one store per lane); 4 = no barriers / waits in phase 2; 8 = no phase 1 (timing only, all of them)
template <int ABL, int J>
__device__ __forceinline__ void c23_block(const Conv23Params& p, float* smem, const float* pars, const C23Lane& L, const int wave,
                                          const float* __restrict__ rbase, float* __restrict__ ybase, const f32x16 (&t2)[2],
                                          float (&res_cur)[16], float (&res_nxt)[16]) {
  // the identity of the NEXT block is requested before this block's matrix instructions: one block of cover (+ whatever the two other
  // workgroups of the CU put in between) instead of a wait on HBM in every epilogue
  if constexpr (J + 1 < 8 && !(ABL & 1)) c23_res_load<J + 1>(rbase, L, res_nxt);
  f32x16 acc;
  c23_block_mfma<J>(smem, L, t2, acc);
  if constexpr (J == 1) {
    // slot 0 is read out: W3 piece 3 goes there once EVERY wavefront is here.  The same barrier tells everybody that pieces 1 and 2 have
    // landed: they are older than the identity of block 0, which this wavefront's epilogue 0 has waited for (one in-order queue)
    if constexpr (!(ABL & 4)) {
      asm volatile("s_waitcnt vmcnt(48)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      c23_barrier();
    }
    c23_dma_w3(p, smem, L, wave, 3, 0);
  }
  if constexpr (ABL & 1) {
    if (p.tiles < 0) c23_block_store<J>(ybase, pars, L, acc, res_cur);
    else asm volatile("" ::"v"(acc));
  } else {
    c23_block_store<J>(ybase, pars, L, acc, res_cur);
  }
  if constexpr (J == 5 && !(ABL & 4)) {
    // piece 3 (requested in block 1) is older than the identity of block 5, which the epilogue above has waited for: landed for this
    // wavefront; the barrier says so for the others'
    asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    c23_barrier();
  }
}

template <int ABL>
__global__ __launch_bounds__(256, 3) void conv23_kernel(const Conv23Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* pars = smem + C23_PAR;
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  C23Lane L;
  L.lane = t & 63;
  L.l31 = L.lane & 31;
  L.hh = L.lane >> 5;
  L.sw = (L.l31 >> 1) & 7;
  L.drow = L.lane >> 3;
  L.pch = L.lane & 7;
  // consecutive workgroups land on consecutive XCDs: give every XCD a contiguous range of image rows, so that the three t1 rows a
  // tile reads are in ITS L2 (fetched by its neighbours) instead of crossing the fabric once per XCD
  const int bid = blockIdx.x;
  const int tile = (bid & 7) * (p.tiles >> 3) + (bid >> 3);
  const int b = tile >> 6, ho = tile & 63;
  const size_t m0 = (size_t)tile * 128;
  // The first 768 workgroups start in the same microsecond, three per CU, and would walk through their phases in lockstep: every
  // matrix pipe contended in phase 1 while HBM idles, then every CU bursting identity reads / y writes at once.  Slots 1 and 2 of a CU
  // (the dispatcher fills all CUs once before it comes back for seconds: slot = bid >> 8) start a third / two thirds of a tile's time
  // late; the workgroup that is alone meanwhile runs that much faster (the CU's resources are shared, not partitioned)
  if (bid < 768 && p.stagger > 0) {
    const int n = (bid >> 8) * p.stagger;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
  }

  // ---- LDS-DMA bookkeeping of phase 1: lane -> (row lane>>3 of the instruction's 8 rows, physical 16-B chunk lane&7) ----
  const float* a_ptr[4];
  int wl[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = wave * 32 + q * 8 + L.drow;        // tile-local pixel: half row>>6, column row&63
    const int lch = L.pch ^ ((row >> 1) & 7);
    wl[q] = row & 63;
    // pixel (b, ho-1, half, wl-1) = tap (0, 0): may lie in front of the tensor, only dereferenced in range
    a_ptr[q] = p.t1 + ((long)((b * 64 + ho - 1) * 128 + (row >> 6) * 64 + wl[q] - 1)) * 64 + lch * 4;
  }
  const float* w_ptr[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = wave * 16 + q * 8 + L.drow;
    const int lch = L.pch ^ ((row >> 1) & 7);
    w_ptr[q] = p.w2 + (size_t)row * 576 + lch * 4;
  }
  auto dma_tile = [&](int kt, int buf) {
    float* As = smem + buf * C23_STAGE;
    float* Ws = As + 128 * C23_BK;
    const int tap = kt >> 1;
    const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
    const int tapoff = (ky * 128 + kx) * 64 + (kt & 1) * 32;
    const bool row_ok = (unsigned)(ho - 1 + ky) < 64u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = row_ok && (unsigned)(wl[q] - 1 + kx) < 64u;
      const float* src = ok ? a_ptr[q] + tapoff : p.zeros;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(As + (wave * 32 + q * 8) * C23_BK), 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_ptr[q] + kt * C23_BK),
                                       (__attribute__((address_space(3))) void*)(Ws + (wave * 16 + q * 8) * C23_BK), 16, 0, 0);
  };

  dma_tile(0, 0);
  dma_tile(1, 1);
  // FrozenBN parameters -> LDS (read after the K loop; its barriers order the writes)
  if (t < 160) {
    const float* src = t < 16 ? p.s2 + t * 4 : t < 32 ? p.b2 + (t - 16) * 4 : t < 96 ? p.s3 + (t - 32) * 4 : p.b3 + (t - 96) * 4;
    *reinterpret_cast<f32x4*>(pars + t * 4) = *reinterpret_cast<const f32x4*>(src);
  }

  // ---- phase 1: t2^T[64 channels][32 pixels of this wavefront] ----------------------------------------------------------------
  f32x16 acc[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
  for (int kt = (ABL & 8) ? C23_KT - 1 : 0; kt < C23_KT; ++kt) {
    LDS_DMA_WAIT_ALL();
    __syncthreads();
    if (kt >= 1 && kt + 1 < C23_KT) dma_tile(kt + 1, (kt + 1) & 1);
    if (kt == C23_KT - 1) c23_dma_w3(p, smem, L, wave, 0, 0);       // stage 0 is free from here on: W3 piece 0 -> slot 0
    const float* As = smem + (kt & 1) * C23_STAGE + (wave * 32 + L.l31) * C23_BK;
    const float* Ws = smem + (kt & 1) * C23_STAGE + 128 * C23_BK + L.l31 * C23_BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch = ((j * 2 + L.hh) ^ L.sw) * 4;
      const f32x4 af = *reinterpret_cast<const f32x4*>(As + ch);
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(Ws + ch);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(Ws + 32 * C23_BK + ch);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[e], af[e], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[e], af[e], acc[1], 0, 0, 0);
      }
    }
  }
  __syncthreads();                                      // every wavefront is done reading stage 1
  const float* __restrict__ rbase = p.residual + (m0 + wave * 32) * 256;   // wave-uniform
  float* __restrict__ ybase = p.y + (m0 + wave * 32) * 256;
  if constexpr (ABL & 2) {
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += acc[0][r] + acc[1][r];
    ybase[L.lane] = sum;
    return;
  }
  c23_dma_w3(p, smem, L, wave, 1, 1);
  c23_dma_w3(p, smem, L, wave, 2, 2);
  float res_a[16], res_b[16];
  if constexpr (!(ABL & 1)) c23_res_load<0>(rbase, L, res_a);
  else {
#pragma unroll
    for (int r = 0; r < 16; ++r) res_a[r] = res_b[r] = 0.f;
  }

  // bn2 + ReLU in the accumulator registers: reg r of block cb = channel 32 cb + (r&3) + 8(r>>2) + 4 half
  f32x16 t2[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 sc = *reinterpret_cast<const f32x4*>(pars + cb * 32 + g * 8 + L.hh * 4);
      const f32x4 bi = *reinterpret_cast<const f32x4*>(pars + 64 + cb * 32 + g * 8 + L.hh * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = fmaf(acc[cb][g * 4 + e], sc[e], bi[e]);
        t2[cb][g * 4 + e] = (v < 0.f) ? 0.f : v;
      }
    }

  // ---- phase 2: 8 blocks of 32 output channels ------------------------------------------------------------------------------------
  // W3 piece 0: everything requested after it may stay in flight: pieces 1, 2 (8 instructions) + the 16 identity loads of block 0
  if constexpr (!(ABL & 4)) {
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    c23_barrier();
  }
  c23_block<ABL, 0>(p, smem, pars, L, wave, rbase, ybase, t2, res_a, res_b);
  c23_block<ABL, 1>(p, smem, pars, L, wave, rbase, ybase, t2, res_b, res_a);
  c23_block<ABL, 2>(p, smem, pars, L, wave, rbase, ybase, t2, res_a, res_b);
  c23_block<ABL, 3>(p, smem, pars, L, wave, rbase, ybase, t2, res_b, res_a);
  c23_block<ABL, 4>(p, smem, pars, L, wave, rbase, ybase, t2, res_a, res_b);
  c23_block<ABL, 5>(p, smem, pars, L, wave, rbase, ybase, t2, res_b, res_a);
  c23_block<ABL, 6>(p, smem, pars, L, wave, rbase, ybase, t2, res_a, res_b);
  c23_block<ABL, 7>(p, smem, pars, L, wave, rbase, ybase, t2, res_b, res_a);
}

// t1 [B][64][128][64] -> y [B][64][128][256]; residual [B][64][128][256]
int launch_conv23(const float* t1, const float* w2, const float* s2, const float* b2, const float* w3, const float* s3, const float* b3,
                  const float* residual, float* y, int B, hipStream_t s) {
  if (B <= 0 || !t1 || !w2 || !s2 || !b2 || !w3 || !s3 || !b3 || !residual || !y) return -1;
  if (((uintptr_t)t1 | (uintptr_t)w2 | (uintptr_t)w3 | (uintptr_t)s2 | (uintptr_t)b2 | (uintptr_t)s3 | (uintptr_t)b3) & 15) return -1;
  Conv23Params p;
  p.t1 = t1; p.w2 = w2; p.s2 = s2; p.b2 = b2; p.w3 = w3; p.s3 = s3; p.b3 = b3; p.residual = residual; p.y = y;
  p.zeros = gemm_zero_buffer();
  if (p.zeros == nullptr) return -2;
  p.tiles = B * 64;
  p.stagger = cotr_num_cus() == 256 ? 5 : 0;   // x 3.6 us per CU slot: 295 -> 247 us at 32 pairs, flat from 2 to 6 (profiles/r5_conv23_probe.txt)
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv23_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, C23_SMEM) != hipSuccess)
      return -2;
    attr_set.set();
  }
  hipLaunchKernelGGL(conv23_kernel<0>, dim3(p.tiles), dim3(256), C23_SMEM, s, p);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
