// Layer1 block 0's two 1x1 convolutions over the pooled stem output for many pairs in ONE launch, fp32 MFMA, gfx950
// (torchvision Bottleneck.forward with the reference's FrozenBatchNorm2d, COTR/models/backbone.py:46-56):
//     identity = bn_d( x . Wd^T )          64 -> 256, no ReLU (the downsample branch)
//     t1       = relu( bn1( x . W1^T ) )   64 -> 64
// Both are bound by the bytes they write (335 MB at 32 pairs against 10.7 GFLOP), not by their matrix work, and the large-tile GEMM is at
// its worst on them: with K = 64 a 128 x 64 tile is two K steps of operand traffic around a 32 KB output tile - every workgroup is all
// prologue and epilogue, 10240 of them for the two launches, 126 + 39 us (2.7 / 3.4 TB/s;
// profiles/r5_final_kernel_times_hip_events_b32_q1000.txt).  Here (conv23.hip's second phase, fed from memory):
//   workgroup = 4 wavefronts = 128 rows; a wavefront keeps its 32 rows IN REGISTERS as the A operand of every product of the tile
//   (lane = row, register (kt, r) = channel 32 kt + (r&3) + 8(r>>2) + 4*half: 16-byte loads straight from the row - the k ORDER of
//   v_mfma_f32_32x32x2_f32 is free as long as both operands use the same one): x is read once for both convolutions.
//   The 320 output channels in five W pieces of 64 (16 KB) through two LDS slots, one piece ahead, global -> registers -> LDS: plain
//   loads on purpose - an LDS-DMA in flight turns every wait the compiler inserts for a register load into vmcnt(0) (a "flat" access is
//   pending as far as its scoreboard knows), which would drain the stores of every epilogue.
//   Per piece: 64 matrix instructions on two alternating accumulators, then both epilogues straight from the accumulators (lane =
//   channel: 128 contiguous bytes per row and half wavefront); one s_barrier per piece.  Four workgroups per CU (126 registers, 32 KB
//   of LDS), the first round's started out of step (conv23.hip).
// 122 us in the 32-pair forward against 165 for the two launches.  What it does NOT reach: its matrix work alone takes 100 us here
// (68 at the peak) and its stores alone 80 (58 at 6.9 TB/s), and the two overlap only partly (tools/micro/expand_probe.hip); the same
// kernel for layer2's K = 128 expansions measured 101 us against 103-106 for the tuned GEMM - not kept.
#include "common.h"

struct ExpandSeg {
  const float* w;         // [32 nblk][64]
  const float* scale;     // [32 nblk] FrozenBN scale, bias
  const float* bias;
  float* y;               // [M][ldy]
  int ldy, relu, nblk;
};
struct ExpandParams {
  const float* x;         // [M][64]
  ExpandSeg seg[2];       // seg[1].nblk == 0: one weight set
  int tiles, stagger;
};

#define EX_SLOT 4096                             // floats per W piece
#define EX_SMEM (2 * EX_SLOT * 4)

struct ExLane {
  int lane, l31, hh, sw;
};

__device__ __forceinline__ void ex_barrier() {   // (not __syncthreads(): its fence waits for every outstanding store)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// W piece pc (rows pc * PBR .. + PBR - 1 of the concatenated weight sets, PBR = 4096 / K) -> registers / registers -> LDS slot:
// [K / 32 k tiles][PBR rows][32 floats], the 16-byte chunks of a row XOR-swizzled with (row >> 1) & 7 (the fragment reads of 32
// consecutive rows then cover all banks)
template <int KC>
__device__ __forceinline__ void ex_w_load(const ExpandParams& p, const int t, const int pc, f32x4 (&wr)[4]) {
  constexpr int K = 32 * KC, C4 = K / 4, PBR = EX_SLOT / K;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + 256 * i;
    const int row = idx / C4, c4 = idx % C4;
    const int grow = pc * PBR + row;                      // row of the concatenation
    const int n0 = p.seg[0].nblk * 32;
    const float* src = grow < n0 ? p.seg[0].w + (size_t)grow * K : p.seg[1].w + (size_t)(grow - n0) * K;
    wr[i] = *reinterpret_cast<const f32x4*>(src + c4 * 4);
  }
}
template <int KC>
__device__ __forceinline__ void ex_w_store(float* slot, const int t, const f32x4 (&wr)[4]) {
  constexpr int K = 32 * KC, C4 = K / 4, PBR = EX_SLOT / K;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + 256 * i;
    const int row = idx / C4, c4 = idx % C4;
    const int kt = c4 >> 3, ch = (c4 & 7) ^ ((row >> 1) & 7);
    *reinterpret_cast<f32x4*>(slot + kt * (PBR * 32) + row * 32 + ch * 4) = wr[i];
  }
}

struct ExBlock {          // where output block j goes (wave-uniform)
  const float* scale;
  const float* bias;
  float* y;
  int ldy, relu;
};
__device__ __forceinline__ ExBlock ex_block(const ExpandParams& p, const int j, const size_t mw) {
  const int s = j < p.seg[0].nblk ? 0 : 1;
  const int jb = j - (s ? p.seg[0].nblk : 0);
  const ExpandSeg& g = p.seg[s];
  ExBlock b;
  b.scale = g.scale + jb * 32;
  b.bias = g.bias + jb * 32;
  b.ldy = g.ldy;
  b.relu = g.relu;
  b.y = g.y + mw * g.ldy + jb * 32;
  return b;
}
// ABL (tools/micro/expand_probe.hip only; the product is <0>): 1 = no stores, 2 = no matrix instructions, 4 = no barriers (timing only)
template <int ABL = 0>
__global__ __launch_bounds__(256, 4) void expand64_kernel(const ExpandParams p) {
  constexpr int KC = 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  ExLane L;
  L.lane = t & 63;
  L.l31 = L.lane & 31;
  L.hh = L.lane >> 5;
  L.sw = (L.l31 >> 1) & 7;
  const int bid = blockIdx.x;
  const size_t mw = (size_t)bid * 128 + wave * 32;
  const int NP = (p.seg[0].nblk + p.seg[1].nblk) / 2;
  constexpr int SLOTS = 4;                              // workgroups per CU
  if (bid < 256 * SLOTS && p.stagger > 0) {
    const int n = (bid >> 8) * p.stagger;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
  }
  f32x16 xa[KC];
  {
    const float* xr = p.x + (mw + L.l31) * 64 + L.hh * 4;
#pragma unroll
    for (int kt = 0; kt < KC; ++kt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + kt * 32 + g * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) xa[kt][g * 4 + e] = v[e];
      }
  }
  f32x4 wr[4];
  ex_w_load<KC>(p, t, 0, wr);
  ex_w_store<KC>(smem, t, wr);
  if (NP > 1) ex_w_load<KC>(p, t, 1, wr);
  float sc[2], bi[2];
  ExBlock blk[2] = {ex_block(p, 0, mw), ex_block(p, 1, mw)};
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    sc[c] = blk[c].scale[L.l31];
    bi[c] = blk[c].bias[L.l31];
  }
  ex_barrier();
  for (int pc = 0; pc < NP; ++pc) {
    if (pc + 1 < NP) ex_w_store<KC>(smem + ((pc + 1) & 1) * EX_SLOT, t, wr);
    if (pc + 2 < NP) ex_w_load<KC>(p, t, pc + 2, wr);
    const float* slot = smem + (pc & 1) * EX_SLOT;
    f32x16 acc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = (ABL & 2) ? xa[c][r] : 0.f;
    if constexpr (!(ABL & 2)) {
#pragma unroll
      for (int kt = 0; kt < KC; ++kt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ch = ((g * 2 + L.hh) ^ L.sw) << 2;
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(slot + kt * 2048 + L.l31 * 32 + ch);
          const f32x4 w1 = *reinterpret_cast<const f32x4*>(slot + kt * 2048 + (32 + L.l31) * 32 + ch);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[kt][g * 4 + e], w0[e], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[kt][g * 4 + e], w1[e], acc[1], 0, 0, 0);
          }
        }
    }
    // this piece's epilogues; the next piece's FrozenBN parameters are requested first
    const int jn = pc + 1 < NP ? 2 * (pc + 1) : 2 * pc;
    const ExBlock nxt[2] = {ex_block(p, jn, mw), ex_block(p, jn + 1, mw)};
    float sc_n[2], bi_n[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      sc_n[c] = nxt[c].scale[L.l31];
      bi_n[c] = nxt[c].bias[L.l31];
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (!(ABL & 1) || p.tiles < 0) {
        const unsigned lo = L.hh * 4u * blk[c].ldy + L.l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = fmaf(acc[c][r], sc[c], bi[c]);
          if (blk[c].relu) v = (v < 0.f) ? 0.f : v;
          blk[c].y[lo + ((r & 3) + 8 * (r >> 2)) * (unsigned)blk[c].ldy] = v;
        }
      } else {
        asm volatile("" ::"v"(acc[c]));
      }
      blk[c] = nxt[c];
      sc[c] = sc_n[c];
      bi[c] = bi_n[c];
    }
    if (!(ABL & 4)) ex_barrier();
  }
}

// x [M][64] (M a multiple of 128); two weight sets (n1 == 0: one), each: w [n][64], scale / bias [n], y [M][n], relu flag; n0, n1
// multiples of 64
int launch_expand(const float* x, int M, const float* w0, const float* s0, const float* b0, int relu0, float* y0, int n0, const float* w1,
                  const float* s1, const float* b1, int relu1, float* y1, int n1, hipStream_t s) {
  if (M <= 0 || M % 128 != 0 || !x || !w0 || !s0 || !b0 || !y0 || n0 <= 0 || n1 < 0 || n0 % 64 != 0 || n1 % 64 != 0) return -1;
  if (n1 > 0 && (!w1 || !s1 || !b1 || !y1)) return -1;
  if (((uintptr_t)x | (uintptr_t)w0 | (uintptr_t)w1) & 15) return -1;
  ExpandParams p;
  p.x = x;
  p.seg[0] = {w0, s0, b0, y0, n0, relu0, n0 / 32};
  p.seg[1] = {w1, s1, b1, y1, n1 > 0 ? n1 : 1, relu1, n1 / 32};
  p.tiles = M / 128;
  p.stagger = cotr_num_cus() == 256 ? 5 : 0;
  hipLaunchKernelGGL(expand64_kernel<0>, dim3(p.tiles), dim3(256), EX_SMEM, s, p);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
