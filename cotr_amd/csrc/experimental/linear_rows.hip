// MEASURED, LOST (round 5; research library only, knob linear_rows_min_rows, off): bit-identical to the tile kernels and 1-14 % SLOWER
// than them on every shape of the batched forward (profiles/r5_ab_linear_rows_lost.txt: 16384 x 768 x 256 76.7 vs 70.4 us, 16384 x 3072 x
// 256 314 vs 272, 65536 x 512 x 128 141 vs 124) - a plain projection has no second contraction to amortise the resident tile over, and
// its per-column-block epilogue (staging, residual loads, stores, and a full vmcnt drain in front of the stores because loads and stores
// complete out of order) costs more than the tile kernels' fixed cost.  The decomposition pays where a tile is reused by a SECOND
// contraction: ffn_rows.hip, att_rows.hip.
//
// K-short projections / 1x1 convolutions for MANY rows (the batched regime), fp32 MFMA, gfx950:
//     C[M,N] = epilogue( A[M,K] . W[N,K]^T ),  K = 128 or 256, N a multiple of 128
// the contract and epilogue of gemm.hip (GemmParams: FrozenBN scale / bias, q scale on the first columns, residual - plain or a
// row-periodic table -, ReLU).  These are the shapes where a tile kernel pays its fixed cost (operand prologue, epilogue, launch
// ramp: 2.4-9 us per 128-row tile, profiles/r4_tile_fixed_cost_vs_k_steps.txt) for a K loop of only 4-8 steps: the encoder's
// packed in-projection (768 x 256), the hoisted decoder K/V projection (3072 x 256), corr_embed (256 x 256) and the 1x1 expansions of
// layer2 / layer3 (512 x 128, 1024 x 256: torchvision Bottleneck.conv3 + bn3 + identity + ReLU).
//
// Same "rows" decomposition as ffn_rows.hip / att_rows.hip: a workgroup (4 wavefronts, one per SIMD) owns 64 rows and ALL N columns.
//   * A tile [64 x K] resident in LDS (k-tiled [K/32][64][32], 16-B chunks XOR-swizzled), loaded once.
//   * The N/32 column blocks are dealt to the wavefronts (wave w: blocks w, w+4, ...).  A block = K/32 pieces of W ([32 rows][32 k],
//     4 KB), wave-private: requested by LDS-DMA into the wavefront's own 4-slot ring three pieces ahead, read back by itself - no barrier
//     after the prologue, ordering by the wavefront's own counted vmcnt.  One piece feeds 32 matrix instructions (two 32-row blocks).
//   * Epilogue per block, wave-private: the 64 x 32 accumulators go through a 32-row LDS tile so that the residual is read and the output
//     written as float4 row segments (one instruction = 8 rows x 128 B) instead of 2 x 128 B per instruction straight from the D layout.
//     The residual rows of a block are requested when the block starts.
#include <utility>

#include "../common.h"

#define LR_BM 64
#define LR_NSLOT 4
#define LR_PIECE 1024
#define LR_SLD 36                // padded row of a wavefront's staging tile [32][36]

template <int KT>
struct LrLds {
  static constexpr int A = KT * LR_BM * 32;                // floats of the A tile
  static constexpr int RING = LR_NSLOT * LR_PIECE;
  static constexpr int STG = 32 * LR_SLD;
  static constexpr size_t BYTES = (size_t)(A + 4 * RING + 4 * STG) * sizeof(float);
};

__device__ __forceinline__ void lr_dma16(const float* src, float* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

struct LrLane {
  const float* At;
  float* ring;
  float* stg;
  int l31, hh, sw;
  int woff_e, woff_o;            // LDS-DMA element offsets of this lane inside a W piece (instruction q even / odd)
};

struct LrFrag {
  f32x4 a, x0, x1;               // W rows of the block; A rows of the two 32-row blocks
};
template <int SLOT, int KTI, int J>
__device__ __forceinline__ LrFrag lr_load_frag(const LrLane& L) {
  LrFrag f;
  const int ch = ((J * 2 + L.hh) ^ L.sw) * 4;
  f.a = *reinterpret_cast<const f32x4*>(L.ring + SLOT * LR_PIECE + L.l31 * 32 + ch);
  const float* Ak = L.At + KTI * (LR_BM * 32) + L.l31 * 32 + ch;
  f.x0 = *reinterpret_cast<const f32x4*>(Ak);
  f.x1 = *reinterpret_cast<const f32x4*>(Ak + 32 * 32);
  return f;
}
// piece kt of column block cb -> ring slot, instruction q of its four
template <int KT>
__device__ __forceinline__ void lr_dma_q(const GemmParams& p, const LrLane& L, const int cb, const int kt, const int slot, const int q) {
  lr_dma16(p.W + ((size_t)cb * 32 + q * 8) * (KT * 32) + kt * 32 + ((q & 1) ? L.woff_o : L.woff_e), L.ring + slot * LR_PIECE + q * 256);
}

// One 8-deep step of a column block (ffn_rows.hip's step order: first MFMA pair, the next step's fragment requests behind the
// counted wait where they open a new piece, then the other pairs with the request of piece +3 interleaved during step (piece, 1)).
// `more`: the wavefront has another block after this one (requests past this block's pieces go to it).
template <int KT, int S_>
__device__ __forceinline__ void lr_step(const GemmParams& p, const LrLane& L, const int cb, const bool more, LrFrag& cur, f32x16 (&acc)[2]) {
  constexpr int NS = KT * 4;
  constexpr int sub = S_ >> 2, j = S_ & 3;
  LrFrag nxt = cur;
  constexpr int q_piece = sub + LR_NSLOT - 1;              // requested during step (sub, 1)
  const bool refill = j == 1 && (q_piece < KT || more);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.x0[e], cur.a[e], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.x1[e], cur.a[e], acc[1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (refill) {
      lr_dma_q<KT>(p, L, q_piece < KT ? cb : cb + 4, q_piece % KT, q_piece % LR_NSLOT, e);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (e == 0) {
      if constexpr (S_ + 1 < NS) {
        if constexpr (j == 3 && sub >= 2) {
          // The next piece (sub+1 >= 3) must have landed; pieces 1 and 2 of a block are known to (the prologue / the end of the
          // previous block waited for them).  Behind it, pieces sub+2 and sub+3 may be in flight - where they exist: inside this
          // block always, past it only if the wavefront has another block.
          if constexpr (sub + 3 < KT) {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          } else {
            if (more) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (KT - 2 - sub == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
        }
        nxt = lr_load_frag<((S_ + 1) >> 2) % LR_NSLOT, ((S_ + 1) >> 2), ((S_ + 1) & 3)>(L);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  cur = nxt;
}
template <int KT, int... S_>
__device__ __forceinline__ void lr_steps(const GemmParams& p, const LrLane& L, const int cb, const bool more, LrFrag& cur, f32x16 (&acc)[2],
                                         std::integer_sequence<int, S_...>) {
  (lr_step<KT, S_>(p, L, cb, more, cur, acc), ...);
}

template <int KT>
__global__ __launch_bounds__(256, 1) void linear_rows_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using LDS = LrLds<KT>;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int m0 = blockIdx.x * LR_BM;
  constexpr int K = KT * 32;

  const int drow = lane >> 3, pch = lane & 7;
  const int lch_e = pch ^ (drow >> 1), lch_o = lch_e ^ 4;
  LrLane L;
  L.At = smem; L.ring = smem + LDS::A + wave * LDS::RING; L.stg = smem + LDS::A + 4 * LDS::RING + wave * LDS::STG;
  L.l31 = l31; L.hh = hh; L.sw = (l31 >> 1) & 7;
  L.woff_e = drow * K + lch_e * 4; L.woff_o = drow * K + lch_o * 4;

  const int nblk = p.N / 128;                              // column blocks of this wavefront: cb = wave + 4 i
  // ---- prologue: the A tile (KT x 8 DMA instructions, 2 KT per wavefront), the first three pieces ----
#pragma unroll
  for (int i = 0; i < 2 * KT; ++i) {
    const int idx = wave * 2 * KT + i;
    const int kt = idx >> 3, rg = idx & 7;
    const int row = rg * 8 + drow;
    const int lch = pch ^ ((row >> 1) & 7);
    const float* src = (m0 + row < p.M) ? p.A + (size_t)(m0 + row) * p.lda + kt * 32 + lch * 4 : p.zeros;
    lr_dma16(src, smem + kt * (LR_BM * 32) + rg * 256);
  }
#pragma unroll
  for (int s = 0; s < LR_NSLOT - 1; ++s)                   // (KT >= 4: all three in the first block)
#pragma unroll
    for (int q = 0; q < 4; ++q) lr_dma_q<KT>(p, L, wave, s, s, q);
  LDS_DMA_WAIT_ALL();
  __syncthreads();

  // epilogue lanes: a staged row = 32 floats = 8 lanes of float4; one wave instruction = 8 rows
  const int er = lane >> 3, ec = (lane & 7) * 4;
  LrFrag cur = lr_load_frag<0, 0, 0>(L);
  for (int i = 0; i < nblk; ++i) {
    const int cb = wave + 4 * i;
    const int n0 = cb * 32;
    const bool more = i + 1 < nblk;
    // the residual rows of this block and its per-column constants, requested now
    f32x4 res[2][4];
    if (p.residual != nullptr) {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int m = m0 + mb * 32 + it * 8 + er;
          const int mr = m < p.M ? (p.res_row_mod > 0 ? fastmod(m, p.fd_resrow) : m) : 0;
          res[mb][it] = *reinterpret_cast<const f32x4*>(p.residual + (size_t)mr * p.ldr + n0 + ec);
        }
    }
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f}, cs;
    if (p.scale != nullptr) sc = *reinterpret_cast<const f32x4*>(p.scale + n0 + ec);
    if (p.bias != nullptr) bi = *reinterpret_cast<const f32x4*>(p.bias + n0 + ec);
#pragma unroll
    for (int e = 0; e < 4; ++e) cs[e] = (n0 + ec + e < p.colscale_n) ? p.colscale : 1.f;

    f32x16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    lr_steps<KT>(p, L, cb, more, cur, acc, std::make_integer_sequence<int, KT * 4>{});
    // The next block's first three pieces (requested during this block's last three) are waited for BEFORE this block's stores go
    // out: loads and stores complete out of order with respect to each other, so a counted wait behind the stores would sit there
    // until they are acknowledged - the next wait is then 2.5 pieces (~2 us) away.  Its first fragments are requested now.
    if (more) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      cur = lr_load_frag<0, 0, 0>(L);
    }
    // ---- epilogue of the block: 32 rows at a time through the wavefront's staging tile ----
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) L.stg[((r & 3) + 8 * (r >> 2) + 4 * hh) * LR_SLD + l31] = acc[mb][r];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + er;
        const int m = m0 + mb * 32 + row;
        f32x4 v = *reinterpret_cast<const f32x4*>(L.stg + row * LR_SLD + ec);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = v[e];
          x = p.scale ? fmaf(x, sc[e], bi[e]) : x + bi[e];
          x *= cs[e];
          if (p.residual) x += res[mb][it][e];
          if (p.relu) x = (x < 0.f) ? 0.f : x;
          v[e] = x;
        }
        if (m < p.M) *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + n0 + ec) = v;
      }
    }
  }
}

// which launches take this kernel (launch_gemm, gemm.hip): dense products - or 1x1 stride-1 convolutions, which are dense products of
// the pixel rows - with K = 128 / 256, N a multiple of 128, no x + pos prologue, from knob linear_rows_min_rows rows on, where the
// 64-row tiles fill the last round of the 256 CUs to at least 3/4
bool linear_rows_applies(int mode, const GemmParams& p) {
  if (mode == GEMM_CONV) {
    if (!(p.ksize == 1 && p.stride == 1 && p.pad == 0 && p.lda == p.Cin)) return false;
  } else if (mode != GEMM_DENSE) {
    return false;
  }
  if (p.K != 128 && p.K != 256) return false;
  if (p.N % 128 != 0 || p.N < 256 || p.A2 != nullptr || p.M < knob(KN_LINEAR_ROWS_MIN_ROWS)) return false;
  if (p.lda % 4 || p.ldc % 4 || ((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15) || ((uintptr_t)p.C & 15)) return false;
  if (p.residual && (p.ldr % 4 || ((uintptr_t)p.residual & 15))) return false;
  if ((p.scale && ((uintptr_t)p.scale & 15)) || (p.bias && ((uintptr_t)p.bias & 15))) return false;
  const long tiles = (p.M + LR_BM - 1) / LR_BM, rounds = (tiles + 255) / 256;
  return tiles * 4 >= rounds * 256 * 3;
}

template <int KT>
static int launch_lr_t(const GemmParams& p0, hipStream_t s) {
  GemmParams p = p0;
  if (p.zeros == nullptr) p.zeros = gemm_zero_buffer();
  if (p.zeros == nullptr) return -2;
  p.fd_resrow = fastdiv_make(p.res_row_mod > 0 ? p.res_row_mod : 1);
  if (p.M > fastdiv_max_n(p.fd_resrow)) return -1;
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(linear_rows_kernel<KT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)LrLds<KT>::BYTES) != hipSuccess)
      return -2;
    attr_set.set();
  }
  hipLaunchKernelGGL(linear_rows_kernel<KT>, dim3((p.M + LR_BM - 1) / LR_BM), dim3(256), LrLds<KT>::BYTES, s, p);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_linear_rows(const GemmParams& p, hipStream_t s) {
  if (p.M <= 0) return 0;
  if (p.N % 128 != 0 || p.A2 != nullptr) return -1;
  if (p.K == 256) return launch_lr_t<8>(p, s);
  if (p.K == 128) return launch_lr_t<4>(p, s);
  return -1;
}
