// Transformer feed-forward block for MANY rows (the batched regime: >= ~8192 token / query rows per launch), fp32 MFMA, gfx950:
//     Y = [post-LN] LayerNorm( X + relu(X . W1^T + b1) . W2^T + b2 )
// = linear1 + ReLU + linear2 + residual + norm of COTR/models/transformer.py:156-158 (encoder) / :199-201 (decoder) [+ decoder.norm,
// transformer.py:110-111] in ONE launch.  It replaces linear1 (K = 256 large tiles), linear2, and the LayerNorm launch of the unfused
// path; the [rows x 1024] hidden tensor never exists, and the per-tile fixed cost of a K = 256 GEMM (a quarter of such a launch,
// profiles/r4_tile_fixed_cost_vs_k_steps.txt) is paid once per 64 rows x 2.1 MFLOP instead of once per 128 x 128 x 256 tile.
//
// Decomposition ("rows"): a workgroup (4 wavefronts, one per SIMD) owns 64 rows and the whole 256-wide output.
//   * X tile [64 x 256] resident in LDS (64 KB, k-tiled [8][64][32] with the 16-B chunk XOR swizzle of gemm_big.hip).
//   * The 1024 hidden units are dealt to the wavefronts in blocks of 32 (wave w: blocks w, w+4, ...).  For a block the wavefront
//       phase 1  H^T[32 hid x 64 rows] = W1_blk . X^T           (A = W1 rows, B = X rows; 2 x 128 MFMAs of 32x32x2)
//                + b1, ReLU in registers
//       phase 2  Yp[64 rows x 256] += H . W2[:, blk]^T          (A = H: the D layout of H^T IS the A-operand layout, lane = row,
//                                                                 register r = hidden unit (r&3) + 8(r>>2) + 4*half - the trick of
//                                                                 attention.hip's P registers; B = W2 rows; 2 x 128 MFMAs)
//     so the hidden activations never leave the registers, and a wavefront's partial Yp (256 accumulator registers) is summed with
//     the other three once per tile.
//   * Weights are WAVE-PRIVATE streams: every weight element is used by exactly one wavefront of the workgroup, so each wavefront
//     requests its own 4 KB pieces ([32 rows][32 k]: 8 of W1, then 8 of W2 per hidden block) by LDS-DMA into its own ring of slots
//     and reads them back itself: NO barrier in the main loop (gemm_wp.hip's scheme); ordering is the wavefront's own counted vmcnt.
//     2 MB of weights per 64 rows = 32 FLOP per byte pulled into the CU (the 128 x 128 tile's ratio), L2-resident for every XCD.
//   * Epilogue: the four partial Yp go through LDS (32 rows at a time, fixed order w0 + w1 + w2 + w3), + b2 + residual (= X), LayerNorm
//     with layernorm_kernel's arithmetic (pointwise.hip), optional second LayerNorm, coalesced float4 row stores.
#include <type_traits>
#include <utility>

#include "common.h"

#define FR_D 256
#define FR_H 1024
#define FR_BM 64
#define FR_NSLOT 4               // ring slots per wavefront (4 KB each, NSLOT divides the 16 pieces of a hidden block: slot = piece % NSLOT is static);
                                 // NSLOT - 2 pieces stay in flight behind the one being consumed
#define FR_PIECE 1024            // floats per piece: 32 rows x 32 k
#define FR_LDT 288               // row of the epilogue tiles: 256 + 32, consecutive rows 32 banks apart

struct FfnRowsParams {
  const float* X;       // [M][256] (also the residual)
  const float* W1;      // [1024][256]
  const float* b1;      // [1024]
  const float* W2;      // [256][1024]
  const float* b2;      // [256]
  const float* ln_w;    // [256]
  const float* ln_b;
  const float* post_w;  // nullptr, or a second LayerNorm applied to the result
  const float* post_b;
  float* Y;             // [M][256]
  const float* zeros;   // >= 16 B of zeros (rows past M)
  int M;
  unsigned long long* dbg;   // ablation instantiations only (tools/micro/ffn_rows_probe.hip): [workgroups][12][2] (shader cycles, 100 MHz wall clock)
};

constexpr int FR_XS = 8 * FR_BM * 32;                    // floats of the X tile
constexpr int FR_RING = FR_NSLOT * FR_PIECE;             // floats of one wavefront's ring
constexpr int FR_MAIN = FR_XS + 4 * FR_RING + FR_H;       // floats of the main loop's LDS: X tile, rings, b1
constexpr int FR_EPI = 4 * 32 * FR_LDT;                   // floats of the epilogue tiles (they alias the main loop's)
constexpr int FR_PAR = FR_MAIN > FR_EPI ? FR_MAIN : FR_EPI;   // behind both: b2, ln_w, ln_b, post_w, post_b (staged once: the epilogue reads them from LDS)
constexpr size_t kFfnRowsSmem = (size_t)(FR_PAR + 5 * FR_D) * sizeof(float);
static_assert(kFfnRowsSmem <= 160 * 1024, "LDS");

// sum over the 8 lanes of an aligned group (all of them get it): two quad permutes and a half-row mirror, DPP - no LDS crossbar
__device__ __forceinline__ float fr_group8_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror: lane i <-> 7 - i
  return v;
}

__device__ __forceinline__ void fr_dma16(const float* src, float* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// fragment set of one 8-deep step: phase 1 uses a (W1) + x0, x1 (X rows of the two 32-row blocks), phase 2 uses a (W2) only
struct FrFrag {
  f32x4 a, x0, x1;
};
// per-lane constants of a wavefront
struct FrLane {
  const float* Xs;     // X tile in LDS
  float* ring;         // this wavefront's ring
  const float* b1s;    // b1 in LDS
  int l31, hh, sw;     // fragment row, k half, chunk swizzle
  int w1_e, w1_o, w2_e, w2_o;   // LDS-DMA element offsets of this lane inside a W1 / W2 piece (instruction q even / odd)
};

// piece `sub` (0-7: W1 k-tile sub; 8-15: W2 column block sub-8) of hidden block hb -> ring slot; instruction q of its four
__device__ __forceinline__ void fr_dma_piece_q(const FfnRowsParams& p, const FrLane& L, int hb, int sub, int slot, int q) {
  float* S = L.ring + slot * FR_PIECE + q * 256;
  if (sub < 8) fr_dma16(p.W1 + ((size_t)hb * 32 * FR_D + sub * 32 + q * 8 * FR_D) + ((q & 1) ? L.w1_o : L.w1_e), S);
  else fr_dma16(p.W2 + ((size_t)(sub - 8) * 32 * FR_H + hb * 32 + q * 8 * FR_H) + ((q & 1) ? L.w2_o : L.w2_e), S);
}

// step s of a hidden block: piece s>>2 (ring slot (s>>2) % NSLOT: 16 pieces per block, NSLOT divides 16), 8-deep slice j = s&3
template <int S_>
__device__ __forceinline__ FrFrag fr_load_frag(const FrLane& L) {
  FrFrag f;
  constexpr int sub = S_ >> 2, j = S_ & 3;
  const int ch = ((j * 2 + L.hh) ^ L.sw) * 4;
  f.a = *reinterpret_cast<const f32x4*>(L.ring + (sub % FR_NSLOT) * FR_PIECE + L.l31 * 32 + ch);
  if constexpr (sub < 8) {
    const float* Xk = L.Xs + sub * (FR_BM * 32) + L.l31 * 32 + ch;
    f.x0 = *reinterpret_cast<const f32x4*>(Xk);
    f.x1 = *reinterpret_cast<const f32x4*>(Xk + 32 * 32);
  } else {
    f.x0 = f.a;
    f.x1 = f.a;
  }
  return f;
}

// One step (8 MFMAs) of a hidden block.  Order inside a step: the first MFMA pair (the only place the wavefront waits for the step's
// fragments - requested a step ago), then the requests for step s+1's fragments, then the other three pairs: whatever conservative
// LDS wait hipcc puts behind an LDS-DMA instruction or the vmcnt asm lands where nothing is pending.  Piece sub+NSLOT-1 is requested
// during step (sub, 1), one DMA instruction behind each MFMA pair, into the slot of piece sub-1 (free: its last fragment reads fed
// MFMAs that are already issued, and LDS reads return in order).  LAST: the wavefront's final block - nothing to request past its
// pieces, nothing to prefetch past its last step.
// ABL (ablation bits, 0 in the product; tools/micro/ffn_rows_probe.hip instantiates the others): 1 = no refill DMA, 2 = no vmcnt waits,
// 4 = phase stamps of wavefront 0
template <bool LAST, int S_, int ABL>
__device__ __forceinline__ void fr_step(const FfnRowsParams& p, const FrLane& L, const int hb, FrFrag& cur, f32x16 (&hacc)[2],
                                        f32x16 (&yacc)[2][8]) {
  constexpr int sub = S_ >> 2, j = S_ & 3;
  constexpr int q_piece = sub + FR_NSLOT - 1;              // the piece requested during step (sub, 1)
  constexpr bool refill = j == 1 && !(LAST && q_piece > 15) && !(ABL & 1);
  FrFrag nxt = cur;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if constexpr (sub < 8) {
      hacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[e], cur.x0[e], hacc[0], 0, 0, 0);
      hacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[e], cur.x1[e], hacc[1], 0, 0, 0);
    } else {
      yacc[0][sub - 8] = __builtin_amdgcn_mfma_f32_32x32x2f32(hacc[0][j * 4 + e], cur.a[e], yacc[0][sub - 8], 0, 0, 0);
      yacc[1][sub - 8] = __builtin_amdgcn_mfma_f32_32x32x2f32(hacc[1][j * 4 + e], cur.a[e], yacc[1][sub - 8], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (refill) {
      fr_dma_piece_q(p, L, q_piece < 16 ? hb : hb + 4, q_piece & 15, q_piece % FR_NSLOT, e);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (e == 0) {
      if constexpr (S_ + 1 < 64 || !LAST) {
        if constexpr (j == 3 && !(ABL & 2)) {
          // the next piece must have landed; behind it at most NSLOT-2 younger pieces may stay in flight (requested so far: up to
          // piece sub+NSLOT-1 of this block, where that exists)
          constexpr int younger = (LAST && sub + FR_NSLOT - 1 > 15) ? 15 - (sub + 1) : FR_NSLOT - 2;
          static_assert(younger >= 0 && younger <= 2, "counted wait");
          if constexpr (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          else if constexpr (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        nxt = fr_load_frag<(S_ + 1) & 63>(L);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if constexpr (S_ == 31) {
    // ---- + b1, ReLU: register r of lane (row l31, half hh) is hidden unit hb*32 + (r&3) + 8(r>>2) + 4hh ----
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(L.b1s + hb * 32 + 8 * g + 4 * L.hh);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = hacc[a][g * 4 + e] + bv[e];
          hacc[a][g * 4 + e] = (v < 0.f) ? 0.f : v;        // NaN passes through like torch.relu
        }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  cur = nxt;
}

template <bool LAST, int ABL, int... S_>
__device__ __forceinline__ void fr_steps(const FfnRowsParams& p, const FrLane& L, const int hb, FrFrag& cur, f32x16 (&hacc)[2],
                                         f32x16 (&yacc)[2][8], std::integer_sequence<int, S_...>) {
  (fr_step<LAST, S_, ABL>(p, L, hb, cur, hacc, yacc), ...);
}

// one hidden block (32 units) of this wavefront: 64 steps
template <bool LAST, int ABL>
__device__ __forceinline__ void fr_hidden_block(const FfnRowsParams& p, const FrLane& L, const int hb, FrFrag& cur, f32x16 (&yacc)[2][8]) {
  f32x16 hacc[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) hacc[a][r] = 0.f;
  fr_steps<LAST, ABL>(p, L, hb, cur, hacc, yacc, std::make_integer_sequence<int, 64>{});
}

template <int ABL>
__global__ __launch_bounds__(256, 1) void ffn_rows_kernel(const FfnRowsParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                                        // [8 k-tiles][64 rows][32]
  float* b1s = smem + FR_XS + 4 * FR_RING;                 // [1024]

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int sw = (l31 >> 1) & 7;                           // chunk swizzle of this lane's fragment rows (row bases are multiples of 32)
  const int m0 = blockIdx.x * FR_BM;
  float* ring = smem + FR_XS + wave * FR_RING;
#define FR_STAMP(slot)                                                                   \
  do {                                                                                   \
    if constexpr ((ABL & 4) != 0) {                                                      \
      if (t == 0) {                                                                      \
        p.dbg[((size_t)blockIdx.x * 12 + (slot)) * 2] = __builtin_readcyclecounter();    \
        p.dbg[((size_t)blockIdx.x * 12 + (slot)) * 2 + 1] = wall_clock64();              \
      }                                                                                  \
    }                                                                                    \
  } while (0)
  FR_STAMP(0);

  // ---- LDS-DMA bookkeeping: one instruction = 8 rows x 128 B; lane -> (row lane>>3, physical 16-B chunk lane&7) fetches the
  // logical chunk pch ^ ((row>>1)&7) of its row, so the tile lands swizzled without padding (gemm_big.hip).  For the four
  // instructions q of a 32-row piece, row = 8q + drow: (row>>1)&7 = (drow>>1) ^ 4(q&1) - two lane offsets serve all four ----
  const int drow = lane >> 3, pch = lane & 7;
  const int lch_e = pch ^ (drow >> 1), lch_o = lch_e ^ 4;
  FrLane L;
  L.Xs = Xs; L.ring = ring; L.b1s = b1s; L.l31 = l31; L.hh = hh; L.sw = sw;
  L.w1_e = drow * FR_D + lch_e * 4; L.w1_o = drow * FR_D + lch_o * 4;
  L.w2_e = drow * FR_H + lch_e * 4; L.w2_o = drow * FR_H + lch_o * 4;

  // ---- prologue: X tile (64 DMA instructions, 16 per wavefront), b1 -> LDS, the first NSLOT-1 weight pieces ----
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int idx = wave * 16 + i;
    const int kt = idx >> 3, rg = idx & 7;
    const int row = rg * 8 + drow;
    const int lch = pch ^ ((row >> 1) & 7);
    const float* src = (m0 + row < p.M) ? p.X + (size_t)(m0 + row) * FR_D + kt * 32 + lch * 4 : p.zeros;
    fr_dma16(src, Xs + kt * (FR_BM * 32) + rg * 256);
  }
  *reinterpret_cast<f32x4*>(b1s + t * 4) = *reinterpret_cast<const f32x4*>(p.b1 + t * 4);
  float* pars = smem + FR_PAR;
  pars[t] = p.b2[t];
  pars[FR_D + t] = p.ln_w[t];
  pars[2 * FR_D + t] = p.ln_b[t];
  if (p.post_w != nullptr) {
    pars[3 * FR_D + t] = p.post_w[t];
    pars[4 * FR_D + t] = p.post_b[t];
  }
#pragma unroll
  for (int s = 0; s < FR_NSLOT - 1; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) fr_dma_piece_q(p, L, wave, s, s, q);
  LDS_DMA_WAIT_ALL();
  __syncthreads();                                         // the X tile, b1 and everybody's first pieces are in LDS

  f32x16 yacc[2][8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) yacc[a][b][r] = 0.f;

  FrFrag cur = fr_load_frag<0>(L);
  FR_STAMP(1);
  for (int i = 0; i < FR_H / 128 - 1; ++i) {
    fr_hidden_block<false, ABL>(p, L, wave + 4 * i, cur, yacc);
    FR_STAMP(2 + i);
  }
  fr_hidden_block<true, ABL>(p, L, wave + 4 * (FR_H / 128 - 1), cur, yacc);
  FR_STAMP(9);

  // ---- epilogue: sum the four partial Yp, + b2 + residual, LayerNorm; 32 rows per pass ----
  // The partials go through LDS as [wavefront][32 rows][FR_LDT]; then EIGHT LANES share a row (lane -> row lane>>3 of the wavefront's 8,
  // columns 32c + 4(lane&7) .. +3, c = 0..7): all 32 rows of a pass are normalised side by side, the row statistics are an in-lane sum
  // of 32 values + three DPP steps inside the 8-lane group - no LDS-crossbar shuffle and no row after row dependent chain (the first
  // version, one wavefront per row with 6-step shuffles: 26.5 k of a tile's 320 k cycles, profiles/r5_ffn_rows_probe.txt).
  // FR_LDT = 288: consecutive rows are 32 banks apart, so the 16-lane groups of a ds_read_b128 (two rows' halves) never collide.
  __syncthreads();                                         // nobody reads the X tile or a ring any more; no DMA is in flight
  FR_STAMP(10);
  float* Pw = smem + wave * (32 * FR_LDT);
  const int erow = wave * 8 + (lane >> 3), eseg = (lane & 7) * 4;
  // the residual rows of both passes are requested now: their latency (the X tile left L2 a tile ago) hides under the first pass's LDS traffic
  f32x4 xr[2][8];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int m = m0 + mb * 32 + erow;
    const float* xres = p.X + (size_t)(m < p.M ? m : 0) * FR_D + eseg;
#pragma unroll
    for (int c = 0; c < 8; ++c) xr[mb][c] = *reinterpret_cast<const f32x4*>(xres + c * 32);
  }
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) Pw[((r & 3) + 8 * (r >> 2) + 4 * hh) * FR_LDT + nb * 32 + l31] = yacc[mb][nb][r];
    __syncthreads();
    const int m = m0 + mb * 32 + erow;
    f32x4 x[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) x[c] = xr[mb][c];
    float s1 = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float* src = smem + erow * FR_LDT + c * 32 + eseg;
      f32x4 v = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
      for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(src + w * (32 * FR_LDT));   // fixed order w0 + w1 + w2 + w3
      v += *reinterpret_cast<const f32x4*>(pars + c * 32 + eseg);
      x[c] += v;
      s1 += (x[c][0] + x[c][1]) + (x[c][2] + x[c][3]);
    }
    const float mean = fr_group8_sum(s1) * (1.f / 256.f);
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x[c][e] -= mean;
        s2 = fmaf(x[c][e], x[c][e], s2);
      }
    }
    const float rstd = 1.f / sqrtf(fr_group8_sum(s2) * (1.f / 256.f) + 1e-5f);
    s1 = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const f32x4 lw = *reinterpret_cast<const f32x4*>(pars + FR_D + c * 32 + eseg);
      const f32x4 lb = *reinterpret_cast<const f32x4*>(pars + 2 * FR_D + c * 32 + eseg);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x[c][e] = x[c][e] * rstd * lw[e] + lb[e];
        s1 += x[c][e];
      }
    }
    if (p.post_w != nullptr) {                             // a second LayerNorm of the result (decoder.norm, transformer.py:110-111)
      const float m2 = fr_group8_sum(s1) * (1.f / 256.f);
      s2 = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          x[c][e] -= m2;
          s2 = fmaf(x[c][e], x[c][e], s2);
        }
      const float r2 = 1.f / sqrtf(fr_group8_sum(s2) * (1.f / 256.f) + 1e-5f);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const f32x4 pw = *reinterpret_cast<const f32x4*>(pars + 3 * FR_D + c * 32 + eseg);
        const f32x4 pb = *reinterpret_cast<const f32x4*>(pars + 4 * FR_D + c * 32 + eseg);
#pragma unroll
        for (int e = 0; e < 4; ++e) x[c][e] = x[c][e] * r2 * pw[e] + pb[e];
      }
    }
    if (m < p.M) {
      float* dst = p.Y + (size_t)m * FR_D + eseg;
#pragma unroll
      for (int c = 0; c < 8; ++c) *reinterpret_cast<f32x4*>(dst + c * 32) = x[c];
    }
    if (mb == 0) __syncthreads();                          // the tiles are overwritten by the second pass
  }
  FR_STAMP(11);
#undef FR_STAMP
}

// Y = [LN_post] LN(X + W2 relu(W1 X + b1) + b2), one launch; Y must not alias X (a workgroup reads its residual rows after other
// workgroups may have written theirs - different rows, but keep the contract simple: distinct buffers)
int launch_ffn_rows(const float* X, const float* W1, const float* b1, const float* W2, const float* b2, const float* ln_w,
                    const float* ln_b, const float* post_w, const float* post_b, float* Y, int M, hipStream_t s) {
  if (M <= 0) return 0;
  if (!X || !W1 || !b1 || !W2 || !b2 || !ln_w || !ln_b || !Y || X == Y) return -1;
  if (((uintptr_t)X | (uintptr_t)Y | (uintptr_t)W1 | (uintptr_t)W2 | (uintptr_t)b1 | (uintptr_t)b2) & 15) return -1;
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_rows_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kFfnRowsSmem) != hipSuccess)
      return -2;
    attr_set.set();
  }
  FfnRowsParams p;
  p.X = X; p.W1 = W1; p.b1 = b1; p.W2 = W2; p.b2 = b2; p.ln_w = ln_w; p.ln_b = ln_b; p.post_w = post_w; p.post_b = post_b;
  p.Y = Y; p.zeros = gemm_zero_buffer(); p.M = M; p.dbg = nullptr;
  if (p.zeros == nullptr) return -2;
  hipLaunchKernelGGL(ffn_rows_kernel<0>, dim3((M + FR_BM - 1) / FR_BM), dim3(256), kFfnRowsSmem, s, p);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
