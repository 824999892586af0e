// RESEARCH LIBRARY COPY of csrc/gemm_big.hip (libcotr_hip_exp.so only): the product file with the research / dead-end paths that used to sit
// behind #ifdef COTR_EXPERIMENTAL in it resolved IN (tools/unifdef_exp.py -D).  The product never compiles this file.
// Large-tile fp32 MFMA GEMM / implicit-GEMM convolution for the batched regime (many image pairs per call: zoom-in
// engine at B=32, dense pass, config 3), gfx950.  Same contract and fused epilogue as gemm.hip (GemmParams).
//
// Workgroup = 4 wavefronts (2 x 2), tile 128 x (64*TN); a wavefront owns 64 x (32*TN) = 2 x TN MFMA blocks of
// v_mfma_f32_32x32x2_f32, so every ds_read_b128 fragment feeds 4*TN (A) / 8 (B) MFMAs.
//
// Operand path: global -> LDS directly (global_load_lds_dwordx4), two stages, ONE barrier per 32-deep K step; the
// transfer of step t+1 is in flight under the 16*2*TN MFMAs per wavefront of step t.  One DMA instruction moves
// 8 rows x 128 B into 1 KB of contiguous LDS (the destination of LDS-DMA is lane-linear), so the tile is stored
// unpadded [row][32 floats] and the 16-byte chunks of a row are XOR-swizzled with (row >> 1) & 7 by choosing which
// global chunk each lane fetches: the fragment reads (32 consecutive rows, same logical chunk) then cover all 64
// banks once per 16 lanes - conflict-free without padding.
//
// Epilogue: accumulators go through LDS (wave-private, 32 rows at a time) so that residual reads and output writes
// are float4 and a wavefront writes 4 x 256 B (TN = 2) contiguous row segments per instruction instead of 64
// scattered dwords; the K-short, residual-carrying 1x1 expansions of the ResNet bottlenecks are bandwidth-bound on
// exactly this traffic.
#include <type_traits>

#include "../common.h"

#define BK 32
#include "gemm_h2.h"

// NST = LDS stages: 2 = every barrier drains the LDS-DMA queue (vmcnt(0)); 3 = ring with TWO tiles in flight: the wait before
// the barrier of step t is a counted vmcnt that covers tile t only, tile t+1 stays in flight across the barrier (raw s_barrier,
// no fence: __syncthreads() would drain the queue) and tile t+2 is requested right after it.
// DIRECT (experiment, configurations 44 / 45): the epilogue stores straight from the accumulators - a wave store = rows r and r + 4 of a
// 32-column block = two full 128-B lines - instead of staging 32 rows at a time through LDS for float4 row stores: no LDS round trip, no
// barrier between the K loop and the epilogue, 4x the store instructions.  Same arithmetic per element: bit-identical.
// X selects an experimental form (libcotr_hip_exp.so only): 0 = the product kernel, 1 = DIRECT, 2 = H2 (configurations 46 / 47,
// experimental/gemm_h2.h): both operands arrive as PACKED SPLIT-f16 dwords (cotr_op_split_h2) and every fp32 product becomes three
// v_mfma_f32_32x32x16_f16 - research, NOT bit-identical to the fp32 path.
// NW = wavefronts (4: the 128-row tile of every product configuration; 8: a 256-row tile, experimental configuration 51: a wavefront still
// fetches 32 A rows, half as many W rows, and the tile moves 25 % fewer bytes per flop)
template <int TN, int MODE, int NST, int X = 0, int NW = 4>
__device__ __forceinline__ void gemm_big_body(const GemmParams& p, const int bid) {
  constexpr bool DIRECT = X == 1;
  [[maybe_unused]] constexpr bool H2 = X == 2;
  constexpr int BM = 32 * NW, BN = 64 * TN;
  constexpr int STAGE = (BM + BN) * BK;   // floats per stage
  constexpr int QW = BN / (8 * NW);       // W-tile DMA instructions per wavefront
  constexpr int EP = 32 * TN + 4;         // padded row of the epilogue staging tile
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int m0, n0;
  if (!gemm_tile_coords(p, BM, BN, bid, m0, n0)) return;
  const int KT = p.K / BK;

  // ---- LDS-DMA bookkeeping: lane -> (row lane>>3 of the instruction's 8 rows, physical 16-B chunk lane&7) ----
  const int drow = lane >> 3, pch = lane & 7;
  const float* a_ptr[4];
  bool a_ok[4];
  int c_hi0[4], c_wi0[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = wave * 32 + q * 8 + drow;          // tile-local A row
    const int lch = pch ^ ((row >> 1) & 7);            // logical chunk this lane fetches
    const int m = m0 + row;
    a_ok[q] = m < p.M;
    const int mm = a_ok[q] ? m : 0;
    if constexpr (MODE == GEMM_DENSE) {
      a_ptr[q] = p.A + (size_t)mm * p.lda + lch * 4;
      c_hi0[q] = c_wi0[q] = 0;
    } else {
      int b, ho, side, wl;
      conv_row_decompose(p, mm, b, ho, side, wl);
      c_hi0[q] = ho * p.stride - p.pad;
      c_wi0[q] = wl * p.stride - p.pad;
      a_ptr[q] = p.A + (long)(((b * p.Hin + c_hi0[q]) * (2 * p.Win) + side * p.Win + c_wi0[q]) * p.Cin) + lch * 4;   // pixel (b, hi0, side, wi0): may lie in front of the tensor, only dereferenced in range
    }
  }
  const float* w_ptr[QW];
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    const int row = wave * (BN / NW) + q * 8 + drow;   // tile-local W row
    const int lch = pch ^ ((row >> 1) & 7);
    w_ptr[q] = p.W + (size_t)(n0 + row) * p.K + lch * 4;
  }

  auto dma_tile = [&](int kt, int buf) {
    float* As = smem + buf * STAGE;
    float* Ws = As + BM * BK;
    if constexpr (MODE == GEMM_DENSE) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* src = a_ok[q] ? a_ptr[q] + kt * BK : p.zeros;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(As + (wave * 32 + q * 8) * BK), 16, 0, 0);
      }
    } else {
      int ky, kx, c0;
      conv_ktile_decompose(p, kt, ky, kx, c0);
      const int tapoff = (ky * (2 * p.Win) + kx) * p.Cin + c0;   // wave-uniform element offset of this tap / channel tile
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int hi = c_hi0[q] + ky, wi = c_wi0[q] + kx;
        const bool ok = a_ok[q] && hi >= 0 && hi < p.Hin && wi >= 0 && wi < p.Win;
        const float* src = ok ? a_ptr[q] + tapoff : p.zeros;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(As + (wave * 32 + q * 8) * BK), 16, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < QW; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_ptr[q] + kt * BK),
                                       (__attribute__((address_space(3))) void*)(Ws + (wave * (BN / NW) + q * 8) * BK), 16, 0, 0);
  };

  // ---- main loop -----------------------------------------------------------------------------------------------
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hh = lane >> 5;
  const int sw = (l31 >> 1) & 7;                        // swizzle of this lane's fragment rows (row bases are multiples of 32)
  f32x16 acc[2][TN];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  f32x16 accx[H2 ? 2 : 1][H2 ? TN : 1];                 // H2: the cross terms (hi x lo + lo x hi), scaled by 2^11
  if constexpr (H2) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) accx[a][b][r] = 0.f;
  }

  // both stages are requested up front; the residual tile (epilogue layout: row it*RPI + er, 4 columns at ec) follows
  // them so that its HBM latency is paid under the first barrier / the MFMAs instead of after them
  dma_tile(0, 0);
  if (KT > 1) dma_tile(1, 1);
  constexpr int C4 = 8 * TN;                            // float4 per staged epilogue row
  constexpr int RPI = 64 / C4;                          // rows per wave instruction
  constexpr int NIT = 32 / RPI;
  const int er = lane / C4, ec = (lane % C4) * 4;
  const int ncol = n0 + wn * 32 * TN + ec;
  f32x4 res[2][NIT];
  if (!DIRECT && p.residual) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int m = m0 + wm * 64 + a * 32 + it * RPI + er;
        const int mr = m < p.M ? (p.res_row_mod > 0 ? fastmod(m, p.fd_resrow) : m) : 0;
        res[a][it] = *reinterpret_cast<const f32x4*>(p.residual + (size_t)mr * p.ldr + ncol);
      }
  }
  int st = 0;                                           // stage of tile kt
  for (int kt = 0; kt < KT; ++kt) {
    if constexpr (NST == 2) {
      LDS_DMA_WAIT_ALL();                               // this wavefront's share of tile kt (and kt+1) has landed ...
      __syncthreads();                                  // ... and so has everybody else's; stage (kt+1)&1 is free
      if (kt >= 1 && kt + 1 < KT) dma_tile(kt + 1, (kt + 1) & 1);
      st = kt & 1;
    } else {
      // in flight, oldest first: [tile kt] [tile kt+1]; one tile = 4 + QW DMA instructions per wavefront.  (In the first
      // steps the residual prefetch sits behind tile 1 and is waited for too - conservative, not wrong.)
      if (kt + 1 < KT) {
        static_assert(QW == 4 || QW == 2, "counted wait: one tile = 4 + QW DMA instructions per wavefront");
        if constexpr (QW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                     // everybody's share of tile kt is in LDS; stage (kt+2)%3 is free
      asm volatile("" ::: "memory");                    // no LDS access of this step may be scheduled above the barrier
      if (kt + 2 < KT) dma_tile(kt + 2, st == 0 ? 2 : st - 1);
    }
    const float* As = smem + st * STAGE + (wm * 64 + l31) * BK;
    const float* Ws = smem + st * STAGE + BM * BK + (wn * 32 * TN + l31) * BK;
    if constexpr (H2) {
      h2_kstep<TN>(As, Ws, hh, sw, acc, accx);
    } else
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch = ((j * 2 + hh) ^ sw) * 4;
      f32x4 af[2], bf[TN];
#pragma unroll
      for (int a = 0; a < 2; ++a) af[a] = *reinterpret_cast<const f32x4*>(As + a * 32 * BK + ch);
#pragma unroll
      for (int b = 0; b < TN; ++b) bf[b] = *reinterpret_cast<const f32x4*>(Ws + b * 32 * BK + ch);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][e], bf[b][e], acc[a][b], 0, 0, 0);
    }
    if constexpr (NST == 3) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this step's fragment reads are retired before the next barrier
      st = st == 2 ? 0 : st + 1;
    }
  }
  if constexpr (H2) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = fmaf(accx[a][b][r], 0x1p-11f, acc[a][b][r]);
  }
  if constexpr (DIRECT) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int col = n0 + wn * 32 * TN + b * 32 + l31;
        const float scv = p.scale ? p.scale[col] : 1.f, biv = p.bias ? p.bias[col] : 0.f;
        const float csv = col < p.colscale_n ? p.colscale : 1.f;
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          rv[r] = 0.f;
          if (p.residual && m < p.M) rv[r] = p.residual[(size_t)(p.res_row_mod > 0 ? fastmod(m, p.fd_resrow) : m) * p.ldr + col];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          float x = acc[a][b][r];
          x = p.scale ? fmaf(x, scv, biv) : x + biv;
          x *= csv;
          if (p.residual) x += rv[r];
          if (p.relu) x = (x < 0.f) ? 0.f : x;
          if (m < p.M) p.C[(size_t)m * p.ldc + col] = x;
        }
      }
    return;
  }
  __syncthreads();                                      // every wavefront is done reading the operand stages

  // ---- epilogue through LDS: 32 rows x (32*TN) columns of this wavefront at a time -------------------------------
  float* Es = smem + wave * 32 * EP;
  f32x4 sc, bi, cs;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sc[e] = p.scale ? p.scale[ncol + e] : 1.f;
    bi[e] = p.bias ? p.bias[ncol + e] : 0.f;
    cs[e] = (ncol + e < p.colscale_n) ? p.colscale : 1.f;
  }
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) Es[((r & 3) + 8 * (r >> 2) + 4 * hh) * EP + b * 32 + l31] = acc[a][b][r];
    const int mb = m0 + wm * 64 + a * 32;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = it * RPI + er;
      const int m = mb + row;
      f32x4 v = *reinterpret_cast<const f32x4*>(Es + row * EP + ec);
      if (m < p.M) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = v[e];
          x = p.scale ? fmaf(x, sc[e], bi[e]) : x + bi[e];
          x *= cs[e];
          if constexpr (H2) {
            const float rv = res[a][it][e];   // (a scalar copy: __builtin_bit_cast on the vector element reads element 0 with this hipcc)
            if (p.residual) x += (p.h2_flags & 2) ? h2_unpack(__float_as_uint(rv)) : rv;
          } else
          if (p.residual) x += res[a][it][e];
          if (p.relu) x = (x < 0.f) ? 0.f : x;
          v[e] = x;
          if constexpr (H2) if (p.h2_flags & 1) v[e] = __uint_as_float(h2_pack_chk(x, p.h2_ovf));
        }
        *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + ncol) = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Wave-specialised form of the same tile (round 3): 8 wavefronts - 4 that ONLY issue the LDS-DMA of the operand tiles, 4 that ONLY
// read fragments and issue MFMAs.  Why: one LDS-DMA instruction costs the issuing wavefront 60-185 cycles of its instruction
// stream (MI355X_MICROARCH.md price table); a K step of the 128 x 128 tile needs 8 of them per wavefront plus the tile's address
// arithmetic (the convolution's tap / bounds logic), ~1000 cycles against the 4096 cycles of its 64 MFMAs - with the 256-workgroup
// grids of layer3 / the 256-wide projections at 32 pairs there is ONE wavefront per SIMD and nothing covers that (matrix pipe 66-69 %
// busy by counter, profiles/r2_mfma_util_and_traffic_b32_q1000.txt).  Here the MFMA wavefronts' stream is ds_read + MFMA only; the
// loader of the same SIMD issues its DMA beside them (separate issue ports) and sleeps at the barrier otherwise.
// Three LDS stages, ONE barrier per K step: the loaders wait (counted vmcnt) until tile kt has landed, everybody meets at the
// barrier, the loaders request tile kt+2 into the stage whose readers passed this very barrier after finishing tile kt-1, the MFMA
// wavefronts consume tile kt.  Same tile decomposition, same k order, same epilogue as gemm_big_body: bit-identical results.
template <int TN, int MODE, bool H2 = false>   // H2 (experimental, configurations 48 / 49): packed split-f16 operands, experimental/gemm_h2.h
__device__ __forceinline__ void gemm_ws_body(const GemmParams& p, const int bid) {
  constexpr int BM = 128, BN = 64 * TN;
  constexpr int STAGE = (BM + BN) * BK;   // floats per stage (one 32-deep K tile)
  constexpr int QW = BN / 32;             // W-tile DMA instructions per loader wavefront
  constexpr int EP = 32 * TN + 4;         // padded row of the epilogue staging tile
  constexpr int NSTG = 4;                 // LDS stages: the tile being consumed + up to three being filled (~12000 cycles of lookahead)
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int t = threadIdx.x, lane = t & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool loader = wave8 >= 4;
  const int wave = wave8 & 3;             // loader: which quarter of the tile rows it fetches; MFMA wavefront: its 64 x 32TN sub-tile
  int m0, n0;
  if (!gemm_tile_coords(p, BM, BN, bid, m0, n0)) return;
  const int KT = p.K / BK;

  if (loader) {
    const int drow = lane >> 3, pch = lane & 7;
    const float* a_ptr[4];
    bool a_ok[4];
    int c_hi0[4], c_wi0[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = wave * 32 + q * 8 + drow;
      const int lch = pch ^ ((row >> 1) & 7);
      const int m = m0 + row;
      a_ok[q] = m < p.M;
      const int mm = a_ok[q] ? m : 0;
      if constexpr (MODE == GEMM_DENSE) {
        a_ptr[q] = p.A + (size_t)mm * p.lda + lch * 4;
        c_hi0[q] = c_wi0[q] = 0;
      } else {
        int b, ho, side, wl;
        conv_row_decompose(p, mm, b, ho, side, wl);
        c_hi0[q] = ho * p.stride - p.pad;
        c_wi0[q] = wl * p.stride - p.pad;
        a_ptr[q] = p.A + (long)(((b * p.Hin + c_hi0[q]) * (2 * p.Win) + side * p.Win + c_wi0[q]) * p.Cin) + lch * 4;   // pixel (b, hi0, side, wi0): may lie in front of the tensor, only dereferenced in range
      }
    }
    const float* w_ptr[QW];
#pragma unroll
    for (int q = 0; q < QW; ++q) {
      const int row = wave * (BN / 4) + q * 8 + drow;
      const int lch = pch ^ ((row >> 1) & 7);
      w_ptr[q] = p.W + (size_t)(n0 + row) * p.K + lch * 4;
    }
    auto dma_tile = [&](int kt, int buf) {
      float* As = smem + buf * STAGE;
      float* Ws = As + BM * BK;
      if constexpr (MODE == GEMM_DENSE) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float* src = a_ok[q] ? a_ptr[q] + kt * BK : p.zeros;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(As + (wave * 32 + q * 8) * BK), 16, 0, 0);
        }
      } else {
        int ky, kx, c0;
        conv_ktile_decompose(p, kt, ky, kx, c0);
        const int tapoff = (ky * (2 * p.Win) + kx) * p.Cin + c0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int hi = c_hi0[q] + ky, wi = c_wi0[q] + kx;
          const bool ok = a_ok[q] && hi >= 0 && hi < p.Hin && wi >= 0 && wi < p.Win;
          const float* src = ok ? a_ptr[q] + tapoff : p.zeros;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(As + (wave * 32 + q * 8) * BK), 16, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < QW; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_ptr[q] + kt * BK),
                                         (__attribute__((address_space(3))) void*)(Ws + (wave * (BN / 4) + q * 8) * BK), 16, 0, 0);
    };
    // Ring of NSTG stages, barrier per tile: before barrier kt the loader makes sure ITS share of tile kt has landed while up to
    // two younger tiles stay in flight (counted vmcnt: one tile = 4 + QW DMA instructions of this wavefront); after the barrier it
    // requests tile kt+3 into the stage whose readers passed this very barrier after finishing tile kt-1.
    constexpr int PER = 4 + QW;
    if (p.ws_flags & 2) __builtin_amdgcn_s_setprio(3);    // the loaders' few instructions go out ahead of the MFMA wavefronts'
    dma_tile(0, 0);
    if (KT > 1) dma_tile(1, 1);
    if (KT > 2) dma_tile(2, 2);
    for (int kt = 0; kt < KT; ++kt) {
      const int younger = (KT - 1 - kt) < 2 ? (KT - 1 - kt) : 2;    // tiles requested after tile kt that may stay in flight
      if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                     // tile kt is in LDS for everybody; the readers of tile kt-1 are done with it
      if (kt + 3 < KT) dma_tile(kt + 3, (kt + 3) & (NSTG - 1));
    }
    __builtin_amdgcn_s_barrier();                       // (the MFMA wavefronts' "operand stages are free" barrier before the epilogue)
    return;
  }

  // ---- MFMA wavefronts ---------------------------------------------------------------------------------------------------------
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hh = lane >> 5;
  const int sw = (l31 >> 1) & 7;
  f32x16 acc[2][TN];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  f32x16 accx[H2 ? 2 : 1][H2 ? TN : 1];                 // H2: the cross terms, scaled by 2^11
  if constexpr (H2) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) accx[a][b][r] = 0.f;
  }
  constexpr int C4 = 8 * TN;
  constexpr int RPI = 64 / C4;
  constexpr int NIT = 32 / RPI;
  const int er = lane / C4, ec = (lane % C4) * 4;
  const int ncol = n0 + wn * 32 * TN + ec;
  f32x4 res[2][NIT];
  if (p.residual) {                                     // requested first: its latency is paid under the K loop
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int m = m0 + wm * 64 + a * 32 + it * RPI + er;
        const int mr = m < p.M ? (p.res_row_mod > 0 ? fastmod(m, p.fd_resrow) : m) : 0;
        res[a][it] = *reinterpret_cast<const f32x4*>(p.residual + (size_t)mr * p.ldr + ncol);
      }
  }
  struct Frag {
    f32x4 a[2], b[TN];
  };
  if (p.ws_flags & 1) __builtin_amdgcn_s_setprio(1);
  for (int kt = 0; kt < KT; ++kt) {
    __builtin_amdgcn_s_barrier();                       // the loaders have seen tile kt land
    asm volatile("" ::: "memory");                      // no LDS access of this tile may be scheduled above the barrier
    const float* As = smem + (kt & (NSTG - 1)) * STAGE + (wm * 64 + l31) * BK;
    const float* Ws = smem + (kt & (NSTG - 1)) * STAGE + BM * BK + (wn * 32 * TN + l31) * BK;
    auto load_frag = [&](int j) {                       // 8-deep slice j of the tile: one ds_read_b128 per 32-row block of A / W
      Frag r;
      const int ch = ((j * 2 + hh) ^ sw) * 4;
#pragma unroll
      for (int a = 0; a < 2; ++a) r.a[a] = *reinterpret_cast<const f32x4*>(As + a * 32 * BK + ch);
#pragma unroll
      for (int b = 0; b < TN; ++b) r.b[b] = *reinterpret_cast<const f32x4*>(Ws + b * 32 * BK + ch);
      return r;
    };
    if constexpr (H2) {
      h2_kstep<TN>(As, Ws, hh, sw, acc, accx);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      continue;
    }
    // the reads of slice j+1 are issued before the 8 TN MFMAs of slice j (hipcc on its own leaves a read two MFMAs of cover)
    Frag cur = load_frag(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      Frag nxt = cur;
      if (j + 1 < 4) nxt = load_frag(j + 1);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[a][e], cur.b[b][e], acc[a][b], 0, 0, 0);
      cur = nxt;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this tile's fragment reads are retired before the next barrier
  }
  if (p.ws_flags & 1) __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_s_barrier();                         // every MFMA wavefront is done reading the operand stages
  asm volatile("" ::: "memory");
  if constexpr (H2) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = fmaf(accx[a][b][r], 0x1p-11f, acc[a][b][r]);
  }

  float* Es = smem + wave * 32 * EP;
  f32x4 sc, bi, cs;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sc[e] = p.scale ? p.scale[ncol + e] : 1.f;
    bi[e] = p.bias ? p.bias[ncol + e] : 0.f;
    cs[e] = (ncol + e < p.colscale_n) ? p.colscale : 1.f;
  }
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) Es[((r & 3) + 8 * (r >> 2) + 4 * hh) * EP + b * 32 + l31] = acc[a][b][r];
    const int mb = m0 + wm * 64 + a * 32;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = it * RPI + er;
      const int m = mb + row;
      f32x4 v = *reinterpret_cast<const f32x4*>(Es + row * EP + ec);
      if (m < p.M) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = v[e];
          x = p.scale ? fmaf(x, sc[e], bi[e]) : x + bi[e];
          x *= cs[e];
          if constexpr (H2) {
            const float rv = res[a][it][e];
            if (p.residual) x += (p.h2_flags & 2) ? h2_unpack(__float_as_uint(rv)) : rv;
          } else
          if (p.residual) x += res[a][it][e];
          if (p.relu) x = (x < 0.f) ? 0.f : x;
          v[e] = x;
          if constexpr (H2) if (p.h2_flags & 1) v[e] = __uint_as_float(h2_pack_chk(x, p.h2_ovf));
        }
        *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + ncol) = v;
      }
    }
  }
}

template <int TN, int MODE, bool H2 = false>
__global__ __launch_bounds__(512) void gemm_ws_kernel(const GemmParams p) {
  gemm_ws_body<TN, MODE, H2>(p, blockIdx.x);
}

// knob KN_WS_FLAGS: bit 0 = s_setprio(1) around the MFMA wavefronts' loop, bit 1 = s_setprio(3) for the loaders

template <int TN, int MODE, bool H2 = false>
static int launch_ws_t(const GemmParams& p0, hipStream_t s) {
  constexpr int BM = 128, BN = 64 * TN;
  constexpr size_t smem = (size_t)4 * (BM + BN) * BK * sizeof(float);
  static_assert(smem >= (size_t)4 * 32 * (32 * TN + 4) * sizeof(float), "epilogue staging fits in the operand stages");
  GemmParams p = p0;
  if (p.N % BN != 0 || p.K % BK != 0 || p.M <= 0 || p.A2 != nullptr) return -1;
  if (p.ldc % 4 != 0 || (p.residual && p.ldr % 4 != 0)) return -1;
  if (((uintptr_t)p.C & 15) || ((uintptr_t)p.residual & 15)) return -1;
  if (p.zeros == nullptr) p.zeros = gemm_zero_buffer();
  if (p.zeros == nullptr) return -2;
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ws_kernel<TN, MODE, H2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem) != hipSuccess)
      return -2;
    attr_set.set();
  }
  if (!gemm_fill_divs(p, MODE, BM, BN)) return -1;
  if constexpr (H2) { p.h2_ovf = h2_overflow_flag(); if (p.h2_ovf == nullptr) return -2; }
  p.ws_flags = knob(KN_WS_FLAGS);
  const int tiles = gemm_grid_tiles(p, BM, BN);
  hipLaunchKernelGGL((gemm_ws_kernel<TN, MODE, H2>), dim3(tiles), dim3(512), smem, s, p);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int TN, int MODE, int NST, int X = 0, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void gemm_big_kernel(const GemmParams p) {
  gemm_big_body<TN, MODE, NST, X, NW>(p, blockIdx.x);
}

// two independent problems in one grid (common.h: launch_gemm_dual_cfg)
template <int TN, int MODE, int NST>
__global__ __launch_bounds__(256, 2) void gemm_big_dual_kernel(const GemmParams p0, const GemmParams p1, const int tiles0) {
  if ((int)blockIdx.x < tiles0) gemm_big_body<TN, MODE, NST>(p0, blockIdx.x);
  else gemm_big_body<TN, MODE, NST>(p1, (int)blockIdx.x - tiles0);
}

template <int TN>
static int launch_big_dual_t(const GemmParams& a, const GemmParams& b, hipStream_t s) {
  constexpr int BM = 128, BN = 64 * TN;
  constexpr size_t smem = (size_t)2 * (BM + BN) * BK * sizeof(float);
  GemmParams p0 = a, p1 = b;
  for (GemmParams* p : {&p0, &p1}) {
    if (p->N % BN != 0 || p->K % BK != 0 || p->M <= 0 || p->A2 != nullptr) return -1;
    if (p->ldc % 4 != 0 || (p->residual && p->ldr % 4 != 0)) return -1;
    if (((uintptr_t)p->C & 15) || ((uintptr_t)p->residual & 15)) return -1;
    if (p->zeros == nullptr) p->zeros = gemm_zero_buffer();
    if (p->zeros == nullptr) return -2;
  }
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_big_dual_kernel<TN, GEMM_CONV, 2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return -2;
    attr_set.set();
  }
  if (!gemm_fill_divs(p0, GEMM_CONV, BM, BN) || !gemm_fill_divs(p1, GEMM_CONV, BM, BN)) return -1;
  const int tiles0 = gemm_grid_tiles(p0, BM, BN), tiles1 = gemm_grid_tiles(p1, BM, BN);
  hipLaunchKernelGGL((gemm_big_dual_kernel<TN, GEMM_CONV, 2>), dim3(tiles0 + tiles1), dim3(256), smem, s, p0, p1, tiles0);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_gemm_big_dual(int mode, int variant, const GemmParams& p0, const GemmParams& p1, hipStream_t s) {
  if (mode != GEMM_CONV) return -1;
  return variant == 0 ? launch_big_dual_t<2>(p0, p1, s) : variant == 1 ? launch_big_dual_t<1>(p0, p1, s) : -1;
}

template <int TN, int MODE, int NST, int X = 0, int NW = 4>
static int launch_big_t(const GemmParams& p0, hipStream_t s) {
  constexpr int BM = 32 * NW, BN = 64 * TN;
  constexpr size_t smem = (size_t)NST * (BM + BN) * BK * sizeof(float);
  static_assert(smem >= (size_t)NW * 32 * (32 * TN + 4) * sizeof(float), "epilogue staging fits in the operand stages");
  GemmParams p = p0;
  if (p.N % BN != 0 || p.K % BK != 0 || p.M <= 0 || p.A2 != nullptr) return -1;
  if (p.ldc % 4 != 0 || (p.residual && p.ldr % 4 != 0)) return -1;
  if (((uintptr_t)p.C & 15) || ((uintptr_t)p.residual & 15)) return -1;
  if (p.zeros == nullptr) p.zeros = gemm_zero_buffer();
  if (p.zeros == nullptr) return -2;
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_big_kernel<TN, MODE, NST, X, NW>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return -2;
    attr_set.set();
  }
  if (!gemm_fill_divs(p, MODE, BM, BN)) return -1;
  if constexpr (X == 2) { p.h2_ovf = h2_overflow_flag(); if (p.h2_ovf == nullptr) return -2; }
  const int tiles = gemm_grid_tiles(p, BM, BN);
  hipLaunchKernelGGL((gemm_big_kernel<TN, MODE, NST, X, NW>), dim3(tiles), dim3(64 * NW), smem, s, p);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// variant 0: 128 x 128 tile, 1: 128 x 64 (two LDS stages); 2, 3: the same tiles with the three-stage ring; 4, 5: the same
// tiles with 4 loader + 4 MFMA wavefronts (gemm_ws_body)
int launch_gemm_big(int mode, int variant, const GemmParams& p, hipStream_t s) {
  if (mode == GEMM_DENSE && p.lda % 4 != 0) return -1;
  if (mode != GEMM_DENSE && mode != GEMM_CONV) return -1;
  const bool d = mode == GEMM_DENSE;
  switch (variant) {
    case 0: return d ? launch_big_t<2, GEMM_DENSE, 2>(p, s) : launch_big_t<2, GEMM_CONV, 2>(p, s);
    case 1: return d ? launch_big_t<1, GEMM_DENSE, 2>(p, s) : launch_big_t<1, GEMM_CONV, 2>(p, s);
    case 2: return d ? launch_big_t<2, GEMM_DENSE, 3>(p, s) : launch_big_t<2, GEMM_CONV, 3>(p, s);
    case 3: return d ? launch_big_t<1, GEMM_DENSE, 3>(p, s) : launch_big_t<1, GEMM_CONV, 3>(p, s);
    case 6: return d ? launch_big_t<2, GEMM_DENSE, 2, 1>(p, s) : launch_big_t<2, GEMM_CONV, 2, 1>(p, s);   // direct epilogue
    case 7: return d ? launch_big_t<1, GEMM_DENSE, 2, 1>(p, s) : launch_big_t<1, GEMM_CONV, 2, 1>(p, s);
    case 8: return d ? launch_big_t<2, GEMM_DENSE, 2, 2>(p, s) : launch_big_t<2, GEMM_CONV, 2, 2>(p, s);   // packed split-f16 operands
    case 9: return d ? launch_big_t<1, GEMM_DENSE, 2, 2>(p, s) : launch_big_t<1, GEMM_CONV, 2, 2>(p, s);
    case 10: return d ? launch_ws_t<2, GEMM_DENSE, true>(p, s) : launch_ws_t<2, GEMM_CONV, true>(p, s);   // packed split-f16, wave-specialised
    case 11: return d ? launch_ws_t<1, GEMM_DENSE, true>(p, s) : launch_ws_t<1, GEMM_CONV, true>(p, s);
    case 12: return d ? launch_big_t<2, GEMM_DENSE, 3, 2, 8>(p, s) : launch_big_t<2, GEMM_CONV, 3, 2, 8>(p, s);   // packed split-f16, 256 x 128 tile, 8 wavefronts, 3 stages
    case 4: return d ? launch_ws_t<2, GEMM_DENSE>(p, s) : launch_ws_t<2, GEMM_CONV>(p, s);   // wave-specialised 128 x 128
    case 5: return d ? launch_ws_t<1, GEMM_DENSE>(p, s) : launch_ws_t<1, GEMM_CONV>(p, s);   // wave-specialised 128 x 64
    default: return -1;
  }
}
