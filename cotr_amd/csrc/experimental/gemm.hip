// RESEARCH LIBRARY COPY of csrc/gemm.hip (libcotr_hip_exp.so only): the product file with the research / dead-end paths that used to sit
// behind #ifdef COTR_EXPERIMENTAL in it resolved IN (tools/unifdef_exp.py -D).  The product never compiles this file.
// fp32 MFMA GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
// One kernel template covers every dense contraction of the COTR forward path: the 43 ResNet
// convolutions (as implicit GEMM over NHWC "side-by-side" activations), the 1x1 input_proj, and
// every Linear of the transformer and the corr MLP - with FrozenBN / bias / q-scale / residual /
// ReLU fused into the epilogue.
//
// Numerics: v_mfma_f32_32x32x2_f32, fp32 operands and fp32 accumulation (bit-equal to an fmaf
// chain).  The 1e-3 px parity bar of BASELINE.json rules out bf16/fp16 operands.
//
// Tiling: 256 threads = 4 wavefronts arranged WM x WN; each wavefront owns TM x TN MFMA blocks of
// 32x32, so the workgroup tile is BM x BN = (WM*TM*32) x (WN*TN*32), K step 32.
// Global -> registers (float4, 128-B row segments, coalesced) -> LDS (rows padded to 36 floats so
// the ds_read_b128 fragment reads of 16 different rows hit 16 different 16-B slots) -> MFMA.
// The loads of K-tile t+1 are issued before the MFMAs of tile t (register prefetch).
//
// Fragment trick: lane l = (row l&31, half l>>5) reads ONE float4 = columns j*8+half*4 .. +3 of its
// row and feeds element e to the e-th of 4 MFMAs; both operands use the same column permutation so
// the contraction is unchanged, and each operand costs one ds_read_b128 per 4 MFMAs.
#include "../common.h"

#define BK 32
#define LDSLD 36


template <int WM, int WN, int TM, int TN, int MODE>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
  const int bid = blockIdx.x;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int RA = BM / 32, RW = BN / 32;
  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDSLD];
  float* As = smem;
  float* Ws = smem + BM * LDSLD;

  const int t = threadIdx.x;
  int m0, n0;
  if (!gemm_tile_coords(p, BM, BN, bid, m0, n0)) return;
  const int lr = t >> 3, lc = (t & 7) * 4;
  const int KT = p.K / BK;

  // ---- per-thread load bookkeeping -----------------------------------------------------------
  const float* a_ptr[RA];
  const float* a2_ptr[RA];
  bool a_ok[RA];
  int c_hi0[RA], c_wi0[RA];
  // x + pos prologue (q | k column tiles of an in-projection): the A2 loads must not sit behind a per-element wave-uniform
  // branch (hipcc then waits for every load before the next one: measured 4 dependent L2 round trips per K step), so a launch
  // with an A2 operand ALWAYS loads it - tiles that do not use it read a zero row with stride 0 - and adds it at the LDS store
  const bool has_a2 = (MODE == GEMM_DENSE) && p.A2 != nullptr;
  const bool use_a2 = has_a2 && fastmod(n0, p.fd_a2per) < p.a2_width;
  const int a2_step = use_a2 ? BK : 0;
  // stem bookkeeping (MODE == GEMM_STEM): one output pixel per thread, 16 k's per tile
  int s_hi0 = 0, s_wi0 = 0;
  const float* s_base = nullptr;
  bool s_ok = false;

  if constexpr (MODE == GEMM_DENSE) {
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int m = m0 + lr + 32 * i;
      a_ok[i] = m < p.M;
      const int mm = a_ok[i] ? m : 0;
      a_ptr[i] = p.A + (size_t)mm * p.lda + lc;
      a2_ptr[i] = use_a2 ? p.A2 + (size_t)(p.a2_row_mod ? fastmod(mm, p.fd_a2row) : mm) * p.lda2 + lc : p.zeros;
    }
  } else if constexpr (MODE == GEMM_CONV) {
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int m = m0 + lr + 32 * i;
      a_ok[i] = m < p.M;
      const int mm = a_ok[i] ? m : 0;
      int b, ho, side, wl;
      conv_row_decompose(p, mm, b, ho, side, wl);
      c_hi0[i] = ho * p.stride - p.pad;
      c_wi0[i] = wl * p.stride - p.pad;
      // pixel (b, 0, side*Win + 0), channel lc
      a_ptr[i] = p.A + (long)(((b * p.Hin + c_hi0[i]) * (2 * p.Win) + side * p.Win + c_wi0[i]) * p.Cin) + lc;   // pixel (b, hi0, side, wi0): may lie in front of the tensor, only dereferenced in range
      a2_ptr[i] = nullptr;
    }
  } else {  // GEMM_STEM
    const int m = m0 + (t & (BM - 1));
    s_ok = m < p.M;
    const int mm = s_ok ? m : 0;
    const int b = mm / (128 * 256);
    const int rem = mm - b * (128 * 256);
    const int ho = rem >> 8;
    const int wo = rem & 255;
    const int side = wo >> 7;
    const int wl = wo & 127;
    s_hi0 = 2 * ho - 3;
    s_wi0 = 2 * wl - 3;
    s_base = p.A + (size_t)b * 3 * 256 * 512 + side * 256;
  }
  const float* w_ptr[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) w_ptr[i] = p.W + (size_t)(n0 + lr + 32 * i) * p.K + lc;

  f32x4 ra[RA], ra2[RA], rw[RW];
  float rs[16];

  auto load_tile = [&](int kt) {
#if defined(COTR_ABL) && COTR_ABL == 1  // ablation: only the first global load
    if (kt > 0) return;
#endif
    if constexpr (MODE == GEMM_DENSE) {
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (a_ok[i]) v = *reinterpret_cast<const f32x4*>(a_ptr[i] + kt * BK);
        ra[i] = v;
      }
      if (has_a2) {
#pragma unroll
        for (int i = 0; i < RA; ++i) ra2[i] = *reinterpret_cast<const f32x4*>(a2_ptr[i] + kt * a2_step);
      }
    } else if constexpr (MODE == GEMM_CONV) {
      int ky, kx, c0;
      conv_ktile_decompose(p, kt, ky, kx, c0);
      const int tapoff = (ky * (2 * p.Win) + kx) * p.Cin + c0;   // wave-uniform element offset of this tap / channel tile
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const int hi = c_hi0[i] + ky, wi = c_wi0[i] + kx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (a_ok[i] && hi >= 0 && hi < p.Hin && wi >= 0 && wi < p.Win)
          v = *reinterpret_cast<const f32x4*>(a_ptr[i] + tapoff);
        ra[i] = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = kt * BK + (t >> 7) + 2 * i;  // BM == 128: two k's per pass over the tile
        const int c = k / 49;
        const int r = k - c * 49;
        const int ky = r / 7;
        const int kx = r - ky * 7;
        const int hi = s_hi0 + ky, wi = s_wi0 + kx;
        float v = 0.f;
        if (s_ok && k < 147 && hi >= 0 && hi < 256 && wi >= 0 && wi < 256)
          v = s_base[((size_t)c * 256 + hi) * 512 + wi];
        rs[i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < RW; ++i) rw[i] = *reinterpret_cast<const f32x4*>(w_ptr[i] + kt * BK);
  };

  auto store_tile = [&]() {
    if constexpr (MODE == GEMM_STEM) {
#pragma unroll
      for (int i = 0; i < 16; ++i) As[(t & (BM - 1)) * LDSLD + (t >> 7) + 2 * i] = rs[i];
    } else {
      if (has_a2) {
#pragma unroll
        for (int i = 0; i < RA; ++i) ra[i] += ra2[i];
      }
#pragma unroll
      for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4*>(&As[(lr + 32 * i) * LDSLD + lc]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < RW; ++i) *reinterpret_cast<f32x4*>(&Ws[(lr + 32 * i) * LDSLD + lc]) = rw[i];
  };

  // ---- main loop -----------------------------------------------------------------------------
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, hh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  load_tile(0);
  for (int kt = 0; kt < KT; ++kt) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (kt + 1 < KT) load_tile(kt + 1);
#if defined(COTR_ABL) && COTR_ABL == 2  // ablation: no LDS reads / MFMAs
    if (p.M > 0) continue;
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a)
        af[a] = *reinterpret_cast<const f32x4*>(&As[((wm * TM + a) * 32 + l31) * LDSLD + j * 8 + hh * 4]);
#pragma unroll
      for (int b = 0; b < TN; ++b)
        bf[b] = *reinterpret_cast<const f32x4*>(&Ws[((wn * TN + b) * 32 + l31) * LDSLD + j * 8 + hh * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][e], bf[b][e], acc[a][b], 0, 0, 0);
    }
  }

  // ---- epilogue: D[i][j], j = lane&31 (column n), i = (r&3) + 8*(r>>2) + 4*(lane>>5) (row m) --
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int n = n0 + (wn * TN + b) * 32 + l31;
    const float sc = p.scale ? p.scale[n] : 1.f;
    const float bi = p.bias ? p.bias[n] : 0.f;
    const float cs = (n < p.colscale_n) ? p.colscale : 1.f;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int mb = m0 + (wm * TM + a) * 32 + 4 * hh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        if (m < p.M) {
          float v = acc[a][b][r];
          v = p.scale ? fmaf(v, sc, bi) : v + bi;
          v *= cs;
          if (p.residual) v += p.residual[(size_t)m * p.ldr + n];
          if (p.relu) v = (v < 0.f) ? 0.f : v;  // NaN passes through like torch.relu (fmaxf would drop it)
          p.C[(size_t)m * p.ldc + n] = v;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// K-split variant for the small-M regime (one image pair: M = 512 .. 8192 rows, 1000 query rows).
// There the spatial tiling above leaves most of the 256 CUs idle (32 .. 128 workgroups) and every
// wavefront walks the whole K serially.  Here ALL NWK wavefronts of a workgroup compute the SAME
// (TM*32) x (TN*32) output tile, each over its own 32-wide slice of a (NWK*32)-deep K step, so a
// K = 2304 contraction is 9 steps of 16 MFMAs per wavefront instead of 72; the NWK partial
// accumulators are summed through LDS in a fixed order (deterministic, no atomics, no second
// launch) and the same fused epilogue is applied.
// ---------------------------------------------------------------------------------------------
template <int NWK, int TM, int TN, int MODE, int DB>
__device__ __forceinline__ void gemm_ks_body(const GemmParams& p, const int bid) {
  constexpr int NT = NWK * 64;
  // TN == 0 selects the 32 x 16 tile built from v_mfma_f32_16x16x4_f32 (two 16-row blocks x one 16-column block):
  // twice the workgroups of the 32 x 32 tile for the M <= 512, N = 256 shapes that otherwise fill half the CUs
  constexpr bool N16 = (TN == 0);
  static_assert(!N16 || TM == 1, "the 16-column tile is 32 rows tall");
  constexpr int BM = TM * 32, BN = N16 ? 16 : TN * 32;
  constexpr int KS = NWK * BK;        // K elements per step
  constexpr int LD = KS + 4;          // padded LDS row
  constexpr int C4 = NWK * 8;         // float4 per row per step
  constexpr int PA = BM / 8, PW = BN / 8;  // passes: 8 rows per pass (NT / C4 == 8)
  constexpr int NB = N16 ? 1 : TM * TN;
  static_assert(NT / C4 == 8, "8 rows per pass");
  // DB == 4 ("patch"): 3x3 stride-1 convolution over Cin = 256 channels (layer3) - a K step is exactly one tap, and the nine
  // A tiles of a 32-pixel row segment are nine shifted views of the same 3 x 34 input pixels.  Those are loaded ONCE (102 rows
  // of 1 KB instead of 9 x 32) and the taps read their A fragments from the patch at the tap's offset; the weights go straight
  // to registers.  Everything the workgroup needs is requested up front: one memory round trip instead of nine half ones.
  constexpr bool PATCH = (DB == 4);
  static_assert(!PATCH || (MODE == GEMM_CONV && N16 && NWK == 8), "the patch variant is the 32 x 16 tile, 8 wavefronts, convolution");
  constexpr int TILE = PATCH ? BN * LD : (BM + BN) * LD;  // floats per LDS stage (A rows then W rows; PATCH: W rows only)
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int t = threadIdx.x;
  int m0, n0;
  if (!gemm_tile_coords(p, BM, BN, bid, m0, n0)) return;
#define KS_STAMP(slot)                                                        \
  do {                                                                        \
    if (p.dbg != nullptr && t == 0) p.dbg[(size_t)bid * 8 + (slot)] = wall_clock64(); \
  } while (0)
  KS_STAMP(0);
  // with 8 wavefronts a wavefront loads whole rows (C4 == 64 lanes per row): the row index, and with it all of the row's
  // addressing (the convolution's pixel decomposition: three integer divisions per row), is wave-uniform - taken through
  // readfirstlane it runs on the scalar unit instead of per lane behind exec-masked branches
  const int lr = (C4 == 64) ? __builtin_amdgcn_readfirstlane(t / C4) : t / C4;
  const int lc4 = t % C4;
  const int ktl = lc4 >> 3;            // which 32-wide k-tile of the step this thread loads
  const int lcc = (lc4 & 7) * 4;       // column inside that k-tile
  const int KT = p.K / BK;
  const int steps = (KT + NWK - 1) / NWK;

  const float* a_ptr[PA];
  const float* a2_ptr[PA];
  bool a_ok[PA];
  int c_hi0[PA], c_wi0[PA];
  // see gemm_kernel: with an A2 operand every tile loads it (zero row, stride 0 where unused), no per-element uniform branch
  const bool has_a2 = (MODE == GEMM_DENSE) && p.A2 != nullptr;
  const bool use_a2 = has_a2 && fastmod(n0, p.fd_a2per) < p.a2_width;
  const int a2_step = use_a2 ? KS : 0;
  if constexpr (MODE == GEMM_DENSE) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int m = m0 + lr + 8 * i;
      a_ok[i] = m < p.M;
      const int mm = a_ok[i] ? m : 0;
      a_ptr[i] = p.A + (size_t)mm * p.lda + lc4 * 4;
      a2_ptr[i] = use_a2 ? p.A2 + (size_t)(p.a2_row_mod ? fastmod(mm, p.fd_a2row) : mm) * p.lda2 + lc4 * 4 : p.zeros;
    }
  } else {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int m = m0 + lr + 8 * i;
      a_ok[i] = m < p.M;
      const int mm = a_ok[i] ? m : 0;
      int b, ho, side, wl;
      conv_row_decompose(p, mm, b, ho, side, wl);
      c_hi0[i] = ho * p.stride - p.pad;
      c_wi0[i] = wl * p.stride - p.pad;
      a_ptr[i] = p.A + (long)(((b * p.Hin + c_hi0[i]) * (2 * p.Win) + side * p.Win + c_wi0[i]) * p.Cin) + lcc;   // pixel (b, hi0, side, wi0): may lie in front of the tensor, only dereferenced in range
      a2_ptr[i] = nullptr;
    }
  }
  const float* w_ptr[PW];
#pragma unroll
  for (int i = 0; i < PW; ++i) w_ptr[i] = p.W + (size_t)(n0 + lr + 8 * i) * p.K + lc4 * 4;

  f32x4 ra[PA], ra2[PA], rw[PW];

  auto load_tile = [&](int st) {
#if defined(COTR_ABL) && COTR_ABL == 1  // ablation: only the first global load
    if (st > 0) return;
#endif
    const int kt = st * NWK + ktl;
    const bool k_ok = kt < KT;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if constexpr (MODE == GEMM_DENSE) {
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        f32x4 v = z;
        if (a_ok[i] && k_ok) v = *reinterpret_cast<const f32x4*>(a_ptr[i] + st * KS);
        ra[i] = v;
      }
      if (has_a2) {   // rows past M re-read a valid row (never stored); k-tiles past K read the zero row (address select, no branch)
#pragma unroll
        for (int i = 0; i < PA; ++i) {
          const float* src = k_ok ? a2_ptr[i] + st * a2_step : p.zeros;
          ra2[i] = *reinterpret_cast<const f32x4*>(src);
        }
      }
    } else {
      int ky, kx, c0;
      conv_ktile_decompose(p, kt, ky, kx, c0);
      const int tapoff = (ky * (2 * p.Win) + kx) * p.Cin + c0;   // wave-uniform element offset of this tap / channel tile
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int hi = c_hi0[i] + ky, wi = c_wi0[i] + kx;
        f32x4 v = z;
        if (a_ok[i] && k_ok && hi >= 0 && hi < p.Hin && wi >= 0 && wi < p.Win)
          v = *reinterpret_cast<const f32x4*>(a_ptr[i] + tapoff);
        ra[i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < PW; ++i) rw[i] = k_ok ? *reinterpret_cast<const f32x4*>(w_ptr[i] + st * KS) : z;
  };
  auto store_tile = [&](int buf) {
    float* As = smem + buf * TILE;
    float* Ws = As + BM * LD;
    if (has_a2) {
#pragma unroll
      for (int i = 0; i < PA; ++i) ra[i] += ra2[i];
    }
#pragma unroll
    for (int i = 0; i < PA; ++i) *reinterpret_cast<f32x4*>(&As[(lr + 8 * i) * LD + lc4 * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < PW; ++i) *reinterpret_cast<f32x4*>(&Ws[(lr + 8 * i) * LD + lc4 * 4]) = rw[i];
  };

  const int lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int l15 = lane & 15, q4 = lane >> 4;
  f32x16 acc[TM][N16 ? 1 : TN];
  f32x4 acc16[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < (N16 ? 1 : TN); ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  auto compute = [&](int buf) {
    const float* As = smem + buf * TILE;
    const float* Ws = As + BM * LD;
#if defined(COTR_ABL) && COTR_ABL == 2  // ablation: no LDS reads / MFMAs
    if (p.M > 0) return;
#endif
    if constexpr (N16) {
      // 16x16x4: lane (row l&15, k group l>>4) reads 4 consecutive k; element e of the float4 feeds the e-th MFMA
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int ko = wave * BK + jj * 16 + q4 * 4;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(&As[l15 * LD + ko]);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(&As[(16 + l15) * LD + ko]);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(&Ws[l15 * LD + ko]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc16[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], bb[e], acc16[0], 0, 0, 0);
          acc16[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], bb[e], acc16[1], 0, 0, 0);
        }
      }
      return;
    }
    constexpr int TNN = N16 ? 1 : TN;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 af[TM], bf[TNN];
#pragma unroll
      for (int a = 0; a < TM; ++a)
        af[a] = *reinterpret_cast<const f32x4*>(&As[(a * 32 + l31) * LD + wave * BK + j * 8 + hh * 4]);
#pragma unroll
      for (int b = 0; b < TN; ++b)
        bf[b] = *reinterpret_cast<const f32x4*>(&Ws[(b * 32 + l31) * LD + wave * BK + j * 8 + hh * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][e], bf[b][e], acc[a][b], 0, 0, 0);
    }
  };

  // DB == 2: global -> LDS directly (global_load_lds_dwordx4, no VGPR staging, no ds_write pass).  With
  // 8 wavefronts one wave instruction moves exactly one padded row (64 lanes x 16 B = 1024 B of data),
  // so the LDS destination is wave-uniform as the instruction requires; out-of-range rows / padded taps
  // read a zero buffer instead.  The transfer of step st+1 runs under the MFMAs of step st.
  auto dma_tile = [&](int st, int buf) {
    static_assert(DB < 2 || C4 == 64, "one wave instruction per LDS row needs 8 wavefronts");
    float* As = smem + buf * TILE;
    float* Ws = As + BM * LD;
    const int kt = st * NWK + ktl;
    const bool k_ok = kt < KT;
    const float* zsrc = p.zeros;
    if constexpr (MODE == GEMM_DENSE) {
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const float* src = (a_ok[i] && k_ok) ? a_ptr[i] + st * KS : zsrc;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(As + (lr + 8 * i) * LD), 16, 0, 0);
      }
    } else {
      int ky, kx, c0;
      conv_ktile_decompose(p, kt, ky, kx, c0);
      const int tapoff = (ky * (2 * p.Win) + kx) * p.Cin + c0;   // wave-uniform element offset of this tap / channel tile
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int hi = c_hi0[i] + ky, wi = c_wi0[i] + kx;
        const bool ok = a_ok[i] && k_ok && hi >= 0 && hi < p.Hin && wi >= 0 && wi < p.Win;
        const float* src = ok ? a_ptr[i] + tapoff : zsrc;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(As + (lr + 8 * i) * LD), 16, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const float* src = k_ok ? w_ptr[i] + st * KS : zsrc;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(Ws + (lr + 8 * i) * LD), 16, 0, 0);
    }
  };

  if constexpr (PATCH) {
    // a 32-row tile is one 32-pixel segment of an output row (Wout a multiple of 32) or, on layer3's 16-wide halves, the two
    // 16-pixel rows of the left and the right half (each with its own zero padding at the seam): nseg segments of seg pixels
    const int seg = p.Wout == 16 ? 16 : 32, segrows = 3 * (seg + 2), nrows = (32 / seg) * segrows;
    const int a1_off = (seg == 16 ? segrows : 16) * LD;  // tile rows 16..31: the second segment, or the same segment 16 pixels on
    // Everything the workgroup needs is requested up front, in the order it is used: [patch rows of input row dy, weights of the
    // three taps ky = dy] for dy = 0, 1, 2.  Measured with the phase stamps (tools/conv_phases.py): the K loop is bound by what
    // one CU can pull while all 256 pull at once (~35-40 GB/s per CU, 9-10 TB/s chip-wide out of the L2s) - 442 KB per workgroup
    // in the 9-tile form (10.6 us), 259 KB here (7.4 us) - and the tile's MFMAs are 4.5 us on top unless they overlap the tail of
    // the loads: the taps of input row dy start as soon as ITS rows and weights have landed (counted vmcnt: every wavefront issues
    // exactly 5 patch DMAs + 6 weight loads per phase, idle DMA slots go to a dummy row).
    f32x4 wreg[9][2];
    {
      int b, ho, side0, wl0;
      conv_row_decompose(p, m0, b, ho, side0, wl0);
      // all of the patch addressing is wave-uniform (a wave instruction moves one pixel): kept on the scalar unit by taking the
      // wavefront index through readfirstlane - per-lane integer divisions and exec-masked branches around every DMA were 4 us
      const int wv = __builtin_amdgcn_readfirstlane(wave);
      // 32-bit element offsets from the pair's image (a pair's activation tensor is far below 2^31 floats), every term that does not
      // depend on the slot hoisted: round 2's form spent ~60 scalar instructions (64-bit multiplies) per DMA, 2.8 us from the first
      // to the last of the 33 requests (profiles/r3_conv_phases_after_fastdiv.txt: "loads issued" 3.5 us after entry)
      const float* img = p.A + (size_t)b * p.Hin * (2 * p.Win) * p.Cin;
      const float* wrow = p.W + (size_t)(n0 + l15) * p.K + wave * BK + q4 * 4;
      const int per_dy = (32 / seg) * (seg + 2);          // 36 (two 16-pixel segments) or 34 patch rows per input row
      const int Cin = 256;                                // (PATCH is the 256-channel 3x3: cfg_fits)
      const int lane_off = lane * 4;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int hi = ho - 1 + dy;
        const bool hi_ok = hi >= 0 && hi < p.Hin;
        const int row_off = (hi * 2 * p.Win + side0 * p.Win + wl0 - 1) * Cin;   // pixel (hi, side0, wl0 - 1)
#pragma unroll
        for (int i = 0; i < 5; ++i) {                     // one wave instruction = one pixel (256 channels = 1 KB)
          const int sl = wv + NWK * i;                    // slot 0..39 of this phase
          const int sg = (seg == 16 && sl >= 18) ? 1 : 0, j = sl - sg * 18;
          const int wi = wl0 - 1 + j;
          const bool slot_ok = sl < per_dy;
          const bool ok = slot_ok && hi_ok && wi >= 0 && wi < p.Win;
          const int off = row_off + (sg * p.Win + j) * Cin;
          const float* src = ok ? img + off + lane_off : p.zeros;
          const int r = slot_ok ? sg * segrows + dy * (seg + 2) + j : nrows;   // dummy row behind the patch
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(smem + r * LD), 16, 0, 0);
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
            // (asm: hipcc's own wait insertion falls back to vmcnt(0) in front of the first MFMA because of the scalar branches
            // between the loads; the counted waits below cover these registers - they are issued in phase order)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wreg[dy * 3 + kx][jj]) : "v"(wrow + (dy * 3 + kx) * KS + jj * 16) : "memory");
      }
    }
    KS_STAMP(1);
    // per wavefront 33 loads are in flight, in use order: [5 patch DMAs, 2 weight loads of tap 3dy], [2 of tap 3dy+1], [2 of tap
    // 3dy+2] for dy = 0, 1, 2.  A tap starts when its own data has landed (counted wait); only the first tap of an input row needs
    // the barrier (patch rows come from every wavefront), the other two need this wavefront's own weight registers only.
#define PATCH_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, kx = tap - dy * 3;
      switch (tap) {
        case 0: PATCH_WAIT(26); break;
        case 1: PATCH_WAIT(24); break;
        case 2: PATCH_WAIT(22); break;
        case 3: PATCH_WAIT(15); break;
        case 4: PATCH_WAIT(13); break;
        case 5: PATCH_WAIT(11); break;
        case 6: PATCH_WAIT(4); break;
        case 7: PATCH_WAIT(2); break;
        default: PATCH_WAIT(0); break;
      }
      if (kx == 0) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tap == 0) KS_STAMP(2);
      }
      const float* As = smem + (dy * (seg + 2) + kx) * LD;   // row i of a segment = pixel (dy, kx + i) of its patch
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int ko = wave * BK + jj * 16 + q4 * 4;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(&As[l15 * LD + ko]);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(&As[l15 * LD + ko + a1_off]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc16[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], wreg[tap][jj][e], acc16[0], 0, 0, 0);
          acc16[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], wreg[tap][jj][e], acc16[1], 0, 0, 0);
        }
      }
    }
#undef PATCH_WAIT
    __syncthreads();
  } else if constexpr (DB == 3) {
    // Three LDS stages, TWO tiles of LDS-DMA in flight.  At one pair a CU holds a single workgroup and a K step's MFMAs
    // (0.2 us) cannot cover the L2 / Infinity-Cache latency of the next tile's DMA (~1.3 us): with two stages every step
    // costs one full DMA latency.  Here the wait before the barrier of step st is a COUNTED vmcnt that covers tile st only
    // (one tile = PA + PW DMA instructions per wavefront), tile st+1 stays in flight across the raw s_barrier
    // (__syncthreads() would drain the queue) and tile st+2 is requested right after it, into the stage whose last reads
    // (step st-1) every wavefront retired (lgkmcnt(0)) before arriving at this barrier.
    static_assert(PA + PW == 6, "the counted wait below is written for 6 DMA instructions per tile (32 x 16 tile, 8 waves)");
    dma_tile(0, 0);
    if (steps > 1) dma_tile(1, 1);
    KS_STAMP(1);
    int stg = 0;
    for (int st = 0; st < steps; ++st) {
      if (st + 1 < steps) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");                    // no LDS access of this step may be scheduled above the barrier
      if (st == 0) KS_STAMP(2);
      if (st + 2 < steps) dma_tile(st + 2, stg == 0 ? 2 : stg - 1);
      compute(stg);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this step's fragment reads are retired before the next barrier
      stg = stg == 2 ? 0 : stg + 1;
    }
    __syncthreads();                                    // the stages are reused by the cross-wave reduction below
  } else if constexpr (DB == 2) {
    dma_tile(0, 0);
    LDS_DMA_WAIT_ALL();
    __syncthreads();
    for (int st = 0; st < steps; ++st) {
      if (st + 1 < steps) dma_tile(st + 1, (st + 1) & 1);
      compute(st & 1);
      LDS_DMA_WAIT_ALL();
      __syncthreads();
    }
  } else if constexpr (DB) {
    load_tile(0);
    // two LDS stages, ONE barrier per K step: a wavefront that has finished its MFMAs of step st
    // writes the tile of step st+1 into the other stage while slower wavefronts still compute,
    // and the global loads of step st+2 are in flight across the whole step.
    store_tile(0);
    if (steps > 1) load_tile(1);
    __syncthreads();
    for (int st = 0; st < steps; ++st) {
      compute(st & 1);
      if (st + 1 < steps) {
        store_tile((st + 1) & 1);          // registers hold tile st+1 (loaded during step st-1 / prologue)
        if (st + 2 < steps) load_tile(st + 2);
      }
      __syncthreads();
    }
  } else {
    load_tile(0);
    for (int st = 0; st < steps; ++st) {
      __syncthreads();
      store_tile(0);
      __syncthreads();
      if (st + 1 < steps) load_tile(st + 1);
      compute(0);
    }
    __syncthreads();
  }

  // ---- cross-wave reduction through LDS: red[wave][block][r][lane] ----------------------------
  KS_STAMP(3);
  float* red = smem;
  if constexpr (N16) {
    // D of 16x16x4: column = lane&15, row = (lane>>4)*4 + reg; 2 blocks x 4 regs = 8 slots per lane
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wave * 8) + blk * 4 + r) * 64 + lane] = acc16[blk][r];
    __syncthreads();
    const int n = n0 + l15;
    const float sc = p.scale ? p.scale[n] : 1.f;
    const float bi = p.bias ? p.bias[n] : 0.f;
    const float cs = (n < p.colscale_n) ? p.colscale : 1.f;
    for (int slot = wave; slot < 8; slot += NWK) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NWK; ++w) v += red[(w * 8 + slot) * 64 + lane];
      const int m = m0 + (slot >> 2) * 16 + q4 * 4 + (slot & 3);
      if (m < p.M) {
        v = p.scale ? fmaf(v, sc, bi) : v + bi;
        v *= cs;
        if (p.residual) v += p.residual[(size_t)m * p.ldr + n];
        if (p.relu) v = (v < 0.f) ? 0.f : v;
        p.C[(size_t)m * p.ldc + n] = v;
      }
    }
    KS_STAMP(4);
    return;
  }
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wave * NB + a * TN + b) * 16 + r) * 64 + lane] = acc[a][b][r];
  __syncthreads();
  // wave w finishes accumulator rows r = w, w+NWK, ... of every block
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int n = n0 + b * 32 + l31;
      const float sc = p.scale ? p.scale[n] : 1.f;
      const float bi = p.bias ? p.bias[n] : 0.f;
      const float cs = (n < p.colscale_n) ? p.colscale : 1.f;
#pragma unroll
      for (int rr = 0; rr < 16 / NWK; ++rr) {
        const int r = wave + rr * NWK;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NWK; ++w) v += red[((w * NB + a * TN + b) * 16 + r) * 64 + lane];
        const int m = m0 + a * 32 + 4 * hh + (r & 3) + 8 * (r >> 2);
        if (m < p.M) {
          v = p.scale ? fmaf(v, sc, bi) : v + bi;
          v *= cs;
          if (p.residual) v += p.residual[(size_t)m * p.ldr + n];
          if (p.relu) v = (v < 0.f) ? 0.f : v;
          p.C[(size_t)m * p.ldc + n] = v;
        }
      }
    }
}

template <int NWK, int TM, int TN, int MODE, int DB>
__global__ __launch_bounds__(NWK * 64) void gemm_ks_kernel(const GemmParams p) {
  gemm_ks_body<NWK, TM, TN, MODE, DB>(p, blockIdx.x);
}

// two independent problems, one grid: workgroups [0, tiles0) work on p0, the rest on p1 (workgroup-uniform branch)
template <int NWK, int TM, int TN, int MODE, int DB>
__global__ __launch_bounds__(NWK * 64) void gemm_ks_dual_kernel(const GemmParams p0, const GemmParams p1, const int tiles0) {
  if ((int)blockIdx.x < tiles0) gemm_ks_body<NWK, TM, TN, MODE, DB>(p0, blockIdx.x);
  else gemm_ks_body<NWK, TM, TN, MODE, DB>(p1, (int)blockIdx.x - tiles0);
}

// ---------------------------------------------------------------------------------------------
// launch configurations; the per-shape choice comes from a table measured on the MI355X
// (tools/tune_gemm.py -> gemm_tuned.inc) with a heuristic for shapes not in the table.
// ---------------------------------------------------------------------------------------------
struct GemmCfg {
  int kind;  // 0 = spatial (WM=WN=2), 1 = k-split, 2 = k-split with two LDS stages, 3 = k-split LDS-DMA (two stages),
             // 4 = large tile, LDS-DMA, swizzled LDS, coalesced epilogue (gemm_big.hip)
  int a, tm, tn;  // spatial: a unused; k-split: a = NWK
};
static const GemmCfg kCfgs[] = {
    {0, 0, 2, 2},   // 0  spatial 128x128
    {0, 0, 2, 1},   // 1  spatial 128x64
    {0, 0, 1, 1},   // 2  spatial 64x64
    {1, 8, 1, 1},   // 3  k-split 8 waves, 32x32
    {1, 4, 1, 1},   // 4  k-split 4 waves, 32x32
    {1, 4, 2, 2},   // 5  k-split 4 waves, 64x64
    {1, 2, 2, 2},   // 6  k-split 2 waves, 64x64
    {1, 8, 2, 1},   // 7  k-split 8 waves, 64x32
    {1, 16, 1, 1},  // 8  k-split 16 waves, 32x32
    {1, 4, 2, 1},   // 9  k-split 4 waves, 64x32
    {1, 8, 1, 2},   // 10 k-split 8 waves, 32x64
    {1, 2, 1, 1},   // 11 k-split 2 waves, 32x32
    {1, 8, 2, 2},   // 12 k-split 8 waves, 64x64
    {2, 8, 1, 1},   // 13 k-split 8 waves, 32x32, double-buffered LDS
    {2, 4, 1, 1},   // 14 k-split 4 waves, 32x32, double-buffered
    {2, 4, 2, 2},   // 15 k-split 4 waves, 64x64, double-buffered
    {2, 8, 1, 2},   // 16 k-split 8 waves, 32x64, double-buffered
    {2, 4, 2, 1},   // 17 k-split 4 waves, 64x32, double-buffered
    {2, 2, 2, 2},   // 18 k-split 2 waves, 64x64, double-buffered
    {3, 8, 1, 1},   // 19 k-split 8 waves, 32x32, LDS-DMA (global_load_lds) double-buffered
    {3, 8, 1, 2},   // 20 k-split 8 waves, 32x64, LDS-DMA
    {3, 8, 2, 1},   // 21 k-split 8 waves, 64x32, LDS-DMA
    {1, 8, 1, 0},   // 22 k-split 8 waves, 32x16 (16x16x4 MFMA)
    {2, 8, 1, 0},   // 23 k-split 8 waves, 32x16, double-buffered
    {3, 8, 1, 0},   // 24 k-split 8 waves, 32x16, LDS-DMA double-buffered
    {1, 4, 1, 0},   // 25 k-split 4 waves, 32x16
    {4, 0, 2, 2},   // 26 large tile 128x128 (gemm_big.hip)
    {4, 0, 2, 1},   // 27 large tile 128x64
    {5, 0, 2, 2},   // 28 large tile 128x128, three LDS stages (two tiles of LDS-DMA in flight, counted vmcnt)
    {5, 0, 2, 1},   // 29 large tile 128x64, three LDS stages
    {6, 8, 1, 0},   // 30 k-split 8 waves, 32x16, LDS-DMA with THREE stages (two tiles in flight, counted vmcnt)
    {7, 8, 1, 0},   // 31 the same for 3x3 stride-1 convolutions over 256 channels with the input patch loaded once (DB == 4)
    // kind 8: wave-private K chunks (gemm_wp.hip): every wavefront requests and reads its own run of 32-wide K chunks through
    // its own ring of LDS slots - no barrier in the K loop, everything (or NSLOT chunks) requested up front; a = variant
    {8, 0, 1, 1},   // 32 wave-private, 8 waves, 32x32, 2 slots
    {8, 1, 1, 0},   // 33 wave-private, 8 waves, 32x16, 3 slots
    {8, 2, 1, 1},   // 34 wave-private, 4 waves, 32x32, 2 slots (two workgroups per CU)
    {8, 3, 1, 2},   // 35 wave-private, 8 waves, 32x64, 1 slot
    {8, 4, 1, 1},   // 36 wave-private, 4 waves, 32x32, 1 slot (four workgroups per CU)
    {8, 5, 2, 1},   // 37 wave-private, 8 waves, 64x32, 1 slot
    {8, 6, 1, 1},   // 38 wave-private, 4 waves, 32x32, 4 slots
    {8, 7, 1, 0},   // 39 wave-private, 8 waves, 32x16, 2 slots
    {5, 0, 2, 2},   // 40 large tile 128x128, wave-specialised: 4 loader + 4 MFMA wavefronts, three LDS stages (gemm_big.hip, gemm_ws_body)
    {5, 0, 2, 1},   // 41 large tile 128x64, wave-specialised
    // kind 9: PERSISTENT large tiles (experimental/gemm_pp.hip): one workgroup per CU walks its tiles, 4 loader wavefronts run ahead across
    // tile boundaries, two groups of 4 MFMA wavefronts alternate tiles so that a tile's epilogue runs beside the next tile's MFMAs
    {9, 0, 2, 1},   // 42 persistent ping-pong 128x64
    {9, 0, 2, 1},   // 43 persistent ping-pong 128x64, LDS-free write-out
    {5, 0, 2, 2},   // 44 large tile 128x128 (26) with the LDS-free epilogue: dword stores straight from the accumulators
    {5, 0, 2, 1},   // 45 large tile 128x64 (27) with the LDS-free epilogue
    {5, 0, 2, 2},   // 46 large tile 128x128 on PACKED SPLIT-f16 operands (experimental/gemm_h2.h): 3 f16 MFMAs per fp32 product
    {5, 0, 2, 1},   // 47 the same, 128x64
    {5, 0, 2, 2},   // 48 packed split-f16 operands on the wave-specialised 128x128 tile (4 loader + 4 MFMA wavefronts)
    {5, 0, 2, 1},   // 49 the same, 128x64
    {10, 0, 2, 2},  // 50 packed split-f16, K = 256 dense: A tile resident in REGISTERS, W through an 8-stage ring across column tiles (experimental/gemm_h2r.hip)
    {11, 0, 2, 2},  // 51 packed split-f16 on a 256 x 128 tile (8 wavefronts, three LDS stages, one workgroup per CU): 25 % fewer bytes per flop
};
static const int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);
int gemm_num_configs() { return kNumCfgs; }

template <int WM, int WN, int TM, int TN, int MODE>
static int launch_t(const GemmParams& p, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  if (p.N % BN != 0 || p.K % BK != 0 || p.M <= 0) return -1;
  GemmParams q = p;
  if (!gemm_fill_divs(q, MODE, BM, BN)) return -1;
  const int tiles = gemm_grid_tiles(q, BM, BN);
  hipLaunchKernelGGL((gemm_kernel<WM, WN, TM, TN, MODE>), dim3(tiles), dim3(256), 0, s, q);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int NWK, int TM, int TN, int DB>
static constexpr size_t ks_smem() {
  size_t rows = (size_t)TM * 32 + (TN == 0 ? 16 : TN * 32);
  size_t tile = (size_t)(DB == 3 ? 3 : DB ? 2 : 1) * rows * (NWK * BK + 4) * sizeof(float);
  if (DB == 4) tile = (size_t)(2 * 3 * 18 + 1) * (NWK * BK + 4) * sizeof(float);  // the input patch + a dummy row (weights go to registers)
  size_t red = (size_t)NWK * (TN == 0 ? 8 : TM * TN * 16) * 64 * sizeof(float);
  return tile > red ? tile : red;
}

template <int NWK, int TM, int TN, int MODE, int DB>
static int launch_ks_impl(const GemmParams& p, hipStream_t s) {
  constexpr int BM = TM * 32, BN = TN == 0 ? 16 : TN * 32;
  if (p.N % BN != 0 || p.K % BK != 0 || p.M <= 0) return -1;
  static PerDeviceFlag attr_set;  // > 64 KB of dynamic LDS needs the opt-in once per kernel and device
  constexpr size_t smem = ks_smem<NWK, TM, TN, DB>();
  if (!attr_set.get()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ks_kernel<NWK, TM, TN, MODE, DB>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return -2;
    attr_set.set();
  }
  GemmParams q = p;
  if (!gemm_fill_divs(q, MODE, BM, BN)) return -1;
  const int tiles = gemm_grid_tiles(q, BM, BN);
  hipLaunchKernelGGL((gemm_ks_kernel<NWK, TM, TN, MODE, DB>), dim3(tiles), dim3(NWK * 64), smem, s, q);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int NWK, int TM, int TN, int MODE, int DB>
static int launch_ks_dual(const GemmParams& p0, const GemmParams& p1, hipStream_t s) {
  static_assert(DB != 2, "the LDS-DMA variant is not instantiated for dual launches");
  constexpr int BM = TM * 32, BN = TN == 0 ? 16 : TN * 32;
  if (p0.N % BN != 0 || p0.K % BK != 0 || p0.M <= 0 || p1.N % BN != 0 || p1.K % BK != 0 || p1.M <= 0) return -1;
  static PerDeviceFlag attr_set;
  constexpr size_t smem = ks_smem<NWK, TM, TN, DB>();
  if (!attr_set.get()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ks_dual_kernel<NWK, TM, TN, MODE, DB>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return -2;
    attr_set.set();
  }
  GemmParams q0 = p0, q1 = p1;
  if (!gemm_fill_divs(q0, MODE, BM, BN) || !gemm_fill_divs(q1, MODE, BM, BN)) return -1;
  const int tiles0 = gemm_grid_tiles(q0, BM, BN), tiles1 = gemm_grid_tiles(q1, BM, BN);
  hipLaunchKernelGGL((gemm_ks_dual_kernel<NWK, TM, TN, MODE, DB>), dim3(tiles0 + tiles1), dim3(NWK * 64), smem, s, q0, q1, tiles0);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int NWK, int TM, int TN, int MODE, int DB = 0>
static int launch_ks(const GemmParams& p, hipStream_t s) {
  if constexpr (DB == 4 && MODE != GEMM_CONV) {
    return -1;
  } else if constexpr (DB >= 2) {
    if (p.A2 != nullptr) return -1;  // the x+pos prologue needs the register path
    if (DB == 4 && !(p.ksize == 3 && p.stride == 1 && p.Cin == 256 && (p.Wout == 16 || p.Wout % 32 == 0) && p.M % 32 == 0)) return -1;
    GemmParams q = p;
    if (q.zeros == nullptr) q.zeros = gemm_zero_buffer();
    if (q.zeros == nullptr) return -2;
    return launch_ks_impl<NWK, TM, TN, MODE, DB>(q, s);
  } else {
    return launch_ks_impl<NWK, TM, TN, MODE, DB>(p, s);
  }
}

template <int MODE>
static int launch_cfg(int cfg, const GemmParams& p, hipStream_t s) {
  switch (cfg) {
    case 0: return launch_t<2, 2, 2, 2, MODE>(p, s);
    case 1: return launch_t<2, 2, 2, 1, MODE>(p, s);
    case 2: return launch_t<2, 2, 1, 1, MODE>(p, s);
    case 3: return launch_ks<8, 1, 1, MODE>(p, s);
    case 4: return launch_ks<4, 1, 1, MODE>(p, s);
    case 5: return launch_ks<4, 2, 2, MODE>(p, s);
    case 6: return launch_ks<2, 2, 2, MODE>(p, s);
    case 7: return launch_ks<8, 2, 1, MODE>(p, s);
    case 8: return launch_ks<16, 1, 1, MODE>(p, s);
    case 9: return launch_ks<4, 2, 1, MODE>(p, s);
    case 10: return launch_ks<8, 1, 2, MODE>(p, s);
    case 11: return launch_ks<2, 1, 1, MODE>(p, s);
    case 12: return launch_ks<8, 2, 2, MODE>(p, s);
    case 13: return launch_ks<8, 1, 1, MODE, 1>(p, s);
    case 14: return launch_ks<4, 1, 1, MODE, 1>(p, s);
    case 15: return launch_ks<4, 2, 2, MODE, 1>(p, s);
    case 16: return launch_ks<8, 1, 2, MODE, 1>(p, s);
    case 17: return launch_ks<4, 2, 1, MODE, 1>(p, s);
    case 18: return launch_ks<2, 2, 2, MODE, 1>(p, s);
    case 19: return launch_ks<8, 1, 1, MODE, 2>(p, s);
    case 20: return launch_ks<8, 1, 2, MODE, 2>(p, s);
    case 21: return launch_ks<8, 2, 1, MODE, 2>(p, s);
    case 22: return launch_ks<8, 1, 0, MODE, 0>(p, s);
    case 23: return launch_ks<8, 1, 0, MODE, 1>(p, s);
    case 24: return launch_ks<8, 1, 0, MODE, 2>(p, s);
    case 25: return launch_ks<4, 1, 0, MODE, 0>(p, s);
    case 26: return launch_gemm_big(MODE, 0, p, s);
    case 27: return launch_gemm_big(MODE, 1, p, s);
    case 28: return launch_gemm_big(MODE, 2, p, s);
    case 29: return launch_gemm_big(MODE, 3, p, s);
    case 30: return launch_ks<8, 1, 0, MODE, 3>(p, s);
    case 31: return launch_ks<8, 1, 0, MODE, 4>(p, s);
    case 32: case 33: case 34: case 35: case 36: case 37: case 38: case 39: return launch_gemm_wp(MODE, kCfgs[cfg].a, p, s);
    case 42: return launch_gemm_pp(MODE, 1, p, s);
    case 43: return launch_gemm_pp(MODE, 3, p, s);
    case 44: return launch_gemm_big(MODE, 6, p, s);
    case 45: return launch_gemm_big(MODE, 7, p, s);
    case 46: return launch_gemm_big(MODE, 8, p, s);
    case 47: return launch_gemm_big(MODE, 9, p, s);
    case 48: return launch_gemm_big(MODE, 10, p, s);
    case 49: return launch_gemm_big(MODE, 11, p, s);
    case 50: return MODE == GEMM_DENSE ? launch_gemm_h2r(p, s) : -1;
    case 51: return launch_gemm_big(MODE, 12, p, s);
    case 40: return launch_gemm_big(MODE, 4, p, s);
    case 41: return launch_gemm_big(MODE, 5, p, s);
    default: return -1;
  }
}

// dual launches are instantiated for the convolution mode only, and only for the configurations the one-pair schedule of
// the bottleneck entry blocks uses (measured table): k-split 4 waves 32x32 (4), 8 waves 32x64 (10), their double-buffered
// forms (14, 16), 8 waves 32x32 (3, 13) and the large tiles (26, 27)
bool gemm_cfg_supports_dual(int cfg) {
  return cfg == 3 || cfg == 4 || cfg == 10 || cfg == 13 || cfg == 14 || cfg == 16 || cfg == 26 || cfg == 27 ||
         cfg == 32 || cfg == 34 || cfg == 35 || cfg == 36;
}

static int launch_dual_conv(int cfg, const GemmParams& p0, const GemmParams& p1, hipStream_t s) {
  switch (cfg) {
    case 3: return launch_ks_dual<8, 1, 1, GEMM_CONV, 0>(p0, p1, s);
    case 4: return launch_ks_dual<4, 1, 1, GEMM_CONV, 0>(p0, p1, s);
    case 10: return launch_ks_dual<8, 1, 2, GEMM_CONV, 0>(p0, p1, s);
    case 13: return launch_ks_dual<8, 1, 1, GEMM_CONV, 1>(p0, p1, s);
    case 14: return launch_ks_dual<4, 1, 1, GEMM_CONV, 1>(p0, p1, s);
    case 16: return launch_ks_dual<8, 1, 2, GEMM_CONV, 1>(p0, p1, s);
    case 26: return launch_gemm_big_dual(GEMM_CONV, 0, p0, p1, s);
    case 27: return launch_gemm_big_dual(GEMM_CONV, 1, p0, p1, s);
    case 32: case 34: case 35: case 36: return launch_gemm_wp_dual(GEMM_CONV, kCfgs[cfg].a, p0, p1, s);
    default: return -1;
  }
}

struct TunedEntry {
  int mode, M, N, K;
  int cfg;      // fastest configuration
  int cfg_reg;  // fastest among the register-staged ones (needed when the x+pos prologue is in use)
};
static const TunedEntry kTuned[] = {
#include "../gemm_tuned.inc"
    {-1, 0, 0, 0, 0, 0}};

static bool cfg_fits(int cfg, const GemmParams& p) {
  const GemmCfg& c = kCfgs[cfg];
  if (c.kind == 10) return p.K == 256 && p.N % 128 == 0 && p.A2 == nullptr && p.lda % 4 == 0;
  if (c.kind == 4 || c.kind == 5 || c.kind == 9 || c.kind == 11) {  // LDS-DMA operands (no x + pos prologue), float4 epilogue
    if (c.kind == 9 && p.K < 64) return false;
    if (p.A2 != nullptr || p.ldc % 4 != 0 || ((uintptr_t)p.C & 15)) return false;
    if (p.residual && (p.ldr % 4 != 0 || ((uintptr_t)p.residual & 15))) return false;
    return p.N % (64 * c.tn) == 0;
  }
  if (c.kind == 8) {  // wave-private chunks: LDS-DMA operands (no x + pos prologue); row-periodic residual supported
    int bm, bn;
    size_t lds;
    if (wp_variant_tile(c.a, &bm, &bn, &lds) != 0) return false;
    return p.A2 == nullptr && p.N % bn == 0 && p.K % BK == 0;
  }
  if (p.res_row_mod > 0) return false;  // row-periodic residual tables: large-tile and wave-private kernels only
  if (c.kind == 7)
    return p.A2 == nullptr && p.ksize == 3 && p.stride == 1 && p.Cin == 256 && (p.Wout == 16 || p.Wout % 32 == 0) && p.M % 32 == 0 && p.N % 16 == 0;
  const int bn = c.tn == 0 ? 16 : (c.kind == 0 ? 2 : 1) * c.tn * 32;
  if (c.kind != 0) {  // dynamic LDS of the k-split kernels must fit the CU's 160 KB
    const size_t rows = (size_t)c.tm * 32 + (c.tn == 0 ? 16 : c.tn * 32);
    const size_t tile = (size_t)(c.kind == 6 ? 3 : c.kind >= 2 ? 2 : 1) * rows * (c.a * BK + 4) * sizeof(float);
    const size_t red = (size_t)c.a * (c.tn == 0 ? 8 : c.tm * c.tn * 16) * 64 * sizeof(float);
    if ((tile > red ? tile : red) > 163840) return false;
    if ((c.kind == 3 || c.kind == 6) && p.A2 != nullptr) return false;  // LDS-DMA has no register prologue (x + pos)
  }
  return p.N % bn == 0;
}

// rough cost model (cycles) for shapes outside the tuned table
static double model_cost(const GemmCfg& c, const GemmParams& p) {
  if (c.kind == 8 || c.kind == 9) return 1e30;   // wave-private / persistent configurations enter through the measured table only
  if (c.kind == 4 || c.kind == 5) {  // pays off once the chip is covered several times over
    const int bm = 128, bn = 64 * c.tn;
    const double wgs = (double)((p.M + bm - 1) / bm) * (p.N / bn);
    const double rounds = ceil(wgs / 512.0);
    return rounds * ((p.K / BK) * (2 * c.tn * 16 * 64.0 * 2 + 150.0) + 3500.0) * (wgs < 1024 ? 4.0 : 1.0);
  }
  const int waves = c.kind == 0 ? 4 : c.a;
  const int bm = (c.kind == 0 ? 2 : 1) * c.tm * 32, bn = c.tn == 0 ? 16 : (c.kind == 0 ? 2 : 1) * c.tn * 32;
  const double wgs = (double)((p.M + bm - 1) / bm) * (p.N / bn);
  const int kt = p.K / BK;
  const int steps = c.kind == 0 ? kt : (kt + c.a - 1) / c.a;
  const double lds = c.kind == 0 ? (bm + bn) * 36 * 4.0 : (double)(c.kind == 6 ? 3 : c.kind >= 2 ? 2 : 1) * (bm + bn) * (c.a * 32 + 4) * 4.0;
  double per_cu = floor(163840.0 / lds);
  if (per_cu > 32.0 / waves) per_cu = 32.0 / waves;
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  const double rounds = ceil(wgs / (256.0 * per_cu));
  const double resident = wgs < 256.0 * per_cu ? ceil(wgs / 256.0) : per_cu;
  const double share = (waves * resident) / 4.0 > 1.0 ? (waves * resident) / 4.0 : 1.0;
  const double step = c.tm * (c.tn == 0 ? 0.5 : (double)c.tn) * 16 * 64.0 * share + (c.kind >= 2 ? 300.0 : 700.0);
  return rounds * (steps * step + 2500.0 + (c.kind != 0 ? 600.0 : 0.0));
}

// knobs KN_KS3: the three-stage LDS-DMA k-split (30) where the measured table says its two-stage form (24); KN_CONV_PATCH: 3x3
// stride-1 convolutions over 256 channels load their input patch once (31) where the table says 24 / 30
static int gemm_pick_config_table(int mode, const GemmParams& p);

int gemm_pick_config(int mode, const GemmParams& p) {
  const int cfg = gemm_pick_config_table(mode, p);
  // few workgroups per CU (one pair): the three-stage form hides the DMA latency the two-stage one exposes at every K step
  if (mode == GEMM_CONV && knob(KN_CONV_PATCH) && (cfg == 24 || cfg == 30) && cfg_fits(31, p)) return 31;
  // (not for the stride-2 3x3: measured 14.2 us on the two-stage form against 15.3 on the three-stage one, tools/conv_cfgs_in_situ.py)
  if (cfg == 24 && knob(KN_KS3) && p.K >= 3 * 256 && !(mode == GEMM_CONV && p.stride == 2 && p.ksize == 3) && cfg_fits(30, p)) return 30;
  return cfg;
}

static int gemm_pick_config_table(int mode, const GemmParams& p) {
  if (mode == GEMM_STEM) return 1;
  // 1. measured table, exact shape; 2. same (N, K) at the nearest measured row count (log distance): the best
  // configuration changes slowly with M; 3. cost model
  const TunedEntry* near = nullptr;
  double near_d = 0;
  for (const TunedEntry* e = kTuned; e->mode >= 0; ++e) {
    if (e->mode != mode || e->N != p.N || e->K != p.K) continue;
    if (e->M == p.M) {
      if (cfg_fits(e->cfg, p)) return e->cfg;
      if (cfg_fits(e->cfg_reg, p)) return e->cfg_reg;
    }
    if (!cfg_fits(e->cfg, p) && !cfg_fits(e->cfg_reg, p)) continue;
    const double d = fabs(log((double)e->M / (double)p.M));
    if (near == nullptr || d < near_d) {
      near = e;
      near_d = d;
    }
  }
  if (near != nullptr && near_d < 1.0) return cfg_fits(near->cfg, p) ? near->cfg : near->cfg_reg;
  int best = -1;
  double best_cost = 0;
  for (int i = 0; i < kNumCfgs && i < 42; ++i) {   // 42 and up: experimental configurations, only ever run when forced
    if (!cfg_fits(i, p)) continue;
    const double c = model_cost(kCfgs[i], p);
    if (best < 0 || c < best_cost) {
      best = i;
      best_cost = c;
    }
  }
  return best;
}

// knobs KN_CONV1X1_DENSE; KN_XCD_MAPPING bits 0-1: 0 = column tiles over XCDs always, 1 = by operand size, 2 = row tiles over XCDs always

// which operand should cross the fabric once: the one that is larger (gemm_tile_coords, common.h)
static void set_xcd_split(int mode, int cfg, GemmParams& p) {
  const GemmCfg& c = kCfgs[cfg];
  const int bm = c.kind == 11 ? 256 : (c.kind == 4 || c.kind == 5 || c.kind == 9) ? 128 : (c.kind == 0 ? 2 : 1) * c.tm * 32;
  const double a_bytes = mode == GEMM_CONV ? (double)p.M * p.stride * p.stride * p.Cin * 4.0 : (double)p.M * p.K * 4.0;
  const double w_bytes = (double)p.N * p.K * 4.0;
  const bool fits = (p.M + bm - 1) / bm >= 8;
  p.xcd_msplit = mode != GEMM_STEM && fits && ((knob(KN_XCD_MAPPING) & 3) == 2 || ((knob(KN_XCD_MAPPING) & 3) == 1 && a_bytes >= 2.0 * w_bytes));
}

int launch_gemm_dual_cfg(int mode, int cfg, const GemmParams& a, const GemmParams& b, hipStream_t s) {
  if (mode != GEMM_CONV || cfg < 0 || cfg >= kNumCfgs || !gemm_cfg_supports_dual(cfg)) return -1;
  if (!cfg_fits(cfg, a) || !cfg_fits(cfg, b)) return -1;
  GemmParams p0 = a, p1 = b;
  for (GemmParams* p : {&p0, &p1}) {
    if (p->N % 16 != 0 || p->Cin % BK != 0 || p->K != p->ksize * p->ksize * p->Cin) return -1;
    set_xcd_split(mode, cfg, *p);
  }
  return launch_dual_conv(cfg, p0, p1, s);
}

int launch_gemm_cfg(int mode, int cfg, const GemmParams& p0, hipStream_t s) {
  if (p0.N % 16 != 0 || cfg < 0 || cfg >= kNumCfgs || !cfg_fits(cfg, p0)) return -1;
  GemmParams p = p0;
  set_xcd_split(mode, cfg, p);
  if (p.A2 != nullptr && p.zeros == nullptr) {   // the x + pos prologue reads a zero row where a tile does not use A2
    p.zeros = gemm_zero_buffer();
    if (p.zeros == nullptr) return -2;
  }
  switch (mode) {
    case GEMM_DENSE:
      if (p.lda % 4 != 0 || (p.A2 && p.lda2 % 4 != 0)) return -1;
      return launch_cfg<GEMM_DENSE>(cfg, p, s);
    case GEMM_CONV:
      if (p.Cin % BK != 0 || p.K != p.ksize * p.ksize * p.Cin) return -1;
      // a 1x1 stride-1 convolution IS the dense product of the pixel rows (row m = pixel m, lda = Cin, no padding): the
      // dense instantiation of the same configuration computes the same sums in the same order without the per-row pixel
      // decomposition - ~640 fewer instructions between workgroup entry and the first load (profiles/r3_prologue_*.txt)
      if (knob(KN_CONV1X1_DENSE) && p.ksize == 1 && p.stride == 1 && p.pad == 0 && p.lda == p.Cin && p.lda % 4 == 0)
        return launch_cfg<GEMM_DENSE>(cfg, p, s);
      return launch_cfg<GEMM_CONV>(cfg, p, s);
    case GEMM_STEM:
      if (p.N != 64 || p.K != 160) return -1;
      return launch_t<2, 2, 2, 1, GEMM_STEM>(p, s);  // BM must be 128
    default:
      return -1;
  }
}

thread_local int cotr_tls_device = -1;

const float* gemm_zero_buffer() {
  static float* z[COTR_MAX_DEVICES] = {};  // one per device, on that device
  float*& zd = z[cotr_current_device()];
  if (zd == nullptr) {
    float* p = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&p), 4096) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 4096) != hipSuccess) { (void)hipFree(p); return nullptr; }
    zd = p;
  }
  return zd;
}

int launch_gemm(int mode, const GemmParams& p, hipStream_t s) {
  if (linear_rows_applies(mode, p)) {   // K-short products over many rows: the A tile resident, the weights streamed (linear_rows.hip)
    const int r = launch_linear_rows(p, s);
    if (r != -1) return r;
  }
  const int cfg = gemm_pick_config(mode, p);
  if (cfg < 0) return -1;
  return launch_gemm_cfg(mode, cfg, p, s);
}
