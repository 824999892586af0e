// "Wave-private" k-split fp32 MFMA GEMM / implicit-GEMM convolution for the latency-bound regime (one image pair: M = 512 .. 8192
// rows, ~1000 query rows), gfx950.  Same contract and fused epilogue as gemm.hip (GemmParams).
//
// What the phase stamps of the k-split kernels (gemm.hip) show at one pair (profiles/r3_conv_phases_after_fastdiv.txt): after
// the prologue the time is the K loop, and the K loop is round trips - a tile of the operands is requested, ~1 us later it is
// usable, a barrier, 0.2 us of MFMAs, the next request.  The bytes a CU can pull per microsecond are (bytes in flight) / (L2 /
// Infinity-Cache latency), and those kernels keep one or two K steps in flight because their wavefronts load ROWS of a shared
// tile: every step needs a barrier, every stage holds all wavefronts' K slices.
//
// Here the K dimension is cut into 32-wide chunks and every wavefront OWNS a contiguous run of chunks: it requests them itself
// (LDS-DMA, global_load_lds_dwordx4: one instruction = 8 rows x 128 B into 1 KB of LDS, 16-byte chunks XOR-swizzled through the
// source address as in gemm_big.hip, so the fragment reads are conflict-free without padding), into its own ring of NSLOT LDS
// slots, and reads them back itself.  Consequences:
//   * no barrier anywhere in the K loop - a wavefront's own `s_waitcnt vmcnt(N)` orders its ds_reads behind its DMA;
//   * everything a wavefront needs (up to NSLOT chunks) is requested in its first instructions - one memory round trip for the
//     whole contraction where the operands fit the CU's 160 KB (most one-pair shapes), a ring refilled chunk by chunk where not;
//   * a wavefront starts its MFMAs when ITS first chunk has landed, not when the slowest wavefront's has.
// The NWK partial accumulators are summed through LDS in a fixed order (deterministic; each wavefront parks its partial in its
// own slot area, one barrier) and the fused epilogue of gemm.hip is applied.  Chunks are dealt to wavefronts in contiguous,
// balanced runs (the first K/32 mod NWK wavefronts get one more), so the summation order is a function of (K, NWK) only.
#include "common.h"

#define BK 32

template <int N>
__device__ __forceinline__ void wp_wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most `chunks` x Q of this wavefront's DMA instructions are outstanding
template <int Q, int MAXC>
__device__ __forceinline__ void wp_wait_chunks(int chunks) {
  if constexpr (MAXC == 0) {
    wp_wait_vmcnt<0>();
  } else {
    if (chunks >= MAXC) wp_wait_vmcnt<(Q * MAXC <= 63 ? Q * MAXC : 63)>();
    else wp_wait_chunks<Q, MAXC - 1>(chunks);
  }
}

// TN == 0 selects the 32 x 16 tile on v_mfma_f32_16x16x4_f32 (two 16-row blocks x one 16-column block)
template <int NWK, int TM, int TN, int MODE, int NSLOT>
__device__ __forceinline__ void gemm_wp_body(const GemmParams& p, const int bid) {
  constexpr bool N16 = (TN == 0);
  static_assert(!N16 || TM == 1, "the 16-column tile is 32 rows tall");
  constexpr int BM = TM * 32, BN = N16 ? 16 : TN * 32;
  constexpr int ROWS = BM + BN;
  constexpr int QA = BM / 8, QW = BN / 8, Q = QA + QW;   // DMA instructions per chunk and wavefront
  constexpr int SLOT = ROWS * BK;                        // floats per slot
  constexpr int NB = N16 ? 1 : TM * TN;
  static_assert(Q * (NSLOT - 1) <= 63, "counted vmcnt");
  static_assert(SLOT >= (N16 ? 8 : NB * 16) * 64, "a wavefront parks its partial accumulators in its own first slot");
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  int m0, n0;
  if (!gemm_tile_coords(p, BM, BN, bid, m0, n0)) return;
  const int NC = p.K / BK;
  const int base = NC / NWK, rem = NC - base * NWK;
  const int c_begin = wave * base + (wave < rem ? wave : rem);   // first chunk of this wavefront
  const int n_my = base + (wave < rem ? 1 : 0);
  float* slots = smem + wave * (NSLOT * SLOT);

  // ---- LDS-DMA bookkeeping: lane -> (row lane>>3 of the instruction's 8 rows, physical 16-B chunk lane&7) ----
  const int drow = lane >> 3, pch = lane & 7;
  const float* a_ptr[QA];
  bool a_ok[QA];
  int c_hi0[QA], c_wi0[QA];
#pragma unroll
  for (int q = 0; q < QA; ++q) {
    const int row = q * 8 + drow;                        // tile-local A row = LDS row
    const int lch = pch ^ ((row >> 1) & 7);              // logical chunk this lane fetches
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
    const int row = q * 8 + drow;                        // tile-local W row; LDS row BM + row (BM is a multiple of 32: same swizzle)
    const int lch = pch ^ ((row >> 1) & 7);
    w_ptr[q] = p.W + (size_t)(n0 + row) * p.K + lch * 4;
  }

  auto dma_chunk = [&](int c, int slot) {                // chunk c of this wavefront's run -> its slot `slot`
    const int kt = c_begin + c;
    float* S = slots + slot * SLOT;
    if constexpr (MODE == GEMM_DENSE) {
#pragma unroll
      for (int q = 0; q < QA; ++q) {
        const float* src = a_ok[q] ? a_ptr[q] + kt * BK : p.zeros;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(S + q * 8 * BK), 16, 0, 0);
      }
    } else {
      int ky, kx, c0;
      conv_ktile_decompose(p, kt, ky, kx, c0);
      const int tapoff = (ky * (2 * p.Win) + kx) * p.Cin + c0;   // wave-uniform element offset of this tap / channel tile
#pragma unroll
      for (int q = 0; q < QA; ++q) {
        const int hi = c_hi0[q] + ky, wi = c_wi0[q] + kx;
        const bool ok = a_ok[q] && hi >= 0 && hi < p.Hin && wi >= 0 && wi < p.Win;
        const float* src = ok ? a_ptr[q] + tapoff : p.zeros;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(S + q * 8 * BK), 16, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < QW; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_ptr[q] + kt * BK),
                                       (__attribute__((address_space(3))) void*)(S + (BM + q * 8) * BK), 16, 0, 0);
  };

#define WP_STAMP(slot_)                                                                 \
  do {                                                                                  \
    if (p.dbg != nullptr && t == 0) p.dbg[(size_t)bid * 8 + (slot_)] = wall_clock64();  \
  } while (0)
  WP_STAMP(0);
  const int pre = n_my < NSLOT ? n_my : NSLOT;
  for (int c = 0; c < pre; ++c) dma_chunk(c, c);
  WP_STAMP(1);

  const int l31 = lane & 31, hh = lane >> 5;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int sw = N16 ? ((l15 >> 1) & 7) : ((l31 >> 1) & 7);   // swizzle of this lane's fragment rows (row bases: multiples of 16 / 32)
  f32x16 acc[TM][N16 ? 1 : TN];
  f32x4 acc16[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < (N16 ? 1 : TN); ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  int slot = 0;
  for (int c = 0; c < n_my; ++c) {
    // requested after chunk c, still allowed in flight: chunks c+1 .. min(n_my, c + NSLOT) - 1
    const int last = (c + NSLOT < n_my ? c + NSLOT : n_my) - 1;
    wp_wait_chunks<Q, NSLOT - 1>(last - c);
    if (c == 0) WP_STAMP(2);
    const float* S = slots + slot * SLOT;
    if constexpr (N16) {
      // all fragment reads of the chunk first: the slot is free for its refill as soon as they have returned, and the 16 MFMAs
      // that follow have nothing to wait for
      f32x4 a0[2], a1[2], bb[2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int ch = ((jj * 4 + q4) ^ sw) * 4;
        a0[jj] = *reinterpret_cast<const f32x4*>(S + l15 * BK + ch);
        a1[jj] = *reinterpret_cast<const f32x4*>(S + (16 + l15) * BK + ch);
        bb[jj] = *reinterpret_cast<const f32x4*>(S + (BM + l15) * BK + ch);
      }
      if (c + NSLOT < n_my) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dma_chunk(c + NSLOT, slot);
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc16[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[jj][e], bb[jj][e], acc16[0], 0, 0, 0);
          acc16[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[jj][e], bb[jj][e], acc16[1], 0, 0, 0);
        }
    } else {
      constexpr int TNN = N16 ? 1 : TN;
      f32x4 af[4][TM], bf[4][TNN];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ch = ((j * 2 + hh) ^ sw) * 4;
#pragma unroll
        for (int a = 0; a < TM; ++a) af[j][a] = *reinterpret_cast<const f32x4*>(S + (a * 32 + l31) * BK + ch);
#pragma unroll
        for (int b = 0; b < TNN; ++b) bf[j][b] = *reinterpret_cast<const f32x4*>(S + (BM + b * 32 + l31) * BK + ch);
      }
      if (c + NSLOT < n_my) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this chunk's fragment reads have returned: its slot may be overwritten
        dma_chunk(c + NSLOT, slot);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TNN; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][a][e], bf[j][b][e], acc[a][b], 0, 0, 0);
    }
    slot = slot + 1 == NSLOT ? 0 : slot + 1;
  }
  WP_STAMP(3);

  // ---- cross-wave reduction: every wavefront parks its partial accumulators at the start of its own slot area (its reads of
  // that area are retired: lgkmcnt(0), and nothing of its DMA is in flight), one barrier, fixed summation order --------------
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  constexpr int WSTRIDE = NSLOT * SLOT;                   // floats between two wavefronts' areas
  if constexpr (N16) {
    // D of 16x16x4: column = lane&15, row = (lane>>4)*4 + reg; 2 blocks x 4 regs = 8 slots per lane
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int r = 0; r < 4; ++r) slots[(blk * 4 + r) * 64 + lane] = acc16[blk][r];
    __syncthreads();
    const int n = n0 + l15;
    const float sc = p.scale ? p.scale[n] : 1.f;
    const float bi = p.bias ? p.bias[n] : 0.f;
    const float cs = (n < p.colscale_n) ? p.colscale : 1.f;
    for (int sl = wave; sl < 8; sl += NWK) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NWK; ++w) v += smem[w * WSTRIDE + sl * 64 + lane];
      const int m = m0 + (sl >> 2) * 16 + q4 * 4 + (sl & 3);
      if (m < p.M) {
        v = p.scale ? fmaf(v, sc, bi) : v + bi;
        v *= cs;
        if (p.residual) v += p.residual[(size_t)(p.res_row_mod > 0 ? fastmod(m, p.fd_resrow) : m) * p.ldr + n];
        if (p.relu) v = (v < 0.f) ? 0.f : v;  // NaN passes through like torch.relu (fmaxf would drop it)
        p.C[(size_t)m * p.ldc + n] = v;
      }
    }
    WP_STAMP(4);
    return;
  } else {
    constexpr int TNN = N16 ? 1 : TN;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TNN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) slots[((a * TNN + b) * 16 + r) * 64 + lane] = acc[a][b][r];
    __syncthreads();
    // wave w finishes accumulator rows r = w, w+NWK, ... of every block
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TNN; ++b) {
        const int n = n0 + b * 32 + l31;
        const float sc = p.scale ? p.scale[n] : 1.f;
        const float bi = p.bias ? p.bias[n] : 0.f;
        const float cs = (n < p.colscale_n) ? p.colscale : 1.f;
        for (int r = wave; r < 16; r += NWK) {
          float v = 0.f;
#pragma unroll
          for (int w = 0; w < NWK; ++w) v += smem[w * WSTRIDE + ((a * TNN + b) * 16 + r) * 64 + lane];
          const int m = m0 + a * 32 + 4 * hh + (r & 3) + 8 * (r >> 2);
          if (m < p.M) {
            v = p.scale ? fmaf(v, sc, bi) : v + bi;
            v *= cs;
            if (p.residual) v += p.residual[(size_t)(p.res_row_mod > 0 ? fastmod(m, p.fd_resrow) : m) * p.ldr + n];
            if (p.relu) v = (v < 0.f) ? 0.f : v;
            p.C[(size_t)m * p.ldc + n] = v;
          }
        }
      }
    WP_STAMP(4);
  }
#undef WP_STAMP
}

template <int NWK, int TM, int TN, int MODE, int NSLOT>
__global__ __launch_bounds__(NWK * 64) void gemm_wp_kernel(const GemmParams p) {
  gemm_wp_body<NWK, TM, TN, MODE, NSLOT>(p, blockIdx.x);
}

// two independent problems, one grid (the entry blocks of the ResNet stages): workgroups [0, tiles0) work on p0, the rest on p1
template <int NWK, int TM, int TN, int MODE, int NSLOT>
__global__ __launch_bounds__(NWK * 64) void gemm_wp_dual_kernel(const GemmParams p0, const GemmParams p1, const int tiles0) {
  if ((int)blockIdx.x < tiles0) gemm_wp_body<NWK, TM, TN, MODE, NSLOT>(p0, blockIdx.x);
  else gemm_wp_body<NWK, TM, TN, MODE, NSLOT>(p1, (int)blockIdx.x - tiles0);
}

template <int NWK, int TM, int TN, int NSLOT>
static constexpr size_t wp_smem() {
  return (size_t)NWK * NSLOT * (TM * 32 + (TN == 0 ? 16 : TN * 32)) * BK * sizeof(float);
}

static bool wp_prepare(GemmParams& p) {
  if (p.K % BK != 0 || p.M <= 0 || p.A2 != nullptr) return false;
  if (p.zeros == nullptr) p.zeros = gemm_zero_buffer();
  return p.zeros != nullptr;
}

template <int NWK, int TM, int TN, int MODE, int NSLOT>
static int launch_wp_t(const GemmParams& p0, hipStream_t s) {
  constexpr int BM = TM * 32, BN = TN == 0 ? 16 : TN * 32;
  constexpr size_t smem = wp_smem<NWK, TM, TN, NSLOT>();
  static_assert(smem <= 163840, "LDS");
  GemmParams p = p0;
  if (p.N % BN != 0 || p.K % BK != 0 || p.M <= 0 || p.A2 != nullptr) return -1;
  if (!wp_prepare(p)) return -2;
  if (!gemm_fill_divs(p, MODE, BM, BN)) return -1;
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wp_kernel<NWK, TM, TN, MODE, NSLOT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return -2;
    attr_set.set();
  }
  const int tiles = gemm_grid_tiles(p, BM, BN);
  hipLaunchKernelGGL((gemm_wp_kernel<NWK, TM, TN, MODE, NSLOT>), dim3(tiles), dim3(NWK * 64), smem, s, p);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int NWK, int TM, int TN, int NSLOT>
static int launch_wp_dual_t(const GemmParams& a, const GemmParams& b, hipStream_t s) {
  constexpr int BM = TM * 32, BN = TN == 0 ? 16 : TN * 32;
  constexpr size_t smem = wp_smem<NWK, TM, TN, NSLOT>();
  GemmParams p0 = a, p1 = b;
  for (GemmParams* p : {&p0, &p1}) {
    if (p->N % BN != 0 || p->K % BK != 0 || p->M <= 0 || p->A2 != nullptr) return -1;
    if (!wp_prepare(*p)) return -2;
    if (!gemm_fill_divs(*p, GEMM_CONV, BM, BN)) return -1;
  }
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wp_dual_kernel<NWK, TM, TN, GEMM_CONV, NSLOT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return -2;
    attr_set.set();
  }
  const int tiles0 = gemm_grid_tiles(p0, BM, BN), tiles1 = gemm_grid_tiles(p1, BM, BN);
  hipLaunchKernelGGL((gemm_wp_dual_kernel<NWK, TM, TN, GEMM_CONV, NSLOT>), dim3(tiles0 + tiles1), dim3(NWK * 64), smem, s, p0, p1,
                     tiles0);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// variants (gemm.hip, kCfgs kind 8): wavefronts x tile x ring slots
//   0: 8 x 32x32 x 2 (128 KB)   1: 8 x 32x16 x 3 (144 KB)   2: 4 x 32x32 x 2 (64 KB, two workgroups per CU)
//   3: 8 x 32x64 x 1 (96 KB)    4: 4 x 32x32 x 1 (32 KB)    5: 8 x 64x32 x 1 (96 KB)    6: 4 x 32x32 x 4 (128 KB)
//   7: 8 x 32x16 x 2 (96 KB)
int wp_variant_tile(int variant, int* bm, int* bn, size_t* lds) {
  switch (variant) {
    case 0: *bm = 32; *bn = 32; *lds = wp_smem<8, 1, 1, 2>(); return 0;
    case 1: *bm = 32; *bn = 16; *lds = wp_smem<8, 1, 0, 3>(); return 0;
    case 2: *bm = 32; *bn = 32; *lds = wp_smem<4, 1, 1, 2>(); return 0;
    case 3: *bm = 32; *bn = 64; *lds = wp_smem<8, 1, 2, 1>(); return 0;
    case 4: *bm = 32; *bn = 32; *lds = wp_smem<4, 1, 1, 1>(); return 0;
    case 5: *bm = 64; *bn = 32; *lds = wp_smem<8, 2, 1, 1>(); return 0;
    case 6: *bm = 32; *bn = 32; *lds = wp_smem<4, 1, 1, 4>(); return 0;
    case 7: *bm = 32; *bn = 16; *lds = wp_smem<8, 1, 0, 2>(); return 0;
    default: return -1;
  }
}

template <int MODE>
static int launch_wp_mode(int variant, const GemmParams& p, hipStream_t s) {
  switch (variant) {
    case 0: return launch_wp_t<8, 1, 1, MODE, 2>(p, s);
    case 1: return launch_wp_t<8, 1, 0, MODE, 3>(p, s);
    case 2: return launch_wp_t<4, 1, 1, MODE, 2>(p, s);
    case 3: return launch_wp_t<8, 1, 2, MODE, 1>(p, s);
    case 4: return launch_wp_t<4, 1, 1, MODE, 1>(p, s);
    case 5: return launch_wp_t<8, 2, 1, MODE, 1>(p, s);
    case 6: return launch_wp_t<4, 1, 1, MODE, 4>(p, s);
    case 7: return launch_wp_t<8, 1, 0, MODE, 2>(p, s);
    default: return -1;
  }
}

int launch_gemm_wp(int mode, int variant, const GemmParams& p, hipStream_t s) {
  if (mode == GEMM_DENSE) return p.lda % 4 != 0 ? -1 : launch_wp_mode<GEMM_DENSE>(variant, p, s);
  if (mode == GEMM_CONV) return launch_wp_mode<GEMM_CONV>(variant, p, s);
  return -1;
}

int launch_gemm_wp_dual(int mode, int variant, const GemmParams& p0, const GemmParams& p1, hipStream_t s) {
  if (mode != GEMM_CONV) return -1;
  switch (variant) {
    case 0: return launch_wp_dual_t<8, 1, 1, 2>(p0, p1, s);
    case 2: return launch_wp_dual_t<4, 1, 1, 2>(p0, p1, s);
    case 3: return launch_wp_dual_t<8, 1, 2, 1>(p0, p1, s);
    case 4: return launch_wp_dual_t<4, 1, 1, 1>(p0, p1, s);
    default: return -1;
  }
}
