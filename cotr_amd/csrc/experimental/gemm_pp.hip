// EXPERIMENTAL (libcotr_hip_exp.so only; measured SLOWER than the kernels it was meant to replace - numbers at the end of this comment).
// Persistent "ping-pong" form of the large-tile fp32 MFMA GEMM / implicit-GEMM convolution (gemm_big.hip) for the batched regime.
// Same contract (GemmParams), same tile decomposition, same k order per accumulator and same epilogue arithmetic as configurations
// 26 / 27 (128 x 128 / 128 x 64 tiles): bit-identical results.  What changes is what happens BETWEEN tiles.
//
// Why (round 4, tools/tile_overhead.py on the MI355X): the one-tile-per-workgroup kernels run their K loop at 0.79-0.87 of the MFMA
// peak, but every tile pays a fixed 2.4-5.4 us on top (first operands' latency, accumulator hand-over, the epilogue's stores) -
// 2.8-4.5 K steps of a 128 x 128 tile.  At K = 4096 that is 3 %; at K = 256 (every transformer projection, every 1x1 convolution of
// layer1-3's expansions: a third of the batched forward) it is 25-35 %: 16384 x 1024 x 256 runs at 0.62, 4096^3 at 0.85.
//
// Here ONE workgroup per CU walks tiles tile = blockIdx.x, + gridDim.x, ... and never drains its pipeline:
//   * 4 LOADER wavefronts only issue LDS-DMA (global_load_lds) into a ring of NSTG stages; their request stream runs NSTG - 1
//     K steps ahead of the consumers and simply continues into the next tile's operands at a tile boundary;
//   * two groups of 4 MFMA wavefronts (A, B) take tiles alternately: while A runs the K loop of tile j, B writes out tile j-1
//     (accumulators -> wave-private LDS staging -> float4 stores, FrozenBN / bias / residual / ReLU as in gemm_big_body), spread
//     over the first two K steps of A's loop, and is ready with zeroed accumulators when A finishes; then they swap.  A SIMD
//     holds one wavefront of each kind; one MFMA wavefront per SIMD saturates the matrix pipe (tools/micro/mfma_lds.hip:
//     64.0 cycles per MFMA), so the pipe always has exactly one feeder and the epilogue, the prologue and the address
//     arithmetic of the next tile all happen beside it.
//   * one s_barrier per K step for all 12 wavefronts: loaders have waited (counted vmcnt) for step g's data, consumers have
//     retired their reads of step g-1 (whose stage the loaders refill right after the barrier).  The idle group hits the same
//     barriers, so nothing is ever signalled through memory.
// LDS: TN = 2: 3 stages x 32 KB + 34 KB staging = 131 KB; TN = 1: 4 x 24 + 18 = 114 KB (one workgroup per CU).
//
// MEASURED (MI355X, round 4, tools/tile_overhead.py; profiles/r4_persistent_pingpong_gemm.txt), 262144 x 256 x K, 128 x 128 tiles:
//   K = 256: configuration 26 311 us (110 TFLOP/s); this kernel with NO write-out 261 us (131 TFLOP/s: the pipeline across tiles
//   works - fixed cost per tile 0.75 us instead of 2.4-5.4); with the write-out 375 us.  The write-out is NOT hidden: its staging
//   round trips run at 30-60 cycles per instruction in a low-priority wavefront beside an MFMA wavefront, so the writers reach every
//   barrier after the MFMA wavefronts (items loop +61 us, stores +41 us, staging writes +5 us of the 16 tiles per CU).  Issuing a
//   piece's reads together (this version) recovers a third of that for the 128 x 64 tile (423 -> 383 us, against 306 us for
//   configuration 27); an LDS-free write-out (dword stores from the accumulators, unrolled) spills.  Bit-identical to 26 / 27
//   throughout (tests/test_experimental_gpu.py).  What it would take: a write-out with no LDS round trip and no per-step barrier
//   coupling to the MFMA wavefronts - i.e. split barriers this ISA does not have.
#include "../common.h"

#define BK 32

template <int TN, int MODE, bool DIRECT>
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3))) void gemm_pp_kernel(const GemmParams p, const int ntiles) {
  constexpr int BM = 128, BN = 64 * TN;
  constexpr int STAGE = (BM + BN) * BK;   // floats per ring stage (one 32-deep K step of a tile)
  constexpr int QW = BN / 32;             // W DMA instructions per loader wavefront and step
  constexpr int PER = 4 + QW;             // DMA instructions per loader wavefront and step
  constexpr int EP = 32 * TN + 4;         // padded row of the epilogue staging tile
  constexpr int NSTG = TN == 2 ? 3 : 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const estage = smem + NSTG * STAGE;   // 4 wave slots of 32 x EP floats (the group that is writing a tile out)

  const int t = threadIdx.x, lane = t & 63;
  const int wave12 = __builtin_amdgcn_readfirstlane(t >> 6);
  const int KT = p.K / BK;
  const int G = gridDim.x, bid = blockIdx.x;

  // ---- the tiles of this workgroup: ids bid, bid + G, ... < ntiles that the XCD-aware mapping (gemm_tile_coords) declares valid ----
  auto tile_coords = [&](int id, int& m0, int& n0) -> bool { return id < ntiles && gemm_tile_coords(p, BM, BN, id, m0, n0); };
  int nvalid = 0;
  for (int id = bid; id < ntiles; id += G) {
    int a, b;
    nvalid += tile_coords(id, a, b) ? 1 : 0;
  }
  if (nvalid == 0) return;
  const int total = nvalid * KT;          // barrier-synchronous steps of this workgroup

  if (wave12 >= 8) {
    // ===================================== loaders ==========================================================================
    const int wave = wave12 - 8;          // which quarter of the tile rows it fetches
    const int drow = lane >> 3, pch = lane & 7;
    const float* a_ptr[4];
    bool a_ok[4];
    int c_hi0[4], c_wi0[4];
    const float* w_ptr[QW];
    int next_id = bid;                    // next tile id to look at
    auto setup_next_tile = [&]() {        // advances to the next valid tile and computes this lane's source pointers
      int m0 = 0, n0 = 0;
      while (!tile_coords(next_id, m0, n0)) next_id += G;    // (only called while a valid tile remains)
      next_id += G;
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
          a_ptr[q] = p.A + (long)(((b * p.Hin + c_hi0[q]) * (2 * p.Win) + side * p.Win + c_wi0[q]) * p.Cin) + lch * 4;
        }
      }
#pragma unroll
      for (int q = 0; q < QW; ++q) {
        const int row = wave * (BN / 4) + q * 8 + drow;
        const int lch = pch ^ ((row >> 1) & 7);
        w_ptr[q] = p.W + (size_t)(n0 + row) * p.K + lch * 4;
      }
    };
    int iss_kt = 0, iss_stage = 0, issued = 0;       // the request stream: K step inside its tile, ring stage, steps requested so far
    auto issue_step = [&]() {
      if (iss_kt == 0) setup_next_tile();
      float* As = smem + iss_stage * STAGE;
      float* Ws = As + BM * BK;
      if constexpr (MODE == GEMM_DENSE) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float* src = a_ok[q] ? a_ptr[q] + iss_kt * BK : p.zeros;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(As + (wave * 32 + q * 8) * BK), 16, 0, 0);
        }
      } else {
        int ky, kx, c0;
        conv_ktile_decompose(p, iss_kt, ky, kx, c0);
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
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_ptr[q] + iss_kt * BK),
                                         (__attribute__((address_space(3))) void*)(Ws + (wave * (BN / 4) + q * 8) * BK), 16, 0, 0);
      iss_kt = iss_kt + 1 == KT ? 0 : iss_kt + 1;
      iss_stage = iss_stage + 1 == NSTG ? 0 : iss_stage + 1;
      ++issued;
    };
    __builtin_amdgcn_s_setprio(3);                   // the loaders' few instructions go out ahead of everybody else's
    for (int i = 0; i < NSTG - 1 && i < total; ++i) issue_step();
    for (int g = 0; g < total; ++g) {
      // before barrier g this wavefront's share of step g must have landed; the (issued - 1 - g) younger steps stay in flight
      const int younger = issued - 1 - g;
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                  // step g is in LDS for everybody; the readers of step g-1 are done with its stage
      if (issued < total) issue_step();              // ... which this request refills (stage (g + NSTG - 1) % NSTG)
    }
    __builtin_amdgcn_s_barrier();                    // (matches the consumers' barrier behind the last step)
    return;
  }

  // ===================================== MFMA wavefronts (groups A = 0, B = 1) ================================================
  const int group = wave12 >> 2, wave = wave12 & 3;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hh = lane >> 5;
  const int sw = (l31 >> 1) & 7;
  constexpr int C4 = 8 * TN;              // float4 per staged epilogue row
  constexpr int RPI = 64 / C4;            // rows per wave instruction
  constexpr int NIT = 32 / RPI;
  const int er = lane / C4, ec = (lane % C4) * 4;
  float* const Es = estage + wave * 32 * EP;
  f32x16 acc[2][TN];
  // The finished tile is written out as 2 * NIT ITEMS (one wave store instruction each: RPI rows x 32 TN columns) spread EVENLY over
  // the K steps of the other group's tile - not in one burst: with every CU in the same phase a burst is 16 MB of stores chip-wide
  // in two K steps (measured: the first version, which wrote a 32-row block per step, had a LARGER fixed cost per tile than the
  // one-tile-per-workgroup kernel).  The residual row of item i+1 is requested while item i is processed (4 registers).
  // (2 * NIT items per tile: RPI rows x 32 TN columns each)
  f32x4 sc, bi, cs;
  int pend_m0 = 0, pend_n0 = 0;           // the tile this group still has to write out
  bool pending = false;
  int item_acc = 0, item_next = 0;        // the write-out schedule of the pending tile

  // Items are processed in PIECES of PIECE consecutive items of one 32-row block with compile-time structure: all staging reads of
  // the piece are issued together and waited for once, then the arithmetic, then the stores.  (Item by item - read, wait, compute,
  // store - the write-out is a chain of LDS round trips that a low-priority wavefront beside an MFMA wavefront gets through at
  // 30-60 cycles per instruction: 16 items took as long as the other group's whole 8-step K loop, and every step ended with the MFMA
  // wavefronts waiting at the barrier for the writers - tools/tile_overhead.py with the ws_flags experiment bits.)
  constexpr int PIECE = NIT / 2;            // items per piece: 4 (TN = 2) / 2 (TN = 1); 4 pieces per tile
  auto write_piece = [&](int piece, int m0, int n0) {
    const int ncol = n0 + wn * 32 * TN + ec;
    if (piece == 0) {                                           // per-column epilogue constants, once per tile
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sc[e] = p.scale ? p.scale[ncol + e] : 1.f;
        bi[e] = p.bias ? p.bias[ncol + e] : 0.f;
        cs[e] = (ncol + e < p.colscale_n) ? p.colscale : 1.f;
      }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      if ((piece >> 1) != a) continue;
      const int half = piece & 1;
      if (half == 0) {                                          // first piece of block a: its accumulators go to the staging tile
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) Es[((r & 3) + 8 * (r >> 2) + 4 * hh) * EP + b * 32 + l31] = acc[a][b][r];
      }
      f32x4 v[PIECE], rr[PIECE];
#pragma unroll
      for (int i = 0; i < PIECE; ++i) {
        const int row = (half * PIECE + i) * RPI + er;
        v[i] = *reinterpret_cast<const f32x4*>(Es + row * EP + ec);
        const int m = m0 + wm * 64 + a * 32 + row;
        rr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.residual && m < p.M) rr[i] = *reinterpret_cast<const f32x4*>(p.residual + (size_t)(p.res_row_mod > 0 ? fastmod(m, p.fd_resrow) : m) * p.ldr + ncol);
      }
#pragma unroll
      for (int i = 0; i < PIECE; ++i) {
        const int row = (half * PIECE + i) * RPI + er;
        const int m = m0 + wm * 64 + a * 32 + row;
        if (m < p.M) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = v[i][e];
            x = p.scale ? fmaf(x, sc[e], bi[e]) : x + bi[e];
            x *= cs[e];
            if (p.residual) x += rr[i][e];
            if (p.relu) x = (x < 0.f) ? 0.f : x;
            v[i][e] = x;
          }
          *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + ncol) = v[i];
        }
      }
    }
  };
  auto write_items = [&](int lo, int hi, int m0, int n0) {      // pieces lo .. hi-1 of the pending tile (4 per tile)
    for (int piece = lo; piece < hi; ++piece) write_piece(piece, m0, n0);
  };
  // LDS-free form of the same write-out (ws_flags bit 6): an item = 4 accumulator registers (4 consecutive rows) of one 32 x 32
  // block, stored straight from the registers - one wave store = two full 128-B lines (rows r and r + 4) - with the epilogue
  // arithmetic applied in place.  No staging writes, no reads, no loop: only VALU + VMEM instructions, which a neighbour
  // wavefront issues beside an MFMA wavefront at no cost to it (tools/micro/mfma_beside.hip), where LDS writes and the scalar
  // bookkeeping of a loop were seen to wait until the MFMA wavefront pauses.
  constexpr int DITEMS = 2 * TN * 4;
  auto write_direct = [&](int lo, int hi, int m0, int n0) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int col = n0 + wn * 32 * TN + b * 32 + l31;
        float scv = 1.f, biv = 0.f, csv = 1.f;
        const int first = (a * TN + b) * 4;
        if (first >= hi || first + 4 <= lo) continue;
        scv = p.scale ? p.scale[col] : 1.f;
        biv = p.bias ? p.bias[col] : 0.f;
        csv = col < p.colscale_n ? p.colscale : 1.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int item = first + q;
          if (item < lo || item >= hi) continue;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + a * 32 + 8 * q + i + 4 * hh;
            if (m < p.M) {
              float x = acc[a][b][q * 4 + i];
              x = p.scale ? fmaf(x, scv, biv) : x + biv;
              x *= csv;
              if (p.residual) x += p.residual[(size_t)(p.res_row_mod > 0 ? fastmod(m, p.fd_resrow) : m) * p.ldr + col];
              if (p.relu) x = (x < 0.f) ? 0.f : x;
              p.C[(size_t)m * p.ldc + col] = x;
            }
          }
        }
      }
  };
  constexpr bool direct = DIRECT;
  constexpr int n_items = direct ? DITEMS : 4;
  struct Frag {
    f32x4 a[2], b[TN];
  };

  int stage = 0, seq = 0;                  // ring stage of the current step; index of the current tile among the valid ones
  for (int id = bid; id < ntiles; id += G) {
    int m0, n0;
    if (!tile_coords(id, m0, n0)) continue;
    const bool mine = (seq & 1) == group;
    ++seq;
    if (mine) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
      __builtin_amdgcn_s_setprio(2);
    }
    for (int kt = 0; kt < KT; ++kt) {
      __builtin_amdgcn_s_barrier();        // the loaders have seen this step land
      asm volatile("" ::: "memory");       // no LDS access of this step may be scheduled above the barrier
      if (mine) {
        const float* As = smem + stage * STAGE + (wm * 64 + l31) * BK;
        const float* Ws = smem + stage * STAGE + BM * BK + (wn * 32 * TN + l31) * BK;
        auto load_frag = [&](int j) {
          Frag r;
          const int ch = ((j * 2 + hh) ^ sw) * 4;
#pragma unroll
          for (int a = 0; a < 2; ++a) r.a[a] = *reinterpret_cast<const f32x4*>(As + a * 32 * BK + ch);
#pragma unroll
          for (int b = 0; b < TN; ++b) r.b[b] = *reinterpret_cast<const f32x4*>(Ws + b * 32 * BK + ch);
          return r;
        };
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
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this step's fragment reads are retired before the next barrier
      } else if (pending) {
        // the other group is in its K loop: write out our previous tile beside it, an equal share of the items per K step
        // (item_acc counts NITEMS per step against KT: no division, nothing to do in most steps of a K-deep tile)
        item_acc += n_items;
        if (item_acc >= KT) {
          int n = 0;
          while (item_acc >= KT) {
            item_acc -= KT;
            ++n;
          }
          if (p.ws_flags & 4) {                       // (bit 2: timing experiment, no epilogue)
          } else if constexpr (direct) write_direct(item_next, item_next + n, pend_m0, pend_n0);
          else write_items(item_next, item_next + n, pend_m0, pend_n0);
          item_next += n;
        }
        if (kt == KT - 1) pending = false;
      }
      stage = stage + 1 == NSTG ? 0 : stage + 1;
    }
    if (mine) {
      __builtin_amdgcn_s_setprio(0);
      pending = true;
      pend_m0 = m0;
      pend_n0 = n0;
      item_acc = 0;
      item_next = 0;
    }
  }
  __builtin_amdgcn_s_barrier();            // behind the last step (the loaders' closing barrier)
  if (pending && !(p.ws_flags & 4)) {
    if constexpr (direct) write_direct(0, DITEMS, pend_m0, pend_n0);
    else write_items(0, 4, pend_m0, pend_n0);
  }   // the last tile of this group: nobody left to hide behind
}

// one persistent workgroup per CU of the current device (a multiple of the 8 XCDs, so that tile id and workgroup id agree mod 8:
// gemm_tile_coords' XCD-aware mapping keeps its meaning)
int gemm_pp_workgroups() {
  static int n[COTR_MAX_DEVICES] = {};
  int& v = n[cotr_current_device()];
  if (v == 0) {
    hipDeviceProp_t prop;
    int dev = 0;
    v = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8)
      v = prop.multiProcessorCount / 8 * 8;
  }
  return v;
}

template <int TN, int MODE, bool DIRECT>
static int launch_pp_t(const GemmParams& p0, hipStream_t s) {
  constexpr int BM = 128, BN = 64 * TN;
  constexpr int NSTG = TN == 2 ? 3 : 4;
  constexpr size_t smem = ((size_t)NSTG * (BM + BN) * BK + (size_t)4 * 32 * (32 * TN + 4)) * sizeof(float);
  GemmParams p = p0;
  if (p.N % BN != 0 || p.K % BK != 0 || p.K < 2 * BK || p.M <= 0 || p.A2 != nullptr) return -1;
  if (p.ldc % 4 != 0 || (p.residual && p.ldr % 4 != 0)) return -1;
  if (((uintptr_t)p.C & 15) || ((uintptr_t)p.residual & 15)) return -1;
  if (p.zeros == nullptr) p.zeros = gemm_zero_buffer();
  if (p.zeros == nullptr) return -2;
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pp_kernel<TN, MODE, DIRECT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem) != hipSuccess)
      return -2;
    attr_set.set();
  }
  if (!gemm_fill_divs(p, MODE, BM, BN)) return -1;
  p.ws_flags = knob(KN_WS_FLAGS);
  const int tiles = gemm_grid_tiles(p, BM, BN);
  const int cus = gemm_pp_workgroups();
  const int grid = tiles < cus ? tiles : cus;
  hipLaunchKernelGGL((gemm_pp_kernel<TN, MODE, DIRECT>), dim3(grid), dim3(768), smem, s, p, tiles);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// variant 0: 128 x 128 tiles, 1: 128 x 64
int launch_gemm_pp(int mode, int variant, const GemmParams& p, hipStream_t s) {
  if (mode == GEMM_DENSE && p.lda % 4 != 0) return -1;
  if (mode != GEMM_DENSE && mode != GEMM_CONV) return -1;
  const bool d = mode == GEMM_DENSE;
  switch (variant) {
    // (the 128 x 128 instantiations are not built: with the piece-wise write-out hipcc keeps half of their accumulators in scratch -
    //  43 TFLOP/s; their item-by-item form measured 375 us against 311 us for configuration 26 at 262144 x 256 x 256)
    case 1: return d ? launch_pp_t<1, GEMM_DENSE, false>(p, s) : launch_pp_t<1, GEMM_CONV, false>(p, s);
    case 3: return d ? launch_pp_t<1, GEMM_DENSE, true>(p, s) : launch_pp_t<1, GEMM_CONV, true>(p, s);    // LDS-free write-out
    default: return -1;
  }
}
