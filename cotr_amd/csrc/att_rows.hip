// Attention sub-layer for MANY rows (the batched regime) in ONE launch, fp32 MFMA, gfx950:
//     Y = LayerNorm( residual + out_proj( MHA( q, K, V ) ) )          8 heads of 32, 512 keys per pair, no masks
//   encoder  (COTR/models/transformer.py:149-155): q = the q columns of the packed in-projection (given, pre-scaled by 32^-0.5)
//   decoder  (transformer.py:192-198):             q = ((tgt + query_pos) . Wq^T + bq) * 32^-0.5 computed here (QP)
// It replaces [q projection,] attention, out-projection (+ residual) and the LayerNorm launch of the unfused path: three K = 256
// launches whose tiles pay their fixed cost per 128 rows (profiles/r4_tile_fixed_cost_vs_k_steps.txt) and the attention output's
// round trip through HBM.  Same "rows" decomposition and wave-private operand streams as ffn_rows.hip:
//
//   workgroup = 64 query rows of ONE pair, 8 wavefronts (two per SIMD); wavefront w owns head w.
//   T    [64 x 256] tile in LDS (k-tiled [8][64][32], 16-B chunks XOR-swizzled): QP: tgt + query_pos, else the q columns; after the
//        attention phase the same 64 KB hold the concatenated head outputs O.
//   QP   q_h^T [32 d x 64 rows] = Wq_h . T^T (8 pieces of Wq_h): lands in the MFMA D layout, which IS the B-operand layout of
//        S^T = K_h . q_h^T (attention.hip's trick: lane = query, register r = head dim (r&3) + 8(r>>2) + 4*half)
//   KV   per 32-key block: one K piece and one V piece (4 KB each) -
//          S^T = K_blk . q^T  (16 MFMAs per 32-query block u), online softmax in the log2 domain (lane = one query, 16 scores in
//          registers + one cross-half swap), O^T += V_blk^T . P^T (16 MFMAs, P registers feed the B operand directly).
//   OUT  after a barrier (all eight O_h are in T): Y[64 x 32 columns of this wavefront] = O . Wo[32w .. 32w+32, :]^T, 8 pieces of Wo,
//        then + bias + residual, LayerNorm over the whole row through LDS (ffn_rows.hip's epilogue: 8 lanes per row, DPP sums).
//   Every operand that is not the T tile - Wq_h, K_h, V_h, the wavefront's rows of Wo - is used by exactly ONE wavefront of the
//   workgroup: each wavefront requests its own 4 KB pieces ([32 rows][32 floats]) by LDS-DMA into its own TWO-slot ring (slot 0: K
//   pieces / even pieces of a GEMM phase, slot 1: V / odd) and reads them back itself - no barrier in the main loops, ordering by the
//   wavefront's own counted vmcnt.  A slot is requested again as soon as its fragments are in registers (the K fragments of block b+1
//   are read during the PV product of block b, the V fragments of block b at the start of its S product), so every piece is one
//   whole block - 64 matrix instructions of this wavefront, ~4000-8000 cycles beside its neighbour - ahead of its use.
//
//   Two wavefronts per SIMD (round 5, second form; the first had 4 wavefronts x 2 heads and four-slot rings): 240 k -> 232 k cycles
//   per encoder tile, 273 k -> 261 k per decoder tile (tools/micro/att_rows_probe.hip), 2107 -> 2016 us of the 32 x 1000 forward.
//   That is all two symmetric wavefronts give: each still pays its own softmax and its own request / wait instructions in the
//   matrix pipe's time - the SIMD grants the pipe to ONE of them until it stalls (two wavefronts of matrix instructions only: the
//   first finishes in 259 k cycles, the second in 518 k), and a wavefront's VALU block waits for ITS matrix results whoever else runs
//   (tools/micro/kv_model.hip: matrix pipe 0.826 busy with one such wavefront per SIMD, 0.862 with two, whatever the start offset or
//   s_setprio; a VALU-only neighbour costs a matrix wavefront nothing - mfma_beside.hip - but handing S / P to one through LDS with
//   a barrier per block gave 0.714).
#include <type_traits>
#include <utility>

#include "common.h"

#define AR_D 256
#define AR_KEYS 512
#define AR_BM 64
#define AR_NSLOT 2               // ring slots per wavefront: slot 0 = K pieces (even pieces of a phase), slot 1 = V pieces (odd)
#define AR_NW 8                  // wavefronts per workgroup: one per head, two per SIMD
#define AR_PIECE 1024            // floats per piece: 32 rows x 32 floats
#define AR_LDT 288               // row of the epilogue tile: consecutive rows 32 banks apart (ffn_rows.hip)

struct AttRowsParams {
  const float* q;        // !QP: [pairs*nq][ldq] query rows, head h at columns 32h (already scaled by 32^-0.5)
  int ldq;
  const float* x;        // QP: [pairs*nq][256] rows to project, or nullptr (decoder layer 0: tgt == 0)
  const float* x2;       // QP: [pairs*nq][256] added to x first (query_pos)
  const float* wq;       // QP: [256][256]
  const float* bq;       // QP: [256]
  float qscale;          // QP: 32^-0.5
  const float* k;        // [pairs*512][ldkv], head h at columns 32h
  const float* v;
  int ldkv;
  const float* wo;       // [256][256] out_proj.weight
  const float* bo;       // [256]
  const float* residual; // [pairs*nq][256] or nullptr
  const float* ln_w;
  const float* ln_b;
  float* Y;              // [pairs*nq][256]
  const float* zeros;
  int nq;                // query rows per pair
  int nb, tpp;           // pairs; 64-row tiles per pair
  int by_xcd;            // 1: 1-D grid, all tiles of a pair on ONE XCD (see att_rows_kernel); 0: grid (tiles per pair, pairs)
  float* dbg;            // debug instantiation only (tools/micro/att_rows_probe.hip): stage dumps of workgroup 0
};

constexpr int AR_T = 8 * AR_BM * 32;                      // floats of the T tile
constexpr int AR_RING = AR_NSLOT * AR_PIECE;
constexpr int AR_MAIN = AR_T + AR_NW * AR_RING + 4 * AR_D;    // T, rings, then bq, bo, ln_w, ln_b (staged once: the epilogue reads them from LDS)
constexpr int AR_EPI = AR_BM * AR_LDT;
constexpr size_t kAttRowsSmem = (size_t)(AR_MAIN > AR_EPI ? AR_MAIN : AR_EPI) * sizeof(float);
static_assert(kAttRowsSmem <= 160 * 1024, "LDS");

__device__ __forceinline__ float ar_group8_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  return v;
}
// Combine lane i with lane i ^ 32 (v_permlane32_swap: lanes 32-63 of the first register <-> lanes 0-31 of the second; no LDS crossbar).
// Inline asm on purpose: given the SAME value for both operands, hipcc (ROCm 7.2) keeps the swap but then uses only its first result
// ("max(r0, r0)": every lane got the LOWER half's value - softmax statistics of half the keys; tools/micro/att_rows_probe.hip found it).
__device__ __forceinline__ void ar_xhalf_swap(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float ar_xhalf_max(float v) {
  float a = v, b = v;
  ar_xhalf_swap(a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float ar_xhalf_sum(float v) {
  float a = v, b = v;
  ar_xhalf_swap(a, b);
  return a + b;
}
__device__ __forceinline__ void ar_dma16(const float* src, float* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

struct ArLane {
  float* T;             // the tile
  float* ring;          // this wavefront's ring
  const float* bqs;     // bq in LDS
  int l31, hh, sw;
  int drow, lch4;       // LDS-DMA: this lane's row (0-7) inside an instruction's 8 rows and 4 x its logical chunk for an even instruction
                        // (odd q: offset ^ 16 - the other half of the chunk swizzle; row strides are multiples of 32 floats)
};

// ---- the piece sequence of a wavefront: [QP: 8 pieces of Wq_h] then 16 x (K block, V block), then 8 pieces of Wo.  The consumer
// walks it in program order; this iterator walks ahead of it for the requests (two pieces in the GEMM phases, three in the KV phase). ----
struct ArPiece {
  const float* base;    // first row, first column of the piece (wave-uniform)
  int kv;               // rows are ldkv apart (K / V) instead of 256
};
template <bool QP>
struct ArIter {
  int phase, i;
  __device__ __forceinline__ void init() { phase = QP ? 0 : 1; i = 0; }
  __device__ __forceinline__ ArPiece next(const AttRowsParams& p, const int wave, const size_t key0) {
    ArPiece r;
    if (QP && phase == 0) {
      r.base = p.wq + (size_t)wave * 32 * AR_D + i * 32;
      r.kv = 0;
      if (++i == 8) { i = 0; phase = 1; }
    } else if (phase == 1) {
      r.base = ((i & 1) ? p.v : p.k) + (key0 + (size_t)(i >> 1) * 32) * p.ldkv + wave * 32;
      r.kv = 1;
      if (++i == 32) { i = 0; phase = 2; }
    } else {
      r.base = p.wo + (size_t)(32 * wave) * AR_D + i * 32;
      r.kv = 0;
      ++i;
    }
    return r;
  }
};
// instruction q (8 rows) of a piece -> ring slot
__device__ __forceinline__ void ar_dma_q(const AttRowsParams& p, const ArLane& L, const ArPiece& pc, const int slot, const int q) {
  const int stride = pc.kv ? p.ldkv : AR_D;
  const int off = (L.drow * stride + L.lch4) ^ ((q & 1) << 4);   // (computed, not selected: hipcc turns a uniform select of two lane values into a scratch array)
  ar_dma16(pc.base + (size_t)(q * 8) * stride + off, L.ring + slot * AR_PIECE + q * 256);
}

// fragment of one 8-deep step of a GEMM-like phase: a = the streamed operand (ring), x0 / x1 = the T rows of the two 32-row blocks
struct ArFrag {
  f32x4 a, x0, x1;
};
template <int SLOT, int KT, int J>
__device__ __forceinline__ ArFrag ar_load_frag(const ArLane& L) {
  ArFrag f;
  const int ch = ((J * 2 + L.hh) ^ L.sw) * 4;
  f.a = *reinterpret_cast<const f32x4*>(L.ring + SLOT * AR_PIECE + L.l31 * 32 + ch);
  const float* Tk = L.T + KT * (AR_BM * 32) + L.l31 * 32 + ch;
  f.x0 = *reinterpret_cast<const f32x4*>(Tk);
  f.x1 = *reinterpret_cast<const f32x4*>(Tk + 32 * 32);
  return f;
}

// ---- QP phase, one head: q^T[d][row] = Wq_h . T^T over 8 pieces (32 steps of 8 MFMAs) ------------------------------------------------
// Step order as in ffn_rows.hip: first MFMA pair, then the next step's fragment requests (behind the counted wait where the next
// step opens a new piece), then the other pairs; during a piece's LAST step (its fragments are all in registers) the request of
// piece +2 goes into its slot.
template <bool QP, int S_>
__device__ __forceinline__ void ar_q_step(const AttRowsParams& p, const ArLane& L, const int wave, const size_t key0, ArIter<QP>& it,
                                          ArFrag& cur, f32x16 (&qacc)[2]) {
  constexpr int sub = S_ >> 2, j = S_ & 3;
  ArFrag nxt = cur;
  ArPiece np = {};
  if constexpr (j == 3) np = it.next(p, wave, key0);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    qacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[e], cur.x0[e], qacc[0], 0, 0, 0);
    qacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[e], cur.x1[e], qacc[1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (e == 0) {
      if constexpr (S_ + 1 < 32) {
        if constexpr (j == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next piece has landed (nothing younger is in flight)
        nxt = ar_load_frag<((S_ + 1) >> 2) % AR_NSLOT, ((S_ + 1) >> 2), ((S_ + 1) & 3)>(L);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (j == 3) {   // this piece's last fragment is in registers (cur): its slot takes piece +2
      ar_dma_q(p, L, np, sub % AR_NSLOT, e);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  cur = nxt;
}
template <bool QP, int... S_>
__device__ __forceinline__ void ar_q_steps(const AttRowsParams& p, const ArLane& L, const int wave, const size_t key0, ArIter<QP>& it,
                                           ArFrag& cur, f32x16 (&qacc)[2], std::integer_sequence<int, S_...>) {
  (ar_q_step<QP, S_>(p, L, wave, key0, it, cur, qacc), ...);
}

// ---- online softmax of one 32-key block for both 32-query blocks, log2 domain, with a LAZY reference ---------------------------------------
// VALU work of a wavefront does NOT overlap with that wavefront's own matrix instructions, however the two are interleaved - measured
// three ways in the 4-wavefront form (compiler-scheduled, sched_group_barrier patterns, one softmax slice pinned behind every matrix
// instruction: the K/V phase took 205 k cycles each time against 131 k of matrix instructions, 132 k with the softmax compiled out;
// profiles/r5_att_rows_probe.txt), and a second wavefront of the same kind on the SIMD hides little of it (header).  So the softmax
// is priced in VALU cycles and written to need few of them:
//   * p = 2^(s - m_ref) with a per-query reference m_ref that is only raised when some query of the wavefront sees a block maximum more
//     than AR_LAZY above it (p <= 2^AR_LAZY: sums of 512 such terms are far inside fp32).  In the common case - every block after the
//     first few - there is no alpha and no rescaling of the 32 output accumulators.  The result, 2^(-m_ref) . (sum_k 2^s_k V_k) /
//     (2^(-m_ref) . sum_k 2^s_k), does not depend on the reference up to rounding.
//   * matrix results live in VGPRs (build flag -amdgpu-mfma-vgpr-form): no v_accvgpr_read / write around every VALU use.
//   * independent instructions back to back (a dependent chain pays its full latency): the 32 subtractions,
//     then the 32 exponentials, then tree sums; 3-input maxima.
#define AR_LAZY 8.f
__device__ __forceinline__ float ar_max16(const f32x16& s) {
  const float a = __builtin_fmaxf(__builtin_fmaxf(s[0], s[1]), s[2]), b = __builtin_fmaxf(__builtin_fmaxf(s[3], s[4]), s[5]);
  const float c = __builtin_fmaxf(__builtin_fmaxf(s[6], s[7]), s[8]), d = __builtin_fmaxf(__builtin_fmaxf(s[9], s[10]), s[11]);
  const float e = __builtin_fmaxf(__builtin_fmaxf(s[12], s[13]), s[14]);
  return __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(a, b), __builtin_fmaxf(c, d)), __builtin_fmaxf(e, s[15]));
}
__device__ __forceinline__ void ar_softmax2(f32x16 (&s)[2], float (&m_ref)[2], float (&l_run)[2], f32x16 (&o)[2]) {
  float mx[2] = {ar_max16(s[0]), ar_max16(s[1])};
  mx[0] = ar_xhalf_max(mx[0]);
  mx[1] = ar_xhalf_max(mx[1]);
  if (__builtin_amdgcn_ballot_w64(mx[0] > m_ref[0] + AR_LAZY || mx[1] > m_ref[1] + AR_LAZY) != 0) {   // rare: raise the references
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float m_new = fmaxf(m_ref[u], mx[u]);
      const float alpha = __builtin_amdgcn_exp2f(m_ref[u] - m_new);   // first block: 2^(-inf) = 0
      l_run[u] *= alpha;
      m_ref[u] = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[u][r] *= alpha;
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[u][r] -= m_ref[u];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[u][r] = __builtin_amdgcn_exp2f(s[u][r]);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    float t8[8], t4[4];
#pragma unroll
    for (int r = 0; r < 8; ++r) t8[r] = s[u][r] + s[u][r + 8];
#pragma unroll
    for (int r = 0; r < 4; ++r) t4[r] = t8[r] + t8[r + 4];
    l_run[u] += (t4[0] + t4[2]) + (t4[1] + t4[3]);
  }
}

// K fragments of a block (A operand of S^T): lane (key l31, half hh) holds K[key][8j + 4hh .. +3], j = 0..3
template <int SLOT>
__device__ __forceinline__ void ar_load_k(const ArLane& L, f32x4 (&kf)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
    kf[j] = *reinterpret_cast<const f32x4*>(L.ring + SLOT * AR_PIECE + L.l31 * 32 + ((j * 2 + L.hh) ^ L.sw) * 4);
}
// V fragments (A operand of O^T): lane (d l31, half hh) holds V[key(r, hh)][d], key(r, hh) = (r&3) + 8(r>>2) + 4hh
template <int SLOT>
__device__ __forceinline__ void ar_load_v(const ArLane& L, float (&vf)[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int key = (r & 3) + 8 * (r >> 2) + 4 * L.hh;
    vf[r] = L.ring[SLOT * AR_PIECE + key * 32 + ((((L.l31 >> 2) ^ ((key >> 1) & 7)) << 2) | (L.l31 & 3))];
  }
}

// ---- one 32-key block of the KV phase (KB_ = parity of the block: pieces 2*KB_ (K) and 2*KB_+1 (V) of the ring) --------------------------
// in: kf = this block's K fragments; out: kf = the next block's (unless `last`: the head's final block of the wavefront's final head)
template <bool QP, int KB_, bool DBG = false, int ABL = 0>   // ABL (probe only): 1 = no softmax VALU, 2 = no requests / waits in this phase
__device__ __forceinline__ void ar_kv_block(const AttRowsParams& p, const ArLane& L, const int wave, const size_t key0, ArIter<QP>& it,
                                            const bool last, f32x4 (&kf)[4], const f32x16 (&q16)[2], f32x16 (&o)[2], float (&m_run)[2],
                                            float (&l_run)[2], const bool dump = false) {
  constexpr int KS = 0, VS = 1;                            // ring slots of the K and V pieces
  f32x16 s[2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[u][r] = 0.f;
  float vf[16];
  // ---- S^T of both query blocks: 32 matrix instructions, the two accumulators alternating (a v_mfma that accumulates onto the result
  // of the one right before it waits for it); the V fragments are read behind the first pair, the next block's V piece is requested
  // into the same slot behind the next four ----
  const ArPiece npa = it.next(p, wave, key0);               // the next block's V piece (the last block: Wo piece 1)
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    s[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[n >> 2][n & 3], q16[0][n], s[0], 0, 0, 0);
    s[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[n >> 2][n & 3], q16[1][n], s[1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (n == 0) {
      if constexpr (!(ABL & 2)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // this block's V piece has landed (the next K piece may be in flight)
      ar_load_v<VS>(L, vf);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (n >= 1 && n <= 4) {
      if (n == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the V fragments are in registers: the slot is free
      if constexpr (!(ABL & 2)) ar_dma_q(p, L, npa, VS, n - 1);                     // -> the V slot
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (!(ABL & 1)) ar_softmax2(s, m_run, l_run, o);
  __builtin_amdgcn_sched_barrier(0);
  // ---- O^T += V^T . P^T of both query blocks: 32 matrix instructions, alternating; the next block's K fragments behind the first
  // pair, the K piece two blocks ahead is requested into the same slot behind the next four ----
  ArPiece npb = {};
  if (!last) npb = it.next(p, wave, key0);                  // the K piece two blocks ahead (the last but one block: Wo piece 0)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[r], s[0][r], o[0], 0, 0, 0);
    o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[r], s[1][r], o[1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (r == 0 && !last) {
      if constexpr (!(ABL & 2)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // the next block's K piece has landed (its V piece may be in flight)
      ar_load_k<KS>(L, kf);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (r >= 1 && r <= 4 && !last) {                                                // (the last block: both slots hold Wo pieces already)
      if (r == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the next block's K fragments are in registers: the slot is free
      if constexpr (!(ABL & 2)) ar_dma_q(p, L, npb, KS, r - 1);                     // -> the K slot
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (DBG) {   // [53248 ..): block dump of one wavefront: vf [16][64], p(u0) [16][64], o(u0) after PV [16][64]
    if (dump) {
      const int lane = L.l31 + 32 * L.hh;
      for (int r = 0; r < 16; ++r) {
        p.dbg[53248 + r * 64 + lane] = vf[r];
        p.dbg[53248 + 1024 + r * 64 + lane] = s[0][r];
        p.dbg[53248 + 2048 + r * 64 + lane] = o[0][r];
      }
    }
  }
}

// ---- OUT phase: Y[64 x 64 columns of this wavefront] = O . Wo_rows^T, 16 pieces (64 steps), the tail of the wavefront's stream ----
template <bool QP, int S_>
__device__ __forceinline__ void ar_out_step(const AttRowsParams& p, const ArLane& L, const int wave, const size_t key0, ArIter<QP>& it,
                                            ArFrag& cur, f32x16 (&yacc)[2]) {
  constexpr int sub = S_ >> 2, j = S_ & 3;
  constexpr bool refill = j == 3 && sub + AR_NSLOT <= 7;
  ArFrag nxt = cur;
  ArPiece np = {};
  if constexpr (refill) np = it.next(p, wave, key0);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    yacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.x0[e], cur.a[e], yacc[0], 0, 0, 0);
    yacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.x1[e], cur.a[e], yacc[1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (e == 0) {
      if constexpr (S_ + 1 < 32) {
        if constexpr (j == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next piece has landed (nothing younger is in flight)
        nxt = ar_load_frag<((S_ + 1) >> 2) % AR_NSLOT, ((S_ + 1) >> 2), ((S_ + 1) & 3)>(L);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (refill) {
      ar_dma_q(p, L, np, sub % AR_NSLOT, e);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  cur = nxt;
}
template <bool QP, int... S_>
__device__ __forceinline__ void ar_out_steps(const AttRowsParams& p, const ArLane& L, const int wave, const size_t key0, ArIter<QP>& it,
                                             ArFrag& cur, f32x16 (&yacc)[2], std::integer_sequence<int, S_...>) {
  (ar_out_step<QP, S_>(p, L, wave, key0, it, cur, yacc), ...);
}

template <bool QP, bool DBG = false, int ABL = 0>
__global__ __launch_bounds__(512, 2) void att_rows_kernel(const AttRowsParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* T = smem;
  float* bqs = smem + AR_T + AR_NW * AR_RING;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  // Workgroup -> (pair, tile).  Workgroups are dealt to the 8 XCDs round-robin by their linear id and every XCD has its own L2, so
  // with the plain (tile, pair) grid the 8 / 16 tiles of a pair land on 8 different XCDs and the pair's K_h / V_h (1 MB per layer) are
  // pulled into all eight L2s (profiles/r5_final_mfma_util_and_traffic_b32_q1000.txt: 1.9 / 2.4 GB per forward).  by_xcd (pairs a
  // multiple of 8; knob xcd_mapping bit 5 turns it off): a 1-D grid; linear id b runs on XCD b & 7 and is the (b >> 3)-th workgroup
  // there; XCD x owns pairs x, x + 8, x + 16, ... and walks them tile by tile, so the tiles of a pair that are resident together
  // share one L2.  Measured (profiles/r6_mfma_util_b32_q1000_{plain_grid,pair_per_xcd}.txt, 32 pairs x 1000 queries): L2 <-> fabric
  // bytes of the encoder form 1927 -> 518 MB, of the decoder form 2385 -> 955 MB per forward (the whole forward 15.6 -> 12.8 GB);
  // time: encoder form 641 -> 622 us (matrix pipes 0.644 -> 0.664 busy), decoder form unchanged (1394 -> 1399 us) - the kernel is
  // bound by its softmax VALU + matrix pipe, not by the fabric.  Other pair counts keep the plain grid (a padded 1-D grid leaves
  // XCDs idle: 4 pairs on 4 of the 8 XCDs ran the dense pass 4 x slower).  Placement is for speed only: any mapping gives the same results.
  int pair, tile;
  if (p.by_xcd) {
    const int j = blockIdx.x >> 3;
    pair = (blockIdx.x & 7) + 8 * (j / p.tpp);
    tile = j % p.tpp;
    if (pair >= p.nb) return;
  } else {
    pair = blockIdx.y;
    tile = blockIdx.x;
  }
  const int q0 = tile * AR_BM;                             // first query of this tile inside its pair
  const size_t row0 = (size_t)pair * p.nq + q0;            // ... its global row
  const size_t key0 = (size_t)pair * AR_KEYS;
  const int nvalid = p.nq - q0 < AR_BM ? p.nq - q0 : AR_BM;

#define AR_STAMP(slot)                                                                                                    \
  do {                                                                                                                    \
    if constexpr (DBG) {                                                                                                  \
      if (t == 0) {                                                                                                       \
        unsigned long long* st = reinterpret_cast<unsigned long long*>(p.dbg + 131072) + ((size_t)(pair * p.tpp + tile) * 8 + (slot)) * 2; \
        st[0] = __builtin_readcyclecounter();                                                                             \
        st[1] = wall_clock64();                                                                                           \
      }                                                                                                                   \
    }                                                                                                                     \
  } while (0)
  AR_STAMP(0);
  const int drow = lane >> 3, pch = lane & 7;
  const int lch_e = pch ^ (drow >> 1);
  ArLane L;
  L.T = T; L.ring = smem + AR_T + wave * AR_RING; L.bqs = bqs; L.l31 = l31; L.hh = hh; L.sw = (l31 >> 1) & 7;
  L.drow = drow; L.lch4 = lch_e * 4;

  // ---- prologue: the T tile, bq -> LDS, the first two pieces ----
  if constexpr (QP) {
    // T = x + x2 through registers, in the LDS-DMA pattern (a wave instruction = 8 rows x 128 B of one k-tile: conflict-free 16-B
    // writes); rows past the pair's last query re-read the tile's first row (never stored)
#pragma unroll 4
    for (int i = 0; i < 8; ++i) {
      const int idx = wave * 8 + i;
      const int kt = idx >> 3, rg = idx & 7;
      const int row = rg * 8 + drow;
      const int lch = pch ^ ((row >> 1) & 7);
      const size_t g = (row0 + (row < nvalid ? row : 0)) * AR_D + kt * 32 + lch * 4;
      f32x4 v = *reinterpret_cast<const f32x4*>(p.x2 + g);
      if (p.x != nullptr) v += *reinterpret_cast<const f32x4*>(p.x + g);
      *reinterpret_cast<f32x4*>(T + kt * (AR_BM * 32) + row * 32 + pch * 4) = v;
    }
    if (t < AR_D) bqs[t] = p.bq[t];
  } else {
    // T = the q columns by LDS-DMA: 64 instructions (k-tile = head, 8 rows each), 16 per wavefront
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = wave * 8 + i;
      const int kt = idx >> 3, rg = idx & 7;
      const int row = rg * 8 + drow;
      const int lch = pch ^ ((row >> 1) & 7);
      ar_dma16(p.q + (row0 + (row < nvalid ? row : 0)) * p.ldq + kt * 32 + lch * 4, T + kt * (AR_BM * 32) + rg * 256);
    }
  }
  if (t < AR_D) {
    bqs[AR_D + t] = p.bo[t];
    bqs[2 * AR_D + t] = p.ln_w[t];
    bqs[3 * AR_D + t] = p.ln_b[t];
  }
  ArIter<QP> it;
  it.init();
#pragma unroll
  for (int s = 0; s < AR_NSLOT; ++s) {   // both slots: Wq pieces 0, 1 / K block 0, V block 0
    const ArPiece pc = it.next(p, wave, key0);
#pragma unroll
    for (int q = 0; q < 4; ++q) ar_dma_q(p, L, pc, s, q);
  }
  LDS_DMA_WAIT_ALL();
  __syncthreads();

  if constexpr (DBG) {   // [57344 .. 54096): wavefront 0's ring right after the prologue
    if (pair == 0 && tile == 0 && wave == 0)
      for (int i = lane; i < AR_RING; i += 64) p.dbg[57344 + i] = L.ring[i];
  }
  AR_STAMP(1);
  // ---- q of this wavefront's head, as B operands of S^T: q16[u][r] = q[row 32u + l31][d = (r&3) + 8(r>>2) + 4hh] * log2(e) ----
  f32x16 q16[2];
  if constexpr (QP) {
    const float qs = p.qscale * 1.44269504088896340736f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) q16[u][r] = 0.f;
    ArFrag cur = ar_load_frag<0, 0, 0>(L);
    ar_q_steps<QP>(p, L, wave, key0, it, cur, q16, std::make_integer_sequence<int, 32>{});
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // K block 0 has landed (V block 0 may be in flight)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bqs + wave * 32 + 8 * g + 4 * hh);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) q16[u][g * 4 + e] = (q16[u][g * 4 + e] + bv[e]) * qs;
    }
  } else {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 qv = *reinterpret_cast<const f32x4*>(T + wave * (AR_BM * 32) + (u * 32 + l31) * 32 + ((j * 2 + hh) ^ L.sw) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) q16[u][j * 4 + e] = qv[e] * 1.44269504088896340736f;
      }
  }

  if constexpr (DBG) {   // [0 .. 16384): q16 as [wave][hd][u][r][lane]
    if (pair == 0 && tile == 0)
      for (int u = 0; u < 2; ++u)
        for (int r = 0; r < 16; ++r) p.dbg[((wave * 2 + u) * 16 + r) * 64 + lane] = q16[u][r];
  }
  AR_STAMP(2);
  // ---- KV phase: 16 key blocks ----
  f32x4 kf[4];
  ar_load_k<0>(L, kf);
  {
    // K block 1 -> the K slot, once block 0's fragments are in registers
    const ArPiece pc = it.next(p, wave, key0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (!(ABL & 2)) {
#pragma unroll
      for (int q = 0; q < 4; ++q) ar_dma_q(p, L, pc, 0, q);
    }
  }
  f32x16 onorm[2];                                         // normalised O_h^T (D layout), written to T behind a barrier
  {
    f32x16 o[2];
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[u][r] = 0.f;
    for (int kb2 = 0; kb2 < 8; ++kb2) {
      ar_kv_block<QP, 0, DBG, ABL>(p, L, wave, key0, it, false, kf, q16, o, m_run, l_run,
                              DBG && kb2 == 0 && wave == 0 && pair == 0 && tile == 0);
      ar_kv_block<QP, 1, DBG, ABL>(p, L, wave, key0, it, kb2 == 7, kf, q16, o, m_run, l_run);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float inv = 1.f / ar_xhalf_sum(l_run[u]);
#pragma unroll
      for (int r = 0; r < 16; ++r) onorm[u][r] = o[u][r] * inv;
      if constexpr (DBG) {   // [16384 .. 32768): onorm as [wave][u][r][lane]; [32768 .. ): l, m per [wave][u][lane]
        if (pair == 0 && tile == 0) {
          for (int r = 0; r < 16; ++r) p.dbg[16384 + ((wave * 2 + u) * 16 + r) * 64 + lane] = onorm[u][r];
          p.dbg[32768 + (wave * 2 + u) * 64 + lane] = l_run[u];
          p.dbg[32768 + 1024 + (wave * 2 + u) * 64 + lane] = m_run[u];
        }
      }
    }
  }
  AR_STAMP(3);
  // ---- O -> T (k-tile = head; register group g of lane (row l31, half hh) = head dims 8g + 4hh .. +3: one 16-B chunk) ----
  __syncthreads();                                         // everybody is done reading T (QP: the rows to project; else: its own q columns)
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 ov = {onorm[u][g * 4], onorm[u][g * 4 + 1], onorm[u][g * 4 + 2], onorm[u][g * 4 + 3]};
      *reinterpret_cast<f32x4*>(T + wave * (AR_BM * 32) + (u * 32 + l31) * 32 + (((g * 2 + hh) ^ L.sw) << 2)) = ov;
    }
  __syncthreads();                                         // all eight heads' outputs are in T

  AR_STAMP(4);
  // ---- OUT phase: Y[64 rows x columns 32 wave .. +31] ----
  f32x16 yacc[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) yacc[a][r] = 0.f;
  {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // Wo piece 0 has landed (piece 1 may be in flight)
    ArFrag cur = ar_load_frag<0, 0, 0>(L);
    ar_out_steps<QP>(p, L, wave, key0, it, cur, yacc, std::make_integer_sequence<int, 32>{});
  }

  if constexpr (DBG) {   // [36864 .. ): yacc as [wave][mb][r][lane]
    if (pair == 0 && tile == 0)
      for (int mb = 0; mb < 2; ++mb)
        for (int r = 0; r < 16; ++r) p.dbg[36864 + ((wave * 2 + mb) * 16 + r) * 64 + lane] = yacc[mb][r];
  }
  AR_STAMP(5);
  // ---- epilogue: Y tile -> LDS, + bias + residual, LayerNorm (8 lanes per row, ffn_rows.hip) ----
  __syncthreads();                                         // nobody reads T any more; no DMA is in flight
  AR_STAMP(6);
  const int erow = wave * 8 + (lane >> 3), eseg = (lane & 7) * 4;   // 8 lanes per row, 64 rows: one pass
  f32x4 xr[8];
  {
    const float* rsrc = p.residual + (row0 + (erow < nvalid ? erow : 0)) * AR_D + eseg;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      xr[c] = p.residual != nullptr ? *reinterpret_cast<const f32x4*>(rsrc + c * 32) : z;
    }
  }
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) smem[(mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * AR_LDT + 32 * wave + l31] = yacc[mb][r];
  __syncthreads();
  {
    const int row = erow;
    f32x4 x[8];
    float s1 = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      x[c] = *reinterpret_cast<const f32x4*>(smem + row * AR_LDT + c * 32 + eseg);
      x[c] += *reinterpret_cast<const f32x4*>(bqs + AR_D + c * 32 + eseg);
      x[c] += xr[c];
      s1 += (x[c][0] + x[c][1]) + (x[c][2] + x[c][3]);
    }
    const float mean = ar_group8_sum(s1) * (1.f / 256.f);
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x[c][e] -= mean;
        s2 = fmaf(x[c][e], x[c][e], s2);
      }
    const float rstd = 1.f / sqrtf(ar_group8_sum(s2) * (1.f / 256.f) + 1e-5f);
    if (row < nvalid) {
      float* dst = p.Y + (row0 + row) * AR_D + eseg;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const f32x4 lw = *reinterpret_cast<const f32x4*>(bqs + 2 * AR_D + c * 32 + eseg);
        const f32x4 lb = *reinterpret_cast<const f32x4*>(bqs + 3 * AR_D + c * 32 + eseg);
        f32x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = x[c][e] * rstd * lw[e] + lb[e];
        *reinterpret_cast<f32x4*>(dst + c * 32) = out;
      }
    }
  }
  AR_STAMP(7);
#undef AR_STAMP
}

// Y = LN(residual + out_proj(MHA(q, K, V)) + bo) for nb pairs x nq query rows; q given (wq == nullptr: [rows][ldq], pre-scaled) or
// projected here from x (+ x2).  Y must not alias the residual / x / x2 (a tile's rows are re-read for the residual after other tiles
// may have written theirs - keep the contract simple: distinct buffers).
int launch_att_rows(const float* q, int ldq, const float* x, const float* x2, const float* wq, const float* bq, float qscale,
                    const float* k, const float* v, int ldkv, const float* wo, const float* bo, const float* residual,
                    const float* ln_w, const float* ln_b, float* Y, int nb, int nq, hipStream_t s) {
  if (nb <= 0 || nq <= 0) return 0;
  const bool qp = wq != nullptr;
  if (!k || !v || !wo || !bo || !ln_w || !ln_b || !Y || ldkv % 32 || nb > 65535) return -1;   // (ldkv % 32: the odd-instruction offset is an XOR)
  if (qp ? (!x2 || !bq) : (!q || ldq % 4)) return -1;
  if (Y == residual || Y == x || Y == x2 || Y == q) return -1;
  if (((uintptr_t)q | (uintptr_t)x | (uintptr_t)x2 | (uintptr_t)k | (uintptr_t)v | (uintptr_t)wo | (uintptr_t)wq | (uintptr_t)residual | (uintptr_t)Y) & 15)
    return -1;
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(att_rows_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kAttRowsSmem) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(att_rows_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kAttRowsSmem) != hipSuccess)
      return -2;
    attr_set.set();
  }
  AttRowsParams p;
  p.q = q; p.ldq = ldq; p.x = x; p.x2 = x2; p.wq = wq; p.bq = bq; p.qscale = qscale; p.k = k; p.v = v; p.ldkv = ldkv;
  p.wo = wo; p.bo = bo; p.residual = residual; p.ln_w = ln_w; p.ln_b = ln_b; p.Y = Y; p.zeros = gemm_zero_buffer(); p.nq = nq;
  p.dbg = nullptr;
  p.nb = nb; p.tpp = (nq + AR_BM - 1) / AR_BM;
  p.by_xcd = (knob(KN_XCD_MAPPING) & 32) == 0 && nb % 8 == 0;
  const dim3 grid = p.by_xcd ? dim3(p.tpp * nb) : dim3(p.tpp, nb);
  if (qp) hipLaunchKernelGGL(att_rows_kernel<true>, grid, dim3(64 * AR_NW), kAttRowsSmem, s, p);
  else hipLaunchKernelGGL(att_rows_kernel<false>, grid, dim3(64 * AR_NW), kAttRowsSmem, s, p);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
