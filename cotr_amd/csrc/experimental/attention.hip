// RESEARCH LIBRARY COPY of csrc/attention.hip (libcotr_hip_exp.so only): the product file with the research / dead-end paths that used to sit
// behind #ifdef COTR_EXPERIMENTAL in it resolved IN (tools/unifdef_exp.py -D).  The product never compiles this file.
// Multi-head softmax attention over the fixed 512-token (16x32) COTR memory, fp32 MFMA, gfx950.
//
// Serves both call sites of nn.MultiheadAttention in the reference:
//   encoder self-attention   (COTR/models/transformer.py:149-153)  nq = 512 queries per pair
//   decoder cross-attention  (COTR/models/transformer.py:192-195)  nq = Q   queries per pair
// 8 heads of 32.  q is expected pre-scaled by head_dim^-0.5 (fused into the q-projection epilogue,
// as torch scales q before q.k^T).  No masks: the key-padding mask is all-False for every caller
// of the reference (input is always exactly 256x512, COTR/models/backbone.py:80).
//
// Work decomposition (v2, "split keys"): one workgroup = 32 queries x 1 head x 1 pair; its NS
// wavefronts each take 512/NS keys, so a wavefront runs 512/NS/32 key blocks instead of 16 and
// the grid is (nq/32) x 8 x pairs workgroups (128 for the encoder of one pair, 256 for 1000
// queries) - the 128-query, whole-K-in-LDS tiling of v1 left 7/8 of the CUs idle at one pair.
// K and V fragments go global -> registers directly in MFMA operand layout (each element is used
// by exactly one wavefront once, so LDS staging would be pure overhead; K_h/V_h of a pair are
// 128 KB and L2-resident).  Per 32-key block, per wavefront:
//   S^T = K_blk . Q^T   16x v_mfma_f32_32x32x2_f32   (lane = one query, 16 keys in registers:
//                        the softmax reduction is in-lane plus one cross-half shuffle)
//   online softmax       running max / sum, exp in fp32
//   O^T += V_blk^T . P^T 16x v_mfma_f32_32x32x2_f32   (P registers feed the B operand directly)
// The NS partial (max, sum, O^T) triples are merged through 17 KB of LDS in a fixed order.
#include "../common.h"

#define ATT_KEYS 512
#define ATT_HD 32

// Optional fusions for the small-row regime (one pair / ~1000 query rows, where every launch costs ~4 us of latency):
//   QP  q-projection prologue: q_h = ((x + x2) . Wq_h^T + bq_h) * qscale computed by the workgroup itself for its 32 queries
//       and its head (COTR/models/transformer.py:192 with nn.MultiheadAttention's packed in_proj): Q^T = Wq_h . X^T lands in
//       the MFMA D layout, which IS the B-operand layout of S^T = K . Q^T below (lane = query, register r = head dim
//       (r&3) + 8*(r>>2) + 4*half) - no transpose, no launch of its own.  The 4 wavefronts split the 256-deep contraction
//       and sum through LDS in a fixed order.
//   OP  out-projection epilogue: the workgroup multiplies its merged 32 x 32 O tile by the head's 32 columns of W_out
//       (transformer.py:153,195: out_proj) and writes a partial [32 x 256] row block; the 8 per-head partials are summed, biased,
//       added to the residual and normalised by ln_reduce_kernel - the launch that followed the projection anyway.
struct AttnFuse {
  const float* x;      // QP: [rows][256] rows to project
  const float* x2;     // QP == 2: [rows][256] added to x first (decoder: tgt + query_pos; layer 0 has tgt == 0 -> x = query_pos)
  const float* wq;     // QP: [256][256] q rows of in_proj_weight
  const float* bq;     // QP: [256]
  float qscale;        // QP: head_dim^-0.5
  const float* wo;     // OP: out_proj.weight [256][256]
  float* part;         // OP: [8][rows_total][256]
  int rows_total;      // OP: rows of one partial
  int wt;              // OP: write-through (sc1) stores for the partials (read once, by every XCD)
  unsigned long long* dbg;  // nullptr, or [workgroups][8] phase timestamps (100 MHz wall clock): cotr_debug_attention_times
  CoopTail ct;         // OP: ct.state != nullptr: the 8 head workgroups of a query tile also sum the partials + bias + residual + LayerNorm
};

template <int NS, int QP, bool OP>   // QP: 0 = q given, 1 = project x, 2 = project x + x2
__global__ __launch_bounds__(NS * 64) void attention_kernel(const float* __restrict__ q, int ldq,
                                                            const float* __restrict__ k,
                                                            const float* __restrict__ v, int ldkv,
                                                            float* __restrict__ o, int ldo, int nq, int head_major,
                                                            const AttnFuse fz) {
  static_assert(!(QP || OP) || NS == 4 || NS == 8, "the fused variants are written for 4 or 8 key splits");
  constexpr int NBO = 8 / (NS >= 4 ? NS : 4);  // OP: 32-column blocks of the out projection per wavefront (2 at NS = 4, 1 at NS = 8)
  constexpr int SLD = 32 * (NBO > 0 ? NBO : 1) + 4;  // OP: padded row of a wave-private staging tile
  constexpr int QCH = 256 / (NS >= 4 ? NS : 4);      // QP: model channels a wavefront contracts over
  constexpr int NBLK = ATT_KEYS / NS / 32;  // key blocks per wavefront
  constexpr int RPW = 16 / NS;              // accumulator rows finished per wavefront in the merge
  // key-split merge buffer [NS][16][64]; the out-projection epilogue reuses it as 4 wave-private staging tiles [32][68]
  constexpr int LDS_BUF = OP ? (NS * 32 * SLD > NS * 16 * 64 ? NS * 32 * SLD : NS * 16 * 64) : NS * 16 * 64;
  __shared__ __attribute__((aligned(16))) float lds_buf[LDS_BUF];
  float (*lds_o)[16][64] = reinterpret_cast<float (*)[16][64]>(lds_buf);
  __shared__ float lds_m[NS][32];
  __shared__ float lds_l[NS][32];
  __shared__ __attribute__((aligned(16))) float lds_out[32][36];  // merged O tile [query][d], rows 16-B aligned

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
#define ATT_STAMP(slot)                                                                                                        \
  do {                                                                                                                         \
    if (fz.dbg != nullptr && t == 0) fz.dbg[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 8 + (slot)] = wall_clock64();      \
  } while (0)
  ATT_STAMP(0);
  // head_major: head index fastest over consecutive workgroups (8 heads <-> 8 XCDs), so K_h/V_h of a pair cross the
  // fabric once chip-wide instead of once per XCD
  const int qtiles = gridDim.x >> 3;
  const int head = head_major ? (blockIdx.x & 7) : (blockIdx.x / qtiles);
  const int qtile = head_major ? (blockIdx.x >> 3) : (blockIdx.x % qtiles);
  const int pair = blockIdx.z;
  const int qi = qtile * 32 + l31;
  const bool q_ok = qi < nq;
  const size_t qrow = (size_t)pair * nq + (q_ok ? qi : 0);

  const size_t key0 = (size_t)pair * ATT_KEYS + (size_t)wave * (ATT_KEYS / NS);
  // K fragment (A operand of S^T): lane (key l31, half hh) reads k[key][j*8 + hh*4 .. +3]
  const float* kg = k + (key0 + l31) * ldkv + head * ATT_HD + hh * 4;
  // V fragment (A operand of O^T): lane (d l31, half hh) reads v[key(r, hh)][d]
  const float* vg = v + (key0 + 4 * hh) * ldkv + head * ATT_HD + l31;
  // everything that does not depend on the queries is requested first: its latency hides under the q projection
  f32x4 kf[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) kf[j] = *reinterpret_cast<const f32x4*>(kg + j * 8);
  f32x4 wof[NBO > 0 ? NBO : 1][4];   // OP: W_out fragment of this wave's 32-column blocks (B operand: lane n = l31, k = head dims)
  if constexpr (OP) {
#pragma unroll
    for (int nb = 0; nb < NBO; ++nb) {
      const float* worow = fz.wo + (size_t)((wave * NBO + nb) * 32 + l31) * 256 + head * ATT_HD + hh * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) wof[nb][j] = *reinterpret_cast<const f32x4*>(worow + j * 8);
    }
  }

  // Q^T fragment (B operand): lane holds q[qi][j*8 + hh*4 + e]
  f32x4 qf[4];
  if constexpr (QP) {
    f32x4 bq4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bq4[j] = *reinterpret_cast<const f32x4*>(fz.bq + head * ATT_HD + j * 8 + hh * 4);
    // wave w contracts over model channels [64w, 64w + 64): A = Wq_h (lane: head dim l31, k half hh), B = X^T (lane: query l31)
    f32x16 qacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) qacc[r] = 0.f;
    // every load is unconditional (rows past nq re-read row 0 of the pair and are never stored): a per-element
    // "load or zero" select would make the compiler branch around each load and wait for it - 16 dependent L2 round trips
    constexpr int QJ = QCH / 8;
    const float* wrow = fz.wq + (size_t)(head * ATT_HD + l31) * 256 + wave * QCH + hh * 4;
    const float* xrow = fz.x + qrow * 256 + wave * QCH + hh * 4;
    f32x4 wa[QJ], xb[QJ];
#pragma unroll
    for (int j = 0; j < QJ; ++j) wa[j] = *reinterpret_cast<const f32x4*>(wrow + j * 8);
#pragma unroll
    for (int j = 0; j < QJ; ++j) xb[j] = *reinterpret_cast<const f32x4*>(xrow + j * 8);
    if constexpr (QP == 2) {
      const float* x2row = fz.x2 + qrow * 256 + wave * QCH + hh * 4;
      f32x4 x2b[QJ];
#pragma unroll
      for (int j = 0; j < QJ; ++j) x2b[j] = *reinterpret_cast<const f32x4*>(x2row + j * 8);
#pragma unroll
      for (int j = 0; j < QJ; ++j) xb[j] += x2b[j];
    }
#pragma unroll
    for (int j = 0; j < QJ; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) qacc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j][e], xb[j][e], qacc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) lds_o[wave][r][lane] = qacc[r];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = j * 4 + e;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < NS; ++w) sum += lds_o[w][r][lane];
        // D row r of this lane = head dim (r&3) + 8*(r>>2) + 4*hh = j*8 + hh*4 + e
        sum += bq4[j][e];
        qf[j][e] = sum * fz.qscale * 1.44269504088896340736f;
      }
    __syncthreads();   // lds_o is reused by the key-split merge below
    ATT_STAMP(1);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      qf[j] = q_ok ? *reinterpret_cast<const f32x4*>(q + qrow * ldq + head * ATT_HD + j * 8 + hh * 4) : z;
      // scores in the log2 domain: softmax(s) = 2^(s*log2e - max) / sum, so every exponential below is ONE v_exp_f32
      // (expf's range reduction is 13 VALU instructions per element: 8.1 instead of 9.6 us per call at one pair)
      qf[j] *= 1.44269504088896340736f;
    }
  }
  f32x16 oacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

#pragma unroll
  for (int kb = 0; kb < NBLK; ++kb) {
    float vf[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) vf[r] = vg[(size_t)(kb * 32 + (r & 3) + 8 * (r >> 2)) * ldkv];
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    // wavefronts of co-resident workgroups run in lock-step phases; giving the MFMA phases issue priority over the
    // other wavefronts' softmax VALU keeps the matrix pipe fed (222 -> 188 us at 32768 query rows)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j][e], qf[j][e], s, 0, 0, 0);
    if (kb + 1 < NBLK) {
#pragma unroll
      for (int j = 0; j < 4; ++j) kf[j] = *reinterpret_cast<const f32x4*>(kg + (size_t)(kb + 1) * 32 * ldkv + j * 8);
    }
    // s[r] = score(key = key0 + kb*32 + (r&3) + 8*(r>>2) + 4*hh, query = l31)
    __builtin_amdgcn_s_setprio(0);
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // first block: 2^(-inf) = 0
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
      psum += s[r];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
    // O^T[d][q] += sum_key V[key][d] * P[q][key]
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[r], s[r], oacc, 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  }
  l_run += __shfl_xor(l_run, 32);
  ATT_STAMP(2);

  // ---- merge the NS key splits -----------------------------------------------------------------
#pragma unroll
  for (int r = 0; r < 16; ++r) lds_o[wave][r][lane] = oacc[r];
  if (hh == 0) {
    lds_m[wave][l31] = m_run;
    lds_l[wave][l31] = l_run;
  }
  __syncthreads();
  float m_all = lds_m[0][l31];
#pragma unroll
  for (int w = 1; w < NS; ++w) m_all = fmaxf(m_all, lds_m[w][l31]);
  float f[NS];
  float l_all = 0.f;
#pragma unroll
  for (int w = 0; w < NS; ++w) {
    f[w] = __builtin_amdgcn_exp2f(lds_m[w][l31] - m_all);
    l_all += f[w] * lds_l[w][l31];
  }
  const float inv = 1.f / l_all;
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int r = wave * RPW + i;
    float acc = 0.f;
#pragma unroll
    for (int w = 0; w < NS; ++w) acc += f[w] * lds_o[w][r][lane];
    lds_out[l31][(r & 3) + 8 * (r >> 2) + 4 * hh] = acc * inv;
  }
  __syncthreads();
  ATT_STAMP(3);
  if constexpr (OP) {
    // partial[head][row][n] = sum_d O[row][d] * Wo[n][head*32 + d]: wave w -> output columns [64w, 64w + 64).  The accumulators
    // go through a wave-private LDS tile so that the partial rows leave as float4 (one instruction = 4 rows x 256 B) instead of
    // 32 scattered dword stores per lane
    float* stage = lds_buf + wave * (32 * SLD);   // aliases lds_o: every wave is past the merge (barrier above)
    f32x4 af[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const f32x4*>(&lds_out[l31][j * 8 + hh * 4]);
#pragma unroll
    for (int nb = 0; nb < NBO; ++nb) {
      f32x16 pacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) pacc[r] = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][e], wof[nb][j][e], pacc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * hh) * SLD + nb * 32 + l31] = pacc[r];
    }
    ATT_STAMP(4);
    float* pbase = fz.part + ((size_t)head * fz.rows_total + (size_t)pair * nq) * 256 + wave * (32 * NBO);
    constexpr int LPR = 8 * NBO;                  // lanes (float4) per staged row
    constexpr int RPI = 64 / LPR;                 // rows per wave instruction
    const int sr = lane / LPR, sc = (lane % LPR) * 4;
#pragma unroll
    for (int it = 0; it < 32 / RPI; ++it) {
      const int row = it * RPI + sr;
      const int qo = qtile * 32 + row;
      const f32x4 val = *reinterpret_cast<const f32x4*>(&stage[row * SLD + sc]);
      if (qo < nq) store_f32x4(pbase + (size_t)qo * 256 + sc, val, fz.wt != 0);
    }
    ATT_STAMP(5);
    if (fz.ct.state != nullptr) {
      __shared__ int coop_flags[2];
      const int rem = nq - qtile * 32;
      coop_tail_run(fz.ct, fz.part, (size_t)fz.rows_total * 256, pair * qtiles + qtile, pair * nq + qtile * 32, rem < 32 ? rem : 32, head, 8,
                    coop_flags);
    }
    if (o == nullptr) return;
  }
  for (int i = t; i < 256; i += NS * 64) {  // 32 rows x 128 B, one float4 per thread: coalesced row stores
    const int row = i >> 3, c4 = (i & 7) * 4;
    const int qo = qtile * 32 + row;
    if (qo < nq)
      *reinterpret_cast<f32x4*>(o + ((size_t)pair * nq + qo) * ldo + head * ATT_HD + c4) =
          *reinterpret_cast<const f32x4*>(&lds_out[row][c4]);
  }
}


// ---- many query rows (the batched engine calls, the dense pass) --------------------------------------------------------------
// Two 32-query tiles per wavefront (64 queries per workgroup): every K and V fragment a wavefront fetches feeds both tiles, which
// halves the L2 -> CU traffic and the load instructions per flop of the kernel above (its 128 KB of K_h/V_h per 32 queries is
// 4.6 TB/s out of L2 at 32768 rows), and the two tiles are independent MFMA chains: the softmax VALU of one tile sits next to
// the matrix instructions of the other in the instruction stream.  Same arithmetic per query as attention_kernel<NS, 0, false>
// (same key split, same merge order) - results are bit-identical to it.
// OCC = wavefronts per SIMD the register budget is set for: 2 keeps the Q fragments in registers (210 VGPRs), 3 parks them in LDS
// (8 KB per workgroup, re-read per key block) to fit 168
template <int NS, int OCC>
__global__ __launch_bounds__(NS * 64, OCC) void attention_wide_kernel(const float* __restrict__ q, int ldq,
                                                                 const float* __restrict__ k,
                                                                 const float* __restrict__ v, int ldkv,
                                                                 float* __restrict__ o, int ldo, int nq, int head_major) {
  constexpr int NBLK = ATT_KEYS / NS / 32;
  constexpr int RPW = 16 / NS;
  __shared__ __attribute__((aligned(16))) float lds_o[NS][2][16][64];
  __shared__ float lds_m[NS][2][32];
  __shared__ float lds_l[NS][2][32];
  __shared__ __attribute__((aligned(16))) float lds_out[64][36];
  constexpr bool QL = OCC >= 3;
  __shared__ __attribute__((aligned(16))) f32x4 lds_q[QL ? 2 * 4 * 64 : 1];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int qtiles = gridDim.x >> 3;
  const int head = head_major ? (blockIdx.x & 7) : (blockIdx.x / qtiles);
  const int qtile = head_major ? (blockIdx.x >> 3) : (blockIdx.x % qtiles);
  const int pair = blockIdx.z;

  const size_t key0 = (size_t)pair * ATT_KEYS + (size_t)wave * (ATT_KEYS / NS);
  const float* kg = k + (key0 + l31) * ldkv + head * ATT_HD + hh * 4;
  const float* vg = v + (key0 + 4 * hh) * ldkv + head * ATT_HD + l31;
  f32x4 kf[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) kf[j] = *reinterpret_cast<const f32x4*>(kg + j * 8);

  f32x4 qf[2][4];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int qi = qtile * 64 + u * 32 + l31;
    const bool q_ok = qi < nq;
    const size_t qrow = (size_t)pair * nq + (q_ok ? qi : 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      qf[u][j] = *reinterpret_cast<const f32x4*>(q + qrow * ldq + head * ATT_HD + j * 8 + hh * 4);
      qf[u][j] *= q_ok ? 1.44269504088896340736f : 0.f;   // log2 domain; rows past nq compute on zeros and are never stored
    }
  }
  if constexpr (QL) {
    if (wave == 0) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_q[(u * 4 + j) * 64 + lane] = qf[u][j];
    }
    __syncthreads();
  }
  f32x16 oacc[2];
  float m_run[2], l_run[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[u][r] = 0.f;
    m_run[u] = -INFINITY;
    l_run[u] = 0.f;
  }

#pragma unroll 1
  for (int kb = 0; kb < NBLK; ++kb) {
    float vf[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) vf[r] = vg[(size_t)(kb * 32 + (r & 3) + 8 * (r >> 2)) * ldkv];
    f32x16 s[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[u][r] = 0.f;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 qv = QL ? lds_q[(u * 4 + j) * 64 + lane] : qf[u][j];
#pragma unroll
        for (int e = 0; e < 4; ++e) s[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j][e], qv[e], s[u], 0, 0, 0);
      }
    if (kb + 1 < NBLK) {
#pragma unroll
      for (int j = 0; j < 4; ++j) kf[j] = *reinterpret_cast<const f32x4*>(kg + (size_t)(kb + 1) * 32 * ldkv + j * 8);
    }
    __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float mx = s[u][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[u][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run[u], mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run[u] - m_new);
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[u][r] = __builtin_amdgcn_exp2f(s[u][r] - m_new);
        psum += s[u][r];
      }
      l_run[u] = l_run[u] * alpha + psum;
      m_run[u] = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[u][r] *= alpha;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[r], s[u][r], oacc[u], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
  }

  // ---- merge the NS key splits (fixed order, as attention_kernel) -------------------------------------------------------------
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    l_run[u] += __shfl_xor(l_run[u], 32);
#pragma unroll
    for (int r = 0; r < 16; ++r) lds_o[wave][u][r][lane] = oacc[u][r];
    if (hh == 0) {
      lds_m[wave][u][l31] = m_run[u];
      lds_l[wave][u][l31] = l_run[u];
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    float m_all = lds_m[0][u][l31];
#pragma unroll
    for (int w = 1; w < NS; ++w) m_all = fmaxf(m_all, lds_m[w][u][l31]);
    float f[NS];
    float l_all = 0.f;
#pragma unroll
    for (int w = 0; w < NS; ++w) {
      f[w] = __builtin_amdgcn_exp2f(lds_m[w][u][l31] - m_all);
      l_all += f[w] * lds_l[w][u][l31];
    }
    const float inv = 1.f / l_all;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = wave * RPW + i;
      float acc = 0.f;
#pragma unroll
      for (int w = 0; w < NS; ++w) acc += f[w] * lds_o[w][u][r][lane];
      lds_out[u * 32 + l31][(r & 3) + 8 * (r >> 2) + 4 * hh] = acc * inv;
    }
  }
  __syncthreads();
  for (int i = t; i < 512; i += NS * 64) {  // 64 rows x 128 B, one float4 per thread
    const int row = i >> 3, c4 = (i & 7) * 4;
    const int qo = qtile * 64 + row;
    if (qo < nq)
      *reinterpret_cast<f32x4*>(o + ((size_t)pair * nq + qo) * ldo + head * ATT_HD + c4) =
          *reinterpret_cast<const f32x4*>(&lds_out[row][c4]);
  }
}

// ---- many query rows per (pair, head): K_h and V_h RESIDENT in LDS ---------------------------------------------------------------------
// Workgroup = (pair, head, chunk of up to 32 query tiles), 8 wavefronts (two per SIMD).  K_h and V_h of the head (512 x 32 floats each,
// 128 KB together, XOR-swizzled instead of padded) are loaded into LDS ONCE and every operand of the key loop comes from there: no
// global load, no barrier and no cross-wavefront merge in the steady state.  A wavefront owns a 32-query tile at a time and runs the
// FOUR key quarters as four independent online-softmax chains (the arithmetic of attention_kernel<4, ..> / attention_wide_kernel<4, ..>,
// where a wavefront ran one quarter each) merged in registers in the same order: results are bit-identical to those kernels, and the
// four chains are independent instruction streams - the softmax VALU of one sits under the matrix instructions of the others.  The
// output tile goes out as float4 stores straight from the D registers (rows = head dim: 4 consecutive dims per register quad).
//   32 pairs x 512 (encoder): attention_wide 94 us;  32 x 1000 (decoder): 171 us - numbers of this kernel in docs/LABNOTES.md 3
constexpr size_t ATT_RES_SMEM = (size_t)2 * ATT_KEYS * ATT_HD * sizeof(float);
__global__ __launch_bounds__(512) void attention_res_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                            const float* __restrict__ v, int ldkv, float* __restrict__ o, int ldo, int nq,
                                                            int tiles_per_chunk) {
  extern __shared__ __attribute__((aligned(16))) float res_smem[];
  float* k_s = res_smem;
  float* v_s = res_smem + ATT_KEYS * ATT_HD;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int head = blockIdx.x & 7, chunk = blockIdx.x >> 3;      // head-major: a head's workgroups land on one XCD (its L2 keeps K_h / V_h)
  const int pair = blockIdx.z;
  {
    const size_t krow0 = (size_t)pair * ATT_KEYS;
    const int c4 = t & 7;
#pragma unroll 4
    for (int r0 = 0; r0 < ATT_KEYS; r0 += 64) {
      const int row = r0 + (t >> 3);
      const f32x4 kk = *reinterpret_cast<const f32x4*>(k + (krow0 + row) * ldkv + head * ATT_HD + c4 * 4);
      const f32x4 vv = *reinterpret_cast<const f32x4*>(v + (krow0 + row) * ldkv + head * ATT_HD + c4 * 4);
      *reinterpret_cast<f32x4*>(k_s + row * ATT_HD + ((c4 ^ (row & 7)) << 2)) = kk;
      *reinterpret_cast<f32x4*>(v_s + row * ATT_HD + ((c4 ^ (row & 7)) << 2)) = vv;
    }
  }
  __syncthreads();
  const int ntiles = (nq + 31) / 32;
  const int t_end = (chunk + 1) * tiles_per_chunk < ntiles ? (chunk + 1) * tiles_per_chunk : ntiles;
  for (int qt = chunk * tiles_per_chunk + wave; qt < t_end; qt += 8) {
    const int qi = qt * 32 + l31;
    const bool q_ok = qi < nq;
    const size_t qrow = (size_t)pair * nq + (q_ok ? qi : 0);
    f32x4 qf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      qf[j] = *reinterpret_cast<const f32x4*>(q + qrow * ldq + head * ATT_HD + j * 8 + hh * 4);
      qf[j] *= q_ok ? 1.44269504088896340736f : 0.f;   // log2 domain; rows past nq compute on zeros and are never stored
    }
    f32x16 oacc[4];
    float m_run[4], l_run[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[c][r] = 0.f;
      m_run[c] = -INFINITY;
      l_run[c] = 0.f;
    }
#pragma unroll 1
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {                    // the four key quarters: independent chains
        const int key0 = c * 128 + kb * 32;
        const int krow = key0 + l31;
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 kf = *reinterpret_cast<const f32x4*>(k_s + krow * ATT_HD + (((j * 2 + hh) ^ (krow & 7)) << 2));
#pragma unroll
          for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[j][e], s, 0, 0, 0);
        }
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run[c], mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run[c] - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
          psum += s[r];
        }
        l_run[c] = l_run[c] * alpha + psum;
        m_run[c] = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[c][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int vr = key0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          const float vf = v_s[vr * ATT_HD + ((((l31 >> 2) ^ (vr & 7)) << 2) | (l31 & 3))];
          oacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, s[r], oacc[c], 0, 0, 0);
        }
      }
    }
    // merge of the four quarters, the arithmetic (and order) of attention_kernel's merge through LDS
#pragma unroll
    for (int c = 0; c < 4; ++c) l_run[c] += __shfl_xor(l_run[c], 32);
    float m_all = m_run[0];
#pragma unroll
    for (int c = 1; c < 4; ++c) m_all = fmaxf(m_all, m_run[c]);
    float f[4];
    float l_all = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f[c] = __builtin_amdgcn_exp2f(m_run[c] - m_all);
      l_all += f[c] * l_run[c];
    }
    const float inv = 1.f / l_all;
    if (q_ok) {
      float* dst = o + qrow * ldo + head * ATT_HD + 4 * hh;
#pragma unroll
      for (int g = 0; g < 4; ++g) {                    // registers 4g .. 4g+3 = head dims 8g + 4hh .. + 3
        f32x4 out;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float acc = 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) acc += f[c] * oacc[c][g * 4 + i];
          out[i] = acc * inv;
        }
        *reinterpret_cast<f32x4*>(dst + 8 * g) = out;
      }
    }
  }
}

// 3 wavefronts per SIMD measured 162-164 us at 32768 query rows against 168-193 for 2 (and 206 for the 32-query kernel).  An in-wave
// software pipeline pinned with sched_group_barrier (softmax of one tile between the MFMAs of the other) measured the same 164 us:
// hipcc honours the pattern for the score MFMAs only, and three wavefronts per SIMD already interleave the phases in hardware.
// knobs: KN_ATTENTION_WIDE_OCCUPANCY (3), KN_ATTENTION_RESIDENT (K_h / V_h resident in LDS from KN_ATTENTION_WIDE_MIN_ROWS rows and
// 256 queries per pair), KN_ATTENTION_WIDE_MIN_ROWS (4096: query rows of a launch from which the 64-query kernel is used)
static const int g_att_wide_head_major = 1;

static thread_local unsigned long long* g_att_dbg = nullptr;   // set_attention_debug_times
void set_attention_debug_times(unsigned long long* p) { g_att_dbg = p; }
// knobs: KN_ATTENTION_SPLITS (0 = automatic), KN_ATTENTION_FUSED_SPLITS (0 = 4; 48 / 84: encoder (q given) / decoder (q projected)
// separately), KN_XCD_MAPPING bit 3 = heads over XCDs (measured: -88 MB of fabric traffic per forward, +0.4 % time -> off)
static const int g_att_part_wt = 1;  // write-through stores for the out-projection partials

int init_attention_attributes() { return 0; }

int launch_attention(const float* q, int ldq, const float* k, const float* v, int ldkv, float* o, int ldo,
                     int nb, int nq, hipStream_t s) {
  if (nb <= 0 || nq <= 0) return 0;
  if (ldq % 4 || ldkv % 4 || ldo % 4) return -1;
  int ns = knob(KN_ATTENTION_SPLITS);
  const int g_att_head_major = (knob(KN_XCD_MAPPING) >> 3) & 1;
  const long g_att_wide_min_rows = knob(KN_ATTENTION_WIDE_MIN_ROWS);
  const int g_att_resident = knob(KN_ATTENTION_RESIDENT), g_att_wide_occ = knob(KN_ATTENTION_WIDE_OCCUPANCY) == 2 ? 2 : 3;
  // query tiles per workgroup: 32 (4 per wavefront) when that still gives every CU a workgroup, down to 8 (one per wavefront: K_h / V_h
  // are then fetched for 8 tiles only); fewer than 128 workgroups even so -> the 64-query kernel below
  const int res_tiles = (nq + 31) / 32;
  int res_tpc = 32;
  while (res_tpc > 8 && (long)((res_tiles + res_tpc - 1) / res_tpc) * 8 * nb < 256) res_tpc /= 2;
  const long res_wgs = (long)((res_tiles + res_tpc - 1) / res_tpc) * 8 * nb;
  if (ns == 0 && g_att_resident && (long)nb * nq >= g_att_wide_min_rows && nq >= 256 && res_wgs >= 128) {
    static PerDeviceFlag attr_set;
    if (!attr_set.get()) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(attention_res_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)ATT_RES_SMEM) != hipSuccess)
        return -2;
      attr_set.set();
    }
    const int chunks = (res_tiles + res_tpc - 1) / res_tpc;
    hipLaunchKernelGGL(attention_res_kernel, dim3(chunks * 8, 1, nb), dim3(512), ATT_RES_SMEM, s, q, ldq, k, v, ldkv, o, ldo, nq, res_tpc);
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  if (ns == 0 && (long)nb * nq >= g_att_wide_min_rows) {
    dim3 wgrid(((nq + 63) / 64) * 8, 1, nb);
    // head-major workgroup order here (head = XCD): with many pairs in flight every XCD then keeps one head's K/V slice of each
    // pair in its L2 instead of all eight (32 pairs x 1000 queries: 176 -> 163 us; one pair: no difference)
    if (g_att_wide_occ == 3)
      hipLaunchKernelGGL((attention_wide_kernel<4, 3>), wgrid, dim3(256), 0, s, q, ldq, k, v, ldkv, o, ldo, nq, g_att_wide_head_major);
    else
      hipLaunchKernelGGL((attention_wide_kernel<4, 2>), wgrid, dim3(256), 0, s, q, ldq, k, v, ldkv, o, ldo, nq, g_att_wide_head_major);
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  dim3 grid(((nq + 31) / 32) * 8, 1, nb);
  AttnFuse fz = {};
  if (ns == 0) ns = 4;
  switch (ns) {
    case 1:
      hipLaunchKernelGGL((attention_kernel<1, 0, false>), grid, dim3(64), 0, s, q, ldq, k, v, ldkv, o, ldo, nq, g_att_head_major, fz);
      break;
    case 2:
      hipLaunchKernelGGL((attention_kernel<2, 0, false>), grid, dim3(128), 0, s, q, ldq, k, v, ldkv, o, ldo, nq, g_att_head_major, fz);
      break;
    case 4:
      hipLaunchKernelGGL((attention_kernel<4, 0, false>), grid, dim3(256), 0, s, q, ldq, k, v, ldkv, o, ldo, nq, g_att_head_major, fz);
      break;
    case 8:
      hipLaunchKernelGGL((attention_kernel<8, 0, false>), grid, dim3(512), 0, s, q, ldq, k, v, ldkv, o, ldo, nq, g_att_head_major, fz);
      break;
    case 16:
      hipLaunchKernelGGL((attention_kernel<16, 0, false>), grid, dim3(1024), 0, s, q, ldq, k, v, ldkv, o, ldo, nq, g_att_head_major, fz);
      break;
    default:
      return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Attention with the q projection in the prologue (qp: x/x2/wq/bq/qscale, q unused) and / or the output projection in the
// epilogue (op: partial outputs [8][nb*nq][256] to `part`, o may be nullptr).  4 key splits.
#define ATT_CT_PARAM , const CoopTail* ct
static int attention_fused_impl(const float* q, int ldq, const float* x, const float* x2, const float* wq, const float* bq,
                                float qscale, const float* k, const float* v, int ldkv, float* o, int ldo, const float* wo,
                                float* part, int nb, int nq, hipStream_t s ATT_CT_PARAM) {
  if (nb <= 0 || nq <= 0) return 0;
  const bool qp = wq != nullptr, op = wo != nullptr;
  if (ct != nullptr && !op) return -1;
  const int g_att_head_major = (knob(KN_XCD_MAPPING) >> 3) & 1;
  const int g_att_fused_splits = knob(KN_ATTENTION_FUSED_SPLITS);
  if (ldkv % 4 || (o && ldo % 4) || (!qp && (q == nullptr || ldq % 4))) return -1;
  if (qp && (bq == nullptr || (x == nullptr && x2 == nullptr))) return -1;
  if (op && part == nullptr) return -1;
  if (!op && o == nullptr) return -1;
  dim3 grid(((nq + 31) / 32) * 8, 1, nb);
  AttnFuse fz = {};
  fz.x = x ? x : x2; fz.x2 = x ? x2 : nullptr; fz.wq = wq; fz.bq = bq; fz.qscale = qscale;
  fz.wo = wo; fz.part = part; fz.rows_total = nb * nq; fz.wt = g_att_part_wt;
  fz.dbg = g_att_dbg;
  if (ct != nullptr) fz.ct = *ct;
  const int qmode = !qp ? 0 : (fz.x2 ? 2 : 1);
  // 8 key splits (8 wavefronts per workgroup) were tried where 4 leave CUs without a workgroup (the encoder of one pair is
  // 16 query tiles x 8 heads = 128 workgroups on 256 CUs): measured 0.978 vs 0.973 ms per forward, the merge of 8 partial
  // softmaxes and the narrower out-projection blocks cost more than the idle CUs -- kept as a knob only
  const int ns = g_att_fused_splits == 48 ? (qp ? 8 : 4) : g_att_fused_splits == 84 ? (qp ? 4 : 8) : g_att_fused_splits ? g_att_fused_splits : 4;
#define ATT_LAUNCH(NSV, QPV, OPV)                                                                                         \
  hipLaunchKernelGGL((attention_kernel<NSV, QPV, OPV>), grid, dim3(NSV * 64), 0, s, q, ldq, k, v, ldkv, o, ldo, nq, g_att_head_major, fz)
#define ATT_PICK(NSV)                              \
  do {                                             \
    if (qmode == 2 && op) ATT_LAUNCH(NSV, 2, true);   \
    else if (qmode == 2) ATT_LAUNCH(NSV, 2, false);   \
    else if (qmode == 1 && op) ATT_LAUNCH(NSV, 1, true); \
    else if (qmode == 1) ATT_LAUNCH(NSV, 1, false);   \
    else if (op) ATT_LAUNCH(NSV, 0, true);            \
    else ATT_LAUNCH(NSV, 0, false);                   \
  } while (0)
  if (ns == 8) ATT_PICK(8);
  else ATT_PICK(4);
#undef ATT_PICK
#undef ATT_LAUNCH
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_attention_fused(const float* q, int ldq, const float* x, const float* x2, const float* wq, const float* bq,
                           float qscale, const float* k, const float* v, int ldkv, float* o, int ldo, const float* wo,
                           float* part, int nb, int nq, hipStream_t s) {
  return attention_fused_impl(q, ldq, x, x2, wq, bq, qscale, k, v, ldkv, o, ldo, wo, part, nb, nq, s, nullptr);
}
// ... with the cooperative tail (experimental/coop_tail.h): the 8 head workgroups of a query tile finish the tile themselves
int launch_attention_fused_coop(const float* q, int ldq, const float* x, const float* x2, const float* wq, const float* bq,
                                float qscale, const float* k, const float* v, int ldkv, float* o, int ldo, const float* wo,
                                float* part, int nb, int nq, hipStream_t s, const CoopTail* ct) {
  return attention_fused_impl(q, ldq, x, x2, wq, bq, qscale, k, v, ldkv, o, ldo, wo, part, nb, nq, s, ct);
}
