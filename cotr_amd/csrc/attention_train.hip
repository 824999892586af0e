// Multi-head attention for the TRAINING step, forward and backward, fp32 MFMA, gfx950.
//
// nn.MultiheadAttention as the reference uses it (COTR/models/transformer.py:127,149-153 self-attention of the encoder,
// :167,192-195 cross-attention of the decoder; 8 heads of 32, 512 keys): o = dropout(softmax(q k^T * hd^-0.5)) v per head.
// The forward is the inference kernel's scheme (attention.hip: one workgroup = 32 queries x 1 head x 1 pair, its 4
// wavefronts split the 512 keys, S^T = K Q^T so that a lane owns one query, online softmax in the log2 domain) with the
// dropout mask applied to the probabilities after the normaliser is accumulated, and the log-sum-exp of every (query, head)
// written out.  The backward recomputes the probabilities from it (nothing of size queries x keys is ever stored):
//     dV = P~^T dO        dP~ = dO V^T       dP = dP~ * mask / (1-p)      dS = P * (dP - delta),  delta = rowsum(dO * O)
//     dQ = dS K * scale   dK = dS^T (Q * scale)
// in two deterministic kernels (no atomics): attn_bwd_dq (a workgroup per 32 queries, wavefronts split the keys, like the
// forward) and attn_bwd_dkv (a workgroup per 32 keys, wavefronts split the query blocks).  Every matrix product is
// v_mfma_f32_32x32x2_f32 with the operand layouts of the forward kernel: the "swapped" products keep the softmax row (or the
// key column) lane-local, and their D registers feed the next product's B operand directly.
#include "common.h"
#include "train.h"

#define ATT_KEYS 512
#define ATT_HD 32
#define LOG2E 1.44269504088896340736f

namespace {

__device__ __forceinline__ uint64_t mask_index(int pair, int head, int nq, int qi, int key) {
  return (((uint64_t)(pair * 8 + head) * nq + qi) * ATT_KEYS + key);
}
// train_keep(seed, mask_index(pair, head, nq, qi, key), thresh) with the part that depends on the ROW (pair, head, query) hoisted: the
// index is base + key with base a multiple of 512 and key < 512, so its low word is base_lo | key and its high word - which enters
// the hash through one additive constant only - is the row's.  Same decisions bit for bit (the first-form kernels and the other
// training kernels keep calling train_keep), two 32-bit multiplies per probability instead of a 64-bit index and four.
struct MaskRow {
  uint32_t lo, c;
};
__device__ __forceinline__ uint32_t mask_hi_const(uint32_t seed, uint32_t hi) {
  return (seed * 0x9E3779B9u) ^ ((hi ^ seed) * 0x85EBCA6Bu + 0xC2B2AE35u);
}
__device__ __forceinline__ MaskRow mask_row(uint32_t seed, int pair, int head, int nq, int qi) {
  const uint64_t base = mask_index(pair, head, nq, qi, 0);
  return MaskRow{(uint32_t)base, mask_hi_const(seed, (uint32_t)(base >> 32))};
}
__device__ __forceinline__ bool mask_keep(MaskRow m, int key, uint32_t thresh) {
  uint32_t x = (m.lo | (uint32_t)key) ^ m.c;
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x >= thresh;
}
// the rows of one query TILE (32 consecutive queries from q0): row ql's base is the tile's (wave-uniform: scalar arithmetic) + ql * 512
struct MaskTile {
  uint32_t lo, c0, c1;
  __device__ __forceinline__ MaskRow row(int ql) const {
    const uint32_t l = lo + ((uint32_t)ql << 9);
    return MaskRow{l, l < lo ? c1 : c0};
  }
};
__device__ __forceinline__ MaskTile mask_tile(uint32_t seed, int pair, int head, int nq, int q0) {
  const uint64_t base = mask_index(pair, head, nq, q0, 0);
  const uint32_t hi = (uint32_t)(base >> 32);
  return MaskTile{(uint32_t)base, mask_hi_const(seed, hi), mask_hi_const(seed, hi + 1)};
}

// -------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_train_fwd_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                             int ldk, const float* __restrict__ v, int ldv, float* __restrict__ o,
                                                             int ldo, float* __restrict__ lse, int nq, float qscale,
                                                             uint32_t thresh, float inv_keep, uint32_t seed,
                                                             const uint32_t* __restrict__ salt) {
  seed = train_salted(seed, salt);
  constexpr int NS = 4, NBLK = ATT_KEYS / NS / 32;
  __shared__ float lds_o[NS][16][64];
  __shared__ float lds_m[NS][32];
  __shared__ float lds_l[NS][32];
  __shared__ __attribute__((aligned(16))) float lds_out[32][36];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int qtiles = gridDim.x >> 3;
  const int head = blockIdx.x / qtiles, qtile = blockIdx.x % qtiles;
  const int pair = blockIdx.z;
  const int qi = qtile * 32 + l31;
  const bool q_ok = qi < nq;
  const size_t qrow = (size_t)pair * nq + (q_ok ? qi : 0);

  f32x4 qf[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    qf[j] = *reinterpret_cast<const f32x4*>(q + qrow * ldq + head * ATT_HD + j * 8 + hh * 4);
    qf[j] *= qscale * LOG2E;
  }
  const int key_w = wave * (ATT_KEYS / NS);
  const size_t key0 = (size_t)pair * ATT_KEYS + key_w;
  const float* kg = k + (key0 + l31) * ldk + head * ATT_HD + hh * 4;
  const float* vg = v + (key0 + 4 * hh) * ldv + head * ATT_HD + l31;
  f32x16 oacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
#pragma unroll
  for (int kb = 0; kb < NBLK; ++kb) {
    f32x4 kf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kf[j] = *reinterpret_cast<const f32x4*>(kg + (size_t)kb * 32 * ldk + j * 8);
    float vf[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) vf[r] = vg[(size_t)(kb * 32 + (r & 3) + 8 * (r >> 2)) * ldv];
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j][e], qf[j][e], s, 0, 0, 0);
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
      psum += s[r];                                   // the normaliser sums the UN-dropped probabilities
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    if (thresh != 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = key_w + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        s[r] = train_keep(seed, mask_index(pair, head, nq, q_ok ? qi : 0, key), thresh) ? s[r] * inv_keep : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[r], s[r], oacc, 0, 0, 0);
  }
  l_run += __shfl_xor(l_run, 32);
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
  {
    const int r = wave;   // NS == 4: accumulator rows 4*wave .. 4*wave+3
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = r * 4 + i;
      float acc = 0.f;
#pragma unroll
      for (int w = 0; w < NS; ++w) acc += f[w] * lds_o[w][rr][lane];
      lds_out[l31][(rr & 3) + 8 * (rr >> 2) + 4 * hh] = acc * inv;
    }
  }
  if (wave == 0 && hh == 0 && q_ok) lse[qrow * 8 + head] = m_all + __builtin_amdgcn_logf(l_all);   // v_log_f32 = log2
  __syncthreads();
  {
    const int row = t >> 3, c4 = (t & 7) * 4;
    const int qo = qtile * 32 + row;
    if (qo < nq)
      *reinterpret_cast<f32x4*>(o + ((size_t)pair * nq + qo) * ldo + head * ATT_HD + c4) =
          *reinterpret_cast<const f32x4*>(&lds_out[row][c4]);
  }
}

// delta[row][head] = sum_d dO[row][head*32 + d] * O[row][head*32 + d]: one wavefront per row, 8 lanes per head
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ o, const float* __restrict__ d_o, int ldo,
                                                         float* __restrict__ delta, int rows) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const f32x4 a = *reinterpret_cast<const f32x4*>(o + (size_t)row * ldo + lane * 4);
  const f32x4 b = *reinterpret_cast<const f32x4*>(d_o + (size_t)row * ldo + lane * 4);
  float sacc = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  sacc += __shfl_xor(sacc, 1);
  sacc += __shfl_xor(sacc, 2);
  sacc += __shfl_xor(sacc, 4);
  if ((lane & 7) == 0) delta[(size_t)row * 8 + (lane >> 3)] = sacc;
}

// -------------------------------------------------------------------------------------------------------------------
// dQ: workgroup = 32 queries x head x pair, the 4 wavefronts split the keys
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                          int ldk, const float* __restrict__ v, int ldv, const float* __restrict__ d_o,
                                                          int ldo, const float* __restrict__ lse, const float* __restrict__ delta,
                                                          float* __restrict__ dq, int lddq, int nq, float qscale, uint32_t thresh,
                                                          float inv_keep, uint32_t seed, const uint32_t* __restrict__ salt) {
  seed = train_salted(seed, salt);
  constexpr int NS = 4, NBLK = ATT_KEYS / NS / 32;
  __shared__ float lds_o[NS][16][64];
  __shared__ __attribute__((aligned(16))) float lds_out[32][36];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int qtiles = gridDim.x >> 3;
  const int head = blockIdx.x / qtiles, qtile = blockIdx.x % qtiles;
  const int pair = blockIdx.z;
  const int qi = qtile * 32 + l31;
  const bool q_ok = qi < nq;
  const size_t qrow = (size_t)pair * nq + (q_ok ? qi : 0);
  f32x4 qf[4], dof[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    qf[j] = *reinterpret_cast<const f32x4*>(q + qrow * ldq + head * ATT_HD + j * 8 + hh * 4);
    qf[j] *= qscale * LOG2E;
    dof[j] = *reinterpret_cast<const f32x4*>(d_o + qrow * ldo + head * ATT_HD + j * 8 + hh * 4);
  }
  const float lse_q = lse[qrow * 8 + head], delta_q = delta[qrow * 8 + head];
  const int key_w = wave * (ATT_KEYS / NS);
  const size_t key0 = (size_t)pair * ATT_KEYS + key_w;
  const float* kg = k + (key0 + l31) * ldk + head * ATT_HD + hh * 4;       // A operand of S^T: lane = key
  const float* vga = v + (key0 + l31) * ldv + head * ATT_HD + hh * 4;      // A operand of dP^T = V dO^T: lane = key
  const float* kt = k + (key0 + 4 * hh) * ldk + head * ATT_HD + l31;       // A operand of dQ^T = K^T dS^T: lane = head dim
  f32x16 dqacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) dqacc[r] = 0.f;
#pragma unroll
  for (int kb = 0; kb < NBLK; ++kb) {
    f32x4 kf[4], vf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      kf[j] = *reinterpret_cast<const f32x4*>(kg + (size_t)kb * 32 * ldk + j * 8);
      vf[j] = *reinterpret_cast<const f32x4*>(vga + (size_t)kb * 32 * ldv + j * 8);
    }
    float ktf[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ktf[r] = kt[(size_t)(kb * 32 + (r & 3) + 8 * (r >> 2)) * ldk];
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j][e], qf[j][e], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[j][e], dof[j][e], dp, 0, 0, 0);
      }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(s[r] - lse_q);
      float g = dp[r];
      if (thresh != 0) {
        const int key = key_w + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        g = train_keep(seed, mask_index(pair, head, nq, q_ok ? qi : 0, key), thresh) ? g * inv_keep : 0.f;
      }
      s[r] = q_ok ? p * (g - delta_q) : 0.f;          // dS[q][key]
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) dqacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ktf[r], s[r], dqacc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) lds_o[wave][r][lane] = dqacc[r];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rr = wave * 4 + i;
    float acc = 0.f;
#pragma unroll
    for (int w = 0; w < NS; ++w) acc += lds_o[w][rr][lane];
    lds_out[l31][(rr & 3) + 8 * (rr >> 2) + 4 * hh] = acc * qscale;   // q_eff = q * qscale
  }
  __syncthreads();
  {
    const int row = t >> 3, c4 = (t & 7) * 4;
    const int qo = qtile * 32 + row;
    if (qo < nq)
      *reinterpret_cast<f32x4*>(dq + ((size_t)pair * nq + qo) * lddq + head * ATT_HD + c4) =
          *reinterpret_cast<const f32x4*>(&lds_out[row][c4]);
  }
}

// -------------------------------------------------------------------------------------------------------------------
// dK, dV: workgroup = 32 keys x head x pair, the 4 wavefronts take every 4th block of 32 queries
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                           int ldk, const float* __restrict__ v, int ldv, const float* __restrict__ d_o,
                                                           int ldo, const float* __restrict__ lse, const float* __restrict__ delta,
                                                           float* __restrict__ dk, int lddk, float* __restrict__ dv, int lddv, int nq,
                                                           float qscale, uint32_t thresh, float inv_keep, uint32_t seed,
                                                           const uint32_t* __restrict__ salt) {
  seed = train_salted(seed, salt);
  constexpr int NS = 4;
  __shared__ float lds_k[NS][16][64];
  __shared__ float lds_v[NS][16][64];
  __shared__ __attribute__((aligned(16))) float lds_out[2][32][36];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int head = blockIdx.x >> 4, ktile = blockIdx.x & 15;     // 16 key tiles of 32
  const int pair = blockIdx.z;
  const int kj = ktile * 32 + l31;
  const size_t krow = (size_t)pair * ATT_KEYS + kj;
  f32x4 kfb[4], vfb[4];                                          // B operands: lane = key
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    kfb[j] = *reinterpret_cast<const f32x4*>(k + krow * ldk + head * ATT_HD + j * 8 + hh * 4);
    vfb[j] = *reinterpret_cast<const f32x4*>(v + krow * ldv + head * ATT_HD + j * 8 + hh * 4);
  }
  f32x16 dkacc, dvacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) dkacc[r] = dvacc[r] = 0.f;
  const int nqb = (nq + 31) / 32;
  for (int qb = wave; qb < nqb; qb += NS) {
    const int qa = qb * 32 + l31;                                // A-operand row of this lane (S^T, dP^T products)
    const size_t qa_row = (size_t)pair * nq + (qa < nq ? qa : 0);
    f32x4 qaf[4], doaf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      qaf[j] = *reinterpret_cast<const f32x4*>(q + qa_row * ldq + head * ATT_HD + j * 8 + hh * 4);
      qaf[j] *= qscale * LOG2E;
      doaf[j] = *reinterpret_cast<const f32x4*>(d_o + qa_row * ldo + head * ATT_HD + j * 8 + hh * 4);
    }
    // rows of the D registers: query qr(r) = qb*32 + (r&3) + 8*(r>>2) + 4*hh
    float qtf[16], dotf[16], lse_r[16], del_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qr = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      const size_t row = (size_t)pair * nq + (qr < nq ? qr : 0);
      qtf[r] = q[row * ldq + head * ATT_HD + l31] * qscale;      // A operand of dK^T = Q^T dS: lane = head dim
      dotf[r] = d_o[row * ldo + head * ATT_HD + l31];            // A operand of dV^T = dO^T P~
      lse_r[r] = lse[row * 8 + head];
      del_r[r] = delta[row * 8 + head];
    }
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(qaf[j][e], kfb[j][e], s, 0, 0, 0);     // S[q][key], lane = key
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(doaf[j][e], vfb[j][e], dp, 0, 0, 0);  // dP~[q][key]
      }
    f32x16 pt;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qr = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      const bool ok = qr < nq;
      const float p = ok ? __builtin_amdgcn_exp2f(s[r] - lse_r[r]) : 0.f;
      float keep = 1.f;
      if (thresh != 0) keep = train_keep(seed, mask_index(pair, head, nq, ok ? qr : 0, kj), thresh) ? inv_keep : 0.f;
      pt[r] = p * keep;                                          // P~[q][key]
      s[r] = p * (dp[r] * keep - del_r[r]);                      // dS[q][key]
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dvacc = __builtin_amdgcn_mfma_f32_32x32x2f32(dotf[r], pt[r], dvacc, 0, 0, 0);
      dkacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qtf[r], s[r], dkacc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    lds_k[wave][r][lane] = dkacc[r];
    lds_v[wave][r][lane] = dvacc[r];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rr = wave * 4 + i;
    float ak = 0.f, av = 0.f;
#pragma unroll
    for (int w = 0; w < NS; ++w) {
      ak += lds_k[w][rr][lane];
      av += lds_v[w][rr][lane];
    }
    const int d = (rr & 3) + 8 * (rr >> 2) + 4 * hh;            // D rows = head dim, column (lane & 31) = key
    lds_out[0][l31][d] = ak;
    lds_out[1][l31][d] = av;
  }
  __syncthreads();
  {
    const int row = t >> 3, c4 = (t & 7) * 4;
    const size_t orow = (size_t)pair * ATT_KEYS + ktile * 32 + row;
    *reinterpret_cast<f32x4*>(dk + orow * lddk + head * ATT_HD + c4) = *reinterpret_cast<const f32x4*>(&lds_out[0][row][c4]);
    *reinterpret_cast<f32x4*>(dv + orow * lddv + head * ATT_HD + c4) = *reinterpret_cast<const f32x4*>(&lds_out[1][row][c4]);
  }
}


// -------------------------------------------------------------------------------------------------------------------
// dK, dV, second form (the one launched): workgroup = 128 keys x head x pair, each wavefront OWNS 32 keys and walks ALL query
// tiles, so there is no cross-wavefront reduction at the end; the four wavefronts share every query tile - Q, dO, lse, delta of 32
// queries go through LDS once per workgroup (global -> registers one tile ahead -> LDS, two buffers, ONE barrier per tile) instead
// of once per wavefront from global memory.  Same products in the same order per (key, head dim) as the first form except that a
// key's sum over the query tiles is now one running accumulation (first form: four partial sums, one per wavefront, added at the end).
//   encoder 32 pairs x 512 queries: 1024 workgroups x 16 tiles;  decoder 16 x 200: 512 workgroups x 7 tiles (the first form gave a
//   wavefront 1-2 tiles and then reduced through LDS).
// -------------------------------------------------------------------------------------------------------------------
template <bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_dkv2_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                            int ldk, const float* __restrict__ v, int ldv, const float* __restrict__ d_o,
                                                            int ldo, const float* __restrict__ lse, const float* __restrict__ delta,
                                                            float* __restrict__ dk, int lddk, float* __restrict__ dv, int lddv, int nq,
                                                            float qscale, uint32_t thresh, float inv_keep, uint32_t seed,
                                                            const uint32_t* __restrict__ salt) {
  seed = train_salted(seed, salt);
  constexpr int TS = 36;                                          // padded tile row (floats)
  __shared__ __attribute__((aligned(16))) float q_s[2][32][TS];   // raw q rows of the tile, this head's 32 columns
  __shared__ __attribute__((aligned(16))) float do_s[2][32][TS];
  __shared__ float ld_s[2][2][32];                                // lse | delta of the tile's queries
  __shared__ __attribute__((aligned(16))) float out_s[4][32][TS]; // per wavefront: D tile -> rows of dk / dv
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int head = blockIdx.x >> 2, kgroup = blockIdx.x & 3;
  const int pair = blockIdx.z;
  const int key0 = kgroup * 128 + wave * 32;
  const int kj = key0 + l31;
  const size_t krow = (size_t)pair * ATT_KEYS + kj;
  f32x4 kfb[4], vfb[4];                                          // B operands: lane = key
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    kfb[j] = *reinterpret_cast<const f32x4*>(k + krow * ldk + head * ATT_HD + j * 8 + hh * 4);
    vfb[j] = *reinterpret_cast<const f32x4*>(v + krow * ldv + head * ATT_HD + j * 8 + hh * 4);
  }
  f32x16 dkacc, dvacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) dkacc[r] = dvacc[r] = 0.f;
  const int nqb = (nq + 31) / 32;
  // tile staging: thread -> (row t >> 3, 4 columns (t & 7) * 4) of Q and dO; threads 0-31 lse, 32-63 delta
  const int srow = t >> 3, sc4 = (t & 7) * 4;
  f32x4 pq, pdo;
  float pld = 0.f;
  auto fetch = [&](int qb) {
    const int qa = qb * 32 + srow;
    const size_t row = (size_t)pair * nq + (qa < nq ? qa : 0);
    pq = *reinterpret_cast<const f32x4*>(q + row * ldq + head * ATT_HD + sc4);
    pdo = *reinterpret_cast<const f32x4*>(d_o + row * ldo + head * ATT_HD + sc4);
    if (t < 64) {
      const int ql = qb * 32 + l31;
      const size_t r2 = (size_t)pair * nq + (ql < nq ? ql : 0);
      pld = (hh == 0 ? lse : delta)[r2 * 8 + head];
    }
  };
  auto stash = [&](int buf) {
    *reinterpret_cast<f32x4*>(&q_s[buf][srow][sc4]) = pq;
    *reinterpret_cast<f32x4*>(&do_s[buf][srow][sc4]) = pdo;
    if (t < 64) ld_s[buf][hh][l31] = pld;
  };
  fetch(0);
  stash(0);
  for (int qb = 0; qb < nqb; ++qb) {
    const int buf = qb & 1;
    __syncthreads();                                             // tile qb is in q_s[buf]; everybody is done with the other buffer
    if (qb + 1 < nqb) fetch(qb + 1);                             // (global loads in flight under this tile's products)
    f32x4 qaf[4], doaf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      qaf[j] = *reinterpret_cast<const f32x4*>(&q_s[buf][l31][j * 8 + hh * 4]);
      qaf[j] *= qscale * LOG2E;
      doaf[j] = *reinterpret_cast<const f32x4*>(&do_s[buf][l31][j * 8 + hh * 4]);
    }
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(qaf[j][e], kfb[j][e], s, 0, 0, 0);     // S[q][key], lane = key
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(doaf[j][e], vfb[j][e], dp, 0, 0, 0);  // dP~[q][key]
      }
    f32x16 pt;
    const MaskTile mt = mask_tile(seed, pair, head, nq, qb * 32);
#pragma unroll
    for (int r = 0; r < 16; ++r) {                               // (branch-free: one basic block for the scheduler to interleave)
      const int ql = (r & 3) + 8 * (r >> 2) + 4 * hh;            // row of D register r: query qb*32 + ql
      const float e = __builtin_amdgcn_exp2f(s[r] - ld_s[buf][0][ql]);
      const float p = qb * 32 + ql < nq ? e : 0.f;
      float keep = 1.f;
      if (DROP) keep = mask_keep(mt.row(ql), kj, thresh) ? inv_keep : 0.f;
      pt[r] = p * keep;                                          // P~[q][key]
      s[r] = p * (dp[r] * keep - ld_s[buf][1][ql]);              // dS[q][key]
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ql = (r & 3) + 8 * (r >> 2) + 4 * hh;
      const float dot = do_s[buf][ql][l31];                      // A operand of dV^T = dO^T P~: lane = head dim
      const float qt = q_s[buf][ql][l31] * qscale;               // A operand of dK^T = Q^T dS
      dvacc = __builtin_amdgcn_mfma_f32_32x32x2f32(dot, pt[r], dvacc, 0, 0, 0);
      dkacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qt, s[r], dkacc, 0, 0, 0);
    }
    if (qb + 1 < nqb) stash(buf ^ 1);
  }
  // D rows = head dim, column (lane & 31) = key: through this wavefront's LDS tile to rows of dk, then of dv
  const size_t orow = (size_t)pair * ATT_KEYS + key0 + (lane >> 1);
  const int oc = (lane & 1) * 16;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
#pragma unroll
    for (int r = 0; r < 16; ++r) out_s[wave][l31][(r & 3) + 8 * (r >> 2) + 4 * hh] = which == 0 ? dkacc[r] : dvacc[r];
    __builtin_amdgcn_wave_barrier();
    float* dst = (which == 0 ? dk + orow * lddk : dv + orow * lddv) + head * ATT_HD + oc;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<f32x4*>(dst + i * 4) = *reinterpret_cast<const f32x4*>(&out_s[wave][lane >> 1][oc + i * 4]);
    __builtin_amdgcn_wave_barrier();
  }
}


// -------------------------------------------------------------------------------------------------------------------
// Forward and dQ, second form: workgroup = 4 query tiles x head x pair, each wavefront OWNS 32 queries and walks all 16 key tiles
// (no cross-wavefront combine of partial softmaxes / partial dQ at the end); the K and V tiles are shared by the four wavefronts
// through LDS (global -> registers one tile ahead -> LDS, two buffers, one barrier per tile).  A wavefront whose tile lies past
// nq only helps with the staging.
// -------------------------------------------------------------------------------------------------------------------
struct KvStage {
  static constexpr int TS = 36;
  f32x4 pk, pv;
  __device__ __forceinline__ void fetch(const float* k, int ldk, const float* v, int ldv, size_t row0, int head, int t) {
    const size_t row = row0 + (t >> 3);
    pk = *reinterpret_cast<const f32x4*>(k + row * ldk + head * ATT_HD + (t & 7) * 4);
    pv = *reinterpret_cast<const f32x4*>(v + row * ldv + head * ATT_HD + (t & 7) * 4);
  }
  __device__ __forceinline__ void stash(float (*ks)[TS], float (*vs)[TS], int t) const {
    *reinterpret_cast<f32x4*>(&ks[t >> 3][(t & 7) * 4]) = pk;
    *reinterpret_cast<f32x4*>(&vs[t >> 3][(t & 7) * 4]) = pv;
  }
};

template <bool DROP>
__global__ __launch_bounds__(256) void attn_train_fwd2_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                              int ldk, const float* __restrict__ v, int ldv, float* __restrict__ o,
                                                              int ldo, float* __restrict__ lse, int nq, int qgroups, float qscale,
                                                              uint32_t thresh, float inv_keep, uint32_t seed,
                                                              const uint32_t* __restrict__ salt) {
  seed = train_salted(seed, salt);
  constexpr int TS = KvStage::TS;
  __shared__ __attribute__((aligned(16))) float k_s[2][32][TS];
  __shared__ __attribute__((aligned(16))) float v_s[2][32][TS];
  __shared__ __attribute__((aligned(16))) float out_s[4][32][TS];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int head = blockIdx.x / qgroups, qtile = (blockIdx.x % qgroups) * 4 + wave;
  const int pair = blockIdx.z;
  const bool active = qtile * 32 < nq;                           // (wave-uniform)
  const int qi = qtile * 32 + l31;
  const bool q_ok = qi < nq;
  const size_t qrow = (size_t)pair * nq + (q_ok ? qi : 0);
  f32x4 qf[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    qf[j] = *reinterpret_cast<const f32x4*>(q + qrow * ldq + head * ATT_HD + j * 8 + hh * 4);
    qf[j] *= qscale * LOG2E;
  }
  const MaskRow mrow = mask_row(seed, pair, head, nq, q_ok ? qi : 0);
  f32x16 oacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const size_t key_row0 = (size_t)pair * ATT_KEYS;
  KvStage st;
  st.fetch(k, ldk, v, ldv, key_row0, head, t);
  st.stash(k_s[0], v_s[0], t);
  constexpr int NKB = ATT_KEYS / 32;
  for (int kb = 0; kb < NKB; ++kb) {
    const int buf = kb & 1;
    __syncthreads();
    if (kb + 1 < NKB) st.fetch(k, ldk, v, ldv, key_row0 + (kb + 1) * 32, head, t);
    if (active) {
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(&k_s[buf][l31][j * 8 + hh * 4]);      // A operand of S^T: lane = key
#pragma unroll
        for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[j][e], s, 0, 0, 0);
      }
      float mx = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
        psum += s[r];                                   // the normaliser sums the UN-dropped probabilities
      }
      l_run = l_run * alpha + psum;
      m_run = m_new;
      if (DROP) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          s[r] = mask_keep(mrow, key, thresh) ? s[r] * inv_keep : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float vf = v_s[buf][(r & 3) + 8 * (r >> 2) + 4 * hh][l31];      // A operand of O^T = V^T P^T: lane = head dim
        oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, s[r], oacc, 0, 0, 0);
      }
    }
    if (kb + 1 < NKB) st.stash(k_s[buf ^ 1], v_s[buf ^ 1], t);
  }
  if (!active) return;
  l_run += __shfl_xor(l_run, 32);
  const float inv = 1.f / l_run;
  // D rows = head dim, column (lane & 31) = query: through this wavefront's LDS tile to rows of o
#pragma unroll
  for (int r = 0; r < 16; ++r) out_s[wave][l31][(r & 3) + 8 * (r >> 2) + 4 * hh] = oacc[r] * inv;
  if (hh == 0 && q_ok) lse[qrow * 8 + head] = m_run + __builtin_amdgcn_logf(l_run);   // v_log_f32 = log2
  __builtin_amdgcn_wave_barrier();
  const int orow = qtile * 32 + (lane >> 1), oc = (lane & 1) * 16;
  if (orow < nq) {
    float* dst = o + ((size_t)pair * nq + orow) * ldo + head * ATT_HD + oc;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<f32x4*>(dst + i * 4) = *reinterpret_cast<const f32x4*>(&out_s[wave][lane >> 1][oc + i * 4]);
  }
}

template <bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_dq2_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                           const float* __restrict__ v, int ldv, const float* __restrict__ d_o, int ldo,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           float* __restrict__ dq, int lddq, int nq, int qgroups, float qscale,
                                                           uint32_t thresh, float inv_keep, uint32_t seed,
                                                           const uint32_t* __restrict__ salt) {
  seed = train_salted(seed, salt);
  constexpr int TS = KvStage::TS;
  __shared__ __attribute__((aligned(16))) float k_s[2][32][TS];
  __shared__ __attribute__((aligned(16))) float v_s[2][32][TS];
  __shared__ __attribute__((aligned(16))) float out_s[4][32][TS];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int head = blockIdx.x / qgroups, qtile = (blockIdx.x % qgroups) * 4 + wave;
  const int pair = blockIdx.z;
  const bool active = qtile * 32 < nq;
  const int qi = qtile * 32 + l31;
  const bool q_ok = qi < nq;
  const size_t qrow = (size_t)pair * nq + (q_ok ? qi : 0);
  f32x4 qf[4], dof[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    qf[j] = *reinterpret_cast<const f32x4*>(q + qrow * ldq + head * ATT_HD + j * 8 + hh * 4);
    qf[j] *= qscale * LOG2E;
    dof[j] = *reinterpret_cast<const f32x4*>(d_o + qrow * ldo + head * ATT_HD + j * 8 + hh * 4);
  }
  const float lse_q = lse[qrow * 8 + head], delta_q = delta[qrow * 8 + head];
  const float qmask = q_ok ? 1.f : 0.f;
  const MaskRow mrow = mask_row(seed, pair, head, nq, q_ok ? qi : 0);
  f32x16 dqacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) dqacc[r] = 0.f;
  const size_t key_row0 = (size_t)pair * ATT_KEYS;
  KvStage st;
  st.fetch(k, ldk, v, ldv, key_row0, head, t);
  st.stash(k_s[0], v_s[0], t);
  constexpr int NKB = ATT_KEYS / 32;
  for (int kb = 0; kb < NKB; ++kb) {
    const int buf = kb & 1;
    __syncthreads();
    if (kb + 1 < NKB) st.fetch(k, ldk, v, ldv, key_row0 + (kb + 1) * 32, head, t);
    if (active) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(&k_s[buf][l31][j * 8 + hh * 4]);      // A operand of S^T: lane = key
        const f32x4 vf = *reinterpret_cast<const f32x4*>(&v_s[buf][l31][j * 8 + hh * 4]);      // A operand of dP^T = V dO^T
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[j][e], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[e], dof[j][e], dp, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[r] - lse_q);
        float g = dp[r];
        if (DROP) {
          const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          g = mask_keep(mrow, key, thresh) ? g * inv_keep : 0.f;
        }
        s[r] = (p * qmask) * (g - delta_q);             // dS[q][key] (0 for the rows past nq)
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float ktf = k_s[buf][(r & 3) + 8 * (r >> 2) + 4 * hh][l31];     // A operand of dQ^T = K^T dS^T: lane = head dim
        dqacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ktf, s[r], dqacc, 0, 0, 0);
      }
    }
    if (kb + 1 < NKB) st.stash(k_s[buf ^ 1], v_s[buf ^ 1], t);
  }
  if (!active) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) out_s[wave][l31][(r & 3) + 8 * (r >> 2) + 4 * hh] = dqacc[r] * qscale;   // q_eff = q * qscale
  __builtin_amdgcn_wave_barrier();
  const int orow = qtile * 32 + (lane >> 1), oc = (lane & 1) * 16;
  if (orow < nq) {
    float* dst = dq + ((size_t)pair * nq + orow) * lddq + head * ATT_HD + oc;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<f32x4*>(dst + i * 4) = *reinterpret_cast<const f32x4*>(&out_s[wave][lane >> 1][oc + i * 4]);
  }
}


// -------------------------------------------------------------------------------------------------------------------
// Backward in ONE pass (third form): workgroup = head x pair; K and V of the head (512 x 32 each) stay in LDS for the whole kernel
// (128 KB, XOR-swizzled instead of padded: the CU has 160 KB); a wavefront owns 128 keys (4 key tiles: their dK / dV accumulators live
// in registers for the whole kernel) and walks the query tiles, which the four wavefronts share (Q, dO, lse, delta of 32 queries
// through LDS, fetched one tile ahead into registers).  Per (query tile, key tile): S and dP~ (32 MFMAs), P~ and dS in registers,
// dV += dO^T P~ and dK += Q^T dS (32 MFMAs, the D registers of the first two are the B operands), and - what the two-kernel forms
// pay a second S / dP~ for - dQ^T += K^T dS^T (16 MFMAs) with dS TRANSPOSED through a 4 KB LDS tile of the wavefront (D layout:
// lane = key; B operand of this product: lane = query).  After its 4 key tiles a wavefront holds dQ^T of the tile over ITS 128 keys;
// the four partial tiles are summed through LDS and written out: 5 products instead of 7, every exp2 and every dropout-mask hash
// once instead of twice, no partial results in HBM.
// -------------------------------------------------------------------------------------------------------------------
#ifndef FB_PIPE
#define FB_PIPE 1   // S / dP~ of key tile kt+1 issued before the softmax arithmetic of tile kt: 258 vs 268 us (encoder shape)
#endif
constexpr int FB_TS = 36;                         // padded row of the Q / dO tiles
constexpr int FB_DS = 33;                         // padded row of a wavefront's dS / output tile
constexpr size_t fb_smem_bytes(int kt) { return (size_t)(2 * kt * 128 * ATT_HD + 2 * 32 * FB_TS + 2 * 32 + 4 * 32 * FB_DS) * sizeof(float); }
__device__ __forceinline__ int fb_swz(int row, int col) { return row * ATT_HD + ((((col >> 2) ^ (row & 7)) << 2) | (col & 3)); }

template <bool DROP, int KT>
__global__ __launch_bounds__(256) void attn_bwd_fused_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                             const float* __restrict__ v, int ldv, const float* __restrict__ d_o, int ldo,
                                                             const float* __restrict__ lse, const float* __restrict__ delta,
                                                             float* __restrict__ dq, int lddq, size_t dq_split_stride,
                                                             float* __restrict__ dk, int lddk,
                                                             float* __restrict__ dv, int lddv, int nq, float qscale, uint32_t thresh,
                                                             float inv_keep, uint32_t seed, const uint32_t* __restrict__ salt) {
  seed = train_salted(seed, salt);
  extern __shared__ __attribute__((aligned(16))) float fb_smem[];
  float* k_s = fb_smem;                           // [512][32] swizzled
  constexpr int WG_KEYS = KT * 128;               // keys of this workgroup: wavefront w owns KT tiles of 32 from w * KT * 32
  float* v_s = k_s + WG_KEYS * ATT_HD;
  float* q_s = v_s + WG_KEYS * ATT_HD;                       // [32][FB_TS] raw q rows of the tile
  float* do_s = q_s + 32 * FB_TS;
  float* ld_s = do_s + 32 * FB_TS;                // lse[32] | delta[32]
  float* w_s = ld_s + 64;                         // [4 wavefronts][32][FB_DS]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  constexpr int SPLITS = 4 / KT;                  // workgroups per (pair, head): each leaves a dQ partial (summed by the caller) when > 1
  const int head = blockIdx.x / SPLITS, ksplit = blockIdx.x % SPLITS, pair = blockIdx.z;
  const int kbase = ksplit * WG_KEYS;
  dq += (size_t)ksplit * dq_split_stride;
  float* my_s = w_s + wave * 32 * FB_DS;
  // K, V of this head -> LDS: thread -> (row t >> 3 (+32 per step), float4 t & 7)
  {
    const size_t krow0 = (size_t)pair * ATT_KEYS + kbase;
    const int c4 = t & 7;
#pragma unroll 4
    for (int r0 = 0; r0 < WG_KEYS; r0 += 32) {
      const int row = r0 + (t >> 3);
      const f32x4 kk = *reinterpret_cast<const f32x4*>(k + (krow0 + row) * ldk + head * ATT_HD + c4 * 4);
      const f32x4 vv = *reinterpret_cast<const f32x4*>(v + (krow0 + row) * ldv + head * ATT_HD + c4 * 4);
      *reinterpret_cast<f32x4*>(k_s + row * ATT_HD + ((c4 ^ (row & 7)) << 2)) = kk;
      *reinterpret_cast<f32x4*>(v_s + row * ATT_HD + ((c4 ^ (row & 7)) << 2)) = vv;
    }
  }
  f32x16 dkacc[KT], dvacc[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dkacc[kt][r] = dvacc[kt][r] = 0.f;
  const int nqb = (nq + 31) / 32;
  const int srow = t >> 3, sc4 = (t & 7) * 4;
  f32x4 pq, pdo;
  float pld = 0.f;
  auto fetch = [&](int qb) {
    const int qa = qb * 32 + srow;
    const size_t row = (size_t)pair * nq + (qa < nq ? qa : 0);
    pq = *reinterpret_cast<const f32x4*>(q + row * ldq + head * ATT_HD + sc4);
    pdo = *reinterpret_cast<const f32x4*>(d_o + row * ldo + head * ATT_HD + sc4);
    if (t < 64) {
      const int ql = qb * 32 + l31;
      const size_t r2 = (size_t)pair * nq + (ql < nq ? ql : 0);
      pld = (hh == 0 ? lse : delta)[r2 * 8 + head];
    }
  };
  fetch(0);
  for (int qb = 0; qb < nqb; ++qb) {
    __syncthreads();                                             // the previous tile's Q / dO and dQ partials are no longer read
    *reinterpret_cast<f32x4*>(q_s + srow * FB_TS + sc4) = pq;
    *reinterpret_cast<f32x4*>(do_s + srow * FB_TS + sc4) = pdo;
    if (t < 64) ld_s[hh * 32 + l31] = pld;
    __syncthreads();                                             // tile qb (and, the first time, K / V) is in LDS
    if (qb + 1 < nqb) fetch(qb + 1);
    f32x4 qaf[4], doaf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      qaf[j] = *reinterpret_cast<const f32x4*>(q_s + l31 * FB_TS + j * 8 + hh * 4);
      qaf[j] *= qscale * LOG2E;
      doaf[j] = *reinterpret_cast<const f32x4*>(do_s + l31 * FB_TS + j * 8 + hh * 4);
    }
    f32x16 dqacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) dqacc[r] = 0.f;
    // S and dP~ of a key tile (32 MFMAs); issued one tile AHEAD of their use, so that the softmax arithmetic of tile kt (VALU) has
    // the matrix pipe busy with tile kt+1 underneath it (one wavefront per SIMD here: nobody else would fill it)
    auto scores = [&](int kt, f32x16& s, f32x16& dp) {
      const int krow = (wave * KT + kt) * 32 + l31;              // row in this workgroup's K / V
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int off = krow * ATT_HD + (((j * 2 + hh) ^ (krow & 7)) << 2);
        const f32x4 kfb = *reinterpret_cast<const f32x4*>(k_s + off);      // B operands: lane = key
        const f32x4 vfb = *reinterpret_cast<const f32x4*>(v_s + off);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(qaf[j][e], kfb[e], s, 0, 0, 0);       // S[q][key], lane = key
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(doaf[j][e], vfb[e], dp, 0, 0, 0);    // dP~[q][key]
        }
      }
    };
    f32x16 s_cur, dp_cur, s_nxt, dp_nxt;
#if FB_PIPE
    scores(0, s_cur, dp_cur);
#endif
    const MaskTile mt = mask_tile(seed, pair, head, nq, qb * 32);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const int key0 = (wave * KT + kt) * 32;                    // (local; + kbase = the key's index in the pair)
      const int kj = kbase + key0 + l31;
#if FB_PIPE
      if (kt + 1 < KT) scores(kt + 1, s_nxt, dp_nxt);
#else
      scores(kt, s_cur, dp_cur);
#endif
      f32x16 pt, ds;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = (r & 3) + 8 * (r >> 2) + 4 * hh;
        const float e = __builtin_amdgcn_exp2f(s_cur[r] - ld_s[ql]);
        const float p = qb * 32 + ql < nq ? e : 0.f;
        float keep = 1.f;
        if (DROP) keep = mask_keep(mt.row(ql), kj, thresh) ? inv_keep : 0.f;
        pt[r] = p * keep;                                        // P~[q][key]
        ds[r] = p * (dp_cur[r] * keep - ld_s[32 + ql]);          // dS[q][key]
      }
      // dS^T through this wavefront's LDS tile: [query][key]
#pragma unroll
      for (int r = 0; r < 16; ++r) my_s[((r & 3) + 8 * (r >> 2) + 4 * hh) * FB_DS + l31] = ds[r];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = (r & 3) + 8 * (r >> 2) + 4 * hh;
        const float dot = do_s[ql * FB_TS + l31];                // A operand of dV^T = dO^T P~: lane = head dim
        const float qt = q_s[ql * FB_TS + l31] * qscale;         // A operand of dK^T = Q^T dS
        dvacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(dot, pt[r], dvacc[kt], 0, 0, 0);
        dkacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qt, ds[r], dkacc[kt], 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) {
        const int kr = key0 + 2 * s2 + hh;
        const float ka = k_s[fb_swz(kr, l31)];                   // A operand of dQ^T = K^T dS^T: lane = head dim, k = key
        const float db = my_s[l31 * FB_DS + 2 * s2 + hh];        // B operand: lane = query
        dqacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka, db, dqacc, 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");                             // (keep later tiles' LDS reads from being hoisted up here)
#if FB_PIPE
      if (kt + 1 < KT) {
        s_cur = s_nxt;
        dp_cur = dp_nxt;
      }
#endif
    }
    // the four wavefronts' dQ^T tiles (rows = head dim, column = query) summed through LDS, 4 head dims x 1 query per thread
#pragma unroll
    for (int r = 0; r < 16; ++r) my_s[r * 64 + lane] = dqacc[r];
    __syncthreads();
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += w_s[w * 32 * FB_DS + (wave * 4 + i) * 64 + lane];
      const int qo = qb * 32 + l31;
      if (qo < nq)
        *reinterpret_cast<f32x4*>(dq + ((size_t)pair * nq + qo) * lddq + head * ATT_HD + 8 * wave + 4 * hh) = acc * qscale;
    }
  }
  __syncthreads();
  // dK / dV tiles: D rows = head dim, column (lane & 31) = key: through the wavefront's LDS tile to rows
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const size_t orow = (size_t)pair * ATT_KEYS + kbase + (wave * KT + kt) * 32 + (lane >> 1);
    const int oc = (lane & 1) * 16;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
#pragma unroll
      for (int r = 0; r < 16; ++r) my_s[l31 * FB_DS + (r & 3) + 8 * (r >> 2) + 4 * hh] = which == 0 ? dkacc[kt][r] : dvacc[kt][r];
      __builtin_amdgcn_wave_barrier();
      float* dst = (which == 0 ? dk + orow * lddk : dv + orow * lddv) + head * ATT_HD + oc;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = my_s[(lane >> 1) * FB_DS + oc + i * 4 + e];
        *reinterpret_cast<f32x4*>(dst + i * 4) = o4;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

}  // namespace

// 1: first form of the three kernels (kept for A/B and as a cross-check); 2: a wavefront owns its keys / queries, dQ and dK/dV in two
// kernels; 3: the backward in one pass (attn_bwd_fused_kernel); 0 (default): 3 where it is the faster one - enough (pair, head)
// workgroups to fill the chip and enough query tiles to amortise parking K / V in LDS (encoder self-attention of a training batch:
// 258 vs 347 us at 32 pairs x 512); with fewer pairs the keys of a head are split over 2 or 4 workgroups (attn_fused_kt) - else 2
#define g_attn_bwd_form knob(KN_TRAIN_ATTENTION_FORM)
// key tiles per wavefront of the one-pass backward (4: one workgroup per (pair, head); 2 / 1: two / four workgroups, each leaving a dQ
// partial in `scratch` that train_sum_parts adds - the decoder's 16 pairs x 200 queries: 128 workgroups would leave half the chip idle),
// 0: use the two-kernel second form.  The split forms need the scratch buffer and a contiguous dq.
static int attn_fused_kt(int nb, int nq, int lddq, const float* scratch) {
  if (g_attn_bwd_form == 1 || g_attn_bwd_form == 2) return 0;
  const bool can_split = scratch != nullptr && lddq == 256;
  if (nb * 8 >= 192 || !can_split) return (g_attn_bwd_form == 3 || (nb * 8 >= 192 && nq >= 256)) ? 4 : 0;
  if (nb * 16 >= 192) return 2;
  if (nb * 32 >= 192 || g_attn_bwd_form == 3) return 1;
  return 0;
}

int train_attention_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo, float* lse,
                        int nb, int nq, float qscale, float p, uint32_t seed, hipStream_t s) {
  if (nb <= 0 || nq <= 0) return 0;
  if (ldq % 4 || ldk % 4 || ldv % 4 || ldo % 4) return -1;
  if (g_attn_bwd_form == 1) {
    dim3 grid(((nq + 31) / 32) * 8, 1, nb);
    hipLaunchKernelGGL(attn_train_fwd_kernel, grid, dim3(256), 0, s, q, ldq, k, ldk, v, ldv, o, ldo, lse, nq, qscale, train_thresh(p),
                       p > 0.f ? 1.f / (1.f - p) : 1.f, seed, train_salt_ptr());
  } else {
    const int qgroups = ((nq + 31) / 32 + 3) / 4;
    const uint32_t thresh = train_thresh(p);
    hipLaunchKernelGGL(thresh ? attn_train_fwd2_kernel<true> : attn_train_fwd2_kernel<false>, dim3(qgroups * 8, 1, nb), dim3(256), 0, s, q,
                       ldq, k, ldk, v, ldv, o, ldo, lse, nq, qgroups, qscale, thresh, p > 0.f ? 1.f / (1.f - p) : 1.f, seed,
                       train_salt_ptr());
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int train_attention_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o, const float* d_o,
                        int ldo, const float* lse, float* delta, float* dq, int lddq, float* dk, int lddk, float* dv, int lddv,
                        int nb, int nq, float qscale, float p, uint32_t seed, float* scratch, hipStream_t s) {
  if (nb <= 0 || nq <= 0) return 0;
  if (ldq % 4 || ldk % 4 || ldv % 4 || ldo % 4 || lddq % 4 || lddk % 4 || lddv % 4) return -1;
  const int rows = nb * nq;
  const uint32_t thresh = train_thresh(p);
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, o, d_o, ldo, delta, rows);
  if (hipGetLastError() != hipSuccess) return -2;
  const int kt = attn_fused_kt(nb, nq, lddq, scratch);
  if (kt != 0) {
    static PerDeviceFlag attr_set;
    if (!attr_set.get()) {
      const void* fns[6] = {reinterpret_cast<const void*>(attn_bwd_fused_kernel<true, 4>), reinterpret_cast<const void*>(attn_bwd_fused_kernel<false, 4>),
                            reinterpret_cast<const void*>(attn_bwd_fused_kernel<true, 2>), reinterpret_cast<const void*>(attn_bwd_fused_kernel<false, 2>),
                            reinterpret_cast<const void*>(attn_bwd_fused_kernel<true, 1>), reinterpret_cast<const void*>(attn_bwd_fused_kernel<false, 1>)};
      const int kts[6] = {4, 4, 2, 2, 1, 1};
      for (int i = 0; i < 6; ++i)
        if (hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)fb_smem_bytes(kts[i])) != hipSuccess) return -2;
      attr_set.set();
    }
    const int splits = 4 / kt;
    float* dq_dst = splits > 1 ? scratch : dq;
    const int ld_dst = splits > 1 ? 256 : lddq;
    const size_t split_stride = splits > 1 ? (size_t)rows * 256 : 0;
    auto fn = kt == 4 ? (thresh ? attn_bwd_fused_kernel<true, 4> : attn_bwd_fused_kernel<false, 4>)
            : kt == 2 ? (thresh ? attn_bwd_fused_kernel<true, 2> : attn_bwd_fused_kernel<false, 2>)
                      : (thresh ? attn_bwd_fused_kernel<true, 1> : attn_bwd_fused_kernel<false, 1>);
    hipLaunchKernelGGL(fn, dim3(8 * splits, 1, nb), dim3(256), fb_smem_bytes(kt), s, q, ldq, k, ldk, v, ldv, d_o, ldo, lse, delta, dq_dst, ld_dst,
                       split_stride, dk, lddk, dv, lddv, nq, qscale, thresh, inv_keep, seed, train_salt_ptr());
    if (hipGetLastError() != hipSuccess) return -2;
    if (splits > 1) return train_sum_parts(scratch, splits, (size_t)rows * 256, dq, s);   // fixed order: deterministic
    return 0;
  }
  if (g_attn_bwd_form == 1)
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(((nq + 31) / 32) * 8, 1, nb), dim3(256), 0, s, q, ldq, k, ldk, v, ldv, d_o, ldo, lse,
                       delta, dq, lddq, nq, qscale, thresh, inv_keep, seed, train_salt_ptr());
  else {
    const int qgroups = ((nq + 31) / 32 + 3) / 4;
    hipLaunchKernelGGL(thresh ? attn_bwd_dq2_kernel<true> : attn_bwd_dq2_kernel<false>, dim3(qgroups * 8, 1, nb), dim3(256), 0, s, q, ldq, k,
                       ldk, v, ldv, d_o, ldo, lse, delta, dq, lddq, nq, qgroups, qscale, thresh, inv_keep, seed, train_salt_ptr());
  }
  if (hipGetLastError() != hipSuccess) return -2;
  if (g_attn_bwd_form == 1)
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(16 * 8, 1, nb), dim3(256), 0, s, q, ldq, k, ldk, v, ldv, d_o, ldo, lse, delta, dk,
                       lddk, dv, lddv, nq, qscale, thresh, inv_keep, seed, train_salt_ptr());
  else
    hipLaunchKernelGGL(thresh ? attn_bwd_dkv2_kernel<true> : attn_bwd_dkv2_kernel<false>, dim3(4 * 8, 1, nb), dim3(256), 0, s, q, ldq, k, ldk,
                       v, ldv, d_o, ldo, lse, delta, dk, lddk, dv, lddv, nq, qscale, thresh, inv_keep, seed, train_salt_ptr());
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
