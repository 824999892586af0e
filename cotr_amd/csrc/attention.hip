// Multi-head softmax attention over the fixed 512-token (16x32) COTR memory, fp32 MFMA, gfx950.
//
// Serves both call sites of nn.MultiheadAttention in the reference:
//   encoder self-attention   (COTR/models/transformer.py:149-153)  nq = 512 queries per pair
//   decoder cross-attention  (COTR/models/transformer.py:192-195)  nq = Q   queries per pair
// 8 heads of 32.  q is expected pre-scaled by head_dim^-0.5 (fused into the q-projection epilogue,
// as torch scales q before q.k^T).  No masks: the key-padding mask is all-False for every caller
// of the reference (input is always exactly 256x512, COTR/models/backbone.py:80).
//
// One workgroup = 128 queries x 1 head x 1 pair; K_h and V_h of that pair/head (512x32 fp32 each,
// 64 KB) are staged ONCE in LDS (139 KB of the CU's 160 KB) and shared by the 4 wavefronts, each
// owning 32 queries.  Per 32-key block, per wavefront:
//   S^T = K_blk . Q^T   16x v_mfma_f32_32x32x2_f32   (lane = one query, 16 keys in registers:
//                        the softmax reduction is in-lane plus one cross-half shuffle)
//   online softmax       running max / sum, exp in fp32
//   O^T += V_blk^T . P^T 16x v_mfma_f32_32x32x2_f32   (P registers feed the B operand directly,
//                        no transpose through LDS)
#include "common.h"

#define ATT_KEYS 512
#define ATT_HD 32
#define ATT_KLD 36  // padded K row (floats): conflict-free ds_read_b128 of 16 different rows
#define ATT_BQ 128

__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ q, int ldq,
                                                        const float* __restrict__ k,
                                                        const float* __restrict__ v, int ldkv,
                                                        float* __restrict__ o, int ldo, int nq) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;                        // [512][36]
  float* Vs = smem + ATT_KEYS * ATT_KLD;   // [512][32]

  const int t = threadIdx.x;
  const int head = blockIdx.y, pair = blockIdx.z;
  const int lr = t >> 3, lc = (t & 7) * 4;
  const float* kg = k + (size_t)pair * ATT_KEYS * ldkv + head * ATT_HD + lc;
  const float* vg = v + (size_t)pair * ATT_KEYS * ldkv + head * ATT_HD + lc;
#pragma unroll 4
  for (int i = 0; i < ATT_KEYS / 32; ++i) {
    const int row = lr + 32 * i;
    *reinterpret_cast<f32x4*>(&Ks[row * ATT_KLD + lc]) = *reinterpret_cast<const f32x4*>(kg + (size_t)row * ldkv);
    *reinterpret_cast<f32x4*>(&Vs[row * ATT_HD + lc]) = *reinterpret_cast<const f32x4*>(vg + (size_t)row * ldkv);
  }

  const int lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int qi = blockIdx.x * ATT_BQ + wave * 32 + l31;
  const bool q_ok = qi < nq;
  const size_t qrow = (size_t)pair * nq + (q_ok ? qi : 0);

  // Q^T fragment (B operand): lane holds q[qi][j*8 + hh*4 + e]
  f32x4 qf[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    qf[j] = q_ok ? *reinterpret_cast<const f32x4*>(q + qrow * ldq + head * ATT_HD + j * 8 + hh * 4) : z;
  }
  __syncthreads();

  f32x16 oacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  for (int kb = 0; kb < ATT_KEYS / 32; ++kb) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 kf = *reinterpret_cast<const f32x4*>(&Ks[(kb * 32 + l31) * ATT_KLD + j * 8 + hh * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[j][e], s, 0, 0, 0);
    }
    // s[r] = score(key = kb*32 + (r&3) + 8*(r>>2) + 4*hh, query = l31)
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = expf(m_run - m_new);  // first block: exp(-inf) = 0
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = expf(s[r] - m_new);
      psum += s[r];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
    // O^T[d][q] += sum_key V[key][d] * P[q][key]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float vf = Vs[(kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * ATT_HD + l31];
      oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, s[r], oacc, 0, 0, 0);
    }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.f / l_tot;
  if (q_ok) {
    float* og = o + qrow * ldo + head * ATT_HD + 4 * hh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 w = {oacc[4 * g] * inv, oacc[4 * g + 1] * inv, oacc[4 * g + 2] * inv, oacc[4 * g + 3] * inv};
      *reinterpret_cast<f32x4*>(og + 8 * g) = w;  // d = 8g + 4hh + (0..3)
    }
  }
}

static const size_t kAttSmem = (size_t)(ATT_KEYS * ATT_KLD + ATT_KEYS * ATT_HD) * sizeof(float);

int init_attention_attributes() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAttSmem) == hipSuccess
             ? 0
             : -2;
}

int launch_attention(const float* q, int ldq, const float* k, const float* v, int ldkv, float* o, int ldo,
                     int nb, int nq, hipStream_t s) {
  if (nb <= 0 || nq <= 0) return 0;
  if (ldq % 4 || ldkv % 4 || ldo % 4) return -1;
  dim3 grid((nq + ATT_BQ - 1) / ATT_BQ, 8, nb);
  hipLaunchKernelGGL(attention_kernel, grid, dim3(256), kAttSmem, s, q, ldq, k, v, ldkv, o, ldo, nq);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
