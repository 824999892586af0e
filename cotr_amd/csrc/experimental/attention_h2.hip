// RESEARCH (knob split_f16 = 3, libcotr_hip_exp.so only): the resident-K/V attention kernel of attention.hip (attention_res_kernel) on
// packed split-f16 tensors - see gemm_h2.h for the format and the error model.
//
// Same decomposition: workgroup = (pair, head, chunk of query tiles), 8 wavefronts, K_h and V_h (512 x 32 PACKED dwords each, 128 KB,
// XOR-swizzled) parked in LDS once; a wavefront owns a 32-query tile at a time and runs the four key quarters as four independent
// online-softmax chains merged in registers.  What changes is the arithmetic of the two products:
//   S^T = K Q^T   (32 keys x 32 queries, 32 channels deep): 2 slices x 3 v_mfma_f32_32x32x16_f16 instead of 16 v_mfma_f32_32x32x2_f32
//   O^T += V^T P^T (32 dims x 32 queries, 32 keys deep):    the same; P (fp32, in the accumulator layout) is split in registers:
//                  hi = f16(p) truncated (v_cvt_pkrtz), lo = f16((p - hi) * 2^11) rounded - the remainder is exact, so 22 bits as everywhere
// hi x lo + lo x hi go to a zero-initialised temporary that joins the accumulator once per key block (scaled by 2^-11): no second set of
// running accumulators to rescale.  k-slot <-> key / channel assignments only have to agree between the A and the B operand of a product.
// q: fp32 or packed (q_packed); o: fp32 or packed (out_packed: its only consumer is the out-projection GEMM).
#include "../common.h"
#include "gemm_h2.h"

#define ATT_KEYS 512
#define ATT_HD 32

namespace {
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr size_t ATT_H2_SMEM = (size_t)2 * ATT_KEYS * ATT_HD * sizeof(float);

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

// 8 fp32 values -> hi / lo MFMA operands (hi truncated, lo = the exact remainder * 2^11 rounded to nearest)
__device__ __forceinline__ void h2_split8(const float (&p)[8], f16x8& hi, f16x8& lo) {
  u32x4 h, l;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f16x2 hp = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(p[2 * i], p[2 * i + 1]));
    const float r0 = (p[2 * i] - (float)hp[0]) * 2048.f, r1 = (p[2 * i + 1] - (float)hp[1]) * 2048.f;
    f16x2 lp;
    lp[0] = (_Float16)r0;
    lp[1] = (_Float16)r1;
    h[i] = __builtin_bit_cast(unsigned int, hp);
    l[i] = __builtin_bit_cast(unsigned int, lp);
  }
  hi = __builtin_bit_cast(f16x8, h);
  lo = __builtin_bit_cast(f16x8, l);
}

__global__ __launch_bounds__(512) void attention_res_h2_kernel(const float* __restrict__ q, int ldq, int q_packed, const float* __restrict__ k,
                                                               const float* __restrict__ v, int ldkv, float* __restrict__ o, int ldo,
                                                               int out_packed, int nq, int tiles_per_chunk, int* ovf) {
  extern __shared__ __attribute__((aligned(16))) float res_smem[];
  float* k_s = res_smem;
  float* v_s = res_smem + ATT_KEYS * ATT_HD;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int head = blockIdx.x & 7, chunk = blockIdx.x >> 3;
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
    // B operand of S^T = K Q^T: this lane's query row, channels 16 s + 8 hh + 0..7 of slice s, scaled into the log2 domain
    f16x8 qh[2], ql[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float qv[8];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const f32x4 raw = *reinterpret_cast<const f32x4*>(q + qrow * ldq + head * ATT_HD + 16 * s2 + 8 * hh + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = raw[e];
          const float f = q_packed ? h2_unpack(__float_as_uint(x)) : x;
          qv[4 * g + e] = q_ok ? f * 1.44269504088896340736f : 0.f;
        }
      }
      u32x4 d0, d1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        d0[e] = h2_pack_chk(qv[e], ovf);
        d1[e] = h2_pack_chk(qv[4 + e], ovf);
      }
      h2_unzip(d0, d1, qh[s2], ql[s2]);
    }
    f32x16 oacc[4];
    float m_run[4], l_run[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      oacc[c] = zero16();
      m_run[c] = -INFINITY;
      l_run[c] = 0.f;
    }
#pragma unroll 1
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {                    // the four key quarters: independent chains
        const int key0 = c * 128 + kb * 32;
        const int krow = key0 + l31;
        f32x16 s = zero16(), sx = zero16();
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {               // A operand: this lane's key row, channels 16 s2 + 8 hh + 0..7 (chunks 4 s2 + 2 hh, + 1)
          const u32x4 k0 = *reinterpret_cast<const u32x4*>(k_s + krow * ATT_HD + (((4 * s2 + 2 * hh) ^ (krow & 7)) << 2));
          const u32x4 k1 = *reinterpret_cast<const u32x4*>(k_s + krow * ATT_HD + (((4 * s2 + 2 * hh + 1) ^ (krow & 7)) << 2));
          f16x8 kh, kl;
          h2_unzip(k0, k1, kh, kl);
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s2], s, 0, 0, 0);
          sx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s2], sx, 0, 0, 0);
          sx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s2], sx, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = fmaf(sx[r], 0x1p-11f, s[r]);
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
        // O^T += V^T P^T: slot i of slice s2 <-> register r = 8 s2 + i <-> key key0 + (r & 3) + 8 (r >> 2) + 4 hh, for P (B) and V (A) alike
        f32x16 x = zero16();
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          float pv[8];
          u32x4 v0, v1;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int r = 8 * s2 + i;
            pv[i] = s[r];
            const int vr = key0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            const unsigned int vd = __float_as_uint(v_s[vr * ATT_HD + ((((l31 >> 2) ^ (vr & 7)) << 2) | (l31 & 3))]);
            if (i < 4) v0[i] = vd; else v1[i - 4] = vd;
          }
          f16x8 ph, pl, vh, vl;
          h2_split8(pv, ph, pl);
          h2_unzip(v0, v1, vh, vl);
          oacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, oacc[c], 0, 0, 0);
          x = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, x, 0, 0, 0);
          x = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, x, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[c][r] = fmaf(x[r], 0x1p-11f, oacc[c][r]);
      }
    }
    // merge of the four quarters (attention_res_kernel's)
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
          const float y = acc * inv;
          out[i] = out_packed ? __uint_as_float(h2_pack_chk(y, ovf)) : y;
        }
        *reinterpret_cast<f32x4*>(dst + 8 * g) = out;
      }
    }
  }
}
}  // namespace

// softmax(q k^T) v per head over ATT_KEYS keys, k / v PACKED [nb * 512, ldkv], q fp32 or packed, o fp32 or packed; nq >= 1
int launch_attention_h2(const float* q, int ldq, int q_packed, const float* k, const float* v, int ldkv, float* o, int ldo, int out_packed,
                        int nb, int nq, hipStream_t s) {
  if (nb <= 0 || nq <= 0) return 0;
  if (ldq % 4 || ldkv % 4 || ldo % 4) return -1;
  const int tiles = (nq + 31) / 32;
  int tpc = 32;
  while (tpc > 8 && (long)((tiles + tpc - 1) / tpc) * 8 * nb < 256) tpc /= 2;
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(attention_res_h2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)ATT_H2_SMEM) != hipSuccess)
      return -2;
    attr_set.set();
  }
  const int chunks = (tiles + tpc - 1) / tpc;
  int* ovf = h2_overflow_flag();
  if (ovf == nullptr) return -2;
  hipLaunchKernelGGL(attention_res_h2_kernel, dim3(chunks * 8, 1, nb), dim3(512), ATT_H2_SMEM, s, q, ldq, q_packed, k, v, ldkv, o, ldo,
                     out_packed, nq, tpc, ovf);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
