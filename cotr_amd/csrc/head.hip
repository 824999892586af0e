// Decoder tail in ONE launch for the small-row regime (one pair / ~1000 query rows), fp32 MFMA, gfx950:
//     hs   = LayerNorm(tgt)                       transformer.decoder.norm   (COTR/models/transformer.py:110-111)
//     h1   = relu(hs . W0^T + b0)                 corr_embed.layers.0        (COTR/models/position_encoding.py:23-26)
//     h2   = relu(h1 . W1^T + b1)                 corr_embed.layers.1
//     pred = h2 . W2^T + b2                       corr_embed.layers.2  (256 -> 2), scattered to out[b][q][0..1]
// on the LAST decoder layer's output only (cotr_model.py:39 keeps [-1]).  Everything is row-local, so a workgroup owns 16
// query rows from the norm to the two output floats: the 16 x 256 activations live in LDS between the stages and nothing but
// the prediction is written - four launches (layernorm, two 256 x 256 linears, head2) and three round trips less.
//
// Workgroup = 8 wavefronts; a wavefront owns 32 output columns (two 16-column blocks) of a 256 x 256 layer and the whole
// K = 256 on v_mfma_f32_16x16x4_f32 (16-row tiles: a 32-row tile would halve the workgroups, 1000 rows are only 63 tiles).
// A fragments (activations) come from LDS, B fragments (weights; each element is used once per workgroup) straight from
// global / L2 in MFMA layout.  Fragment trick as in gemm.hip: a lane reads ONE float4 = 4 consecutive k and feeds element e
// to the e-th of 4 MFMAs; both operands use the same k permutation.
#include "common.h"

#define HD_D 256
#define HD_LD 260   // padded LDS row

struct HeadParams {
  const float* X;      // [rows][256] output of the last decoder layer (after its norm3)
  const float *nw, *nb;            // decoder.norm
  const float *w0, *b0, *w1, *b1;  // corr_embed.layers.0 / .1  [256][256], [256]
  const float *w2, *b2;            // corr_embed.layers.2      [2][256], [2]
  float* hs;           // optional [rows][256]: the normalised rows ('hs' debug tap), or nullptr
  float* out;          // [nb][q_total][2]
  int rows, nq, q_total;
};

__device__ __forceinline__ float head_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// dst[16][256] = relu(src[16][256] . W^T + b) for this wavefront's 32 columns
__device__ __forceinline__ void head_linear(const float* src, float* dst, const float* __restrict__ W, const float* __restrict__ bias,
                                            int wave, int lane) {
  const int l15 = lane & 15, kq = lane >> 4;
  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const float* wrow0 = W + (size_t)(32 * wave + l15) * HD_D + kq * 4;
  const float* wrow1 = wrow0 + 16 * HD_D;
  const float* arow = src + l15 * HD_LD + kq * 4;
#pragma unroll
  for (int jg = 0; jg < 4; ++jg) {   // 4 groups of 4 k-steps (16 k each): 8 weight float4 in flight per group
    f32x4 b0[4], b1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      b0[j] = *reinterpret_cast<const f32x4*>(wrow0 + (jg * 4 + j) * 16);
      b1[j] = *reinterpret_cast<const f32x4*>(wrow1 + (jg * 4 + j) * 16);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(arow + (jg * 4 + j) * 16);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b0[j][e], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b1[j][e], acc[1], 0, 0, 0);
      }
    }
  }
  // D: column = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int n = 32 * wave + cb * 16 + l15;
    const float bv = bias[n];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = acc[cb][r] + bv;
      v = (v < 0.f) ? 0.f : v;   // NaN passes through like torch.relu
      dst[(kq * 4 + r) * HD_LD + n] = v;
    }
  }
}

__global__ __launch_bounds__(512) void dec_head_kernel(const HeadParams p) {
  __shared__ __attribute__((aligned(16))) float xs[16 * HD_LD];
  __shared__ __attribute__((aligned(16))) float ys[16 * HD_LD];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int m0 = blockIdx.x * 16;

  // ---- decoder.norm: wave w -> rows 2w, 2w+1 (same arithmetic as layernorm_kernel, pointwise.hip) ----
  {
    const f32x4 ww = *reinterpret_cast<const f32x4*>(p.nw + lane * 4);
    const f32x4 bb = *reinterpret_cast<const f32x4*>(p.nb + lane * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int lr = wave * 2 + i, row = m0 + lr;
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      if (row < p.rows) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p.X + (size_t)row * HD_D + lane * 4);
        const float mean = head_wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256.f);
        const f32x4 d = {v[0] - mean, v[1] - mean, v[2] - mean, v[3] - mean};
        const float var = head_wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / 256.f);
        const float rstd = 1.f / sqrtf(var + 1e-5f);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = d[e] * rstd * ww[e] + bb[e];
        if (p.hs != nullptr) *reinterpret_cast<f32x4*>(p.hs + (size_t)row * HD_D + lane * 4) = o;
      }
      *reinterpret_cast<f32x4*>(&xs[lr * HD_LD + lane * 4]) = o;
    }
  }
  __syncthreads();
  head_linear(xs, ys, p.w0, p.b0, wave, lane);
  __syncthreads();
  head_linear(ys, xs, p.w1, p.b1, wave, lane);
  __syncthreads();
  // ---- last layer 256 -> 2: wave w -> rows 2w, 2w+1 (same arithmetic as head2_kernel) ----
  {
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(p.w2 + lane * 4);
    const f32x4 w1 = *reinterpret_cast<const f32x4*>(p.w2 + 256 + lane * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int lr = wave * 2 + i, row = m0 + lr;
      const f32x4 v = *reinterpret_cast<const f32x4*>(&xs[lr * HD_LD + lane * 4]);
      float s0 = v[0] * w0[0] + v[1] * w0[1] + v[2] * w0[2] + v[3] * w0[3];
      float s1 = v[0] * w1[0] + v[1] * w1[1] + v[2] * w1[2] + v[3] * w1[3];
      s0 = head_wave_sum(s0);
      s1 = head_wave_sum(s1);
      if (lane == 0 && row < p.rows) {
        const int bi = row / p.nq, qi = row - bi * p.nq;
        float* dst = p.out + ((size_t)bi * p.q_total + qi) * 2;
        dst[0] = s0 + p.b2[0];
        dst[1] = s1 + p.b2[1];
      }
    }
  }
}

int launch_dec_head(const float* x, const float* nw, const float* nb, const float* w0, const float* b0, const float* w1,
                    const float* b1, const float* w2, const float* b2, float* hs, float* out, int nb_pairs, int nq, int q_total,
                    hipStream_t s) {
  const int rows = nb_pairs * nq;
  if (rows <= 0) return 0;
  HeadParams p;
  p.X = x; p.nw = nw; p.nb = nb; p.w0 = w0; p.b0 = b0; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2;
  p.hs = hs; p.out = out; p.rows = rows; p.nq = nq; p.q_total = q_total;
  hipLaunchKernelGGL(dec_head_kernel, dim3((rows + 15) / 16), dim3(512), 0, s, p);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
