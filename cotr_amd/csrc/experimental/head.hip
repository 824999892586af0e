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
#include "../common.h"

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

// One half (8 k-steps of 16 = 128 of the 256 input channels) of this wavefront's weight fragment: 2 column blocks x 8 float4.
// The loads are issued a stage ahead of their use (the kernel is otherwise bound by the L2 round trip of each group).
struct HeadW {
  f32x4 b[2][8];
};
__device__ __forceinline__ void head_load_w(HeadW& w, const float* __restrict__ W, int wave, int lane, int half) {
  const float* wrow = W + (size_t)(32 * wave + (lane & 15)) * HD_D + (lane >> 4) * 4 + half * 128;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int j = 0; j < 8; ++j) w.b[cb][j] = *reinterpret_cast<const f32x4*>(wrow + cb * 16 * HD_D + j * 16);
}
__device__ __forceinline__ void head_mma(f32x4 (&acc)[2], const float* src, const HeadW& w, int lane, int half) {
  const float* arow = src + (lane & 15) * HD_LD + (lane >> 4) * 4 + half * 128;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(arow + j * 16);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], w.b[0][j][e], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], w.b[1][j][e], acc[1], 0, 0, 0);
    }
  }
}
// dst[16][32 columns of this wavefront] = relu(acc + bias); D: column = lane & 15, row = (lane >> 4) * 4 + reg
__device__ __forceinline__ void head_store(const f32x4 (&acc)[2], float* dst, const float* __restrict__ bias, int wave, int lane) {
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int n = 32 * wave + cb * 16 + (lane & 15);
    const float bv = bias[n];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = acc[cb][r] + bv;
      v = (v < 0.f) ? 0.f : v;   // NaN passes through like torch.relu
      dst[((lane >> 4) * 4 + r) * HD_LD + n] = v;
    }
  }
}

__global__ __launch_bounds__(512) void dec_head_kernel(const HeadParams p) {
  __shared__ __attribute__((aligned(16))) float xs[16 * HD_LD];
  __shared__ __attribute__((aligned(16))) float ys[16 * HD_LD];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int m0 = blockIdx.x * 16;
  HeadW wa, wb;                               // both halves of layer 0's weights are requested before the norm
  head_load_w(wa, p.w0, wave, lane, 0);
  head_load_w(wb, p.w0, wave, lane, 1);

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
  {
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    head_mma(acc, xs, wa, lane, 0);
    head_load_w(wa, p.w1, wave, lane, 0);     // layer 1's first half streams in under the second half of layer 0
    head_mma(acc, xs, wb, lane, 1);
    head_load_w(wb, p.w1, wave, lane, 1);
    head_store(acc, ys, p.b0, wave, lane);
  }
  __syncthreads();
  {
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    head_mma(acc, ys, wa, lane, 0);
    head_mma(acc, ys, wb, lane, 1);
    head_store(acc, xs, p.b1, wave, lane);   // xs is free: every wave passed the barrier after its layer-0 reads
  }
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
