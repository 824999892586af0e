// HBM-bound kernels of the COTR forward path: LayerNorm, lin_sine positional encoding,
// 3x3/2 max-pool on NHWC side-by-side activations, and the final 256 -> 2 regression head.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// LayerNorm over rows of 256 (nn.LayerNorm(256), eps 1e-5, biased variance): one wavefront per
// row, one float4 per lane, two-pass statistics in registers, wave reductions by shuffles.
// Call sites: COTR/models/transformer.py:155,158 (encoder), :198,201 (decoder), :110-111 (decoder.norm).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ y,
                                                        int rows) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)row * 256 + lane * 4);
  const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256.f);
  const f32x4 d = {v[0] - mean, v[1] - mean, v[2] - mean, v[3] - mean};
  const float var = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / 256.f);
  const float rstd = 1.f / sqrtf(var + 1e-5f);
  const f32x4 ww = *reinterpret_cast<const f32x4*>(w + lane * 4);
  const f32x4 bb = *reinterpret_cast<const f32x4*>(b + lane * 4);
  f32x4 out;
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = d[i] * rstd * ww[i] + bb[i];
  *reinterpret_cast<f32x4*>(y + (size_t)row * 256 + lane * 4) = out;
}

// LayerNorm of (sum of `np` partial outputs [np][rows][256] + bias + residual): the tail of the fused FFN block
// (ffn.hip): y = LN(residual + linear2(...)) with linear2's bias (transformer.py:156-158, 199-201).
__global__ __launch_bounds__(256) void ln_reduce_kernel(const float* __restrict__ parts, int np, const float* __restrict__ bias,
                                                        const float* __restrict__ residual, const float* __restrict__ w,
                                                        const float* __restrict__ b, const float* __restrict__ post_w,
                                                        const float* __restrict__ post_b, float* __restrict__ y, int rows) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  f32x4 v = *reinterpret_cast<const f32x4*>(bias + lane * 4);
  f32x4 rr = {0.f, 0.f, 0.f, 0.f};
  if (residual != nullptr) rr = *reinterpret_cast<const f32x4*>(residual + (size_t)row * 256 + lane * 4);
  v += rr;
  // the partial rows are independent loads: request 8 at a time and add them in order afterwards (a plain
  // "for c: v += load" is np dependent L2 round trips - the loop is not unrolled for a runtime np); same sum order as before
  const float* prow = parts + (size_t)row * 256 + lane * 4;
  const size_t pstride = (size_t)rows * 256;
  int c = 0;
  for (; c + 8 <= np; c += 8) {
    f32x4 t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = *reinterpret_cast<const f32x4*>(prow + (size_t)(c + k) * pstride);
#pragma unroll
    for (int k = 0; k < 8; ++k) v += t[k];
  }
  for (; c < np; ++c) v += *reinterpret_cast<const f32x4*>(prow + (size_t)c * pstride);
  const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256.f);
  const f32x4 d = {v[0] - mean, v[1] - mean, v[2] - mean, v[3] - mean};
  const float var = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / 256.f);
  const float rstd = 1.f / sqrtf(var + 1e-5f);
  const f32x4 ww = *reinterpret_cast<const f32x4*>(w + lane * 4);
  const f32x4 bb = *reinterpret_cast<const f32x4*>(b + lane * 4);
  f32x4 out;
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = d[i] * rstd * ww[i] + bb[i];
  if (post_w != nullptr) {   // a second LayerNorm of the result: decoder.norm after the last layer's norm3 (transformer.py:110-111)
    const float m2 = wave_sum(out[0] + out[1] + out[2] + out[3]) * (1.f / 256.f);
    const f32x4 d2 = {out[0] - m2, out[1] - m2, out[2] - m2, out[3] - m2};
    const float v2 = wave_sum(d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2] + d2[3] * d2[3]) * (1.f / 256.f);
    const float r2 = 1.f / sqrtf(v2 + 1e-5f);
    const f32x4 w2 = *reinterpret_cast<const f32x4*>(post_w + lane * 4);
    const f32x4 b2 = *reinterpret_cast<const f32x4*>(post_b + lane * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = d2[i] * r2 * w2[i] + b2[i];
  }
  *reinterpret_cast<f32x4*>(y + (size_t)row * 256 + lane * 4) = out;
}

int launch_ln_reduce_post(const float* parts, int np, const float* bias, const float* residual, const float* w, const float* b,
                          const float* post_w, const float* post_b, float* y, int rows, hipStream_t s) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(ln_reduce_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, parts, np, bias, residual, w, b,
                     post_w, post_b, y, rows);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}


int launch_ln_reduce(const float* parts, int np, const float* bias, const float* residual, const float* w, const float* b,
                     float* y, int rows, hipStream_t s) {
  return launch_ln_reduce_post(parts, np, bias, residual, w, b, nullptr, nullptr, y, rows, s);
}

int launch_layernorm(const float* x, const float* w, const float* b, float* y, int rows, hipStream_t s) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, w, b, y, rows);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------------
// lin_sine ("NeRF") encoding, COTR/models/position_encoding.py:30-45 with depth 64:
//   channel c < 128 : sin(k*pi*p[axis]),  k = c/2 + 1, axis = c%2 (0 = x, 1 = y)
//   channel c >= 128: cos(...) of the same (k, axis) for c-128
// `k*math.pi` is a Python double that torch rounds to fp32 before the fp32 multiply; the same
// two roundings are done here (arguments reach 64*pi ~ 201, where one fp32 ulp is 1.5e-5).
// sinf/cosf are the full-range ocml versions, not the fast intrinsics.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float lin_sine(float px, float py, int c) {
  const int cc = c & 127;
  const int kk = (cc >> 1) + 1;
  const float a = (float)((double)kk * 3.141592653589793);
  const float arg = __fmul_rn(a, (cc & 1) ? py : px);
  return (c < 128) ? sinf(arg) : cosf(arg);
}

__global__ __launch_bounds__(256) void posenc_kernel(const float* __restrict__ pts, float* __restrict__ y,
                                                     int nq, int q_total, int rows, int rpw) {
  // rpw rows per workgroup: one row per workgroup is 32000 workgroups of one sinf per thread at 32 x 1000 (launch-rate bound);
  // few rows keep one row per workgroup (1000 rows = 1000 workgroups fill the chip)
  for (int i = 0; i < rpw; ++i) {
    const int row = blockIdx.x * rpw + i;        // bi*nq + qi
    if (row >= rows) return;
    const int bi = row / nq, qi = row - bi * nq;
    const float* p = pts + ((size_t)bi * q_total + qi) * 2;
    y[(size_t)row * 256 + threadIdx.x] = lin_sine(p[0], p[1], threadIdx.x);
  }
}

int launch_posenc(const float* pts, float* y, int nb, int nq, int q_total, hipStream_t s) {
  if (nb * nq <= 0) return 0;
  const int rows = nb * nq;
  const int rpw = rows >= 16384 ? 8 : rows >= 4096 ? 2 : 1;
  hipLaunchKernelGGL(posenc_kernel, dim3((rows + rpw - 1) / rpw), dim3(256), 0, s, pts, y, nq, q_total, rows, rpw);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Image grid encoding, COTR/models/position_encoding.py:60-72 with an all-False mask on the 16x32
// feature map: x = (j + 0.5)/(32 + 1e-6), y = (i + 0.5)/(16 + 1e-6) evaluated in fp32 as the reference
// does: 32 + 1e-6 rounds to 32, but 16 + 1e-6 rounds UP to 16.0000019 (half an ulp of 16 is 0.95e-6),
// which moves sin(64*pi*y) by 2e-5 - so the sums are formed in fp32, not simplified.
// Token l = i*32 + j (flatten of [16,32], transformer.py:50-51).  Constant -> built once per handle.
__global__ __launch_bounds__(256) void pos_table_kernel(float* __restrict__ y) {
  const int l = blockIdx.x;
  const float px = __fdiv_rn((float)(l & 31) + 0.5f, 32.f + 1e-6f);
  const float py = __fdiv_rn((float)(l >> 5) + 0.5f, 16.f + 1e-6f);
  y[(size_t)l * 256 + threadIdx.x] = lin_sine(px, py, threadIdx.x);
}

int launch_pos_table(float* y, hipStream_t s) {
  hipLaunchKernelGGL(pos_table_kernel, dim3(512), dim3(256), 0, s, y);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------------
// 3x3 stride-2 pad-1 max-pool (torchvision resnet50.maxpool) on NHWC side-by-side
// [B,Hin,2*Win,C] -> [B,Hin/2,2*(Win/2),C]; padding never wins a max (treated as -inf) and a
// window never crosses the seam between the two halves.  One thread = one pixel x 4 channels.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      int total, int Hin, int Win, int C) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c4 = C / 4;
  const int Ho = Hin / 2, Wo = Win / 2;
  const int cg = idx % c4;
  int pix = idx / c4;
  const int wo = pix % (2 * Wo);
  pix /= (2 * Wo);
  const int ho = pix % Ho;
  const int b = pix / Ho;
  const int side = wo / Wo, wl = wo - side * Wo;
  f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int hi = 2 * ho - 1 + dy;
    if (hi < 0 || hi >= Hin) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int wi = 2 * wl - 1 + dx;
      if (wi < 0 || wi >= Win) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(
          x + (((size_t)b * Hin + hi) * (2 * Win) + side * Win + wi) * C + cg * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) m[i] = fmaxf(m[i], v[i]);
    }
  }
  *reinterpret_cast<f32x4*>(y + (((size_t)b * Ho + ho) * (2 * Wo) + wo) * C + cg * 4) = m;
}

int launch_maxpool(const float* x, float* y, int B, int Hin, int Win, int C, hipStream_t s) {
  const long total = (long)B * (Hin / 2) * (Win) * (C / 4);  // 2*(Win/2) = Win output columns
  if (total <= 0) return 0;
  hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, y, (int)total,
                     Hin, Win, C);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------------
// Last corr_embed layer, Linear(256, 2) without activation (COTR/models/position_encoding.py:23-26,
// cotr_model.py:21,38): one wavefront per query row, scattered to out[b][q][0..1].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ b, float* __restrict__ y,
                                                    int rows, int nq, int q_total) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)row * 256 + lane * 4);
  const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + lane * 4);
  const f32x4 w1 = *reinterpret_cast<const f32x4*>(w + 256 + lane * 4);
  float s0 = v[0] * w0[0] + v[1] * w0[1] + v[2] * w0[2] + v[3] * w0[3];
  float s1 = v[0] * w1[0] + v[1] * w1[1] + v[2] * w1[2] + v[3] * w1[3];
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  if (lane == 0) {
    const int bi = row / nq, qi = row - bi * nq;
    float* dst = y + ((size_t)bi * q_total + qi) * 2;
    dst[0] = s0 + b[0];
    dst[1] = s1 + b[1];
  }
}

int launch_head2(const float* x, const float* w, const float* b, float* y, int nb, int nq, int q_total,
                 hipStream_t s) {
  const int rows = nb * nq;
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(head2_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, w, b, y, rows, nq, q_total);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
