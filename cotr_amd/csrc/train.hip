// Kernels of the TRAINING step (COTR/trainers/cotr_trainer.py:118-150 on COTR/models/transformer.py
// with dropout active), fp32, gfx950.  The contractions of the forward pass reuse the inference GEMM kernels (gemm.hip); this
// file holds what only training needs:
//   add_drop_ln_fwd / ln_bwd      y = LayerNorm(x + dropout(a)) of every sub-layer (transformer.py:154-158,196-201) and its
//                                 backward (dx, da, dgamma, dbeta); plain LayerNorm with x == nullptr, p == 0
//   dropout_fwd / relu_drop_bwd   dropout(relu(linear1(x))) of the feed-forward block (:156,199)
//   colsum / sum_parts            bias gradients and every other fixed-order cross-workgroup reduction (no atomics:
//                                 a training step is bit-repeatable)
//   gemm_tn                       dW = dY^T . X with both operands row-major (contraction over the strided row index),
//                                 split over M, partial products summed in a fixed order
//   transpose                     W -> W^T once per optimiser step for dX = dY . W on the inference GEMM kernels
//   head_bwd                      backward of the 256 -> 2 output layer (position_encoding.py:23-26)
// Attention forward (with dropout on the probabilities and the log-sum-exp saved) and backward live in attention_train.hip.
//
// Dropout masks are counter-based: keep(element) = hash(seed, element index) >= p * 2^32.  The backward kernels recompute
// them from the same (seed, index), nothing is stored.  The masks are not torch's (no parity requirement on a random mask;
// with p = 0 every kernel is exact and that is what the goldens pin).
#include <atomic>

#include "common.h"
#include "train.h"

namespace {

__device__ __forceinline__ float tr_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// y = LayerNorm(s), s = x + dropout(a)   (x may be nullptr: s = dropout(a));  one wavefront per row of 256, one float4 per
// lane, two-pass statistics like layernorm_kernel (pointwise.hip).  s and (mean, rstd) are kept for the backward.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_drop_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                              const float* __restrict__ w, const float* __restrict__ b,
                                                              float* __restrict__ s_out, float* __restrict__ y,
                                                              float* __restrict__ stats, int rows, uint32_t thresh, float inv_keep,
                                                              uint32_t seed, const uint32_t* __restrict__ salt) {
  seed = train_salted(seed, salt);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const size_t base = (size_t)row * 256 + lane * 4;
  f32x4 v = *reinterpret_cast<const f32x4*>(a + base);
  if (thresh != 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = train_keep(seed, base + e, thresh) ? v[e] * inv_keep : 0.f;
  }
  if (x != nullptr) v += *reinterpret_cast<const f32x4*>(x + base);
  const float mean = tr_wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256.f);
  const f32x4 d = {v[0] - mean, v[1] - mean, v[2] - mean, v[3] - mean};
  const float var = tr_wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / 256.f);
  const float rstd = 1.f / sqrtf(var + 1e-5f);
  const f32x4 ww = *reinterpret_cast<const f32x4*>(w + lane * 4);
  const f32x4 bb = *reinterpret_cast<const f32x4*>(b + lane * 4);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = d[e] * rstd * ww[e] + bb[e];
  *reinterpret_cast<f32x4*>(y + base) = o;
  if (s_out != nullptr) *reinterpret_cast<f32x4*>(s_out + base) = v;
  if (lane == 0) {
    stats[2 * row] = mean;
    stats[2 * row + 1] = rstd;
  }
}

// Backward of the above: with xhat = (s - mean) * rstd and g = dy * gamma,
//   ds = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat))       (= dx; da = ds * mask / (1 - p))
//   dgamma = sum_rows dy * xhat,  dbeta = sum_rows dy           (per-workgroup partial sums, finished by sum_parts)
// A workgroup owns rows_per_wg consecutive rows (a wavefront every 4th of them); the column sums stay in registers (lane =
// 4 channels) and are combined across the 4 wavefronts through LDS in a fixed order.
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ s,
                                                     const float* __restrict__ stats, const float* __restrict__ w,
                                                     float* __restrict__ ds, float* __restrict__ da, float* __restrict__ part,
                                                     int rows, int rows_per_wg, uint32_t thresh, float inv_keep, uint32_t seed,
                                                     const uint32_t* __restrict__ salt) {
  seed = train_salted(seed, salt);
  __shared__ __attribute__((aligned(16))) float red[2][4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f32x4 ww = *reinterpret_cast<const f32x4*>(w + lane * 4);
  f32x4 gw = {0.f, 0.f, 0.f, 0.f}, gb = {0.f, 0.f, 0.f, 0.f};
  const int r0 = blockIdx.x * rows_per_wg;
  for (int i = wave; i < rows_per_wg; i += 4) {
    const int row = r0 + i;
    if (row >= rows) break;
    const size_t base = (size_t)row * 256 + lane * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + base);
    const f32x4 sv = *reinterpret_cast<const f32x4*>(s + base);
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    f32x4 xh, gg;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      xh[e] = (sv[e] - mean) * rstd;
      gg[e] = g[e] * ww[e];
      gw[e] += g[e] * xh[e];
      gb[e] += g[e];
    }
    const float m1 = tr_wave_sum(gg[0] + gg[1] + gg[2] + gg[3]) * (1.f / 256.f);
    const float m2 = tr_wave_sum(gg[0] * xh[0] + gg[1] * xh[1] + gg[2] * xh[2] + gg[3] * xh[3]) * (1.f / 256.f);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = rstd * (gg[e] - m1 - xh[e] * m2);
    *reinterpret_cast<f32x4*>(ds + base) = o;
    if (da != nullptr) {
      if (thresh != 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = train_keep(seed, base + e, thresh) ? o[e] * inv_keep : 0.f;
      }
      *reinterpret_cast<f32x4*>(da + base) = o;
    }
  }
  *reinterpret_cast<f32x4*>(&red[0][wave][lane * 4]) = gw;
  *reinterpret_cast<f32x4*>(&red[1][wave][lane * 4]) = gb;
  __syncthreads();
  const int c = threadIdx.x;   // 256 threads = 256 channels
  float sw = 0.f, sb = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    sw += red[0][k][c];
    sb += red[1][k][c];
  }
  part[(size_t)blockIdx.x * 512 + c] = sw;          // [workgroup][dgamma 256 | dbeta 256]
  part[(size_t)blockIdx.x * 512 + 256 + c] = sb;
}

// out[i] = sum_p part[p][i], p in order (deterministic)
__global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ part, int nparts, size_t numel,
                                                        float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= numel) return;
  float acc = 0.f;
  int p = 0;
  for (; p + 8 <= nparts; p += 8) {   // independent loads first, adds in order afterwards
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = part[(size_t)(p + k) * numel + i];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += t[k];
  }
  for (; p < nparts; ++p) acc += part[(size_t)p * numel + i];
  out[i] = acc;
}

// ---------------------------------------------------------------------------------------------------------------------
// Deferred gradient reduction (cotr_train_reduce_jobs): ONE launch finishes every split-M / per-workgroup partial of a whole
// backward pass and accumulates it into the persistent gradient buffers -
//   dst[perm(e)] += scale[row(e)] * sum_p part[p][e]   for every source of the job in order, p in order (deterministic),
// instead of one sum_parts launch per weight followed by autograd's own add_ into .grad (and, for the convolutions, a row
// scale and a layout transpose).  The partials of a step stay where the producing kernels wrote them until the flush (about
// 1-2 GB at 16 pairs x 200 queries: the HBM is 288 GB).  A job = one destination (a parameter gradient, or a row slice of one);
// its sources = the uses of that parameter in the step, in the order autograd ran them (the decoder's weights are used by
// the first and by the cycle pass).  Workgroup = 1024 consecutive elements of one job (chunk_job[workgroup] = its job).  Same additions in the same order as sum_parts + scale_rows + add_: bit-identical gradients.
//   perm: dst index of packed element e = row * row_len + tap * cin + c  is  row * row_len + c * taps + tap  (the packed
//   [Cout][k][k][Cin] layout of the conv kernels back to torch's [Cout][Cin][k][k]); taps == 1: identity.
// ---------------------------------------------------------------------------------------------------------------------
// (rounds of 32 independent loads, added in order afterwards: a thread that owns a LayerNorm gradient element walks ~500 partials
// per use of the weight - with 8 loads per round that walk was the whole launch's critical path)
__device__ __forceinline__ float reduce_one(const TrainReduceSrc& sc, unsigned e) {
  const float* p0 = sc.part + e;
  float acc = 0.f;
  unsigned p = 0;
  for (; p + 32 <= sc.nparts; p += 32) {
    float t[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) t[k] = p0[(size_t)(p + k) * sc.pstride];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc += t[k];
  }
  for (; p + 8 <= sc.nparts; p += 8) {
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = p0[(size_t)(p + k) * sc.pstride];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += t[k];
  }
  for (; p < sc.nparts; ++p) acc += p0[(size_t)p * sc.pstride];
  return acc;
}
__device__ __forceinline__ f32x4 reduce_four(const TrainReduceSrc& sc, unsigned e) {
  const float* p0 = sc.part + e;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  unsigned p = 0;
  for (; p + 32 <= sc.nparts; p += 32) {
    f32x4 t[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) t[k] = *reinterpret_cast<const f32x4*>(p0 + (size_t)(p + k) * sc.pstride);
#pragma unroll
    for (int k = 0; k < 32; ++k) acc += t[k];
  }
  for (; p + 8 <= sc.nparts; p += 8) {
    f32x4 t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = *reinterpret_cast<const f32x4*>(p0 + (size_t)(p + k) * sc.pstride);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += t[k];
  }
  for (; p < sc.nparts; ++p) acc += *reinterpret_cast<const f32x4*>(p0 + (size_t)p * sc.pstride);
  return acc;
}

__global__ __launch_bounds__(256) void reduce_jobs_kernel(const TrainReduceJob* __restrict__ jobs,
                                                          const TrainReduceSrc* __restrict__ srcs,
                                                          const unsigned* __restrict__ chunk_job) {
  const TrainReduceJob j = jobs[chunk_job[blockIdx.x]];
  const unsigned base = (blockIdx.x - j.chunk0) * 1024u;
  if (j.vec) {                                         // every pointer 16-byte aligned, every count a multiple of 4
    const unsigned e = base + threadIdx.x * 4;
    if (e >= j.numel) return;
    unsigned d = e;
    float sc = 1.f;
    if (j.row_len) {
      const unsigned row = e / j.row_len, col = e - row * j.row_len;
      if (j.scale != nullptr) sc = j.scale[row];
      if (j.taps > 1) {
        const unsigned tap = col / j.cin, c = col - tap * j.cin;
        d = row * j.row_len + c * j.taps + tap;
      }
    }
    f32x4 tot;
    if (j.taps > 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) tot[i] = j.dst[d + i * j.taps];
    } else
      tot = *reinterpret_cast<const f32x4*>(j.dst + d);
    for (unsigned s = 0; s < j.n_src; ++s) {
      f32x4 a = reduce_four(srcs[j.first_src + s], e);
      if (j.scale != nullptr) {                        // (a separate rounding, as scale_rows + add_ make: no fused multiply-add)
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = __fmul_rn(a[i], sc);
      }
      tot += a;
    }
    if (j.taps > 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) j.dst[d + i * j.taps] = tot[i];
    } else
      *reinterpret_cast<f32x4*>(j.dst + d) = tot;
    return;
  }
#pragma unroll 1
  for (int i = 0; i < 4; ++i) {
    const unsigned e = base + i * 256 + threadIdx.x;
    if (e >= j.numel) return;
    unsigned d = e;
    float sc = 1.f;
    if (j.row_len) {
      const unsigned row = e / j.row_len, col = e - row * j.row_len;
      if (j.scale != nullptr) sc = j.scale[row];
      if (j.taps > 1) {
        const unsigned tap = col / j.cin, c = col - tap * j.cin;
        d = row * j.row_len + c * j.taps + tap;
      }
    }
    float tot = j.dst[d];
    for (unsigned s = 0; s < j.n_src; ++s) {
      float a = reduce_one(srcs[j.first_src + s], e);
      if (j.scale != nullptr) a = __fmul_rn(a, sc);
      tot += a;
    }
    j.dst[d] = tot;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Adam over every trainable parameter in ONE launch (cotr_train_adam): torch.optim.Adam's update (train_cotr.py:49-57: betas
// (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad) as torch's multi-tensor path computes it -
//   m += (g - m) (1 - b1);  v = v b2 + (1 - b2) g g;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// - on the GradSink's flat gradient buffer and flat m / v buffers of the same layout; the parameters themselves stay where the
// model keeps them (one job per parameter: pointer, offset into the flat buffers, learning-rate group).  torch's own step is
// ~15 multi-tensor launches that each stream the state once (0.57 ms for the 10.4 M stage-1 parameters); this is one pass.
// step_ptr != nullptr: the step count is read from device memory (captured steps: the host does not know it).
// ---------------------------------------------------------------------------------------------------------------------
struct AdamArgs {
  float lr[8];                   // per parameter group
  float b1, b2, eps;
  float omb1, omb2;              // 1 - b1, 1 - b2 rounded from DOUBLE as torch passes them (1 - 0.999f in float is 4.7e-5 off 0.001)
  float bc1, bc2_sqrt;           // 1 - b1^t, sqrt(1 - b2^t) (host, double precision) when step_ptr == nullptr
  const float* step_ptr;
};
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float step_size, float bc2_sqrt, float omb1, float b2,
                                         float omb2, float eps) {
  m = fmaf(omb1, g - m, m);
  v = fmaf(omb2 * g, g, v * b2);
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p = p - step_size * (m / denom);
}
__global__ __launch_bounds__(256) void adam_jobs_kernel(const TrainAdamJob* __restrict__ jobs, const unsigned* __restrict__ chunk_job,
                                                        const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                        AdamArgs a) {
  const TrainAdamJob j = jobs[chunk_job[blockIdx.x]];
  float bc1 = a.bc1, bc2_sqrt = a.bc2_sqrt;
  if (a.step_ptr != nullptr) {
    const float t = *a.step_ptr;
    bc1 = 1.f - powf(a.b1, t);
    bc2_sqrt = sqrtf(1.f - powf(a.b2, t));
  }
  const float step_size = a.lr[j.group] / bc1;
  const unsigned e = (blockIdx.x - j.chunk0) * 1024u + threadIdx.x * 4;
  if (e >= j.numel) return;
  const size_t o = j.off + e;
  if (e + 4 <= j.numel && j.vec) {
    f32x4 pp = *reinterpret_cast<const f32x4*>(j.p + e);
    const f32x4 gg = *reinterpret_cast<const f32x4*>(g + o);
    f32x4 mm = *reinterpret_cast<const f32x4*>(m + o), vv = *reinterpret_cast<const f32x4*>(v + o);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float p1 = pp[i], m1 = mm[i], v1 = vv[i];
      adam_one(p1, gg[i], m1, v1, step_size, bc2_sqrt, a.omb1, a.b2, a.omb2, a.eps);
      pp[i] = p1;
      mm[i] = m1;
      vv[i] = v1;
    }
    *reinterpret_cast<f32x4*>(j.p + e) = pp;
    *reinterpret_cast<f32x4*>(m + o) = mm;
    *reinterpret_cast<f32x4*>(v + o) = vv;
    return;
  }
  for (unsigned i = 0; i < 4 && e + i < j.numel; ++i) {
    float pp = j.p[e + i], mm = m[o + i], vv = v[o + i];
    adam_one(pp, g[o + i], mm, vv, step_size, bc2_sqrt, a.omb1, a.b2, a.omb2, a.eps);
    j.p[e + i] = pp;
    m[o + i] = mm;
    v[o + i] = vv;
  }
}

// y[m][:] = x[m][:] + x2[m % mod][:] over rows of 256 (mod == 0: x2 has one row per row of x): src + pos, tgt + query_pos
__global__ __launch_bounds__(256) void add_rowmod_kernel(const float* __restrict__ x, const float* __restrict__ x2, int mod,
                                                         float* __restrict__ y, int rows) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int r2 = mod ? row % mod : row;
  const f32x4 a = *reinterpret_cast<const f32x4*>(x + (size_t)row * 256 + lane * 4);
  const f32x4 b = *reinterpret_cast<const f32x4*>(x2 + (size_t)r2 * 256 + lane * 4);
  *reinterpret_cast<f32x4*>(y + (size_t)row * 256 + lane * 4) = a + b;
}

// x *= mask / (1 - p), element index = offset of the element in x
__global__ __launch_bounds__(256) void dropout_fwd_kernel(float* __restrict__ x, size_t n4, uint32_t thresh, float inv_keep,
                                                          uint32_t seed, const uint32_t* __restrict__ salt) {
  seed = train_salted(seed, salt);
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 v = *reinterpret_cast<f32x4*>(x + i * 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = train_keep(seed, i * 4 + e, thresh) ? v[e] * inv_keep : 0.f;
  *reinterpret_cast<f32x4*>(x + i * 4) = v;
}

// backward of y = dropout(relu(h)): dh = y > 0 ? dy / (1 - p) : 0 (a dropped or non-positive element has y == 0; torch's relu
// backward is (result > 0) as well); p == 0: plain relu backward
__global__ __launch_bounds__(256) void relu_drop_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                            float* __restrict__ dx, size_t n4, float inv_keep) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 g = *reinterpret_cast<const f32x4*>(dy + i * 4);
  const f32x4 v = *reinterpret_cast<const f32x4*>(y + i * 4);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = v[e] > 0.f ? g[e] * inv_keep : 0.f;
  *reinterpret_cast<f32x4*>(dx + i * 4) = o;
}

// part[wg][n] = sum over the workgroup's rows of x[row][n]  (bias gradients); N <= 4096, N % 4 == 0
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ part, int M, int N,
                                                     int rows_per_wg) {
  const int r0 = blockIdx.x * rows_per_wg;
  const int r1 = (r0 + rows_per_wg < M) ? r0 + rows_per_wg : M;
  for (int c = threadIdx.x * 4; c < N; c += 1024) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = r0; r < r1; ++r) acc += *reinterpret_cast<const f32x4*>(x + (size_t)r * N + c);
    *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.x * N + c) = acc;
  }
}

// dst[C][R] = src[R][C]^T through a 32 x 33 LDS tile (coalesced both ways)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < C && r < R) dst[(size_t)c * R + r] = tile[tx][ty + 8 * i];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// part[split][n][k] = sum_{m in split} A[m][n] * B[m][k]:  dW = dY^T . X with A = dY [M][N], B = X [M][K] both ROW-major (the
// contraction index m is the strided one: no operand is transposed).  Workgroup = 4 wavefronts = 64 x 64 output tile (each a
// 32 x 32 block of v_mfma_f32_32x32x2_f32), walking its share of the rows 32 at a time through LDS: a fragment is one dword
// per lane per MFMA (lane = output row / column, the two halves of the wavefront = the two rows of the K = 2 step), read
// with ds_read_b32 at consecutive addresses (conflict-free).  Rows past M are zero-filled.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                      float* __restrict__ part, int M, int N, int K, int rows_per_split,
                                                      int with_colsum) {
  __shared__ __attribute__((aligned(16))) float As[32][68];
  __shared__ __attribute__((aligned(16))) float Bs[32][68];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int tiles_k = K / 64;
  const int n0 = (blockIdx.x / tiles_k) * 64, k0 = (blockIdx.x % tiles_k) * 64;
  const int split = blockIdx.y;
  const int m_begin = split * rows_per_split;
  const int m_end = (m_begin + rows_per_split < M) ? m_begin + rows_per_split : M;
  const int wn = wave >> 1, wk = wave & 1;
  // staging: thread -> (row t >> 4 (+16), float4 column (t & 15) * 4) of the 32 x 64 tile
  const int sr = t >> 4, sc = (t & 15) * 4;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // two chunks of 32 rows in flight in registers (the loop is otherwise bound by one global-load latency per 16 MFMAs)
  f32x4 ra0[2], rb0[2], ra1[2], rb1[2];
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};      // column sums of A (the bias gradient) over this thread's rows; k-tile 0 only
  const bool do_colsum = with_colsum && k0 == 0;
  auto load = [&](int m, f32x4 (&ra)[2], f32x4 (&rb)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = m + sr + 16 * i;
      const int rc = row < m_end ? row : m_begin;          // clamped address, value zeroed below (no branch around the load)
      ra[i] = *reinterpret_cast<const f32x4*>(A + (size_t)rc * N + n0 + sc);
      rb[i] = *reinterpret_cast<const f32x4*>(B + (size_t)rc * K + k0 + sc);
      if (row >= m_end) {
        ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        rb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto step = [&](f32x4 (&ra)[2], f32x4 (&rb)[2], int m_next) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<f32x4*>(&As[sr + 16 * i][sc]) = ra[i];
      *reinterpret_cast<f32x4*>(&Bs[sr + 16 * i][sc]) = rb[i];
      if (do_colsum) csum += ra[i];
    }
    __syncthreads();
    if (m_next < m_end) load(m_next, ra, rb);             // this register set is free again: request the chunk after next
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      const float a = As[2 * s2 + hh][wn * 32 + l31];   // A operand: lane i = output row n, k = row 2*s2 + hh of the chunk
      const float b = Bs[2 * s2 + hh][wk * 32 + l31];   // B operand: lane j = output column k
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  };
  load(m_begin, ra0, rb0);
  if (m_begin + 32 < m_end) load(m_begin + 32, ra1, rb1);
  for (int m = m_begin; m < m_end; m += 64) {
    step(ra0, rb0, m + 64);
    if (m + 32 < m_end) step(ra1, rb1, m + 96);
  }
  // D[i][j]: column j = lane & 31 (k), rows i = (r&3) + 8*(r>>2) + 4*hh (n).  One partial = [N*K products | N column sums]
  const size_t pstride = (size_t)N * K + (with_colsum ? N : 0);
  float* out = part + (size_t)split * pstride;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
    out[(size_t)n * K + k0 + wk * 32 + l31] = acc[r];
  }
  if (do_colsum) {   // 16 row-threads per column group: fixed-order sum through LDS (As is free after the last barrier pair)
    __syncthreads();
    *reinterpret_cast<f32x4*>(&As[sr][sc]) = csum;
    __syncthreads();
    if (t < 64) {
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += As[r][t];
      out[(size_t)N * K + n0 + t] = sum;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same contraction for the WIDE weights (linear1 / linear2 1024 x 256, input_proj, the packed q|k slice): 128 x 128 output
// tile, 4 wavefronts each 64 x 64 (2 x 2 MFMA blocks: four MFMAs per pair of ds_read2_b32, against one MFMA per two ds_read_b32
// in the 64 x 64 kernel), operands by LDS-DMA - a wave instruction moves two 128-wide tile rows (lanes 0-31 / 32-63), the tile
// is stored unpadded (a fragment read is 32 consecutive dwords per half-wave: conflict-free at any row stride) - two stages,
// two workgroups per CU.  Column sums of A (bias gradient) by the k-tile-0 workgroups from the landed A tile.
// ---------------------------------------------------------------------------------------------------------------------
// CONV (round 6): B is not a matrix but the NHWC side-by-side activation x of a convolution, and "row m, columns k0 .. k0+127" of
// its im2col image is gathered by the DMA itself - pixel m shifted by the tap of this column block, 128 of its channels; zeros
// where the tap leaves the half (the same zeros im2col_kernel writes).  The weight gradient of a 3x3 / strided convolution then
// needs no im2col launch and no [M][k*k*Cin] buffer (75 MB per layer3 convolution at 16 pairs): same operands in the same order
// as gemm_tn on the explicit image - the same bits.  Needs Cin % 128 == 0 (a column block lies inside one tap).
struct ConvGeo {
  int B, Hin, Win, Cin, Hout, Wout, ksize, stride, pad;
};
template <bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_tn_big_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                             float* __restrict__ part, int M, int N, int K, int rows_per_split,
                                                             int with_colsum, const float* __restrict__ zeros, const ConvGeo g) {
  extern __shared__ __attribute__((aligned(16))) float tn_smem[];      // [2 stages][A 32 x 128 | B 32 x 128]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int tiles_k = K / 128;
  const int n0 = (blockIdx.x / tiles_k) * 128, k0 = (blockIdx.x % tiles_k) * 128;
  const int split = blockIdx.y;
  const int m_begin = split * rows_per_split;
  const int m_end = (m_begin + rows_per_split < M) ? m_begin + rows_per_split : M;
  const int wn = wave >> 1, wk = wave & 1;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const bool do_colsum = with_colsum && k0 == 0;
  float csum = 0.f;                                      // threads 0..127: column n0 + t of A over this split's rows

  // CONV: the tap (ky, kx) and first channel c0 of this workgroup's column block (wave-uniform)
  int ky = 0, kx = 0, c0 = 0;
  if constexpr (CONV) {
    const int tap = k0 / g.Cin;
    c0 = k0 - tap * g.Cin;
    ky = tap / g.ksize;
    kx = tap - ky * g.ksize;
  }
  // wavefront w moves tile rows 8w .. 8w+7 of both operands: 4 DMA instructions per operand, each two rows
  auto dma_chunk = [&](int m, int st) {
    float* As = tn_smem + st * (2 * 32 * 128);
    float* Bs = As + 32 * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = wv * 8 + 2 * i;                      // (wave-uniform: the LDS destination)
      const int row = m + r + hh;
      const bool ok = row < m_end;
      const float* sa = ok ? A + (size_t)row * N + n0 + l31 * 4 : zeros;
      const float* sb;
      if constexpr (CONV) {
        // output pixel `row` = (b, ho, side, wl) of [B][Hout][2*Wout]; its input pixel under this tap
        const int w2o = 2 * g.Wout, rr = ok ? row : 0;
        const int bq = rr / (g.Hout * w2o), rem = rr - bq * (g.Hout * w2o);
        const int ho = rem / w2o, wo = rem - ho * w2o;
        const int side = wo >= g.Wout ? 1 : 0, wl = wo - side * g.Wout;
        const int hi = ho * g.stride - g.pad + ky, wi = wl * g.stride - g.pad + kx;
        const bool in = ok && hi >= 0 && hi < g.Hin && wi >= 0 && wi < g.Win;
        sb = in ? B + (((size_t)bq * g.Hin + hi) * (2 * g.Win) + side * g.Win + wi) * g.Cin + c0 + l31 * 4 : zeros;
      } else {
        sb = ok ? B + (size_t)row * K + k0 + l31 * 4 : zeros;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sa,
                                       (__attribute__((address_space(3))) void*)(As + r * 128), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sb,
                                       (__attribute__((address_space(3))) void*)(Bs + r * 128), 16, 0, 0);
    }
  };
  dma_chunk(m_begin, 0);
  int st = 0;
  for (int m = m_begin; m < m_end; m += 32) {
    LDS_DMA_WAIT_ALL();
    __syncthreads();                                     // chunk m has landed for everybody; the other stage is free
    if (m + 32 < m_end) dma_chunk(m + 32, st ^ 1);
    const float* As = tn_smem + st * (2 * 32 * 128);
    const float* Bs = As + 32 * 128;
    if (do_colsum && t < 128) {
#pragma unroll
      for (int r = 0; r < 32; ++r) csum += As[r * 128 + t];   // fixed order: rows ascending
    }
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      const float* ar = As + (2 * s2 + hh) * 128 + wn * 64 + l31;
      const float* br = Bs + (2 * s2 + hh) * 128 + wk * 64 + l31;
      const float a0 = ar[0], a1 = ar[32], b0 = br[0], b1 = br[32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    st ^= 1;
  }
  const size_t pstride = (size_t)N * K + (with_colsum ? N : 0);
  float* out = part + (size_t)split * pstride;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        out[(size_t)n * K + k0 + wk * 64 + b * 32 + l31] = acc[a][b][r];
      }
  if (do_colsum && t < 128) out[(size_t)N * K + n0 + t] = csum;
}

// ---------------------------------------------------------------------------------------------------------------------
// Convolution backward of the trainable backbone stages (layer2 / layer3, COTR/models/backbone.py:66-69) by explicit
// im2col: with col[m][(ky*k + kx)*Cin + c] = x[pixel(m) shifted by the tap][c] (zero where the tap leaves the 256-wide half:
// the two halves of a side-by-side pair are padded separately, as in the forward kernels),
//     wgrad   dW[Cout][k*k*Cin] = dz^T . col          (gemm_tn)
//     dgrad   dcol = dz . W ,  dx[pixel] = sum of the dcol entries that read it   (GEMM + col2im gather, fixed tap order)
// Geometry as in GemmParams: activations NHWC side-by-side [B][Hin][2*Win][Cin], outputs [B][Hout][2*Wout].
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, float* __restrict__ col, ConvGeo g, size_t total4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int c4 = g.Cin / 4;
  const int c = (int)(i % c4) * 4;
  size_t r = i / c4;
  const int tap = (int)(r % (g.ksize * g.ksize));
  const size_t m = r / (g.ksize * g.ksize);
  const int W2o = 2 * g.Wout;
  const int wo = (int)(m % W2o);
  const int ho = (int)((m / W2o) % g.Hout);
  const int b = (int)(m / ((size_t)W2o * g.Hout));
  const int side = wo / g.Wout, wl = wo - side * g.Wout;
  const int ky = tap / g.ksize, kx = tap - ky * g.ksize;
  const int hi = ho * g.stride - g.pad + ky, wi = wl * g.stride - g.pad + kx;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (hi >= 0 && hi < g.Hin && wi >= 0 && wi < g.Win)
    v = *reinterpret_cast<const f32x4*>(x + (((size_t)b * g.Hin + hi) * (2 * g.Win) + side * g.Win + wi) * g.Cin + c);
  *reinterpret_cast<f32x4*>(col + i * 4) = v;
}

__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcol, float* __restrict__ dx, ConvGeo g, size_t total4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int c4 = g.Cin / 4;
  const int c = (int)(i % c4) * 4;
  size_t pix = i / c4;
  const int wi2 = (int)(pix % (2 * g.Win));
  const int hi = (int)((pix / (2 * g.Win)) % g.Hin);
  const int b = (int)(pix / ((size_t)2 * g.Win * g.Hin));
  const int side = wi2 / g.Win, wi = wi2 - side * g.Win;
  const int K = g.ksize * g.ksize * g.Cin;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int ky = 0; ky < g.ksize; ++ky) {
    const int hn = hi + g.pad - ky;
    if (hn < 0 || hn % g.stride) continue;
    const int ho = hn / g.stride;
    if (ho >= g.Hout) continue;
    for (int kx = 0; kx < g.ksize; ++kx) {
      const int wn = wi + g.pad - kx;
      if (wn < 0 || wn % g.stride) continue;
      const int wo = wn / g.stride;
      if (wo >= g.Wout) continue;
      const size_t m = ((size_t)b * g.Hout + ho) * (2 * g.Wout) + side * g.Wout + wo;
      acc += *reinterpret_cast<const f32x4*>(dcol + m * K + (ky * g.ksize + kx) * g.Cin + c);
    }
  }
  *reinterpret_cast<f32x4*>(dx + i * 4) = acc;
}

// out[r][c] = w[r][c] * scale[r]   (FrozenBN folded into the weights for the backward: conv(x, W) * scale = conv(x, W * scale))
// (w and out may be the SAME buffer - ConvBN.backward scales in place - so neither is __restrict__)
__global__ __launch_bounds__(256) void scale_rows_kernel(const float* w, const float* __restrict__ scale, float* out, int rows,
                                                         int cols4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)rows * cols4) return;
  const float sc = scale[i / cols4];
  f32x4 v = *reinterpret_cast<const f32x4*>(w + i * 4);
  v *= sc;
  *reinterpret_cast<f32x4*>(out + i * 4) = v;
}

// per batch z: dst[z][C][R] = src[z][R][C]^T  (conv weights [Cout][Cin][k*k] <-> [Cout][k*k][Cin])
__global__ __launch_bounds__(256) void transpose_batched_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C) {
  __shared__ float tile[32][33];
  const size_t off = (size_t)blockIdx.z * R * C;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < R && c < C) ? src[off + (size_t)r * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < C && r < R) dst[off + (size_t)c * R + r] = tile[tx][ty + 8 * i];
  }
}

// backward of pred = h . W2^T + b2 (W2 [2][256]): dh[r][c] = dy[r][0] * W2[0][c] + dy[r][1] * W2[1][c];
// part[wg][0..511] = per-workgroup sums of dy[r][j] * h[r][c] (j = 0, 1), part[wg][512..513] = sums of dy[r][j]
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ hin,
                                                       const float* __restrict__ w2, float* __restrict__ dh,
                                                       float* __restrict__ part, int rows, int rows_per_wg) {
  __shared__ float red[4][2][256];
  __shared__ float redb[4][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f32x4 w0 = *reinterpret_cast<const f32x4*>(w2 + lane * 4);
  const f32x4 w1 = *reinterpret_cast<const f32x4*>(w2 + 256 + lane * 4);
  f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = {0.f, 0.f, 0.f, 0.f};
  float b0 = 0.f, b1 = 0.f;
  const int r0 = blockIdx.x * rows_per_wg;
  for (int i = wave; i < rows_per_wg; i += 4) {
    const int row = r0 + i;
    if (row >= rows) break;
    const float d0 = dy[2 * row], d1 = dy[2 * row + 1];
    const f32x4 h = *reinterpret_cast<const f32x4*>(hin + (size_t)row * 256 + lane * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = d0 * w0[e] + d1 * w1[e];
      g0[e] += d0 * h[e];
      g1[e] += d1 * h[e];
    }
    *reinterpret_cast<f32x4*>(dh + (size_t)row * 256 + lane * 4) = o;
    b0 += d0;
    b1 += d1;
  }
  *reinterpret_cast<f32x4*>(&red[wave][0][lane * 4]) = g0;
  *reinterpret_cast<f32x4*>(&red[wave][1][lane * 4]) = g1;
  if (lane == 0) {
    redb[wave][0] = b0;
    redb[wave][1] = b1;
  }
  __syncthreads();
  const int c = threadIdx.x;
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    s0 += red[k][0][c];
    s1 += red[k][1][c];
  }
  float* p = part + (size_t)blockIdx.x * 514;
  p[c] = s0;
  p[256 + c] = s1;
  if (c < 2) p[512 + c] = redb[0][c] + redb[1][c] + redb[2][c] + redb[3][c];
}

}  // namespace

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? 0 : -2)

// The salt word is per DEVICE and process-wide, not per host thread: PyTorch runs the backward of a CUDA autograd Function on
// the autograd engine's per-device worker thread, and the kernels that RECOMPUTE a dropout mask there (ln_bwd, attn_bwd_dq,
// attn_bwd_dkv) must see the salt the forward kernels drew it with on the Python thread.  (Round 2 had it thread_local: with a
// salt registered, forward masks came from seed ^ salt and backward masks from seed alone.)
static std::atomic<const uint32_t*> g_salt[COTR_MAX_DEVICES];
const uint32_t* train_salt_ptr() { return g_salt[cotr_current_device()].load(std::memory_order_acquire); }
void train_set_salt_ptr(const uint32_t* p) { g_salt[cotr_current_device()].store(p, std::memory_order_release); }
// clear the registration only if it is still `expected` (a closing GraphedTrainStep must not unregister another one's salt)
bool train_clear_salt_ptr_if(const uint32_t* expected) {
  const uint32_t* e = expected;
  return g_salt[cotr_current_device()].compare_exchange_strong(e, nullptr, std::memory_order_acq_rel);
}

int train_add_drop_ln_fwd(const float* x, const float* a, const float* w, const float* b, float* s_out, float* y, float* stats,
                          int rows, float p, uint32_t seed, hipStream_t s) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(add_drop_ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, a, w, b, s_out, y, stats, rows,
                     train_thresh(p), p > 0.f ? 1.f / (1.f - p) : 1.f, seed, train_salt_ptr());
  return LAUNCH_OK();
}

int train_ln_bwd_parts(int rows) {
  int per = (rows + 511) / 512;
  per = (per + 3) / 4 * 4;
  return (rows + per - 1) / per;
}

// ds (and da = ds * mask / (1 - p) when da != nullptr); dwb [512] = dgamma | dbeta; part holds train_ln_bwd_parts(rows) * 512
int train_ln_bwd(const float* dy, const float* s_in, const float* stats, const float* w, float* ds, float* da, float* part,
                 float* dwb, int rows, float p, uint32_t seed, hipStream_t s) {
  if (rows <= 0) return 0;
  int per = (rows + 511) / 512;
  per = (per + 3) / 4 * 4;
  const int nwg = (rows + per - 1) / per;
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(nwg), dim3(256), 0, s, dy, s_in, stats, w, ds, da, part, rows, per, train_thresh(p),
                     p > 0.f ? 1.f / (1.f - p) : 1.f, seed, train_salt_ptr());
  if (hipGetLastError() != hipSuccess) return -2;
  if (dwb == nullptr) return 0;                          // the caller finishes the partials later (train_reduce_jobs)
  return train_sum_parts(part, nwg, (size_t)512, dwb, s);
}

int train_add_rowmod(const float* x, const float* x2, int mod, float* y, int rows, hipStream_t s) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(add_rowmod_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, x2, mod, y, rows);
  return LAUNCH_OK();
}

int train_sum_parts(const float* part, int nparts, size_t numel, float* out, hipStream_t s) {
  if (numel == 0) return 0;
  hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, s, part, nparts, numel, out);
  return LAUNCH_OK();
}

int train_reduce_jobs(const TrainReduceJob* jobs, const TrainReduceSrc* srcs, const unsigned* chunk_job, int njobs, int nchunks,
                      hipStream_t s) {
  if (njobs <= 0 || nchunks <= 0) return 0;
  hipLaunchKernelGGL(reduce_jobs_kernel, dim3((unsigned)nchunks), dim3(256), 0, s, jobs, srcs, chunk_job);
  return LAUNCH_OK();
}

__global__ __launch_bounds__(256) void perm_jobs_kernel(const TrainPermJob* __restrict__ jobs, const unsigned* __restrict__ tile_job) {
  __shared__ float tile[32][33];
  const TrainPermJob j = jobs[tile_job[blockIdx.x]];
  unsigned t = blockIdx.x - j.tile0;
  const unsigned per_z = j.tiles_r * j.tiles_c;
  const unsigned z = t / per_z;
  t -= z * per_z;
  const unsigned tr = t / j.tiles_c, tc = t - tr * j.tiles_c;
  const unsigned r0 = tr * 32, c0 = tc * 32;
  const unsigned tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* src = j.src + (size_t)((j.pad & 1) ? j.Z - 1 - z : z) * j.sz;   // flag bit 0: the source's batches in reverse (a flipped 3x3 kernel)
  float* dst = j.dst + (size_t)z * j.dz;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned r = r0 + ty + 8 * i, c = c0 + tx;
    float v = 0.f;
    if (r < j.R && c < j.C) {
      v = src[(size_t)r * j.sr + (size_t)c * j.sc];
      if (j.scale != nullptr) v = __fmul_rn(v, j.scale[r]);      // (one rounding, as scale_rows makes)
    }
    tile[ty + 8 * i][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < j.C && r < j.R) dst[(size_t)c * j.dc + r] = tile[tx][ty + 8 * i];
  }
}

int train_perm_jobs(const TrainPermJob* jobs, const unsigned* tile_job, int njobs, int ntiles, hipStream_t s) {
  if (njobs <= 0 || ntiles <= 0) return 0;
  hipLaunchKernelGGL(perm_jobs_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, jobs, tile_job);
  return LAUNCH_OK();
}

int train_adam(const TrainAdamJob* jobs, const unsigned* chunk_job, int nchunks, const float* g, float* m, float* v, const float* lr,
               int ngroups, double b1, double b2, double eps, double bc1, double bc2_sqrt, const float* step_ptr, hipStream_t s) {
  if (nchunks <= 0) return 0;
  if (ngroups < 1 || ngroups > 8 || lr == nullptr) return -1;
  AdamArgs a;
  for (int i = 0; i < 8; ++i) a.lr[i] = i < ngroups ? lr[i] : 0.f;
  a.b1 = (float)b1;
  a.b2 = (float)b2;
  a.eps = (float)eps;
  a.omb1 = (float)(1.0 - b1);
  a.omb2 = (float)(1.0 - b2);
  a.bc1 = (float)bc1;
  a.bc2_sqrt = (float)bc2_sqrt;
  a.step_ptr = step_ptr;
  hipLaunchKernelGGL(adam_jobs_kernel, dim3((unsigned)nchunks), dim3(256), 0, s, jobs, chunk_job, g, m, v, a);
  return LAUNCH_OK();
}

int train_dropout_fwd(float* x, size_t n, float p, uint32_t seed, hipStream_t s) {
  if (n == 0 || p <= 0.f) return 0;
  if (n % 4) return -1;
  hipLaunchKernelGGL(dropout_fwd_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, x, n / 4, train_thresh(p),
                     1.f / (1.f - p), seed, train_salt_ptr());
  return LAUNCH_OK();
}

int train_relu_drop_bwd(const float* dy, const float* y, float* dx, size_t n, float p, hipStream_t s) {
  if (n == 0) return 0;
  if (n % 4) return -1;
  hipLaunchKernelGGL(relu_drop_bwd_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, dy, y, dx, n / 4,
                     p > 0.f ? 1.f / (1.f - p) : 1.f);
  return LAUNCH_OK();
}

int train_colsum_parts(int M) {
  const int per = (M + 255) / 256;
  return (M + per - 1) / per;
}

int train_colsum(const float* x, float* part, float* out, int M, int N, hipStream_t s) {
  if (M <= 0 || N <= 0) return 0;
  if (N % 4 || N > 4096) return -1;
  const int per = (M + 255) / 256;
  const int nwg = (M + per - 1) / per;
  hipLaunchKernelGGL(colsum_kernel, dim3(nwg), dim3(256), 0, s, x, part, M, N, per);
  if (hipGetLastError() != hipSuccess) return -2;
  return train_sum_parts(part, nwg, (size_t)N, out, s);
}

int train_transpose(const float* src, float* dst, int R, int C, hipStream_t s) {
  if (R <= 0 || C <= 0) return 0;
  hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, s, src, dst, R, C);
  return LAUNCH_OK();
}

static ConvGeo conv_geo(int B, int Hin, int Win, int Cin, int ksize, int stride) {
  ConvGeo g;
  g.B = B; g.Hin = Hin; g.Win = Win; g.Cin = Cin; g.ksize = ksize; g.stride = stride; g.pad = ksize / 2;
  g.Hout = (Hin + 2 * g.pad - ksize) / stride + 1;
  g.Wout = (Win + 2 * g.pad - ksize) / stride + 1;
  return g;
}

// col [B*Hout*2*Wout][k*k*Cin] of x [B][Hin][2*Win][Cin]
int train_im2col(const float* x, float* col, int B, int Hin, int Win, int Cin, int ksize, int stride, hipStream_t s) {
  if (Cin % 4 || B <= 0) return -1;
  const ConvGeo g = conv_geo(B, Hin, Win, Cin, ksize, stride);
  const size_t total4 = (size_t)B * g.Hout * 2 * g.Wout * ksize * ksize * (Cin / 4);
  hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, x, col, g, total4);
  return LAUNCH_OK();
}

// dx [B][Hin][2*Win][Cin] = gather of dcol [B*Hout*2*Wout][k*k*Cin]
int train_col2im(const float* dcol, float* dx, int B, int Hin, int Win, int Cin, int ksize, int stride, hipStream_t s) {
  if (Cin % 4 || B <= 0) return -1;
  const ConvGeo g = conv_geo(B, Hin, Win, Cin, ksize, stride);
  const size_t total4 = (size_t)B * Hin * 2 * Win * (Cin / 4);
  hipLaunchKernelGGL(col2im_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, dcol, dx, g, total4);
  return LAUNCH_OK();
}

int train_scale_rows(const float* w, const float* scale, float* out, int rows, int cols, hipStream_t s) {
  if (cols % 4 || rows <= 0) return -1;
  const size_t total = (size_t)rows * (cols / 4);
  hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, scale, out, rows, cols / 4);
  return LAUNCH_OK();
}

int train_transpose_batched(const float* src, float* dst, int batch, int R, int C, hipStream_t s) {
  if (batch <= 0 || R <= 0 || C <= 0) return 0;
  if (batch > 65535) return -1;
  hipLaunchKernelGGL(transpose_batched_kernel, dim3((C + 31) / 32, (R + 31) / 32, batch), dim3(256), 0, s, src, dst, R, C);
  return LAUNCH_OK();
}

static bool gemm_tn_use_big(int M, int N, int K) {
  // (256 x 256 outputs measured the same on either kernel: four 128 x 128 tiles need 64 splits to fill the chip)
  return N % 128 == 0 && K % 128 == 0 && (size_t)N * K >= (size_t)512 * 256 && M >= 2048;
}

int train_gemm_tn_splits(int M, int N, int K) {
  if (gemm_tn_use_big(M, N, K)) {
    const int tiles = (N / 128) * (K / 128);
    int splits = (512 + tiles - 1) / tiles;               // two workgroups per CU
    const int max_splits = (M + 255) / 256;               // at least 256 rows (8 chunks) per split
    if (splits > max_splits) splits = max_splits;
    return splits < 1 ? 1 : splits;
  }
  const int tiles = (N / 64) * (K / 64);
  int splits = (512 + tiles - 1) / tiles;                 // about two workgroups per CU: more splits = more partial traffic
  const int max_splits = (M + 127) / 128;                 // at least 128 rows per split
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  return splits;
}

// out[N][K] = A[M][N]^T . B[M][K] (and, with colsum != nullptr, colsum[N] = column sums of A: dW and db of a Linear from ONE
// pass over dY); part holds train_gemm_tn_splits(M, N, K) * (N * K + N) floats; out and colsum must be ONE contiguous
// [N*K + N] buffer when both are wanted (colsum == out + N*K): the partials are then finished by a single launch
int train_gemm_tn(const float* A, const float* B, float* part, float* out, float* colsum, int M, int N, int K, hipStream_t s) {
  if (N % 64 || K % 64 || M < 0) return -1;
  if (colsum != nullptr && colsum != out + (size_t)N * K) return -1;
  if (M == 0) {   // an empty batch (no queries) contributes zero gradients: dW = 0, db = 0, like the other wrappers' no-ops
    return hipMemsetAsync(out, 0, ((size_t)N * K + (colsum != nullptr ? N : 0)) * sizeof(float), s) == hipSuccess ? 0 : -2;
  }
  const int nsplit = train_gemm_tn_parts(A, B, part, M, N, K, colsum != nullptr ? 1 : 0, s);
  if (nsplit < 0) return nsplit;
  return train_sum_parts(part, nsplit, (size_t)N * K + (colsum ? N : 0), out, s);
}

// The partials alone: part[split][N*K (+ N column sums with with_colsum)]; -> number of splits written (0 for M == 0: nothing
// to add), < 0 on error.  The caller sums them in split order (train_sum_parts, or later through train_reduce_jobs).
int train_gemm_tn_parts(const float* A, const float* B, float* part, int M, int N, int K, int with_colsum, hipStream_t s) {
  if (N % 64 || K % 64 || M < 0) return -1;
  if (M == 0) return 0;
  const int splits = train_gemm_tn_splits(M, N, K);
  int per = (M + splits - 1) / splits;
  per = (per + 31) / 32 * 32;
  const int nsplit = (M + per - 1) / per;
  if (gemm_tn_use_big(M, N, K)) {
    const float* zeros = gemm_zero_buffer();
    if (zeros == nullptr) return -2;
    static PerDeviceFlag attr_set;
    constexpr size_t smem = 2 * 2 * 32 * 128 * sizeof(float);
    if (!attr_set.get()) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_big_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem) != hipSuccess)
        return -2;
      attr_set.set();
    }
    hipLaunchKernelGGL(gemm_tn_big_kernel<false>, dim3((N / 128) * (K / 128), nsplit), dim3(256), smem, s, A, B, part, M, N, K, per,
                       with_colsum ? 1 : 0, zeros, ConvGeo{});
  } else
    hipLaunchKernelGGL(gemm_tn_kernel, dim3((N / 64) * (K / 64), nsplit), dim3(256), 0, s, A, B, part, M, N, K, per,
                       with_colsum ? 1 : 0);
  if (hipGetLastError() != hipSuccess) return -2;
  return nsplit;
}

// The weight gradient of a convolution WITHOUT its im2col image: part[split][Cout][k*k*Cin] = dz[rows of the split]^T . im2col(x)[same
// rows], the image gathered by the kernel (gemm_tn_big_kernel<true>).  -> number of splits (the same as train_gemm_tn_splits(M, Cout,
// k*k*Cin): the caller sizes `part` with it), or -1 where this form does not apply (Cin % 128, the small-shape kernel's territory):
// the caller then forms the image (train_im2col) and calls train_gemm_tn_parts - the two give the same bits.
int train_conv_wgrad_parts(const float* dz, const float* x, float* part, int B, int Hin, int Win, int Cin, int Cout, int ksize, int stride,
                           hipStream_t s) {
  if (B <= 0 || Cin % 128 != 0 || ksize < 1 || stride < 1) return -1;
  const ConvGeo g = conv_geo(B, Hin, Win, Cin, ksize, stride);
  const int M = B * g.Hout * 2 * g.Wout, N = Cout, K = ksize * ksize * Cin;
  if (!gemm_tn_use_big(M, N, K)) return -1;
  const int splits = train_gemm_tn_splits(M, N, K);
  int per = (M + splits - 1) / splits;
  per = (per + 31) / 32 * 32;
  const int nsplit = (M + per - 1) / per;
  const float* zeros = gemm_zero_buffer();
  if (zeros == nullptr) return -2;
  static PerDeviceFlag attr_set;
  constexpr size_t smem = 2 * 2 * 32 * 128 * sizeof(float);
  if (!attr_set.get()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_big_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem) != hipSuccess)
      return -2;
    attr_set.set();
  }
  hipLaunchKernelGGL(gemm_tn_big_kernel<true>, dim3((N / 128) * (K / 128), nsplit), dim3(256), smem, s, dz, x, part, M, N, K, per, 0, zeros, g);
  if (hipGetLastError() != hipSuccess) return -2;
  return nsplit;
}

int train_head_bwd_parts(int rows) {
  int per = (rows + 255) / 256;
  per = (per + 3) / 4 * 4;
  return (rows + per - 1) / per;
}

// dh [rows][256]; dw2 [2][256] and db2 [2] contiguous in `dwb` (514 floats)
int train_head_bwd(const float* dy, const float* h, const float* w2, float* dh, float* part, float* dwb, int rows, hipStream_t s) {
  if (rows <= 0) return 0;
  int per = (rows + 255) / 256;
  per = (per + 3) / 4 * 4;
  const int nwg = (rows + per - 1) / per;
  hipLaunchKernelGGL(head_bwd_kernel, dim3(nwg), dim3(256), 0, s, dy, h, w2, dh, part, rows, per);
  if (hipGetLastError() != hipSuccess) return -2;
  if (dwb == nullptr) return 0;                          // deferred (train_reduce_jobs)
  return train_sum_parts(part, nwg, (size_t)514, dwb, s);
}
