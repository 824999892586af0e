// Training-step kernels (train.hip, attention_train.hip): launch helpers and the counter-based dropout mask shared by the
// forward and backward kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// keep(element) = hash(seed, index) >= p * 2^32: a 64 -> 32 bit integer mixer (two multiply / xor-shift rounds over both
// words of the index); the same (seed, index) gives the same decision in the forward and in the backward kernel.
__device__ __forceinline__ bool train_keep(uint32_t seed, uint64_t idx, uint32_t thresh) {
  uint32_t x = (uint32_t)idx ^ (seed * 0x9E3779B9u);
  const uint32_t hi = (uint32_t)(idx >> 32) ^ seed;
  x ^= hi * 0x85EBCA6Bu + 0xC2B2AE35u;
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x >= thresh;
}
// Step salt for captured (hipGraph) training steps: a launch's seed is baked into the graph, so the kernels XOR it with a word
// read from device memory that the captured step itself advances (cotr_train_set_dropout_salt; nullptr = plain seeds)
const uint32_t* train_salt_ptr();
bool train_clear_salt_ptr_if(const uint32_t* expected);
void train_set_salt_ptr(const uint32_t* p);
__device__ __forceinline__ uint32_t train_salted(uint32_t seed, const uint32_t* salt) { return salt ? seed ^ *salt : seed; }
static inline uint32_t train_thresh(float p) {
  if (!(p > 0.f)) return 0u;
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
}

int train_add_drop_ln_fwd(const float* x, const float* a, const float* w, const float* b, float* s_out, float* y, float* stats,
                          int rows, float p, uint32_t seed, hipStream_t s);
int train_ln_bwd_parts(int rows);
int train_ln_bwd(const float* dy, const float* s_in, const float* stats, const float* w, float* ds, float* da, float* part,
                 float* dwb, int rows, float p, uint32_t seed, hipStream_t s);
int train_add_rowmod(const float* x, const float* x2, int mod, float* y, int rows, hipStream_t s);
int train_sum_parts(const float* part, int nparts, size_t numel, float* out, hipStream_t s);
// deferred gradient reduction (train.hip: reduce_jobs_kernel); the layouts are part of the C ABI (include/cotr_hip.h)
struct TrainReduceSrc {
  const float* part;             // first partial (already offset to this destination's part of a partial record)
  unsigned long long pstride;    // floats from one partial to the next
  unsigned nparts, pad_;
};
struct TrainReduceJob {
  float* dst;                    // accumulated into: dst[perm(e)] += scale[row] * sum of the partials, source after source
  const float* scale;            // per-row factor (FrozenBN scale of a conv weight gradient) or nullptr
  unsigned numel, first_src, n_src, chunk0;   // chunk0 = index of this job's first 1024-element chunk in the launch
  unsigned row_len, cin, taps;   // row_len 0: flat; taps > 1: packed [row][tap][cin] -> dst [row][cin][tap]
  unsigned vec;                  // 1: every pointer 16-byte aligned and numel / pstride / row_len / cin multiples of 4
};
int train_reduce_jobs(const TrainReduceJob* jobs, const TrainReduceSrc* srcs, const unsigned* chunk_job, int njobs, int nchunks,
                      hipStream_t s);
// One launch that re-derives EVERY weight-shaped operand a training step needs from the parameters (after the optimiser step): the W^T
// slices of the dX GEMMs, the packed [Cout][k][k][Cin] convolution weights, the FrozenBN-scaled transposed convolution weights.  A job is
// a batched strided transpose with an optional per-source-row factor: dst[z * dz + c * dc + r] = src[z * sz + r * sr + c * sc] * scale[r]
struct TrainPermJob {
  const float* src;
  float* dst;
  const float* scale;          // [R] or nullptr
  unsigned Z, R, C;            // batches, rows, columns of the source view
  unsigned sz, sr, sc, dz, dc; // element strides (the destination's row stride is 1)
  unsigned tile0, tiles_r, tiles_c, pad;   // pad: flags - bit 0: source batch Z-1-z feeds destination batch z
};
int train_perm_jobs(const TrainPermJob* jobs, const unsigned* tile_job, int njobs, int ntiles, hipStream_t s);
struct TrainAdamJob {
  float* p;                      // the parameter
  unsigned long long off;        // its offset (floats) in the flat gradient / m / v buffers
  unsigned numel, chunk0, group, vec;   // vec 1: p is 16-byte aligned
};
// lr: HOST array of ngroups (<= 8) learning rates, passed by value
int train_adam(const TrainAdamJob* jobs, const unsigned* chunk_job, int nchunks, const float* g, float* m, float* v, const float* lr,
               int ngroups, double b1, double b2, double eps, double bc1, double bc2_sqrt, const float* step_ptr, hipStream_t s);
int train_gemm_tn_parts(const float* A, const float* B, float* part, int M, int N, int K, int with_colsum, hipStream_t s);
int train_dropout_fwd(float* x, size_t n, float p, uint32_t seed, hipStream_t s);
int train_relu_drop_bwd(const float* dy, const float* y, float* dx, size_t n, float p, hipStream_t s);
int train_colsum_parts(int M);
int train_colsum(const float* x, float* part, float* out, int M, int N, hipStream_t s);
int train_transpose(const float* src, float* dst, int R, int C, hipStream_t s);
int train_im2col(const float* x, float* col, int B, int Hin, int Win, int Cin, int ksize, int stride, hipStream_t s);
// weight-gradient partials of a convolution straight from x (no im2col image); -1 where that form does not apply (train.hip)
int train_conv_wgrad_parts(const float* dz, const float* x, float* part, int B, int Hin, int Win, int Cin, int Cout, int ksize, int stride,
                           hipStream_t s);
int train_col2im(const float* dcol, float* dx, int B, int Hin, int Win, int Cin, int ksize, int stride, hipStream_t s);
int train_scale_rows(const float* w, const float* scale, float* out, int rows, int cols, hipStream_t s);
int train_transpose_batched(const float* src, float* dst, int batch, int R, int C, hipStream_t s);
int train_gemm_tn_splits(int M, int N, int K);
int train_gemm_tn(const float* A, const float* B, float* part, float* out, float* colsum, int M, int N, int K, hipStream_t s);
int train_head_bwd_parts(int rows);
int train_head_bwd(const float* dy, const float* h, const float* w2, float* dh, float* part, float* dwb, int rows, hipStream_t s);

// attention_train.hip: softmax(q k^T * qscale) with dropout on the probabilities; q [nb*nq][ldq], k [nb*512][ldk], v [nb*512][ldv]
// (8 heads x 32); lse [nb*nq][8] = log2-domain log-sum-exp of the scaled scores (for the backward)
int train_attention_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo, float* lse,
                        int nb, int nq, float qscale, float p, uint32_t seed, hipStream_t s);
// dq [nb*nq][lddq], dk [nb*512][lddk], dv [nb*512][lddv]; delta [nb*nq][8] scratch (rowsum(dO * O) per head); o / d_o share ldo
int train_attention_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o, const float* d_o,
                        int ldo, const float* lse, float* delta, float* dq, int lddq, float* dk, int lddk, float* dv, int lddv,
                        int nb, int nq, float qscale, float p, uint32_t seed, float* scratch, hipStream_t s);
// floats of `scratch` train_attention_bwd can use (dQ partials of the key-split one-pass backward); scratch == nullptr is allowed
static inline size_t train_attention_bwd_scratch(int nb, int nq) { return (size_t)4 * nb * nq * 256; }
