// Declarations of the MEASURED DEAD ENDS (docs/LABNOTES.md 4b / 3c): compiled only into libcotr_hip_exp.so (-DCOTR_EXPERIMENTAL), never into
// the product library.  Each is correct and tested (tests/test_experimental_gpu.py runs against the experimental library); each lost
// its A/B on the MI355X and is kept so that the measurement can be repeated.
//   coop_tail.h      row tiles finished by their own workgroups instead of ln_reduce launches      +7.4 us per tail
//   head.hip         decoder.norm + corr_embed as one row-local launch                               1.004 vs 0.998 ms
//   gemm_ln.hip      256-wide projection + LayerNorm as one 128 x 256 tile launch                    nothing inside the forward
//   ffn tail / preln last-arriver reduce inside the FFN launch; norm1 folded into the FFN block      1.285 vs 1.043 ms; neutral
//   GEMM configs 28 / 29  three LDS stages in the large-tile kernel                                  within +-5 %
//   gemm_pp.hip      persistent ping-pong large tiles (configs 42 / 43): loaders run ahead across tiles,   383 vs 306 us (K = 256):
//                    a tile's write-out beside the next tile's MFMAs                                     the write-out is not hidden
//   gemm_h2r.hip     split-f16 GEMM for K = 256 with the A tile resident in registers (configuration 50): 138 vs 74 us at
//                    32000 x 1024 x 256 - one wavefront per SIMD, nothing covers its LDS reads / perms
//   GEMM config 51   split-f16 on a 256 x 128 tile (8 wavefronts, 3 stages): 355 vs 340 TFLOP/s at 4096^3, equal or slower on the model's shapes
//   gemm_h2.h/.hip   RESEARCH, not a dead end: fp32 products from three f16 MFMAs on packed split-f16 operands (configs 46 / 47);
//                    not bit-identical to the fp32 path and range-limited (|x| < 65504) - see the header of gemm_h2.h
#pragma once
#include "coop_tail.h"

int launch_attention_fused_coop(const float* q, int ldq, const float* x, const float* x2, const float* wq, const float* bq,
                                float qscale, const float* k, const float* v, int ldkv, float* o, int ldo, const float* wo,
                                float* part, int nb, int nq, hipStream_t s, const CoopTail* ct);
int launch_ffn_fused_coop(const float* X, const float* W1, const float* b1, const float* W2, float* P, int M, int nch,
                          const CoopTail& ct, hipStream_t s);
// fused FFN + the reduce / bias / residual / LayerNorm tail inside the kernel (last-arriving workgroup of a row tile)
int launch_ffn_fused_ln(const float* X, const float* W1, const float* b1, const float* W2, float* P, int M, int nch,
                        const float* b2, const float* residual, const float* ln_w, const float* ln_b, float* Y, hipStream_t s);
// fused FFN / ln_reduce with the LayerNorm that precedes the FFN folded in: X / residual are the PRE-norm rows
int launch_ffn_fused_pre(const float* X, const float* pre_w, const float* pre_b, const float* W1, const float* b1,
                         const float* W2, float* P, int M, int nch, hipStream_t s);
int launch_ln_reduce_pre(const float* parts, int np, const float* bias, const float* residual, const float* pre_w,
                         const float* pre_b, const float* w, const float* b, float* y, int rows, hipStream_t s);
// gemm_ln.hip: y [M][256] = LayerNorm(x [M][K] . w [256][K]^T + bias + residual) * ln_w + ln_b, a workgroup owns 128 complete rows
int launch_gemm_ln(const float* x, int lda, const float* w, const float* bias, const float* residual, int ldr, const float* ln_w,
                   const float* ln_b, float* y, int M, int K, hipStream_t s);
// head.hip: decoder.norm + corr_embed (256 -> 256 -> 256 -> 2) in one row-local launch; hs (normalised rows) optional
int launch_dec_head(const float* x, const float* nw, const float* nb, const float* w0, const float* b0, const float* w1,
                    const float* b1, const float* w2, const float* b2, float* hs, float* out, int nb_pairs, int nq, int q_total,
                    hipStream_t s);
// pointwise.hip: the next ln_reduce launch also pulls two regions (the next launch's weights) through every XCD's L2 (knob l2_warm)
void set_ln_reduce_warm(const float* p0, size_t bytes0, const float* p1, size_t bytes1);
// gemm_pp.hip: persistent ping-pong large tiles (1 = 128 x 64 staged write-out, 3 = 128 x 64 LDS-free write-out)
int launch_gemm_pp(int mode, int variant, const GemmParams& p, hipStream_t s);
int gemm_pp_workgroups();
// gemm_h2.hip / gemm_h2.h (research): fp32 -> packed split-f16 dwords, the operand format of GEMM configurations 46 / 47
int launch_split_h2(const float* x, void* y, size_t n, hipStream_t s, const float* x2 = nullptr);   // y = pack(x [+ x2])
int launch_unsplit_h2(const void* x, float* y, size_t n, hipStream_t s);
int* h2_overflow_flag();   // per-device flag raised by every kernel that packs an activation outside f16's range (round 5: range safety)
// y = LayerNorm(x) (bits of launch_layernorm) and yp = pack(y [+ add]) in one launch
// gemm_h2r.hip: K = 256 dense GEMM on packed operands, A tile resident in registers (configuration 50)
int launch_gemm_h2r(const GemmParams& p, hipStream_t s);
int launch_layernorm_h2(const float* x, const float* w, const float* b, float* y, void* yp, const float* add, int rows, hipStream_t s);
// attention_h2.hip: the resident-K/V attention kernel on packed k / v (q fp32 or packed, o fp32 or packed)
int launch_attention_h2(const float* q, int ldq, int q_packed, const float* k, const float* v, int ldkv, float* o, int ldo, int out_packed,
                        int nb, int nq, hipStream_t s);
//   // persistent workgroups per launch on the current device
