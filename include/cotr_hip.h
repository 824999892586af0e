/*
 * cotr_hip.h - C ABI of libcotr_hip.so: the MI355X (gfx950) implementation of COTR's
 * batched correspondence-query forward path.
 *
 * The reference (ubc-vision/COTR) has no FFI: its seam is the Python call
 *     COTR.forward(samples, queries) -> {'pred_corrs'}      (COTR/models/cotr_model.py:26-40)
 * made by SparseEngine.infer_batch (COTR/inference/sparse_engine.py:47-56),
 * FasterSparseEngine.infer_batch_grouped (:277-282), cotr_patch_flow_exhaustive
 * (COTR/inference/inference_helper.py:106-145) and cotr_corr_base (:186-204).
 * This header is what a binding for that seam binds to; cotr_amd/models (ctypes) is
 * the binding this repository ships, INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *  - plain C, no exceptions; every call returns COTR_OK (0) or a negative COTR_ERR_* and
 *    leaves a message retrievable with cotr_last_error().
 *  - all tensors are fp32, contiguous; `img`, `queries`, `out` are DEVICE pointers owned by
 *    the caller (torch tensors' data_ptr()); weight pointers may be host or device.
 *  - `stream` is a hipStream_t (NULL = the default stream).  All work is enqueued on it;
 *    nothing synchronises the device except cotr_load_weights / cotr_destroy / cotr_debug_tap.
 *  - packed weights and scratch live in memory owned by the handle (grow-only arenas).
 *  - one handle per device per caller thread; a handle is not thread-safe.  Handles on different
 *    devices may coexist in one process: every call makes its handle's device current for its
 *    duration and restores the caller's current device before returning; one-time kernel
 *    attributes and helper buffers are kept per device.
 *  - a NaN in the inputs/weights yields NaN outputs, never a trap: the reference's engines
 *    raise ValueError('NaN in prediction') themselves (sparse_engine.py:54-55).
 *
 * Model geometry is COTR's default and only published configuration
 * (COTR/options/options.py:41-51): resnet50 to layer3, hidden 256, 8 heads, FFN 1024,
 * lin_sine; the number of encoder/decoder layers is read from the weight names.
 */
#ifndef COTR_HIP_H
#define COTR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden and linked with a version script made from this header (cotr_amd/build.py): the
 * functions declared here are its ONLY dynamic symbols. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define COTR_HIP_ABI_VERSION 2   /* 2: tuning knobs per handle (cotr_set_knob(h, ...)), the cotr_set_<knob>(int) functions are gone;
                                  cotr_train_attention_bwd takes a scratch argument; experiments live in libcotr_hip_exp.so */

#define COTR_OK 0
#define COTR_ERR_ARG (-1)    /* bad argument (null pointer, bad shape)                     */
#define COTR_ERR_HIP (-2)    /* a HIP runtime call failed; see cotr_last_error()            */
#define COTR_ERR_STATE (-3)  /* weights not loaded / decode without a matching encode       */
#define COTR_ERR_WEIGHTS (-4) /* a required tensor is missing or has the wrong element count */

typedef struct cotr_ctx* cotr_handle;
typedef void* cotr_stream; /* hipStream_t */

int cotr_abi_version(void);

/* Per-device context.  Replaces: model = build_model(opt).cuda()  (demo_single_pair.py:26-27) */
int cotr_create(cotr_handle* out, int device);
void cotr_destroy(cotr_handle h);
const char* cotr_last_error(cotr_handle h); /* h may be NULL: last error of cotr_create */

/* Load a state-dict: n tensors, names as in the reference's checkpoints
 * ('backbone.0.body.layer1.0.conv1.weight', 'transformer.encoder.layers.0.self_attn.in_proj_weight',
 * ...; full table: cotr_amd/models/spec.py).  Re-lays-out conv weights to [Cout][kh][kw][Cin],
 * folds the four FrozenBN buffers into (scale, bias) exactly as COTR/models/backbone.py:46-56
 * computes them, concatenates the decoder K/V projections.  Unknown names are ignored
 * (e.g. decoder norm1, never applied: COTR/models/transformer.py:173,185-201).
 * Replaces: utils.safe_load_weights(model, weights)  (COTR/utils/utils.py:164-193). */
int cotr_load_weights(cotr_handle h, const char* const* names, const float* const* ptrs,
                      const int64_t* numels, int n);

/* Query-independent half: backbone on both 256x256 halves, input_proj, 6 encoder layers and the
 * K/V projections of every decoder layer; result cached in the handle for cotr_decode.
 * img: [B,3,256,512] NCHW, ImageNet-normalised side-by-side pair.
 * Follows COTR/models/backbone.py:79-92,114-123, cotr_model.py:37, transformer.py:49-55,143-159,192-195. */
int cotr_encode(cotr_handle h, const float* img, int B, cotr_stream stream);

/* Backbone only: features [B,16,32,1024] (NHWC over the side-by-side pair = 512 token rows of 1024 channels per pair,
 * token h*32+w like flatten(2) of the reference's [B,1024,16,32]).  Replaces self.backbone(samples)[0][-1].tensors
 * (COTR/models/backbone.py:79-92) for callers that run the rest themselves: the training step with a frozen backbone
 * (train_cotr.py:54-55 with --lr_backbone=0, the reference's stage 1). */
int cotr_backbone(cotr_handle h, const float* img, int B, float* features, cotr_stream stream);
/* Same, stopping after `stage` = 1, 2 or 3 (layer1 [B,64,128,256], layer2 [B,32,64,512], layer3 [B,16,32,1024], NHWC over
 * the pair): the part of the backbone that stays frozen when the reference trains layer2/layer3 (--lr_backbone > 0 trains
 * only parameters whose name contains layer2/3/4, COTR/models/backbone.py:66-69; stages 2-3 of readme.md:50-52). */
int cotr_backbone_upto(cotr_handle h, const float* img, int B, int stage, float* features, cotr_stream stream);

/* Query-dependent half against the cached encode: lin_sine query encoding, 6 cross-attention
 * decoder layers, decoder.norm and the corr_embed MLP on the LAST layer only (the reference runs
 * them on all 6 and keeps [-1]: transformer.py:107-117, cotr_model.py:38-39).
 * queries, out: [B,Q,2]; may be called repeatedly per encode (cycle pass of
 * inference_helper.py:197-198, query chunks of :131-136).  Q == 0 is a no-op. */
int cotr_decode(cotr_handle h, const float* queries, int B, int Q, float* out, cotr_stream stream);

/* cotr_encode + cotr_decode: the drop-in for COTR.forward (cotr_model.py:26-40). */
int cotr_forward(cotr_handle h, const float* img, const float* queries, int B, int Q, float* out,
                 cotr_stream stream);

/* Bytes of device memory behind a call of that size (packed weights + encode cache + scratch). */
int cotr_workspace_bytes(cotr_handle h, int B, int Q, size_t* bytes);

/* Scratch from the CALLER's allocator ("torch allocates, passes in").  By default the encode cache and the two
 * scratch arenas are grow-only hipMalloc'ed memory owned by the handle; growing one (a larger batch arrives) costs a
 * hipFree + hipMalloc, i.e. a device synchronisation in the middle of the caller's stream.  With a workspace the handle
 * carves the three regions from [ws, ws + bytes) instead (256-byte aligned `ws`, device memory, e.g. a torch uint8 tensor
 * from the caching allocator) and never allocates or frees.  cotr_scratch_bytes(B, Q) is the size a call of that shape needs;
 * a call that does not fit returns COTR_ERR_ARG.  keep_encode != 0: a cached encode is carried over into the new workspace
 * (device copy on `stream`, so the old workspace must outlive the work enqueued so far; meant for a decode with more queries
 * than the workspace was sized for - same number of pairs); otherwise, and with ws == NULL (back to handle-owned memory), it
 * is dropped.  The packed weights (74 MB) and the 0.5 MB position table stay handle-owned. */
int cotr_scratch_bytes(cotr_handle h, int B, int Q, size_t* bytes);
int cotr_set_workspace(cotr_handle h, void* ws, size_t bytes, int keep_encode, cotr_stream stream);

/* How a (B, Q) call is cut into passes under the handle's knobs (knob batch_split): which = 0 the encode passes, 1 the decode passes;
 * sizes[0 .. min(return, cap)) = pairs per pass; returns the number of passes or < 0.  Pairs are independent, so a call computes each
 * pass exactly as a call on those pairs alone would (tests/test_parity_gpu.py). */
int cotr_batch_chunks(cotr_handle h, int B, int Q, int which, int* sizes, int cap);

/* ---- test / profiling hooks (not needed by a binding) ------------------------------------ */

/* Keep copies of scratch intermediates for cotr_debug_tap (off by default: costs D2D copies). */
int cotr_set_debug_taps(cotr_handle h, int enable);

/* Copy an intermediate of the LAST cotr_encode / cotr_decode to dst (device or host pointer).
 * always available: "memory" ([B*512,256]), "kv" ([B*512, L*512]), "pos" ([512,256]);
 * with debug taps on: "stem" "pool" "layer1" "layer2" "layer3" (NHWC side-by-side [B,H,2W,C] of the
 * last encode chunk), "src" ([B*512,256]), "query_pos" "hs" ([rows,256] of the last decode chunk).
 * Synchronises the stream. */
int cotr_debug_tap(cotr_handle h, const char* name, float* dst, size_t max_elems, size_t* n_elems,
                   cotr_stream stream);

/* HIP-event timings (ms) of the last cotr_encode + cotr_decode: enable = 1 per stage, 2 per kernel launch
 * (names carry the GEMM shape and launch configuration), 0 off. */
int cotr_set_profiling(cotr_handle h, int enable);
int cotr_get_profile(cotr_handle h, const char** names, float* ms, int max_entries, int* n_entries);

/* Single-kernel entry points used by tests/test_ops_gpu.py for op-level parity.
 * All pointers are device pointers. */
/* y[M,N] = epilogue(x[M,K] (+ x2[M % x2_row_mod, K]) . w[N,K]^T) ; any of scale/bias/residual may be NULL */
int cotr_op_linear(const float* x, const float* x2, int x2_row_mod, const float* w, const float* scale,
                   const float* bias, const float* residual, int relu, float* y, int M, int N, int K,
                   cotr_stream stream);
/* NHWC side-by-side conv + FrozenBN(scale,bias) (+residual) (+ReLU):
 * x [B,Hin,2*Win,Cin], w [Cout][k][k][Cin], y [B,Hout,2*Wout,Cout] */
int cotr_op_conv(const float* x, const float* w, const float* scale, const float* bias,
                 const float* residual, int relu, float* y, int B, int Hin, int Win, int Cin, int Cout,
                 int ksize, int stride, cotr_stream stream);
/* stem: img NCHW [B,3,256,512] -> y [B,128,256,64]; w [64][160] (k = c*49+ky*7+kx, zero padded) */
int cotr_op_stem(const float* img, const float* w, const float* scale, const float* bias, float* y, int B,
                 cotr_stream stream);
int cotr_op_maxpool(const float* x, float* y, int B, int Hin, int Win, int C, cotr_stream stream);
/* conv1 + FrozenBN + ReLU + maxpool fused (stem_pool.hip): img [B,3,256,512] NCHW, w [64][160] -> y [B,64,128,64] NHWC sbs */
int cotr_op_stem_pool(const float* img, const float* w, const float* scale, const float* bias, float* y, int B,
                      cotr_stream stream);
/* q [nb*nq, ldq] (pre-scaled), k/v [nb*512, ldkv]; 8 heads x 32; o [nb*nq, ldo] */
int cotr_op_attention(const float* q, int ldq, const float* k, const float* v, int ldkv, float* o, int ldo,
                      int nb, int nq, cotr_stream stream);
/* the same kernel with the q projection as prologue (wq != NULL: q = ((x + x2) . wq_h^T + bq_h) * qscale per head, `q`
 * unused; x or x2 may be NULL) and / or the output projection as epilogue (wo != NULL: per-head partial outputs
 * part[8][nb*nq][256], part[h] = O_h . wo[:, 32h:32h+32]^T; `o` may then be NULL) */
int cotr_op_attention_fused(const float* q, int ldq, const float* x, const float* x2, const float* wq, const float* bq,
                            float qscale, const float* k, const float* v, int ldkv, float* o, int ldo, const float* wo,
                            float* part, int nb, int nq, cotr_stream stream);
/* y = LayerNorm(sum_c parts[c] + bias + residual) over rows of 256; parts [np][rows][256]; residual may be NULL */
int cotr_op_ln_reduce(const float* parts, int np, const float* bias, const float* residual, const float* w, const float* b,
                      float* y, int rows, cotr_stream stream);
int cotr_op_layernorm(const float* x, const float* w, const float* b, float* y, int rows, cotr_stream stream);
/* fused FFN block y = LN(x + W2 relu(W1 x + b1) + b2) in two launches (ffn.hip + ln_reduce); scratch holds
 * cotr_op_ffn_chunks(M) * M * 256 floats */
int cotr_op_ffn_block(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, const float* ln_w,
                      const float* ln_b, float* scratch, float* y, int M, cotr_stream stream);
int cotr_op_ffn_chunks(int M);
/* the attention sub-layer in ONE launch for many rows (att_rows.hip: transformer.py:149-155 encoder, :192-198 decoder):
 * y = LayerNorm(residual + out_proj(MHA(q, k, v)) + bo), 8 heads of 32, 512 keys per pair; q [nb*nq][ldq] given pre-scaled by
 * 32^-0.5 (wq == NULL), or projected here: q = ((x + x2) . wq^T + bq) * qscale (x may be NULL: decoder layer 0).  residual may be
 * NULL; y must not alias residual / x / x2 / q */
int cotr_op_att_rows(const float* q, int ldq, const float* x, const float* x2, const float* wq, const float* bq, float qscale,
                     const float* k, const float* v, int ldkv, const float* wo, const float* bo, const float* residual,
                     const float* ln_w, const float* ln_b, float* y, int nb, int nq, cotr_stream stream);
/* the same block in ONE launch for many rows (ffn_rows.hip: transformer.py:156-158 / 199-201 [+ :110-111 with post_w / post_b, a
 * second LayerNorm of the result]); y must not alias x */
int cotr_op_ffn_rows(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, const float* ln_w,
                     const float* ln_b, const float* post_w, const float* post_b, float* y, int M, cotr_stream stream);
/* the same for a layer2 bottleneck (conv23m.hip): t1 [B,32 S,64 S,128] NHWC side-by-side (S = stride of the 3x3: 1 or 2), w2 [128][3][3][128],
 * w3 [512][128], residual / y [B,32,64,512] */
int cotr_op_conv23m(const float* t1, const float* w2, const float* s2, const float* b2, const float* w3, const float* s3, const float* b3,
                    const float* residual, float* y, int B, int stride, cotr_stream stream);
/* one or two 1x1 convolutions with K = 64 over the same x in ONE launch for many rows (expand.hip; layer1 block 0's downsample branch
 * and conv1 of torchvision's Bottleneck with COTR/models/backbone.py:46-56): y_s = [relu](FrozenBN_s(x . w_s^T)); x [M][64], M a multiple
 * of 128; w_s [n_s][64], y_s [M][n_s], n_s multiples of 64; n1 == 0: one set.  Shapes it is not written for are declined (-1) */
int cotr_op_expand(const float* x, int M, const float* w0, const float* s0, const float* b0, int relu0, float* y0, int n0, const float* w1,
                   const float* s1, const float* b1, int relu1, float* y1, int n1, cotr_stream stream);
/* conv2 (3x3, padding 1 per half) + FrozenBN + ReLU -> conv3 (1x1) + FrozenBN + identity + ReLU of a layer1 bottleneck in ONE launch
 * (conv23.hip; torchvision Bottleneck.forward with COTR/models/backbone.py:46-56): t1 [B,64,128,64] NHWC side-by-side, w2 [64][3][3][64],
 * w3 [256][64], residual / y [B,64,128,256] */
int cotr_op_conv23(const float* t1, const float* w2, const float* s2, const float* b2, const float* w3, const float* s3, const float* b3,
                   const float* residual, float* y, int B, cotr_stream stream);
/* lin_sine encoding of pts [n,2] -> y [n,256] (COTR/models/position_encoding.py:41-45) */
int cotr_op_posenc(const float* pts, float* y, int n, cotr_stream stream);


/* ---- training step ----------------------------------------------------------
 * The kernels the autograd tape of cotr_amd/training.py runs besides the GEMMs above; they replace, op for op, what
 * COTRTrainer.train_batch (COTR/trainers/cotr_trainer.py:118-150) has torch/cuDNN compute in COTR/models/transformer.py
 * with dropout active.  All device pointers, fp32, rows of 256 channels unless stated.  Dropout is counter-based
 * (keep = hash(seed, element index) >= p * 2^32): backward kernels recompute the mask from (seed, index), nothing is stored;
 * p == 0 makes every kernel exact (what the gradient goldens pin).  Reductions across workgroups go through `part` scratch and
 * are summed in a fixed order (no atomics: a step is bit-repeatable).  cotr_train_*_parts / _splits give the scratch sizes. */
/* y[m] = x[m] + x2[m % mod] (mod == 0: row m): src + pos (transformer.py:147), tgt + query_pos (:192) */
int cotr_train_add_rowmod(const float* x, const float* x2, int mod, float* y, int rows, cotr_stream stream);
/* y = LayerNorm(s), s = x + dropout(a) (x may be NULL); s_out (may be NULL) and stats [rows][2] = (mean, rstd) for the backward:
 * transformer.py:154-155,157-158 (encoder), :196-198,200-201 (decoder), :110-111 (decoder.norm with x NULL, p 0) */
int cotr_train_add_drop_ln_fwd(const float* x, const float* a, const float* w, const float* b, float* s_out, float* y, float* stats,
                               int rows, float p, uint32_t seed, cotr_stream stream);
int cotr_train_ln_bwd_parts(int rows);
/* ds = d loss / d s (= dx), da = ds * mask / (1-p) (may be NULL), dwb [512] = dgamma | dbeta; part: parts * 512 floats */
int cotr_train_ln_bwd(const float* dy, const float* s_in, const float* stats, const float* w, float* ds, float* da, float* part,
                      float* dwb, int rows, float p, uint32_t seed, cotr_stream stream);
/* Captured (hipGraph) training steps: the seed argument of a dropout launch is baked into the graph, so every training kernel that
 * draws a mask XORs its seed with the word at `salt` (device memory, read at kernel start); the captured step advances that word
 * itself (any device-side add), so each replay draws fresh masks and forward / backward of one step agree.  NULL (default) = the
 * seeds alone.  The registration is per DEVICE (the current one) and process-wide: the autograd engine runs the backward kernels
 * that recompute a mask on its own worker thread, and they must see the salt the forward drew the mask with.
 * cotr_train_clear_dropout_salt unregisters `salt` only if it is still the registered word (COTR_ERR_STATE otherwise). */
int cotr_train_set_dropout_salt(const unsigned int* salt);
int cotr_train_clear_dropout_salt(const unsigned int* salt);
/* in place x *= mask / (1-p) (n % 4 == 0); backward of y = dropout(relu(h)): dx = y > 0 ? dy / (1-p) : 0 (p == 0: relu backward) */
int cotr_train_dropout_fwd(float* x, size_t n, float p, uint32_t seed, cotr_stream stream);
int cotr_train_relu_drop_bwd(const float* dy, const float* y, float* dx, size_t n, float p, cotr_stream stream);
/* out[N] = column sums of x [M][N] (bias gradients); part: parts(M) * N floats */
int cotr_train_colsum_parts(int M);
int cotr_train_colsum(const float* x, float* part, float* out, int M, int N, cotr_stream stream);
/* dst [C][R] = src [R][C]^T (W^T once per optimiser step, for dX = dY . W on the GEMM kernels) */
int cotr_train_transpose(const float* src, float* dst, int R, int C, cotr_stream stream);
/* Convolution backward of the trainable backbone stages (layer2 / layer3, backbone.py:66-69) by explicit im2col over the NHWC
 * side-by-side activations [B][Hin][2*Win][Cin] (each half padded on its own, like the forward kernels):
 *   col [B*Hout*2*Wout][k*k*Cin] = im2col(x);  wgrad dW [Cout][k*k*Cin] = dz^T . col (cotr_train_gemm_tn);
 *   dgrad dcol = dz . W (GEMM), dx = col2im(dcol) (gather over the taps in a fixed order).
 * cotr_train_scale_rows: out[r][:] = w[r][:] * scale[r] (FrozenBN folded into the weights / unfolded from their gradient);
 * cotr_train_transpose_batched: per batch element dst[C][R] = src[R][C]^T (torch's [Cout][Cin][k*k] <-> packed [Cout][k*k][Cin]) */
int cotr_train_im2col(const float* x, float* col, int B, int Hin, int Win, int Cin, int ksize, int stride, cotr_stream stream);
int cotr_train_col2im(const float* dcol, float* dx, int B, int Hin, int Win, int Cin, int ksize, int stride, cotr_stream stream);
int cotr_train_scale_rows(const float* w, const float* scale, float* out, int rows, int cols, cotr_stream stream);
int cotr_train_transpose_batched(const float* src, float* dst, int batch, int R, int C, cotr_stream stream);
/* out [N][K] = A [M][N]^T . B [M][K] (dW = dY^T . X, both row-major, no transposed copies); N, K multiples of 64.
 * colsum != NULL: also colsum [N] = column sums of A (db) from the same pass over dY; it must be out + N*K (one buffer
 * [N*K + N]) so that one launch finishes both.  part: splits(M, N, K) * (N * K + N) floats */
int cotr_train_gemm_tn_splits(int M, int N, int K);
int cotr_train_gemm_tn(const float* A, const float* B, float* part, float* out, float* colsum, int M, int N, int K,
                       cotr_stream stream);
/* The partials of cotr_train_gemm_tn alone (part [splits][N*K (+ N with with_colsum)]): -> number of partials written (0 for
 * M == 0), < 0 = error.  cotr_train_ln_bwd / cotr_train_head_bwd with dwb == NULL likewise leave their partials unsummed.
 * cotr_train_reduce_jobs finishes ALL partials of a backward pass in one launch and accumulates them into the gradient buffers
 * (cotr_amd/train_ops.py: GradSink - replaces one reduction launch per weight plus autograd's add_ into .grad):
 *   for each job, each source in order:  dst[perm(e)] += scale[e / row_len] * sum_{p < nparts, in order} part[p * pstride + e]
 * jobs / srcs / chunk_job live in DEVICE memory; one workgroup per 1024-element chunk: chunk0 = a job's first chunk
 * ((numel + 1023) / 1024 chunks per job), nchunks = their total, chunk_job[c] = index of the job chunk c belongs to; taps > 1: element e = (row, tap, c) of a packed [row][taps][cin] record goes to (row, c, tap) (conv weight
 * gradients back to torch's [Cout][Cin][k][k]); vec = 1 promises 16-byte aligned pointers and counts that are multiples of 4. */
typedef struct cotr_reduce_src {
  const float* part;
  unsigned long long pstride;
  unsigned nparts, pad_;
} cotr_reduce_src;
typedef struct cotr_reduce_job {
  float* dst;
  const float* scale;
  unsigned numel, first_src, n_src, chunk0;
  unsigned row_len, cin, taps, vec;
} cotr_reduce_job;
int cotr_train_gemm_tn_parts(const float* A, const float* B, float* part, int M, int N, int K, int with_colsum, cotr_stream stream);
/* The same partials for the weight gradient of a convolution WITHOUT its im2col image (round 6): A = dz [B*Hout*2*Wout][Cout], the
 * second operand is gathered from x [B][Hin][2*Win][Cin] by the kernel (zeros where a tap leaves the half) - the bits of
 * cotr_train_im2col + cotr_train_gemm_tn_parts(dz, col, ..., 0).  part as for cotr_train_gemm_tn_parts with M = B*Hout*2*Wout,
 * N = Cout, K = ksize*ksize*Cin.  Returns the number of partials, or -1 where this form does not apply (Cin % 128 != 0, small
 * shapes): form the image and call cotr_train_gemm_tn_parts. */
int cotr_train_conv_wgrad_parts(const float* dz, const float* x, float* part, int B, int Hin, int Win, int Cin, int Cout, int ksize,
                                int stride, cotr_stream stream);
/* out[i] = part[0][i] + part[1][i] + ... in split order (i < n; part [nparts][n]): what cotr_train_gemm_tn does after its partials */
int cotr_train_sum_parts(const float* part, int nparts, size_t n, float* out, cotr_stream stream);
int cotr_train_reduce_jobs(const cotr_reduce_job* jobs, const cotr_reduce_src* srcs, const unsigned* chunk_job, int njobs, int nchunks,
                           cotr_stream stream);
/* torch.optim.Adam's update (train_cotr.py:49-57: betas (0.9, 0.999), eps 1e-8, no weight decay / amsgrad) over EVERY trainable
 * parameter in one launch, on flat gradient / exp_avg / exp_avg_sq buffers of one layout (cotr_amd/training.py: FusedAdam on the
 * GradSink's buffer):  m += (g - m)(1 - b1);  v = v b2 + (1 - b2) g g;  p -= lr[group] / bc1 * m / (sqrt(v) / bc2_sqrt + eps).
 * jobs / chunk_job (device memory): one job per parameter, one workgroup per 1024 elements, as for cotr_train_reduce_jobs; lr = HOST
 * array of ngroups <= 8 rates.  step == NULL: bias_correction1 = 1 - b1^t and bias_correction2_sqrt = sqrt(1 - b2^t) as given
 * (torch computes them on the host in double precision); step != NULL: *step (device, float) = t, corrections computed in the kernel
 * (captured steps). */
typedef struct cotr_adam_job {
  float* p;
  unsigned long long off;
  unsigned numel, chunk0, group, vec;
} cotr_adam_job;
int cotr_train_adam(const cotr_adam_job* jobs, const unsigned* chunk_job, int nchunks, const float* g, float* m, float* v, const float* lr,
                    int ngroups, double beta1, double beta2, double eps, double bias_correction1, double bias_correction2_sqrt,
                    const float* step, cotr_stream stream);
/* Every weight-shaped operand of a training step re-derived from the parameters in ONE launch (after the optimiser step): the W^T
 * slices of the dX GEMMs, the packed [Cout][k][k][Cin] convolution weights, the FrozenBN-scaled transposed convolution weights (one
 * fp32 multiply per element, as cotr_train_scale_rows makes).  A job is a batched strided transpose of 32 x 32 tiles:
 * dst[z*dz + c*dc + r] = src[z*sz + r*sr + c*sc] * (scale ? scale[r] : 1); tile_job[workgroup] = its job (device memory, like
 * cotr_train_reduce_jobs); tile0 = the job's first workgroup, tiles_r / tiles_c = ceil(R / 32) / ceil(C / 32). */
typedef struct cotr_perm_job {
  const float* src;
  float* dst;
  const float* scale;
  unsigned Z, R, C;
  unsigned sz, sr, sc, dz, dc;
  unsigned tile0, tiles_r, tiles_c, pad;
} cotr_perm_job;
int cotr_train_perm_jobs(const cotr_perm_job* jobs, const unsigned* tile_job, int njobs, int ntiles, cotr_stream stream);
/* last corr_embed layer 256 -> 2 (position_encoding.py:23-26): y [nb][nq][2]; backward: dh [rows][256], dwb [514] = dW2 | db2 */
int cotr_train_head_fwd(const float* x, const float* w, const float* b, float* y, int nb, int nq, cotr_stream stream);
int cotr_train_head_bwd_parts(int rows);
int cotr_train_head_bwd(const float* dy, const float* h, const float* w2, float* dh, float* part, float* dwb, int rows,
                        cotr_stream stream);
/* o = dropout(softmax(q k^T * qscale)) v per head (8 x 32; q [nb*nq][ldq], k [nb*512][ldk], v [nb*512][ldv]); lse [nb*nq][8]
 * (log2-domain log-sum-exp) for the backward */
int cotr_train_attention_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo, float* lse,
                             int nb, int nq, float qscale, float p, uint32_t seed, cotr_stream stream);
/* dq [nb*nq][lddq], dk [nb*512][lddk], dv [nb*512][lddv]; delta: nb*nq*8 floats of scratch; o / d_o share ldo.
 * scratch: cotr_train_attention_bwd_scratch(nb, nq) floats or NULL - room for the dQ partials of the one-pass backward when the keys
 * of a head are split over several workgroups (few pairs); without it that shape takes the two-kernel form */
int cotr_train_attention_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                             const float* d_o, int ldo, const float* lse, float* delta, float* dq, int lddq, float* dk, int lddk,
                             float* dv, int lddv, int nb, int nq, float qscale, float p, uint32_t seed, float* scratch,
                             cotr_stream stream);
size_t cotr_train_attention_bwd_scratch(int nb, int nq);

/* ---- engine-side input construction (one launch per zoom level) -------------
 * For each of n tasks: crop the square box (xa, ya, size_a) of image A and (xb, yb, size_b) of image B
 * (uint8 HWC RGB, DEVICE pointers; boxes int32 [n][6] on the device, inside the images), resize both to
 * 256x256 with Pillow's 8-bit BILINEAR resample (bit-exact), place them side by side, convert to float/255
 * and ImageNet-normalise: out float32 [n,3,256,512], directly consumable by cotr_encode.
 * Replaces per task: PIL resize x2 + two_images_side_by_side + to_tensor + normalize
 * (COTR/inference/refinement_task.py:105-120) and the H2D copy of sparse_engine.py:50.
 * max_size: largest crop edge among the boxes (sizes LDS; <= 16384). */
int cotr_crop_resize_pairs(const uint8_t* img_a, int ha, int wa, const uint8_t* img_b, int hb, int wb,
                           const int32_t* boxes, int n, float* out, int max_size, cotr_stream stream);

/* ---- dense initial pass: post-processing on the device ---------------------------------------
 * Replaces the host round trip of COTR/inference/inference_helper.py:137-160 (cotr_patch_flow_exhaustive.one_pass
 * tail) and :61-75 (merge_flow_patches) + COTR/utils/utils.py:69-83 (float_image_resize).
 *
 * cotr_dense_cycle: pred [n_pairs,256,512,2] = the network's answer for the query grid (j/512, i/256) ->
 *   maps [n_pairs,256,512,3] = (x, y, cycle error): self-composition through bilinear grid_sample (zeros padding,
 *   align_corners=False), x re-centred per half (:140-142), then the 2x3 affine of that half applied in double
 *   (:157-158).  affine [n_pairs][2][6] doubles (device): [p][0] left half -> image b, [p][1] right half -> image a.
 * cotr_dense_merge: for one image (side 0 = a / left halves, 1 = b / right halves) resize each entry's 256x256x3
 *   half to its patch boxes[p] = (x, y, size) with Pillow's mode-'F' BILINEAR (bit-exact) and keep per pixel the
 *   entry with the lowest cycle error, later entries winning ties -> flow [H,W,2], conf [H,W] (100 where uncovered). */
int cotr_dense_cycle(const float* pred, int n_pairs, const double* affine, float* maps, cotr_stream stream);
int cotr_dense_merge(const float* maps, const int32_t* boxes, int n_pairs, int side, int H, int W, float* flow,
                     float* conf, cotr_stream stream);

/* src [Hs,Ws,C] float -> dst [Hd,Wd,C] with Pillow's mode-'F' BILINEAR resample, channel by channel, bit-exact:
 * utils.float_image_resize (COTR/utils/utils.py:69-83), used by SparseEngine's 'stretching' mode to bring the dense maps
 * of the squared images back to the image shape (sparse_engine.py:124-129). */
int cotr_resize_f32(const float* src, int Hs, int Ws, int C, float* dst, int Hd, int Wd, cotr_stream stream);

/* ---- tuning knobs -------------------------------------------------------------------------------------------
 * Named integer switches that choose between launch schedules / kernel variants with the SAME results (bit-identical unless a
 * knob's line says otherwise).  They are not part of the drop-in boundary: a binding never needs them.  ONE SET PER HANDLE:
 * cotr_set_knob(h, ...) affects only that handle's cotr_encode / cotr_decode / cotr_forward / cotr_backbone calls (two handles, or
 * two threads, can differ); a new handle starts from the shipped defaults.  h == NULL addresses the process-wide set read by the
 * handle-less op-level entry points below (cotr_op_*, cotr_bench_*, cotr_train_*: tests and tools).  count / name enumerate the
 * knobs, get returns the current and the shipped default value, reset puts every knob of that set back to its default; a value
 * outside a knob's range is refused with COTR_ERR_ARG.
 *
 *   encode_chunk               pairs per backbone / encoder pass inside cotr_encode (1..128, default 64; 64 is +2-3 % over 32 from 64
 *                              pairs up, 128 is -20 %).  Scratch of a pass is ~66 MB per pair: set BEFORE sizing a caller-supplied
 *                              workspace (cotr_scratch_bytes uses the handle's value)
 *   attention_fusion_max_rows  up to this many rows (default 4096; 0 = never) the attention kernel also does the output projection
 *                              (per-head partials summed + bias + residual + LayerNorm by one ln_reduce launch) and, in the
 *                              decoder, the q projection of its own queries - above 1024 rows only where its grid of 32-row tiles x 8
 *                              heads fills whole rounds of the CUs to >= 90 % (round 6)
 *   ffn_fusion_max_rows        the fused FFN block (ffn.hip) for at most this many rows (default 4096; 0 = never; above 1024 rows
 *                              under the same fill rule), with at most
 *   ffn_fused_max_chunks       hidden-unit chunks = partial output slabs per row tile (2, 4, 8 or 16; default 16)
 *   ks3                        1 (default): K-deep small-M GEMMs the table gives to the two-stage LDS-DMA k-split run its three-stage form
 *   dual_conv                  1 (default): downsample + conv1 of a ResNet stage's entry block as one launch at few pairs
 *   fused_stem                 1 (default): conv1 + bn1 + relu + maxpool as one launch (stem_pool.hip)
 *   xcd_mapping                workgroup -> XCD mapping, a bit field (default 1): bits 0-1 GEMM / conv tiles (0 = column tiles over the
 *                              8 XCDs, 2 = row tiles, 1 = per launch by operand size); bit 2 fused FFN: hidden-unit chunks over XCDs;
 *                              bit 3 attention: heads over XCDs; bit 4 fused FFN: plain instead of write-through partial stores;
 *                              bit 5 att_rows: the plain (tile, pair) grid also where the pairs are a multiple of 8 (default there: all tiles of a
 *                              pair on ONE XCD - round 6: fabric traffic of the kernel / 2.5-3.7, its time -0 ... -3 %)
 *   attention_fused_splits     key splits of the fused attention: 0 (default = 4), 4, 8, 48 / 84 (encoder / decoder separately)
 *   conv_patch                 1 (default): layer3's 3x3 convolutions at few pairs load their input patch once
 *   pos_table_min_rows         token rows from which the encoder in-projections take pos . W^T from tables built at cotr_load_weights
 *                              (default 8192; >= 2^30 also turns the K/V projection's table off; +1 fp32 addition per output)
 *   attention_wide_min_rows    query rows of a launch from which the 64-query / resident-K/V attention kernels are used (default 4096)
 *   attention_wide_occupancy   wavefronts per SIMD of the 64-query kernel: 3 (default) or 2
 *   attention_splits           key splits of the plain attention kernel: 0 (automatic), 1, 2, 4, 8, 16
 *   attention_resident         1 (default): K_h / V_h resident in LDS for many rows (attention_res_kernel)
 *   conv1x1_dense              1 (default): a 1x1 stride-1 convolution is launched as the dense product of its pixel rows
 *   ws_flags                   wave-specialised large tiles (configurations 40 / 41): bit 0 priority for the MFMA wavefronts' loop,
 *                              bit 1 (default) for the loaders
 *   bottleneck_max_pairs       layer1's bottlenecks as ONE launch each (bottleneck.hip) up to this many pairs per pass (default 5)
 *   train_attention_form       training attention backward: 0 (default) = by shape, 1-3 = force the first / second / one-pass form
 *   att_rows_min_rows, ffn_rows_min_rows   query rows / rows from which the attention sub-layer / the FFN block run as ONE launch
 *                              (att_rows.hip / ffn_rows.hip; default 8192), and
 *   rows_min_fill              the least fill, in percent, of the last round of their 64-row tiles over the CUs (default 75)
 *   conv23_min_pairs, conv23m_min_pairs, expand_min_rows   layer1 / layer2 conv2 -> conv3 and layer1.0's downsample + conv1 as one
 *                              launch from this many pairs per pass / rows (defaults 5 / 16 / 65536; conv23 above bottleneck_max_pairs only;
 *                              conv23m not where its single round fills the CUs unevenly: 17 ... 27 pairs on 256 CUs)
 *   batch_split                1 (default): a batch is walked in the passes the measured time-against-pairs staircase prefers - encode
 *                              passes from a measured table (csrc/enc_split.inc, tools/batch_cost.py: 17 pairs = 16 + 1, 33 = 32 + 1),
 *                              decode passes where a prefix of the pairs fills the one-launch rows kernels and the whole does not;
 *                              0: passes of encode_chunk pairs / 32768 query rows only (cotr_batch_chunks reports the passes)
 *   side_stream                cotr_forward with few rows (B * Q <= 8192, one pass): bit 0 the query encoding, bit 1 the K/V projections of
 *                              decoder layers 1-5 run on a second stream owned by the handle, beside the chain; joined before the
 *                              call returns (default 0: measured, profiles/r6_ab_side_stream_*.txt)
 * libcotr_hip_exp.so (the experimental build, -DCOTR_EXPERIMENTAL) adds the knobs of the measured dead ends, all off by default:
 *   head_fusion_max_rows, ffn_preln, ffn_tail, coop_tail, coop_tail_spin, gemm_ln_min_rows, l2_warm  (cotr_amd/csrc/experimental/experimental.h)
 * and the RESEARCH path of docs/LABNOTES.md 3e (not a dead end; off by default, results as close to fp64 as the fp32 path but not its bits):
 *   split_f16                  0 (default) = off; 1 = the backbone + input_proj of a pass on packed split-f16 tensors (three f16 MFMAs per
 *                              fp32 product); 2 = also the projections / FFN blocks / corr_embed of the many-row (>= 8192 rows) transformer
 *                              path; 3 = also attention in both stacks.  |activations| must stay below 65504
 *   split_f16_min_pairs        the backbone pass takes that path only from this many pairs per pass (default 8) */
int cotr_knob_count(void);
const char* cotr_knob_name(int i);
int cotr_get_knob(cotr_handle h, const char* name, int* value, int* default_value);
int cotr_set_knob(cotr_handle h, const char* name, int value);
/* COTR_OK if cotr_set_knob would accept (name, value), COTR_ERR_ARG otherwise; changes no knob set */
int cotr_check_knob(const char* name, int value);
int cotr_reset_knobs(cotr_handle h);
/* 1 in libcotr_hip_exp.so, 0 in the product library */
int cotr_is_experimental(void);

/* ---- GEMM configuration tuning (tools/tune_gemm.py) and per-config tests ------------------- */
int cotr_gemm_num_configs(void);
/* one layer1 bottleneck from unpacked device weights (tests): x [B][64][128][cin] -> y [B][64][128][256]; cin = 64 with the
 * downsample branch (wd != NULL) or 256 without; w1 [64][cin], w2 [64][3][3][64], w3 [256][64], wd [256][64]; s* / b* FrozenBN
 * scale / bias per output channel */
int cotr_op_bottleneck(const float* x, float* y, int B, int cin, const float* w1, const float* w2, const float* w3, const float* wd,
                       const float* s1, const float* b1, const float* s2, const float* b2, const float* s3, const float* b3,
                       const float* sd, const float* bd, cotr_stream stream);
/* microseconds per launch of one shape under config `cfg` (-1: the library's own choice), measured
 * with HIP events around a captured graph of `iters` launches on a private stream */
int cotr_bench_linear(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int cfg,
                      int iters, float* us);
int cotr_bench_conv(const float* x, const float* w, const float* scale, const float* bias, float* y, int B, int Hin,
                    int Win, int Cin, int Cout, int ksize, int stride, int cfg, int iters, float* us);
int cotr_op_linear_cfg(const float* x, const float* w, const float* bias, const float* residual, int relu, float* y,
                       int M, int N, int K, int cfg, cotr_stream stream);
int cotr_op_conv_cfg(const float* x, const float* w, const float* scale, const float* bias, const float* residual,
                     int relu, float* y, int B, int Hin, int Win, int Cin, int Cout, int ksize, int stride, int cfg,
                     cotr_stream stream);

/* phase timestamps of the fused FFN launches that follow into `times` (device, [workgroups][8] uint64, 100 MHz wall clock); NULL = off */
int cotr_debug_ffn_times(unsigned long long* times);
/* the same for the fused attention launches (cotr_op_attention_fused / the forward's small-row path) */
int cotr_debug_attention_times(unsigned long long* times);
/* the launch configuration the library picks for this convolution (tools) */
int cotr_gemm_pick_conv(int B, int Hin, int Win, int Cin, int Cout, int ksize, int stride);
/* one convolution launch whose k-split kernel writes phase timestamps (100 MHz wall clock) of every workgroup to `times`
 * (device memory, [workgroups][8] uint64; slots 0..4 = entry, loads issued, first data usable, K loop done, stored) */
int cotr_debug_conv_times(const float* x, const float* w, const float* scale, const float* bias, float* y, int B, int Hin, int Win,
                          int Cin, int Cout, int ksize, int stride, int cfg, unsigned long long* times, cotr_stream stream);

/* two independent convolutions of the SAME input in one launch under config `cfg` (the dual-launch path of the entry blocks) */
int cotr_op_conv_dual_cfg(const float* x, const float* w0, const float* scale0, const float* bias0, int relu0, float* y0, int Cout0,
                          int ksize0, int stride0, const float* w1, const float* scale1, const float* bias1, int relu1, float* y1,
                          int Cout1, int ksize1, int stride1, int B, int Hin, int Win, int Cin, int cfg, cotr_stream stream);

/* ---- libcotr_hip_exp.so only (COTR_EXPERIMENTAL): op-level entry points of two measured dead ends ---- */
#ifdef COTR_EXPERIMENTAL
/* decoder tail: out[b][q][0..1] = corr_embed(LayerNorm(x)) for x [nb_pairs*nq, 256]; hs (optional) receives the normalised rows */
int cotr_op_dec_head(const float* x, const float* nw, const float* nb, const float* w0, const float* b0, const float* w1,
                     const float* b1, const float* w2, const float* b2, float* hs, float* out, int nb_pairs, int nq, int q_total,
                     cotr_stream stream);
/* y [M][256] = LayerNorm(x [M][K] . w [256][K]^T + bias + residual [M][256]) * ln_w + ln_b in one launch (op-level entry for the tests) */
int cotr_op_linear_ln(const float* x, const float* w, const float* bias, const float* residual, const float* ln_w, const float* ln_b,
                      float* y, int M, int K, cotr_stream stream);
/* RESEARCH (csrc/experimental/gemm_h2.h): y[i] = packed split-f16 form of x[i] (f16(x) | f16((x - f16(x)) * 2^11) << 16), n a multiple of 4,
 * 16-byte aligned, in place allowed.  cotr_op_linear / cotr_op_conv with the forced configurations 46 / 47 take BOTH operands (activations
 * and weights) in this form and return fp32; |x| must stay below 65504.  Not bit-identical to the fp32-MFMA path. */
int cotr_op_split_h2(const float* x, void* y, size_t n, cotr_stream stream);
int cotr_op_unsplit_h2(const void* x, float* y, size_t n, cotr_stream stream);   /* the inverse: exact */
/* flags of the following cotr_op_linear_cfg / cotr_op_conv_cfg calls of this thread on configurations 46 / 47: bit 0 = y is written packed,
 * bit 1 = the residual is packed (what a chain of such launches passes from one to the next); 0 restores fp32 residual / output */
int cotr_op_set_h2_flags(int flags);
/* RANGE SAFETY of the research path: a cotr_encode / cotr_decode / cotr_forward that ran with split_f16 on and packed an activation outside f16's
 * range (|x| >= 65504) is run again on the fp32-MFMA kernels before it returns; this counts those re-runs (process-wide) */
long cotr_h2_fallbacks(void);
/* cotr_op_attention on PACKED k / v (q fp32 or packed: q_packed; o fp32 or packed: out_packed), csrc/experimental/attention_h2.hip */
int cotr_op_attention_h2(const float* q, int ldq, int q_packed, const float* k, const float* v, int ldkv, float* o, int ldo, int out_packed,
                         int nb, int nq, cotr_stream stream);
#endif

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* COTR_HIP_H */
