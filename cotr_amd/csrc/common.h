// Shared declarations of the gfx950 kernels behind libcotr_hip.so.
#pragma once
// The research library (libcotr_hip_exp.so, -DCOTR_EXPERIMENTAL: split-f16 products and the measured dead ends) is built from PATCHED
// copies of the translation units it changes (csrc/experimental/patches/*.patch applied to api.hip, attention.hip, ffn.hip, gemm.hip,
// gemm_big.hip, pointwise.hip and to this header at build time -> csrc/experimental/gen/, cotr_amd/build.py) plus the untouched
// product ones, which find its declarations through this redirect - the only mention of it in the product sources.
#ifdef COTR_EXPERIMENTAL
#include "experimental/gen/common_exp.h"
#else
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// LDS-DMA (global_load_lds) data is ordered for OTHER wavefronts' ds_reads only by the issuing wavefront's vmcnt wait
// followed by a barrier.  The compiler does not always put that wait before s_barrier (it may sink it to the issuing
// wavefront's own first ds_read, which leaves the other wavefronts' reads unordered): state it explicitly.
#define LDS_DMA_WAIT_ALL() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

// 16-byte store, optionally write-through (sc1): for buffers that another kernel reads exactly once from every XCD (the
// per-chunk / per-head partial outputs), so that nothing dirty is left in this XCD's L2 for the launch boundary to write
// back.  Scalar sc1 stores cost ~6x a 16-byte one per byte (MI355X_MICROARCH.md), hence the float4 form.
__device__ __forceinline__ void store_f32x4(float* dst, const f32x4 v, const bool write_through) {
  if (write_through) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
  else *reinterpret_cast<f32x4*>(dst) = v;
}

// ---------------------------------------------------------------------------------------------
// Per-device one-time state.  A process may hold handles on several GPUs (cotr_create(&h, device)): function attributes
// (the > 64 KB dynamic-LDS opt-in is per device), the zero buffer of the LDS-DMA kernels and the arrival counters of the
// fused FFN tail live on ONE device each, so they are keyed by the device that is current when a launch helper runs.
// The ABI entry points make the handle's device current (DeviceScope, api.hip) and record it here so that the launch
// helpers do not have to ask the runtime on every launch.
// ---------------------------------------------------------------------------------------------
#define COTR_MAX_DEVICES 64
extern thread_local int cotr_tls_device;   // device made current by the innermost DeviceScope of this thread, or -1
static inline int cotr_current_device() {
  if (cotr_tls_device >= 0) return cotr_tls_device;
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= COTR_MAX_DEVICES) d = 0;
  return d;
}
// Compute units of the current device, asked once per device.  The dispatch rules that count "rounds of the chip" (api.hip) and the
// first-round stagger of conv23 / conv23m / expand (CU slot = blockIdx >> 8) are written for the MI355X's 256 CUs in one partition:
// on another count (CPX / NPS partition modes) the rules use the real number and the stagger is skipped.
static inline int cotr_num_cus() {
  static int cus[COTR_MAX_DEVICES] = {};
  const int d = cotr_current_device();
  if (cus[d] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256;
    cus[d] = n;
  }
  return cus[d];
}
struct PerDeviceFlag {
  bool done[COTR_MAX_DEVICES] = {};
  bool get() const { return done[cotr_current_device()]; }
  void set() { done[cotr_current_device()] = true; }
};

// ---------------------------------------------------------------------------------------------
// Tuning knobs.  ONE set per handle (cotr_ctx::knobs; cotr_set_knob(h, name, value)) plus one process-wide set for the
// handle-less op-level entry points (cotr_op_*, cotr_bench_*, cotr_train_*: tests and tools; cotr_set_knob(NULL, ...)).  Every
// ABI call makes its set current for the calling thread (KnobScope, api.hip); the launch helpers read it through knob(): two
// handles - or two threads - can run with different settings, and a new handle always starts from the shipped defaults.
// (The research library appends its own knobs in its copy of this header; the product has neither those knobs nor the code behind them.)
// ---------------------------------------------------------------------------------------------
enum KnobId {
  KN_ENCODE_CHUNK,               // pairs per backbone / encoder pass (<= 128).  64: +2-3 % over 32 from 64 pairs up, 128: -20 %
  KN_ATTENTION_FUSION_MAX_ROWS,  // attention with the out-projection (decoder: + q projection) fused in, up to this many rows
  KN_FFN_FUSION_MAX_ROWS,        // fused FFN block up to this many rows (slower than the two GEMMs from ~1300 rows on)
  KN_KS3,                        // three-stage LDS-DMA k-split (config 30) where the table says its two-stage form (24)
  KN_DUAL_CONV,                  // downsample + conv1 of a stage's entry block as one launch (few pairs)
  KN_FUSED_STEM,                 // conv1 7x7 + bn + relu + maxpool in one launch
  KN_XCD_MAPPING,                // bits 0-1: GEMM tiles over XCDs (0 columns, 1 by operand size, 2 rows); bit 2: FFN chunks over XCDs;
                                 // bit 3: attention heads over XCDs; bit 4: plain (not write-through) stores for the FFN partials;
                                 // bit 5: att_rows on the plain (tile, pair) grid also where the pairs are a multiple of 8 (default: all tiles of a pair on ONE XCD there)
  KN_ATTENTION_FUSED_SPLITS,     // key splits of the fused attention: 0 (= 4), 4, 8, 48 / 84 (encoder / decoder separately)
  KN_CONV_PATCH,                 // layer3's 3x3 convolutions load their input patch once (config 31)
  KN_POS_TABLE_MIN_ROWS,         // token rows from which the encoder in-projections take pos . W^T from the tables
  KN_ATTENTION_WIDE_OCCUPANCY,   // wavefronts per SIMD of the 64-query attention kernel (2 or 3)
  KN_ATTENTION_WIDE_MIN_ROWS,    // query rows of a launch from which the 64-query / resident-K/V kernels are used
  KN_ATTENTION_SPLITS,           // key splits of the plain 32-query attention kernel (0 = automatic)
  KN_CONV1X1_DENSE,              // 1x1 stride-1 convolutions run the dense instantiation of their configuration
  KN_WS_FLAGS,                   // wave-specialised large tiles: bit 0 s_setprio(1) around the MFMA loop, bit 1 s_setprio(3) for the loaders
  KN_BOTTLENECK_MAX_PAIRS,       // layer1 bottlenecks as one launch each up to this many pairs per pass (-4.3 % at 1 pair, +1.4 % at 32)
  KN_TRAIN_ATTENTION_FORM,       // training attention backward: 0 = by shape, 1-3 = force a form
  KN_ATTENTION_RESIDENT,         // K_h / V_h resident in LDS (attention_res_kernel) for many rows
  KN_ATT_ROWS_MIN_ROWS,          // attention sub-layer ([q projection,] attention, out projection, residual, LayerNorm) as ONE launch (att_rows.hip) from this many query rows
  KN_FFN_ROWS_MIN_ROWS,          // FFN block + residual + LayerNorm as ONE launch (ffn_rows.hip: 64-row tiles, hidden units dealt to the wavefronts) from this many rows
  KN_CONV23_MIN_PAIRS,           // layer1: conv2 (3x3) -> conv3 (1x1 expansion) + identity + ReLU as ONE launch (conv23.hip) from this many pairs per pass
  KN_CONV23M_MIN_PAIRS,          // layer2: conv2 (3x3) -> conv3 (1x1 expansion) + identity + ReLU as ONE launch (conv23m.hip) from this many pairs per pass
  KN_EXPAND_MIN_ROWS,            // layer1 block 0's downsample + conv1 as one launch (expand.hip) from this many rows
  KN_ROWS_MIN_FILL,              // att_rows / ffn_rows: least fill (percent) of their 64-row tiles' last round over the CUs for the one-launch form to be taken
  KN_SIDE_STREAM,                // cotr_forward, few rows: query-only / memory-only work on a second stream of the handle, beside the chain
                                 // (bit 0: the query encoding beside the backbone; bit 1: the K/V projections of decoder layers 1-5 beside decoder layer 0)
  KN_FFN_FUSED_MAX_CHUNKS,       // fused FFN (ffn.hip): most hidden-unit chunks = partial output slabs per row tile (16; 8 = half as many, twice as large
                                 // producers - round 6: measured with its ln_reduce, slower)
  KN_BATCH_SPLIT,                // 1: a batch is walked in the chunks the measured staircase prefers (17 pairs = 16 + 1, 33 = 32 + 1: enc_split.inc, api.hip enc_next_chunk / dec_next_pairs); 0: in chunks of encode_chunk pairs / 32768 query rows only
  KN_COUNT
};
struct KnobSet {
  int v[KN_COUNT];
};
extern thread_local const KnobSet* cotr_tls_knobs;   // the set of the ABI call in flight on this thread (api.hip)
static inline int knob(int id) { return cotr_tls_knobs->v[id]; }

// ---------------------------------------------------------------------------------------------
// C[M,N] = epilogue( A'[M,K] . W[N,K]^T )
//   A' row m, column k:
//     mode DENSE : A[m*lda + k]  (+ A2[(m % a2_row_mod)*lda2 + k] when the tile's first column n0
//                  satisfies (n0 % a2_period) < a2_width; a2_row_mod == 0 -> row m)
//     mode CONV  : implicit im2col of an NHWC "side-by-side" activation [B,Hin,2*Win,Cin]
//                  (the two 256x256 halves of a pair sit next to each other on W, each padded
//                  on its own: a tap never crosses the seam), k = (ky*ks + kx)*Cin + c
//     mode STEM  : implicit im2col of the NCHW input image [B,3,256,512], 7x7/2 pad 3,
//                  k = c*49 + ky*7 + kx, K padded to 160 with zero weights
//   epilogue, per element (m,n):  v = acc
//     scale != null : v = v*scale[n] + bias[n]        (FrozenBN, COTR/models/backbone.py:54-56)
//     else bias     : v = v + bias[n]
//     n < colscale_n: v = v * colscale                (q * head_dim^-0.5 of nn.MultiheadAttention)
//     residual      : v = v + residual[m*ldr + n]
//     relu          : v = max(v, 0)
// ---------------------------------------------------------------------------------------------
enum { GEMM_DENSE = 0, GEMM_CONV = 1, GEMM_STEM = 2 };

// Division by a launch-time constant without a hardware divide.  hipcc expands `n / d` with a runtime d into ~25 dependent
// instructions (v_rcp_iflag + a Newton step + two corrections); the convolution kernels did a dozen of them per row before
// their first load could be issued - 500 of the ~800 instructions (1.7 us) between workgroup entry and "loads issued" in the
// phase stamps of round 2.  Every divisor here is known on the host: q = umulhi(n, m) with m = 2^32 / d for a power of two
// (all of COTR's geometry; exact), floor(2^32 / d) + 1 otherwise (exact for n * d < 2^32; the launch helpers check the range).
struct FastDiv {
  unsigned mul;  // q = umulhi(n, mul) + n * one
  int one;       // 1 only for d == 1 (the multiplier would be 2^32)
  int d;
};
static inline FastDiv fastdiv_make(int d) {
  FastDiv f;
  f.d = d > 0 ? d : 1;
  f.one = f.d == 1;
  if (f.d == 1) f.mul = 0;
  else if ((f.d & (f.d - 1)) == 0) f.mul = (unsigned)(0x100000000ull / (unsigned)f.d);        // exact for every n
  else f.mul = (unsigned)(0x100000000ull / (unsigned)f.d) + 1u;                                // exact for n * d < 2^32
  return f;
}
// largest n for which fastdiv is exact (n * e < 2^32 with e = d - 2^32 mod d <= d)
static inline long long fastdiv_max_n(const FastDiv& f) {
  return (f.d & (f.d - 1)) == 0 ? 0x7fffffffll : (long long)(0x100000000ull / (unsigned)f.d) - 1;
}
__device__ __forceinline__ int fastdiv(int n, const FastDiv& f) {   // n >= 0; branch-free: multiply-high, multiply, add
  return (int)(__umulhi((unsigned)n, f.mul) + (unsigned)n * (unsigned)f.one);
}

struct GemmParams {
  int M, N, K;
  const float* A;
  int lda;
  const float* A2;
  int lda2, a2_row_mod, a2_period, a2_width;
  const float* W;  // [N][K]
  float* C;
  int ldc;
  // conv geometry (per half)
  int Hin, Win, Cin, Hout, Wout, ksize, stride, pad;
  // epilogue
  const float* scale;
  const float* bias;
  const float* residual;
  int ldr;
  int res_row_mod;     // > 0: the residual is a row-periodic table, row m reads residual[m % res_row_mod] (large-tile kernels only)
  int relu;
  float colscale;
  int colscale_n;
  const float* zeros;  // >= 16 B of zeros in global memory (source of padded / out-of-range tiles for LDS-DMA)
  int xcd_msplit;      // workgroup -> tile mapping, see gemm_tile_coords
  int ws_flags;        // wave-specialised large tiles (gemm_big.hip): priorities, see gemm_set_ws_flags
  // launch-time divisors (gemm_fill_divs, called by every launch helper): column tiles of the launch's tile shape; the
  // convolution's pixel decomposition (Hout * 2*Wout, 2*Wout, Wout), channel tiles per tap (Cin / 32), ksize; the x + pos
  // prologue's row period and column period; the row period of a table residual
  FastDiv fd_tiles_n, fd_hw, fd_w2o, fd_wout, fd_tpt, fd_ks, fd_a2row, fd_a2per, fd_resrow;
  unsigned long long* dbg;  // nullptr, or [workgroups][8] phase timestamps (100 MHz wall clock) written by wavefront 0 of the
                            // k-split kernels: entry, loads issued, first data usable, K loop done, stored (cotr_debug_conv_times)
};

// blockIdx -> (row tile, column tile).  Consecutive workgroups land on consecutive XCDs (8, each with its own 4 MB L2), so the
// mapping decides which operand every XCD's L2 pulls a private copy of through the fabric:
//   default  column tile fastest: with tiles_n a multiple of 8 an XCD owns 1/8 of the column tiles and walks ALL row tiles:
//            W crosses the fabric once chip-wide, the activations once per XCD (right when weights >> activations: layer3,
//            the transformer at one pair)
//   M-split  an XCD owns every 8th ROW tile and walks all column tiles: activations once chip-wide, W once per XCD
//            (right when the activation operand is the larger one: layer1/layer2 at any batch, everything when batched)
// Time-neutral at one pair (the path is latency-bound there); it is what the L2<->fabric byte counters see.
__device__ __forceinline__ bool gemm_tile_coords(const GemmParams& p, int bm, int bn, int bid, int& m0, int& n0) {
  const int tiles_n = p.fd_tiles_n.d;   // == p.N / bn (gemm_fill_divs)
  if (p.xcd_msplit) {
    const int tiles_m = (p.M + bm - 1) / bm;   // bm is a power of two at every call site
    const int a = bid >> 3;
    const int q = fastdiv(a, p.fd_tiles_n);    // bid / (8 * tiles_n)
    const int mt = q * 8 + (bid & 7);
    if (mt >= tiles_m) return false;
    m0 = mt * bm;
    n0 = (a - q * tiles_n) * bn;
  } else {
    const int q = fastdiv(bid, p.fd_tiles_n);
    m0 = q * bm;
    n0 = (bid - q * tiles_n) * bn;
  }
  return true;
}
__device__ __forceinline__ int fastmod(int n, const FastDiv& f) { return n - fastdiv(n, f) * f.d; }
// output row mm of a convolution -> (pair b, output row ho, half `side`, column wl inside the half)
__device__ __forceinline__ void conv_row_decompose(const GemmParams& p, int mm, int& b, int& ho, int& side, int& wl) {
  b = fastdiv(mm, p.fd_hw);
  const int rem = mm - b * p.fd_hw.d;
  ho = fastdiv(rem, p.fd_w2o);
  const int wo = rem - ho * p.fd_w2o.d;
  side = fastdiv(wo, p.fd_wout);
  wl = wo - side * p.fd_wout.d;
}
// K tile kt of a convolution -> (ky, kx, first channel c0)
__device__ __forceinline__ void conv_ktile_decompose(const GemmParams& p, int kt, int& ky, int& kx, int& c0) {
  const int tap = fastdiv(kt, p.fd_tpt);
  c0 = (kt - tap * p.fd_tpt.d) * 32;
  ky = fastdiv(tap, p.fd_ks);
  kx = tap - ky * p.fd_ks.d;
}
static inline int gemm_grid_tiles(const GemmParams& p, int bm, int bn) {
  const int tiles_m = (p.M + bm - 1) / bm;
  return (p.xcd_msplit ? (tiles_m + 7) / 8 * 8 : tiles_m) * (p.N / bn);
}
// fills the launch-time divisors for a launch with column tiles of bn; false if a quotient would leave fastdiv's exact range
static inline bool gemm_fill_divs(GemmParams& p, int mode, int bm, int bn) {
  p.fd_tiles_n = fastdiv_make(p.N / bn);
  const long long grid = gemm_grid_tiles(p, bm, bn);
  bool ok = grid <= fastdiv_max_n(p.fd_tiles_n);
  p.fd_a2row = fastdiv_make(p.a2_row_mod > 0 ? p.a2_row_mod : 1);
  p.fd_a2per = fastdiv_make(p.a2_period > 0 ? p.a2_period : 1);
  p.fd_resrow = fastdiv_make(p.res_row_mod > 0 ? p.res_row_mod : 1);
  ok = ok && p.M <= fastdiv_max_n(p.fd_a2row) && p.N <= fastdiv_max_n(p.fd_a2per) && p.M <= fastdiv_max_n(p.fd_resrow);
  if (mode == GEMM_CONV) {
    p.fd_hw = fastdiv_make(p.Hout * 2 * p.Wout);
    p.fd_w2o = fastdiv_make(2 * p.Wout);
    p.fd_wout = fastdiv_make(p.Wout);
    p.fd_tpt = fastdiv_make(p.Cin / 32);
    p.fd_ks = fastdiv_make(p.ksize);
    // the convolution kernels address the activation tensor with 32-bit element offsets
    const long long pairs = (p.M + (long long)p.Hout * 2 * p.Wout - 1) / ((long long)p.Hout * 2 * p.Wout);
    ok = ok && (pairs + 1) * p.Hin * 2 * p.Win * p.Cin < 0x7fffffffll;
    ok = ok && p.M <= fastdiv_max_n(p.fd_hw) && p.Hout * 2 * p.Wout <= fastdiv_max_n(p.fd_w2o) &&
         2 * p.Wout <= fastdiv_max_n(p.fd_wout) && p.K / 32 <= fastdiv_max_n(p.fd_tpt) && p.ksize * p.ksize <= fastdiv_max_n(p.fd_ks);
  } else {
    p.fd_hw = p.fd_w2o = p.fd_wout = p.fd_tpt = p.fd_ks = fastdiv_make(1);
  }
  return ok;
}

int launch_gemm(int mode, const GemmParams& p, hipStream_t s);            // tuned / modelled config
int launch_gemm_cfg(int mode, int cfg, const GemmParams& p, hipStream_t s);  // explicit config (tuning, tests)
int gemm_pick_config(int mode, const GemmParams& p);
int launch_gemm_big(int mode, int variant, const GemmParams& p, hipStream_t s);  // gemm_big.hip: 0 = 128x128, 1 = 128x64
int launch_gemm_wp(int mode, int variant, const GemmParams& p, hipStream_t s);   // gemm_wp.hip: wave-private K chunks
int launch_gemm_wp_dual(int mode, int variant, const GemmParams& p0, const GemmParams& p1, hipStream_t s);
int wp_variant_tile(int variant, int* bm, int* bn, size_t* lds);
// Two INDEPENDENT problems of the same mode in ONE launch (grid = tiles of p0 followed by tiles of p1, same kernel
// configuration): the downsample branch of a bottleneck next to its conv1 (torchvision Bottleneck.forward: both read the block
// input).  Supported configurations: the k-split ones without LDS-DMA (kinds 1, 2) and the large tiles (kind 4).
int launch_gemm_dual_cfg(int mode, int cfg, const GemmParams& p0, const GemmParams& p1, hipStream_t s);
int launch_gemm_big_dual(int mode, int variant, const GemmParams& p0, const GemmParams& p1, hipStream_t s);
bool gemm_cfg_supports_dual(int cfg);
int gemm_num_configs();
const float* gemm_zero_buffer();  // per-DEVICE buffer of zeros (LDS-DMA padding source), on the current device
// softmax(q k^T) v for 8 heads of 32; q rows are [nb][nq], keys/values [nb][512]
int launch_attention(const float* q, int ldq, const float* k, const float* v, int ldkv, float* o, int ldo,
                     int nb, int nq, hipStream_t s);

// the same with the q projection as prologue (wq != nullptr: q = ((x + x2) . wq_h^T + bq_h) * qscale, `q` unused) and / or the
// output projection as epilogue (wo != nullptr: per-head partial outputs [8][nb*nq][256] to `part`; `o` may be nullptr)
int launch_attention_fused(const float* q, int ldq, const float* x, const float* x2, const float* wq, const float* bq,
                           float qscale, const float* k, const float* v, int ldkv, float* o, int ldo, const float* wo,
                           float* part, int nb, int nq, hipStream_t s);

int launch_layernorm(const float* x, const float* w, const float* b, float* y, int rows, hipStream_t s);
// lin_sine encoding; point (bi, qi) read from pts[((bi*q_total) + qi)*2], written to row bi*nq+qi
int launch_posenc(const float* pts, float* y, int nb, int nq, int q_total, hipStream_t s);
int launch_pos_table(float* y /*[512][256]*/, hipStream_t s);
int launch_maxpool(const float* x, float* y, int B, int Hin, int Win, int C, hipStream_t s);
// y[(bi*q_total + qi)*2 + j] = x[bi*nq+qi, :] . w[j, :] + b[j]   (last corr_embed layer, 256 -> 2)
int launch_head2(const float* x, const float* w, const float* b, float* y, int nb, int nq, int q_total,
                 hipStream_t s);

// conv1 7x7/2 + FrozenBN + ReLU + maxpool 3x3/2 in one launch (stem_pool.hip): img NCHW [B,3,256,512] -> [B,64,128,64]
int launch_stem_pool(const float* img, const float* w, int wk, const float* scale, const float* bias, float* out, int B,
                     hipStream_t s);

// batched crop + Pillow-bilinear resize to 256x256 + side-by-side + ImageNet normalise (crop_resize.hip)
int launch_crop_resize(const uint8_t* img_a, int ha, int wa, const uint8_t* img_b, int hb, int wb,
                       const int32_t* boxes, int n, float* out, int max_size, hipStream_t s);

// dense-pass post-processing (dense_post.hip): cycle error + affine, then Pillow float resize + merge per image
int launch_dense_cycle(const float* pred, const double* aff, float* maps, int n_pairs, hipStream_t s);
int launch_dense_merge(const float* maps, const int32_t* boxes, int n_pairs, int side, int H, int W, float* flow,
                       float* conf, hipStream_t s);

int launch_resize_f32(const float* src, int Hs, int Ws, int C, float* dst, int Hd, int Wd, hipStream_t s);

// fused feed-forward block (ffn.hip) and its reduce + bias + residual + LayerNorm tail (pointwise.hip)
int ffn_fused_chunks(int M);
int launch_ffn_fused(const float* X, const float* W1, const float* b1, const float* W2, float* P, int M, int nch, hipStream_t s);
int launch_ln_reduce_post(const float* parts, int np, const float* bias, const float* residual, const float* w, const float* b,
                          const float* post_w, const float* post_b, float* y, int rows, hipStream_t s);
int launch_ln_reduce(const float* parts, int np, const float* bias, const float* residual, const float* w, const float* b,
                     float* y, int rows, hipStream_t s);
// the FFN block for many rows in one launch (ffn_rows.hip): Y = [LN_post] LN(X + W2 relu(W1 X + b1) + b2); Y != X
int launch_ffn_rows(const float* X, const float* W1, const float* b1, const float* W2, const float* b2, const float* ln_w,
                    const float* ln_b, const float* post_w, const float* post_b, float* Y, int M, hipStream_t s);
// the attention sub-layer for many rows in one launch (att_rows.hip): Y = LN(residual + out_proj(MHA(q, K, V)) + bo); q given
// (wq == nullptr) or projected from x (+ x2)
int launch_att_rows(const float* q, int ldq, const float* x, const float* x2, const float* wq, const float* bq, float qscale,
                    const float* k, const float* v, int ldkv, const float* wo, const float* bo, const float* residual,
                    const float* ln_w, const float* ln_b, float* Y, int nb, int nq, hipStream_t s);
// layer1 block 0's downsample + conv1 over the same x [M][64] in one launch for many pairs (expand.hip)
int launch_expand(const float* x, int M, const float* w0, const float* s0, const float* b0, int relu0, float* y0, int n0, const float* w1,
                  const float* s1, const float* b1, int relu1, float* y1, int n1, hipStream_t s);
// the same for a layer2 bottleneck (conv23m.hip): t1 [B][32 S][64 S][128] -> y [B][32][64][512], stride S = 1 / 2
int launch_conv23m(const float* t1, const float* w2, const float* s2, const float* b2, const float* w3, const float* s3, const float* b3,
                   const float* residual, float* y, int B, int stride, hipStream_t s);
// conv2 (3x3) -> conv3 (1x1) of a layer1 bottleneck in one launch for many pairs (conv23.hip): t1 [B][64][128][64] -> y [B][64][128][256]
int launch_conv23(const float* t1, const float* w2, const float* s2, const float* b2, const float* w3, const float* s3, const float* b3,
                  const float* residual, float* y, int B, hipStream_t s);
// one whole layer1 bottleneck in one launch (bottleneck.hip); w2p / w3p / wdp are the packed fragment arrays (bottleneck_pack_*)
int launch_bottleneck(const float* x, float* y, int B, int cin, const float* w1, const float* w2p, const float* w3p, const float* wdp,
                      const float* s1, const float* b1, const float* s2, const float* b2, const float* s3, const float* b3,
                      const float* sd, const float* bd, hipStream_t s);
void bottleneck_pack_w2(const float* w2 /*[64][576]*/, float* w2p /*[36864]*/);
void bottleneck_pack_w3(const float* w3 /*[256][64]*/, float* w3p /*[16384]*/);
#endif   // COTR_EXPERIMENTAL
