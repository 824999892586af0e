// libcotr_hip.so - C ABI (include/cotr_hip.h), weight packing and the launch schedule of the COTR
// correspondence-query forward path on one MI355X.
//
// Path and reference call sites (relative to /root/reference):
//   encode: COTR/models/backbone.py:79-92 (two halves -> torchvision resnet50 to layer3, FrozenBN :46-56)
//           COTR/models/cotr_model.py:37 (input_proj) ; position_encoding.py:60-72 (image grid encoding)
//           COTR/models/transformer.py:143-159 x6 (encoder) ; :192-195 (decoder K/V projections, hoisted:
//           they depend on the memory only, not on the queries)
//   decode: cotr_model.py:34-36 (query encoding) ; transformer.py:185-201 x6 ; :110-111 (decoder.norm)
//           position_encoding.py:23-26 (corr_embed) - on the last layer only, the reference keeps [-1]
//
// HBM layout: activations are NHWC "side-by-side" [B, H, 2W, C] - both 256x256 halves of a pair share
// one tensor (the reference concatenates them on W only after layer3, backbone.py:85), convolutions
// pad each half on its own; after layer3 the tensor IS the [B*512, 1024] token matrix.  Tokens are
// batch-major [B*512, 256] (the reference is sequence-first [512,B,256]; internal only).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include <limits.h>

#include "../../include/cotr_hip.h"
#include "common.h"
#include "train.h"

int init_attention_attributes();
void set_ffn_debug_times(unsigned long long* p);  // ffn.hip
void set_attention_debug_times(unsigned long long* p);  // attention.hip

namespace {

constexpr int D = 256, FFN = 1024, TOK = 512, CFEAT = 1024;
constexpr int ENC_CHUNK_MAX = 128; // largest settable encode chunk (scratch ~66 MB per pair of a pass: 8.4 GB at 128 - of 288 GB)
constexpr int DEC_ROWS = 32768;   // query rows per decoder pass (scratch ~9 KB per row)
constexpr float QSCALE = 0.17677669529663687f;  // 32^-0.5, float(head_dim) ** -0.5 in torch

struct ConvW {
  const float *w, *scale, *bias;
  int cin, cout, k, stride;
};
struct EncW {
  const float *in_w, *in_b, *out_w, *out_b, *l1w, *l1b, *l2w, *l2b, *n1w, *n1b, *n2w, *n2b;
};
struct DecW {
  const float *q_w, *q_b, *out_w, *out_b, *l1w, *l1b, *l2w, *l2b, *n2w, *n2b, *n3w, *n3b;
};
struct StageSpec {
  int planes, blocks, stride;
};
const StageSpec kStages[3] = {{64, 3, 1}, {128, 4, 2}, {256, 6, 2}};

struct Arena {
  float* ptr = nullptr;
  size_t cap = 0;  // floats
  bool external = false;  // carved from the caller's workspace (cotr_set_workspace): never freed here
};

thread_local std::string g_create_error;

// ---- tuning knobs (KnobId in common.h): name, shipped default, accepted range; a few have a value set instead of a range ----
struct KnobDesc {
  int id;            // its KnobId: the table below must list the knobs in the enum's order (checked at compile time)
  const char* name;
  int def, lo, hi;
};
constexpr KnobDesc kKnobs[] = {
    {KN_ENCODE_CHUNK, "encode_chunk", 64, 1, ENC_CHUNK_MAX},
    {KN_ATTENTION_FUSION_MAX_ROWS, "attention_fusion_max_rows", 4096, 0, INT_MAX},
    {KN_FFN_FUSION_MAX_ROWS, "ffn_fusion_max_rows", 4096, 0, INT_MAX},
    {KN_KS3, "ks3", 1, 0, 1},
    {KN_DUAL_CONV, "dual_conv", 1, 0, 1},
    {KN_FUSED_STEM, "fused_stem", 1, 0, 1},
    {KN_XCD_MAPPING, "xcd_mapping", 1, 0, 63},
    {KN_ATTENTION_FUSED_SPLITS, "attention_fused_splits", 0, 0, 84},
    {KN_CONV_PATCH, "conv_patch", 1, 0, 1},
    {KN_POS_TABLE_MIN_ROWS, "pos_table_min_rows", 8192, 0, INT_MAX},
    {KN_ATTENTION_WIDE_OCCUPANCY, "attention_wide_occupancy", 3, 2, 3},
    {KN_ATTENTION_WIDE_MIN_ROWS, "attention_wide_min_rows", 4096, 0, INT_MAX},
    {KN_ATTENTION_SPLITS, "attention_splits", 0, 0, 16},
    {KN_CONV1X1_DENSE, "conv1x1_dense", 1, 0, 1},
    {KN_WS_FLAGS, "ws_flags", 2, 0, 3},
    {KN_BOTTLENECK_MAX_PAIRS, "bottleneck_max_pairs", 5, 0, INT_MAX},
    {KN_TRAIN_ATTENTION_FORM, "train_attention_form", 0, 0, 3},
    {KN_ATTENTION_RESIDENT, "attention_resident", 1, 0, 1},
    {KN_ATT_ROWS_MIN_ROWS, "att_rows_min_rows", 8192, 0, INT_MAX},
    {KN_FFN_ROWS_MIN_ROWS, "ffn_rows_min_rows", 8192, 0, INT_MAX},
    {KN_CONV23_MIN_PAIRS, "conv23_min_pairs", 5, 1, INT_MAX},
    {KN_CONV23M_MIN_PAIRS, "conv23m_min_pairs", 16, 1, INT_MAX},
    {KN_EXPAND_MIN_ROWS, "expand_min_rows", 65536, 0, INT_MAX},
    {KN_ROWS_MIN_FILL, "rows_min_fill", 75, 0, 100},
    {KN_SIDE_STREAM, "side_stream", 0, 0, 3},
    {KN_FFN_FUSED_MAX_CHUNKS, "ffn_fused_max_chunks", 16, 2, 16},
    {KN_BATCH_SPLIT, "batch_split", 1, 0, 1},
};
constexpr bool knobs_in_enum_order() {
  for (int i = 0; i < (int)(sizeof(kKnobs) / sizeof(kKnobs[0])); ++i)
    if (kKnobs[i].id != i) return false;
  return true;
}
static_assert(sizeof(kKnobs) / sizeof(kKnobs[0]) == KN_COUNT && knobs_in_enum_order(), "kKnobs must follow enum KnobId entry by entry");
bool knob_value_ok(int id, int v) {
  if (v < kKnobs[id].lo || v > kKnobs[id].hi) return false;
  if (id == KN_XCD_MAPPING) return (v & 3) != 3;
  if (id == KN_ATTENTION_FUSED_SPLITS) return v == 0 || v == 4 || v == 8 || v == 48 || v == 84;
  if (id == KN_ATTENTION_SPLITS) return v == 0 || v == 1 || v == 2 || v == 4 || v == 8 || v == 16;
  if (id == KN_FFN_FUSED_MAX_CHUNKS) return v == 2 || v == 4 || v == 8 || v == 16;
  return true;
}
KnobSet default_knobs() {
  KnobSet k;
  for (int i = 0; i < KN_COUNT; ++i) k.v[i] = kKnobs[i].def;
  return k;
}
// the set of the handle-less op-level entry points (cotr_op_*, cotr_bench_*, cotr_train_*); cotr_set_knob(NULL, ...)
KnobSet g_process_knobs = default_knobs();
}  // namespace
thread_local const KnobSet* cotr_tls_knobs = &g_process_knobs;

struct cotr_ctx {
  int device = 0;
  std::string err;
  KnobSet knobs = default_knobs();   // this handle's tuning knobs (cotr_set_knob)
  // weights
  float* wbuf = nullptr;
  size_t wfloats = 0;
  bool loaded = false;
  std::vector<ConvW> convs;  // execution order: stem, then per block conv1, conv2, conv3, [downsample]
  // packed MFMA-fragment images of layer1's conv2 / conv3 / downsample weights for the fused bottleneck kernel (bottleneck.hip)
  struct FusedBlock { const float *w2p = nullptr, *w3p = nullptr, *wdp = nullptr; };
  FusedBlock l1_fused[3];
  const float *ip_w = nullptr, *ip_b = nullptr;
  std::vector<EncW> enc;
  std::vector<DecW> dec;
  const float *kv_w = nullptr, *kv_b = nullptr;  // [L*512][256], [L*512]
  const float *dn_w = nullptr, *dn_b = nullptr;
  const float* mlp_w[3] = {nullptr, nullptr, nullptr};
  const float* mlp_b[3] = {nullptr, nullptr, nullptr};
  float* pos = nullptr;  // [512][256]
  // pos . W^T of the encoder in-projections [L][512][768] (q|k columns, q pre-scaled; v columns zero) and of the hoisted decoder
  // K/V projection [512][L*512] (k columns; v zero): (x + pos) . W^T = x . W^T + pos . W^T, and the second term depends on the
  // weights only - with many rows it enters as a row-periodic residual of a plain GEMM on the LDS-DMA large-tile kernel instead
  // of an x + pos prologue that only the register-staged kernels have (transformer.py:147-153, 192-195)
  float* tab_qkv = nullptr;
  float* tab_kv = nullptr;
  // cached encode
  Arena memkv;  // memory [B*512*256] then kv [B*512*L*512]
  int enc_B = 0;
  // scratch
  Arena enc_scr, dec_scr;
  // taps: name -> (pointer, floats).  Persistent buffers (memory, kv, pos) are referenced in place;
  // scratch intermediates are only kept when debug taps are on (copied aside as they are produced)
  std::map<std::string, std::pair<const float*, size_t>> taps;
  bool keep_taps = false;
  std::map<std::string, Arena> tap_store;
  // caller-supplied scratch (cotr_set_workspace): the three arenas are carved from it instead of hipMalloc'ed
  char* ws = nullptr;
  size_t ws_bytes = 0, ws_used = 0;
  // second stream of the handle (knob side_stream): work that depends on the queries only / on the memory only runs beside the chain
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_q = nullptr, ev_mem = nullptr, ev_kv = nullptr;
  int side_mode = 0;   // the knob's bits for the cotr_forward call in flight (0 outside one, or where its conditions do not hold)
  // profiling
  int prof = 0;  // 0 off, 1 per stage, 2 per kernel launch
  std::vector<std::string> prof_names;
  std::vector<hipEvent_t> prof_ev;
};

namespace {

#define HIPCHK(h, expr)                                                                  \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                      \
      return COTR_ERR_HIP;                                                               \
    }                                                                                    \
  } while (0)

#define KCHK(h, expr, what)                                                              \
  do {                                                                                   \
    int r_ = (expr);                                                                     \
    if (r_ != 0) {                                                                       \
      (h)->err = std::string("launch failed: ") + (what) + (r_ == -1 ? " (bad shape)" : " (hip error)"); \
      return r_ == -1 ? COTR_ERR_ARG : COTR_ERR_HIP;                                     \
    }                                                                                    \
  } while (0)

// Makes the handle's device current for the duration of an ABI call and puts the caller's device back afterwards (a
// destroy running from a Python finaliser must not change the thread's current device behind the caller's back).
struct DeviceScope {
  int prev = -1, prev_tls = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceScope(int dev) {
    prev_tls = cotr_tls_device;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) {
      err = hipSetDevice(dev);
      switched = err == hipSuccess;
    }
    if (err == hipSuccess) cotr_tls_device = dev;
  }
  ~DeviceScope() {
    cotr_tls_device = prev_tls;
    if (switched && prev >= 0) (void)hipSetDevice(prev);
  }
};
// ... and the handle's knob set, for the launch helpers of every translation unit (knob(), common.h)
struct KnobScope {
  const KnobSet* prev;
  explicit KnobScope(const KnobSet* k) : prev(cotr_tls_knobs) { cotr_tls_knobs = k; }
  ~KnobScope() { cotr_tls_knobs = prev; }
};
#define DEVICE_SCOPE(h)                                                                  \
  DeviceScope dev_scope_((h)->device);                                                   \
  KnobScope knob_scope_(&(h)->knobs);                                                    \
  do {                                                                                   \
    if (dev_scope_.err != hipSuccess) {                                                  \
      (h)->err = std::string("hipSetDevice: ") + hipGetErrorString(dev_scope_.err);      \
      return COTR_ERR_HIP;                                                               \
    }                                                                                    \
  } while (0)

int ensure(cotr_ctx* h, Arena& a, size_t floats) {
  if (a.cap >= floats) return COTR_OK;
  if (h->ws != nullptr && (&a == &h->memkv || &a == &h->enc_scr || &a == &h->dec_scr)) {
    // caller-supplied workspace: the three regions lie in the fixed order [encode cache | encoder scratch | decoder scratch],
    // 256-byte aligned; nothing is freed or allocated on the device (no synchronisation in the middle of a stream).  A region
    // that has to grow grows in place and the regions behind it are re-carved at their next use (they hold scratch only: the
    // encode cache is the first region and only grows inside encode, which rewrites it) - so a workspace sized with
    // cotr_scratch_bytes for the largest (B, Q) serves every smaller shape in any order.
    Arena* order[3] = {&h->memkv, &h->enc_scr, &h->dec_scr};
    size_t off = 0;
    int idx = 0;
    for (; order[idx] != &a; ++idx)
      if (order[idx]->ptr) off = (size_t)(reinterpret_cast<char*>(order[idx]->ptr) - h->ws) + order[idx]->cap * sizeof(float);
    off = (off + 255) & ~size_t(255);
    if (off + floats * sizeof(float) > h->ws_bytes) {
      char msg[160];
      snprintf(msg, sizeof msg, "workspace too small: %zu bytes given, %zu needed so far (size it with cotr_scratch_bytes)",
               h->ws_bytes, off + floats * sizeof(float));
      h->err = msg;
      return COTR_ERR_ARG;
    }
    a.ptr = reinterpret_cast<float*>(h->ws + off);
    a.cap = floats;
    a.external = true;
    for (int j = idx + 1; j < 3; ++j) *order[j] = Arena();
    h->ws_used = off + floats * sizeof(float);
    return COTR_OK;
  }
  if (a.ptr && !a.external) HIPCHK(h, hipFree(a.ptr));
  a.ptr = nullptr;
  a.cap = 0;
  a.external = false;
  HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&a.ptr), floats * sizeof(float)));
  a.cap = floats;
  return COTR_OK;
}

void prof_mark(cotr_ctx* h, const char* name, hipStream_t s, int level = 1) {
  if (h->prof < level || (h->prof >= 2 && level == 1 && strcmp(name, "begin") != 0 && strcmp(name, "dec_begin") != 0)) return;
  hipEvent_t ev;
  if (hipEventCreate(&ev) != hipSuccess) return;
  (void)hipEventRecord(ev, s);
  h->prof_names.push_back(name);
  h->prof_ev.push_back(ev);
}

void prof_reset(cotr_ctx* h) {
  for (auto ev : h->prof_ev) (void)hipEventDestroy(ev);
  h->prof_ev.clear();
  h->prof_names.clear();
}

// remember an intermediate for cotr_debug_tap: scratch is recycled by later stages, so it is copied
int tap_save(cotr_ctx* h, const char* name, const float* src, size_t n, hipStream_t s) {
  if (!h->keep_taps) return COTR_OK;
  Arena& a = h->tap_store[name];
  int r = ensure(h, a, n);
  if (r) return r;
  HIPCHK(h, hipMemcpyAsync(a.ptr, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
  h->taps[name] = {a.ptr, n};
  return COTR_OK;
}

GemmParams base_params() {
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.colscale = 1.f;
  p.a2_period = 1;
  return p;
}


// y[M,N] = epi( (x (+x2)) . w^T )
int linear(cotr_ctx* h, const float* x, const float* x2, int x2_row_mod, int a2_period, int a2_width,
           const float* w, const float* bias, const float* residual, int relu, float colscale, int colscale_n,
           float* y, int M, int N, int K, hipStream_t s, int ldc = 0, int res_row_mod = 0, int ldr = 0) {
  GemmParams p = base_params();
  p.M = M; p.N = N; p.K = K;
  p.A = x; p.lda = K;
  p.A2 = x2; p.lda2 = K; p.a2_row_mod = x2_row_mod; p.a2_period = a2_period; p.a2_width = a2_width;
  p.W = w; p.C = y; p.ldc = ldc ? ldc : N;
  p.bias = bias; p.residual = residual; p.ldr = ldr ? ldr : N; p.res_row_mod = res_row_mod; p.relu = relu;
  p.colscale = colscale; p.colscale_n = colscale_n;
  KCHK(h, launch_gemm(GEMM_DENSE, p, s), "linear");
  if (h->prof >= 2) { char nm[96]; snprintf(nm, sizeof nm, "linear %dx%dx%d cfg%d", M, N, K, gemm_pick_config(GEMM_DENSE, p)); prof_mark(h, nm, s, 2); }
  return COTR_OK;
}

int layernorm(cotr_ctx* h, const float* x, const float* w, const float* b, float* y, int M, hipStream_t s) {
  KCHK(h, launch_layernorm(x, w, b, y, M, s), "layernorm");
  prof_mark(h, "layernorm", s, 2);
  return COTR_OK;
}


// The small-row fused kernels (attention + out-projection partials, fused FFN: attention.hip / ffn.hip, each followed by ln_reduce):
// always up to 1024 rows - the one-pair regime they were built for; above, up to their knob (4096 rows), only where their grid of
// 32-row tiles x 8 heads / x hidden chunks fills whole rounds of the 256 CUs to >= 90 % (2048 and 4096 encoder rows = 4 and 8 pairs,
// 2000 and 4000 query rows: -2.5 ... -4 % per forward there; 1536 / 3072 rows = 1.5 rounds: +2 ... +4 %, left to the unfused
// launches; profiles/r6_frac_by_batch_sweep.txt)
bool fused_fill_ok(long wgs) {
  const long cus = cotr_num_cus(), rounds = (wgs + cus - 1) / cus;
  return wgs * 10 >= rounds * cus * 9;
}
bool att_fused_applies(long rows) {
  if (rows > knob(KN_ATTENTION_FUSION_MAX_ROWS)) return false;
  return rows <= 1024 || fused_fill_ok((rows + 31) / 32 * 8);
}
bool ffn_fused_applies(long rows) {
  if (rows > knob(KN_FFN_FUSION_MAX_ROWS)) return false;
  return rows <= 1024 || fused_fill_ok((rows + 31) / 32 * ffn_fused_chunks((int)rows));
}

// The FFN block as ONE launch (ffn_rows.hip): from knob ffn_rows_min_rows rows on, where its 64-row tiles - one workgroup per CU,
// 256 CUs - fill their last round of the chip to at least knob rows_min_fill percent (75: 500 tiles of 32 x 1000 rows: 0.98; 313
// tiles of 20 000 rows: 0.61 -> the three launches, whose tiles are finer)
bool ffn_rows_applies(int M) {
  if (M < knob(KN_FFN_ROWS_MIN_ROWS) || ffn_fused_applies(M)) return false;
  const long cus = cotr_num_cus(), tiles = (M + 63) / 64, rounds = (tiles + cus - 1) / cus;
  return tiles * 100 >= rounds * cus * knob(KN_ROWS_MIN_FILL);
}

// The attention sub-layer as ONE launch (att_rows.hip): from knob att_rows_min_rows query rows on, where its 64-query tiles (per
// pair) fill the last round of the 256 CUs to at least rows_min_fill percent and a pair's last tile is not mostly padding
bool att_rows_applies(int nb, int nq) {
  const long R = (long)nb * nq;
  if (R < knob(KN_ATT_ROWS_MIN_ROWS)) return false;
  const long cus = cotr_num_cus(), tpp = (nq + 63) / 64, tiles = tpp * nb, rounds = (tiles + cus - 1) / cus;
  return tiles * 100 >= rounds * cus * knob(KN_ROWS_MIN_FILL) && (long)nq * 8 >= tpp * 64 * 7;
}

// conv23m (layer2's conv2 -> conv3 in one launch) runs TWO workgroups per CU, 16 per pair: where its grid is a single round that fills
// the chip unevenly - more than one workgroup per CU, fewer than 7/8 of two (17 ... 27 pairs on 256 CUs) - the CUs that hold two set the
// time of the launch (the 32-pair time) and the two launches it replaces are faster: 20 pairs -2.9 ... -4.1 % per forward, 24 pairs
// -1.0 ... -1.8 %; at 16 (one per CU), 28, 32, 40, 48, 64 pairs the one launch wins by 0.5 ... 2 % (profiles/r6_frac_by_batch_odd_pairs.txt)
bool conv23m_fill_ok(int pairs) {
  const long cus = cotr_num_cus(), wgs = 16L * pairs;
  return wgs <= cus || wgs * 8 >= 2 * cus * 7;
}

// ---- how a batch is cut into passes (knob batch_split) -------------------------------------------------------------------------
// The forward's time against the pair count is a staircase (profiles/r6_frac_by_batch_every_pair_count.txt: 16 pairs 4.16 ms, 17 pairs
// 5.82; 32 pairs 7.23, 33 pairs 10.63 - tiles quantise to rounds of the 256 CUs, and the one-launch rows kernels need their last round
// filled): a batch just above a step runs faster as the step + a small remainder.
//  * encode (query-independent): kEncFirst[n] = the first pass of the cheapest partition of n pairs, from MEASURED encode times of
//    1 ... 64 pairs (tools/batch_cost.py: dynamic programme over the table, a split must win at least 2 %; enc_split.inc).
//  * decode: where the rows kernels do not take the whole pass but take a prefix of it, that prefix is a pass of its own (17 x 1000
//    queries: 16 pairs on att_rows / ffn_rows + 1 pair on the small-row fused kernels: 1.7 against 2.7 ms) - from the dispatch
//    predicates, so it follows the knobs and the query count.
// Pairs are independent (no cross-pair arithmetic anywhere): any partition computes every pair exactly as a call on that pass alone would.
#include "enc_split.inc"
int enc_next_chunk(int remaining, int cap) {
  const int n = remaining < cap ? remaining : cap;
  if (!knob(KN_BATCH_SPLIT) || n > 64) return n;
  const int c = kEncFirst[n];
  return (c >= 1 && c <= n) ? c : n;
}
int dec_next_pairs(int remaining, int cap, int nq) {
  const int n = remaining < cap ? remaining : cap;
  if (!knob(KN_BATCH_SPLIT) || n < 2) return n;
  if (att_rows_applies(n, nq) && ffn_rows_applies(n * nq)) return n;
  const long min_rows = knob(KN_ATT_ROWS_MIN_ROWS) > knob(KN_FFN_ROWS_MIN_ROWS) ? knob(KN_ATT_ROWS_MIN_ROWS) : knob(KN_FFN_ROWS_MIN_ROWS);
  for (int k = n - 1; k >= 1 && (long)k * nq >= min_rows; --k)   // (below either threshold no prefix can take both rows kernels)
    if (att_rows_applies(k, nq) && ffn_rows_applies(k * nq)) return k;
  return n;
}

// y = LayerNorm(x + linear2(relu(linear1(x))))  (transformer.py:156-158 / 199-201; x is already normalised).
// Where ffn_fused_applies: ONE fused launch that keeps the hidden activations on the CU and writes per-chunk
// partial outputs + ln_reduce (sum, bias, residual, norm [, a second norm: decoder.norm after the last layer]); above: linear1,
// linear2 (+residual), layernorm.  `hid` holds hid_cap >= M*1024 floats (the fused form needs chunks*M*256), `tmp` M*256.
// post_w only where the caller knows the fused form (or ffn_rows) applies.
int ffn_block(cotr_ctx* h, const float* x, const float* l1w, const float* l1b, const float* l2w, const float* l2b,
              const float* nw, const float* nb, float* hid, size_t hid_cap, float* tmp, float* y, int M, hipStream_t s,
              const float* post_w = nullptr, const float* post_b = nullptr) {
  if (ffn_rows_applies(M) && y != x) {
    // many rows: the whole block - linear1, ReLU, linear2, bias, residual, norm [, decoder.norm] - in one launch (ffn_rows.hip)
    KCHK(h, launch_ffn_rows(x, l1w, l1b, l2w, l2b, nw, nb, post_w, post_b, y, M, s), "ffn_rows");
    if (h->prof >= 2) { char nm[64]; snprintf(nm, sizeof nm, "ffn_rows %d rows", M); prof_mark(h, nm, s, 2); }
    return COTR_OK;
  }
  if (post_w != nullptr || (ffn_fused_applies(M) && (size_t)ffn_fused_chunks(M) * M * D <= hid_cap)) {
    const int nch = ffn_fused_chunks(M);
    if ((size_t)nch * M * D > hid_cap) { h->err = "ffn_block: partial-output scratch too small"; return COTR_ERR_STATE; }
    KCHK(h, launch_ffn_fused(x, l1w, l1b, l2w, hid, M, nch, s), "ffn_fused");
    if (h->prof >= 2) { char nm[64]; snprintf(nm, sizeof nm, "ffn_fused %d rows x%d", M, nch); prof_mark(h, nm, s, 2); }
    KCHK(h, launch_ln_reduce_post(hid, nch, l2b, x, nw, nb, post_w, post_b, y, M, s), "ln_reduce");
    prof_mark(h, post_w ? "ln_reduce +norm" : "ln_reduce", s, 2);
    return COTR_OK;
  }
  int r;
  if ((r = linear(h, x, nullptr, 0, 1, 0, l1w, l1b, nullptr, 1, 1.f, 0, hid, M, FFN, D, s))) return r;
  if ((r = linear(h, hid, nullptr, 0, 1, 0, l2w, l2b, x, 0, 1.f, 0, tmp, M, D, FFN, s))) return r;
  return layernorm(h, tmp, nw, nb, y, M, s);
}

GemmParams conv_params(const ConvW& c, const float* x, const float* residual, int relu, float* y, int B, int Hin, int Win) {
  GemmParams p = base_params();
  const int pad = c.k / 2;
  p.Hin = Hin; p.Win = Win; p.Cin = c.cin;
  p.Hout = (Hin + 2 * pad - c.k) / c.stride + 1;
  p.Wout = (Win + 2 * pad - c.k) / c.stride + 1;
  p.ksize = c.k; p.stride = c.stride; p.pad = pad;
  p.M = B * p.Hout * 2 * p.Wout; p.N = c.cout; p.K = c.k * c.k * c.cin;
  p.A = x; p.lda = c.cin;
  p.W = c.w; p.C = y; p.ldc = c.cout;
  p.scale = c.scale; p.bias = c.bias; p.residual = residual; p.ldr = c.cout; p.relu = relu;
  return p;
}

// Entry block of a ResNet stage: the downsample branch and conv1 both read the block input (torchvision Bottleneck.forward)
// and are independent - in the latency-bound regime (a pair or two per pass) they go out as ONE launch whose grid is the
// tiles of both problems.  Returns 1 if it launched them, 0 if the caller should launch them one by one, < 0 on error.
int conv_pair(cotr_ctx* h, const ConvW& cd, const ConvW& c1, const float* x, float* yd, float* y1, int B, int Hin, int Win,
              hipStream_t s) {
  if (!knob(KN_DUAL_CONV)) return 0;
  const GemmParams pd = conv_params(cd, x, nullptr, 0, yd, B, Hin, Win), p1 = conv_params(c1, x, nullptr, 1, y1, B, Hin, Win);
  if (p1.M > 16384) return 0;   // batched: throughput-bound, each problem keeps its own best configuration
  const int cfd = gemm_pick_config(GEMM_CONV, pd), cf1 = gemm_pick_config(GEMM_CONV, p1);
  const bool d_larger = (double)pd.M * pd.N >= (double)p1.M * p1.N;
  const int order[2] = {d_larger ? cfd : cf1, d_larger ? cf1 : cfd};   // the larger problem's configuration first
  for (int cfg : order) {
    if (cfg < 0 || !gemm_cfg_supports_dual(cfg)) continue;
    const int r = launch_gemm_dual_cfg(GEMM_CONV, cfg, pd, p1, s);
    if (r == -1) continue;                                              // does not fit one of the two shapes
    if (r != 0) { h->err = "launch failed: dual conv (hip error)"; return COTR_ERR_HIP; }
    if (h->prof >= 2) {
      char nm[128];
      snprintf(nm, sizeof nm, "conv1x1/%d+conv1x1/1 %dx%dx%d+%dx%dx%d cfg%d", cd.stride, pd.M, pd.N, pd.K, p1.M, p1.N, p1.K, cfg);
      prof_mark(h, nm, s, 2);
    }
    return 1;
  }
  return 0;
}

int conv(cotr_ctx* h, const ConvW& c, const float* x, const float* residual, int relu, float* y, int B,
         int Hin, int Win, hipStream_t s) {
  const GemmParams p = conv_params(c, x, residual, relu, y, B, Hin, Win);
  KCHK(h, launch_gemm(GEMM_CONV, p, s), "conv");
  if (h->prof >= 2) { char nm[96]; snprintf(nm, sizeof nm, "conv%dx%d/%d %dx%dx%d cfg%d", c.k, c.k, c.stride, p.M, p.N, p.K, gemm_pick_config(GEMM_CONV, p)); prof_mark(h, nm, s, 2); }
  return COTR_OK;
}

int stem(cotr_ctx* h, const ConvW& c, const float* img, float* y, int B, hipStream_t s) {
  GemmParams p = base_params();
  p.M = B * 128 * 256; p.N = 64; p.K = 160;
  p.A = img; p.W = c.w; p.C = y; p.ldc = 64;
  p.scale = c.scale; p.bias = c.bias; p.relu = 1;
  KCHK(h, launch_gemm(GEMM_STEM, p, s), "stem");
  if (h->prof >= 2) prof_mark(h, "stem conv7x7", s, 2);
  return COTR_OK;
}

}  // namespace

// ================================================================================================
extern "C" {

int cotr_abi_version(void) { return COTR_HIP_ABI_VERSION; }

const char* cotr_last_error(cotr_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int cotr_create(cotr_handle* out, int device) {
  if (!out) return COTR_ERR_ARG;
  *out = nullptr;
  if (device < 0 || device >= COTR_MAX_DEVICES) {
    g_create_error = "cotr_create: device index out of range";
    return COTR_ERR_ARG;
  }
  DeviceScope scope(device);
  hipError_t e = scope.err;
  if (e != hipSuccess) {
    g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
    return COTR_ERR_HIP;
  }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    g_create_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
    return COTR_ERR_HIP;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_create_error = std::string("libcotr_hip is built for gfx950 (MI355X) only; device is ") + prop.gcnArchName;
    return COTR_ERR_HIP;
  }
  if (init_attention_attributes() != 0) {
    g_create_error = "hipFuncSetAttribute(attention, 139 KB dynamic LDS) failed";
    return COTR_ERR_HIP;
  }
  cotr_ctx* h = new cotr_ctx();
  h->device = device;
  if (hipMalloc(reinterpret_cast<void**>(&h->pos), (size_t)TOK * D * sizeof(float)) != hipSuccess ||
      launch_pos_table(h->pos, nullptr) != 0 || hipStreamSynchronize(nullptr) != hipSuccess) {
    g_create_error = "building the image position table failed";
    delete h;
    return COTR_ERR_HIP;
  }
  *out = h;
  return COTR_OK;
}

void cotr_destroy(cotr_handle h) {
  if (!h) return;
  DeviceScope scope(h->device);
  (void)hipDeviceSynchronize();
  prof_reset(h);
  if (h->wbuf) (void)hipFree(h->wbuf);
  if (h->pos) (void)hipFree(h->pos);
  for (hipEvent_t ev : {h->ev_fork, h->ev_q, h->ev_mem, h->ev_kv})
    if (ev) (void)hipEventDestroy(ev);
  if (h->side) (void)hipStreamDestroy(h->side);
  for (Arena* a : {&h->memkv, &h->enc_scr, &h->dec_scr})
    if (a->ptr && !a->external) (void)hipFree(a->ptr);
  for (auto& kv : h->tap_store)
    if (kv.second.ptr) (void)hipFree(kv.second.ptr);
  delete h;
}

int cotr_load_weights(cotr_handle h, const char* const* names, const float* const* ptrs,
                      const int64_t* numels, int n) {
  if (!h || !names || !ptrs || !numels || n <= 0) return COTR_ERR_ARG;
  DEVICE_SCOPE(h);
  std::map<std::string, int> idx;
  for (int i = 0; i < n; ++i) idx[names[i]] = i;

  int n_enc = 0, n_dec = 0;
  while (idx.count("transformer.encoder.layers." + std::to_string(n_enc) + ".linear1.weight")) ++n_enc;
  while (idx.count("transformer.decoder.layers." + std::to_string(n_dec) + ".linear1.weight")) ++n_dec;
  if (n_enc == 0 || n_dec == 0) {
    h->err = "state dict has no transformer.encoder/decoder layers";
    return COTR_ERR_WEIGHTS;
  }

  std::vector<float> host;  // packed image of every tensor, 64-float aligned
  host.reserve(19u << 20);
  std::string missing;
  std::vector<float> tmp;
  auto fetch = [&](const std::string& name, size_t expect) -> const float* {
    auto it = idx.find(name);
    if (it == idx.end() || (size_t)numels[it->second] != expect) {
      if (missing.empty()) missing = name + (it == idx.end() ? " (missing)" : " (wrong size)");
      tmp.assign(expect, 0.f);
      return tmp.data();
    }
    tmp.resize(expect);
    if (hipMemcpy(tmp.data(), ptrs[it->second], expect * sizeof(float), hipMemcpyDefault) != hipSuccess) {
      if (missing.empty()) missing = name + " (hipMemcpy failed)";
      tmp.assign(expect, 0.f);
    }
    return tmp.data();
  };
  auto reserve = [&](size_t nfl) -> size_t {
    size_t off = (host.size() + 63) & ~size_t(63);
    host.resize(off + nfl, 0.f);
    return off;
  };
  auto put = [&](const std::string& name, size_t nfl) -> size_t {
    const float* src = fetch(name, nfl);
    size_t off = reserve(nfl);
    memcpy(&host[off], src, nfl * sizeof(float));
    return off;
  };

  struct ConvOff {
    size_t w, scale, bias;
    int cin, cout, k, stride;
  };
  std::vector<ConvOff> convs;
  auto put_conv = [&](const std::string& conv, const std::string& bn, int cout, int cin, int k, int stride) {
    ConvOff c;
    c.cin = cin; c.cout = cout; c.k = k; c.stride = stride;
    const std::string p = "backbone.0.body.";
    if (cin == 3) {  // stem: [64][3][7][7] -> [64][160], k = c*49 + ky*7 + kx, zero padded
      const float* src = fetch(p + conv + ".weight", (size_t)cout * 147);
      c.w = reserve((size_t)cout * 160);
      for (int o = 0; o < cout; ++o) memcpy(&host[c.w + (size_t)o * 160], src + (size_t)o * 147, 147 * sizeof(float));
    } else {  // [Cout][Cin][k][k] -> [Cout][k][k][Cin]
      const float* src = fetch(p + conv + ".weight", (size_t)cout * cin * k * k);
      c.w = reserve((size_t)cout * cin * k * k);
      for (int o = 0; o < cout; ++o)
        for (int ci = 0; ci < cin; ++ci)
          for (int t = 0; t < k * k; ++t)
            host[c.w + ((size_t)o * k * k + t) * cin + ci] = src[((size_t)o * cin + ci) * k * k + t];
    }
    // FrozenBatchNorm2d.forward, COTR/models/backbone.py:46-56
    std::vector<float> bw(cout), bb(cout), rm(cout), rv(cout);
    memcpy(bw.data(), fetch(p + bn + ".weight", cout), cout * sizeof(float));
    memcpy(bb.data(), fetch(p + bn + ".bias", cout), cout * sizeof(float));
    memcpy(rm.data(), fetch(p + bn + ".running_mean", cout), cout * sizeof(float));
    memcpy(rv.data(), fetch(p + bn + ".running_var", cout), cout * sizeof(float));
    c.scale = reserve(cout);
    c.bias = reserve(cout);
    for (int o = 0; o < cout; ++o) {
      const float scale = bw[o] * (1.0f / sqrtf(rv[o] + 1e-5f));
      host[c.scale + o] = scale;
      host[c.bias + o] = bb[o] - rm[o] * scale;
    }
    convs.push_back(c);
  };
  put_conv("conv1", "bn1", 64, 3, 7, 2);
  int inplanes = 64;
  for (int st = 0; st < 3; ++st) {
    for (int b = 0; b < kStages[st].blocks; ++b) {
      const int planes = kStages[st].planes, s = (b == 0) ? kStages[st].stride : 1;
      const std::string p = "layer" + std::to_string(st + 1) + "." + std::to_string(b) + ".";
      put_conv(p + "conv1", p + "bn1", planes, inplanes, 1, 1);
      put_conv(p + "conv2", p + "bn2", planes, planes, 3, s);
      put_conv(p + "conv3", p + "bn3", planes * 4, planes, 1, 1);
      if (b == 0) put_conv(p + "downsample.0", p + "downsample.1", planes * 4, inplanes, 1, s);
      inplanes = planes * 4;
    }
  }
  // layer1 (convs[1..10]: block 0 = conv1, conv2, conv3, downsample; blocks 1, 2 = conv1, conv2, conv3): fragment images
  size_t l1_w2p[3], l1_w3p[3], l1_wdp = 0;
  for (int b = 0; b < 3; ++b) {
    const int c1i = b == 0 ? 1 : 5 + 3 * (b - 1);
    l1_w2p[b] = reserve(36864);
    l1_w3p[b] = reserve(16384);
    bottleneck_pack_w2(&host[convs[c1i + 1].w], &host[l1_w2p[b]]);
    bottleneck_pack_w3(&host[convs[c1i + 2].w], &host[l1_w3p[b]]);
    if (b == 0) {
      l1_wdp = reserve(16384);
      bottleneck_pack_w3(&host[convs[c1i + 3].w], &host[l1_wdp]);
    }
  }
  const size_t ip_w = put("input_proj.weight", (size_t)D * CFEAT), ip_b = put("input_proj.bias", D);

  struct EncOff { size_t v[12]; };
  std::vector<EncOff> enc(n_enc), dec(n_dec);
  for (int i = 0; i < n_enc; ++i) {
    const std::string p = "transformer.encoder.layers." + std::to_string(i) + ".";
    size_t* v = enc[i].v;
    v[0] = put(p + "self_attn.in_proj_weight", (size_t)3 * D * D);
    v[1] = put(p + "self_attn.in_proj_bias", 3 * D);
    v[2] = put(p + "self_attn.out_proj.weight", (size_t)D * D);
    v[3] = put(p + "self_attn.out_proj.bias", D);
    v[4] = put(p + "linear1.weight", (size_t)FFN * D);
    v[5] = put(p + "linear1.bias", FFN);
    v[6] = put(p + "linear2.weight", (size_t)D * FFN);
    v[7] = put(p + "linear2.bias", D);
    v[8] = put(p + "norm1.weight", D);
    v[9] = put(p + "norm1.bias", D);
    v[10] = put(p + "norm2.weight", D);
    v[11] = put(p + "norm2.bias", D);
  }
  // decoder: q projection per layer; K|V projections of all layers concatenated [L*512][256]
  const size_t kv_w = reserve((size_t)n_dec * 2 * D * D), kv_b = reserve((size_t)n_dec * 2 * D);
  for (int i = 0; i < n_dec; ++i) {
    const std::string p = "transformer.decoder.layers." + std::to_string(i) + ".";
    size_t* v = dec[i].v;
    {
      const float* w = fetch(p + "multihead_attn.in_proj_weight", (size_t)3 * D * D);
      v[0] = reserve((size_t)D * D);
      memcpy(&host[v[0]], w, (size_t)D * D * sizeof(float));
      memcpy(&host[kv_w + (size_t)i * 2 * D * D], w + (size_t)D * D, (size_t)2 * D * D * sizeof(float));
      const float* bq = fetch(p + "multihead_attn.in_proj_bias", 3 * D);
      v[1] = reserve(D);
      memcpy(&host[v[1]], bq, D * sizeof(float));
      memcpy(&host[kv_b + (size_t)i * 2 * D], bq + D, 2 * D * sizeof(float));
    }
    v[2] = put(p + "multihead_attn.out_proj.weight", (size_t)D * D);
    v[3] = put(p + "multihead_attn.out_proj.bias", D);
    v[4] = put(p + "linear1.weight", (size_t)FFN * D);
    v[5] = put(p + "linear1.bias", FFN);
    v[6] = put(p + "linear2.weight", (size_t)D * FFN);
    v[7] = put(p + "linear2.bias", D);
    v[8] = put(p + "norm2.weight", D);
    v[9] = put(p + "norm2.bias", D);
    v[10] = put(p + "norm3.weight", D);
    v[11] = put(p + "norm3.bias", D);
  }
  const size_t dn_w = put("transformer.decoder.norm.weight", D), dn_b = put("transformer.decoder.norm.bias", D);
  size_t mw[3], mb[3];
  for (int i = 0; i < 3; ++i) {
    const int o = (i == 2) ? 2 : D;
    mw[i] = put("corr_embed.layers." + std::to_string(i) + ".weight", (size_t)o * D);
    mb[i] = put("corr_embed.layers." + std::to_string(i) + ".bias", o);
  }
  if (!missing.empty()) {
    h->err = "state dict: " + missing;
    return COTR_ERR_WEIGHTS;
  }

  const size_t tab_qkv = reserve((size_t)n_enc * TOK * 3 * D), tab_kv = reserve((size_t)TOK * n_dec * 2 * D);  // zeros; filled below
  HIPCHK(h, hipDeviceSynchronize());
  if (h->wbuf && h->wfloats < host.size()) {
    HIPCHK(h, hipFree(h->wbuf));
    h->wbuf = nullptr;
  }
  if (!h->wbuf) {
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->wbuf), host.size() * sizeof(float)));
    h->wfloats = host.size();
  }
  HIPCHK(h, hipMemcpy(h->wbuf, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  const float* base = h->wbuf;
  h->convs.clear();
  for (const auto& c : convs) h->convs.push_back({base + c.w, base + c.scale, base + c.bias, c.cin, c.cout, c.k, c.stride});
  for (int b = 0; b < 3; ++b) {
    h->l1_fused[b].w2p = base + l1_w2p[b];
    h->l1_fused[b].w3p = base + l1_w3p[b];
    h->l1_fused[b].wdp = b == 0 ? base + l1_wdp : nullptr;
  }
  h->ip_w = base + ip_w; h->ip_b = base + ip_b;
  h->enc.clear();
  for (const auto& e : enc)
    h->enc.push_back({base + e.v[0], base + e.v[1], base + e.v[2], base + e.v[3], base + e.v[4], base + e.v[5],
                      base + e.v[6], base + e.v[7], base + e.v[8], base + e.v[9], base + e.v[10], base + e.v[11]});
  h->dec.clear();
  for (const auto& e : dec)
    h->dec.push_back({base + e.v[0], base + e.v[1], base + e.v[2], base + e.v[3], base + e.v[4], base + e.v[5],
                      base + e.v[6], base + e.v[7], base + e.v[8], base + e.v[9], base + e.v[10], base + e.v[11]});
  h->kv_w = base + kv_w; h->kv_b = base + kv_b;
  h->dn_w = base + dn_w; h->dn_b = base + dn_b;
  for (int i = 0; i < 3; ++i) { h->mlp_w[i] = base + mw[i]; h->mlp_b[i] = base + mb[i]; }
  // pos . W^T tables (see cotr_ctx): q|k rows of every encoder in_proj (q scaled as the projection's epilogue scales it), k rows of
  // every decoder layer's slice of the hoisted K/V weight; the v columns stay zero
  h->tab_qkv = h->wbuf + tab_qkv;
  h->tab_kv = h->wbuf + tab_kv;
  for (int l = 0; l < n_enc; ++l)
    if (int r = linear(h, h->pos, nullptr, 0, 1, 0, h->enc[l].in_w, nullptr, nullptr, 0, QSCALE, D,
                       h->tab_qkv + (size_t)l * TOK * 3 * D, TOK, 2 * D, D, nullptr, 3 * D)) return r;
  for (int l = 0; l < n_dec; ++l)
    if (int r = linear(h, h->pos, nullptr, 0, 1, 0, h->kv_w + (size_t)l * 2 * D * D, nullptr, nullptr, 0, 1.f, 0,
                       h->tab_kv + (size_t)l * 2 * D, TOK, D, D, nullptr, n_dec * 2 * D)) return r;
  HIPCHK(h, hipStreamSynchronize(nullptr));
  h->loaded = true;
  h->enc_B = 0;  // a cached encode belongs to the old weights
  return COTR_OK;
}

// feat_out == nullptr: the whole query-independent half (cotr_encode); otherwise only the backbone, its layer3 output
// [B,16,32,1024] (NHWC, both halves side by side = 512 token rows per pair) copied to feat_out (cotr_backbone)
static int encode_impl(cotr_handle h, const float* img, int B, cotr_stream stream, float* feat_out, int upto = 3) {
  if (!h) return COTR_ERR_ARG;
  if (!h->loaded) { h->err = "cotr_encode before cotr_load_weights"; return COTR_ERR_STATE; }
  if (!img || B <= 0) { h->err = "cotr_encode: null image or B <= 0"; return COTR_ERR_ARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  DEVICE_SCOPE(h);
  const int L = (int)h->dec.size();
  const size_t KVLD = (size_t)L * 2 * D;
  h->enc_B = 0;
  h->taps.clear();
  if (!feat_out) {
    int r = ensure(h, h->memkv, (size_t)B * TOK * (D + KVLD));
    if (r) return r;
  }
  float* memory = h->memkv.ptr;
  float* kv = h->memkv.ptr + (size_t)B * TOK * D;

  const int ENC_CHUNK = knob(KN_ENCODE_CHUNK);
  const int Bc_max = B < ENC_CHUNK ? B : ENC_CHUNK;
  // scratch carve (floats per pair)
  const size_t n_stem = (size_t)128 * 256 * 64, n_pool = (size_t)64 * 128 * 64, n_act = (size_t)64 * 128 * 256;
  const size_t n_tok = (size_t)TOK * D;
  const size_t per_pair = n_stem + n_pool + 5 * n_act + 6 * n_tok + (size_t)TOK * 3 * D + (size_t)TOK * 4 * FFN;
  // per-head partial outputs of the fused attention + out_proj launch (small-row regime only); a buffer of their own: the
  // fused FFN's partials (t_hid) are written with write-through stores right after these were read
  // (sized for the largest pass that can take that path: with batch_split a small remainder follows a large first pass)
  const size_t part_rows_cap = knob(KN_ATTENTION_FUSION_MAX_ROWS) > 1024 ? knob(KN_ATTENTION_FUSION_MAX_ROWS) : 1024;
  const size_t part_rows = (size_t)Bc_max * TOK < part_rows_cap ? (size_t)Bc_max * TOK : part_rows_cap;
  const size_t n_part = (size_t)8 * part_rows * D;
  {
    int r = ensure(h, h->enc_scr, per_pair * Bc_max + n_part);
    if (r) return r;
  }
  float* p = h->enc_scr.ptr;
  float* b_stem = p; p += n_stem * Bc_max;
  float* b_pool = p; p += n_pool * Bc_max;
  float* b_x = p; p += n_act * Bc_max;
  float* b_y = p; p += n_act * Bc_max;
  float* b_t1 = p; p += n_act * Bc_max;
  float* b_t2 = p; p += n_act * Bc_max;
  float* b_d = p; p += n_act * Bc_max;
  float* t_src = p; p += n_tok * Bc_max;
  float* t_alt = p; p += n_tok * Bc_max;
  float* t_tmp = p; p += n_tok * Bc_max;
  float* t_x1 = p; p += n_tok * Bc_max;
  float* t_pre2 = p; p += n_tok * Bc_max;
  float* t_ao = p; p += n_tok * Bc_max;
  float* t_qkv = p; p += (size_t)TOK * 3 * D * Bc_max;
  float* t_hid = p; p += (size_t)TOK * 4 * FFN * Bc_max;  // hidden activations, or up to 16 partial outputs of the fused FFN
  float* t_part = p; p += n_part;

  if (h->prof) prof_reset(h);
  prof_mark(h, "begin", s);
  for (int b0 = 0, Bc = 0; b0 < B; b0 += Bc) {
    Bc = enc_next_chunk(B - b0, ENC_CHUNK);
    const float* img_c = img + (size_t)b0 * 3 * 256 * 512;
    // ---- backbone -------------------------------------------------------------------------
    int ci = 0;
    if (knob(KN_FUSED_STEM) && !h->keep_taps) {  // conv1 + bn1 + relu + maxpool in one launch; the 'stem' tap needs the unfused pair
      const ConvW& c0 = h->convs[ci++];
      KCHK(h, launch_stem_pool(img_c, c0.w, 160, c0.scale, c0.bias, b_pool, Bc, s), "stem_pool");
      prof_mark(h, "stem_pool conv7x7+bn+relu+maxpool", s, 2);
    } else {
      { int r = stem(h, h->convs[ci++], img_c, b_stem, Bc, s); if (r) return r; }
      KCHK(h, launch_maxpool(b_stem, b_pool, Bc, 128, 128, 64, s), "maxpool");
      prof_mark(h, "maxpool", s, 2);
    }
    prof_mark(h, "stem+pool", s);
    if (int r = tap_save(h, "stem", b_stem, n_stem * Bc, s)) return r;
    if (int r = tap_save(h, "pool", b_pool, n_pool * Bc, s)) return r;
    const float* x = b_pool;
    float* outbuf[2] = {b_x, b_y};
    int flip = 0, H = 64, W = 64;
    for (int st = 0; st < 3; ++st) {
      for (int b = 0; b < kStages[st].blocks; ++b) {
        const int stride = (b == 0) ? kStages[st].stride : 1;
        const int Ho = H / stride, Wo = W / stride;
        const ConvW& c1 = h->convs[ci++];
        const ConvW& c2 = h->convs[ci++];
        const ConvW& c3 = h->convs[ci++];
        float* y = outbuf[flip];
        flip ^= 1;
        int r;
        bool one_launch = st == 0 && Bc <= knob(KN_BOTTLENECK_MAX_PAIRS) && H == 64 && W == 64;
        if (one_launch) {
          // the whole bottleneck - conv1, conv2, conv3, (downsample,) FrozenBN, identity, ReLU - in one launch (bottleneck.hip)
          const ConvW* cd = (b == 0) ? &h->convs[ci++] : nullptr;
          KCHK(h, launch_bottleneck(x, y, Bc, c1.cin, c1.w, h->l1_fused[b].w2p, h->l1_fused[b].w3p, h->l1_fused[b].wdp, c1.scale,
                                    c1.bias, c2.scale, c2.bias, c3.scale, c3.bias, cd ? cd->scale : nullptr, cd ? cd->bias : nullptr, s),
               "bottleneck");
          if (h->prof >= 2) { char nm[64]; snprintf(nm, sizeof nm, "bottleneck layer1.%d %d pairs", b, Bc); prof_mark(h, nm, s, 2); }
          x = y;
          continue;
        }
        const float* idt = x;
        bool c1_done = false;
        if (b == 0) {  // downsample branch (1x1, strided)
          const ConvW& cd = h->convs[ci++];
          if (st == 0 && cd.stride == 1 && c1.cin == 64 && Bc * H * 2 * W >= knob(KN_EXPAND_MIN_ROWS)) {
            // many pairs: downsample branch and conv1 read the pooled stem output once, in one launch (expand.hip)
            KCHK(h, launch_expand(x, Bc * H * 2 * W, cd.w, cd.scale, cd.bias, 0, b_d, cd.cout, c1.w, c1.scale, c1.bias, 1, b_t1, c1.cout, s),
                 "expand");
            if (h->prof >= 2) { char nm[64]; snprintf(nm, sizeof nm, "expand ds+conv1 layer1.0 %d pairs", Bc); prof_mark(h, nm, s, 2); }
            c1_done = true;
          } else {
            const int pr = conv_pair(h, cd, c1, x, b_d, b_t1, Bc, H, W, s);
            if (pr < 0) return pr;
            c1_done = pr == 1;
            if (!c1_done && (r = conv(h, cd, x, nullptr, 0, b_d, Bc, H, W, s))) return r;
          }
          idt = b_d;
        }
        if (!c1_done && (r = conv(h, c1, x, nullptr, 1, b_t1, Bc, H, W, s))) return r;
        if (st == 0 && H == 64 && W == 64 && Bc >= knob(KN_CONV23_MIN_PAIRS)) {
          // many pairs: conv2 -> conv3 + identity + ReLU in one launch, t2 never leaves the CU (conv23.hip)
          KCHK(h, launch_conv23(b_t1, c2.w, c2.scale, c2.bias, c3.w, c3.scale, c3.bias, idt, y, Bc, s), "conv23");
          if (h->prof >= 2) { char nm[64]; snprintf(nm, sizeof nm, "conv23 layer1.%d %d pairs", b, Bc); prof_mark(h, nm, s, 2); }
        } else if (st == 1 && Ho == 32 && Wo == 32 && Bc >= knob(KN_CONV23M_MIN_PAIRS) && conv23m_fill_ok(Bc)) {
          // many pairs: the same fusion for layer2 (conv23m.hip)
          KCHK(h, launch_conv23m(b_t1, c2.w, c2.scale, c2.bias, c3.w, c3.scale, c3.bias, idt, y, Bc, stride, s), "conv23m");
          if (h->prof >= 2) { char nm[64]; snprintf(nm, sizeof nm, "conv23m layer2.%d %d pairs", b, Bc); prof_mark(h, nm, s, 2); }
        } else {
          if ((r = conv(h, c2, b_t1, nullptr, 1, b_t2, Bc, H, W, s))) return r;
          if ((r = conv(h, c3, b_t2, idt, 1, y, Bc, Ho, Wo, s))) return r;
        }
        x = y;
        H = Ho; W = Wo;
      }
      const char* names[3] = {"layer1", "layer2", "layer3"};
      if (int r = tap_save(h, names[st], x, (size_t)Bc * H * 2 * W * kStages[st].planes * 4, s)) return r;
      prof_mark(h, names[st], s);
      if (feat_out && st + 1 == upto) {  // cotr_backbone: [Bc, H, 2W, 4*planes] of this stage, NHWC over the pair
        const size_t per_pair = (size_t)H * 2 * W * kStages[st].planes * 4;
        HIPCHK(h, hipMemcpyAsync(feat_out + (size_t)b0 * per_pair, x, (size_t)Bc * per_pair * sizeof(float),
                                 hipMemcpyDeviceToDevice, s));
        break;
      }
    }
    if (feat_out) continue;
    // ---- input_proj: x is [Bc*512, 1024] --------------------------------------------------
    const int M = Bc * TOK;
    int r;
    if ((r = linear(h, x, nullptr, 0, 1, 0, h->ip_w, h->ip_b, nullptr, 0, 1.f, 0, t_src, M, D, CFEAT, s))) return r;
    if ((r = tap_save(h, "src", t_src, (size_t)M * D, s))) return r;
    prof_mark(h, "input_proj", s);
    // ---- encoder (transformer.py:143-159, post-norm) ----------------------------------------
    float* mem_c = memory + (size_t)b0 * TOK * D;
    const float* xin = t_src;
    for (size_t li = 0; li < h->enc.size(); ++li) {
      const EncW& e = h->enc[li];
      // q|k use src+pos, v uses src; q scaled by 32^-0.5 (transformer.py:147-153)
      if (M >= knob(KN_POS_TABLE_MIN_ROWS)) {
        if ((r = linear(h, xin, nullptr, 0, 1, 0, e.in_w, e.in_b, h->tab_qkv + li * TOK * 3 * D, 0, QSCALE, D, t_qkv, M, 3 * D, D, s, 0, TOK)))
          return r;
      } else if ((r = linear(h, xin, h->pos, TOK, 3 * D, 2 * D, e.in_w, e.in_b, nullptr, 0, QSCALE, D, t_qkv, M, 3 * D, D, s))) return r;
      float* y = (li + 1 == h->enc.size()) ? mem_c : (xin == t_alt ? t_pre2 : t_alt);
      bool fused = n_part != 0 && att_fused_applies(M);   // (the FFN block decides for itself: ffn_block)
      if (fused) {
        // few rows: out_proj inside the attention kernel (8 per-head partial outputs), summed + bias + residual + norm1 by ln_reduce
        KCHK(h, launch_attention_fused(t_qkv, 3 * D, nullptr, nullptr, nullptr, nullptr, 0.f, t_qkv + D, t_qkv + 2 * D, 3 * D,
                                       nullptr, 0, e.out_w, t_part, Bc, TOK, s), "attention+out_proj");
        prof_mark(h, "attention+oproj enc", s, 2);
        KCHK(h, launch_ln_reduce(t_part, 8, e.out_b, xin, e.n1w, e.n1b, t_x1, M, s), "ln_reduce");
        prof_mark(h, "ln_reduce heads", s, 2);
      } else if (att_rows_applies(Bc, TOK)) {
        // many rows: attention, out_proj, residual and norm1 in one launch (att_rows.hip)
        KCHK(h, launch_att_rows(t_qkv, 3 * D, nullptr, nullptr, nullptr, nullptr, 0.f, t_qkv + D, t_qkv + 2 * D, 3 * D, e.out_w, e.out_b,
                                xin, e.n1w, e.n1b, t_x1, Bc, TOK, s), "att_rows");
        prof_mark(h, "att_rows enc", s, 2);
      } else {
        KCHK(h, launch_attention(t_qkv, 3 * D, t_qkv + D, t_qkv + 2 * D, 3 * D, t_ao, D, Bc, TOK, s), "attention");
        prof_mark(h, "attention enc", s, 2);
        if ((r = linear(h, t_ao, nullptr, 0, 1, 0, e.out_w, e.out_b, xin, 0, 1.f, 0, t_tmp, M, D, D, s))) return r;
        if ((r = layernorm(h, t_tmp, e.n1w, e.n1b, t_x1, M, s))) return r;
      }
      if ((r = ffn_block(h, t_x1, e.l1w, e.l1b, e.l2w, e.l2b, e.n2w, e.n2b, t_hid, (size_t)TOK * 4 * FFN * Bc_max, fused ? t_tmp : t_ao, y, M, s)))
        return r;
      xin = y;
    }
    prof_mark(h, "encoder", s);
    // ---- decoder K/V of every layer: k = Wk(memory+pos), v = Wv(memory) (transformer.py:192-195)
    float* kv_c = kv + (size_t)b0 * TOK * KVLD;
    // (the hoisted K/V projection takes the table at any row count: 3072 columns fill the chip with large tiles even at one pair -
    // 18 -> 14 us there; the encoder in-projections only from knob pos_table_min_rows on)
    if ((h->side_mode & 2) && B <= ENC_CHUNK && knob(KN_POS_TABLE_MIN_ROWS) < (1 << 30)) {
      // cotr_forward with few rows (knob side_stream bit 1): decoder layer 0 needs its own K / V columns only - the other layers'
      // 5/6 of this product run on the handle's second stream beside decoder layer 0 (decode_chunk waits for them before layer 1)
      const int n0 = 2 * D, n1 = (int)KVLD - n0;
      if ((r = linear(h, mem_c, nullptr, 0, 1, 0, h->kv_w, h->kv_b, h->tab_kv, 0, 1.f, 0, kv_c, M, n0, D, s, (int)KVLD, TOK, (int)KVLD))) return r;
      HIPCHK(h, hipEventRecord(h->ev_mem, s));
      HIPCHK(h, hipStreamWaitEvent(h->side, h->ev_mem, 0));
      if ((r = linear(h, mem_c, nullptr, 0, 1, 0, h->kv_w + (size_t)n0 * D, h->kv_b + n0, h->tab_kv + n0, 0, 1.f, 0, kv_c + n0, M, n1, D, h->side,
                      (int)KVLD, TOK, (int)KVLD))) return r;
      HIPCHK(h, hipEventRecord(h->ev_kv, h->side));
    } else if (knob(KN_POS_TABLE_MIN_ROWS) < (1 << 30)) {
      if ((r = linear(h, mem_c, nullptr, 0, 1, 0, h->kv_w, h->kv_b, h->tab_kv, 0, 1.f, 0, kv_c, M, (int)KVLD, D, s, 0, TOK))) return r;
    } else if ((r = linear(h, mem_c, h->pos, TOK, 2 * D, D, h->kv_w, h->kv_b, nullptr, 0, 1.f, 0, kv_c, M, (int)KVLD, D, s))) return r;
    prof_mark(h, "dec_kv", s);
  }
  if (feat_out) return COTR_OK;
  h->taps["memory"] = {memory, (size_t)B * TOK * D};
  h->taps["kv"] = {kv, (size_t)B * TOK * KVLD};
  h->taps["pos"] = {h->pos, (size_t)TOK * D};
  h->enc_B = B;
  return COTR_OK;
}

int cotr_encode(cotr_handle h, const float* img, int B, cotr_stream stream) { return encode_impl(h, img, B, stream, nullptr); }

int cotr_backbone(cotr_handle h, const float* img, int B, float* features, cotr_stream stream) {
  if (!features) return COTR_ERR_ARG;
  return encode_impl(h, img, B, stream, features);
}

int cotr_backbone_upto(cotr_handle h, const float* img, int B, int stage, float* features, cotr_stream stream) {
  if (!features || stage < 1 || stage > 3) return COTR_ERR_ARG;
  return encode_impl(h, img, B, stream, features, stage);
}

}  // extern "C"

namespace {

struct DecPlan {
  int q_chunk = 0, nb_max = 0;
  size_t Rmax = 0, hid_cap = 0;   // most rows of a pass; floats behind `hid`
  float *qpos = nullptr, *tgt = nullptr, *q = nullptr, *ao = nullptr, *pre2 = nullptr, *t2 = nullptr, *pre3 = nullptr,
        *hid = nullptr, *part = nullptr;
  bool single_chunk = false;
};

int dec_plan(cotr_ctx* h, int B, int Q, DecPlan& d) {
  d.q_chunk = Q < DEC_ROWS ? Q : DEC_ROWS;
  const int pairs_per = Q < DEC_ROWS ? (DEC_ROWS / Q) : 1;
  d.nb_max = B < pairs_per ? B : pairs_per;
  d.Rmax = (size_t)d.nb_max * d.q_chunk;
  d.single_chunk = d.nb_max >= B && d.q_chunk >= Q;
  // A pass has at most Rmax rows - and with batch_split a small remainder follows a large first pass: `hid` (FFN hidden activations, FFN
  // floats per row; or the fused FFN's up to 16 partial outputs, 4 * FFN per row) and `part` (per-head partials of attention + out_proj,
  // 8 * D per row) are sized for the largest pass of EITHER kind that the fusion thresholds admit
  const size_t fr = d.Rmax < (size_t)knob(KN_FFN_FUSION_MAX_ROWS) ? d.Rmax : (size_t)knob(KN_FFN_FUSION_MAX_ROWS);
  const size_t ar = d.Rmax < (size_t)knob(KN_ATTENTION_FUSION_MAX_ROWS) ? d.Rmax : (size_t)knob(KN_ATTENTION_FUSION_MAX_ROWS);
  d.hid_cap = d.Rmax * FFN > fr * 4 * FFN ? d.Rmax * FFN : fr * 4 * FFN;
  const size_t part_cap = ar * 8 * D;
  int r = ensure(h, h->dec_scr, d.Rmax * 7 * D + d.hid_cap + part_cap);
  if (r) return r;
  float* p = h->dec_scr.ptr;
  d.qpos = p; p += d.Rmax * D;
  d.tgt = p; p += d.Rmax * D;
  d.q = p; p += d.Rmax * D;
  d.ao = p; p += d.Rmax * D;
  d.pre2 = p; p += d.Rmax * D;
  d.t2 = p; p += d.Rmax * D;
  d.pre3 = p; p += d.Rmax * D;
  d.hid = p; p += d.hid_cap;
  d.part = part_cap ? p : nullptr; p += part_cap;
  return COTR_OK;
}

// query-side prologue of one chunk: lin_sine encoding of the queries (cotr_model.py:34-36) and layer 0's
// q = Wq(0 + query_pos) * 32^-0.5 (tgt == 0 at layer 0, transformer.py:54).  Depends on the queries only.
int dec_prologue(cotr_ctx* h, const DecPlan& d, const float* qsrc, int nb, int nq, int Q, hipStream_t s, bool fused) {
  if (h->side_mode & 1) {
    // (knob side_stream bit 0) cotr_forward already ran the query encoding on the second stream, beside the backbone
    HIPCHK(h, hipStreamWaitEvent(s, h->ev_q, 0));
  } else {
    KCHK(h, launch_posenc(qsrc, d.qpos, nb, nq, Q, s), "posenc");
    prof_mark(h, "posenc", s, 2);
  }
  if (fused) return COTR_OK;   // the attention kernel projects its own queries
  const DecW& w = h->dec[0];
  return linear(h, d.qpos, nullptr, 0, 1, 0, w.q_w, w.q_b, nullptr, 0, QSCALE, D, d.q, nb * nq, D, D, s);
}

// one chunk of query rows through the decoder: rows [row0, row0 + nb*nq) of the scratch buffers
int decode_chunk(cotr_ctx* h, const DecPlan& d0, size_t row0, const float* qsrc, float* odst, const float* kv_c, int nb, int nq,
                 int Q, hipStream_t s) {
  DecPlan d = d0;
  d.qpos += row0 * D; d.tgt += row0 * D; d.q += row0 * D; d.ao += row0 * D; d.pre2 += row0 * D; d.t2 += row0 * D;
  d.pre3 += row0 * D; d.hid += row0 * FFN;   // (row0 is 0 in every caller; d.part is only valid for row0 == 0)
  const int L = (int)h->dec.size();
  const int KVLD = L * 2 * D;
  const int R = nb * nq;
  int r;
  const size_t hid_cap = d0.hid_cap;
  bool fused = d.part != nullptr && att_fused_applies(R) && ffn_fused_applies(R) && (size_t)ffn_fused_chunks(R) * R * D <= hid_cap;
  bool hs_normed = false;
  const bool rows = !fused && att_rows_applies(nb, nq);   // many rows: q projection, attention, out_proj, residual, norm2 in one launch
  if ((r = dec_prologue(h, d, qsrc, nb, nq, Q, s, fused || rows))) return r;
  // transformer.py:185-201 per layer (cross-attention only, post-norm)
  for (int li = 0; li < L; ++li) {
    const DecW& w = h->dec[li];
    const float* kl = kv_c + (size_t)li * 2 * D;          // this layer's K (then V) columns of the hoisted projection
    const float* tgt_in = li == 0 ? nullptr : d.tgt;      // tgt == 0 at layer 0 (transformer.py:54)
    if (li == 1 && (h->side_mode & 2)) HIPCHK(h, hipStreamWaitEvent(s, h->ev_kv, 0));   // K / V of layers 1-5 from the second stream
    if (fused) {
      // few rows: q = Wq(tgt + query_pos) * 32^-0.5 in the attention kernel's prologue, out_proj in its epilogue (8 per-head
      // partials), then ln_reduce: sum + bias + residual + norm2; FFN block; 4 launches per layer.  Last layer: decoder.norm rides
      // in the FFN block's ln_reduce launch (its input has no other consumer); pre2 = the normed 'hs'
      KCHK(h, launch_attention_fused(nullptr, 0, tgt_in, d.qpos, w.q_w, w.q_b, QSCALE, kl, kl + D, KVLD, nullptr, 0, w.out_w, d.part,
                                     nb, nq, s), "q_proj+attention+out_proj");
      prof_mark(h, "qproj+attention+oproj dec", s, 2);
      KCHK(h, launch_ln_reduce(d.part, 8, w.out_b, tgt_in, w.n2w, w.n2b, d.t2, R, s), "ln_reduce");
      prof_mark(h, "ln_reduce heads", s, 2);
      const bool post = li + 1 == L;
      if ((r = ffn_block(h, d.t2, w.l1w, w.l1b, w.l2w, w.l2b, w.n3w, w.n3b, d.hid, hid_cap, d.pre3, post ? d.pre2 : d.tgt, R, s,
                         post ? h->dn_w : nullptr, post ? h->dn_b : nullptr))) return r;
      hs_normed = post;
    } else {
      if (rows) {
        KCHK(h, launch_att_rows(nullptr, 0, tgt_in, d.qpos, w.q_w, w.q_b, QSCALE, kl, kl + D, KVLD, w.out_w, w.out_b, tgt_in, w.n2w, w.n2b,
                                d.t2, nb, nq, s), "att_rows");
        prof_mark(h, "att_rows dec", s, 2);
      } else {
        // q = Wq(tgt + query_pos) * 32^-0.5 (layer 0: computed by dec_prologue)
        if (li > 0 && (r = linear(h, d.tgt, d.qpos, 0, 1, 1, w.q_w, w.q_b, nullptr, 0, QSCALE, D, d.q, R, D, D, s))) return r;
        KCHK(h, launch_attention(d.q, D, kl, kl + D, KVLD, d.ao, D, nb, nq, s), "attention");
        prof_mark(h, "attention dec", s, 2);
        if ((r = linear(h, d.ao, nullptr, 0, 1, 0, w.out_w, w.out_b, tgt_in, 0, 1.f, 0, d.pre2, R, D, D, s))) return r;
        if ((r = layernorm(h, d.pre2, w.n2w, w.n2b, d.t2, R, s))) return r;
      }
      // (many rows, last layer: decoder.norm rides in the one-launch FFN block's epilogue; pre2 = the normed 'hs')
      const bool post = li + 1 == L && ffn_rows_applies(R);
      if ((r = ffn_block(h, d.t2, w.l1w, w.l1b, w.l2w, w.l2b, w.n3w, w.n3b, d.hid, hid_cap, d.pre3, post ? d.pre2 : d.tgt, R, s,
                         post ? h->dn_w : nullptr, post ? h->dn_b : nullptr))) return r;
      hs_normed = post;
    }
  }
  // decoder.norm + corr_embed on the last layer only (the reference computes all 6 and keeps [-1])
  if (!hs_normed && (r = layernorm(h, d.tgt, h->dn_w, h->dn_b, d.pre2, R, s))) return r;
  if ((r = linear(h, d.pre2, nullptr, 0, 1, 0, h->mlp_w[0], h->mlp_b[0], nullptr, 1, 1.f, 0, d.ao, R, D, D, s))) return r;
  if ((r = linear(h, d.ao, nullptr, 0, 1, 0, h->mlp_w[1], h->mlp_b[1], nullptr, 1, 1.f, 0, d.q, R, D, D, s))) return r;
  KCHK(h, launch_head2(d.q, h->mlp_w[2], h->mlp_b[2], odst, nb, nq, Q, s), "head2");
  prof_mark(h, "head2", s, 2);
  return COTR_OK;
}

int decode_impl(cotr_ctx* h, const float* queries, int B, int Q, float* out, hipStream_t s, const DecPlan& d) {
  const int L = (int)h->dec.size();
  const int KVLD = L * 2 * D;
  const float* kv = h->memkv.ptr + (size_t)B * TOK * D;
  prof_mark(h, "dec_begin", s);
  for (int b0 = 0, nb = 0; b0 < B; b0 += nb) {
    nb = dec_next_pairs(B - b0, d.nb_max, d.q_chunk);
    for (int q0 = 0; q0 < Q; q0 += d.q_chunk) {
      const int nq = (Q - q0) < d.q_chunk ? (Q - q0) : d.q_chunk;
      const int R = nb * nq;
      const float* qsrc = queries + ((size_t)b0 * Q + q0) * 2;
      float* odst = out + ((size_t)b0 * Q + q0) * 2;
      const float* kv_c = kv + (size_t)b0 * TOK * KVLD;
      int r;
      if ((r = decode_chunk(h, d, 0, qsrc, odst, kv_c, nb, nq, Q, s))) return r;
      if ((r = tap_save(h, "query_pos", d.qpos, (size_t)R * D, s))) return r;
      if ((r = tap_save(h, "hs", d.pre2, (size_t)R * D, s))) return r;
    }
  }
  prof_mark(h, "decoder", s);
  return COTR_OK;
}

int decode_check(cotr_ctx* h, const float* queries, int B, int Q, float* out) {
  if (!h->loaded) { h->err = "cotr_decode before cotr_load_weights"; return COTR_ERR_STATE; }
  if (B <= 0 || Q < 0) { h->err = "cotr_decode: B <= 0 or Q < 0"; return COTR_ERR_ARG; }
  if (Q > 0 && (!queries || !out)) { h->err = "cotr_decode: null queries/out"; return COTR_ERR_ARG; }
  return COTR_OK;
}

}  // namespace

extern "C" {

int cotr_decode(cotr_handle h, const float* queries, int B, int Q, float* out, cotr_stream stream) {
  if (!h) return COTR_ERR_ARG;
  if (int r = decode_check(h, queries, B, Q, out)) return r;
  if (h->enc_B != B) {
    h->err = "cotr_decode: no cached encode for this batch size (call cotr_encode first)";
    return COTR_ERR_STATE;
  }
  if (Q == 0) return COTR_OK;
  DEVICE_SCOPE(h);
  DecPlan d;
  if (int r = dec_plan(h, B, Q, d)) return r;
  return decode_impl(h, queries, B, Q, out, static_cast<hipStream_t>(stream), d);
}

// cotr_forward after its argument checks (the research library wraps this call; keep the entry point itself a one-liner)
static int forward_impl(cotr_ctx* h, const float* img, const float* queries, int B, int Q, float* out, hipStream_t s) {
  int side = 0;
  if (Q > 0) {
    // Few rows (one chunk of pairs, one chunk of queries, the fused small-row kernels): the chain is bound by its ~94 dependent
    // launches, not by the chip - work that depends on the queries only (their lin_sine encoding, cotr_model.py:34-36) or on the memory
    // only (K / V of decoder layers 1-5, transformer.py:192-195) leaves the chain for a second stream.  Both streams join before
    // cotr_forward returns: the caller sees one stream.
    side = knob(KN_SIDE_STREAM);
    if (h->dec.size() < 2) side &= ~2;
    if (h->prof || h->keep_taps || B > knob(KN_ENCODE_CHUNK) || (long)B * Q > 8192 || enc_next_chunk(B, knob(KN_ENCODE_CHUNK)) != B) side = 0;
    if (side) {
      if (!h->side) {   // the handle's second stream and its events, created at first use
        HIPCHK(h, hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
        for (hipEvent_t* ev : {&h->ev_fork, &h->ev_q, &h->ev_mem, &h->ev_kv}) HIPCHK(h, hipEventCreateWithFlags(ev, hipEventDisableTiming));
      }
      if (side & 1) HIPCHK(h, hipEventRecord(h->ev_fork, s));   // the previous call on `s` may still read the query encodings
    }
  }
  h->side_mode = side;
  struct Reset { cotr_ctx* h; ~Reset() { h->side_mode = 0; } } reset{h};
  int r = encode_impl(h, img, B, s, nullptr);
  if (r || Q == 0) return r;
  DecPlan d;
  if ((r = dec_plan(h, B, Q, d))) return r;
  if (side & 1) {
    HIPCHK(h, hipStreamWaitEvent(h->side, h->ev_fork, 0));
    KCHK(h, launch_posenc(queries, d.qpos, B, Q, Q, h->side), "posenc");
    HIPCHK(h, hipEventRecord(h->ev_q, h->side));
  }
  return decode_impl(h, queries, B, Q, out, s, d);
}

int cotr_forward(cotr_handle h, const float* img, const float* queries, int B, int Q, float* out,
                 cotr_stream stream) {
  if (!h) return COTR_ERR_ARG;
  if (int r = decode_check(h, queries, B, Q, out)) return r;
  DEVICE_SCOPE(h);
  return forward_impl(h, img, queries, B, Q, out, static_cast<hipStream_t>(stream));
}

// the passes a (B, Q) call is cut into under the handle's knobs (tests, tools): which = 0 encode passes, 1 decode passes; pairs per pass
int cotr_batch_chunks(cotr_handle h, int B, int Q, int which, int* sizes, int cap) {
  if (!h || B <= 0 || Q < 0 || (which != 0 && which != 1) || (cap > 0 && !sizes)) return COTR_ERR_ARG;
  KnobScope knob_scope_(&h->knobs);
  int n = 0;
  if (which == 0) {
    for (int b0 = 0, c = 0; b0 < B; b0 += c) {
      c = enc_next_chunk(B - b0, knob(KN_ENCODE_CHUNK));
      if (n < cap) sizes[n] = c;
      ++n;
    }
  } else {
    if (Q == 0) return 0;
    const int q_chunk = Q < DEC_ROWS ? Q : DEC_ROWS;
    const int pairs_per = Q < DEC_ROWS ? (DEC_ROWS / Q) : 1;
    const int nb_max = B < pairs_per ? B : pairs_per;
    for (int b0 = 0, c = 0; b0 < B; b0 += c) {
      c = dec_next_pairs(B - b0, nb_max, q_chunk);
      if (n < cap) sizes[n] = c;
      ++n;
    }
  }
  return n;
}

// bytes of the three arenas a call of that size carves (each rounded up to 256 B): what cotr_set_workspace must be given
int cotr_scratch_bytes(cotr_handle h, int B, int Q, size_t* bytes) {
  if (!h || !bytes || B <= 0 || Q < 0) return COTR_ERR_ARG;
  const size_t L = h->dec.empty() ? 6 : h->dec.size();
  const int* kn = h->knobs.v;
  const size_t Bc = B < kn[KN_ENCODE_CHUNK] ? B : kn[KN_ENCODE_CHUNK];
  const size_t per_pair = (size_t)128 * 256 * 64 + (size_t)64 * 128 * 64 + 5 * (size_t)64 * 128 * 256 +
                          6 * (size_t)TOK * D + (size_t)TOK * 3 * D + (size_t)TOK * 4 * FFN;
  // Monotone in B and in Q, and an upper bound over the fusion thresholds' settings: a workspace sized for (B, Q) must serve
  // every (B' <= B, Q' <= Q) - whose decoder passes can have MORE rows than (B, Q)'s own (2 x 16000 rows against 1 x 20000)
  // and, below the thresholds, more scratch per row - and flipping a tuning knob must not make a sized workspace too small.
  const size_t R = (size_t)B * Q < (size_t)DEC_ROWS ? (size_t)B * Q : (size_t)DEC_ROWS;
  const size_t thr_a = kn[KN_ATTENTION_FUSION_MAX_ROWS] > 1024 ? kn[KN_ATTENTION_FUSION_MAX_ROWS] : 1024;
  const size_t thr_f = kn[KN_FFN_FUSION_MAX_ROWS] > 1024 ? kn[KN_FFN_FUSION_MAX_ROWS] : 1024;
  const size_t f_memkv = (size_t)B * TOK * (D + L * 2 * D);
  const size_t enc_rows = Bc * TOK;
  const size_t f_enc = per_pair * Bc + 8 * (enc_rows < thr_a ? enc_rows : thr_a) * D;
  const size_t f_dec = R * (7 * D + FFN) + (R < thr_f ? R : thr_f) * 3 * FFN + (R < thr_a ? R : thr_a) * 8 * D;
  size_t total = 0;
  for (size_t f : {f_memkv, f_enc, f_dec}) total = ((total + 255) & ~size_t(255)) + f * sizeof(float);
  *bytes = total + 256;
  return COTR_OK;
}

// Scratch from the CALLER's allocator (torch's caching allocator in the Python binding): the handle carves its encode cache
// and its two scratch arenas from [ws, ws + bytes) instead of owning hipMalloc'ed memory that it would have to hipFree +
// hipMalloc (a device synchronisation in the middle of the stream) whenever a larger batch arrives.  With keep_encode a cached
// encode is carried over (copied on `stream`; its region is the first carve, so the workspace should be sized for the same
// number of pairs); ws == NULL returns to handle-owned memory and drops it.  The memory must stay valid until the next cotr_set_workspace /
// cotr_destroy and all work enqueued on it has finished.
int cotr_set_workspace(cotr_handle h, void* ws, size_t bytes, int keep_encode, cotr_stream stream) {
  if (!h || (ws != nullptr && (bytes == 0 || ((uintptr_t)ws & 255)))) return COTR_ERR_ARG;
  DEVICE_SCOPE(h);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // a cached encode moves with the workspace (copied on `stream`, ordered after the work that produced it): a caller that
  // encodes once and then decodes a larger query set than ever before keeps its encode
  const size_t L = h->dec.size();
  const size_t keep = (ws != nullptr && keep_encode && h->enc_B > 0 && h->memkv.ptr) ? (size_t)h->enc_B * TOK * (D + L * 2 * D) : 0;
  if (keep * sizeof(float) > bytes) { h->err = "cotr_set_workspace: smaller than the cached encode"; return COTR_ERR_ARG; }
  if (keep) HIPCHK(h, hipMemcpyAsync(ws, h->memkv.ptr, keep * sizeof(float), hipMemcpyDeviceToDevice, s));
  bool synced = false;
  for (Arena* a : {&h->memkv, &h->enc_scr, &h->dec_scr}) {
    if (a->ptr && !a->external) {
      if (!synced) HIPCHK(h, hipDeviceSynchronize());
      synced = true;
      HIPCHK(h, hipFree(a->ptr));
    }
    *a = Arena();
  }
  h->ws = static_cast<char*>(ws);
  h->ws_bytes = ws ? bytes : 0;
  h->ws_used = 0;
  h->taps.clear();
  if (keep) {   // first carve = the encode cache, already filled
    h->memkv.ptr = static_cast<float*>(ws);
    h->memkv.cap = keep;
    h->memkv.external = true;
    h->ws_used = keep * sizeof(float);
    h->taps["memory"] = {h->memkv.ptr, (size_t)h->enc_B * TOK * D};
    h->taps["kv"] = {h->memkv.ptr + (size_t)h->enc_B * TOK * D, (size_t)h->enc_B * TOK * L * 2 * D};
    h->taps["pos"] = {h->pos, (size_t)TOK * D};
  } else {
    h->enc_B = 0;
  }
  return COTR_OK;
}

int cotr_workspace_bytes(cotr_handle h, int B, int Q, size_t* bytes) {
  if (!h || !bytes || B <= 0 || Q < 0) return COTR_ERR_ARG;
  const size_t L = h->dec.empty() ? 6 : h->dec.size();
  const int* kn = h->knobs.v;
  const size_t Bc = B < kn[KN_ENCODE_CHUNK] ? B : kn[KN_ENCODE_CHUNK];
  const size_t per_pair = (size_t)128 * 256 * 64 + (size_t)64 * 128 * 64 + 5 * (size_t)64 * 128 * 256 +
                          6 * (size_t)TOK * D + (size_t)TOK * 3 * D + (size_t)TOK * 4 * FFN;
  const size_t q_chunk = Q < DEC_ROWS ? Q : DEC_ROWS;
  const size_t pairs_per = (Q > 0 && Q < DEC_ROWS) ? (DEC_ROWS / Q) : 1;
  const size_t nb = (size_t)B < pairs_per ? B : pairs_per;
  const size_t thr_a = kn[KN_ATTENTION_FUSION_MAX_ROWS] > 1024 ? kn[KN_ATTENTION_FUSION_MAX_ROWS] : 1024;
  const size_t R = nb * q_chunk;
  const size_t fr = R < (size_t)kn[KN_FFN_FUSION_MAX_ROWS] ? R : (size_t)kn[KN_FFN_FUSION_MAX_ROWS];
  const size_t ar = R < (size_t)kn[KN_ATTENTION_FUSION_MAX_ROWS] ? R : (size_t)kn[KN_ATTENTION_FUSION_MAX_ROWS];
  size_t fl = h->wfloats + (size_t)TOK * D + (size_t)B * TOK * (D + L * 2 * D) + per_pair * Bc +
              8 * (Bc * TOK < thr_a ? Bc * TOK : thr_a) * D +
              R * 7 * D + (R * FFN > fr * 4 * FFN ? R * FFN : fr * 4 * FFN) + ar * 8 * D;
  *bytes = fl * sizeof(float);
  return COTR_OK;
}

int cotr_debug_tap(cotr_handle h, const char* name, float* dst, size_t max_elems, size_t* n_elems,
                   cotr_stream stream) {
  if (!h || !name) return COTR_ERR_ARG;
  auto it = h->taps.find(name);
  if (it == h->taps.end()) { h->err = std::string("no tap named ") + name; return COTR_ERR_ARG; }
  if (n_elems) *n_elems = it->second.second;
  if (!dst) return COTR_OK;
  if (max_elems < it->second.second) { h->err = "tap buffer too small"; return COTR_ERR_ARG; }
  DEVICE_SCOPE(h);
  HIPCHK(h, hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  HIPCHK(h, hipMemcpy(dst, it->second.first, it->second.second * sizeof(float), hipMemcpyDefault));
  return COTR_OK;
}

int cotr_set_debug_taps(cotr_handle h, int enable) {
  if (!h) return COTR_ERR_ARG;
  h->keep_taps = enable != 0;
  return COTR_OK;
}

int cotr_set_profiling(cotr_handle h, int enable) {
  if (!h) return COTR_ERR_ARG;
  h->prof = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
  if (!h->prof) prof_reset(h);
  return COTR_OK;
}

int cotr_get_profile(cotr_handle h, const char** names, float* ms, int max_entries, int* n_entries) {
  if (!h || !n_entries) return COTR_ERR_ARG;
  *n_entries = 0;
  if (h->prof_ev.size() < 2) return COTR_OK;
  HIPCHK(h, hipEventSynchronize(h->prof_ev.back()));
  int n = 0;
  for (size_t i = 1; i < h->prof_ev.size() && n < max_entries; ++i) {
    if (h->prof_names[i] == "dec_begin") continue;  // interval between encode and decode calls
    float t = 0.f;
    HIPCHK(h, hipEventElapsedTime(&t, h->prof_ev[i - 1], h->prof_ev[i]));
    if (names) names[n] = h->prof_names[i].c_str();  // valid until the next profiled call
    if (ms) ms[n] = t;
    ++n;
  }
  *n_entries = n;
  return COTR_OK;
}

// ---- op-level entry points (tests) --------------------------------------------------------------
static int op_ret(int r) { return r == 0 ? COTR_OK : (r == -1 ? COTR_ERR_ARG : COTR_ERR_HIP); }

int cotr_op_linear(const float* x, const float* x2, int x2_row_mod, const float* w, const float* scale,
                   const float* bias, const float* residual, int relu, float* y, int M, int N, int K,
                   cotr_stream stream) {
  GemmParams p = base_params();
  p.M = M; p.N = N; p.K = K; p.A = x; p.lda = K;
  p.A2 = x2; p.lda2 = K; p.a2_row_mod = x2_row_mod; p.a2_period = 1; p.a2_width = 1;
  p.W = w; p.C = y; p.ldc = N; p.scale = scale; p.bias = bias; p.residual = residual; p.ldr = N; p.relu = relu;
  return op_ret(launch_gemm(GEMM_DENSE, p, static_cast<hipStream_t>(stream)));
}

int cotr_op_conv(const float* x, const float* w, const float* scale, const float* bias,
                 const float* residual, int relu, float* y, int B, int Hin, int Win, int Cin, int Cout,
                 int ksize, int stride, cotr_stream stream) {
  GemmParams p = base_params();
  const int pad = ksize / 2;
  p.Hin = Hin; p.Win = Win; p.Cin = Cin;
  p.Hout = (Hin + 2 * pad - ksize) / stride + 1;
  p.Wout = (Win + 2 * pad - ksize) / stride + 1;
  p.ksize = ksize; p.stride = stride; p.pad = pad;
  p.M = B * p.Hout * 2 * p.Wout; p.N = Cout; p.K = ksize * ksize * Cin;
  p.A = x; p.lda = Cin; p.W = w; p.C = y; p.ldc = Cout;
  p.scale = scale; p.bias = bias; p.residual = residual; p.ldr = Cout; p.relu = relu;
  return op_ret(launch_gemm(GEMM_CONV, p, static_cast<hipStream_t>(stream)));
}

int cotr_op_stem(const float* img, const float* w, const float* scale, const float* bias, float* y, int B,
                 cotr_stream stream) {
  GemmParams p = base_params();
  p.M = B * 128 * 256; p.N = 64; p.K = 160; p.A = img; p.W = w; p.C = y; p.ldc = 64;
  p.scale = scale; p.bias = bias; p.relu = 1;
  return op_ret(launch_gemm(GEMM_STEM, p, static_cast<hipStream_t>(stream)));
}

int cotr_op_stem_pool(const float* img, const float* w, const float* scale, const float* bias, float* y, int B,
                      cotr_stream stream) {
  return op_ret(launch_stem_pool(img, w, 160, scale, bias, y, B, static_cast<hipStream_t>(stream)));
}

int cotr_op_maxpool(const float* x, float* y, int B, int Hin, int Win, int C, cotr_stream stream) {
  return op_ret(launch_maxpool(x, y, B, Hin, Win, C, static_cast<hipStream_t>(stream)));
}

int cotr_op_attention(const float* q, int ldq, const float* k, const float* v, int ldkv, float* o, int ldo,
                      int nb, int nq, cotr_stream stream) {
  if (init_attention_attributes() != 0) return COTR_ERR_HIP;
  return op_ret(launch_attention(q, ldq, k, v, ldkv, o, ldo, nb, nq, static_cast<hipStream_t>(stream)));
}

int cotr_op_attention_fused(const float* q, int ldq, const float* x, const float* x2, const float* wq, const float* bq,
                            float qscale, const float* k, const float* v, int ldkv, float* o, int ldo, const float* wo,
                            float* part, int nb, int nq, cotr_stream stream) {
  return op_ret(launch_attention_fused(q, ldq, x, x2, wq, bq, qscale, k, v, ldkv, o, ldo, wo, part, nb, nq,
                                       static_cast<hipStream_t>(stream)));
}


int cotr_op_ln_reduce(const float* parts, int np, const float* bias, const float* residual, const float* w, const float* b,
                      float* y, int rows, cotr_stream stream) {
  if (!parts || np < 1 || !bias || !w || !b || !y) return COTR_ERR_ARG;
  return op_ret(launch_ln_reduce(parts, np, bias, residual, w, b, y, rows, static_cast<hipStream_t>(stream)));
}

int cotr_op_layernorm(const float* x, const float* w, const float* b, float* y, int rows, cotr_stream stream) {
  return op_ret(launch_layernorm(x, w, b, y, rows, static_cast<hipStream_t>(stream)));
}

int cotr_op_posenc(const float* pts, float* y, int n, cotr_stream stream) {
  return op_ret(launch_posenc(pts, y, 1, n, n, static_cast<hipStream_t>(stream)));
}

// ---- tuning hooks (tools/tune_gemm.py): time one GEMM/conv shape under an explicit config -------
static int bench_launches(int mode, int cfg, const GemmParams& p, int iters, float* us) {
  if (!us || iters <= 0) return COTR_ERR_ARG;
  hipStream_t s;
  if (hipStreamCreate(&s) != hipSuccess) return COTR_ERR_HIP;
  int rc = COTR_OK;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < 3 && rc == COTR_OK; ++i) rc = op_ret(launch_gemm_cfg(mode, cfg, p, s));  // warm-up (+ attribute opt-in)
  if (rc == COTR_OK && hipStreamSynchronize(s) != hipSuccess) rc = COTR_ERR_HIP;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  if (rc == COTR_OK) {
    // a captured chain of `iters` dependent launches: GPU-paced, no host launch cost in the timing
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) rc = COTR_ERR_HIP;
    for (int i = 0; i < iters && rc == COTR_OK; ++i) rc = op_ret(launch_gemm_cfg(mode, cfg, p, s));
    if (hipStreamEndCapture(s, &graph) != hipSuccess) rc = rc ? rc : COTR_ERR_HIP;
    if (rc == COTR_OK && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) rc = COTR_ERR_HIP;
  }
  if (rc == COTR_OK) {
    (void)hipGraphLaunch(exec, s);  // warm
    (void)hipStreamSynchronize(s);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0, s);
      (void)hipGraphLaunch(exec, s);
      (void)hipEventRecord(e1, s);
      if (hipEventSynchronize(e1) != hipSuccess) { rc = COTR_ERR_HIP; break; }
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    *us = best * 1000.f / iters;
  }
  if (exec) (void)hipGraphExecDestroy(exec);
  if (graph) (void)hipGraphDestroy(graph);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipStreamDestroy(s);
  return rc;
}

int cotr_gemm_num_configs(void) { return gemm_num_configs(); }

// one layer1 bottleneck from UNPACKED weights (tests): w1 [64][cin], w2 [64][3][3][64], w3 [256][64], wd [256][64] or NULL (device
// pointers); packs the fragment images on the host and launches bottleneck.hip
int cotr_op_bottleneck(const float* x, float* y, int B, int cin, const float* w1, const float* w2, const float* w3, const float* wd,
                       const float* s1, const float* b1, const float* s2, const float* b2, const float* s3, const float* b3,
                       const float* sd, const float* bd, cotr_stream stream) {
  if (!x || !y || !w1 || !w2 || !w3 || B <= 0) return COTR_ERR_ARG;
  std::vector<float> hw2(36864), hw3(16384), hwd(16384), packed(36864 + 2 * 16384);
  if (hipMemcpy(hw2.data(), w2, hw2.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return COTR_ERR_HIP;
  if (hipMemcpy(hw3.data(), w3, hw3.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return COTR_ERR_HIP;
  bottleneck_pack_w2(hw2.data(), packed.data());
  bottleneck_pack_w3(hw3.data(), packed.data() + 36864);
  if (wd) {
    if (hipMemcpy(hwd.data(), wd, hwd.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return COTR_ERR_HIP;
    bottleneck_pack_w3(hwd.data(), packed.data() + 36864 + 16384);
  }
  float* dev = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&dev), packed.size() * 4) != hipSuccess) return COTR_ERR_HIP;
  int rc = COTR_ERR_HIP;
  if (hipMemcpy(dev, packed.data(), packed.size() * 4, hipMemcpyHostToDevice) == hipSuccess) {
    rc = op_ret(launch_bottleneck(x, y, B, cin, w1, dev, dev + 36864, wd ? dev + 36864 + 16384 : nullptr, s1, b1, s2, b2, s3, b3, sd, bd,
                                  static_cast<hipStream_t>(stream)));
    if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) rc = COTR_ERR_HIP;
  }
  (void)hipFree(dev);
  return rc;
}

// ---- tuning knobs (KnobId / KnobSet in common.h, kKnobs above): per handle; h == NULL addresses the process-wide set that the
// handle-less op-level entry points (cotr_op_*, cotr_bench_*, cotr_train_*) read ----
static int knob_index(const char* name) {
  if (name)
    for (int i = 0; i < KN_COUNT; ++i)
      if (strcmp(kKnobs[i].name, name) == 0) return i;
  return -1;
}
int cotr_knob_count(void) { return KN_COUNT; }
const char* cotr_knob_name(int i) { return (i >= 0 && i < KN_COUNT) ? kKnobs[i].name : nullptr; }
int cotr_get_knob(cotr_handle h, const char* name, int* value, int* default_value) {
  const int i = knob_index(name);
  if (i < 0) return COTR_ERR_ARG;
  if (value) *value = (h ? h->knobs : g_process_knobs).v[i];
  if (default_value) *default_value = kKnobs[i].def;
  return COTR_OK;
}
int cotr_set_knob(cotr_handle h, const char* name, int value) {
  const int i = knob_index(name);
  if (i < 0) {
    if (h) h->err = std::string("cotr_set_knob: no knob named ") + (name ? name : "(null)");
    return COTR_ERR_ARG;
  }
  if (!knob_value_ok(i, value)) {
    if (h) h->err = std::string("cotr_set_knob: value out of range for ") + name;
    return COTR_ERR_ARG;
  }
  (h ? h->knobs : g_process_knobs).v[i] = value;
  return COTR_OK;
}
// would cotr_set_knob accept this?  Changes nothing (a binding that remembers knobs before its handle exists asks this instead of trying
// the value on the process-wide set, which a concurrent handle-less call of another thread would see)
int cotr_check_knob(const char* name, int value) {
  const int i = knob_index(name);
  return (i >= 0 && knob_value_ok(i, value)) ? COTR_OK : COTR_ERR_ARG;
}
int cotr_reset_knobs(cotr_handle h) {
  (h ? h->knobs : g_process_knobs) = default_knobs();
  return COTR_OK;
}
int cotr_is_experimental(void) {
  return 0;
}

int cotr_bench_linear(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int cfg,
                      int iters, float* us) {
  GemmParams p = base_params();
  p.M = M; p.N = N; p.K = K; p.A = x; p.lda = K; p.W = w; p.C = y; p.ldc = N; p.bias = bias;
  if (cfg < 0) cfg = gemm_pick_config(GEMM_DENSE, p);
  return bench_launches(GEMM_DENSE, cfg, p, iters, us);
}

int cotr_bench_conv(const float* x, const float* w, const float* scale, const float* bias, float* y, int B, int Hin,
                    int Win, int Cin, int Cout, int ksize, int stride, int cfg, int iters, float* us) {
  GemmParams p = base_params();
  const int pad = ksize / 2;
  p.Hin = Hin; p.Win = Win; p.Cin = Cin;
  p.Hout = (Hin + 2 * pad - ksize) / stride + 1;
  p.Wout = (Win + 2 * pad - ksize) / stride + 1;
  p.ksize = ksize; p.stride = stride; p.pad = pad;
  p.M = B * p.Hout * 2 * p.Wout; p.N = Cout; p.K = ksize * ksize * Cin;
  p.A = x; p.lda = Cin; p.W = w; p.C = y; p.ldc = Cout; p.scale = scale; p.bias = bias; p.relu = 1;
  if (cfg < 0) cfg = gemm_pick_config(GEMM_CONV, p);
  return bench_launches(GEMM_CONV, cfg, p, iters, us);
}

// explicit-config variants of the op entry points (tests check every config against torch)
int cotr_op_linear_cfg(const float* x, const float* w, const float* bias, const float* residual, int relu, float* y,
                       int M, int N, int K, int cfg, cotr_stream stream) {
  GemmParams p = base_params();
  p.M = M; p.N = N; p.K = K; p.A = x; p.lda = K; p.W = w; p.C = y; p.ldc = N;
  p.bias = bias; p.residual = residual; p.ldr = N; p.relu = relu;
  return op_ret(launch_gemm_cfg(GEMM_DENSE, cfg, p, static_cast<hipStream_t>(stream)));
}

int cotr_op_conv_cfg(const float* x, const float* w, const float* scale, const float* bias, const float* residual,
                     int relu, float* y, int B, int Hin, int Win, int Cin, int Cout, int ksize, int stride, int cfg,
                     cotr_stream stream) {
  GemmParams p = base_params();
  const int pad = ksize / 2;
  p.Hin = Hin; p.Win = Win; p.Cin = Cin;
  p.Hout = (Hin + 2 * pad - ksize) / stride + 1;
  p.Wout = (Win + 2 * pad - ksize) / stride + 1;
  p.ksize = ksize; p.stride = stride; p.pad = pad;
  p.M = B * p.Hout * 2 * p.Wout; p.N = Cout; p.K = ksize * ksize * Cin;
  p.A = x; p.lda = Cin; p.W = w; p.C = y; p.ldc = Cout;
  p.scale = scale; p.bias = bias; p.residual = residual; p.ldr = Cout; p.relu = relu;
  return op_ret(launch_gemm_cfg(GEMM_CONV, cfg, p, static_cast<hipStream_t>(stream)));
}

// phase timestamps of the fused FFN launches that follow (device memory [workgroups][8], 100 MHz wall clock; slots: entry, loads
// issued, first tile usable, H of sub-chunk 0 complete, phase 2 of sub-chunk 0 issued, second W1 usable, loop done, stored);
// nullptr = off.  tools/ffn_phases.py
int cotr_debug_ffn_times(unsigned long long* times) {
  set_ffn_debug_times(times);
  return COTR_OK;
}

// the same for the fused attention launches (slots: entry, q projected, key loop done, merged, out projection staged, stored)
int cotr_debug_attention_times(unsigned long long* times) {
  set_attention_debug_times(times);
  return COTR_OK;
}

// the configuration the library would pick for this convolution (tools)
int cotr_gemm_pick_conv(int B, int Hin, int Win, int Cin, int Cout, int ksize, int stride) {
  GemmParams p = base_params();
  const int pad = ksize / 2;
  p.Hin = Hin; p.Win = Win; p.Cin = Cin;
  p.Hout = (Hin + 2 * pad - ksize) / stride + 1;
  p.Wout = (Win + 2 * pad - ksize) / stride + 1;
  p.ksize = ksize; p.stride = stride; p.pad = pad;
  p.M = B * p.Hout * 2 * p.Wout; p.N = Cout; p.K = ksize * ksize * Cin;
  p.lda = Cin; p.ldc = Cout; p.ldr = Cout;
  return gemm_pick_config(GEMM_CONV, p);
}

// one convolution launch with the k-split kernels' phase timestamps written to `times` (device, [workgroups][8] uint64, 100 MHz
// wall clock; slots 0..4 = entry, loads issued, first data usable, K loop done, stored): tools/conv_phases.py
int cotr_debug_conv_times(const float* x, const float* w, const float* scale, const float* bias, float* y, int B, int Hin, int Win,
                          int Cin, int Cout, int ksize, int stride, int cfg, unsigned long long* times, cotr_stream stream) {
  GemmParams p = base_params();
  const int pad = ksize / 2;
  p.Hin = Hin; p.Win = Win; p.Cin = Cin;
  p.Hout = (Hin + 2 * pad - ksize) / stride + 1;
  p.Wout = (Win + 2 * pad - ksize) / stride + 1;
  p.ksize = ksize; p.stride = stride; p.pad = pad;
  p.M = B * p.Hout * 2 * p.Wout; p.N = Cout; p.K = ksize * ksize * Cin;
  p.A = x; p.lda = Cin; p.W = w; p.C = y; p.ldc = Cout;
  p.scale = scale; p.bias = bias; p.relu = 1;
  p.dbg = times;
  return op_ret(launch_gemm_cfg(GEMM_CONV, cfg, p, static_cast<hipStream_t>(stream)));
}

// two independent convolutions of the same block input in one launch (tests): both outputs, explicit config
int cotr_op_conv_dual_cfg(const float* x, const float* w0, const float* scale0, const float* bias0, int relu0, float* y0, int Cout0,
                          int ksize0, int stride0, const float* w1, const float* scale1, const float* bias1, int relu1, float* y1,
                          int Cout1, int ksize1, int stride1, int B, int Hin, int Win, int Cin, int cfg, cotr_stream stream) {
  ConvW c0 = {w0, scale0, bias0, Cin, Cout0, ksize0, stride0}, c1 = {w1, scale1, bias1, Cin, Cout1, ksize1, stride1};
  const GemmParams p0 = conv_params(c0, x, nullptr, relu0, y0, B, Hin, Win), p1 = conv_params(c1, x, nullptr, relu1, y1, B, Hin, Win);
  return op_ret(launch_gemm_dual_cfg(GEMM_CONV, cfg, p0, p1, static_cast<hipStream_t>(stream)));
}

// ---- training step: kernels under the autograd tape of cotr_amd/training.py ---------------------------
#define TS static_cast<hipStream_t>(stream)
int cotr_train_add_rowmod(const float* x, const float* x2, int mod, float* y, int rows, cotr_stream stream) {
  return op_ret(train_add_rowmod(x, x2, mod, y, rows, TS));
}
int cotr_train_add_drop_ln_fwd(const float* x, const float* a, const float* w, const float* b, float* s_out, float* y, float* stats,
                               int rows, float p, uint32_t seed, cotr_stream stream) {
  return op_ret(train_add_drop_ln_fwd(x, a, w, b, s_out, y, stats, rows, p, seed, TS));
}
int cotr_train_ln_bwd_parts(int rows) { return train_ln_bwd_parts(rows); }
int cotr_train_ln_bwd(const float* dy, const float* s_in, const float* stats, const float* w, float* ds, float* da, float* part,
                      float* dwb, int rows, float p, uint32_t seed, cotr_stream stream) {
  return op_ret(train_ln_bwd(dy, s_in, stats, w, ds, da, part, dwb, rows, p, seed, TS));
}
int cotr_train_set_dropout_salt(const unsigned int* salt) {
  train_set_salt_ptr(salt);
  return COTR_OK;
}
int cotr_train_clear_dropout_salt(const unsigned int* salt) { return train_clear_salt_ptr_if(salt) ? COTR_OK : COTR_ERR_STATE; }

int cotr_train_dropout_fwd(float* x, size_t n, float p, uint32_t seed, cotr_stream stream) {
  return op_ret(train_dropout_fwd(x, n, p, seed, TS));
}
int cotr_train_relu_drop_bwd(const float* dy, const float* y, float* dx, size_t n, float p, cotr_stream stream) {
  return op_ret(train_relu_drop_bwd(dy, y, dx, n, p, TS));
}
int cotr_train_colsum_parts(int M) { return train_colsum_parts(M); }
int cotr_train_colsum(const float* x, float* part, float* out, int M, int N, cotr_stream stream) {
  return op_ret(train_colsum(x, part, out, M, N, TS));
}
int cotr_train_transpose(const float* src, float* dst, int R, int C, cotr_stream stream) {
  return op_ret(train_transpose(src, dst, R, C, TS));
}
int cotr_train_im2col(const float* x, float* col, int B, int Hin, int Win, int Cin, int ksize, int stride, cotr_stream stream) {
  return op_ret(train_im2col(x, col, B, Hin, Win, Cin, ksize, stride, TS));
}
int cotr_train_col2im(const float* dcol, float* dx, int B, int Hin, int Win, int Cin, int ksize, int stride, cotr_stream stream) {
  return op_ret(train_col2im(dcol, dx, B, Hin, Win, Cin, ksize, stride, TS));
}
int cotr_train_scale_rows(const float* w, const float* scale, float* out, int rows, int cols, cotr_stream stream) {
  return op_ret(train_scale_rows(w, scale, out, rows, cols, TS));
}
int cotr_train_transpose_batched(const float* src, float* dst, int batch, int R, int C, cotr_stream stream) {
  return op_ret(train_transpose_batched(src, dst, batch, R, C, TS));
}
int cotr_train_gemm_tn_splits(int M, int N, int K) { return train_gemm_tn_splits(M, N, K); }
int cotr_train_gemm_tn(const float* A, const float* B, float* part, float* out, float* colsum, int M, int N, int K,
                       cotr_stream stream) {
  return op_ret(train_gemm_tn(A, B, part, out, colsum, M, N, K, TS));
}
int cotr_train_gemm_tn_parts(const float* A, const float* B, float* part, int M, int N, int K, int with_colsum, cotr_stream stream) {
  const int r = train_gemm_tn_parts(A, B, part, M, N, K, with_colsum, TS);     // >= 0: number of partials written
  return r >= 0 ? r : op_ret(r);
}
int cotr_train_sum_parts(const float* part, int nparts, size_t n, float* out, cotr_stream stream) {
  if (!part || !out || nparts < 1) return COTR_ERR_ARG;
  return op_ret(train_sum_parts(part, nparts, n, out, TS));
}
int cotr_train_conv_wgrad_parts(const float* dz, const float* x, float* part, int B, int Hin, int Win, int Cin, int Cout, int ksize,
                                int stride, cotr_stream stream) {
  const int r = train_conv_wgrad_parts(dz, x, part, B, Hin, Win, Cin, Cout, ksize, stride, TS);   // >= 0: partials written; -1: not this form
  return r >= -1 ? r : op_ret(r);
}
static_assert(sizeof(cotr_reduce_src) == sizeof(TrainReduceSrc) && sizeof(cotr_reduce_job) == sizeof(TrainReduceJob),
              "include/cotr_hip.h and train.h disagree on the reduction records");
int cotr_train_reduce_jobs(const cotr_reduce_job* jobs, const cotr_reduce_src* srcs, const unsigned* chunk_job, int njobs, int nchunks,
                           cotr_stream stream) {
  if (njobs < 0 || nchunks < 0 || (njobs > 0 && (jobs == nullptr || srcs == nullptr || chunk_job == nullptr))) return COTR_ERR_ARG;
  return op_ret(train_reduce_jobs(reinterpret_cast<const TrainReduceJob*>(jobs), reinterpret_cast<const TrainReduceSrc*>(srcs), chunk_job,
                                  njobs, nchunks, TS));
}
static_assert(sizeof(cotr_adam_job) == sizeof(TrainAdamJob), "include/cotr_hip.h and train.h disagree on cotr_adam_job");
int cotr_train_adam(const cotr_adam_job* jobs, const unsigned* chunk_job, int nchunks, const float* g, float* m, float* v, const float* lr,
                    int ngroups, double beta1, double beta2, double eps, double bias_correction1, double bias_correction2_sqrt,
                    const float* step, cotr_stream stream) {
  if (nchunks < 0 || (nchunks > 0 && (jobs == nullptr || chunk_job == nullptr || g == nullptr || m == nullptr || v == nullptr)))
    return COTR_ERR_ARG;
  return op_ret(train_adam(reinterpret_cast<const TrainAdamJob*>(jobs), chunk_job, nchunks, g, m, v, lr, ngroups, beta1, beta2, eps,
                           bias_correction1, bias_correction2_sqrt, step, TS));
}
static_assert(sizeof(cotr_perm_job) == sizeof(TrainPermJob), "include/cotr_hip.h and train.h disagree on cotr_perm_job");
int cotr_train_perm_jobs(const cotr_perm_job* jobs, const unsigned* tile_job, int njobs, int ntiles, cotr_stream stream) {
  if (njobs < 0 || ntiles < 0 || (njobs > 0 && (jobs == nullptr || tile_job == nullptr))) return COTR_ERR_ARG;
  return op_ret(train_perm_jobs(reinterpret_cast<const TrainPermJob*>(jobs), tile_job, njobs, ntiles, TS));
}
int cotr_train_head_fwd(const float* x, const float* w, const float* b, float* y, int nb, int nq, cotr_stream stream) {
  return op_ret(launch_head2(x, w, b, y, nb, nq, nq, TS));
}
int cotr_train_head_bwd_parts(int rows) { return train_head_bwd_parts(rows); }
int cotr_train_head_bwd(const float* dy, const float* h, const float* w2, float* dh, float* part, float* dwb, int rows,
                        cotr_stream stream) {
  return op_ret(train_head_bwd(dy, h, w2, dh, part, dwb, rows, TS));
}
int cotr_train_attention_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo, float* lse,
                             int nb, int nq, float qscale, float p, uint32_t seed, cotr_stream stream) {
  return op_ret(train_attention_fwd(q, ldq, k, ldk, v, ldv, o, ldo, lse, nb, nq, qscale, p, seed, TS));
}
int cotr_train_attention_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                             const float* d_o, int ldo, const float* lse, float* delta, float* dq, int lddq, float* dk, int lddk,
                             float* dv, int lddv, int nb, int nq, float qscale, float p, uint32_t seed, float* scratch,
                             cotr_stream stream) {
  return op_ret(train_attention_bwd(q, ldq, k, ldk, v, ldv, o, d_o, ldo, lse, delta, dq, lddq, dk, lddk, dv, lddv, nb, nq, qscale, p,
                                    seed, scratch, TS));
}
size_t cotr_train_attention_bwd_scratch(int nb, int nq) { return nb > 0 && nq > 0 ? train_attention_bwd_scratch(nb, nq) : 0; }
#undef TS

// ---- engine-side input construction --------------------------------------------------
int cotr_crop_resize_pairs(const uint8_t* img_a, int ha, int wa, const uint8_t* img_b, int hb, int wb,
                           const int32_t* boxes, int n, float* out, int max_size, cotr_stream stream) {
  if (n < 0 || (n > 0 && (!img_a || !img_b || !boxes || !out))) return COTR_ERR_ARG;
  if (ha <= 0 || wa <= 0 || hb <= 0 || wb <= 0) return COTR_ERR_ARG;
  return op_ret(launch_crop_resize(img_a, ha, wa, img_b, hb, wb, boxes, n, out, max_size, static_cast<hipStream_t>(stream)));
}

int cotr_dense_cycle(const float* pred, int n_pairs, const double* affine, float* maps, cotr_stream stream) {
  if (n_pairs < 0 || (n_pairs > 0 && (!pred || !affine || !maps))) return COTR_ERR_ARG;
  return op_ret(launch_dense_cycle(pred, affine, maps, n_pairs, static_cast<hipStream_t>(stream)));
}

int cotr_dense_merge(const float* maps, const int32_t* boxes, int n_pairs, int side, int H, int W, float* flow,
                     float* conf, cotr_stream stream) {
  if (n_pairs <= 0 || !maps || !boxes || !flow || !conf || (side != 0 && side != 1) || H <= 0 || W <= 0 ||
      (int64_t)H * W > (int64_t)1 << 30)
    return COTR_ERR_ARG;
  return op_ret(launch_dense_merge(maps, boxes, n_pairs, side, H, W, flow, conf, static_cast<hipStream_t>(stream)));
}

int cotr_resize_f32(const float* src, int Hs, int Ws, int C, float* dst, int Hd, int Wd, cotr_stream stream) {
  if (!src || !dst || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0 || C <= 0 || (int64_t)Hd * Wd > (int64_t)1 << 30) return COTR_ERR_ARG;
  return op_ret(launch_resize_f32(src, Hs, Ws, C, dst, Hd, Wd, static_cast<hipStream_t>(stream)));
}

// fused FFN block: y = LayerNorm(x + linear2(relu(linear1(x)))) in two launches; scratch >= ffn chunks * M * 256 floats
int cotr_op_ffn_block(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, const float* ln_w,
                      const float* ln_b, float* scratch, float* y, int M, cotr_stream stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nch = ffn_fused_chunks(M);
  int r = launch_ffn_fused(x, w1, b1, w2, scratch, M, nch, s);
  if (r == 0) r = launch_ln_reduce(scratch, nch, b2, x, ln_w, ln_b, y, M, s);
  return op_ret(r);
}

int cotr_op_ffn_chunks(int M) { return ffn_fused_chunks(M); }

// the attention sub-layer in ONE launch for many rows (att_rows.hip): y = LN(residual + out_proj(MHA(q, k, v)) + bo); q [rows][ldq]
// given pre-scaled (wq == NULL) or projected here: q = ((x + x2) . wq^T + bq) * qscale (x may be NULL); residual may be NULL
int cotr_op_att_rows(const float* q, int ldq, const float* x, const float* x2, const float* wq, const float* bq, float qscale,
                     const float* k, const float* v, int ldkv, const float* wo, const float* bo, const float* residual,
                     const float* ln_w, const float* ln_b, float* y, int nb, int nq, cotr_stream stream) {
  return op_ret(launch_att_rows(q, ldq, x, x2, wq, bq, qscale, k, v, ldkv, wo, bo, residual, ln_w, ln_b, y, nb, nq,
                                static_cast<hipStream_t>(stream)));
}

// two 1x1 convolutions (K = 64) over the same x in ONE launch (expand.hip: layer1 block 0's downsample + conv1)
int cotr_op_expand(const float* x, int M, const float* w0, const float* s0, const float* b0, int relu0, float* y0, int n0, const float* w1,
                   const float* s1, const float* b1, int relu1, float* y1, int n1, cotr_stream stream) {
  return op_ret(launch_expand(x, M, w0, s0, b0, relu0, y0, n0, w1, s1, b1, relu1, y1, n1, static_cast<hipStream_t>(stream)));
}

// conv2 (3x3, 128 -> 128, stride 1 / 2) -> conv3 (1x1, 128 -> 512) + identity + ReLU of a layer2 bottleneck in ONE launch (conv23m.hip)
int cotr_op_conv23m(const float* t1, const float* w2, const float* s2, const float* b2, const float* w3, const float* s3, const float* b3,
                    const float* residual, float* y, int B, int stride, cotr_stream stream) {
  return op_ret(launch_conv23m(t1, w2, s2, b2, w3, s3, b3, residual, y, B, stride, static_cast<hipStream_t>(stream)));
}

// conv2 (3x3, 64 -> 64) -> conv3 (1x1, 64 -> 256) + identity + ReLU of a layer1 bottleneck in ONE launch (conv23.hip)
int cotr_op_conv23(const float* t1, const float* w2, const float* s2, const float* b2, const float* w3, const float* s3, const float* b3,
                   const float* residual, float* y, int B, cotr_stream stream) {
  return op_ret(launch_conv23(t1, w2, s2, b2, w3, s3, b3, residual, y, B, static_cast<hipStream_t>(stream)));
}

// the same block in ONE launch for many rows (ffn_rows.hip); post_w / post_b: optional second LayerNorm (decoder.norm); y != x
int cotr_op_ffn_rows(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, const float* ln_w,
                     const float* ln_b, const float* post_w, const float* post_b, float* y, int M, cotr_stream stream) {
  return op_ret(launch_ffn_rows(x, w1, b1, w2, b2, ln_w, ln_b, post_w, post_b, y, M, static_cast<hipStream_t>(stream)));
}

}  // extern "C"
