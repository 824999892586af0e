// Fused transformer feed-forward block for the small-M (one pair / ~1000 query rows) regime, fp32 MFMA, gfx950:
//     P[c] = relu(X . W1_c^T + b1_c) . W2[:, c]^T          c = hidden-unit chunk
// i.e. linear1 + ReLU + linear2 of COTR/models/transformer.py:156,199 (`linear2(dropout(activation(linear1(x))))`)
// in ONE launch; the hidden activations [rows x 1024] never leave the CU.  The NCH per-chunk partial outputs are
// summed, biased, added to the residual and LayerNorm-ed by ln_reduce_kernel (pointwise.hip) - the launch that
// followed linear2 anyway - so the block costs two launches instead of three and one prologue/epilogue latency less.
//
// Grid: (rows/32) x NCH workgroups of 8 wavefronts; a workgroup owns 32 rows and 1024/NCH hidden units, walked in
// sub-chunks of 64.  Per sub-chunk:
//   phase 1  H[32 x 64] = X[32 x 256] . W1_sub[64 x 256]^T as 2 x 4 tiles of 16 x 16 (v_mfma_f32_16x16x4_f32): wave w -> ONE tile
//            over the whole K = 256 (64 MFMAs), + b1, ReLU -> H in LDS; no cross-wave reduction
//   phase 2  out[32 x 256] += H . W2[:, sub]^T: wave w -> output columns 32w..32w+31, K = 64: 32 MFMAs; the W2 operand
//            goes global -> registers in MFMA layout (each element is used once per workgroup), prefetched under phase 1
// X and W1 sub-chunks arrive by LDS-DMA (one wave instruction = one padded 1040-B row).
#include <string.h>

#include "common.h"

#define FF_D 256
#define FF_H 1024
#define FF_LD 260   // padded LDS row of 256 floats
#define FF_HLD 68   // padded LDS row of the 64-wide hidden sub-chunk

struct FfnParams {
  const float* X;    // [M][256]
  const float* W1;   // [1024][256]
  const float* b1;   // [1024]
  const float* W2;   // [256][1024]
  float* P;          // [nch][M][256] partial outputs
  const float* zeros;
  int M, nch, chunk_major;
  int wt_partials;   // partial outputs with write-through stores
  unsigned long long* dbg;  // nullptr, or [workgroups][8] phase timestamps (100 MHz wall clock), cotr_debug_ffn_times
};

__device__ __forceinline__ float ffn_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__global__ __launch_bounds__(512) void ffn_fused_kernel(const FfnParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                       // [32][260]
  float* W1s = Xs + 32 * FF_LD;           // [64][260]
  float* Hs = W1s + 64 * FF_LD;           // [32][68]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
#define FFN_STAMP(slot)                                                                      \
  do {                                                                                       \
    if (p.dbg != nullptr && t == 0) p.dbg[(size_t)blockIdx.x * 8 + (slot)] = wall_clock64(); \
  } while (0)
  FFN_STAMP(0);
  // chunk_major: chunk index fastest over consecutive workgroups (= consecutive XCDs): an XCD works on 1/8 of the hidden
  // units, so W1/W2 (2 MB per block) cross the fabric once chip-wide instead of once per XCD; X (<= 1 MB) is replicated
  const int tiles = gridDim.x / p.nch;
  const int chunk = p.chunk_major ? blockIdx.x % p.nch : blockIdx.x / tiles;
  const int m0 = (p.chunk_major ? blockIdx.x / p.nch : blockIdx.x % tiles) * 32;
  const int hw = FF_H / p.nch;            // hidden units of this workgroup
  const int h0 = chunk * hw;
  const int nsub = hw / 64;

  // X tile: 32 rows, one DMA row per wave instruction (4 per wave)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave + 8 * i;
    const float* src = (m0 + row < p.M) ? p.X + (size_t)(m0 + row) * FF_D + lane * 4 : p.zeros;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(Xs + row * FF_LD), 16, 0, 0);
  }
  auto dma_w1 = [&](int sub) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = wave + 8 * i;
      const float* src = p.W1 + (size_t)(h0 + sub * 64 + row) * FF_D + lane * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(W1s + row * FF_LD), 16, 0, 0);
    }
  };
  dma_w1(0);
  FFN_STAMP(1);


  f32x16 acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc2[r] = 0.f;

  for (int sub = 0; sub < nsub; ++sub) {
    // W2 fragment of this sub-chunk for this wave's 32 output columns: lane (n = l31, half hh) holds
    // W2[32*wave + l31][h0 + sub*64 + j*8 + hh*4 .. +3], j = 0..7
    LDS_DMA_WAIT_ALL();
    __syncthreads();  // X and W1_sub have landed, previous phase 2 is done with Hs
    if (sub == 0) FFN_STAMP(2);
    else if (sub == 1) FFN_STAMP(5);
    f32x4 w2f[8];     // issued after the barrier (which drains vmcnt), in flight under phase 1
    const float* w2g = p.W2 + (size_t)(32 * wave + l31) * FF_H + h0 + sub * 64 + hh * 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) w2f[j] = *reinterpret_cast<const f32x4*>(w2g + j * 8);

    // ---- phase 1 ---------------------------------------------------------------------------------
    // H[32 x 64] as 2 x 4 tiles of 16 x 16 on v_mfma_f32_16x16x4_f32: each of the 8 wavefronts owns ONE tile over the whole
    // K = 256 (64 MFMAs = the matrix-pipe time of the former 32 x 32 blocks with K split four ways), so there is no cross-wave
    // reduction: no 32 KB of partial accumulators through LDS, one barrier less (phase stamps: 4.1 us of which 1.9 were MFMA).
    // Two accumulators (even / odd k groups) keep the MFMAs from waiting on their own result.
    {
      const int l15 = lane & 15, q4 = lane >> 4;
      const int rb = wave & 1, cb = wave >> 1;
      f32x4 ae = {0.f, 0.f, 0.f, 0.f}, ao = {0.f, 0.f, 0.f, 0.f};
      const float* xr = &Xs[(rb * 16 + l15) * FF_LD + q4 * 4];
      const float* wr = &W1s[(cb * 16 + l15) * FF_LD + q4 * 4];
#pragma unroll
      for (int kk = 0; kk < 16; kk += 2) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(xr + kk * 16), b0 = *reinterpret_cast<const f32x4*>(wr + kk * 16);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(xr + kk * 16 + 16), b1 = *reinterpret_cast<const f32x4*>(wr + kk * 16 + 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ae = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], b0[e], ae, 0, 0, 0);
          ao = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], b1[e], ao, 0, 0, 0);
        }
      }
      // D of 16x16x4: column = lane & 15 (hidden unit), row = (lane >> 4) * 4 + reg
      const float b1v = p.b1[h0 + sub * 64 + cb * 16 + l15];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = ae[r] + ao[r] + b1v;
        v = (v < 0.f) ? 0.f : v;
        Hs[(rb * 16 + q4 * 4 + r) * FF_HLD + cb * 16 + l15] = v;
      }
    }
    __syncthreads();  // H complete
    if (sub == 0) FFN_STAMP(3);
    if (sub + 1 < nsub) dma_w1(sub + 1);  // next W1 sub-chunk streams in under phase 2 (drained by the next barrier)

    // ---- phase 2 ---------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4 af = *reinterpret_cast<const f32x4*>(&Hs[l31 * FF_HLD + j * 8 + hh * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], w2f[j][e], acc2, 0, 0, 0);
    }
    if (sub == 0) FFN_STAMP(4);
  }

  FFN_STAMP(6);
  // partial output block of this wave: rows m0 + (r&3) + 8*(r>>2) + 4*hh, columns 32*wave + l31
  float* out = p.P + (size_t)chunk * p.M * FF_D;
  {
    // through a wave-private LDS tile (the W1 stage is free: no DMA is in flight after the last sub-chunk and every wave is
    // past its last read of it, barrier "H complete") so that the rows leave as float4 - one instruction = 8 rows x 128 B.
    // Write-through (sc1): the 8-16 MB of partial outputs are read once, by ln_reduce on all XCDs; left dirty in L2 they are
    // written back at the kernel boundary (MI355X_MICROARCH.md price table: + dirty bytes / 6 TB/s per boundary).
    float* stage = W1s + wave * (32 * 36);
#pragma unroll
    for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * hh) * 36 + l31] = acc2[r];
    const int sr = lane >> 3, sc = (lane & 7) * 4;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + sr;
      const f32x4 val = *reinterpret_cast<const f32x4*>(&stage[row * 36 + sc]);
      if (m0 + row < p.M) store_f32x4(out + (size_t)(m0 + row) * FF_D + 32 * wave + sc, val, p.wt_partials != 0);
    }
    FFN_STAMP(7);
    return;
  }
}

static const size_t kFfnSmem = (size_t)(32 * FF_LD + 64 * FF_LD + 32 * FF_HLD) * sizeof(float);

static thread_local unsigned long long* g_ffn_dbg = nullptr;   // set_ffn_debug_times: phase stamps of the next launches
void set_ffn_debug_times(unsigned long long* p) { g_ffn_dbg = p; }
// knob KN_XCD_MAPPING bit 4 set = plain stores for the partials; bit 2 = hidden-unit chunks over XCDs (measured: -112 MB of fabric
// traffic per forward but +2 % time -> off)

// hidden-unit chunks per row tile: enough workgroups to cover the 256 CUs, at most 16 (knob ffn_fused_max_chunks) partial outputs
int ffn_fused_chunks(int M) {
  const int tiles = (M + 31) / 32;
  int nch = 2;
  while (nch < knob(KN_FFN_FUSED_MAX_CHUNKS) && tiles * nch < 256) nch *= 2;
  return nch;
}


int launch_ffn_fused(const float* X, const float* W1, const float* b1, const float* W2, float* P, int M, int nch,
                     hipStream_t s) {
  if (M <= 0) return 0;
  if (nch < 1 || nch > 16 || FF_H % (nch * 64) != 0) return -1;
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kFfnSmem) != hipSuccess)
      return -2;
    attr_set.set();
  }
  FfnParams p;
  p.X = X; p.W1 = W1; p.b1 = b1; p.W2 = W2; p.P = P; p.zeros = gemm_zero_buffer(); p.M = M; p.nch = nch;
  p.chunk_major = (knob(KN_XCD_MAPPING) >> 2) & 1;
  p.wt_partials = ((knob(KN_XCD_MAPPING) >> 4) & 1) == 0;
  p.dbg = g_ffn_dbg;
  if (p.zeros == nullptr) return -2;
  hipLaunchKernelGGL(ffn_fused_kernel, dim3(((M + 31) / 32) * nch), dim3(512), kFfnSmem, s, p);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
