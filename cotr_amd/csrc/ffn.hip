// Fused transformer feed-forward block for the small-M (one pair / ~1000 query rows) regime, fp32 MFMA, gfx950:
//     P[c] = relu(X . W1_c^T + b1_c) . W2[:, c]^T          c = hidden-unit chunk
// i.e. linear1 + ReLU + linear2 of COTR/models/transformer.py:156,199 (`linear2(dropout(activation(linear1(x))))`)
// in ONE launch; the hidden activations [rows x 1024] never leave the CU.  The NCH per-chunk partial outputs are
// summed, biased, added to the residual and LayerNorm-ed by ln_reduce_kernel (pointwise.hip) - the launch that
// followed linear2 anyway - so the block costs two launches instead of three and one prologue/epilogue latency less.
//
// Grid: (rows/32) x NCH workgroups of 8 wavefronts; a workgroup owns 32 rows and 1024/NCH hidden units, walked in
// sub-chunks of 64.  Per sub-chunk:
//   phase 1  H[32 x 64] = X[32 x 256] . W1_sub[64 x 256]^T: wave w -> 32x32 block (w&1), K quarter (w>>1): 32 MFMAs,
//            the 4 K-quarters summed through LDS in a fixed order, + b1, ReLU -> H in LDS
//   phase 2  out[32 x 256] += H . W2[:, sub]^T: wave w -> output columns 32w..32w+31, K = 64: 32 MFMAs; the W2 operand
//            goes global -> registers in MFMA layout (each element is used once per workgroup), prefetched under phase 1
// X and W1 sub-chunks arrive by LDS-DMA (one wave instruction = one padded 1040-B row).
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
};

__global__ __launch_bounds__(512) void ffn_fused_kernel(const FfnParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                       // [32][260]
  float* W1s = Xs + 32 * FF_LD;           // [64][260]
  float* red = W1s + 64 * FF_LD;          // [8 waves][16][64]
  float* Hs = red + 8 * 16 * 64;          // [32][68]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
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

  f32x16 acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
  const int blk = wave & 1, kq = wave >> 1;

  for (int sub = 0; sub < nsub; ++sub) {
    // W2 fragment of this sub-chunk for this wave's 32 output columns: lane (n = l31, half hh) holds
    // W2[32*wave + l31][h0 + sub*64 + j*8 + hh*4 .. +3], j = 0..7
    LDS_DMA_WAIT_ALL();
    __syncthreads();  // X and W1_sub have landed, previous phase 2 is done with Hs
    f32x4 w2f[8];     // issued after the barrier (which drains vmcnt), in flight under phase 1
    const float* w2g = p.W2 + (size_t)(32 * wave + l31) * FF_H + h0 + sub * 64 + hh * 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) w2f[j] = *reinterpret_cast<const f32x4*>(w2g + j * 8);

    // ---- phase 1 ---------------------------------------------------------------------------------
    f32x16 acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ko = kq * 64 + j * 8 + hh * 4;
      const f32x4 af = *reinterpret_cast<const f32x4*>(&Xs[l31 * FF_LD + ko]);
      const f32x4 bf = *reinterpret_cast<const f32x4*>(&W1s[(blk * 32 + l31) * FF_LD + ko]);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc1[r];
    __syncthreads();  // partials visible; every wave is done reading W1s
    {
      // wave w finishes block (w&1), accumulator rows 4*(w>>1) .. +3: sum of the 4 K-quarters, + b1, ReLU -> Hs
      const float b1v = p.b1[h0 + sub * 64 + blk * 32 + l31];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = kq * 4 + i;
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) v += red[((blk + 2 * q) * 16 + r) * 64 + lane];
        v += b1v;
        v = (v < 0.f) ? 0.f : v;
        const int m = (r & 3) + 8 * (r >> 2) + 4 * hh;
        Hs[m * FF_HLD + blk * 32 + l31] = v;
      }
    }
    __syncthreads();  // H complete
    if (sub + 1 < nsub) dma_w1(sub + 1);  // next W1 sub-chunk streams in under phase 2 (drained by the next barrier)

    // ---- phase 2 ---------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4 af = *reinterpret_cast<const f32x4*>(&Hs[l31 * FF_HLD + j * 8 + hh * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], w2f[j][e], acc2, 0, 0, 0);
    }
  }

  // partial output block of this wave: rows m0 + (r&3) + 8*(r>>2) + 4*hh, columns 32*wave + l31
  float* out = p.P + (size_t)chunk * p.M * FF_D;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
    if (m < p.M) out[(size_t)m * FF_D + 32 * wave + l31] = acc2[r];
  }
}

static const size_t kFfnSmem = (size_t)(32 * FF_LD + 64 * FF_LD + 8 * 16 * 64 + 32 * FF_HLD) * sizeof(float);

static int g_ffn_chunk_major = 0;  // measured: -112 MB of fabric traffic per forward but +2 % time -> off (cotr_set_xcd_mapping bit 2)
void set_ffn_chunk_major(int v) { g_ffn_chunk_major = v; }

// hidden-unit chunks per row tile: enough workgroups to cover the 256 CUs, at most 16 partial outputs
int ffn_fused_chunks(int M) {
  const int tiles = (M + 31) / 32;
  int nch = 2;
  while (nch < 16 && tiles * nch < 256) nch *= 2;
  return nch;
}

int launch_ffn_fused(const float* X, const float* W1, const float* b1, const float* W2, float* P, int M, int nch,
                     hipStream_t s) {
  if (M <= 0) return 0;
  if (nch < 1 || nch > 16 || FF_H % (nch * 64) != 0) return -1;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kFfnSmem) != hipSuccess)
      return -2;
    attr_set = true;
  }
  FfnParams p;
  p.X = X; p.W1 = W1; p.b1 = b1; p.W2 = W2; p.P = P; p.zeros = gemm_zero_buffer(); p.M = M; p.nch = nch; p.chunk_major = g_ffn_chunk_major;
  if (p.zeros == nullptr) return -2;
  hipLaunchKernelGGL(ffn_fused_kernel, dim3(((M + 31) / 32) * nch), dim3(512), kFfnSmem, s, p);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
