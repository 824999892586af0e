// conv1 7x7/2 (3 -> 64) + FrozenBN + ReLU + max-pool 3x3/2 of the ResNet stem in ONE launch, fp32 MFMA, gfx950.
// Replaces torchvision resnet50's conv1 / bn1 / relu / maxpool as the reference runs them per 256x256 half
// (COTR/models/backbone.py:71,79-92; FrozenBN :46-56).  The unfused pair (implicit-GEMM stem + maxpool kernel) wrote the
// 128x128x64 conv output of every half to HBM and read it back: 8.4 MB each way per pair, 268 MB each way at 32 pairs.
//
// One workgroup (4 wavefronts) = 4 x 8 pooled pixels of one half: it needs the 9 x 17 conv outputs around them and, for
// those, a 23 x 39 x 3 input patch (zero outside the half: the conv's padding), staged in LDS straight from the NCHW image.
// The conv is a [153 (-> 160) x 147 (-> 148)] x [148 x 64] contraction on v_mfma_f32_16x16x4_f32: wavefront w owns filters
// 16w .. 16w+15 (its 37 weight fragments stay in registers for all rows) and walks the 10 row blocks of 16 conv pixels;
// the A fragment is gathered from the patch with a per-lane table of the 37 (c,ky,kx) offsets.  Conv outputs go through
// BN + ReLU into LDS, the 3x3/2 max (conv positions outside the half are padding, i.e. skipped) is taken there, and only the
// pooled [64 ch] rows reach HBM, NHWC over the side-by-side pair.  25 % more MFMAs than the unfused conv (tile halo),
// a quarter of its HBM traffic.
#include "common.h"

#define SP_PH 4                    // pooled rows per workgroup
#define SP_PW 8                    // pooled columns per workgroup
#define SP_CH (2 * SP_PH + 1)      // 9 conv rows
#define SP_CW (2 * SP_PW + 1)      // 17 conv columns
#define SP_NPIX (SP_CH * SP_CW)    // 153 conv pixels
#define SP_MB 10                   // row blocks of 16
#define SP_IH (2 * SP_CH + 5)      // 23 input rows
#define SP_IW (2 * SP_CW + 5)      // 39 input columns
#define SP_KS 37                   // k steps of 4 (147 -> 148)
#define SP_CP 64                   // row (64 channels) of the conv tile in LDS: unpadded so that 3 workgroups fit a CU

typedef float f32x4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stem_pool_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                        const float* __restrict__ scale, const float* __restrict__ bias,
                                                        float* __restrict__ out, int wk) {
  __shared__ float patch[3 * SP_IH * SP_IW];                                 // 10.8 KB
  __shared__ __attribute__((aligned(16))) float ctile[SP_MB * 16 * SP_CP];   // 43.5 KB
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int tx = blockIdx.x % (64 / SP_PW), ty = blockIdx.x / (64 / SP_PW);  // pooled tile inside the half
  const int side = blockIdx.y, b = blockIdx.z;
  const int cy0 = 2 * ty * SP_PH - 1, cx0 = 2 * tx * SP_PW - 1;              // first conv row / column of the tile
  const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;                            // first input row / column of the patch
  const float* src = img + (size_t)b * 3 * 256 * 512 + side * 256;

  // all 11 loads of a thread are issued before the first LDS store (a rolled loop paid the HBM latency 11 times)
  constexpr int NLD = (3 * SP_IH * SP_IW + 255) / 256;
  float pv[NLD];
#pragma unroll
  for (int u = 0; u < NLD; ++u) {
    const int i = t + 256 * u;
    const int c = i / (SP_IH * SP_IW), r = i - c * (SP_IH * SP_IW);
    const int py = r / SP_IW, px = r - py * SP_IW;
    const int y = iy0 + py, x = ix0 + px;
    pv[u] = (i < 3 * SP_IH * SP_IW && y >= 0 && y < 256 && x >= 0 && x < 256) ? src[((size_t)c * 256 + y) * 512 + x] : 0.f;
  }
#pragma unroll
  for (int u = 0; u < NLD; ++u)
    if (t + 256 * u < 3 * SP_IH * SP_IW) patch[t + 256 * u] = pv[u];

  // B operand: lane (filter n = lane & 15 of this wavefront's 16, k group g = lane >> 4) holds w[n][4*ks + g]
  const int g = lane >> 4, n = wave * 16 + (lane & 15);
  float wf[SP_KS];
  int koff[SP_KS];
#pragma unroll
  for (int ks = 0; ks < SP_KS; ++ks) {
    const int k = 4 * ks + g;
    wf[ks] = k < 147 ? w[(size_t)n * wk + k] : 0.f;
    const int kk = k < 147 ? k : 0;
    const int c = kk / 49, r = kk - c * 49, ky = r / 7, kx = r - ky * 7;
    koff[ks] = c * (SP_IH * SP_IW) + ky * SP_IW + kx;
  }
  __syncthreads();

  const float sc = scale[n], bi = bias[n];
  // two row blocks per pass: two independent accumulator chains keep the matrix pipe busy (a single chain of dependent
  // 16x16x4 MFMAs leaves bubbles) and twice the LDS gathers are in flight
#pragma unroll 1
  for (int mb = 0; mb < SP_MB; mb += 2) {
    // A operand: lane (conv pixel m = mb*16 + (lane & 15), k group g) reads patch[c][2*cy + ky][2*cx + kx]
    int base[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int m = (mb + u) * 16 + (lane & 15);
      if (m >= SP_NPIX) m = SP_NPIX - 1;                   // rows 153..159 of the last block: duplicates, never pooled
      const int cy = m / SP_CW, cx = m - cy * SP_CW;
      base[u] = 2 * cy * SP_IW + 2 * cx;
    }
    f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float a0[SP_KS], a1[SP_KS];                            // all gathers of the pass in flight before the first MFMA
#pragma unroll
    for (int ks = 0; ks < SP_KS; ++ks) {
      a0[ks] = patch[base[0] + koff[ks]];
      a1[ks] = patch[base[1] + koff[ks]];
    }
#pragma unroll
    for (int ks = 0; ks < SP_KS; ++ks) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[ks], wf[ks], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[ks], wf[ks], acc1, 0, 0, 0);
    }
    // D: lane holds conv pixel mb*16 + 4*g + r, filter n
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v0 = fmaf(acc0[r], sc, bi), v1 = fmaf(acc1[r], sc, bi);
      ctile[(mb * 16 + 4 * g + r) * SP_CP + n] = (v0 < 0.f) ? 0.f : v0;
      ctile[((mb + 1) * 16 + 4 * g + r) * SP_CP + n] = (v1 < 0.f) ? 0.f : v1;
    }
  }
  __syncthreads();

  // 3x3/2 max-pool, pad 1: pooled (py, px) <- conv rows 2py-1 .. 2py+1, i.e. tile rows 2*ly .. 2*ly+2
  for (int i = t; i < SP_PH * SP_PW * 16; i += 256) {
    const int c4 = (i & 15) * 4, p = i >> 4;
    const int ly = p / SP_PW, lx = p - ly * SP_PW;
    f32x4v mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int cyg = cy0 + 2 * ly + dy;
      if (cyg < 0 || cyg >= 128) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int cxg = cx0 + 2 * lx + dx;
        if (cxg < 0 || cxg >= 128) continue;
        const f32x4v v = *reinterpret_cast<const f32x4v*>(&ctile[((2 * ly + dy) * SP_CW + 2 * lx + dx) * SP_CP + c4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) mx[e] = fmaxf(mx[e], v[e]);
      }
    }
    const int oy = ty * SP_PH + ly, ox = side * 64 + tx * SP_PW + lx;
    *reinterpret_cast<f32x4v*>(out + (((size_t)b * 64 + oy) * 128 + ox) * 64 + c4) = mx;
  }
}

// img [B,3,256,512] NCHW -> out [B,64,128,64] (NHWC over the side-by-side pair); w [64][wk] with k = c*49 + ky*7 + kx
int launch_stem_pool(const float* img, const float* w, int wk, const float* scale, const float* bias, float* out, int B,
                     hipStream_t s) {
  if (B <= 0) return 0;
  if (wk < 147) return -1;
  hipLaunchKernelGGL(stem_pool_kernel, dim3((64 / SP_PH) * (64 / SP_PW), 2, B), dim3(256), 0, s, img, w, scale, bias, out, wk);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
