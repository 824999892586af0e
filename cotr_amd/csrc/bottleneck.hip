// One whole ResNet bottleneck of layer1 in ONE launch (fp32 MFMA, gfx950) - torchvision Bottleneck.forward with the reference's
// FrozenBatchNorm2d (COTR/models/backbone.py:46-56) on the NHWC "side-by-side" layout:
//     t1 = relu(bn1(conv1x1(x)))        C_in -> 64
//     t2 = relu(bn2(conv3x3(t1)))       64 -> 64, padding 1, each 64-wide half padded on its own
//     y  = relu(bn3(conv1x1(t2)) + idt) 64 -> 256, idt = x (blocks 1, 2) or bn_d(conv1x1_d(x)) (block 0, C_in = 64)
//
// Why: at one pair every launch of the backbone pays ~1.7 us of dispatch (the 8 XCDs start a grid one after the other, 1.3 us
// first to last: profiles/r3_launch_ramp_xcd_skew.txt) plus ~2 us of its own ramp (kernel arguments, first-load latency, epilogue)
// around 2-4 us of useful work, and layer1's weights are small (70 KB ... 280 KB per block): the three convolutions of a block
// fit one workgroup's LDS + registers.  9 launches of layer1 become 3.
//
// Work decomposition: one workgroup (8 wavefronts) = a 4 x 8 tile of output pixels of one half (32 pixels; 256 workgroups per pair).
//   phase A  conv1 on the tile's 6 x 10 halo patch (60 pixels, padded to 64 rows): [64 x C_in] . [64 x C_in]^T, four 32 x 32 MFMA
//            blocks, the contraction split over two wavefront groups, summed through LDS in a fixed order; + bn1 + ReLU; pixels
//            outside the half become 0 (the zero padding of conv2 applies to t1, not to x) -> t1 patch in LDS.  The halo makes
//            this phase do 2x the algorithmic work of conv1 (60 of 64 rows are real, 32 are the tile) - the price of not
//            exchanging halos between workgroups.
//   phase B  conv2 from the t1 patch: [32 x 576] . [64 x 576]^T; a wavefront = one 32-column block x one quarter of the 72
//            8-deep K slices (slice = tap, 8 channels); its A rows are the patch rows shifted by the tap; the W2 fragments come
//            global -> registers from an array packed at load time in exactly the register image (one wave instruction = 1 KB
//            contiguous); 4 partial sums through LDS, fixed order; + bn2 + ReLU -> t2 in LDS
//   phase C  conv3: [32 x 64] . [256 x 64]^T, a wavefront = one 32-column block, W3 fragments packed like W2; block 0 also runs the
//            downsample product on the x tile; + bn3 (+ bn_d) + identity (read back from the x patch in LDS) + ReLU -> y
// All weight fragments are requested in the first instructions, in the order of use: x patch + W1 (phase A), W2, W3 (, Wd); the
// compiler counts the waits (plain loads), so phase A starts when ITS operands have landed while W2 / W3 are still streaming.
#include "common.h"

struct BottleneckParams {
  const float* x;    // [B][64][128][CIN]
  float* y;          // [B][64][128][256]
  const float* w1;   // [64][CIN]
  const float* w2p;  // packed: [2 nb][4 kg][18][64 lanes] float4
  const float* w3p;  // packed: [8 waves][8][64 lanes] float4
  const float* wdp;  // packed like w3p (block 0) or nullptr
  const float *s1, *b1, *s2, *b2, *s3, *b3, *sd, *bd;
  const float* zeros;
  int B;
};

// layer1 geometry: halves of 64 x 64 pixels
#define BT_H 64
#define BT_W 64
#define BT_LDT 68   // padded LDS row of a 64-channel tile

template <int CIN, bool DS>
__global__ __launch_bounds__(512) void bottleneck_kernel(const BottleneckParams p) {
  static_assert(CIN == 64 || CIN == 256, "layer1: 64 (block 0) or 256 input channels");
  static_assert(!DS || CIN == 64, "the downsample branch belongs to block 0");
  constexpr int LDX = CIN + 4;
  constexpr int C4 = CIN / 4;                   // float4 per row
  constexpr int NX = 64 * C4 / 512;             // float4 per thread for a 64-row operand
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                             // [64][LDX]  x on the halo patch (row = py * 10 + px, rows 60..63 unused)
  float* W1s = Xs + 64 * LDX;                   // [64][LDX]  conv1 weights; later the partial-sum buffers
  float* T1s = W1s + 64 * LDX;                  // [64][68]   t1 on the halo patch
  float* T2s = T1s + 64 * BT_LDT;               // [32][68]   t2 on the tile
  // partial sums, phase A: [4 blocks][16][64], phase B: [4 kg][2 nb][16][64] = 32 KB: on top of W1s where that is large enough
  // (CIN = 256: 65 KB, dead after phase A's MFMAs), a region of its own behind T2s otherwise
  float* RED = (CIN == 256) ? W1s : T2s + 32 * BT_LDT;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  // workgroup -> (pair, tile row, tile column): 16 x 16 tiles of 4 x 8 pixels per pair
  const int bid = blockIdx.x;
  const int tx = bid & 15, ty = (bid >> 4) & 15, b = bid >> 8;
  const int side = tx >> 3, x0 = (tx & 7) * 8, y0 = ty * 4;
  const size_t pair_base = (size_t)b * BT_H * (2 * BT_W);

  // ---- requests, in the order of use -----------------------------------------------------------------------------------------
  f32x4 xr[NX], wr[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int idx = t + 512 * i;
    const int row = idx / C4, c4 = idx % C4;     // C4 is a power of two
    const int py = (row * 205) >> 11, px = row - py * 10;    // row / 10 for row < 64
    const int yy = y0 - 1 + py, xx = x0 - 1 + px;
    const bool ok = row < 60 && yy >= 0 && yy < BT_H && xx >= 0 && xx < BT_W;
    const float* src = ok ? p.x + (pair_base + (size_t)yy * (2 * BT_W) + side * BT_W + xx) * CIN + c4 * 4 : p.zeros;
    xr[i] = *reinterpret_cast<const f32x4*>(src);
    wr[i] = *reinterpret_cast<const f32x4*>(p.w1 + (size_t)row * CIN + c4 * 4);
  }
  // phase B / C roles and their weight fragments
  const int nbB = wave & 1, kgB = wave >> 1;
  f32x4 w2f[18], w3f[8], wdf[DS ? 8 : 1];
  {
    const f32x4* w2g = reinterpret_cast<const f32x4*>(p.w2p) + (size_t)((nbB * 4 + kgB) * 18) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 18; ++i) w2f[i] = w2g[i * 64];
    const f32x4* w3g = reinterpret_cast<const f32x4*>(p.w3p) + (size_t)(wave * 8) * 64 + lane;
#pragma unroll
    for (int j = 0; j < 8; ++j) w3f[j] = w3g[j * 64];
    if constexpr (DS) {
      const f32x4* wdg = reinterpret_cast<const f32x4*>(p.wdp) + (size_t)(wave * 8) * 64 + lane;
#pragma unroll
      for (int j = 0; j < 8; ++j) wdf[j] = wdg[j * 64];
    }
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int idx = t + 512 * i;
    const int row = idx / C4, c4 = idx % C4;
    *reinterpret_cast<f32x4*>(&Xs[row * LDX + c4 * 4]) = xr[i];
    *reinterpret_cast<f32x4*>(&W1s[row * LDX + c4 * 4]) = wr[i];
  }
  __syncthreads();

  // ---- phase A: t1 = relu(bn1(x_patch . W1^T)), 64 x 64, wavefront = (row block mb, column block nb, K half kh) ---------------
  {
    const int mb = wave & 1, nb = (wave >> 1) & 1, kh = wave >> 2;
    constexpr int NS = CIN / 16;                // 8-deep slices per K half
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* ar = &Xs[(32 * mb + l31) * LDX + kh * (CIN / 2) + hh * 4];
    const float* br = &W1s[(32 * nb + l31) * LDX + kh * (CIN / 2) + hh * 4];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ar + j * 8);
      const f32x4 bb = *reinterpret_cast<const f32x4*>(br + j * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], bb[e], acc, 0, 0, 0);
    }
    __syncthreads();                            // every wavefront is done reading W1s: it becomes the partial-sum buffer
    if (kh == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) RED[((mb * 2 + nb) * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (kh == 0) {
      const int ch = 32 * nb + l31;
      const float sc = p.s1[ch], bi = p.b1[ch];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * hh;       // patch row of this accumulator element
        const int py = (row * 205) >> 11, px = row - py * 10;
        const int yy = y0 - 1 + py, xx = x0 - 1 + px;
        const bool ok = row < 60 && yy >= 0 && yy < BT_H && xx >= 0 && xx < BT_W;
        float v = acc[r] + RED[((mb * 2 + nb) * 16 + r) * 64 + lane];
        v = fmaf(v, sc, bi);
        v = (v < 0.f) ? 0.f : v;                // NaN passes through like torch.relu
        T1s[row * BT_LDT + ch] = ok ? v : 0.f;  // zero padding of conv2 (a NaN outside the half cannot exist: x there is 0)
      }
    }
    __syncthreads();
  }

  // ---- phase B: t2 = relu(bn2(conv3x3(t1))), 32 x 64, wavefront = (column block nbB, K quarter kgB: 18 slices of 8) ------------
  {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int rowbase = (l31 >> 3) * 10 + (l31 & 7);                     // patch row of tap (0, 0) of output pixel l31
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      const int s = 18 * kgB + i;                                        // wave-uniform
      const int tap = s >> 3, kc = s & 7;
      const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;                 // tap / 3 for tap < 9
      const f32x4 a = *reinterpret_cast<const f32x4*>(&T1s[(rowbase + ky * 10 + kx) * BT_LDT + kc * 8 + hh * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], w2f[i][e], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) RED[((kgB * 2 + nbB) * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    // wavefront w finishes column block nb = w & 1, accumulator rows (w >> 1) * 4 .. + 3; partials summed in K order
    const int nb = wave & 1, rg = wave >> 1;
    const int ch = 32 * nb + l31;
    const float sc = p.s2[ch], bi = p.b2[ch];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = rg * 4 + rr;
      float v = 0.f;
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) v += RED[((kg * 2 + nb) * 16 + r) * 64 + lane];
      v = fmaf(v, sc, bi);
      v = (v < 0.f) ? 0.f : v;
      T2s[((r & 3) + 8 * (r >> 2) + 4 * hh) * BT_LDT + ch] = v;
    }
    __syncthreads();
  }

  // ---- phase C: y = relu(bn3(t2 . W3^T) + identity), 32 x 256, wavefront = column block `wave` ---------------------------------
  {
    f32x16 acc, accd;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = accd[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(&T2s[l31 * BT_LDT + j * 8 + hh * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], w3f[j][e], acc, 0, 0, 0);
    }
    if constexpr (DS) {                                                  // identity = bn_d(x_tile . Wd^T), K = 64
      const int crow = ((l31 >> 3) + 1) * 10 + (l31 & 7) + 1;            // patch row of tile pixel l31
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(&Xs[crow * LDX + j * 8 + hh * 4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) accd = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], wdf[j][e], accd, 0, 0, 0);
      }
    }
    const int n = 32 * wave + l31;
    const float sc = p.s3[n], bi = p.b3[n];
    float scd = 0.f, bid_ = 0.f;
    if constexpr (DS) { scd = p.sd[n]; bid_ = p.bd[n]; }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int px = (r & 3) + 8 * (r >> 2) + 4 * hh;                    // tile pixel of this accumulator element
      const int oy = px >> 3, ox = px & 7;
      float v = fmaf(acc[r], sc, bi);
      if constexpr (DS) v += fmaf(accd[r], scd, bid_);
      else v += Xs[((oy + 1) * 10 + ox + 1) * LDX + n];                  // the block input at this pixel (CIN == 256 channels)
      v = (v < 0.f) ? 0.f : v;
      p.y[(pair_base + (size_t)(y0 + oy) * (2 * BT_W) + side * BT_W + x0 + ox) * 256 + n] = v;
    }
  }
}

template <int CIN>
static constexpr size_t bottleneck_smem() {
  return (size_t)(2 * 64 * (CIN + 4) + 64 * BT_LDT + 32 * BT_LDT + (CIN == 256 ? 0 : 8 * 16 * 64)) * sizeof(float);
}

// x [B][64][128][cin] -> y [B][64][128][256]; cin 64 with the downsample branch (wdp != nullptr) or 256 without
int launch_bottleneck(const float* x, float* y, int B, int cin, const float* w1, const float* w2p, const float* w3p, const float* wdp,
                      const float* s1, const float* b1, const float* s2, const float* b2, const float* s3, const float* b3,
                      const float* sd, const float* bd, hipStream_t s) {
  if (B <= 0) return 0;
  if (!((cin == 64 && wdp != nullptr) || (cin == 256 && wdp == nullptr))) return -1;
  BottleneckParams p;
  p.x = x; p.y = y; p.w1 = w1; p.w2p = w2p; p.w3p = w3p; p.wdp = wdp;
  p.s1 = s1; p.b1 = b1; p.s2 = s2; p.b2 = b2; p.s3 = s3; p.b3 = b3; p.sd = sd; p.bd = bd;
  p.zeros = gemm_zero_buffer();
  p.B = B;
  if (p.zeros == nullptr) return -2;
  static_assert(bottleneck_smem<256>() <= 163840, "LDS");
  const int grid = B * 256;
  if (cin == 64) {
    static PerDeviceFlag attr_set;
    if (!attr_set.get()) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(bottleneck_kernel<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)bottleneck_smem<64>()) != hipSuccess)
        return -2;
      attr_set.set();
    }
    hipLaunchKernelGGL((bottleneck_kernel<64, true>), dim3(grid), dim3(512), bottleneck_smem<64>(), s, p);
  } else {
    static PerDeviceFlag attr_set;
    if (!attr_set.get()) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(bottleneck_kernel<256, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)bottleneck_smem<256>()) != hipSuccess)
        return -2;
      attr_set.set();
    }
    hipLaunchKernelGGL((bottleneck_kernel<256, false>), dim3(grid), dim3(512), bottleneck_smem<256>(), s, p);
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// host-side packing of the fragment arrays (called once per bottleneck by cotr_load_weights / the op-level test entry):
//   w2 [64][576] (k = (ky*3+kx)*64 + c) -> w2p [2 nb][4 kg][18][64 lanes][4]:  W2[32 nb + (lane & 31)][(18 kg + i) * 8 + (lane >> 5) * 4 + e]
//   w3 [256][64]                        -> w3p [8 w][8 j][64 lanes][4]:         W3[32 w + (lane & 31)][j * 8 + (lane >> 5) * 4 + e]
void bottleneck_pack_w2(const float* w2, float* w2p) {
  for (int nb = 0; nb < 2; ++nb)
    for (int kg = 0; kg < 4; ++kg)
      for (int i = 0; i < 18; ++i)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 4; ++e)
            w2p[((((size_t)(nb * 4 + kg) * 18 + i) * 64 + lane) * 4) + e] =
                w2[(size_t)(32 * nb + (lane & 31)) * 576 + (18 * kg + i) * 8 + (lane >> 5) * 4 + e];
}
void bottleneck_pack_w3(const float* w3, float* w3p) {
  for (int w = 0; w < 8; ++w)
    for (int j = 0; j < 8; ++j)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 4; ++e)
          w3p[(((size_t)(w * 8 + j) * 64 + lane) * 4) + e] = w3[(size_t)(32 * w + (lane & 31)) * 64 + j * 8 + (lane >> 5) * 4 + e];
}
