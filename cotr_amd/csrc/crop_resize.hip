// Device-side construction of the network input for a batch of zoom-in tasks:
// for every task, crop a square patch around the query in image A and around the current estimate in
// image B, resize both to 256x256 with Pillow's 8-bit BILINEAR resample, put them side by side, convert
// to float and ImageNet-normalise -> float32 [N,3,256,512].  Replaces, per task, the host-side
//   PIL.Image.fromarray(patch).resize((256,256), BILINEAR) x2, two_images_side_by_side, to_tensor, normalize
// of COTR/inference/refinement_task.py:105-120 (and the 1.5 MB H2D copy per task that follows it).
//
// Bit-exact with Pillow (src/libImaging/Resample.c, 8-bit path): separable, horizontal pass first with the
// result rounded to uint8, double-precision triangle coefficients whose support scales with the down-scale
// factor, quantised to 22 fractional bits, int32 accumulation with a rounding bias, clip to [0,255]; then
// x/255, -mean, /std as separate IEEE fp32 operations like torchvision's to_tensor + normalize.
// Integer/byte work, HBM/L2-bound: no MFMA here.  Compiled with -ffp-contract=off.
//
// One workgroup = R output rows x 256 columns of one half of one task: the horizontal pass of the input
// rows those R output rows need goes to LDS (uchar4 per pixel), the vertical pass reads it back.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

#define OUT 256
#define PRECISION_BITS 22

struct CropParams {
  const uint8_t* img[2];  // HWC uint8, 3 channels
  int h[2], w[2];
  const int32_t* boxes;   // [N][6]: xa, ya, size_a, xb, yb, size_b
  float* out;             // [N][3][256][512]
  int rows_per_wg;        // R
  int max_rows;           // LDS rows available for the horizontal-pass result
  int ksize_max;          // taps reserved per output column / row
};

__device__ __forceinline__ double tri(double x) {
  if (x < 0.0) x = -x;
  return x < 1.0 ? 1.0 - x : 0.0;
}

// Pillow precompute_coeffs + normalize_coeffs_8bpc for one output index (box = [0, in_size))
__device__ void coeffs_for(int in_size, int xx, int32_t* k, int& xmin_out, int& xmax_out) {
  const double scale = (double)in_size / (double)OUT;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  const double ss = 1.0 / filterscale;
  const double center = ((double)xx + 0.5) * scale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) ww += tri(((double)(x + xmin) - center + 0.5) * ss);
  for (int x = 0; x < xmax; ++x) {
    double w = tri(((double)(x + xmin) - center + 0.5) * ss);
    if (ww != 0.0) w /= ww;
    const double v = w * (double)(1 << PRECISION_BITS);
    k[x] = w < 0.0 ? (int32_t)(-0.5 + v) : (int32_t)(0.5 + v);
  }
  xmin_out = xmin;
  xmax_out = xmax;
}

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(256) void crop_resize_kernel(const CropParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x;  // output column
  const int side = blockIdx.y, task = blockIdx.z;
  const int R = p.rows_per_wg;
  const int yy0 = blockIdx.x * R;
  const int32_t* box = p.boxes + (size_t)task * 6 + side * 3;
  const int bx = box[0], by = box[1], size = box[2];
  const uint8_t* img = p.img[side];
  const int W = p.w[side];

  // LDS carve: tmp[max_rows][256] uchar4 | kh[256][ksize_max] int32 | kv[R][ksize_max] int32 | vb[R][2] int32
  uchar4* tmp = reinterpret_cast<uchar4*>(smem);
  int32_t* kh = reinterpret_cast<int32_t*>(smem + (size_t)p.max_rows * OUT * 4);
  int32_t* kv = kh + OUT * p.ksize_max;
  int32_t* vb = kv + R * p.ksize_max;

  // vertical coefficients of this workgroup's R output rows (threads 0..R-1), horizontal ones per column
  if (t < R) {
    int ymin, ymax;
    coeffs_for(size, yy0 + t, kv + t * p.ksize_max, ymin, ymax);
    vb[2 * t] = ymin;
    vb[2 * t + 1] = ymax;
  }
  int xmin, xmax;
  int32_t* my_kh = kh + t * p.ksize_max;
  coeffs_for(size, t, my_kh, xmin, xmax);
  __syncthreads();
  const int r0 = vb[0];
  const int r1 = vb[2 * (R - 1)] + vb[2 * (R - 1) + 1];  // one past the last input row needed

  // horizontal pass: input rows r0..r1-1 of the crop -> tmp (uint8 like Pillow's intermediate image)
  if (size == OUT) {
    for (int r = r0; r < r1; ++r) {
      const uint8_t* px = img + ((size_t)(by + r) * W + bx + t) * 3;
      tmp[(r - r0) * OUT + t] = make_uchar4(px[0], px[1], px[2], 0);
    }
  } else {
    for (int r = r0; r < r1; ++r) {
      const uint8_t* row = img + ((size_t)(by + r) * W + bx + xmin) * 3;
      int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
      for (int x = 0; x < xmax; ++x) {
        const int k = my_kh[x];
        a0 += row[3 * x] * k;
        a1 += row[3 * x + 1] * k;
        a2 += row[3 * x + 2] * k;
      }
      tmp[(r - r0) * OUT + t] = make_uchar4((unsigned char)clip8(a0), (unsigned char)clip8(a1), (unsigned char)clip8(a2), 0);
    }
  }
  __syncthreads();

  // vertical pass + to_tensor + normalize
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float stdv[3] = {0.229f, 0.224f, 0.225f};
  float* outp = p.out + (size_t)task * 3 * OUT * 2 * OUT + side * OUT + t;
  for (int i = 0; i < R; ++i) {
    const int yy = yy0 + i;
    int v0, v1, v2;
    if (size == OUT) {
      const uchar4 q = tmp[(yy - r0) * OUT + t];
      v0 = q.x; v1 = q.y; v2 = q.z;
    } else {
      const int ymin = vb[2 * i], ymax = vb[2 * i + 1];
      const int32_t* k = kv + i * p.ksize_max;
      int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
      for (int y = 0; y < ymax; ++y) {
        const uchar4 q = tmp[(ymin - r0 + y) * OUT + t];
        a0 += q.x * k[y];
        a1 += q.y * k[y];
        a2 += q.z * k[y];
      }
      v0 = clip8(a0); v1 = clip8(a1); v2 = clip8(a2);
    }
    const int v[3] = {v0, v1, v2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float f = __fdiv_rn((float)v[c], 255.0f);
      outp[((size_t)c * OUT + yy) * (2 * OUT)] = __fdiv_rn(__fsub_rn(f, mean[c]), stdv[c]);
    }
  }
}

// Host launcher.  max_size = largest crop edge among the boxes (host knows it: it computed the boxes).
int launch_crop_resize(const uint8_t* img_a, int ha, int wa, const uint8_t* img_b, int hb, int wb,
                       const int32_t* boxes, int n, float* out, int max_size, hipStream_t s) {
  if (n <= 0) return 0;
  if (max_size < 2 || max_size > 16384) return -1;
  const double scale = max_size > OUT ? (double)max_size / OUT : 1.0;
  const int sup = (int)ceil(scale);
  const int ksize = sup * 2 + 1;
  int R = 8;
  size_t bytes = 0;
  int max_rows = 0;
  for (; R >= 1; R >>= 1) {
    max_rows = (int)ceil(R * scale) + 2 * sup + 3;
    bytes = (size_t)max_rows * OUT * 4 + (size_t)OUT * ksize * 4 + (size_t)R * ksize * 4 + (size_t)R * 8;
    if (bytes <= 160 * 1024) break;
  }
  if (R < 1) return -1;
  static PerDeviceFlag attr_set;
  if (!attr_set.get()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(crop_resize_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(160 * 1024)) != hipSuccess)
      return -2;
    attr_set.set();
  }
  CropParams p;
  p.img[0] = img_a; p.img[1] = img_b;
  p.h[0] = ha; p.h[1] = hb; p.w[0] = wa; p.w[1] = wb;
  p.boxes = boxes; p.out = out;
  p.rows_per_wg = R; p.max_rows = max_rows; p.ksize_max = ksize;
  hipLaunchKernelGGL(crop_resize_kernel, dim3(OUT / R, 2, n), dim3(256), bytes, s, p);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
