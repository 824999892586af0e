// Device-side post-processing of the dense initial pass.  The reference pulls the
// [256,512,2] answer of every patch pair to the host and there (COTR/inference/inference_helper.py)
//   :137-145  composes the answer with itself through torch's grid_sample -> per-pixel cycle error,
//             re-centres x of both halves, appends the error as third channel
//   :150-158  moves the two halves to image coordinates (3-point affine of the patch corners)
//   :159-160  resizes each half to its patch with Pillow's mode-'F' BILINEAR (COTR/utils/utils.py:69-83)
//   :61-75    merges the patches of an image: per pixel the patch with the lowest cycle error wins
// Here: one launch for the first two steps (dense_cycle_kernel) and one launch per image for the last two
// (dense_merge_kernel); only the merged [H,W,2] flow and [H,W] error map go back to the host.
//
// Float/byte work, HBM/L2-bound, a few MB: no MFMA.  The Pillow part is bit-exact (32-bit float path of
// src/libImaging/Resample.c: un-quantised double coefficients, double accumulation in tap order, float
// intermediate after the horizontal pass); compiled with -ffp-contract=off for that.  The grid_sample + norm part
// reproduces torch's CPU kernels bit for bit (their association and their FMA contractions, spelled out with explicit
// round-to-nearest intrinsics), so the `con < 0.02` mask and the random draw behind it are the reference's.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NET_H 256
#define NET_W 512

// ---- step 1: cycle error + re-centring + affine -------------------------------------------------------
// pred  [P][256][512][2]  network answer for the query grid q(i,j) = (j/512, i/256)
// aff   [P][2][6]         row-major 2x3 affine per pair: [0] for the left half (into image b), [1] for the right half
// maps  [P][256][512][3]  (x, y, cycle error)
__global__ __launch_bounds__(256) void dense_cycle_kernel(const float* __restrict__ pred, const double* __restrict__ aff,
                                                          float* __restrict__ maps, int n_pairs) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_pairs * NET_H * NET_W) return;
  const int j = idx % NET_W, i = (idx / NET_W) % NET_H, p = idx / (NET_W * NET_H);
  const float* g = pred + (size_t)p * NET_H * NET_W * 2;
  // out_grid = out * 2 - 1  (:138)
  const float gx = __fsub_rn(__fmul_rn(g[(i * NET_W + j) * 2], 2.f), 1.f);
  const float gy = __fsub_rn(__fmul_rn(g[(i * NET_W + j) * 2 + 1], 2.f), 1.f);
  // grid_sample(out_grid as a 2-channel image, out_grid), bilinear / zeros / align_corners=False, in the association of torch's
  // CPU kernel (aten/src/ATen/native/cpu/GridSamplerKernel.cpp, the path the reference runs: the maps are host tensors there):
  //   x = (g + 1) * (size / 2) - 0.5 ; w = x - floor(x), e = 1 - w (likewise n, s) ; weights nw = s*e, ne = s*w, sw = n*e, se = n*w ;
  //   a neighbour outside the map contributes the VALUE 0 (its weight is still multiplied: a NaN / inf coordinate gives NaN) ;
  //   nw_val*nw + ne_val*ne + sw_val*sw + se_val*se evaluated left to right with every multiply-add contracted into one FMA
  //   (that is what the shipped AVX2 / AVX512 builds do - pinned bit for bit against torch CPU by tests/test_zoom_engine_cpu.py
  //   on the C restatement oracle/dense_cycle_ref.c and by the GPU tests on this kernel).
  const float ix = __fsub_rn(__fmul_rn(__fadd_rn(gx, 1.f), 0.5f * NET_W), 0.5f);
  const float iy = __fsub_rn(__fmul_rn(__fadd_rn(gy, 1.f), 0.5f * NET_H), 0.5f);
  const float fx = floorf(ix), fy = floorf(iy);
  const float ww = __fsub_rn(ix, fx), we = __fsub_rn(1.f, ww);
  const float wn = __fsub_rn(iy, fy), wsth = __fsub_rn(1.f, wn);
  const float w[4] = {__fmul_rn(wsth, we), __fmul_rn(wsth, ww), __fmul_rn(wn, we), __fmul_rn(wn, ww)};   // nw, ne, sw, se
  // bounds on the float coordinates (a NaN or a coordinate beyond the int range is outside, as torch's converted index is)
  const bool x_in[2] = {fx >= 0.f && fx <= (float)(NET_W - 1), fx >= -1.f && fx <= (float)(NET_W - 2)};
  const bool y_in[2] = {fy >= 0.f && fy <= (float)(NET_H - 1), fy >= -1.f && fy <= (float)(NET_H - 2)};
  const int x0 = x_in[0] || x_in[1] ? (int)fx : 0, y0 = y_in[0] || y_in[1] ? (int)fy : 0;
  float cx = 0.f, cy = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float vx = 0.f, vy = 0.f;
    if (x_in[c & 1] && y_in[c >> 1]) {
      const float* v = g + ((y0 + (c >> 1)) * NET_W + x0 + (c & 1)) * 2;
      vx = __fsub_rn(__fmul_rn(v[0], 2.f), 1.f);
      vy = __fsub_rn(__fmul_rn(v[1], 2.f), 1.f);
    }
    cx = c == 0 ? __fmul_rn(vx, w[0]) : __fmaf_rn(vx, w[c], cx);
    cy = c == 0 ? __fmul_rn(vy, w[0]) : __fmaf_rn(vy, w[c], cy);
  }
  // in_grid = q * 2 - 1 with q = (j/512, i/256): exact in fp32.  torch.norm(., dim=-1) over the two components on the CPU:
  // acc = dx*dx ; acc = fma(dy, dy, acc) ; sqrt (correctly rounded)
  const float dx = __fsub_rn(cx, (float)j / NET_W * 2.f - 1.f);
  const float dy = __fsub_rn(cy, (float)i / NET_H * 2.f - 1.f);
  // (float)sqrt((double)x) is the correctly rounded float square root (53 >= 2*24 + 2 bits: the double rounding is harmless);
  // __fsqrt_rn is v_sqrt_f32 in this toolchain (1 ulp), sqrtf depends on a compiler default
  const float err = (float)sqrt((double)__fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
  // :140-142  x of the left half lives in the right image and vice versa
  const int half = j >= NET_W / 2;
  const float X = half ? gx * 2.f + 1.f : gx * 2.f - 1.f;
  // :157-158  c[..., :2] @ T[:2,:2] + T[:,2] in double, stored as float
  const double* T = aff + ((size_t)p * 2 + half) * 6;
  const double ox = ((double)X * T[0] + (double)gy * T[3]) + T[2];
  const double oy = ((double)X * T[1] + (double)gy * T[4]) + T[5];
  float* o = maps + (size_t)idx * 3;
  o[0] = (float)ox;
  o[1] = (float)oy;
  o[2] = err;
}

// ---- step 2: Pillow mode-'F' bilinear resize of every patch + merge ------------------------------------
__device__ __forceinline__ double tri(double x) {
  if (x < 0.0) x = -x;
  return x < 1.0 ? 1.0 - x : 0.0;
}

// taps of output index xx when resizing NET_H (=256) samples to `out` samples: first tap, tap count, 1/ww
struct Taps {
  int lo, n;
  double center, ss, ww;
};

__device__ __forceinline__ Taps taps_for(int out, int xx) {
  Taps t;
  const double scale = (double)NET_H / (double)out;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  t.ss = 1.0 / filterscale;
  t.center = ((double)xx + 0.5) * scale;
  int lo = (int)(t.center - support + 0.5);
  if (lo < 0) lo = 0;
  int hi = (int)(t.center + support + 0.5);
  if (hi > NET_H) hi = NET_H;
  t.lo = lo;
  t.n = hi - lo;
  t.ww = 0.0;
  for (int x = 0; x < t.n; ++x) t.ww += tri(((double)(x + lo) - t.center + 0.5) * t.ss);
  return t;
}

__device__ __forceinline__ double tap_weight(const Taps& t, int x) {
  double w = tri(((double)(x + t.lo) - t.center + 0.5) * t.ss);
  if (t.ww != 0.0) w /= t.ww;
  return w;
}

// maps   [P][256][512][3] from step 1;  side 0: left halves (image a), 1: right halves (image b)
// boxes  [P][3] (x, y, size) of the patch each entry covers in this image
// flow   [H][W][2], conf [H][W]
__global__ __launch_bounds__(256) void dense_merge_kernel(const float* __restrict__ maps, const int32_t* __restrict__ boxes,
                                                          int n_pairs, int side, int H, int W, float* __restrict__ flow,
                                                          float* __restrict__ conf) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= H * W) return;
  const int px = idx % W, py = idx / W;
  float best = 100.f, fx = 0.f, fy = 0.f;               // merge_flow_patches :62-64
  for (int p = 0; p < n_pairs; ++p) {
    const int bx = boxes[p * 3], by = boxes[p * 3 + 1], size = boxes[p * 3 + 2];
    const bool inside = px >= bx && px < bx + size && py >= by && py < by + size;
    float v[3] = {0.f, 0.f, 100.f};
    if (inside) {
      const float* src = maps + ((size_t)p * NET_H * NET_W + side * (NET_W / 2)) * 3;   // [256][256][3], row stride 512*3
      const int xx = px - bx, yy = py - by;
      if (size == NET_H) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = src[((size_t)yy * NET_W + xx) * 3 + c];
      } else {
        const Taps tx = taps_for(size, xx), ty = taps_for(size, yy);
        double acc[3] = {0.0, 0.0, 0.0};
        for (int y = 0; y < ty.n; ++y) {
          const double ky = tap_weight(ty, y);
          double row[3] = {0.0, 0.0, 0.0};
          const float* r = src + (size_t)(ty.lo + y) * NET_W * 3;
          for (int x = 0; x < tx.n; ++x) {
            const double kx = tap_weight(tx, x);
#pragma unroll
            for (int c = 0; c < 3; ++c) row[c] += (double)r[(tx.lo + x) * 3 + c] * kx;
          }
#pragma unroll
          for (int c = 0; c < 3; ++c) acc[c] += (double)(float)row[c] * ky;      // float intermediate image
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (float)acc[c];
      }
    }
    // np.stack([temp, confidence]).argmin(-1) == 0: temp <= confidence, a NaN counts as the minimum
    const bool take = (v[2] != v[2]) || (best == best && v[2] <= best);
    if (take) {
      best = v[2];
      fx = v[0];
      fy = v[1];
    }
  }
  flow[(size_t)idx * 2] = fx;
  flow[(size_t)idx * 2 + 1] = fy;
  conf[idx] = best;
}

// ---- generic Pillow mode-'F' BILINEAR resize of a [Hs,Ws,C] float map to [Hd,Wd,C] (utils.float_image_resize) ----
struct GTaps {
  int lo, n;
  double center, ss, ww;
};

__device__ __forceinline__ GTaps gtaps_for(int in_size, int out_size, int xx) {
  GTaps t;
  const double scale = (double)in_size / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  t.ss = 1.0 / filterscale;
  t.center = ((double)xx + 0.5) * scale;
  int lo = (int)(t.center - support + 0.5);
  if (lo < 0) lo = 0;
  int hi = (int)(t.center + support + 0.5);
  if (hi > in_size) hi = in_size;
  t.lo = lo;
  t.n = hi - lo;
  t.ww = 0.0;
  for (int x = 0; x < t.n; ++x) t.ww += tri(((double)(x + lo) - t.center + 0.5) * t.ss);
  return t;
}

__device__ __forceinline__ double gtap_weight(const GTaps& t, int x) {
  double w = tri(((double)(x + t.lo) - t.center + 0.5) * t.ss);
  if (t.ww != 0.0) w /= t.ww;
  return w;
}

// Pillow runs the horizontal pass only if the width changes and the vertical pass only if the height changes; the
// intermediate image is float, so rounding the row sums to float in between reproduces it in every case.
__global__ __launch_bounds__(256) void resize_f32_kernel(const float* __restrict__ src, int Hs, int Ws, int C,
                                                         float* __restrict__ dst, int Hd, int Wd) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Hd * Wd) return;
  const int px = idx % Wd, py = idx / Wd;
  const bool horiz = Ws != Wd, vert = Hs != Hd;
  GTaps tx, ty;
  if (horiz) tx = gtaps_for(Ws, Wd, px);
  if (vert) ty = gtaps_for(Hs, Hd, py);
  const int ny = vert ? ty.n : 1, y0 = vert ? ty.lo : py;
  const int nx = horiz ? tx.n : 1, x0 = horiz ? tx.lo : px;
  for (int c = 0; c < C; ++c) {
    double acc = 0.0;
    float last = 0.f;
    for (int y = 0; y < ny; ++y) {
      const float* r = src + ((size_t)(y0 + y) * Ws) * C + c;
      float rowv;
      if (horiz) {
        double row = 0.0;
        for (int x = 0; x < nx; ++x) row += (double)r[(size_t)(x0 + x) * C] * gtap_weight(tx, x);
        rowv = (float)row;
      } else {
        rowv = r[(size_t)x0 * C];
      }
      if (vert) acc += (double)rowv * gtap_weight(ty, y);
      last = rowv;
    }
    dst[(size_t)idx * C + c] = vert ? (float)acc : last;
  }
}

int launch_resize_f32(const float* src, int Hs, int Ws, int C, float* dst, int Hd, int Wd, hipStream_t s) {
  if (Hd <= 0 || Wd <= 0) return 0;
  hipLaunchKernelGGL(resize_f32_kernel, dim3((Hd * Wd + 255) / 256), dim3(256), 0, s, src, Hs, Ws, C, dst, Hd, Wd);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_dense_cycle(const float* pred, const double* aff, float* maps, int n_pairs, hipStream_t s) {
  if (n_pairs <= 0) return 0;
  const int total = n_pairs * NET_H * NET_W;
  hipLaunchKernelGGL(dense_cycle_kernel, dim3((total + 255) / 256), dim3(256), 0, s, pred, aff, maps, n_pairs);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_dense_merge(const float* maps, const int32_t* boxes, int n_pairs, int side, int H, int W, float* flow,
                       float* conf, hipStream_t s) {
  if (H <= 0 || W <= 0) return 0;
  hipLaunchKernelGGL(dense_merge_kernel, dim3((H * W + 255) / 256), dim3(256), 0, s, maps, boxes, n_pairs, side, H, W, flow,
                     conf);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
