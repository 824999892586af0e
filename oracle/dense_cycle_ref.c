/* TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) - never linked into or loaded by cotr_amd.
 *
 * Plain-C restatement of the arithmetic behind the cycle-error map of the reference's dense pass,
 * COTR/inference/inference_helper.py:137-139:
 *     out_grid   = out * 2 - 1
 *     cycle_grid = torch.nn.functional.grid_sample(out_grid.permute(0,3,1,2), out_grid)   # bilinear, zeros, align_corners=False
 *     confidence = torch.norm(cycle_grid - in_grid, dim=-1)
 * as torch's CPU kernels evaluate it (the reference runs these three lines on host tensors):
 *   - aten/src/ATen/native/cpu/GridSamplerKernel.cpp, ApplyGridSample<float, 2, Bilinear, Zeros, align_corners=false>:
 *       x = (g + 1) * (W / 2) - 0.5 ; w = x - floor(x) ; e = 1 - w ; (n, s likewise) ; nw = s*e, ne = s*w, sw = n*e, se = n*w ;
 *       out-of-map neighbours are gathered as the value 0 ; nw_val*nw + ne_val*ne + sw_val*sw + se_val*se left to right,
 *       each multiply-add contracted into an FMA by the compiler of the shipped AVX2 / AVX512 builds;
 *   - aten/src/ATen/native/cpu/ReduceOpsKernel.cpp, the p = 2 last-dimension path: acc = dx*dx ; acc = fma(dy, dy, acc) ; sqrt.
 * Which products are contracted is not visible in the source: it was determined by trying the 16 combinations against torch
 * 2.10 CPU (only this one reproduces it, on random, out-of-range, NaN and inf coordinates, under ATEN_CPU_CAPABILITY avx2 and
 * avx512) and is pinned by tests/test_zoom_engine_cpu.py::test_c_cycle_restatement_equals_torch_cpu on every CPU test run.
 * cotr_amd/csrc/dense_post.hip (dense_cycle_kernel) is the device version of exactly this function.
 *
 * Build: gcc -O1 -ffp-contract=off -shared -fPIC (oracle/build_c.py); every rounding step goes through a volatile so that
 * nothing is re-associated or contracted behind the code's back. */
#include <math.h>

#define NET_H 256
#define NET_W 512

/* pred [P][256][512][2] (network answer, 0..1) -> cyc [P][256][512][2] (grid_sample result), err [P][256][512] */
void dense_cycle_ref(const float* pred, float* cyc, float* err, int n_pairs) {
  for (int p = 0; p < n_pairs; ++p) {
    const float* g = pred + (long)p * NET_H * NET_W * 2;
    for (int i = 0; i < NET_H; ++i)
      for (int j = 0; j < NET_W; ++j) {
        volatile float gx = g[(i * NET_W + j) * 2] * 2.f, gy = g[(i * NET_W + j) * 2 + 1] * 2.f;
        gx = gx - 1.f;
        gy = gy - 1.f;
        volatile float tx = gx + 1.f, ty = gy + 1.f;
        volatile float mx = tx * (0.5f * NET_W), my = ty * (0.5f * NET_H);
        const float ix = mx - 0.5f, iy = my - 0.5f;
        const float fx = floorf(ix), fy = floorf(iy);
        volatile float w = ix - fx, n = iy - fy;
        volatile float e = 1.f - w, s = 1.f - n;
        volatile float wt[4];
        wt[0] = s * e; wt[1] = s * w; wt[2] = n * e; wt[3] = n * w;
        const int x_in[2] = {fx >= 0.f && fx <= (float)(NET_W - 1), fx >= -1.f && fx <= (float)(NET_W - 2)};
        const int y_in[2] = {fy >= 0.f && fy <= (float)(NET_H - 1), fy >= -1.f && fy <= (float)(NET_H - 2)};
        const int x0 = (x_in[0] || x_in[1]) ? (int)fx : 0, y0 = (y_in[0] || y_in[1]) ? (int)fy : 0;
        float c[2];
        for (int ch = 0; ch < 2; ++ch) {
          volatile float r = 0.f;
          for (int k = 0; k < 4; ++k) {
            volatile float v = 0.f;
            if (x_in[k & 1] && y_in[k >> 1]) {
              v = g[((y0 + (k >> 1)) * NET_W + x0 + (k & 1)) * 2 + ch] * 2.f;
              v = v - 1.f;
            }
            if (k == 0) r = v * wt[0];
            else r = fmaf(v, wt[k], r);
          }
          c[ch] = r;
        }
        volatile float qx = (float)j / NET_W * 2.f - 1.f, qy = (float)i / NET_H * 2.f - 1.f;
        volatile float dx = c[0] - qx, dy = c[1] - qy;
        volatile float acc = dx * dx;
        acc = fmaf(dy, dy, acc);
        const long o = ((long)p * NET_H + i) * NET_W + j;
        cyc[o * 2] = c[0];
        cyc[o * 2 + 1] = c[1];
        err[o] = sqrtf(acc);
      }
  }
}
