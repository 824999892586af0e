// Micro-benchmark: dependent chains of the library's small kernels (graph-captured), gfx950.
#include "../../cotr_amd/csrc/pointwise.hip"
#include <stdio.h>
__global__ void k_copy4(const float4* a, float4* b, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) { float4 v = a[i]; v.x += 1.f; b[i] = v; } }
__global__ void k_copy1(const float* a, float* b, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) b[i] = a[i] + 1.f; }
template <typename F> float run(F launch, int n) {
  hipStream_t s; (void)hipStreamCreate(&s);
  hipGraph_t g; hipGraphExec_t e;
  (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < n; ++i) launch(s, i);
  (void)hipStreamEndCapture(s, &g); (void)hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  (void)hipGraphLaunch(e, s); (void)hipStreamSynchronize(s);
  float best = 1e9;
  for (int r = 0; r < 5; ++r) { (void)hipEventRecord(a, s); (void)hipGraphLaunch(e, s); (void)hipEventRecord(b, s); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
  return best * 1000.f / n;
}
int main() {
  float *x, *y, *w; (void)hipMalloc(&x, 64 << 20); (void)hipMalloc(&y, 64 << 20); (void)hipMalloc(&w, 4096);
  (void)hipMemset(x, 0, 64 << 20); (void)hipMemset(y, 0, 64 << 20); (void)hipMemset(w, 0, 4096);
  for (int rows : {32, 512, 1000, 16384}) {
    printf("layernorm rows=%5d: %.2f us/launch\n", rows, run([&](hipStream_t s, int i) { launch_layernorm((i & 1) ? y : x, w, w + 256, (i & 1) ? x : y, rows, s); }, 100));
    int n4 = rows * 64;
    printf("copy4     rows=%5d: %.2f us/launch\n", rows, run([&](hipStream_t s, int i) { hipLaunchKernelGGL(k_copy4, dim3((n4 + 255) / 256), dim3(256), 0, s, (const float4*)((i & 1) ? y : x), (float4*)((i & 1) ? x : y), n4); }, 100));
    printf("copy1     rows=%5d: %.2f us/launch\n", rows, run([&](hipStream_t s, int i) { hipLaunchKernelGGL(k_copy1, dim3((n4 * 4 + 255) / 256), dim3(256), 0, s, (i & 1) ? y : x, (i & 1) ? x : y, n4 * 4); }, 100));
  }
  printf("posenc 1000: %.2f us\n", run([&](hipStream_t s, int i) { launch_posenc(x, y, 1, 1000, 1000, s); }, 50));
  printf("head2 1000: %.2f us\n", run([&](hipStream_t s, int i) { launch_head2(x, w, w, y, 1, 1000, 1000, s); }, 50));
  printf("maxpool B=1: %.2f us\n", run([&](hipStream_t s, int i) { launch_maxpool(x, y, 1, 128, 128, 64, s); }, 50));
  return 0;
}
