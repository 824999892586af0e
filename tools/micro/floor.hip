// Micro-benchmark: cost of a dependent chain of tiny kernels (graph-captured), gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_empty() {}
__global__ void k_copy(const float* a, float* b) { b[blockIdx.x * 256 + threadIdx.x] = a[blockIdx.x * 256 + threadIdx.x] + 1.f; }
__global__ void k_copy4(const float4* a, float4* b) { float4 v = a[blockIdx.x * 256 + threadIdx.x]; v.x += 1.f; b[blockIdx.x * 256 + threadIdx.x] = v; }
template <typename F> float run(F launch, int n) {
  hipStream_t s; hipStreamCreate(&s);
  hipGraph_t g; hipGraphExec_t e;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < n; ++i) launch(s, i);
  hipStreamEndCapture(s, &g); hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipGraphLaunch(e, s); hipStreamSynchronize(s);
  float best = 1e9;
  for (int r = 0; r < 5; ++r) { hipEventRecord(a, s); hipGraphLaunch(e, s); hipEventRecord(b, s); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
  return best * 1000.f / n;
}
int main() {
  float *x, *y; hipMalloc(&x, 64 << 20); hipMalloc(&y, 64 << 20); hipMemset(x, 0, 64 << 20);
  for (int wgs : {1, 256, 1024}) {
    printf("empty  %4d wgs: %.2f us/launch\n", wgs, run([&](hipStream_t s, int) { hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(256), 0, s); }, 100));
    printf("copy   %4d wgs: %.2f us/launch (ping-pong dependent)\n", wgs, run([&](hipStream_t s, int i) { hipLaunchKernelGGL(k_copy, dim3(wgs), dim3(256), 0, s, (i & 1) ? y : x, (i & 1) ? x : y); }, 100));
    printf("copy4  %4d wgs: %.2f us/launch\n", wgs, run([&](hipStream_t s, int i) { hipLaunchKernelGGL(k_copy4, dim3(wgs), dim3(256), 0, s, (const float4*)((i & 1) ? y : x), (float4*)((i & 1) ? x : y)); }, 100));
  }
  // eager (no graph) chain
  hipStream_t s; hipStreamCreate(&s); hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a, s);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, s, (i & 1) ? y : x, (i & 1) ? x : y);
    hipEventRecord(b, s); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
    printf("eager copy 256 wgs: %.2f us/launch\n", ms * 1000.f / 200);
  }
  return 0;
}
