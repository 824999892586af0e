// What v_permlane32_swap returns (gfx950): lane id in, both results out.   hipcc --offload-arch=gfx950 -O3 permlane_probe.hip -o permlane_probe.exe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
  const unsigned v = threadIdx.x, w = 100 + threadIdx.x;
  const auto r = __builtin_amdgcn_permlane32_swap(v, w, false, false);
  o[threadIdx.x] = r[0];
  o[64 + threadIdx.x] = r[1];
  const auto r2 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  o[128 + threadIdx.x] = r2[0];
  o[192 + threadIdx.x] = r2[1];
}
int main() {
  unsigned* d;
  hipMalloc(&d, 1024);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[256];
  hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  const char* names[4] = {"swap(v=lane, w=100+lane)[0]", "swap(v, w)[1]", "swap(v, v)[0]", "swap(v, v)[1]"};
  for (int a = 0; a < 4; ++a) {
    printf("%s:", names[a]);
    for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[a * 64 + i]);
    printf("\n");
  }
  return 0;
}
