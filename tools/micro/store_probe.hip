// Write-only bandwidth by store pattern (268 MB = 262144 rows x 256 floats): what does the epilogue's store shape cost?
//   0  float4, lane-linear (1 KB contiguous per wave instruction: torch's fill)
//   1  dword, lane&31 = column, lane>>5 = row +4: two 128-B row segments per instruction (the MFMA D layout stored directly:
//      conv23.hip / expand.hip), 16 instructions = 32 rows x 32 columns
//   2  float4, 8 lanes per row segment of 128 B: eight rows per instruction (gemm_big.hip's epilogue after LDS staging, 32-column block)
//   3  float4, 16 lanes per 256-B row segment, four rows per instruction
//   4  = 1, but ~2 us of sleep between the 128-B column blocks of a row (how the epilogues of expand.hip / conv23.hip see time pass
//      between the pieces of one output row);  5 = 4 with three workgroups per CU started 0 / 1/3 / 2/3 of a block time apart
//   hipcc --offload-arch=gfx950 -O3 store_probe.hip -o store_probe.exe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k_store(float* y, int ld) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const size_t row0 = (size_t)blockIdx.x * 128 + wave * 32;      // a wavefront owns 32 rows x ld columns, like the kernels
  const float v = (float)lane;
  if (MODE == 0) {
    f32x4 q = {v, v, v, v};
    float* base = y + row0 * ld;
    for (int i = 0; i < 32 * ld / 256; ++i) *reinterpret_cast<f32x4*>(base + i * 256 + lane * 4) = q;
  } else if (MODE == 1) {
    for (int cb = 0; cb < ld / 32; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) y[(row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * ld + cb * 32 + (lane & 31)] = v;
  } else if (MODE == 4 || MODE == 5) {
    if (MODE == 5) for (int i = 0; i < (int)(blockIdx.x >> 8) % 3; ++i) __builtin_amdgcn_s_sleep(127);
    for (int cb = 0; cb < ld / 32; ++cb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) y[(row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * ld + cb * 32 + (lane & 31)] = v;
      __builtin_amdgcn_s_sleep(40);
    }
  } else if (MODE == 2) {
    f32x4 q = {v, v, v, v};
    for (int cb = 0; cb < ld / 32; ++cb)
#pragma unroll
      for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(y + (row0 + it * 8 + (lane >> 3)) * ld + cb * 32 + (lane & 7) * 4) = q;
  } else {
    f32x4 q = {v, v, v, v};
    for (int cb = 0; cb < ld / 64; ++cb)
#pragma unroll
      for (int it = 0; it < 8; ++it) *reinterpret_cast<f32x4*>(y + (row0 + it * 4 + (lane >> 4)) * ld + cb * 64 + (lane & 15) * 4) = q;
  }
}
template <int MODE>
static void run(const char* tag, float* y, int rows, int ld) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_store<MODE>, dim3(rows / 128), dim3(256), 0, 0, y, ld);
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_store<MODE>, dim3(rows / 128), dim3(256), 0, 0, y, ld);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 20, mb = (double)rows * ld * 4 / 1e6;
  printf("%-64s ld %4d: %7.1f us  %5.2f TB/s\n", tag, ld, us, mb / us);
}
int main() {
  float* y;
  (void)hipMalloc(&y, (size_t)262144 * 256 * 4);
  for (int ld : {256, 64, 512}) {
    const int rows = 262144 * 256 / ld;
    run<0>("float4 lane-linear", y, rows, ld);
    run<1>("dword, D layout direct (2 x 128 B per instruction)", y, rows, ld);
    run<2>("float4, 8 lanes per 128-B row segment (8 rows per instruction)", y, rows, ld);
    run<3>("float4, 16 lanes per 256-B row segment (4 rows per instruction)", y, rows, ld);
    run<4>("dword D layout, 1.1 us of sleep between a row's column blocks", y, rows, ld);
    run<5>("  ... and staggered workgroups", y, rows, ld);
  }
  return 0;
}
