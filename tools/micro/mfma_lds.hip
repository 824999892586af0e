// Micro-benchmark: the inner loop of an MFMA wavefront of the large-tile GEMM in isolation - 16 v_mfma_f32_32x32x2_f32 on four
// accumulators per 8-deep K slice, operands from LDS by 4 ds_read_b128 per slice (swizzled [row][32 floats] tile as in
// gemm_big.hip), ONE wavefront per SIMD (256-thread workgroups, 1 per CU), no barrier, no loads: what does the slice cost in
// shader cycles, as a function of where the reads are issued?   hipcc --offload-arch=gfx950 -O3 mfma_lds.hip -o mfma_lds.exe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VARIANT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_loop(float* out, unsigned long long* stamps, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 tiles of (128 + 128) x 32 floats = 64 KB
  const int t = threadIdx.x, lane = t & 63, wave = (t >> 6) & 3;
  for (int i = t; i < 2 * 256 * 32; i += WAVES * 64) smem[i] = (float)((i * 2654435761u) >> 8 & 0xffff) * (1.f / 65536.f) - 0.5f;
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hh = lane >> 5, sw = (l31 >> 1) & 7;
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  struct Frag { f32x4 a[2], b[2]; };
  auto load_frag = [&](int f) {
    Frag r;
    const float* As = smem + (f >> 2) * (256 * 32) + (wm * 64 + l31) * 32;
    const float* Ws = smem + (f >> 2) * (256 * 32) + 128 * 32 + (wn * 64 + l31) * 32;
    const int ch = (((f & 3) * 2 + hh) ^ sw) * 4;
    for (int a = 0; a < 2; ++a) r.a[a] = *reinterpret_cast<const f32x4*>(As + a * 32 * 32 + ch);
    for (int b = 0; b < 2; ++b) r.b[b] = *reinterpret_cast<const f32x4*>(Ws + b * 32 * 32 + ch);
    return r;
  };
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (VARIANT == 0) {          // reads of a slice right in front of its MFMAs (what a naive loop does)
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        Frag cur = load_frag(f);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[a][e], cur.b[b][e], acc[a][b], 0, 0, 0);
      }
    } else if constexpr (VARIANT == 1) {   // reads one slice ahead, issued in a block
      Frag cur = load_frag(0);
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        Frag nxt = cur;
        if (f + 1 < 8) nxt = load_frag(f + 1);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[a][e], cur.b[b][e], acc[a][b], 0, 0, 0);
        cur = nxt;
      }
    } else if constexpr (VARIANT == 2) {   // MFMAs only (operands loaded once): the floor
      Frag cur = load_frag(it & 7);
#pragma unroll
      for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[a][(e + f) & 3], cur.b[b][e], acc[a][b], 0, 0, 0);
    } else {                               // reads one slice ahead, pinned one by one behind the first MFMAs
      Frag cur = load_frag(0);
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        Frag nxt = cur;
        if (f + 1 < 8) nxt = load_frag(f + 1);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[a][e], cur.b[b][e], acc[a][b], 0, 0, 0);
        if (f + 1 < 8) {
          for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
          __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        }
        cur = nxt;
      }
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[blockIdx.x * WAVES * 64 + t] = s;
  if (t == 0) stamps[blockIdx.x] = c1 - c0;
}

template <int VARIANT, int WAVES>
void run(const char* name, float* out, unsigned long long* st) {
  const int wgs = 256, iters = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_loop<VARIANT, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_loop<VARIANT, WAVES>), dim3(wgs), dim3(WAVES * 64), 65536, 0, out, st, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> s(wgs);
    hipMemcpy(s.data(), st, wgs * 8, hipMemcpyDeviceToHost);
    double cyc = 0; for (auto v : s) cyc += v;
    const double mf = (double)iters * 128;
    if (rep == 1)
      printf("%-64s %d waves/WG: %6.1f cycles per MFMA per wavefront, %6.1f TFLOP/s\n", name, WAVES, cyc / wgs / mf,
             (double)wgs * WAVES * mf * 4096.0 / ms / 1e9);
  }
}

int main() {
  float* out; unsigned long long* st;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&st, 256 * 8);
  run<2, 4>("MFMAs only", out, st);
  run<0, 4>("4 ds_read_b128 in front of each slice's 16 MFMAs", out, st);
  run<1, 4>("reads one slice ahead, in a block", out, st);
  run<3, 4>("reads one slice ahead, one by one behind the first MFMAs", out, st);
  run<2, 8>("MFMAs only", out, st);
  run<0, 8>("4 ds_read_b128 in front of each slice's 16 MFMAs", out, st);
  run<1, 8>("reads one slice ahead, in a block", out, st);
  run<3, 8>("reads one slice ahead, one by one behind the first MFMAs", out, st);
  return 0;
}
