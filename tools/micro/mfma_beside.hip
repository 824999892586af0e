// Micro-benchmark: what does work of ANOTHER wavefront on the same SIMD cost an MFMA wavefront?  (round 4: the persistent ping-pong
// GEMM hides a tile's epilogue "beside" the other group's MFMAs - and measured that the epilogue still costs 3.8 us of MFMA time per
// 128 x 128 tile although no wavefront ever waits for it.)  8 wavefronts per workgroup, one workgroup per CU: wavefronts 0-3 run the
// large tile's MFMA loop (16 v_mfma_f32_32x32x2_f32 per 8-deep slice, fragments from LDS one slice ahead: 64.0 cycles per MFMA alone);
// wavefronts 4-7 (one per SIMD) run `n_other` iterations of a neighbour loop of KIND:
//   0 nothing   1 independent VALU FMAs (32 per iteration)   2 ds_read_b128 + 8 VALU   3 ds_write_b32 x 8   4 dependent VALU chain (32)
//   5 global_store_dwordx4 x 2 + 8 VALU   6 SALU only (32 s_add)
// and the MFMA wavefronts report their cycles per MFMA.     hipcc --offload-arch=gfx950 -O3 mfma_beside.hip -o mfma_beside.exe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(512) void k_beside(float* out, unsigned long long* stamps, int iters, int n_other, int prio) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 tiles of 256 x 32 floats + 16 KB neighbour area
  const int t = threadIdx.x, lane = t & 63, wave8 = t >> 6, wave = wave8 & 3;
  for (int i = t; i < 2 * 256 * 32 + 4096; i += 512) smem[i] = (float)((i * 2654435761u) >> 8 & 0xffff) * (1.f / 65536.f) - 0.5f;
  __syncthreads();
  if (wave8 >= 4) {
    // ---- the neighbour ----
    if (prio) __builtin_amdgcn_s_setprio(0);
    float* mine = smem + 2 * 256 * 32 + wave * 1024;
    float v[32];
    for (int i = 0; i < 32; ++i) v[i] = mine[lane + (i & 7) * 64] + i;
    unsigned long long n_instr = 0;
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < n_other; ++it) {
      if constexpr (KIND == 1) {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 31]), "v"(v[(i + 2) & 31]));
        n_instr += 32;
      } else if constexpr (KIND == 2) {
        f32x4 r = *reinterpret_cast<volatile f32x4*>(mine + ((lane * 4 + it * 4) & 1020));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(r[i & 3]), "v"(v[(i + 2) & 31]));
        n_instr += 9;
      } else if constexpr (KIND == 3) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<volatile float*>(mine + ((lane + i * 68 + it) & 1023)) = v[i];
        n_instr += 8;
      } else if constexpr (KIND == 4) {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[0]));
        n_instr += 32;
      } else if constexpr (KIND == 5) {
        f32x4 r = {v[0], v[1], v[2], v[3]};
        float* dst = out + 1048576 + ((size_t)blockIdx.x * 4 + wave) * 16384 + ((it & 15) * 2) * 256 + lane * 4;
        *reinterpret_cast<f32x4*>(dst) = r;
        *reinterpret_cast<f32x4*>(dst + 256) = r;
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 31]), "v"(v[(i + 2) & 31]));
        n_instr += 10;
      } else if constexpr (KIND == 7) {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("v_exp_f32 %0, %1" : "=v"(v[i]) : "v"(v[(i + 1) & 31]));
        n_instr += 32;
      } else if constexpr (KIND == 8) {   // the softmax mix of att_rows.hip per 16 scores: 5 max3, 16 sub, 16 exp, 15 add
#pragma unroll
        for (int i = 0; i < 5; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[i + 8]), "v"(v[i + 16]));
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[31]));
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
#pragma unroll
        for (int i = 0; i < 15; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[16 + (i & 7)]) : "v"(v[i]));
        n_instr += 52;
      } else if constexpr (KIND == 6) {
        int s = it;
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s));
        if (s == 123456789) v[0] += 1.f;
        n_instr += 32;
      }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += v[i];
    out[blockIdx.x * 512 + t] = s;
    if (lane == 0) { stamps[(blockIdx.x * 8 + wave8) * 2] = c1 - c0; stamps[(blockIdx.x * 8 + wave8) * 2 + 1] = n_instr; }
    return;
  }
  // ---- the MFMA wavefront ----
  if (prio) __builtin_amdgcn_s_setprio(2);
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
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[blockIdx.x * 512 + t] = s;
  if (lane == 0) { stamps[(blockIdx.x * 8 + wave8) * 2] = c1 - c0; stamps[(blockIdx.x * 8 + wave8) * 2 + 1] = (unsigned long long)iters * 128; }
}

template <int KIND>
void run(const char* name, float* out, unsigned long long* st, int prio) {
  const int wgs = 256, iters = 400;
  const size_t smem = (2 * 256 * 32 + 4096) * sizeof(float);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_beside<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  for (int n_other : {0, 200, 800, 3200}) {
    if (KIND == 0 && n_other) continue;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_beside<KIND>, dim3(wgs), dim3(512), smem, 0, out, st, iters, n_other, prio);
    hipDeviceSynchronize();
    std::vector<unsigned long long> s(wgs * 16);
    hipMemcpy(s.data(), st, wgs * 16 * 8, hipMemcpyDeviceToHost);
    double mc = 0, mn = 0, oc = 0, on = 0;
    for (int b = 0; b < wgs; ++b)
      for (int w = 0; w < 8; ++w) {
        if (w < 4) { mc += s[(b * 8 + w) * 2]; mn += s[(b * 8 + w) * 2 + 1]; }
        else { oc += s[(b * 8 + w) * 2]; on += s[(b * 8 + w) * 2 + 1]; }
      }
    const double base = 64.0;
    printf("%-28s prio %d  neighbour iterations %5d: MFMA wavefront %6.2f cycles per MFMA", name, prio, n_other, mc / mn);
    if (on > 0) printf("  | neighbour: %8.0f instructions in %8.0f cycles (%5.1f each); MFMA cycles lost per neighbour instruction %5.2f", on / (wgs * 4), oc / (wgs * 4), oc / on,
                       (mc / mn - base) * (mn / (wgs * 4)) / (on / (wgs * 4)));
    printf("\n");
  }
}

int main() {
  float* out; unsigned long long* st;
  hipMalloc(&out, (size_t)(1048576 + 256 * 4 * 16384) * 4 + 4096); hipMalloc(&st, 256 * 16 * 8);
  for (int prio = 0; prio < 2; ++prio) {
    run<0>("alone", out, st, prio);
    run<1>("VALU fma, independent", out, st, prio);
    run<4>("VALU fma, dependent chain", out, st, prio);
    run<2>("ds_read_b128 + 8 VALU", out, st, prio);
    run<3>("ds_write_b32", out, st, prio);
    run<5>("2 global_store_x4 + 8 VALU", out, st, prio);
    run<7>("v_exp_f32 x 32", out, st, prio);
    run<8>("softmax mix (max3/sub/exp/add)", out, st, prio);
    run<6>("SALU", out, st, prio);
  }
  return 0;
}
