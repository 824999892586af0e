// Model of att_rows.hip's K/V phase without memory: every wavefront repeats [32 v_mfma_f32_32x32x2_f32 on two alternating accumulators,
// then a softmax-like VALU block on the results (5 max3, 16 sub, 16 exp, 15 add per accumulator), then 32 more matrix instructions that
// take the VALU results as operands].  One workgroup per CU; 4 wavefronts (one per SIMD) or 8 (two per SIMD, the second one optionally
// started `stagger` x 64 cycles late).  Question: do two such wavefronts on a SIMD fill each other's VALU phases?
//   hipcc --offload-arch=gfx950 -O3 kv_model.hip -o kv_model.exe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VALU>
__global__ __launch_bounds__(512, 2) void k_model(float* out, unsigned long long* stamps, int blocks, int stagger) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float kf[16], vf[16];
  f32x16 q0, q1, o0, o1;
  for (int i = 0; i < 16; ++i) { kf[i] = (float)(lane + i) * 1e-3f; vf[i] = (float)(lane - i) * 1e-3f; q0[i] = 0.01f * i; q1[i] = 0.02f * i; o0[i] = 0.f; o1[i] = 0.f; }
  if (wave >= 4) for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(1);
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int b = 0; b < blocks; ++b) {
    f32x16 s0, s1;
    for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[n], q0[n], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[n], q1[n], s1, 0, 0, 0);
    }
    if constexpr (VALU) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        f32x16& s = u ? s1 : s0;
        float m = s[15];
#pragma unroll
        for (int i = 0; i < 5; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(s[3 * i]), "v"(s[3 * i + 1]));
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s[i]) : "v"(m));
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(s[i]));
        float l = 0.f;
#pragma unroll
        for (int i = 0; i < 15; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(l) : "v"(s[i]));
        q0[u] += l * 1e-9f;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[r], s0[r], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[r], s1[r], o1, 0, 0, 0);
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += o0[i] + o1[i];
  out[blockIdx.x * 512 + t] = s;
  if (lane == 0) { stamps[(blockIdx.x * 8 + wave) * 2] = c0; stamps[(blockIdx.x * 8 + wave) * 2 + 1] = c1; }
}

// Specialised form: wavefronts 0-3 issue ONLY matrix instructions (S of block k, then O += V . P of block k-2), wavefronts 4-7 (one per
// SIMD) do ONLY the softmax-like VALU block, on block k-1, in place in LDS (the D layout of S^T is the B-operand layout of P^T: every lane
// reads and writes its own 32 values).  One s_barrier per block for all eight.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512, 2) void k_spec(float* out, unsigned long long* stamps, int blocks) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [4 pairs][2 buffers][2 u][4 g][64 lanes][4]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, pair = wave & 3;
  float* B = smem + pair * 2 * 2048;
  const unsigned long long c0 = __builtin_readcyclecounter();
  float res = 0.f;
  if (wave < 4) {
    float kf[16], vf[16];
    f32x16 q0, q1, o0, o1;
    for (int i = 0; i < 16; ++i) { kf[i] = (float)(lane + i) * 1e-3f; vf[i] = (float)(lane - i) * 1e-3f; q0[i] = 0.01f * i; q1[i] = 0.02f * i; o0[i] = 0.f; o1[i] = 0.f; }
    for (int k = 0; k < blocks + 2; ++k) {
      float* buf = B + (k & 1) * 2048;
      f32x4 pf[2][4];
      if (k >= 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int g = 0; g < 4; ++g) pf[u][g] = *reinterpret_cast<const f32x4*>(buf + ((u * 4 + g) * 64 + lane) * 4);
      }
      f32x16 s0, s1;
      for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
      if (k < blocks) {
#pragma unroll
        for (int n = 0; n < 16; ++n) {
          s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[n], q0[n], s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[n], q1[n], s1, 0, 0, 0);
        }
      }
      if (k >= 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[r], pf[0][r >> 2][r & 3], o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[r], pf[1][r >> 2][r & 3], o1, 0, 0, 0);
        }
      }
      if (k < blocks) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 a = {s0[g * 4], s0[g * 4 + 1], s0[g * 4 + 2], s0[g * 4 + 3]}, b = {s1[g * 4], s1[g * 4 + 1], s1[g * 4 + 2], s1[g * 4 + 3]};
          *reinterpret_cast<f32x4*>(buf + ((0 * 4 + g) * 64 + lane) * 4) = a;
          *reinterpret_cast<f32x4*>(buf + ((1 * 4 + g) * 64 + lane) * 4) = b;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    for (int i = 0; i < 16; ++i) res += o0[i] + o1[i];
  } else {
    float acc = 0.f;
    for (int k = 0; k < blocks + 2; ++k) {
      if (k >= 1 && k <= blocks) {
        float* buf = B + ((k - 1) & 1) * 2048;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float s[16];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(buf + ((u * 4 + g) * 64 + lane) * 4);
            s[g * 4] = v[0]; s[g * 4 + 1] = v[1]; s[g * 4 + 2] = v[2]; s[g * 4 + 3] = v[3];
          }
          float m = s[15];
#pragma unroll
          for (int i = 0; i < 5; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(s[3 * i]), "v"(s[3 * i + 1]));
#pragma unroll
          for (int i = 0; i < 16; ++i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s[i]) : "v"(m));
#pragma unroll
          for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(s[i]));
          float l = 0.f;
#pragma unroll
          for (int i = 0; i < 15; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(l) : "v"(s[i]));
          acc += l;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v = {s[g * 4], s[g * 4 + 1], s[g * 4 + 2], s[g * 4 + 3]};
            *reinterpret_cast<f32x4*>(buf + ((u * 4 + g) * 64 + lane) * 4) = v;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    res = acc;
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  out[blockIdx.x * 512 + t] = res;
  if (lane == 0) { stamps[(blockIdx.x * 8 + wave) * 2] = c0; stamps[(blockIdx.x * 8 + wave) * 2 + 1] = c1; }
}
static void run_spec(float* out, unsigned long long* st) {
  const int wgs = 256, blocks = 128;    // 4 matrix wavefronts do the work of run<1>(.., 512, ..)'s 8: 128 blocks each
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_spec, dim3(wgs), dim3(512), 65536, 0, out, st, blocks);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> s(wgs * 16);
  (void)hipMemcpy(s.data(), st, wgs * 16 * 8, hipMemcpyDeviceToHost);
  double span = 0;
  for (int b = 0; b < wgs; ++b) {
    unsigned long long lo = ~0ull, hi = 0;
    for (int w = 0; w < 8; ++w) {
      lo = s[(b * 8 + w) * 2] < lo ? s[(b * 8 + w) * 2] : lo;
      hi = s[(b * 8 + w) * 2 + 1] > hi ? s[(b * 8 + w) * 2 + 1] : hi;
    }
    span += (double)(hi - lo) / wgs;
  }
  printf("%-44s matrix wavefront + softmax wavefront per SIMD, S / P through LDS, one barrier per block: span %9.0f cycles | matrix pipe busy %.3f of the span\n",
         "specialised", span, (double)blocks * 64 * 64 / span);
}

template <int VALU>
static void run(const char* tag, int nthreads, int stagger, float* out, unsigned long long* st) {
  const int wgs = 256, blocks = 64;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_model<VALU>, dim3(wgs), dim3(nthreads), 0, 0, out, st, blocks, stagger);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> s(wgs * 16);
  (void)hipMemcpy(s.data(), st, wgs * 16 * 8, hipMemcpyDeviceToHost);
  const int nw = nthreads / 64;
  double span = 0, each = 0;
  for (int b = 0; b < wgs; ++b) {
    unsigned long long lo = ~0ull, hi = 0;
    for (int w = 0; w < nw; ++w) {
      lo = s[(b * 8 + w) * 2] < lo ? s[(b * 8 + w) * 2] : lo;
      hi = s[(b * 8 + w) * 2 + 1] > hi ? s[(b * 8 + w) * 2 + 1] : hi;
      each += (double)(s[(b * 8 + w) * 2 + 1] - s[(b * 8 + w) * 2]) / (wgs * nw);
    }
    span += (double)(hi - lo) / wgs;
  }
  const double mfma_cycles = (double)blocks * 64 * 64 * (nw / 4);   // matrix-pipe cycles per SIMD
  printf("%-44s %d wavefronts per SIMD, stagger %3d: workgroup span %9.0f cycles, a wavefront %9.0f | matrix pipe busy %.3f of the span\n", tag, nw / 4,
         stagger, span, each, mfma_cycles / span);
}

int main() {
  float* out; unsigned long long* st;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&st, 256 * 16 * 8);
  run<0>("matrix instructions only", 256, 0, out, st);
  run<0>("matrix instructions only", 512, 0, out, st);
  run<1>("with the softmax-like VALU block", 256, 0, out, st);
  run<1>("with the softmax-like VALU block", 512, 0, out, st);
  for (int sg : {8, 16, 24, 32, 48, 64}) run<1>("with the softmax-like VALU block", 512, sg, out, st);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_spec), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  run_spec(out, st);
  return 0;
}
