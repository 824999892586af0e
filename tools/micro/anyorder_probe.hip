// Can a DEPENDENT launch start under its predecessor's tail on gfx950?  The one-pair forward is a chain of 94 dependent launches and
// 385 us of its 708 us floor is their fixed cost (profiles/r6_one_pair_floor_table.txt).  hipExtLaunchKernelGGL(..., hipExtAnyOrderLaunch)
// clears the AQL barrier bit of a dispatch, so the command processor may start it while the previous kernel of the SAME queue still
// runs (hip_ext.h says "not supported on AMD GFX9xx boards"; this probe asks the chip).  If it works, a consumer can run its
// producer-independent prologue (weight loads) early and wait for its input on a device-scope counter instead of the kernel boundary.
//   part 1: a 200 us spin kernel, then a stamp kernel - launched plainly and with the any-order flag: when does the stamp kernel start?
//   part 2: a chain of N links, every link = G workgroups that (a) pull WKB KB of "weights" each (independent of the predecessor),
//           (b) read a slice the predecessor wrote (from another workgroup, so it crosses XCDs), (c) write their own slice.
//           plain    : ordinary launches, the kernel boundary is the dependency
//           anyorder : any-order launches; (b) waits until the predecessor's arrival counter has reached G (bounded poll - no hang
//                      if the flag is not honoured or the dispatch order is not what we think), slices are stored write-through
//                      (sc1) and drained before the arrival, the consumer acquires at agent scope once per workgroup.
//           Every link adds 1 to the data: the final values check that no link read stale data.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 anyorder_probe.hip -o anyorder_probe.exe && ./anyorder_probe.exe
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin_kernel(unsigned long long* out, unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(20);
  if (threadIdx.x == 0) { out[0] = t0; out[1] = wall_clock64(); }
}
__global__ void stamp_kernel(unsigned long long* out) {
  if (threadIdx.x == 0) out[0] = wall_clock64();
}

struct Link {
  const float* w;           // [G][wfloats]
  const float* in;          // [G][SLICE]
  float* out;               // [G][SLICE]
  const unsigned* wait;     // predecessor's arrival counter (nullptr: first link)
  unsigned* arrive;         // this link's arrival counter
  unsigned* err;            // bounded polls that ran out
  unsigned long long* stamps;   // [2]: first workgroup's entry, last arrival (anyorder) / any workgroup's exit (max)
  int wfloats, mode, G;
};
constexpr int SLICE = 4096;   // floats per workgroup and link: 16 KB

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_wt(float4* dst, float4 v4) {
  const f32x4 v = {v4.x, v4.y, v4.z, v4.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
}

__global__ __launch_bounds__(256) void link_kernel(const Link p) {
  __shared__ float red[256];
  const int t = threadIdx.x, wg = blockIdx.x;
  if (wg == 0 && t == 0) atomicMin(p.stamps, wall_clock64());
  // (a) the predecessor-independent part: this workgroup's weights
  float acc = 0.f;
  const float4* w4 = reinterpret_cast<const float4*>(p.w + (size_t)wg * p.wfloats);
  for (int i = t; i < p.wfloats / 4; i += 256) {
    const float4 v = w4[i];
    acc += (v.x + v.y) + (v.z + v.w);
  }
  red[t] = acc;
  __syncthreads();
  // (b) the dependency
  if (p.mode == 1 && p.wait != nullptr) {
    if (t == 0) {
      int it = 0;
      while (__hip_atomic_load(p.wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p.G) {
        if (++it > 50000) { atomicAdd(p.err, 1u); break; }
        __builtin_amdgcn_s_sleep(2);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  const int src = (wg * 7 + 3) % p.G;
  const float4* in4 = reinterpret_cast<const float4*>(p.in + (size_t)src * SLICE);
  float4* out4 = reinterpret_cast<float4*>(p.out + (size_t)wg * SLICE);
  const float z = red[(t + 1) & 255] * 0.f;   // keeps the weight loads alive (weights are finite)
  for (int i = t; i < SLICE / 4; i += 256) {
    float4 v = in4[i];
    v.x += 1.f + z; v.y += 1.f; v.z += 1.f; v.w += 1.f;
    if (p.mode == 1) store_wt(out4 + i, v);
    else out4[i] = v;
  }
  if (p.mode == 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave drains its write-through stores
    __syncthreads();
    if (t == 0) __hip_atomic_fetch_add(p.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (t == 0) atomicMax(p.stamps + 1, wall_clock64());
}

int main() {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  unsigned long long* st;
  CK(hipMalloc(&st, 64 * 8));
  // ---- part 1 ----
  for (int any = 0; any < 2; ++any) {
    CK(hipMemset(st, 0, 64 * 8));
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, st, 20000ull);   // 200 us at 100 MHz
      if (any) hipExtLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, st + 2);
      else hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, st + 2);
      CK(hipStreamSynchronize(s));
      unsigned long long h[3];
      CK(hipMemcpy(h, st, 24, hipMemcpyDeviceToHost));
      printf("part 1 %-9s rep %d: spin kernel ran %.1f us; the next kernel started %.1f us after the spin kernel started (%s)\n",
             any ? "any-order" : "plain", rep, (h[1] - h[0]) * 0.01, ((long long)h[2] - (long long)h[0]) * 0.01,
             h[2] < h[1] ? "OVERLAPPED" : "after it ended");
    }
  }
  // ---- part 2 ----
  const int N = 96;
  for (int G : {250, 128}) {
    for (int wkb : {0, 64, 192}) {
      const int wfloats = wkb * 256;
      float *w, *buf0, *buf1;
      CK(hipMalloc(&w, (size_t)G * (wfloats ? wfloats : 4) * 4));
      CK(hipMemset(w, 0, (size_t)G * (wfloats ? wfloats : 4) * 4));
      CK(hipMalloc(&buf0, (size_t)G * SLICE * 4));
      CK(hipMalloc(&buf1, (size_t)G * SLICE * 4));
      unsigned* ctr;
      CK(hipMalloc(&ctr, (N + 2) * 4));
      unsigned long long* stamps;
      CK(hipMalloc(&stamps, 16));
      for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f;
        unsigned errs = 0;
        float first = -1.f;
        for (int rep = 0; rep < 5; ++rep) {
          CK(hipMemsetAsync(buf0, 0, (size_t)G * SLICE * 4, s));
          CK(hipMemsetAsync(ctr, 0, (N + 2) * 4, s));
          const unsigned long long init[2] = {~0ull, 0ull};
          CK(hipMemcpyAsync(stamps, init, 16, hipMemcpyHostToDevice, s));
          CK(hipStreamSynchronize(s));
          // a blocker so that the whole chain is queued before its first link runs (the host needs ~4 us per launch)
          hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, st + 8, 150000ull);   // 1.5 ms
          for (int i = 0; i < N; ++i) {
            Link p;
            p.w = w; p.in = (i & 1) ? buf1 : buf0; p.out = (i & 1) ? buf0 : buf1;
            p.wait = i ? ctr + i - 1 : nullptr; p.arrive = ctr + i; p.err = ctr + N; p.stamps = stamps;
            p.wfloats = wfloats; p.mode = mode; p.G = G;
            // the first link is an ordinary launch (behind the blocker) in both modes
            if (mode == 1 && i > 0) hipExtLaunchKernelGGL(link_kernel, dim3(G), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, p);
            else hipLaunchKernelGGL(link_kernel, dim3(G), dim3(256), 0, s, p);
          }
          CK(hipStreamSynchronize(s));
          unsigned long long h[2];
          CK(hipMemcpy(h, stamps, 16, hipMemcpyDeviceToHost));
          unsigned e;
          CK(hipMemcpy(&e, ctr + N, 4, hipMemcpyDeviceToHost));
          errs += e;
          const float us = (h[1] - h[0]) * 0.01f / N;
          if (us < best) best = us;
          if (rep == 4) {
            std::vector<float> host((size_t)G * SLICE);
            CK(hipMemcpy(host.data(), (N & 1) ? buf1 : buf0, host.size() * 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (float v : host) bad += (v != (float)N);
            first = (float)bad / host.size();
          }
        }
        printf("part 2  G %3d  weights %3d KB per workgroup  %-8s: %.2f us per link (best of 5), polls that ran out %u, wrong values %.4f %%\n", G, wkb,
               mode ? "anyorder" : "plain", best, errs, first * 100.f);
      }
      (void)hipFree(w); (void)hipFree(buf0); (void)hipFree(buf1); (void)hipFree(ctr); (void)hipFree(stamps);
    }
  }
  return 0;
}
