// What does a launch pay for COLD CODE?  The one-pair forward is a dependent chain of 94 launches of ~30 different kernels, each run once
// per forward between launches that stream megabytes through the 4 MB L2s; its cheapest launches cost 4.1-4.9 us in situ against
// 1.7 us for an empty dependent chain and 2.5 us for the same small kernel launched back to back (profiles/r4_ln_reduce_vs_partial_slabs.txt).
// This probe separates the candidates: a graph-captured dependent chain of small kernels that execute ~NI straight-line instructions
// before one load / store, either the SAME kernel every time or 64 DISTINCT instantiations (distinct code addresses), with and without
// a streaming kernel (the same code every time) between them that pulls MB megabytes through the L2s.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 icache_probe.hip -o icache_probe.exe && ./icache_probe.exe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <utility>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define R256(x) R4(R64(x))

// ~NI straight-line VALU instructions (4 bytes each) executed by every wavefront before its one load / store: the code a kernel's
// prologue runs through.  ID makes every instantiation its own function at its own address.
template <int ID, int NI>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ a, float* __restrict__ b, unsigned long long* pc_slot) {
  if (pc_slot != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {   // discovery pass: where does this kernel's code live?
    unsigned long long pc;
    asm volatile("s_getpc_b64 %0" : "=s"(pc));
    *pc_slot = pc;
  }
  float v = (float)ID, w = 1.0009765625f;
  if constexpr (NI >= 256) { R256(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v) : "v"(w));) }
  if constexpr (NI >= 512) { R256(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v) : "v"(w));) }
  if constexpr (NI >= 1024) { R256(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v) : "v"(w));) R256(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v) : "v"(w));) }
  const int i = blockIdx.x * 256 + threadIdx.x;
  b[i] = a[i] + v * 1e-30f;
}

// warm: the NEXT launch's code (address from the discovery pass) is pulled through this XCD's L2 by the first wavefront of workgroups
// 0-7 (consecutive workgroups run on consecutive XCDs) - plain data loads of the code bytes, one 64-B line per lane and step
__global__ __launch_bounds__(256) void stream(const float4* __restrict__ a, float4* __restrict__ b, int n4, const char* warm, int warm_bytes) {
  if (warm != nullptr && blockIdx.x < 8 && threadIdx.x < 64) {
    for (int off = threadIdx.x * 64; off < warm_bytes; off += 64 * 64) {
      unsigned int t;
      asm volatile("global_load_dword %0, %1, off" : "=v"(t) : "v"(warm + off) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
    float4 v = a[i];
    v.x += 1.f;
    b[i] = v;
  }
}

using Fn = void (*)(const float*, float*, unsigned long long*);
template <int NI, int... I>
std::vector<Fn> table(std::integer_sequence<int, I...>) { return {probe<I, NI>...}; }

int main() {
  float *x, *y, *big0, *big1;
  CK(hipMalloc(&x, 1 << 20)); CK(hipMalloc(&y, 1 << 20));
  CK(hipMalloc(&big0, 256 << 20)); CK(hipMalloc(&big1, 256 << 20));
  CK(hipMemset(x, 0, 1 << 20)); CK(hipMemset(big0, 0, 256 << 20));
  hipStream_t s; CK(hipStreamCreate(&s));
  unsigned long long* pcs;
  CK(hipMalloc(&pcs, 64 * 8));
  std::vector<unsigned long long> pc_host(64);
  auto discover = [&](const std::vector<Fn>& fns) {
    for (int i = 0; i < 64; ++i) hipLaunchKernelGGL(fns[i], dim3(1), dim3(256), 0, s, x, y, pcs + i);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(pc_host.data(), pcs, 64 * 8, hipMemcpyDeviceToHost);
  };
  // warm: 0 = no; 1 = the streaming kernel before a small kernel touches that kernel's code (code_bytes from its recorded pc on)
  auto run = [&](const std::vector<Fn>& fns, bool distinct, int stream_mb, int wgs, int n, int warm, int code_bytes) -> float {
    hipGraph_t g; hipGraphExec_t e;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < n; ++i) {
      Fn f = fns[distinct ? i % fns.size() : 0];
      hipLaunchKernelGGL(f, dim3(wgs), dim3(256), 0, s, (i & 1) ? y : x, (i & 1) ? x : y, (unsigned long long*)nullptr);
      if (stream_mb) {
        // a different 'stream_mb' window of the big buffers every time: always cold in the L2s, like the weights of the next layer
        const size_t off = ((size_t)i * stream_mb % 192) << 20;
        const int nxt = distinct ? (i + 1) % (int)fns.size() : 0;
        const char* wp = warm ? (const char*)((pc_host[nxt] & ~63ull)) : nullptr;
        hipLaunchKernelGGL(stream, dim3(1024), dim3(256), 0, s, (const float4*)((char*)big0 + off), (float4*)((char*)big1 + off), stream_mb << 16, wp, code_bytes);
      }
    }
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipGraphLaunch(e, s); hipStreamSynchronize(s);
    float best = 1e9;
    for (int r = 0; r < 7; ++r) {
      hipEventRecord(a, s); hipGraphLaunch(e, s); hipEventRecord(b, s); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    hipGraphExecDestroy(e); hipGraphDestroy(g);
    return best * 1000.f / n;
  };
  const auto t256 = table<256>(std::make_integer_sequence<int, 64>());
  const auto t1024 = table<1024>(std::make_integer_sequence<int, 64>());
  printf("dependent chain of 128 small kernels (256 / 1024 straight-line instructions, then one load + store), us per link of the chain\n");
  printf("(a link = the small kernel [+ the streaming kernel]; 'cold code' = distinct - same)\n");
  for (int wgs : {250, 1000})
    for (int mb : {0, 2, 8, 32}) {
      for (int ni : {256, 1024}) {
        const auto& t = ni == 256 ? t256 : t1024;
        discover(t);
        const float same = run(t, false, mb, wgs, 128, 0, 0), dist = run(t, true, mb, wgs, 128, 0, 0);
        const float warmed = mb ? run(t, true, mb, wgs, 128, 1, ni * 4 + 256) : 0.f;
        printf("  %4d workgroups, %2d MB streamed between, %4d instructions: same kernel %6.2f  64 distinct kernels %6.2f  cold code %+5.2f us",
               wgs, mb, ni, same, dist, dist - same);
        if (mb) printf("  | next kernel's code touched by the streaming kernel: %6.2f (%+5.2f vs same)", warmed, warmed - same);
        printf("\n");
      }
    }
  return 0;
}
