// Cooperative tail of the kernels that leave per-workgroup PARTIAL outputs of a row tile (the fused FFN block: one partial per
// hidden-unit chunk; attention with the out-projection fused in: one partial per head): the workgroups of a row tile sum the
// partials, add bias + residual and apply LayerNorm THEMSELVES, each for its own share of the tile's 32 rows, instead of a
// second launch (ln_reduce_kernel) doing it.  At one pair a launch costs ~1.7 us of dispatch (the 8 XCDs start a grid one after
// the other, profiles/r3_launch_ramp_xcd_skew.txt) plus its own ramp - 24 such launches per forward.
//
// Protocol (per row tile; `n` members; device memory `state` = one arrival word + 16 claim words of 64 bits, generation-tagged
// so that nothing ever has to be reset):
//   1. a member writes its partial with write-through (sc1) stores, drains them (vmcnt(0), barrier) and ARRIVES: the arrival
//      word holds (generation << 8 | count); the first arriver of a generation starts the count at 1;
//   2. the LAST arriver (count == n) knows every partial is in memory: it finishes its own share and then every share that
//      nobody has claimed;  any other member polls the arrival word a BOUNDED number of times; if it sees the tile complete it
//      claims its own share (compare-and-swap of the claim word to the generation) and finishes it, otherwise it leaves.
// Nobody waits without bound, so no placement, residency or co-scheduling assumption is needed for correctness or progress
// (MI355X_MICROARCH.md: "placement-independent protocols only"): a member that is not resident, or gave up, simply has its share
// finished by the last arriver - which exists by definition.  Every share is finished exactly once (the claim word), by the same
// arithmetic in the same order as ln_reduce_kernel: bit-identical to the two-launch form.  Partials are read back with sc1 loads
// (they bypass this CU's L1; the producers stored write-through), 8 in flight per lane.
//
// MEASURED (MI355X, 1 pair x 1000 queries): correct - bit-identical to the two-launch form with the normal wait, with no waiting at
// all, and with three forwards in flight on three streams (tests/test_parity_gpu.py) - but SLOWER: 1000.7 vs 823.7 us per forward,
// +7.4 us per tail.  The hand-off is four dependent memory-side round trips (arrive, poll, claim, sc1 partial reads) of 1-2 us
// each - device-scope atomics and sc1 accesses are served behind the per-XCD L2s - against 1.7 us of dispatch + 2.3 us of kernel
// for the ln_reduce launch it replaces.  Off by default (cotr_set_coop_tail); kept as the measured answer to "fewer launches".
#pragma once
#include "../common.h"

struct CoopTail {
  unsigned long long* state;   // [row tiles][COOP_WORDS]; nullptr = no tail (separate ln_reduce launch)
  unsigned long long gen;      // generation of this launch (host counter, > 0, unique per launch on this state array)
  int spin_limit;              // polls before a member stops waiting (0: the last arriver finishes the whole tile)
  const float* bias;           // [256]
  const float* residual;       // [rows][256] or nullptr
  const float *w, *b;          // LayerNorm
  const float *post_w, *post_b;  // second LayerNorm (decoder.norm) or nullptr
  float* y;                    // [rows][256]
};
#define COOP_WORDS 17

__device__ __forceinline__ float coop_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// 8 float4 loads that bypass the vector L1 (sc1), one wait
__device__ __forceinline__ void coop_load8_sc1(const float* p0, size_t stride, f32x4* t) {
  const float *a0 = p0, *a1 = p0 + stride, *a2 = p0 + 2 * stride, *a3 = p0 + 3 * stride, *a4 = p0 + 4 * stride, *a5 = p0 + 5 * stride,
              *a6 = p0 + 6 * stride, *a7 = p0 + 7 * stride;
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc1\n\t"
      "global_load_dwordx4 %1, %9, off sc1\n\t"
      "global_load_dwordx4 %2, %10, off sc1\n\t"
      "global_load_dwordx4 %3, %11, off sc1\n\t"
      "global_load_dwordx4 %4, %12, off sc1\n\t"
      "global_load_dwordx4 %5, %13, off sc1\n\t"
      "global_load_dwordx4 %6, %14, off sc1\n\t"
      "global_load_dwordx4 %7, %15, off sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
      : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7)
      : "memory");
}

// one row: y = LN(bias + residual + sum of the np partials in order) [then the post norm]; one wavefront, lane = 4 columns.
// Same arithmetic and order as ln_reduce_kernel (pointwise.hip).
__device__ __forceinline__ void coop_finish_row(const CoopTail& ct, const float* parts, size_t pstride, int np, int row, int lane) {
  f32x4 v = *reinterpret_cast<const f32x4*>(ct.bias + lane * 4);
  f32x4 rr = {0.f, 0.f, 0.f, 0.f};
  if (ct.residual != nullptr) rr = *reinterpret_cast<const f32x4*>(ct.residual + (size_t)row * 256 + lane * 4);
  v += rr;
  const float* prow = parts + (size_t)row * 256 + lane * 4;
  for (int c = 0; c + 8 <= np; c += 8) {         // np is 8 or 16
    f32x4 t[8];
    coop_load8_sc1(prow + (size_t)c * pstride, pstride, t);
#pragma unroll
    for (int k = 0; k < 8; ++k) v += t[k];
  }
  const float mean = coop_wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256.f);
  const f32x4 d = {v[0] - mean, v[1] - mean, v[2] - mean, v[3] - mean};
  const float var = coop_wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / 256.f);
  const float rstd = 1.f / sqrtf(var + 1e-5f);
  const f32x4 ww = *reinterpret_cast<const f32x4*>(ct.w + lane * 4);
  const f32x4 bb = *reinterpret_cast<const f32x4*>(ct.b + lane * 4);
  f32x4 out;
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = d[i] * rstd * ww[i] + bb[i];
  if (ct.post_w != nullptr) {
    const float m2 = coop_wave_sum(out[0] + out[1] + out[2] + out[3]) * (1.f / 256.f);
    const f32x4 d2 = {out[0] - m2, out[1] - m2, out[2] - m2, out[3] - m2};
    const float v2 = coop_wave_sum(d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2] + d2[3] * d2[3]) * (1.f / 256.f);
    const float r2 = 1.f / sqrtf(v2 + 1e-5f);
    const f32x4 w2 = *reinterpret_cast<const f32x4*>(ct.post_w + lane * 4);
    const f32x4 b2 = *reinterpret_cast<const f32x4*>(ct.post_b + lane * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = d2[i] * r2 * w2[i] + b2[i];
  }
  *reinterpret_cast<f32x4*>(ct.y + (size_t)row * 256 + lane * 4) = out;
}

// Called by EVERY thread of a member workgroup after its partial stores (sc1) were issued.  `tile` = index of the row tile in
// `state`, row0 = its first row, nvalid = its valid rows (<= 32), `member` in [0, n), n in {8, 16} = partials per tile; the
// workgroup has at least 32 / n wavefronts.  `flags` = 2 ints of LDS.
__device__ __forceinline__ void coop_tail_run(const CoopTail& ct, const float* parts, size_t pstride, int tile, int row0, int nvalid,
                                              int member, int n, int* flags) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's write-through partial stores have reached memory ...
  __syncthreads();                                   // ... and so have the whole workgroup's
  unsigned long long* st = ct.state + (size_t)tile * COOP_WORDS;
  if (t == 0) {
    // arrive: (generation << 8 | count); a stale word of an older generation restarts at 1
    unsigned long long old = __hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), want;
    do {
      want = ((old >> 8) == ct.gen) ? old + 1 : ((ct.gen << 8) | 1ull);
    } while (!__hip_atomic_compare_exchange_strong(st, &old, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    int role = ((int)(want & 255) == n) ? 2 : 0;     // 2 = last arriver
    if (role == 0) {
      for (int it = 0; it < ct.spin_limit; ++it) {
        const unsigned long long a = __hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((a >> 8) == ct.gen && (int)(a & 255) >= n) { role = 1; break; }   // 1 = saw the tile complete
        __builtin_amdgcn_s_sleep(4);
      }
    }
    flags[0] = role;
  }
  __syncthreads();
  const int role = flags[0];
  if (role == 0) return;                             // gave up waiting: the last arriver finishes this member's share
  const int rps = 32 / n;                            // rows per share
  const int first = member, count = (role == 2) ? n : 1;
  for (int k = 0; k < count; ++k) {
    const int s = (first + k) % n;                   // own share first, then (last arriver only) everybody else's
    __syncthreads();
    if (t == 0) {                                    // claim share s for this generation
      unsigned long long old = __hip_atomic_load(st + 1 + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int got = 0;
      while (old != ct.gen) {
        if (__hip_atomic_compare_exchange_strong(st + 1 + s, &old, ct.gen, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          got = 1;
          break;
        }
      }
      flags[1] = got;
    }
    __syncthreads();
    if (flags[1] && wave < rps) {
      const int r = s * rps + wave;
      if (r < nvalid) coop_finish_row(ct, parts, pstride, n, row0 + r, lane);
    }
  }
}
