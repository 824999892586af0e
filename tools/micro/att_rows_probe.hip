// Stage-by-stage check of cotr_amd/csrc/att_rows.hip against a host computation (one tile: 1 pair x 64 queries): q fragments,
// normalised head outputs, softmax statistics, out-projection accumulators, final rows.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../cotr_amd/csrc att_rows_probe.hip -o att_rows_probe.exe
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "../../cotr_amd/csrc/att_rows.hip"

thread_local int cotr_tls_device = -1;
static KnobSet g_knobs = {};
thread_local const KnobSet* cotr_tls_knobs = &g_knobs;
static float* g_zero = nullptr;
const float* gemm_zero_buffer() { return g_zero; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main() {
  const int nq = 64, ldkv = 768, ldq = 768;
  std::vector<float> q((size_t)nq * ldq), kv((size_t)512 * ldkv), wo(256 * 256), bo(256, 0.f), lw(256, 1.f), lb(256, 0.f);
  unsigned s = 777;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) * (1.f / 65536.f) - 0.5f; };
  for (auto& v : q) v = rnd() * 2.f;
  for (auto& v : kv) v = rnd() * 2.f;
  for (int i = 0; i < 256; ++i) for (int j = 0; j < 256; ++j) wo[i * 256 + j] = i == j ? 1.f : 0.f;
  float *qd, *kvd, *wod, *bod, *lwd, *lbd, *yd, *dbg;
  CK(hipMalloc(&qd, q.size() * 4)); CK(hipMalloc(&kvd, kv.size() * 4)); CK(hipMalloc(&wod, wo.size() * 4));
  CK(hipMalloc(&bod, 1024)); CK(hipMalloc(&lwd, 1024)); CK(hipMalloc(&lbd, 1024)); CK(hipMalloc(&yd, nq * 256 * 4));
  CK(hipMalloc(&dbg, 262144 * 4)); CK(hipMemset(dbg, 0, 262144 * 4));
  CK(hipMalloc(&g_zero, 256)); CK(hipMemset(g_zero, 0, 256));
  CK(hipMemcpy(qd, q.data(), q.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(kvd, kv.data(), kv.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(wod, wo.data(), wo.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bod, bo.data(), 1024, hipMemcpyHostToDevice));
  CK(hipMemcpy(lwd, lw.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(lbd, lb.data(), 1024, hipMemcpyHostToDevice));
  AttRowsParams p = {};
  p.q = qd; p.ldq = ldq; p.k = kvd + 256; p.v = kvd + 512; p.ldkv = ldkv; p.wo = wod; p.bo = bod; p.ln_w = lwd; p.ln_b = lbd; p.Y = yd;
  p.zeros = g_zero; p.nq = nq; p.dbg = dbg; p.nb = 1; p.tpp = 1; p.by_xcd = 0;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(att_rows_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAttRowsSmem));
  hipLaunchKernelGGL((att_rows_kernel<false, true>), dim3(1, 1), dim3(512), kAttRowsSmem, 0, p);
  CK(hipDeviceSynchronize());
  std::vector<float> d(65536), y(nq * 256);
  CK(hipMemcpy(d.data(), dbg, d.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(y.data(), yd, y.size() * 4, hipMemcpyDeviceToHost));
  // host: per head O, l
  std::vector<double> O((size_t)nq * 256), Lsum((size_t)nq * 8), Mx((size_t)nq * 8);
  const double LOG2E = 1.4426950408889634;
  for (int m = 0; m < nq; ++m)
    for (int h = 0; h < 8; ++h) {
      std::vector<double> sc(512);
      double mx = -1e300;
      for (int key = 0; key < 512; ++key) {
        double a = 0;
        for (int dd = 0; dd < 32; ++dd) a += (double)q[(size_t)m * ldq + h * 32 + dd] * kv[(size_t)key * ldkv + 256 + h * 32 + dd];
        sc[key] = a * LOG2E;
        mx = fmax(mx, sc[key]);
      }
      double l = 0;
      for (int key = 0; key < 512; ++key) { sc[key] = exp2(sc[key] - mx); l += sc[key]; }
      Lsum[m * 8 + h] = l; Mx[m * 8 + h] = mx;
      for (int dd = 0; dd < 32; ++dd) {
        double a = 0;
        for (int key = 0; key < 512; ++key) a += sc[key] * kv[(size_t)key * ldkv + 512 + h * 32 + dd];
        O[(size_t)m * 256 + h * 32 + dd] = a / l;
      }
    }
  double eq = 0, eo = 0, el = 0, em = 0, ey = 0, ef = 0;
  for (int w = 0; w < 8; ++w)
    for (int u = 0; u < 2; ++u)
      for (int r = 0; r < 16; ++r)
        for (int lane = 0; lane < 64; ++lane) {
          const int h = w, m = u * 32 + (lane & 31), dd = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const size_t idx = (((size_t)w * 2 + u) * 16 + r) * 64 + lane;
          eq = fmax(eq, fabs(d[idx] - q[(size_t)m * ldq + h * 32 + dd] * LOG2E));
          eo = fmax(eo, fabs(d[16384 + idx] - O[(size_t)m * 256 + h * 32 + dd]));
        }
  for (int w = 0; w < 8; ++w)
    for (int u = 0; u < 2; ++u)
      for (int lane = 0; lane < 32; ++lane) {
        const int h = w, m = u * 32 + lane;
        const size_t idx = ((size_t)w * 2 + u) * 64;
        const double l = d[32768 + idx + lane] + d[32768 + idx + lane + 32];
        const double mm = d[32768 + 1024 + idx + lane];
        // compare l * 2^m (scale-free)
        el = fmax(el, fabs(log2(l) + mm - (log2(Lsum[m * 8 + h]) + Mx[m * 8 + h])));
        em = fmax(em, fabs(mm - Mx[m * 8 + h]));
      }
  for (int w = 0; w < 8; ++w)
    for (int mb = 0; mb < 2; ++mb)
      for (int r = 0; r < 16; ++r)
        for (int lane = 0; lane < 64; ++lane) {
          const int m = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), n = 32 * w + (lane & 31);
          ey = fmax(ey, fabs(d[36864 + (((size_t)w * 2 + mb) * 16 + r) * 64 + lane] - O[(size_t)m * 256 + n]));
        }
  for (int m = 0; m < nq; ++m) {
    double mean = 0, var = 0;
    for (int n = 0; n < 256; ++n) mean += O[(size_t)m * 256 + n];
    mean /= 256;
    for (int n = 0; n < 256; ++n) var += (O[(size_t)m * 256 + n] - mean) * (O[(size_t)m * 256 + n] - mean);
    var /= 256;
    for (int n = 0; n < 256; ++n) ef = fmax(ef, fabs(y[(size_t)m * 256 + n] - (O[(size_t)m * 256 + n] - mean) / sqrt(var + 1e-5)));
  }
  {  // ring of wavefront 0 after the prologue: slots 0, 1 = K block 0, V block 0 of head 0, rows swizzled
    for (int slot = 0; slot < 2; ++slot) {
      double e = 0;
      const int col0 = slot == 1 ? 512 : 256, key00 = slot == 2 ? 32 : 0;
      for (int row = 0; row < 32; ++row)
        for (int c = 0; c < 32; ++c) {
          const int phys = (((c >> 2) ^ ((row >> 1) & 7)) << 2) | (c & 3);
          e = fmax(e, fabs(d[57344 + slot * 1024 + row * 32 + phys] - kv[(size_t)(key00 + row) * ldkv + col0 + c]));
        }
      printf("ring slot %d after the prologue: max abs error %.3e\n", slot, e);
      for (int pos : {0, 4, 32, 36, 256, 260, 512, 768}) {   // where do these LDS floats come from?
        const float val = d[57344 + slot * 1024 + pos];
        int fr = -1, fc = -1;
        for (size_t i = 0; i < kv.size(); ++i) if (kv[i] == val) { fr = (int)(i / ldkv); fc = (int)(i % ldkv); break; }
        int qr = -1, qc = -1;
        for (size_t i = 0; i < q.size(); ++i) if (q[i] == val) { qr = (int)(i / ldq); qc = (int)(i % ldq); break; }
        printf("   ring[%d][%4d] = %9.6f  <- kv[%d][%d]  q[%d][%d]\n", slot, pos, val, fr, fc, qr, qc);
      }
    }
  }
  {  // block 0 of head 0 (wavefront 0): V fragments, K fragments, P, O after the first PV
    double ev = 0, ep = 0, eo0 = 0;
    for (int r = 0; r < 16; ++r)
      for (int lane = 0; lane < 64; ++lane) {
        const int l31 = lane & 31, hh = lane >> 5, key = (r & 3) + 8 * (r >> 2) + 4 * hh;
        ev = fmax(ev, fabs(d[53248 + r * 64 + lane] - kv[(size_t)key * ldkv + 512 + l31]));
        // p = 2^(s - m_block) with m_block = max over the block's 32 keys for query l31
        double sc[32], mx = -1e300;
        for (int k2 = 0; k2 < 32; ++k2) {
          double a = 0;
          for (int dd = 0; dd < 32; ++dd) a += (double)q[(size_t)l31 * ldq + dd] * kv[(size_t)k2 * ldkv + 256 + dd];
          sc[k2] = a * LOG2E;
          mx = fmax(mx, sc[k2]);
        }
        ep = fmax(ep, fabs(d[53248 + 1024 + r * 64 + lane] - exp2(sc[key] - mx)));
        // o after the first block: O^T[d = (r&3)+8(r>>2)+4hh][m = l31] = sum_key p * V[key][d]
        const int dd0 = (r & 3) + 8 * (r >> 2) + 4 * hh;
        double a = 0;
        for (int k2 = 0; k2 < 32; ++k2) a += exp2(sc[k2] - mx) * kv[(size_t)k2 * ldkv + 512 + dd0];
        eo0 = fmax(eo0, fabs(d[53248 + 2048 + r * 64 + lane] - a));
      }
    printf("block 0, head 0: V fragments %.3e | P %.3e | O after the first PV %.3e\n", ev, ep, eo0);
  }
  // ---- timing: 64 / 128 / 256 tiles (pairs of 512 queries), both forms, phase stamps of wavefront 0 ----
  {
    const int NP = 32;
    float *qb, *kvb, *x2b, *wqb, *bqb, *yb;
    CK(hipMalloc(&qb, (size_t)NP * 512 * 768 * 4)); CK(hipMalloc(&kvb, (size_t)NP * 512 * 3072 * 4)); CK(hipMalloc(&x2b, (size_t)NP * 512 * 256 * 4));
    CK(hipMalloc(&wqb, 256 * 256 * 4)); CK(hipMalloc(&bqb, 1024)); CK(hipMalloc(&yb, (size_t)NP * 512 * 256 * 4));
    std::vector<float> h((size_t)NP * 512 * 3072);
    for (auto& v : h) v = rnd() * 2.f;
    CK(hipMemcpy(kvb, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(qb, h.data(), (size_t)NP * 512 * 768 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(x2b, h.data() + 12345, (size_t)NP * 512 * 256 * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < 65536; ++i) h[i] = rnd() * 0.12f;
    CK(hipMemcpy(wqb, h.data(), 65536 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bqb, h.data() + 70000, 1024, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(att_rows_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAttRowsSmem));
    AttRowsParams pb = p;
    pb.q = qb; pb.ldq = 768; pb.k = kvb; pb.v = kvb + 256; pb.ldkv = 3072; pb.Y = yb; pb.nq = 512; pb.tpp = 8; pb.residual = x2b;
    pb.x = nullptr; pb.x2 = x2b; pb.wq = wqb; pb.bq = bqb; pb.qscale = 0.17677669f;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(att_rows_kernel<false, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAttRowsSmem));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(att_rows_kernel<false, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAttRowsSmem));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(att_rows_kernel<false, true, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAttRowsSmem));
    for (int qp = 0; qp < 5; ++qp)
      for (int pairs : {8, 32}) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto launch = [&]() {
          if (qp == 1) hipLaunchKernelGGL((att_rows_kernel<true, true>), dim3(8, pairs), dim3(512), kAttRowsSmem, 0, pb);
          else if (qp == 0) hipLaunchKernelGGL((att_rows_kernel<false, true>), dim3(8, pairs), dim3(512), kAttRowsSmem, 0, pb);
          else if (qp == 2) hipLaunchKernelGGL((att_rows_kernel<false, true, 1>), dim3(8, pairs), dim3(512), kAttRowsSmem, 0, pb);
          else if (qp == 3) hipLaunchKernelGGL((att_rows_kernel<false, true, 2>), dim3(8, pairs), dim3(512), kAttRowsSmem, 0, pb);
          else hipLaunchKernelGGL((att_rows_kernel<false, true, 3>), dim3(8, pairs), dim3(512), kAttRowsSmem, 0, pb);
        };
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const int tiles = 8 * pairs;
        std::vector<unsigned long long> st((size_t)tiles * 16);
        CK(hipMemcpy(st.data(), reinterpret_cast<unsigned long long*>(dbg + 131072), st.size() * 8, hipMemcpyDeviceToHost));
        double cyc[7] = {}, wall = 0;
        for (int w = 0; w < tiles; ++w) {
          for (int k2 = 0; k2 < 7; ++k2) cyc[k2] += (double)(st[((size_t)w * 8 + k2 + 1) * 2] - st[((size_t)w * 8 + k2) * 2]) / tiles;
          wall += (double)(st[((size_t)w * 8 + 7) * 2 + 1] - st[(size_t)w * 8 * 2 + 1]) / tiles;
        }
        double tc = 0;
        for (double c : cyc) tc += c;
        printf("%s %2d pairs x 512 (%3d tiles): %7.1f us/launch | wave 0 of a tile: %.0f cycles in %.1f us = %.2f GHz | prologue %.0f | q phase %.0f (ideal %d) | K/V phase %.0f (ideal 131072) | barrier + O -> T %.0f | out-proj %.0f (ideal 32768) | barrier %.0f | epilogue %.0f\n",
               qp == 1 ? "decoder form (q projected)" : qp == 0 ? "encoder form (q given)    " : qp == 2 ? "encoder, NO softmax VALU  " : qp == 3 ? "encoder, NO K/V requests  " : "encoder, neither          ", pairs, tiles, ms * 100.f, tc, wall / 100.0, tc / (wall * 10.0), cyc[0], cyc[1],
               qp == 1 ? 32768 : 0, cyc[2], cyc[3], cyc[4], cyc[5], cyc[6]);
      }
  }
  printf("max abs error: q fragments %.3e | running max %.3e | log2(sum)+max %.3e | normalised O (registers) %.3e | out-proj accumulators (Wo = I) %.3e | final rows %.3e\n",
         eq, em, el, eo, ey, ef);
  return 0;
}
