// tools/microbench6.hip -- grouped xor(v,v) / bcnt schedules: does v_xor_b32 v,v keep its
// 2.4-cycle rate when issued in runs between runs of v_bcnt?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
constexpr int ITERS = 4000;

// G = group size (G xors then G bcnts), 16 pairs per iteration. SV: xor takes an SGPR operand.
template <int G, bool SV>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, unsigned long long* stamps) {
  uint32_t a[16], wv[16], t[16], acc[4] = {0, 0, 0, 0};
  uint32_t ws = seed | 1;
  for (int i = 0; i < 16; i++) { a[i] = threadIdx.x * 2654435761u + i * 40503u + seed; wv[i] = a[i] * 31u + 7u; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int g = 0; g < 16; g += G) {
#pragma unroll
      for (int i = g; i < g + G; i++) {
        if (SV) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t[i]) : "s"(ws), "v"(a[i]));
        else asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t[i]) : "v"(wv[i]), "v"(a[i]));
      }
#pragma unroll
      for (int i = g; i < g + G; i++) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i & 3]) : "v"(t[i]));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int G, bool SV>
void run(int waves_per_simd) {
  int blocks = 256 * waves_per_simd;
  uint32_t* out; unsigned long long* st;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4); (void)hipMalloc(&st, blocks * 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<G, SV>), dim3(blocks), dim3(256), 0, 0, out, 12345u, st);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<G, SV>), dim3(blocks), dim3(256), 0, 0, out, 12345u, st);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 2); (void)hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
  std::vector<double> mhz;
  for (int b = 0; b < blocks; b++) mhz.push_back(100.0 * h[2 * b] / (double)h[2 * b + 1]);
  std::sort(mhz.begin(), mhz.end());
  double clk = mhz[blocks / 2];
  double pairs = (double)ITERS * 16;
  printf("group=%2d xor %s  w/SIMD=%d  %.3f ms clk %4.0f  %.2f SIMD-cycles per (xor+bcnt) pair\n", G, SV ? "s,v" : "v,v", waves_per_simd, ms, clk,
         clk * 1e6 * ms * 1e-3 / (pairs * waves_per_simd));
  (void)hipFree(out); (void)hipFree(st);
}

int main() {
  for (int w : {4, 8}) {
    run<1, false>(w); run<2, false>(w); run<4, false>(w); run<8, false>(w); run<16, false>(w);
    run<1, true>(w); run<4, true>(w); run<16, true>(w);
  }
  return 0;
}
