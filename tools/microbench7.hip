// tools/microbench7.hip -- issue cost of the ternary inner loops:
//   TB  (cnvW1A2/lfcW1A2):  v_bitop3 (za & (sa ^ w)) + v_bcnt                  per 32 synapses
//   TB' the same as v_xor + v_and + v_bcnt
//   TT  (cnvW2A2):          v_and + v_bcnt + v_bitop3 + v_bcnt                  per 32 synapses
//   TT' v_and + v_bitop3 + v_bcnt + v_bcnt (logic ops adjacent)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
constexpr int ITERS = 4000;

template <int FORM>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, unsigned long long* stamps) {
  uint32_t as[16], az[16], acc[4] = {0, 0, 0, 0}, zcc[4] = {0, 0, 0, 0};
  uint32_t ws = seed | 1, wz = seed * 77u + 5u;
  for (int i = 0; i < 16; i++) { as[i] = threadIdx.x * 2654435761u + i * 40503u + seed; az[i] = as[i] * 31u + 7u; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      uint32_t t, u;
      if (FORM == 0)
        asm volatile("v_bitop3_b32 %0, %2, %3, %4 bitop3:0x28\n\tv_bcnt_u32_b32 %1, %0, %1" : "=&v"(t), "+v"(acc[i & 3]) : "s"(ws), "v"(as[i]), "v"(az[i]));
      else if (FORM == 1)
        asm volatile("v_xor_b32 %0, %2, %3\n\tv_and_b32 %0, %0, %4\n\tv_bcnt_u32_b32 %1, %0, %1" : "=&v"(t), "+v"(acc[i & 3]) : "s"(ws), "v"(as[i]), "v"(az[i]));
      else if (FORM == 2)
        asm volatile("v_and_b32 %0, %4, %5\n\tv_bcnt_u32_b32 %2, %0, %2\n\tv_bitop3_b32 %1, %6, %7, %0 bitop3:0x28\n\tv_bcnt_u32_b32 %3, %1, %3"
                     : "=&v"(t), "=&v"(u), "+v"(zcc[i & 3]), "+v"(acc[i & 3]) : "s"(wz), "v"(az[i]), "s"(ws), "v"(as[i]));
      else if (FORM == 3)
        asm volatile("v_and_b32 %0, %4, %5\n\tv_bitop3_b32 %1, %6, %7, %0 bitop3:0x28\n\tv_bcnt_u32_b32 %2, %0, %2\n\tv_bcnt_u32_b32 %3, %1, %3"
                     : "=&v"(t), "=&v"(u), "+v"(zcc[i & 3]), "+v"(acc[i & 3]) : "s"(wz), "v"(az[i]), "s"(ws), "v"(as[i]));
      else if (FORM == 4)  // TT with plain VOP2 logic: and, bcnt, xor, and, bcnt
        asm volatile("v_and_b32 %0, %4, %5\n\tv_bcnt_u32_b32 %2, %0, %2\n\tv_xor_b32 %1, %6, %7\n\tv_and_b32 %1, %1, %0\n\tv_bcnt_u32_b32 %3, %1, %3"
                     : "=&v"(t), "=&v"(u), "+v"(zcc[i & 3]), "+v"(acc[i & 3]) : "s"(wz), "v"(az[i]), "s"(ws), "v"(as[i]));
      else if (FORM == 5)  // XNOR pair for reference
        asm volatile("v_xor_b32 %0, %2, %3\n\tv_bcnt_u32_b32 %1, %0, %1" : "=&v"(t), "+v"(acc[i & 3]) : "s"(ws), "v"(as[i]));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + zcc[0] + zcc[1] + zcc[2] + zcc[3];
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int FORM>
void run(int waves_per_simd, const char* name) {
  int blocks = 256 * waves_per_simd;
  uint32_t* out; unsigned long long* st;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4); (void)hipMalloc(&st, blocks * 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<FORM>), dim3(blocks), dim3(256), 0, 0, out, 12345u, st);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<FORM>), dim3(blocks), dim3(256), 0, 0, out, 12345u, st);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 2); (void)hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
  std::vector<double> mhz;
  for (int b = 0; b < blocks; b++) mhz.push_back(100.0 * h[2 * b] / (double)h[2 * b + 1]);
  std::sort(mhz.begin(), mhz.end());
  double clk = mhz[blocks / 2];
  double units = (double)ITERS * 16;
  printf("%-44s w/SIMD=%d  %.3f ms clk %4.0f  %.2f SIMD-cycles per 32 synapses\n", name, waves_per_simd, ms, clk,
         clk * 1e6 * ms * 1e-3 / (units * waves_per_simd));
  (void)hipFree(out); (void)hipFree(st);
}

int main() {
  for (int w : {4, 8}) {
    run<5>(w, "XNOR  xor(s,v) bcnt");
    run<0>(w, "TB    bitop3(s,v,v) bcnt");
    run<1>(w, "TB'   xor(s,v) and bcnt");
    run<2>(w, "TT    and(s,v) bcnt bitop3(s,v,v) bcnt");
    run<3>(w, "TT'   and bitop3 bcnt bcnt");
    run<4>(w, "TT''  and bcnt xor and bcnt");
  }
  return 0;
}
