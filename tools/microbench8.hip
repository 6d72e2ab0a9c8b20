// tools/microbench8.hip -- does the s_nop the compiler puts behind every inline-asm statement cost issue
// slots?  The same xor(s,v)+bcnt stream as 1, 4 or 16 pairs per asm statement.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
constexpr int ITERS = 4000;

#define PAIR(i, acc) "v_xor_b32 %0, %5, %" #i "\n\tv_bcnt_u32_b32 %" #acc ", %0, %" #acc "\n\t"

template <int PER>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, unsigned long long* stamps) {
  uint32_t a[16], acc[4] = {0, 0, 0, 0};
  uint32_t ws = seed | 1;
  for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 2654435761u + i * 40503u + seed;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; it++) {
    uint32_t t;
    if (PER == 1) {
#pragma unroll
      for (int i = 0; i < 16; i++)
        asm volatile("v_xor_b32 %0, %2, %3\n\tv_bcnt_u32_b32 %1, %0, %1" : "=&v"(t), "+v"(acc[i & 3]) : "s"(ws), "v"(a[i]));
    } else if (PER == 4) {
#pragma unroll
      for (int g = 0; g < 16; g += 4)
        asm volatile(PAIR(6, 1) PAIR(7, 2) PAIR(8, 3) PAIR(9, 4)
                     : "=&v"(t), "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
                     : "s"(ws), "v"(a[g]), "v"(a[g + 1]), "v"(a[g + 2]), "v"(a[g + 3]));
    } else {
      asm volatile(PAIR(6, 1) PAIR(7, 2) PAIR(8, 3) PAIR(9, 4) PAIR(10, 1) PAIR(11, 2) PAIR(12, 3) PAIR(13, 4)
                   PAIR(14, 1) PAIR(15, 2) PAIR(16, 3) PAIR(17, 4) PAIR(18, 1) PAIR(19, 2) PAIR(20, 3) PAIR(21, 4)
                   : "=&v"(t), "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
                   : "s"(ws), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]),
                     "v"(a[9]), "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15]));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int PER>
void run(int waves_per_simd) {
  int blocks = 256 * waves_per_simd;
  uint32_t* out; unsigned long long* st;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4); (void)hipMalloc(&st, blocks * 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<PER>), dim3(blocks), dim3(256), 0, 0, out, 12345u, st);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<PER>), dim3(blocks), dim3(256), 0, 0, out, 12345u, st);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 2); (void)hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
  std::vector<double> mhz;
  for (int b = 0; b < blocks; b++) mhz.push_back(100.0 * h[2 * b] / (double)h[2 * b + 1]);
  std::sort(mhz.begin(), mhz.end());
  double clk = mhz[blocks / 2];
  printf("%2d pairs per asm statement  w/SIMD=%d  %.3f ms clk %4.0f  %.2f SIMD-cycles per (xor+bcnt) pair\n", PER, waves_per_simd, ms, clk,
         clk * 1e6 * ms * 1e-3 / ((double)ITERS * 16 * waves_per_simd));
  (void)hipFree(out); (void)hipFree(st);
}

int main() {
  for (int w : {4, 8}) { run<1>(w); run<4>(w); run<16>(w); }
  return 0;
}
