// tools/microbench.hip -- VALU issue-rate microbenchmarks for the instructions the
// hot path is made of (v_xor_b32, v_bcnt_u32_b32, v_dot4c_i32_i8, ...).
// Prints lane-ops per clock per CU so that bench.py's "valu" ceiling is a measured one.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 4096;

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, long long* cycles) {
  uint32_t a[8], w = seed | 1;
  for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 2654435761u + i * 40503u + seed;
  long long t0 = clock64();
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (MODE == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "s"(w));
      if (MODE == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
      if (MODE == 2) { uint32_t t; asm volatile("v_xor_b32 %0, %2, %1\n\tv_bcnt_u32_b32 %1, %0, %1" : "=&v"(t), "+v"(a[i]) : "s"(w)); }
      if (MODE == 3) asm volatile("v_dot4c_i32_i8 %0, %1, %2" : "+v"(a[i]) : "s"(w), "v"(a[(i + 1) & 7]));
      if (MODE == 4) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "s"(w));
      if (MODE == 5) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
      if (MODE == 6) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
      if (MODE == 7) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "s"(w));
      if (MODE == 8) asm volatile("v_min_i32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
    }
  }
  long long t1 = clock64();
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s ^= a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
int run(const char* name, int ops_per_inner, int waves_per_simd) {
  int blocks = 256 * waves_per_simd;  // 4 waves per block, 1 block per CU per "wave per SIMD"
  uint32_t* out; long long* cyc;
  CHK(hipMalloc(&out, (size_t)blocks * 256 * 4)); CHK(hipMalloc(&cyc, blocks * 8));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 12345u, cyc);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 12345u, cyc);
  CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> h(blocks); CHK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
  double avg = 0; for (auto c : h) avg += c; avg /= blocks;
  double inst = (double)ITERS * 8 * ops_per_inner;            // VALU instr per wave
  double laneops = inst * 64 * 4 * blocks;                     // total lane-ops
  printf("%-28s waves/SIMD=%d  %.3f ms  %.2f T lane-op/s  cycles/inst/wave=%.2f (clock64 ticks)\n", name, waves_per_simd, ms,
         laneops / (ms * 1e-3) / 1e12, avg / inst);
  hipFree(out); hipFree(cyc);
  return 0;
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_xor_b32 (sgpr)", 1, w);
    run<1>("v_bcnt_u32_b32 (vgpr)", 1, w);
    run<7>("v_bcnt_u32_b32 (sgpr src)", 1, w);
    run<2>("v_xor+v_bcnt pair", 2, w);
    run<3>("v_dot4c_i32_i8", 1, w);
    run<4>("v_and_b32", 1, w);
    run<5>("v_add3_u32", 1, w);
    run<6>("v_add_u32", 1, w);
    run<8>("v_min_i32", 1, w);
  }
  return 0;
}
