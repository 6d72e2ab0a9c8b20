// tools/microbench2.hip -- second-generation VALU microbench: independent
// destinations (no RAW chains), long runs, and the real shader clock
// (s_memtime ticks per s_memrealtime 100 MHz tick).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 20000;

// 16 independent destinations d[i] written from 16 sources s[i]; sources are loop-invariant
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, unsigned long long* stamps) {
  uint32_t s[16], d[16], w = seed | 1;
  for (int i = 0; i < 16; i++) { s[i] = threadIdx.x * 2654435761u + i * 40503u + seed; d[i] = 0; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (MODE == 0) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(d[i]) : "s"(w), "v"(s[i]));
      if (MODE == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d[i]) : "v"(s[i]));
      if (MODE == 2) { uint32_t t; asm volatile("v_xor_b32 %0, %2, %3\n\tv_bcnt_u32_b32 %1, %0, %1" : "=&v"(t), "+v"(d[i]) : "s"(w), "v"(s[i])); }
      if (MODE == 3) asm volatile("v_dot4c_i32_i8 %0, %1, %2" : "+v"(d[i]) : "s"(w), "v"(s[i]));
      if (MODE == 4) asm volatile("v_add_u32 %0, %1, %2" : "=v"(d[i]) : "s"(w), "v"(s[i]));
      if (MODE == 5) asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(d[i]) : "v"(s[i]), "v"(s[(i + 1) & 15]), "v"(s[(i + 2) & 15]));
      if (MODE == 6) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(d[i]) : "v"(s[i]), "v"(s[(i + 1) & 15]));
      if (MODE == 7) asm volatile("v_and_b32 %0, %1, %2" : "=v"(d[i]) : "s"(w), "v"(s[i]));
      if (MODE == 8) asm volatile("v_min_i32 %0, %1, %2" : "=v"(d[i]) : "s"(w), "v"(s[i]));
      if (MODE == 9) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(d[i]) : "v"(s[i]), "v"(s[(i + 1) & 15]));
      if (MODE == 10) { uint32_t t; asm volatile("v_xor_b32 %0, %2, %3\n\tv_add_u32 %1, %0, %1" : "=&v"(t), "+v"(d[i]) : "s"(w), "v"(s[i])); }
      if (MODE == 11) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(d[i]) : "v"(s[(i + 1) & 15]), "v"(s[i]));
      if (MODE == 12) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(d[i]) : "v"(s[i]), "v"(s[(i + 1) & 15]));
      if (MODE == 13) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(d[i]) : "v"(s[i]), "v"(s[(i + 1) & 15]));
      if (MODE == 14) asm volatile("v_bcnt_u32_b32 %0, %1, 0" : "=v"(d[i]) : "v"(s[i]));
      if (MODE == 15) asm volatile("v_pk_add_u16 %0, %1, %2" : "=v"(d[i]) : "v"(s[i]), "v"(s[(i + 1) & 15]));
      if (MODE == 16) asm volatile("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(d[i]) : "v"(s[i]), "v"(s[(i + 1) & 15]));
      if (MODE == 17) asm volatile("v_xad_u32 %0, %1, %2, %0" : "+v"(d[i]) : "v"(s[i]), "v"(s[(i + 1) & 15]));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t x = 0;
  for (int i = 0; i < 16; i++) x ^= d[i];
  out[blockIdx.x * 256 + threadIdx.x] = x;
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int MODE>
int run(const char* name, int ops_per_inner, int waves_per_simd) {
  int blocks = 256 * waves_per_simd;
  uint32_t* out; unsigned long long* st;
  CHK(hipMalloc(&out, (size_t)blocks * 256 * 4)); CHK(hipMalloc(&st, blocks * 16));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 12345u, st);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 12345u, st);
  CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(blocks * 2); CHK(hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost));
  std::vector<double> mhz;
  double avgt = 0;
  for (int b = 0; b < blocks; b++) { avgt += h[2 * b]; mhz.push_back(100.0 * h[2 * b] / (double)h[2 * b + 1]); }
  avgt /= blocks;
  std::sort(mhz.begin(), mhz.end());
  double inst = (double)ITERS * 16 * ops_per_inner;
  double laneops = inst * 64 * 4 * blocks;
  double tops = laneops / (ms * 1e-3) / 1e12;
  double clk = mhz[blocks / 2];
  double lanes_per_clk_simd = tops * 1e12 / (clk * 1e6) / 1024.0;
  printf("%-26s w/SIMD=%d %7.3f ms %6.2f T lane-op/s  clk=%5.0f MHz  lanes/clk/SIMD=%5.2f  ticks/inst/wave=%5.2f\n", name,
         waves_per_simd, ms, tops, clk, lanes_per_clk_simd, avgt / inst);
  (void)hipFree(out); (void)hipFree(st);
  return 0;
}

int main() {
  for (int w : {2, 8}) {
    run<0>("v_xor_b32 s,v", 1, w);
    run<11>("v_xor_b32 v,v", 1, w);
    run<7>("v_and_b32 s,v", 1, w);
    run<1>("v_bcnt_u32_b32 acc", 1, w);
    run<14>("v_bcnt_u32_b32 +0", 1, w);
    run<2>("xor+bcnt pair", 2, w);
    run<10>("xor+add pair", 2, w);
    run<3>("v_dot4c_i32_i8", 1, w);
    run<13>("v_dot4_u32_u8", 1, w);
    run<12>("v_sad_u8", 1, w);
    run<4>("v_add_u32", 1, w);
    run<5>("v_add3_u32", 1, w);
    run<8>("v_min_i32", 1, w);
    run<9>("v_mad_u32_u24", 1, w);
    run<15>("v_pk_add_u16", 1, w);
    run<16>("v_lshl_or_b32", 1, w);
    run<17>("v_xad_u32", 1, w);
    run<6>("v_fma_f32", 1, w);
  }
  return 0;
}
