// tools/microbench5.hip -- does the FP pipe co-issue with the integer pipe ACROSS waves?
// Even waves of a block run an FP stream, odd waves an INT stream; compare with all-FP / all-INT.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
constexpr int ITERS = 8000;

enum { FMA_VV, FMA_SV, PKFMA_VV, PKFMA_SV, BCNT, XORBCNT, DOT4C, FMAC_S, MUL_S };

template <int OP>
__device__ __forceinline__ void body(uint32_t (&d)[8], uint32_t (&s)[8], uint32_t w, uint64_t w2) {
#pragma unroll
  for (int i = 0; i < 8; i++) {
    if (OP == FMA_VV) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(d[i]) : "v"(s[i]), "v"(s[(i + 1) & 7]));
    if (OP == FMA_SV) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(d[i]) : "s"(w), "v"(s[i]));
    if (OP == FMAC_S) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(d[i]) : "s"(w), "v"(s[i]));
    if (OP == MUL_S) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d[i]) : "s"(w), "v"(s[i]));
    if (OP == PKFMA_VV) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(uint64_t*)&d[i & 6]) : "v"(*(uint64_t*)&s[i & 6]), "v"(*(uint64_t*)&s[(i + 2) & 6]));
    if (OP == PKFMA_SV) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(uint64_t*)&d[i & 6]) : "s"(w2), "v"(*(uint64_t*)&s[i & 6]));
    if (OP == BCNT) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d[i]) : "v"(s[i]));
    if (OP == XORBCNT) { uint32_t t; asm volatile("v_xor_b32 %0, %2, %3\n\tv_bcnt_u32_b32 %1, %0, %1" : "=&v"(t), "+v"(d[i]) : "s"(w), "v"(s[i])); }
    if (OP == DOT4C) asm volatile("v_dot4c_i32_i8 %0, %1, %2" : "+v"(d[i]) : "s"(w), "v"(s[i]));
  }
}

template <int A, int B>  // even waves run A, odd waves run B
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, unsigned long long* stamps) {
  uint32_t s[8], d[8], w = seed | 1;
  uint64_t w2 = ((uint64_t)seed << 32) | 77u;
  for (int i = 0; i < 8; i++) { s[i] = threadIdx.x * 2654435761u + i * 40503u + seed; d[i] = 0; }
  const bool odd = (threadIdx.x >> 6) & 1;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (!odd) { for (int it = 0; it < ITERS; it++) body<A>(d, s, w, w2); }
  else      { for (int it = 0; it < ITERS; it++) body<B>(d, s, w, w2); }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t x = 0;
  for (int i = 0; i < 8; i++) x ^= d[i];
  out[blockIdx.x * 256 + threadIdx.x] = x;
  if ((threadIdx.x & 63) == 0) { int wv = blockIdx.x * 4 + (threadIdx.x >> 6); stamps[wv * 2] = t1 - t0; stamps[wv * 2 + 1] = r1 - r0; }
}

template <int A, int B>
void run(const char* name, int waves_per_simd) {
  int blocks = 256 * waves_per_simd;
  uint32_t* out; unsigned long long* st;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4); (void)hipMalloc(&st, blocks * 4 * 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<A, B>), dim3(blocks), dim3(256), 0, 0, out, 12345u, st);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<A, B>), dim3(blocks), dim3(256), 0, 0, out, 12345u, st);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 8); (void)hipMemcpy(h.data(), st, blocks * 64, hipMemcpyDeviceToHost);
  std::vector<double> mhz;
  for (int b = 0; b < blocks * 4; b++) mhz.push_back(100.0 * h[2 * b] / (double)h[2 * b + 1]);
  std::sort(mhz.begin(), mhz.end());
  double clk = mhz[mhz.size() / 2];
  double inst_per_wave = (double)ITERS * 8 * ((A == XORBCNT) ? 2 : 1);  // approx for A
  printf("%-36s w/SIMD=%d  %.3f ms  clk %4.0f MHz  SIMD-cycles per (A-wave instr): %.2f\n", name, waves_per_simd, ms, clk,
         clk * 1e6 * ms * 1e-3 / ((double)ITERS * 8 * waves_per_simd));
  (void)inst_per_wave;
  (void)hipFree(out); (void)hipFree(st);
}

int main() {
  for (int w : {4, 8}) {
    run<FMA_VV, FMA_VV>("all waves fma v,v", w);
    run<FMA_SV, FMA_SV>("all waves fma s,v", w);
    run<FMAC_S, FMAC_S>("all waves fmac s,v", w);
    run<MUL_S, MUL_S>("all waves mul s,v", w);
    run<PKFMA_VV, PKFMA_VV>("all waves pk_fma v,v", w);
    run<PKFMA_SV, PKFMA_SV>("all waves pk_fma s,v", w);
    run<BCNT, BCNT>("all waves bcnt", w);
    run<DOT4C, DOT4C>("all waves dot4c", w);
    run<XORBCNT, XORBCNT>("all waves xor+bcnt (2 instr)", w);
    run<FMA_VV, BCNT>("half fma v,v | half bcnt", w);
    run<FMA_SV, BCNT>("half fma s,v | half bcnt", w);
    run<PKFMA_VV, BCNT>("half pk_fma v,v | half bcnt", w);
    run<PKFMA_SV, XORBCNT>("half pk_fma s,v | half xor+bcnt", w);
    run<FMA_SV, XORBCNT>("half fma s,v | half xor+bcnt", w);
    run<DOT4C, XORBCNT>("half dot4c | half xor+bcnt", w);
  }
  return 0;
}
