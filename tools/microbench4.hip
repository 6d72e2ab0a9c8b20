// tools/microbench4.hip -- schedules of the XNOR-popcount inner loop (one neuron x a 2x2 quad,
// 9 weight words): what does a 64-bit word-MAC cost per wave per SIMD in the best case?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
constexpr int ITERS = 3000;

// V: 0 = xor(s,v)+bcnt alternating, 4 independent accumulators
//    1 = 4 xors then 4 bcnts (grouped)
//    2 = like 0 but weights first moved to VGPRs (v_mov) and xor v,v
//    3 = like 0 + epilogue per neuron: v_min3, v_min, v_alignbit (new bit insert)
//    4 = like 0 + epilogue: cmp + cndmask + or (what the compiler emits today)
//    5 = 8 xors then 8 bcnts
template <int V>
__global__ __launch_bounds__(256) void k(uint32_t* out, const uint32_t* wts, unsigned long long* stamps) {
  uint32_t win[32];
  for (int i = 0; i < 32; i++) win[i] = threadIdx.x * 2654435761u + i * 40503u;
  uint32_t bits = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; it++) {
    // 18 weight dwords "in SGPRs": derive from it so the compiler keeps them scalar
    uint32_t w[18];
#pragma unroll
    for (int j = 0; j < 18; j++) w[j] = __builtin_amdgcn_readfirstlane(wts[(it * 18 + j) & 1023]);
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
    for (int j = 0; j < 18; j++) {
      // 4 pixels share weight dword j; their activation dwords are win[(j + 2p) & 31]
      uint32_t x0 = win[j & 31], x1 = win[(j + 2) & 31], x2 = win[(j + 8) & 31], x3 = win[(j + 10) & 31];
      uint32_t t0_, t1_, t2_, t3_;
      if (V == 0 || V == 3 || V == 4) {
        asm volatile("v_xor_b32 %0, %2, %3\n\tv_bcnt_u32_b32 %1, %0, %1" : "=&v"(t0_), "+v"(a0) : "s"(w[j]), "v"(x0));
        asm volatile("v_xor_b32 %0, %2, %3\n\tv_bcnt_u32_b32 %1, %0, %1" : "=&v"(t1_), "+v"(a1) : "s"(w[j]), "v"(x1));
        asm volatile("v_xor_b32 %0, %2, %3\n\tv_bcnt_u32_b32 %1, %0, %1" : "=&v"(t2_), "+v"(a2) : "s"(w[j]), "v"(x2));
        asm volatile("v_xor_b32 %0, %2, %3\n\tv_bcnt_u32_b32 %1, %0, %1" : "=&v"(t3_), "+v"(a3) : "s"(w[j]), "v"(x3));
      } else if (V == 1) {
        asm volatile("v_xor_b32 %0, %8, %9\n\tv_xor_b32 %1, %8, %10\n\tv_xor_b32 %2, %8, %11\n\tv_xor_b32 %3, %8, %12\n\t"
                     "v_bcnt_u32_b32 %4, %0, %4\n\tv_bcnt_u32_b32 %5, %1, %5\n\tv_bcnt_u32_b32 %6, %2, %6\n\tv_bcnt_u32_b32 %7, %3, %7"
                     : "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                     : "s"(w[j]), "v"(x0), "v"(x1), "v"(x2), "v"(x3));
      } else if (V == 2) {
        uint32_t wv;
        asm volatile("v_mov_b32 %0, %1" : "=v"(wv) : "s"(w[j]));
        asm volatile("v_xor_b32 %0, %2, %3\n\tv_bcnt_u32_b32 %1, %0, %1" : "=&v"(t0_), "+v"(a0) : "v"(wv), "v"(x0));
        asm volatile("v_xor_b32 %0, %2, %3\n\tv_bcnt_u32_b32 %1, %0, %1" : "=&v"(t1_), "+v"(a1) : "v"(wv), "v"(x1));
        asm volatile("v_xor_b32 %0, %2, %3\n\tv_bcnt_u32_b32 %1, %0, %1" : "=&v"(t2_), "+v"(a2) : "v"(wv), "v"(x2));
        asm volatile("v_xor_b32 %0, %2, %3\n\tv_bcnt_u32_b32 %1, %0, %1" : "=&v"(t3_), "+v"(a3) : "v"(wv), "v"(x3));
      } else if (V == 5) {
        if ((j & 1) == 0) {
          uint32_t y0 = win[(j + 1) & 31], y1 = win[(j + 3) & 31], y2 = win[(j + 9) & 31], y3 = win[(j + 11) & 31];
          uint32_t u0, u1, u2, u3;
          asm volatile("v_xor_b32 %0, %12, %14\n\tv_xor_b32 %1, %12, %15\n\tv_xor_b32 %2, %12, %16\n\tv_xor_b32 %3, %12, %17\n\t"
                       "v_xor_b32 %4, %13, %18\n\tv_xor_b32 %5, %13, %19\n\tv_xor_b32 %6, %13, %20\n\tv_xor_b32 %7, %13, %21\n\t"
                       "v_bcnt_u32_b32 %8, %0, %8\n\tv_bcnt_u32_b32 %9, %1, %9\n\tv_bcnt_u32_b32 %10, %2, %10\n\tv_bcnt_u32_b32 %11, %3, %11\n\t"
                       "v_bcnt_u32_b32 %8, %4, %8\n\tv_bcnt_u32_b32 %9, %5, %9\n\tv_bcnt_u32_b32 %10, %6, %10\n\tv_bcnt_u32_b32 %11, %7, %11"
                       : "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_), "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3),
                         "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                       : "s"(w[j]), "s"(w[j + 1]), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(y0), "v"(y1), "v"(y2), "v"(y3));
        }
      }
    }
    if (V == 3) {
      uint32_t m;
      asm volatile("v_min3_i32 %0, %1, %2, %3\n\tv_min_i32 %0, %0, %4\n\tv_alignbit_b32 %5, %5, %0, 31"
                   : "=&v"(m), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(bits));
    } else if (V == 4) {
      int mn = min(min((int)a0, (int)a1), min((int)a2, (int)a3));
      bits |= (mn < (int)w[0]) ? (1u << (it & 31)) : 0u;
    } else {
      bits += a0 + a1 + a2 + a3;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = bits;
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int V>
void run(const char* name, int waves_per_simd, const uint32_t* wts) {
  int blocks = 256 * waves_per_simd;
  uint32_t* out; unsigned long long* st;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4); (void)hipMalloc(&st, blocks * 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<V>), dim3(blocks), dim3(256), 0, 0, out, wts, st);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<V>), dim3(blocks), dim3(256), 0, 0, out, wts, st);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 2); (void)hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
  std::vector<double> mhz;
  for (int b = 0; b < blocks; b++) mhz.push_back(100.0 * h[2 * b] / (double)h[2 * b + 1]);
  std::sort(mhz.begin(), mhz.end());
  double clk = mhz[blocks / 2];
  double wordmacs = (double)ITERS * 9 * 4;  // 64-bit word-MACs per wave (9 weight words x 4 pixels)
  double cyc = (clk * 1e6 * ms * 1e-3) / (wordmacs * waves_per_simd);
  printf("%-44s w/SIMD=%d  %.3f ms  clk %4.0f MHz  %.2f SIMD-cycles per 64-bit word-MAC\n", name, waves_per_simd, ms, clk, cyc);
  (void)hipFree(out); (void)hipFree(st);
}

int main() {
  uint32_t* wts; (void)hipMalloc(&wts, 4096);
  std::vector<uint32_t> h(1024); for (int i = 0; i < 1024; i++) h[i] = i * 2654435761u;
  (void)hipMemcpy(wts, h.data(), 4096, hipMemcpyHostToDevice);
  for (int w : {4, 8}) {
    run<0>("alternating xor(s,v)/bcnt", w, wts);
    run<1>("4 xor then 4 bcnt", w, wts);
    run<5>("8 xor then 8 bcnt", w, wts);
    run<2>("v_mov weights to VGPR, xor v,v / bcnt", w, wts);
    run<3>("alternating + min3/min/alignbit epilogue", w, wts);
    run<4>("alternating + compiler cmp/cndmask/or epilogue", w, wts);
  }
  return 0;
}
