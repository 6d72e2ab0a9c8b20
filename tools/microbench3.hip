// tools/microbench3.hip -- which VALU instructions co-issue on a CDNA4 SIMD?
// Two independent instruction streams A and B (no shared registers) interleaved
// 1:1 inside one wave; reports lanes/clk/SIMD for A alone, B alone and A+B.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 8000;

enum Op { NONE, XOR_SV, XOR_VV, AND_VV, OR_VV, BCNT, BCNT0, ADD_VV, ADD_SV, DOT4C_S, DOT4C_V, MIN_VV, MOV_S, MOV_V, ADD3, CNDMASK, CMP, LSHL_OR, FMA, BFI, PERM, XNOR_VV, ADDC, MAD24, SUB_VV, LSHLREV, ANDOR };

template <int OP>
__device__ __forceinline__ void emit(uint32_t& d, uint32_t s0, uint32_t s1, uint32_t w) {
  if (OP == XOR_SV) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(d) : "s"(w), "v"(s0));
  if (OP == XOR_VV) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(d) : "v"(s1), "v"(s0));
  if (OP == XNOR_VV) asm volatile("v_xnor_b32 %0, %1, %2" : "=v"(d) : "v"(s1), "v"(s0));
  if (OP == AND_VV) asm volatile("v_and_b32 %0, %1, %2" : "=v"(d) : "v"(s1), "v"(s0));
  if (OP == OR_VV) asm volatile("v_or_b32 %0, %1, %2" : "=v"(d) : "v"(s1), "v"(s0));
  if (OP == BCNT) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d) : "v"(s0));
  if (OP == BCNT0) asm volatile("v_bcnt_u32_b32 %0, %1, 0" : "=v"(d) : "v"(s0));
  if (OP == ADD_VV) asm volatile("v_add_u32 %0, %1, %2" : "=v"(d) : "v"(s1), "v"(s0));
  if (OP == SUB_VV) asm volatile("v_sub_u32 %0, %1, %2" : "=v"(d) : "v"(s1), "v"(s0));
  if (OP == ADD_SV) asm volatile("v_add_u32 %0, %1, %2" : "=v"(d) : "s"(w), "v"(s0));
  if (OP == DOT4C_S) asm volatile("v_dot4c_i32_i8 %0, %1, %2" : "+v"(d) : "s"(w), "v"(s0));
  if (OP == DOT4C_V) asm volatile("v_dot4c_i32_i8 %0, %1, %2" : "+v"(d) : "v"(s1), "v"(s0));
  if (OP == MIN_VV) asm volatile("v_min_i32 %0, %1, %2" : "=v"(d) : "v"(s1), "v"(s0));
  if (OP == MOV_S) asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "s"(w));
  if (OP == MOV_V) asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(s0));
  if (OP == ADD3) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(d) : "v"(s1), "v"(s0));
  if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(d) : "v"(s1), "v"(s0) : "vcc");
  if (OP == CMP) asm volatile("v_cmp_lt_i32 vcc, %0, %1" : : "v"(s1), "v"(s0) : "vcc");
  if (OP == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(d) : "v"(s1), "v"(s0));
  if (OP == FMA) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(d) : "v"(s1), "v"(s0));
  if (OP == BFI) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(d) : "v"(s1), "v"(s0));
  if (OP == PERM) asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(d) : "v"(s1), "v"(s0));
  if (OP == ADDC) asm volatile("v_addc_co_u32 %0, vcc, %1, %2, vcc" : "=v"(d) : "v"(s1), "v"(s0) : "vcc");
  if (OP == MAD24) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(d) : "v"(s1), "v"(s0));
  if (OP == LSHLREV) asm volatile("v_lshlrev_b32 %0, 1, %1" : "=v"(d) : "v"(s0));
  if (OP == ANDOR) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(d) : "v"(s1), "v"(s0));
}

template <int A, int B>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, unsigned long long* stamps) {
  uint32_t sa[8], sb[8], da[8], db[8], w = seed | 1;
  for (int i = 0; i < 8; i++) { sa[i] = threadIdx.x * 2654435761u + i * 40503u + seed; sb[i] = sa[i] * 7u + 3u; da[i] = 0; db[i] = 0; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      emit<A>(da[i], sa[i], sa[(i + 1) & 7], w);
      emit<B>(db[i], sb[i], sb[(i + 1) & 7], w);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t x = 0;
  for (int i = 0; i < 8; i++) x ^= da[i] ^ db[i];
  out[blockIdx.x * 256 + threadIdx.x] = x;
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int A, int B>
double run(int waves_per_simd) {
  int blocks = 256 * waves_per_simd;
  uint32_t* out; unsigned long long* st;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4); (void)hipMalloc(&st, blocks * 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<A, B>), dim3(blocks), dim3(256), 0, 0, out, 12345u, st);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<A, B>), dim3(blocks), dim3(256), 0, 0, out, 12345u, st);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 2); (void)hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
  std::vector<double> mhz;
  for (int b = 0; b < blocks; b++) mhz.push_back(100.0 * h[2 * b] / (double)h[2 * b + 1]);
  std::sort(mhz.begin(), mhz.end());
  int nops = (A != NONE) + (B != NONE);
  double inst = (double)ITERS * 8 * nops;
  double laneops = inst * 64 * 4 * blocks;
  double clk = mhz[blocks / 2];
  (void)hipFree(out); (void)hipFree(st);
  // cycles per wave-instruction per SIMD
  return (clk * 1e6 * ms * 1e-3) / (inst * waves_per_simd);
}

#define SINGLE(NAME, OP) printf("%-12s alone: %5.2f cyc/inst (8 w/SIMD)  %5.2f (2 w/SIMD)\n", NAME, run<OP, NONE>(8), run<OP, NONE>(2));
#define PAIR(NA, A, NB, B) printf("%-12s + %-12s: %5.2f cyc per PAIR (8 w/SIMD)   [%5.2f at 2 w/SIMD]\n", NA, NB, 2 * run<A, B>(8), 2 * run<A, B>(2));

int main() {
  SINGLE("xor s,v", XOR_SV) SINGLE("xor v,v", XOR_VV) SINGLE("xnor v,v", XNOR_VV) SINGLE("and v,v", AND_VV) SINGLE("or v,v", OR_VV)
  SINGLE("bcnt acc", BCNT) SINGLE("bcnt +0", BCNT0) SINGLE("add v,v", ADD_VV) SINGLE("sub v,v", SUB_VV) SINGLE("add s,v", ADD_SV)
  SINGLE("dot4c s", DOT4C_S) SINGLE("dot4c v", DOT4C_V) SINGLE("min v,v", MIN_VV) SINGLE("mov s", MOV_S) SINGLE("mov v", MOV_V)
  SINGLE("add3", ADD3) SINGLE("cndmask", CNDMASK) SINGLE("cmp", CMP) SINGLE("lshl_or", LSHL_OR) SINGLE("fma", FMA)
  SINGLE("bfi", BFI) SINGLE("perm", PERM) SINGLE("addc", ADDC) SINGLE("mad24", MAD24) SINGLE("lshlrev", LSHLREV) SINGLE("and_or", ANDOR)
  PAIR("bcnt", BCNT, "bcnt", BCNT)
  PAIR("bcnt", BCNT, "xor v,v", XOR_VV)
  PAIR("bcnt", BCNT, "xor s,v", XOR_SV)
  PAIR("bcnt", BCNT, "and v,v", AND_VV)
  PAIR("bcnt", BCNT, "add v,v", ADD_VV)
  PAIR("bcnt", BCNT, "dot4c v", DOT4C_V)
  PAIR("bcnt", BCNT, "fma", FMA)
  PAIR("bcnt", BCNT, "mov s", MOV_S)
  PAIR("bcnt", BCNT, "cmp", CMP)
  PAIR("xor s,v", XOR_SV, "xor v,v", XOR_VV)
  PAIR("xor s,v", XOR_SV, "add v,v", ADD_VV)
  PAIR("xor s,v", XOR_SV, "xor s,v", XOR_SV)
  PAIR("xor v,v", XOR_VV, "xor v,v", XOR_VV)
  PAIR("dot4c s", DOT4C_S, "xor v,v", XOR_VV)
  PAIR("dot4c v", DOT4C_V, "add v,v", ADD_VV)
  PAIR("dot4c s", DOT4C_S, "dot4c s", DOT4C_S)
  PAIR("add3", ADD3, "xor v,v", XOR_VV)
  PAIR("min v,v", MIN_VV, "xor v,v", XOR_VV)
  return 0;
}
