// tools/mfma_fp4_probe.hip -- facts the matrix-pipe side experiment ("Layer 1 on the matrix
// pipe") rests on, measured rather than assumed:
//   1. operand lane map of v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (E2M1) A and B: lane l = (r = l & 31,
//      h = l >> 5), nibble j (0..31, low nibble of byte 0 first) of the lane's 16 operand bytes is
//      k = 32 h + j of row r (A) / column r (B)  -- checked as "the same (h, j) -> k on both operands" with
//      exact random data against a host product;
//   2. C/D map: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5);
//   3. scale operand 0x7F (E8M0 2^0) leaves the product unscaled; +-1 are the nibbles 0x2 / 0xA;
//   4. issue rate: cycles per MFMA with 4 independent accumulators, 1 and 2 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_fp4_probe mfma_fp4_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
  } while (0)

__global__ void k_one(const uint4 *a, const uint4 *b, float *d) {
  const int l = threadIdx.x;
  const uint4 av = a[l], bv = b[l];
  v8i A = {(int)av.x, (int)av.y, (int)av.z, (int)av.w, 0, 0, 0, 0};
  v8i B = {(int)bv.x, (int)bv.y, (int)bv.z, (int)bv.w, 0, 0, 0, 0};
  v16f acc = {0};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  for (int i = 0; i < 16; i++) d[l * 16 + i] = acc[i];
}

__global__ void k_rate(const uint4 *a, const uint4 *b, float *d, int iters, long long *cycles) {
  const int l = threadIdx.x & 63;
  const uint4 av = a[l], bv = b[l];
  v8i A = {(int)av.x, (int)av.y, (int)av.z, (int)av.w, 0, 0, 0, 0};
  v8i B = {(int)bv.x, (int)bv.y, (int)bv.z, (int)bv.w, 0, 0, 0, 0};
  v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c0, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c1, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c2, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c3, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
  v16f s = c0 + c1 + c2 + c3;
  float acc = 0;
  for (int i = 0; i < 16; i++) acc += s[i];
  d[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static float fp4(int nib) {
  static const float mag[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
  return (nib & 8) ? -mag[nib & 7] : mag[nib & 7];
}

int main() {
  std::vector<uint8_t> a(64 * 16), b(64 * 16);
  srand(7);
  for (auto &x : a) x = (uint8_t)(rand() & 0xFF);
  for (auto &x : b) x = (uint8_t)(rand() & 0xFF);
  uint4 *da, *db;
  float *dd;
  long long *dc;
  CK(hipMalloc(&da, 1024));
  CK(hipMalloc(&db, 1024));
  CK(hipMalloc(&dd, 1 << 22));
  CK(hipMalloc(&dc, 8));
  CK(hipMemcpy(da, a.data(), 1024, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, b.data(), 1024, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, da, db, dd);
  std::vector<float> d(64 * 16);
  CK(hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost));
  auto nib = [](const std::vector<uint8_t> &v, int lane, int j) { return (v[lane * 16 + j / 2] >> (4 * (j & 1))) & 15; };
  int bad = 0;
  for (int l = 0; l < 64; l++)
    for (int reg = 0; reg < 16; reg++) {
      const int col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
      float want = 0;
      for (int h = 0; h < 2; h++)
        for (int j = 0; j < 32; j++) want += fp4(nib(a, row + 32 * h, j)) * fp4(nib(b, col + 32 * h, j));
      if (want != d[l * 16 + reg]) {
        if (bad < 5) printf("mismatch lane %d reg %d: got %g want %g\n", l, reg, d[l * 16 + reg], want);
        bad++;
      }
    }
  printf("FP4 32x32x64 operand/result map as assumed: %s (%d mismatches of 1024)\n", bad ? "NO" : "yes", bad);
  // +-1 only
  for (auto &x : a) x = (uint8_t)(((rand() & 1) ? 0x2 : 0xA) | (((rand() & 1) ? 0x2 : 0xA) << 4));
  for (auto &x : b) x = (uint8_t)(((rand() & 1) ? 0x2 : 0xA) | (((rand() & 1) ? 0x2 : 0xA) << 4));
  CK(hipMemcpy(da, a.data(), 1024, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, b.data(), 1024, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, da, db, dd);
  CK(hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost));
  bad = 0;
  for (int l = 0; l < 64; l++)
    for (int reg = 0; reg < 16; reg++) {
      const int col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
      int want = 0;
      for (int h = 0; h < 2; h++)
        for (int j = 0; j < 32; j++) want += (nib(a, row + 32 * h, j) == 2 ? 1 : -1) * (nib(b, col + 32 * h, j) == 2 ? 1 : -1);
      bad += (float)want != d[l * 16 + reg];
    }
  printf("+-1 operands (nibbles 0x2 / 0xA): %s\n", bad ? "MISMATCH" : "exact");
  // issue rate
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  for (int wps : {1, 2, 4}) {
    const int iters = 20000, threads = 256 * wps, blocks = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(threads), 0, 0, da, db, dd, 100, dc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(threads), 0, 0, da, db, dd, iters, dc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    long long cyc = 0;
    CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost));
    const double mfma = (double)iters * 4 * wps * 4 * blocks;  // per chip
    printf("waves/SIMD=%d: %.3f ms, %.1f shader ticks per MFMA per SIMD (s_memtime, 100 MHz units x clock ratio not applied), "
           "%.2f P MAC/s = %.2f PFLOP/s dense\n",
           wps, ms, (double)cyc / (iters * 4.0 * wps), mfma * 65536 / (ms * 1e-3) / 1e15, 2 * mfma * 65536 / (ms * 1e-3) / 1e15);
  }
  return 0;
}
