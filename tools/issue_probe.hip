// tools/issue_probe.hip -- the integer-pipe issue ceiling, measured on the device that runs the benchmark.
//
// bench.py quotes the networks' rates against an "issue floor": word-MACs x SIMD-cycles per (logic op, v_bcnt) pair.
// Rounds 1-3 took the cycles per pair (6.3 / 6.6 / 2 x 6.35) and the clock (2.38 GHz) from microbenchmarks run once, on
// another box (profiles/r01_microbench9_nop_cadence.txt); MI355X devices differ by several percent in the clock they hold
// under load.  This is the same loop -- 16 pairs per asm statement in the cadence the product kernels are built to
// (pair, one s_nop 0), four accumulator chains, weights from an SGPR -- as a tiny library bench.py calls OUTSIDE its
// timed region: SIMD-cycles per pair from s_memtime, the clock held under this load from s_memtime / s_memrealtime
// (100 MHz), at 4 and 8 waves per SIMD.  Not part of the product libraries.
//
//   hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 tools/issue_probe.hip -o tools/libissue_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace {
constexpr int ITERS = 10000;  // ~1.7 ms per launch at 4 waves per SIMD: the launch itself is then ~1 % of the wall time

#define A16 "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15])
// operands: %0 dummy, %1..%4 accumulators, %5..%8 temporaries, %9 the SGPR weight, %10..%25 the 16 input dwords
#define XB(t, acc, in) "v_xor_b32 %" #t ", %9, %" #in "\n\tv_bcnt_u32_b32 %" #acc ", %" #t ", %" #acc "\n\ts_nop 0\n\t"
#define TB(t, acc, in, in2) "v_bitop3_b32 %" #t ", %9, %" #in ", %" #in2 " bitop3:0x28\n\tv_bcnt_u32_b32 %" #acc ", %" #t ", %" #acc "\n\ts_nop 0\n\t"
// W2A2: z = za & w_nz -> bcnt; m = bitop3(za & w_nz, sa, w_s) -> bcnt: two pairs per 32 synapses
#define TT(t, u, accz, accm, in, in2) \
  "v_and_b32 %" #t ", %9, %" #in "\n\tv_bcnt_u32_b32 %" #accz ", %" #t ", %" #accz "\n\ts_nop 0\n\tv_bitop3_b32 %" #u ", %9, %" #in2 ", %" #t \
  " bitop3:0x28\n\tv_bcnt_u32_b32 %" #accm ", %" #u ", %" #accm "\n\ts_nop 0\n\t"

template <int P>
__global__ __launch_bounds__(256) void k_probe(uint32_t *out, uint32_t seed, unsigned long long *stamps) {
  uint32_t a[16], acc[4] = {0, 0, 0, 0}, t0_, t1_, t2_, t3_, dummy;
  const uint32_t ws = seed | 1;
  for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 2654435761u + i * 40503u + seed;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; it++) {
    if constexpr (P == 0)
      asm volatile(XB(5, 1, 10) XB(6, 2, 11) XB(7, 3, 12) XB(8, 4, 13) XB(5, 1, 14) XB(6, 2, 15) XB(7, 3, 16) XB(8, 4, 17) XB(5, 1, 18) XB(6, 2, 19)
                       XB(7, 3, 20) XB(8, 4, 21) XB(5, 1, 22) XB(6, 2, 23) XB(7, 3, 24) XB(8, 4, 25)
                   : "=&v"(dummy), "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_)
                   : "s"(ws), A16
                   : "memory");
    else if constexpr (P == 1)
      asm volatile(TB(5, 1, 10, 11) TB(6, 2, 11, 12) TB(7, 3, 12, 13) TB(8, 4, 13, 14) TB(5, 1, 14, 15) TB(6, 2, 15, 16) TB(7, 3, 16, 17) TB(8, 4, 17, 18)
                       TB(5, 1, 18, 19) TB(6, 2, 19, 20) TB(7, 3, 20, 21) TB(8, 4, 21, 22) TB(5, 1, 22, 23) TB(6, 2, 23, 24) TB(7, 3, 24, 25) TB(8, 4, 25, 10)
                   : "=&v"(dummy), "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_)
                   : "s"(ws), A16
                   : "memory");
    else
      asm volatile(TT(5, 7, 1, 3, 10, 18) TT(6, 8, 2, 4, 11, 19) TT(5, 7, 1, 3, 12, 20) TT(6, 8, 2, 4, 13, 21) TT(5, 7, 1, 3, 14, 22) TT(6, 8, 2, 4, 15, 23)
                       TT(5, 7, 1, 3, 16, 24) TT(6, 8, 2, 4, 17, 25)
                   : "=&v"(dummy), "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_)
                   : "s"(ws), A16
                   : "memory");
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 2] = c1 - c0;
    stamps[blockIdx.x * 2 + 1] = r1 - r0;
  }
}

template <int P>
int run(int blocks, uint32_t *out, unsigned long long *st, hipEvent_t e0, hipEvent_t e1, float *ms) {
  if (hipEventRecord(e0, nullptr) != hipSuccess) return -1;
  hipLaunchKernelGGL((k_probe<P>), dim3(blocks), dim3(256), 0, nullptr, out, 12345u, st);
  if (hipEventRecord(e1, nullptr) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return -1;
  return hipEventElapsedTime(ms, e0, e1) == hipSuccess ? 0 : -1;
}
}  // namespace

// pattern 0: v_xor(s, v) + v_bcnt (W1A1), 1: v_bitop3 + v_bcnt (W1A2), 2: the W2A2 form, two pairs per 32 synapses.
// waves_per_simd: 4 or 8 (one 256-thread block is one wave on each SIMD of a CU).  Keeps the GPU busy with the same kernel
// for `settle_ms` first (the clock under load is what is asked for), then takes the fastest of 3 launches.
// Out: cycles_per_pair_wall -- the launch's duration by HIP events x the clock / pairs issued per SIMD (what the round-1
// microbenchmarks reported; contains the launch's own few microseconds); cycles_per_pair_kernel -- the median block's
// in-kernel cycle count (s_memtime around the loop) / pairs issued per SIMD, meaningful when all blocks are resident
// together (4 waves per SIMD: one 256-thread block per wave slot row; at 8 the figure is reported as measured, the blocks
// of this grid do not all run side by side); the clock the kernel ran at (MHz, median over blocks); the launch's
// duration (ms).  Returns 0, or -1 on a HIP error.
extern "C" int issue_probe_run(int pattern, int waves_per_simd, int settle_ms, double *cycles_per_pair_wall, double *cycles_per_pair_kernel,
                               double *clock_mhz, double *launch_ms) {
  if (pattern < 0 || pattern > 2 || (waves_per_simd != 4 && waves_per_simd != 8)) return -1;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
  const int blocks = prop.multiProcessorCount * waves_per_simd;
  uint32_t *out = nullptr;
  unsigned long long *st = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = -1;
  float ms = 0.f, best = 1e30f;
  std::vector<unsigned long long> h((size_t)blocks * 2), hb;
  auto one = [&](float *t) { return pattern == 0 ? run<0>(blocks, out, st, e0, e1, t) : pattern == 1 ? run<1>(blocks, out, st, e0, e1, t) : run<2>(blocks, out, st, e0, e1, t); };
  if (hipMalloc(&out, (size_t)blocks * 256 * 4) != hipSuccess || hipMalloc(&st, (size_t)blocks * 16) != hipSuccess) goto done;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) goto done;
  for (float spent = 0.f; spent < (float)settle_ms;) {
    if (one(&ms)) goto done;
    spent += ms > 0.01f ? ms : 0.01f;
  }
  for (int rep = 0; rep < 3; rep++) {
    if (one(&ms)) goto done;
    if (hipMemcpy(h.data(), st, (size_t)blocks * 16, hipMemcpyDeviceToHost) != hipSuccess) goto done;
    if (ms < best) { best = ms; hb = h; }
  }
  {
    std::vector<double> cyc, mhz;
    for (int b = 0; b < blocks; b++) {
      cyc.push_back((double)hb[2 * b]);
      mhz.push_back(100.0 * (double)hb[2 * b] / (double)hb[2 * b + 1]);
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(mhz.begin(), mhz.end());
    // a SIMD issues for `waves_per_simd` waves: ITERS x 16 pairs each
    const double pairs = (double)ITERS * 16.0 * waves_per_simd;
    *cycles_per_pair_kernel = cyc[blocks / 2] / pairs;
    *clock_mhz = mhz[blocks / 2];
    *cycles_per_pair_wall = (double)best * 1e-3 * (*clock_mhz) * 1e6 / pairs;
    *launch_ms = best;
    rc = 0;
  }
done:
  if (out) (void)hipFree(out);
  if (st) (void)hipFree(st);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  return rc;
}
