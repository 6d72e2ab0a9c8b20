// tools/launch_latency_probe.hip -- what a host pays around ONE short kernel, by the way it launches and waits:
//   A  hipExtLaunchKernelGGL with start/stop events, poll hipEventQuery(stop)          (the direct calls' way, round 4)
//   B  hipExtLaunchKernelGGL with events, hipStreamSynchronize                          (round 3)
//   C  plain launch, the kernel's last act is a store to pinned host memory, the host spins on it (no HIP call in the wait)
//   D  as C, then hipEventQuery polled until the stop event is ready as well (what timing by events would still cost)
// The kernel spins on the 100 MHz clock for SPIN_US microseconds (default 5), one block of 1024 threads.
//   hipcc -O2 --offload-arch=gfx950 tools/launch_latency_probe.hip -o tools/launch_latency_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(1024) void k_spin(volatile unsigned *flag, unsigned seq, int ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) {}
  __syncthreads();
  if (threadIdx.x == 0 && flag) {
    __threadfence_system();
    *flag = seq;
  }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
  const int spin_us = argc > 1 ? std::atoi(argv[1]) : 5, reps = 200;
  hipStream_t s;
  (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  unsigned *h = nullptr, *d = nullptr;
  (void)hipHostMalloc(reinterpret_cast<void **>(&h), 64, hipHostMallocMapped);
  (void)hipHostGetDevicePointer(reinterpret_cast<void **>(&d), h, 0);
  *h = 0;
  unsigned seq = 0;
  for (int mode = 0; mode < 4; mode++) {
    std::vector<double> total, launch, dev;
    for (int r = 0; r < reps + 20; r++) {
      seq++;
      const double t0 = now_us();
      if (mode <= 1 || mode == 3) hipExtLaunchKernelGGL(k_spin, dim3(1), dim3(1024), 0, s, e0, e1, 0, mode == 3 ? d : nullptr, seq, spin_us * 100);
      else hipLaunchKernelGGL(k_spin, dim3(1), dim3(1024), 0, s, d, seq, spin_us * 100);
      const double t1 = now_us();
      if (mode == 0) { while (hipEventQuery(e1) == hipErrorNotReady) {} }
      else if (mode == 1) (void)hipStreamSynchronize(s);
      else { while (*(volatile unsigned *)h != seq) {} }
      const double t2 = now_us();
      double t3 = t2;
      if (mode == 3) { while (hipEventQuery(e1) == hipErrorNotReady) {} t3 = now_us(); }
      float ms = 0.f;
      if (mode != 2) (void)hipEventElapsedTime(&ms, e0, e1);
      else (void)hipStreamSynchronize(s);
      if (r >= 20) { total.push_back((mode == 3 ? t3 : t2) - t0); launch.push_back(t1 - t0); dev.push_back(mode == 3 ? t2 - t0 : ms * 1e3); }
    }
    std::sort(total.begin(), total.end()); std::sort(launch.begin(), launch.end()); std::sort(dev.begin(), dev.end());
    const char *names[4] = {"A ext launch + events, poll hipEventQuery", "B ext launch + events, hipStreamSynchronize", "C plain launch, spin on a flag in pinned memory",
                            "D ext launch + events, flag seen (3rd figure), then poll the stop event"};
    std::printf("%-72s launch call %5.1f us | launch -> done %5.1f us (min %5.1f) | %s %5.1f us\n", names[mode], launch[reps / 2], total[reps / 2], total[0],
                mode == 3 ? "flag seen at" : (mode == 2 ? "-" : "device time by events"), dev[reps / 2]);
  }
  return 0;
}
