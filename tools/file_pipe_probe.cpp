// tools/file_pipe_probe.cpp -- the transfer pipe of the file entry point on its own (no stages): reader threads pread() a file
// from the page cache into a ring of pinned pieces, the main thread sends every full piece to HBM with an asynchronous copy.
// What rate does the pipe sustain, and what changes it?   hipcc -O2 -o file_pipe_probe file_pipe_probe.cpp -lpthread
//   file_pipe_probe FILE [threads=6] [piece_MB=4] [slots=12] [mode]     mode: 0 plain, 1 clflushopt every line behind the pread,
//   2 pread into a 256 KB cached bounce buffer + non-temporal stores into the ring, 3 = mode 0 with the DMAs alternating
//   over two streams, 4 = pread only (no DMA), 5 = DMA only (ring prefilled once)
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <immintrin.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
  if (argc < 2) return 1;
  const int nt = argc > 2 ? atoi(argv[2]) : 6, slots = argc > 4 ? atoi(argv[4]) : 12, mode = argc > 5 ? atoi(argv[5]) : 0;
  const size_t piece = (size_t)(argc > 3 ? atoi(argv[3]) : 4) << 20;
  const int fd = open(argv[1], O_RDONLY);
  struct stat st;
  if (fd < 0 || fstat(fd, &st)) return 2;
  const size_t total = (size_t)st.st_size, np = (total + piece - 1) / piece;
  uint8_t *ring, *dev;
  if (hipHostMalloc((void **)&ring, slots * piece, hipHostMallocDefault) != hipSuccess || hipMalloc((void **)&dev, total) != hipSuccess) return 3;
  memset(ring, 1, slots * piece);
  hipStream_t s[2];
  hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking);
  std::vector<hipEvent_t> ev(slots);
  for (auto &e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  std::vector<std::atomic<int>> filled(np);
  for (int rep = 0; rep < 5; rep++) {
    for (auto &f : filled) f = 0;
    std::atomic<size_t> next{0}, released{0};
    const double t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++)
      th.emplace_back([&] {
        static thread_local uint8_t *bounce = (uint8_t *)aligned_alloc(64, 256 << 10);
        for (;;) {
          const size_t p = next.fetch_add(1);
          if (p >= np) return;
          while (p >= released.load(std::memory_order_acquire) + slots) std::this_thread::yield();
          const size_t off = p * piece, b = off + piece <= total ? piece : total - off;
          uint8_t *dst = ring + (p % slots) * piece;
          if (mode == 5) {
          } else if (mode == 2) {
            for (size_t o = 0; o < b; o += 256 << 10) {
              const size_t c = b - o < (256u << 10) ? b - o : (256u << 10);
              if (pread(fd, bounce, c, off + o) != (ssize_t)c) abort();
              for (size_t i = 0; i + 64 <= c; i += 64) {
                const __m256i a = _mm256_load_si256((const __m256i *)(bounce + i)), bb = _mm256_load_si256((const __m256i *)(bounce + i + 32));
                _mm256_stream_si256((__m256i *)(dst + o + i), a);
                _mm256_stream_si256((__m256i *)(dst + o + i + 32), bb);
              }
            }
            _mm_sfence();
          } else {
            size_t done = 0;
            while (done < b) {
              const ssize_t g = pread(fd, dst + done, b - done, off + done);
              if (g <= 0) abort();
              done += g;
            }
            if (mode == 1) {
              for (size_t i = 0; i < b; i += 64) _mm_clflushopt(dst + i);
              _mm_sfence();
            }
          }
          filled[p].store(1, std::memory_order_release);
        }
      });
    size_t issued = 0, rel = 0;
    auto release = [&] {
      while (rel < issued && hipEventQuery(ev[rel % slots]) == hipSuccess) rel++;
      released.store(rel, std::memory_order_release);
    };
    for (size_t p = 0; p < np; p++) {
      while (!filled[p].load(std::memory_order_acquire)) { release(); std::this_thread::yield(); }
      const size_t off = p * piece, b = off + piece <= total ? piece : total - off;
      if (mode == 4) {
        issued = p + 1; rel = issued; released.store(rel);
        continue;
      }
      hipStream_t ss = s[mode == 3 ? (p & 1) : 0];
      hipMemcpyAsync(dev + off, ring + (p % slots) * piece, b, hipMemcpyHostToDevice, ss);
      hipEventRecord(ev[p % slots], ss);
      issued = p + 1;
      release();
    }
    hipStreamSynchronize(s[0]);
    hipStreamSynchronize(s[1]);
    const double dt = now() - t0;
    for (auto &t : th) t.join();
    if (rep) printf("threads %2d piece %2zu MB slots %2d mode %d: %.2f ms = %.1f GB/s\n", nt, piece >> 20, slots, mode, dt * 1e3, total / dt / 1e9);
  }
  return 0;
}
