// pack_inputs.cpp -- binarizeAndPack on the host (see pack_inputs.h).
#include "pack_inputs.h"

#include <cstring>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace bnn {
namespace {

// the most significant bits of 8 consecutive bytes, gathered into one byte (byte 0 -> bit 0)
inline uint64_t msb8(const uint8_t *p) {
  uint64_t v;
  std::memcpy(&v, p, 8);
  // (x >> 7) & 0x01..01 leaves byte k's bit at position 8k; the multiplier has a 1 at 7j for j = 1..8, so the product
  // has byte k's bit at 8k + 7j: all 64 positions distinct (no carries), and position 56 + m is reached by k = m, j = 8 - m only
  return (((v >> 7) & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56;
}

}  // namespace

void binarize_pack_portable(const uint8_t *pixels, size_t n, uint64_t *words) {
  for (size_t i = 0; i < n; i++) {
    const uint8_t *p = pixels + i * kLfcPixels;
    uint64_t w[kLfcWords];
    for (int k = 0; k < 12; k++) {
      uint64_t x = 0;
      for (int b = 0; b < 8; b++) x |= msb8(p + 64 * k + 8 * b) << (8 * b);
      w[k] = x;
    }
    w[12] = msb8(p + 768) | (msb8(p + 776) << 8);  // pixels 768..783; bits 784..831 are padding (0)
    std::memcpy(words + i * kLfcWords, w, sizeof(w));
  }
}

#if defined(__x86_64__)
namespace {
__attribute__((target("avx2"))) void binarize_pack_avx2(const uint8_t *pixels, size_t n, uint64_t *words) {
  for (size_t i = 0; i < n; i++) {
    const uint8_t *p = pixels + i * kLfcPixels;
    uint64_t w[kLfcWords];
    for (int k = 0; k < 12; k++) {
      const uint32_t lo = (uint32_t)_mm256_movemask_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + 64 * k)));
      const uint32_t hi = (uint32_t)_mm256_movemask_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + 64 * k + 32)));
      w[k] = (uint64_t)lo | ((uint64_t)hi << 32);
    }
    w[12] = (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(p + 768)));
    std::memcpy(words + i * kLfcWords, w, sizeof(w));
  }
}
bool have_avx2() {
  static const bool v = __builtin_cpu_supports("avx2");
  return v;
}
}  // namespace
#endif

void binarize_pack(const uint8_t *pixels, size_t n, uint64_t *words) {
#if defined(__x86_64__)
  if (have_avx2()) return binarize_pack_avx2(pixels, n, words);
#endif
  binarize_pack_portable(pixels, n, words);
}

const char *binarize_pack_isa() {
#if defined(__x86_64__)
  if (have_avx2()) return "avx2";
#endif
  return "portable";
}

}  // namespace bnn
