// preprocess.hip -- picture -> CIFAR-10 record on the GPU.
//
// Replaces the host-side PIL work of CnvClassifier.image_to_cifar (bnn/bnn.py:226-242):
//   img.thumbnail((32, 32), ANTIALIAS); paste centred on an RGBA (255,255,255,0) canvas;
//   write label byte 1 and the R, G, B planes.
// Integer arithmetic throughout (Pillow's 8-bit path: 22-bit fixed-point coefficients, the
// horizontal pass rounded to 8 bits before the vertical one), so the record is bit-identical to
// the one PIL produces.  Both passes are reductions over up to several hundred taps per output
// sample; the source row is read with the lanes of a wave on CONSECUTIVE bytes (coalesced), each
// lane accumulates the band its bytes belong to, and a wave reduction (DPP adds) finishes the sum:
// sums wrap mod 2^32 exactly like Pillow's int accumulators, so the order of additions is free.
#include "preprocess.h"

namespace bnn {
namespace {

constexpr int kPrec = 22;

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ uint8_t clip8(uint32_t ss) {
  const int v = (int)ss >> kPrec;
  return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
}

// Horizontal pass: one wave per (row y, output column xx); 4 waves per block.
//   dst[y][xx][b] = clip8(2^21 + sum_x src[y][xmin + x][b] * k[xx][x])
template <int BANDS>
__global__ __launch_bounds__(256) void k_resample_h(const uint8_t *__restrict__ src, long stride, int h, uint8_t *__restrict__ dst,
                                                     int out_w, const int32_t *__restrict__ kk, const int32_t *__restrict__ bounds,
                                                     int ksize) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wave >= h * out_w) return;
  const int y = wave / out_w, xx = wave - y * out_w;
  const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
  const uint8_t *__restrict__ row = src + (size_t)y * stride + (size_t)xmin * BANDS;
  const int32_t *__restrict__ k = kk + (size_t)xx * ksize;
  uint32_t acc[BANDS];
#pragma unroll
  for (int b = 0; b < BANDS; b++) acc[b] = 0;
  if constexpr (BANDS == 1) {
    for (int j = lane; j < xmax; j += 64) acc[0] += (uint32_t)row[j] * (uint32_t)k[j];
  } else if constexpr (BANDS == 4) {
    // 64 = 0 mod 4: a lane stays on band (lane & 3); its sum goes to acc[0], the reduction below keeps bands apart
    const int nbytes = xmax * 4;
    for (int j = lane; j < nbytes; j += 64) acc[0] += (uint32_t)row[j] * (uint32_t)k[j >> 2];
  } else {
    // byte j of the window belongs to band j % 3; 192 = lcm(64, 3): a lane keeps its band
    // from one trip to the next only in steps of 192, so walk three interleaved strides
    const int nbytes = xmax * 3;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const int b = (lane + 64 * r) % 3;
      uint32_t a = 0;
      for (int j = lane + 64 * r; j < nbytes; j += 192) a += (uint32_t)row[j] * (uint32_t)k[j / 3];
      if (b == 0) acc[0] += a; else if (b == 1) acc[1] += a; else acc[2] += a;
    }
  }
  if constexpr (BANDS == 4) {
    uint32_t v = acc[0];
#pragma unroll
    for (int off = 32; off >= 4; off >>= 1) v += __shfl_xor(v, off, 64);  // lanes 0..3 end up with bands 0..3
    if (lane < 4) dst[((size_t)y * out_w + xx) * 4 + lane] = clip8(v + (1u << (kPrec - 1)));
  } else {
#pragma unroll
    for (int b = 0; b < BANDS; b++) {
      const uint32_t s = wave_sum(acc[b]) + (1u << (kPrec - 1));
      if (lane == 0) dst[((size_t)y * out_w + xx) * BANDS + b] = clip8(s);
    }
  }
}

// RGBA pictures are resampled with premultiplied alpha (Image.resize converts to "RGBa" and back): Pillow's
// rgbA2rgba, c' = MULDIV255(c, a) = ((t = c * a + 128), ((t >> 8) + t) >> 8), alpha itself unchanged
__global__ __launch_bounds__(256) void k_premultiply(uint8_t *__restrict__ px, size_t n_pixels) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pixels) return;
  uint32_t v = reinterpret_cast<uint32_t *>(px)[i];
  const uint32_t a = v >> 24;
  uint32_t out = v & 0xFF000000u;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const uint32_t t = ((v >> (8 * c)) & 0xFF) * a + 128;
    out |= ((((t >> 8) + t) >> 8) & 0xFF) << (8 * c);
  }
  reinterpret_cast<uint32_t *>(px)[i] = out;
}

// Vertical pass + paste: one 1024-thread block per canvas row.  A canvas row holds `out_w x BANDS`
// resampled samples (<= 96); the block's 16 waves split the taps of every sample, partial sums meet
// in LDS.  `src` is the horizontal pass's output (row pitch = out_w * BANDS) or, when no horizontal
// pass ran, the picture itself (row pitch = stride).
template <int BANDS>
__global__ __launch_bounds__(1024) void k_resample_v_paste(const uint8_t *__restrict__ src, long pitch, int out_w, int out_h,
                                                            const int32_t *__restrict__ kk, const int32_t *__restrict__ bounds, int ksize,
                                                            int vertical, int premultiplied, uint8_t *__restrict__ record) {
  __shared__ uint32_t part[16][128];
  __shared__ uint8_t line[128];
  const int cy = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int off_x = (32 - out_w) / 2, off_y = (32 - out_h) / 2;  // int((32 - size) / 2), size <= 32
  const int yy = cy - off_y;
  const bool inside = yy >= 0 && yy < out_h;  // block-uniform
  const int cols = out_w * BANDS;
  if (inside) {
    if (vertical) {
      const int ymin = bounds[2 * yy], ymax = bounds[2 * yy + 1];
      const int32_t *__restrict__ k = kk + (size_t)yy * ksize;
      for (int c = lane; c < cols; c += 64) {
        uint32_t a = 0;
        for (int y = wave; y < ymax; y += 16) a += (uint32_t)src[(size_t)(ymin + y) * pitch + c] * (uint32_t)k[y];
        part[wave][c] = a;
      }
      __syncthreads();
      if (t < cols) {
        uint32_t s = 1u << (kPrec - 1);
#pragma unroll
        for (int w = 0; w < 16; w++) s += part[w][t];
        line[t] = clip8(s);
      }
    } else if (t < cols) {
      line[t] = src[(size_t)yy * pitch + t];
    }
  }
  __syncthreads();
  if (cy == 0 && t == 0) record[0] = 1;  // the label byte the reference writes (np.identity(1))
  if (t < 96) {
    const int ch = t >> 5, cx = t & 31, xx = cx - off_x;
    uint8_t v = 255;  // the canvas: (255, 255, 255, 0), alpha is not part of the record
    if (inside && xx >= 0 && xx < out_w) {
      v = line[xx * BANDS + (BANDS >= 3 ? ch : 0)];
      if constexpr (BANDS == 4) {
        // back from premultiplied alpha (Pillow's rgba2rgbA): copied when alpha is 0 or 255, else 255 c / a clipped
        const uint32_t a = line[xx * 4 + 3];
        if (premultiplied && a != 255 && a != 0) {
          const uint32_t q = (255u * v) / a;
          v = (uint8_t)(q > 255 ? 255 : q);
        }
      }
    }
    record[1 + ch * 1024 + cy * 32 + cx] = v;
  }
}

// Vertical pass alone (the first pass of very tall pictures): one thread per output sample.
//   dst[yy][c] = clip8(2^21 + sum_y src[ymin + y][c] * k[yy][y]),  c over the w x bands bytes of a row
__global__ __launch_bounds__(256) void k_resample_v(const uint8_t *__restrict__ src, long pitch, int cols, uint8_t *__restrict__ dst,
                                                     const int32_t *__restrict__ kk, const int32_t *__restrict__ bounds, int ksize) {
  const int c = blockIdx.x * 256 + threadIdx.x, yy = blockIdx.y;
  if (c >= cols) return;
  const int ymin = bounds[2 * yy], ymax = bounds[2 * yy + 1];
  const int32_t *__restrict__ k = kk + (size_t)yy * ksize;
  uint32_t s = 1u << (kPrec - 1);
  for (int y = 0; y < ymax; y++) s += (uint32_t)src[(size_t)(ymin + y) * pitch + c] * (uint32_t)k[y];
  dst[(size_t)yy * cols + c] = clip8(s);
}

// parse_cifar10 on the device: record i = [label][3072 bytes] at byte 3073 i.  One lane per output
// dword; the source is unaligned by (3073 i + 1) % 4, so two aligned dwords and a v_alignbyte.
__global__ __launch_bounds__(256) void k_strip_records(const uint32_t *__restrict__ raw, int rec_bytes, int skip, uint32_t *__restrict__ out,
                                                        int img_dwords, size_t n_dwords) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_dwords) return;
  const size_t rec = t / img_dwords;
  const int j = (int)(t - rec * img_dwords);
  const size_t a = rec * (size_t)rec_bytes + skip + 4 * (size_t)j;  // byte address of the source dword
  const uint32_t lo = raw[a >> 2], hi = raw[(a >> 2) + 1];          // (the buffer has 4 bytes of slack)
  out[t] = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(a & 3));
}

}  // namespace

hipError_t launch_strip_records(const uint8_t *raw, int rec_bytes, int skip, uint8_t *out, int n_records, hipStream_t s) {
  const int img_dwords = (rec_bytes - skip) / 4;
  const size_t n = (size_t)n_records * img_dwords;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_strip_records, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const uint32_t *>(raw), rec_bytes,
                     skip, reinterpret_cast<uint32_t *>(out), img_dwords, n);
  return hipGetLastError();
}

hipError_t launch_image_to_cifar(const ResampleJob &j, hipStream_t s) {
  const bool horizontal = j.out_w != j.w;
  bool vertical = j.out_h != j.h;
  const int premultiplied = (j.bands == 4 && (horizontal || vertical)) ? 1 : 0;
  const uint8_t *in = j.src;
  long pitch = j.stride;
  int rows = j.h;
  uint8_t *free_tmp = j.tmp;
  if (premultiplied) {  // in place: j.src is the runtime's own copy of the picture, rows packed
    const size_t n = (size_t)j.w * j.h;
    hipLaunchKernelGGL(k_premultiply, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, const_cast<uint8_t *>(j.src), n);
  }
  if (vertical && horizontal && j.vertical_first) {
    const int cols = j.w * j.bands;
    hipLaunchKernelGGL(k_resample_v, dim3((unsigned)((cols + 255) / 256), (unsigned)j.out_h), dim3(256), 0, s, in, pitch, cols, j.tmp,
                       j.kv, j.bv, j.ksize_v);
    in = j.tmp;
    pitch = cols;
    rows = j.out_h;
    free_tmp = j.tmp + (size_t)j.out_h * cols;
    vertical = false;
  }
  if (horizontal) {
    // Pillow runs this pass over the rows the vertical pass will read; running it over all rows
    // gives the same samples
    const int waves = rows * j.out_w;
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    if (j.bands == 3) hipLaunchKernelGGL(k_resample_h<3>, grid, block, 0, s, in, pitch, rows, free_tmp, j.out_w, j.kh, j.bh, j.ksize_h);
    else if (j.bands == 4) hipLaunchKernelGGL(k_resample_h<4>, grid, block, 0, s, in, pitch, rows, free_tmp, j.out_w, j.kh, j.bh, j.ksize_h);
    else hipLaunchKernelGGL(k_resample_h<1>, grid, block, 0, s, in, pitch, rows, free_tmp, j.out_w, j.kh, j.bh, j.ksize_h);
    in = free_tmp;
    pitch = (long)j.out_w * j.bands;
  }
  const int v = vertical ? 1 : 0;
  if (j.bands == 3)
    hipLaunchKernelGGL(k_resample_v_paste<3>, dim3(32), dim3(1024), 0, s, in, pitch, j.out_w, j.out_h, j.kv, j.bv, j.ksize_v, v, 0, j.record);
  else if (j.bands == 4)
    hipLaunchKernelGGL(k_resample_v_paste<4>, dim3(32), dim3(1024), 0, s, in, pitch, j.out_w, j.out_h, j.kv, j.bv, j.ksize_v, v, premultiplied, j.record);
  else
    hipLaunchKernelGGL(k_resample_v_paste<1>, dim3(32), dim3(1024), 0, s, in, pitch, j.out_w, j.out_h, j.kv, j.bv, j.ksize_v, v, 0, j.record);
  return hipGetLastError();
}

}  // namespace bnn
