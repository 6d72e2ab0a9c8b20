// Device half of image_to_cifar (see resample.h): the two fixed-point resampling passes and the
// paste onto the 32x32 canvas, written as the body of a CIFAR-10 record.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bnn {

struct ResampleJob {
  const uint8_t *src;  // H x W x bands, rows `stride` bytes apart (device)
  int w, h, bands;     // bands: 1 (L), 3 (RGB) or 4 (RGBA: resampled with premultiplied alpha, like Pillow; modified in place)
  long stride;
  int out_w, out_h;    // 1..32 each; == w / h: that pass is skipped, like Pillow does
  const int32_t *kh, *bh;  // horizontal coefficients [out_w][ksize_h], bounds [out_w][2] (device)
  int ksize_h;
  const int32_t *kv, *bv;  // vertical
  int ksize_v;
  bool vertical_first; // Pillow's order for very tall pictures (resample.h); needs out_w != w to matter
  uint8_t *tmp;        // max(h x out_w, out_h x w) x bands + 3072 bytes (device): output of the first pass
                       // (+ of the second one when the vertical pass runs first)
  uint8_t *record;     // 3073 bytes (device): label byte 1, then the R, G, B planes
};

hipError_t launch_image_to_cifar(const ResampleJob &job, hipStream_t s);

// Records of an input file as they lie on disk -> packed image bodies: drops the first `skip` bytes of
// every `rec_bytes`-byte record (the label byte of a CIFAR-10 record).  `raw` starts at a record
// boundary and is 4-byte aligned; img_bytes = rec_bytes - skip is a multiple of 4.
hipError_t launch_strip_records(const uint8_t *raw, int rec_bytes, int skip, uint8_t *out, int n_records, hipStream_t s);

}  // namespace bnn
