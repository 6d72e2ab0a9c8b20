// Host half of the device-side CnvClassifier.image_to_cifar (reference: bnn/bnn.py:226-242).
//
// The reference shrinks an arbitrary picture with PIL's Image.thumbnail((32, 32), ANTIALIAS) and
// pastes it centred on a white 32x32 canvas.  PIL (Pillow, a third-party dependency of the
// reference, not vendored in it) resamples 8-bit images in fixed point: two separable passes
// (horizontal, then vertical on the 8-bit result of the first), each output sample
//     clip8((2^21 + sum_x pixel[x] * k[x]) >> 22)
// with integer coefficients k = round(2^22 * normalised Lanczos-3 weight).  This file restates the
// two pieces that are double-precision host arithmetic -- the thumbnail size rule and the
// coefficient table -- and csrc/preprocess.hip does the integer passes on the GPU.  Pinned by
// tests/test_image_to_cifar.py against Pillow itself (bit-exact records).
#pragma once
#include <cstdint>
#include <vector>

namespace bnn {

// Image.thumbnail's aspect-preserving target for a (w, h) picture and a (32, 32) box.
// Returns false when PIL leaves the picture alone (it already fits).
bool thumbnail_size(int w, int h, int box, int *out_w, int *out_h);

// Image.resize of current Pillow (12.x) runs the vertical pass FIRST on very tall, thin pictures
// (height > 100 x width, and the height shrinks); the 8-bit rounding between the passes makes the
// order visible in the result.
inline bool vertical_pass_first(int w, int h, int out_h) { return (long)h > (long)w * 100 && out_h < h; }

// precompute_coeffs + normalize_coeffs_8bpc of Pillow's Resample.c for the Lanczos filter
// (support 3), box = the whole axis.  kk: out_size rows of `ksize` int32 (zero padded), bounds:
// out_size pairs {first input index, tap count}.  Returns ksize.
int lanczos_coeffs(int in_size, int out_size, std::vector<int32_t> &kk, std::vector<int32_t> &bounds);

}  // namespace bnn
