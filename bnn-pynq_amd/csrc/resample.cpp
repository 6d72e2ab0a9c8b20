// see resample.h.  Compiled with -ffp-contract=off: the tables must round like Pillow's do.
#include "resample.h"

#include <cmath>

namespace bnn {
namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;  // Pillow: PRECISION_BITS for 8 bits per channel

double sinc(double x) {
  if (x == 0.0) return 1.0;
  x = x * M_PI;
  return std::sin(x) / x;
}

double lanczos3(double x) {  // truncated sinc, support 3
  if (-3.0 <= x && x < 3.0) return sinc(x) * sinc(x / 3);
  return 0.0;
}

}  // namespace

bool thumbnail_size(int w, int h, int box, int *out_w, int *out_h) {
  // Image.thumbnail -> preserve_aspect_ratio(): nothing to do when the box already holds the picture
  if (box >= w && box >= h) return false;
  double x = box, y = box;
  const double aspect = (double)w / (double)h;
  // round_aspect(number, key) = max(min(floor(number), ceil(number), key=key), 1): of the two
  // integer neighbours the one whose aspect ratio is nearer (floor on a tie)
  if (x / y >= aspect) {
    const double number = y * aspect;
    const double lo = std::floor(number), hi = std::ceil(number);
    const double klo = std::fabs(aspect - lo / y), khi = std::fabs(aspect - hi / y);
    x = (klo <= khi) ? lo : hi;
    if (x < 1) x = 1;
  } else {
    const double number = x / aspect;
    const double lo = std::floor(number), hi = std::ceil(number);
    const double klo = (lo == 0) ? 0.0 : std::fabs(aspect - x / lo), khi = (hi == 0) ? 0.0 : std::fabs(aspect - x / hi);
    y = (klo <= khi) ? lo : hi;
    if (y < 1) y = 1;
  }
  *out_w = (int)x;
  *out_h = (int)y;
  return true;
}

int lanczos_coeffs(int in_size, int out_size, std::vector<int32_t> &kk, std::vector<int32_t> &bounds) {
  const double in0 = 0.0, in1 = in_size;
  double filterscale, scale;
  filterscale = scale = (in1 - in0) / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 3.0 * filterscale;
  const int ksize = (int)std::ceil(support) * 2 + 1;
  kk.assign((size_t)out_size * ksize, 0);
  bounds.assign((size_t)out_size * 2, 0);
  std::vector<double> k(ksize);
  for (int xx = 0; xx < out_size; xx++) {
    const double center = in0 + (xx + 0.5) * scale;
    double ww = 0.0;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; x++) {
      const double w = lanczos3((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    int32_t *row = &kk[(size_t)xx * ksize];
    for (int x = 0; x < xmax; x++) {
      double v = k[x];
      if (ww != 0.0) v /= ww;
      row[x] = (v < 0) ? (int32_t)(-0.5 + v * (1 << kPrecisionBits)) : (int32_t)(0.5 + v * (1 << kPrecisionBits));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
  return ksize;
}

}  // namespace bnn
