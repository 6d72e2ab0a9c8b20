// pack_inputs.h -- host side of the LFC input hand-over: binarizeAndPack.
//
// The reference binarises MNIST images on the HOST and ships 13 words per image to the accelerator
// (bnn/src/library/host/foldedmv-offload.cpp:82-98 binarizeAndPack; :186-188 the loop over the images;
// :194 the copy of count * psi words).  The entry points of this runtime that take HOST data
// (inference_multiple(path), bnn_mi355x_inference_buffer / _raw) do the same: 104 bytes per image cross the
// PCIe link instead of 784.  (Images that already lie in HBM -- bnn_mi355x_inference_device -- are binarised
// by the kernels.)
//
// Layout: 13 little-endian u64 words per image, bit i of the image = (pixel i >= 128) -- tiny-cnn's
// parse_mnist_images scales a pixel to -1 + 2 p / 255, binarizeAndPack sets the bit where that is >= 0 --,
// bits 784..831 zero (FOLDEDMV_INPUT_PADCHAR = 0).  Exactly what k_lfc_binarize writes on the device.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace bnn {

constexpr int kLfcPixels = 784, kLfcWords = 13;  // paddedSize(784, 64) / 64

// n images of 784 bytes -> n x 13 words.  `words` need not be aligned; plain stores.
// AVX2 (vpmovmskb: 32 pixels per instruction) where the CPU has it, a portable multiply-gather otherwise.
void binarize_pack(const uint8_t *pixels, size_t n, uint64_t *words);
// the portable form alone (tests compare the two)
void binarize_pack_portable(const uint8_t *pixels, size_t n, uint64_t *words);
// which form binarize_pack uses on this CPU: "avx2" or "portable"
const char *binarize_pack_isa();

}  // namespace bnn
