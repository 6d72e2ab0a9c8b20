// packed_params.h -- the device-resident parameter blob.
//
// load_parameters() reads the reference's bnn/params/<dataset>/<net>/
// L-P-weights.bin / L-P-thres.bin files unchanged (format: SURVEY.md A1;
// reader being replaced: FoldedMVLoadLayerMem, bnn/src/library/host/
// foldedmv-offload.cpp:281-336) and repacks them ONCE into this blob: one
// contiguous, position-independent byte string (offsets, no pointers) that is
// uploaded to HBM, and that rank 0 can broadcast to the other GPUs over RCCL
// as plain bytes.
//
// Layout of a layer = `rows` neuron rows of `row_dwords` 32-bit words, neuron
// n at row n (the PE interleave of the files is undone).  A row is what ONE
// wave needs to evaluate one neuron for 64 work items, laid out so that it is
// fetched with a couple of wide scalar loads (s_load_dwordx8/x16) and then
// used straight from SGPRs as the scalar operand of v_xor / v_and / v_dot4:
//
//   dword 0,1      t0, t1   pre-transformed thresholds (see below)
//   dword 2..      the weight words of the row
//
//   AR_INT8   7 dwords of int8 taps in {-1,0,+1}: tap tau = 3*(c*3+ky)+kx (channel-plane, row,
//             then the 3 horizontally adjacent pixels) at byte tau%4 of dword tau/4, byte 27 = 0;
//             3 pad dwords.                              fire_i = t_i < dot
//             t_i = floor(T_i / 2)   (T in 2^-8 units, accumulator = 2*dot)
//   AR_XNOR   KW x u64, bit j = 1 <=> weight +1.  m = popcount(w ^ a) = # mismatches
//             fire_i = m < t_i,  t_i = MW - T_i           (T_i < MW - m)
//             signed form (lfcW1A2 L0): t_i = floor((MW - T_i + 1) / 2)   (T_i < MW - 2m)
//   AR_TB     KW x u64, bit j = 1 <=> weight -1.  acc = nz(a) - 2*popcount(za & (sa ^ w))
//             fire_i = T_i < acc
//   AR_TT     KW x {u64 sign (1 <=> -1), u64 non-zero}, then KW x u64 "weight is -2" (ap_int<2> 0b10: set by
//             bit flips only; such a column is also set in both planes), a flag dword (any -2 in the row), a pad.
//             acc = popcount(z) - 2*popcount(z & (sa ^ sw)),  z = za & zw;  fire_i = T_i < acc
//
// Column order j inside a row is the reference's: conv (ky*3+kx)*Cin + c, i.e.
// window pixel-major, channel-minor -- exactly the order in which the
// bit-packed HWC activation words of a 3x3 window are laid out.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "topology.h"

namespace bnn {

constexpr uint32_t kBlobMagic0 = 0x4D4E4E42u;  // "BNNM"
constexpr uint32_t kBlobMagic1 = 0x35353349u;  // "I355"
constexpr uint32_t kBlobVersion = 4;

struct PackedLayer {
  uint32_t offset;      // bytes from blob start, 256-byte aligned
  uint32_t row_dwords;
  uint32_t rows;        // MH
  uint32_t kw;          // 64-bit words per activation plane (AR_INT8: 0)
};

struct PackedHeader {
  uint32_t magic0, magic1, version, net_id;
  uint32_t nlayers, total_bytes;
  uint32_t l0_mfma_offset;  // CNV nets: layer 0 once more, as the A operand of v_mfma_i32_32x32x32_i8 (0: none)
  uint32_t reserved1;
  PackedLayer layer[9];
};

// Layer 0 for the matrix pipe: two tables of 64 rows of 32 int8 = {27 taps in the order of the AR_INT8
// rows, a0, a1, 0, 0, 0} with  a0 + 64*a1 = -t - 1  (t clamped to the reachable range of the dot
// product, +-3456): the first with t = t0, the second with t = t1 (second threshold of the 2-bit
// nets; a copy of the first for the 1-bit net).  The activation operand carries the constants 1 and
// 64 in K slots 27 and 28, so the MFMA result is  dot - t - 1 : its sign bit is !fire.
constexpr uint32_t kL0MfmaPixelBytes = 2 * 64 * 32;
// Behind them, the same layer in the operand form of k_conv0_tile (kernels.hip), which feeds the matrix pipe
// from quantised images staged in LDS: a 3-tap run (c, ky) of a window is ONE unaligned dword read -- its
// three taps plus a don't-care byte whose weight is 0 -- so K grows to 9 runs + 1 constant dword and is
// spread over a K = 32 and a K = 16 instruction (v_mfma_i32_32x32x32_i8 + v_mfma_i32_32x32x16_i8):
//   big   [tile ct 0..1][row i 0..31][h 0..1][16 int8]   k = 16h + 4s + b: tap kx = b (b < 3, else 0) of run
//         R[h][s], R[0] = (c,ky) (0,0) (0,1) (0,2) (2,0), R[1] = (1,0) (1,1) (1,2) (2,2)   (the two lane halves
//         then read LDS addresses a channel plane / two image rows apart: different banks)
//   small [threshold 0..1][ct][row i][h][8 int8]   h = 0: taps of run (2,1), 0...; h = 1: a0, a1, 0... (a0 +
//         64*a1 = -t - 1 as above, against the activation constants 1 and 64)
// Row i of tile ct is neuron 32ct + 16h' + 4g + q with i = 8g + 4h' + q: lane half h' of the MFMA result then
// holds, in register order, 16 CONSECUTIVE neurons -- their sign bits need no interleaving.
constexpr uint32_t kL0MfmaBigBytes = 2 * 32 * 2 * 16, kL0MfmaSmallBytes = 2 * 2 * 32 * 2 * 8;
constexpr uint32_t kL0MfmaTileOffset = kL0MfmaPixelBytes;  // from l0_mfma_offset
constexpr uint32_t kL0MfmaBytes = kL0MfmaPixelBytes + kL0MfmaBigBytes + kL0MfmaSmallBytes;
static_assert(sizeof(PackedHeader) == 32 + 9 * 16, "blob header layout");

uint32_t row_dwords_for(const LayerSpec &L);

// The reference's PE memories exactly as the files hold them (what DoMemInit / DoMemRead see,
// top.cpp:78-178): per layer, per PE, WMEM weight words and TMEM*nThr threshold words.  Kept on the
// host next to the blob so that single words can be modified (fault injection,
// FoldedMVMemRead/FoldedMVMemSet, foldedmv-offload.h:146-214) and the affected blob row rebuilt.
struct RawParams {
  std::vector<std::vector<uint64_t>> w[9], t[9];
  bool empty() const { return w[0].empty(); }
};

std::string read_raw_params(const NetSpec &net, const std::string &dir, RawParams &raw);
void pack_blob(const NetSpec &net, const RawParams &raw, std::vector<uint8_t> &blob);
// rebuild row n of layer l inside an existing blob; returns the byte offset and size of that row
void repack_row(const NetSpec &net, const RawParams &raw, int l, int n, std::vector<uint8_t> &blob, size_t *offset,
                size_t *bytes);

// Reads the param directory and fills `blob`.  Returns "" on success, else the
// error text (missing file: the reference throws "Could not open file",
// foldedmv-offload.cpp:321-323).  Short files are zero-filled like the
// reference's reader (foldedmv-offload.cpp:283-284).
std::string pack_params_from_dir(const NetSpec &net, const std::string &dir, std::vector<uint8_t> &blob);

// Number of rows (2-bit-weight layers only) holding a weight of -2, i.e. with their flag dword set.
int count_two_rows(const NetSpec &net, const std::vector<uint8_t> &blob);

// Size of this network's blob: a function of the topology alone (every rank of a multi-GPU job knows it
// without having seen the parameter files).
size_t blob_bytes(const NetSpec &net);

// Sanity-check a blob received from elsewhere (e.g. an RCCL broadcast).
std::string validate_blob(const NetSpec &net, const void *blob, size_t bytes);

}  // namespace bnn
