// kernels.h -- launch interface of kernels.hip (host side).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "topology.h"

namespace bnn {

struct CnvLaunch {
  const uint8_t *images;      // device, n x 3072 bytes, planar CHW uint8 (CIFAR-10 record bodies)
  int n;
  void *buf0, *buf1;          // device ping-pong activation buffers (cnv_workspace_bytes per image)
  const uint32_t *rows[9];    // device, per-layer packed rows (packed_params.h)
  const uint8_t *l0_mfma;     // device, layer-0 MFMA table (packed_params.h); null: integer-pipe k_conv0
  const uint8_t *l1_mfma;     // device, cnvW1A1 layer 1 as FP4 MFMA operands (l1_mfma_table); null: the XNOR-popcount kernel.
                              // Side experiment only (BNN_MI355X_L1=mfma): never the default path.
  bool l1_literal;            // cnvW1A1, BNN_MI355X_L1=lds: layer 1 in the north-star's literal formulation (comparison figure only)
  bool has_two;               // cnvW2A2: some row holds a weight of -2 (fault injection): the -2-aware kernel variants
  int16_t *scores;            // device, n x 64, may be null
  int32_t *classes;           // device, n, may be null
  int number_class;
  hipStream_t stream;
  hipEvent_t *events;         // optional: kCnvStages+1 events, recorded around every stage
  int last_stage;             // run stages 0..last_stage only (debug / per-layer tests); kCnvStages-1 = all
  hipEvent_t t0, t1;          // optional (both or neither): the batch's device time is t0 -> t1 (see LfcLaunch)
  // optional, n == 1 only (one image: the last launch is one block): a word in pinned host memory that the last kernel
  // sets to done_seq behind its results -- the host spins on it; no t0 / t1 packets are queued then
  unsigned *done_flag;
  unsigned done_seq;
};

struct LfcLaunch {
  const uint8_t *images;      // device, n x 784 bytes -- or, `packed`, n x 13 binarised words (csrc/pack_inputs.h), 8-byte aligned
  bool packed;
  int n;
  void *buf0, *buf1;
  const uint32_t *rows[4];
  uint64_t *words;            // device, n raw output words (required)
  int32_t *classes;           // device, n, may be null
  int number_class;
  hipStream_t stream;
  hipEvent_t *events;         // optional: kLfcStages+1 events
  int last_stage;             // run stages 0..last_stage only; kLfcStages-1 = all
  // optional (both or neither): the batch's device time is t0 -> t1.  Several launches: events recorded in front of
  // the first and behind the last.  ONE launch (k_lfc_fused*, k_lfc_block_s): the dispatch's own start / end
  // timestamps (hipExtLaunchKernelGGL) -- what a kernel trace reports for it -- instead of two more packets around
  // it, whose processing would be booked as compute (3.5 us on a 7 us kernel).
  hipEvent_t t0, t1;
  // t0 / t1 may be bound to the one dispatch (above).  Only for a call that is ONE chunk: hipEventElapsedTime between
  // such events of DIFFERENT dispatches is not an interval a multi-chunk call could place its chunks by; those get
  // ordinary recorded events around the launch.
  bool t_dispatch;
  unsigned *done_flag;        // as CnvLaunch::done_flag (n == 1: the one-block form of k_lfc_fused*)
  unsigned done_seq;
};

// layer-0 MFMA table (packed_params.h): the tile-form operands sit behind the pixel-form ones
constexpr int kL0TileOffset = 2 * 64 * 32, kL0BigBytes = 2 * 32 * 2 * 16;

constexpr int kCnvStages = 9;  // conv0, L1..L7, L8+decode
constexpr int kLfcStages = 6;  // binarize, L0..L3, decode
const char *stage_name(bool is_cnv, int stage);

size_t stage_output_bytes(bool is_cnv, int abits, int stage, int *in_buf1);
void cnv_workspace_bytes(int abits, size_t *buf0, size_t *buf1);
void lfc_workspace_bytes(int abits, size_t *buf0, size_t *buf1);

// cnvW1A1 layer 1 for the matrix pipe (side experiment): A operands of v_mfma_scale_f32_32x32x64_f8f6f4 with
// FP4 (E2M1) weights -- +1 = 0x2, -1 = 0xA -- [tap 0..8][neuron tile 0..1][lane 0..63][16 bytes], lane (r, h)
// holding channels 32h..32h+31 of neuron 32*tile + r; then the accumulator seeds [tile][h][16] floats
// -(theta + 1) with theta = 576 - 2 * t the threshold on the signed sum (the row's t is the XNOR form's
// "mismatches < t").  rows: layer 1 of the packed blob (host pointer).  Returns the table's bytes.
constexpr size_t kL1MfmaWeights = 9 * 2 * 64 * 16, kL1MfmaBytes = kL1MfmaWeights + 2 * 2 * 16 * 4;
void l1_mfma_table(const uint32_t *rows, uint8_t *dst);

// enqueue all stages of one batch on a.stream; returns the launch error, if any
hipError_t run_cnv(NetId net, const CnvLaunch &a);
hipError_t run_lfc(NetId net, const LfcLaunch &a);

}  // namespace bnn
