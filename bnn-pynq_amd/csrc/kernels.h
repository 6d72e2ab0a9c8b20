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
  bool has_two;               // cnvW2A2: some row holds a weight of -2 (fault injection): the -2-aware kernel variants
  int16_t *scores;            // device, n x 64, may be null
  int32_t *classes;           // device, n, may be null
  int number_class;
  hipStream_t stream;
  hipEvent_t *events;         // optional: kCnvStages+1 events, recorded around every stage
  int last_stage;             // run stages 0..last_stage only (debug / per-layer tests); kCnvStages-1 = all
};

struct LfcLaunch {
  const uint8_t *images;      // device, n x 784 bytes
  int n;
  void *buf0, *buf1;
  const uint32_t *rows[4];
  uint64_t *words;            // device, n raw output words (required)
  int32_t *classes;           // device, n, may be null
  int number_class;
  hipStream_t stream;
  hipEvent_t *events;         // optional: kLfcStages+1 events
  int last_stage;             // run stages 0..last_stage only; kLfcStages-1 = all
};

constexpr int kCnvStages = 9;  // conv0, L1..L7, L8+decode
constexpr int kLfcStages = 6;  // binarize, L0..L3, decode
const char *stage_name(bool is_cnv, int stage);

size_t stage_output_bytes(bool is_cnv, int abits, int stage, int *in_buf1);
void cnv_workspace_bytes(int abits, size_t *buf0, size_t *buf1);
void lfc_workspace_bytes(int abits, size_t *buf0, size_t *buf1);

// enqueue all stages of one batch on a.stream; returns the launch error, if any
hipError_t run_cnv(NetId net, const CnvLaunch &a);
hipError_t run_lfc(NetId net, const LfcLaunch &a);

}  // namespace bnn
