// faults.h -- random bit/word-flip injection into the parameter memories.
//
// Mirrors the reference's fault campaign machinery (bnn/src/library/host/faults.h:30-148,
// topology.h:29-178, foldedmv-offload.h:146-214): faults land at uniformly distributed
// positions of the weight / threshold memories (selection weighted by the bit count of each
// layer) and at uniformly distributed image indices; a fault flips `word_size` adjacent bits
// of one memory word through the FoldedMVMemRead / FoldedMVMemSet path and stays in place.
// The reference seeds every generator from std::random_device; here the seed can be fixed
// (bnn_mi355x_set_fault_seed) so that a campaign can be replayed and checked.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "packed_params.h"
#include "topology.h"

namespace bnn {

struct Fault {
  int image;      // injected before this image is classified
  int target;     // 0 weights, 1 thresholds ("activations" in the reference's naming)
  int layer, mem, ind, thresh, bit, word_size;
};

// target_type: < 0 any, 0 weights, > 0 thresholds (main_python.cpp:105-108).  Returns "" and the plan, or the
// reason why the request cannot be honoured as asked (a target layer outside the network; thresholds-only
// faults on layers without threshold memory): never a silently different campaign.
std::string plan_faults(const NetSpec &net, uint64_t seed, int num_images, unsigned flip_count, int word_size,
                        int target_type, const int *target_layers, unsigned num_layers, std::vector<Fault> &out);

// inject_fault_impl on the raw memories; returns the matrix row whose packed form changed, or -1
int apply_fault(const NetSpec &net, RawParams &raw, const Fault &f);

}  // namespace bnn
