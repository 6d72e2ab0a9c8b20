// faults.cpp -- see faults.h (host only).
#include "faults.h"

#include <algorithm>
#include <random>

namespace bnn {
namespace {

// FINNTopology: bits of the weight / threshold memories of one layer (topology.h:57-75)
uint32_t weight_bits(const LayerSpec &L) { return (uint32_t)L.wbits * L.fold.simd * L.fold.pe * L.fold.wmem; }
uint32_t elem_bits(const LayerSpec &L) { return L.nthr == 0 ? 0u : (L.thr24 ? 24u : 16u); }
uint32_t thresh_bits(const LayerSpec &L) { return (uint32_t)L.fold.tmem * L.fold.pe * L.nthr * elem_bits(L); }

}  // namespace

std::string plan_faults(const NetSpec &net, uint64_t seed, int num_images, unsigned flip_count, int word_size,
                        int target_type, const int *target_layers, unsigned num_layers, std::vector<Fault> &out) {
  out.clear();
  if (num_images <= 0 || flip_count == 0) return "";
  std::mt19937_64 gen(seed ? seed : (uint64_t)std::random_device{}());
  // candidate layers: the caller's list, or every layer (the reference leaves the empty list to
  // std::discrete_distribution over two zero weights; "all layers" is the evident intent)
  std::vector<int> layers;
  for (unsigned i = 0; target_layers && i < num_layers; i++) {
    // (the reference indexes its topology tables with whatever it is given; a campaign that silently ran on
    // other layers than the ones asked for would be mislabelled, so this is an error here)
    if (target_layers[i] < 0 || target_layers[i] >= net.nlayers) return "fault injection: target layer out of range";
    layers.push_back(target_layers[i]);
  }
  if (layers.empty())
    for (int l = 0; l < net.nlayers; l++) layers.push_back(l);
  std::vector<double> wb, tb;
  double wsum = 0, tsum = 0;
  for (int l : layers) {
    wb.push_back(weight_bits(net.L[l])); wsum += wb.back();
    tb.push_back(thresh_bits(net.L[l])); tsum += tb.back();
  }
  if (target_type > 0 && tsum == 0) return "fault injection: no threshold memory in the targeted layers";
  // fault times: uniform over the image indices (faults.h:124-131)
  std::uniform_int_distribution<int> when(0, num_images - 1);
  std::vector<int> times(flip_count);
  for (auto &t : times) t = when(gen);
  std::stable_sort(times.begin(), times.end());
  for (int t : times) {
    Fault f{};
    f.image = t;
    f.word_size = word_size < 1 ? 1 : (word_size > 64 ? 64 : word_size);
    bool weights;
    if (target_type < 0) weights = std::discrete_distribution<int>({wsum, tsum})(gen) == 0;
    else weights = (target_type == 0);
    const std::vector<double> &space = weights ? wb : tb;
    std::discrete_distribution<int> pick(space.begin(), space.end());
    const int l = layers[pick(gen)];
    const LayerSpec &L = net.L[l];
    f.layer = l;
    if (weights) {
      // inject_fault, foldedmv-offload.h:189-199
      const uint32_t bit = std::uniform_int_distribution<uint32_t>(0, weight_bits(L) - 1)(gen);
      const uint32_t esz = (uint32_t)L.fold.simd * L.wbits, element = bit / esz;
      f.target = 0; f.thresh = 0;
      f.ind = (int)(element % L.fold.wmem);
      f.mem = (int)((element / L.fold.wmem) % L.fold.pe);
      f.bit = (int)(bit % esz);
    } else {
      // foldedmv-offload.h:200-210
      const uint32_t bit = std::uniform_int_distribution<uint32_t>(0, thresh_bits(L) - 1)(gen);
      const uint32_t esz = elem_bits(L), element = bit / esz;
      f.target = 1;
      f.thresh = (int)(element % L.nthr);
      f.ind = (int)((element / L.nthr) % L.fold.tmem);
      f.mem = (int)(((element / L.nthr) / L.fold.tmem) % L.fold.pe);
      f.bit = (int)(bit % esz);
    }
    out.push_back(f);
  }
  return "";
}

int apply_fault(const NetSpec &net, RawParams &raw, const Fault &f) {
  if (f.layer < 0 || f.layer >= net.nlayers || f.word_size < 1 || f.word_size > 64) return -1;
  const LayerSpec &L = net.L[f.layer];
  uint64_t flip = f.word_size >= 64 ? ~0ull : ((1ull << f.word_size) - 1);
  flip <<= (f.bit / f.word_size) * f.word_size;  // aligns bit_pos to a multiple of word_size
  if (f.target == 0) {
    if (f.mem >= L.fold.pe || f.ind >= L.fold.wmem) return -1;
    const int ebits = L.fold.simd * L.wbits;  // m_weights[pe][ind] is ap_uint<SIMD*WPI>
    const uint64_t emask = ebits >= 64 ? ~0ull : ((1ull << ebits) - 1);
    uint64_t &w = raw.w[f.layer][f.mem][f.ind];
    w = ((w & emask) ^ flip) & emask;
    return (f.ind / (L.fold.wmem / L.fold.tmem)) * L.fold.pe + f.mem;
  }
  if (L.nthr == 0 || f.mem >= L.fold.pe || f.ind >= L.fold.tmem || f.thresh >= L.nthr) return -1;
  uint64_t &t = raw.t[f.layer][f.mem][(size_t)f.ind * L.nthr + f.thresh];
  int64_t v;
  if (L.thr24) {
    // DoMemRead returns the INTEGER part of the ap_fixed<24,16> threshold (top.cpp:143); DoMemInit
    // reinterprets the 64-bit word as ap_fixed<64,56> (top.cpp:84): mirrored as is
    int32_t t24 = (int32_t)(t & 0xFFFFFF);
    if (t24 & 0x800000) t24 -= 0x1000000;
    v = (int64_t)(t24 >> 8);
  } else {
    v = (int64_t)(int16_t)(t & 0xFFFF);
  }
  t = (uint64_t)v ^ flip;
  return f.ind * L.fold.pe + f.mem;
}

}  // namespace bnn
