// topology.cpp -- builds the NetSpec tables (see topology.h for the sources).
#include "topology.h"

#include <cstring>

namespace bnn {
namespace {

struct ConvGeom { int ifm_ch, ifm_dim, ofm_ch, ofm_dim; Shape shape; bool pool; };
// cnv*/hw/config.h Lx_IFM_CH, Lx_IFM_DIM, Lx_OFM_CH, Lx_OFM_DIM
const ConvGeom kCnvConv[6] = {
    {3, 32, 64, 30, SH_CONV0, false},  {64, 30, 64, 28, SH_QUAD, true},
    {64, 14, 128, 12, SH_QUAD, false}, {128, 12, 128, 10, SH_QUAD, true},
    {128, 5, 256, 3, SH_SINGLE, false}, {256, 3, 256, 1, SH_FC, false}};
const int kCnvFc[3][2] = {{256, 512}, {512, 512}, {512, 64}};  // Lx_MW, Lx_MH

// Lx_PE, Lx_SIMD, Lx_WMEM, Lx_TMEM
const FileFold kCnvFoldW1[9] = {{16, 3, 36, 4},     {32, 32, 36, 2},    {16, 32, 144, 8},
                                {16, 32, 288, 8},   {4, 32, 2304, 64},  {1, 32, 18432, 256},
                                {1, 4, 32768, 512}, {1, 8, 32768, 512}, {4, 1, 8192, 16}};
const FileFold kCnvFoldW2[9] = {{8, 3, 72, 8},      {16, 16, 144, 4},   {8, 16, 576, 16},
                                {8, 16, 1152, 16},  {4, 8, 9216, 64},   {1, 8, 73728, 256},
                                {1, 2, 65536, 512}, {2, 2, 65536, 256}, {4, 1, 8192, 16}};
const int kLfcDims[4][2] = {{832, 1024}, {1024, 1024}, {1024, 1024}, {1024, 64}};
const FileFold kLfcFold[4] = {{32, 64, 416, 32}, {64, 32, 512, 16}, {32, 64, 512, 32}, {16, 8, 512, 4}};

NetSpec make_cnv(NetId id, const char *name, int wbits, int abits) {
  NetSpec n{};
  n.id = id; n.name = name; n.is_cnv = true; n.wbits = wbits; n.abits = abits; n.nlayers = 9;
  const FileFold *fold = (wbits == 2) ? kCnvFoldW2 : kCnvFoldW1;
  const Arith inner = (abits == 1) ? AR_XNOR : (wbits == 1 ? AR_TB : AR_TT);
  for (int l = 0; l < 9; l++) {
    LayerSpec &L = n.L[l];
    L.fold = fold[l];
    L.wbits = wbits;
    L.out_planes = abits;
    L.nthr = abits;
    L.arith = inner;
    if (l < 6) {
      const ConvGeom &g = kCnvConv[l];
      L.shape = g.shape; L.pool = g.pool;
      L.ifm_ch = g.ifm_ch; L.ifm_dim = g.ifm_dim; L.ofm_ch = g.ofm_ch; L.ofm_dim = g.ofm_dim;
      if (l == 5) { L.ifm_ch = 9 * 256; L.ifm_dim = 1; }  // 3x3x256 window == the whole map: an FC over 2304 bits
    } else {
      L.shape = (l == 8) ? SH_FCLAST : SH_FC;
      L.ifm_ch = kCnvFc[l - 6][0]; L.ofm_ch = kCnvFc[l - 6][1]; L.ifm_dim = L.ofm_dim = 1;
    }
    if (l == 0) { L.arith = AR_INT8; L.thr24 = true; }
    if (l == 8) { L.nthr = 0; L.out_planes = 0; }  // PassThroughActivation<ap_uint<16>>
  }
  return n;
}

NetSpec make_lfc(NetId id, const char *name, int abits) {
  NetSpec n{};
  n.id = id; n.name = name; n.is_cnv = false; n.wbits = 1; n.abits = abits; n.nlayers = 4;
  for (int l = 0; l < 4; l++) {
    LayerSpec &L = n.L[l];
    L.shape = SH_FC;
    L.fold = kLfcFold[l];
    L.wbits = 1;
    L.ifm_ch = kLfcDims[l][0]; L.ofm_ch = kLfcDims[l][1]; L.ifm_dim = L.ofm_dim = 1;
    if (abits == 1) {
      L.arith = AR_XNOR; L.nthr = 1; L.out_planes = 1;
    } else {
      // lfcW1A2: L0 Recast<Binary> x Recast<Binary> (signed sum of +-1 products),
      // L1..3 Slice<ap_int<2>> x Recast<Binary>; last layer one threshold, 1-bit out
      L.arith = (l == 0) ? AR_XNOR : AR_TB;
      L.signed_bb = (l == 0);
      L.nthr = (l == 3) ? 1 : 2;
      L.out_planes = (l == 3) ? 1 : 2;
    }
  }
  return n;
}

const NetSpec kNets[NET_COUNT] = {
    make_cnv(NET_CNVW1A1, "cnvW1A1", 1, 1), make_cnv(NET_CNVW1A2, "cnvW1A2", 1, 2),
    make_cnv(NET_CNVW2A2, "cnvW2A2", 2, 2), make_lfc(NET_LFCW1A1, "lfcW1A1", 1),
    make_lfc(NET_LFCW1A2, "lfcW1A2", 2)};

}  // namespace

const NetSpec &net_spec(NetId id) { return kNets[id]; }

int net_from_name(const char *name) {
  for (int i = 0; i < NET_COUNT; i++)
    if (!std::strcmp(kNets[i].name, name)) return i;
  return -1;
}

}  // namespace bnn
