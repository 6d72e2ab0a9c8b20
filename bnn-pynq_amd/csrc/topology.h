// topology.h -- network tables of the five FINN overlays shipped with BNN-PYNQ,
// as the MI355X runtime sees them.
//
// Numbers come from the reference's per-network config.h
// (bnn/src/network/cnvW1A1/hw/config.h:18-202, cnvW2A2/hw/config.h,
// lfcW1A1/hw/config.h:17-83, lfcW1A2/hw/config.h) and the operand modes from
// the layer instantiations in top.cpp (cnvW1A1/hw/top.cpp:214-235,
// cnvW1A2/hw/top.cpp:213-233, cnvW2A2/hw/top.cpp:215-236,
// lfcW1A1/hw/top.cpp:155-164, lfcW1A2/hw/top.cpp:155-164).
//
// PE/SIMD/WMEM/TMEM ("folding") are FPGA resource knobs: here they only
// describe how the bnn/params files are cut up.  What the GPU kernels need is
// the logical matrix (MH x MW), the arithmetic of the layer and the geometry.
#pragma once
#include <cstdint>

namespace bnn {

enum NetId { NET_CNVW1A1 = 0, NET_CNVW1A2 = 1, NET_CNVW2A2 = 2, NET_LFCW1A1 = 3, NET_LFCW1A2 = 4, NET_COUNT = 5 };

// how one layer multiplies and accumulates
enum Arith : uint32_t {
  AR_INT8 = 0,   // int8 activation x {-1,0,+1} weight, v_dot4 (first CNV layer)
  AR_XNOR = 1,   // 1-bit x 1-bit, acc = # mismatches; thresholds pre-transformed
  AR_TB = 2,     // {-1,0,1} activation x +-1 weight, two activation planes
  AR_TT = 3,     // {-1,0,1} x {-1,0,1}, two planes each
};

// how the GPU stage walks the feature map
enum Shape : uint32_t {
  SH_CONV0 = 0,    // 32x32x3 uint8 image -> 30x30 map
  SH_QUAD = 1,     // 3x3 conv, one work item = 2x2 output pixels (4x4 window); optional OR/max pool
  SH_SINGLE = 2,   // 3x3 conv, one work item = one output pixel
  SH_FC = 3,       // one work item = one image, KW words in, thresholded
  SH_FCLAST = 4,   // CNV layer 8: raw 16-bit accumulators + class decode
};

struct FileFold { int pe, simd, wmem, tmem; };

struct LayerSpec {
  Shape shape;
  Arith arith;
  int ifm_ch, ifm_dim, ofm_ch, ofm_dim;  // FC: ifm_ch = MW, ofm_ch = MH, dims 1
  bool pool;                             // 2x2 max-pool follows (StreamingMaxPool_Batch)
  int wbits;                             // Lx_WPI
  int nthr;                              // thresholds per neuron in the files (0: pass-through)
  int out_planes;                        // 1: 1-bit activations out, 2: {-1,0,1} out (sign, non-zero)
  bool thr24;                            // thresholds are ap_fixed<24,16> (2^-8 units)
  bool signed_bb;                        // AR_XNOR hardware, but the reference accumulates +-1 products (lfcW1A2 L0)
  FileFold fold;
  int mw() const { return (fold.wmem / fold.tmem) * fold.simd; }
  int mh() const { return fold.tmem * fold.pe; }
};

struct NetSpec {
  NetId id;
  const char *name;
  bool is_cnv;
  int wbits, abits;
  int nlayers;
  LayerSpec L[9];
  int image_bytes() const { return is_cnv ? 3072 : 784; }
};

const NetSpec &net_spec(NetId id);
int net_from_name(const char *name);  // -1 if unknown

}  // namespace bnn
