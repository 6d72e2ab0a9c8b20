/*
 * bnn_oracle.c -- CPU restatement of the BNN-PYNQ SW-runtime ("python_sw")
 * hot path: param load -> input quantise/binarise -> DoCompute -> decode.
 *
 * TEST INFRASTRUCTURE ONLY (see bnn_oracle.h).  Parity status: PINNED against
 * the reference's recorded outputs (tests/test_oracle_golden.py).
 *
 * Two implementations of the same function live here on purpose:
 *   *_ref   one multiply-accumulate at a time on unpacked integers, the same
 *           loop nest the HLS C-simulation runs (faithful, slow);
 *   *_fast  64-bit XNOR/AND + popcount words, OpenMP over images (used as the
 *           CPU baseline in bench.py and as the checker at batch sizes where
 *           *_ref would take minutes).  tests/ check *_fast == *_ref.
 *
 * Reference files followed (all under /root/reference/bnn/src):
 *   network/<net>/hw/config.h, top.cpp        topology, operand modes, types
 *   library/host/foldedmv-offload.{h,cpp}     load, pack, decode
 *   library/host/rawhls-offload.cpp           SW shim onto BlackBoxJam
 *   network/<net>/sw/main_python.cpp          which loader args / decode
 *   training/finnthesizer.py                  the on-disk param layout
 */
#include "bnn_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

enum { IN_INT8 = 0, IN_BIN = 1, IN_TER = 2 };

typedef struct {
  int is_conv;                 /* 3x3 valid stride-1 conv (ConvLayer_Batch) or FC */
  int ifm_ch, ifm_dim, ofm_ch, ofm_dim;
  int pe, simd, wmem, tmem;    /* folding: file layout only (config.h) */
  int wbits;                   /* Lx_WPI */
  int nthr;                    /* thresholds per neuron passed to FoldedMVLoadLayerMem; 0 = pass-through */
  int in_mode;                 /* operand type of the input stream */
  int xnor;                    /* Recast<XnorMul>: acc = popcount of matches */
  int pool;                    /* StreamingMaxPool_Batch<.,2,.> follows */
  int thr24;                   /* thresholds are ap_fixed<24,16> (CNV layer 0, top.cpp:61,84) */
  int mw, mh;                  /* derived */
} lcfg;

struct bnn_oracle {
  char name[16];
  int is_cnv;
  int nl;
  int abits;
  lcfg L[9];
  int8_t *W[9];     /* [mh][mw] value domain */
  int32_t *T[9];    /* [mh][2] */
  /* fast path: bit planes, [mh][kw] */
  uint64_t *Wp[9];  /* 1-bit: bit=1 <=> +1 ; 2-bit: sign plane (bit=1 <=> -1) */
  uint64_t *Wn[9];  /* 2-bit: non-zero plane */
  uint64_t *Wt[9];  /* 2-bit: columns whose weight is -2 (ap_int<2> 0b10: reachable by bit flips only) */
  /* the PE memories as the files hold them, [pe][wmem] / [pe][tmem*nthr] (DoMemInit's targets,
   * top.cpp:78-135): kept so that single words can be modified (fault injection) */
  uint64_t *wraw[9], *traw[9];
};

/* --------------------------------------------------------------------------
 * Topology tables: cnv*: network/cnvW1A1/hw/config.h:18-202 (W1A2 differs
 * only in Lx_API), network/cnvW2A2/hw/config.h; lfc*: network/lfcW1A1/hw/
 * config.h:17-83, network/lfcW1A2/hw/config.h.
 * -------------------------------------------------------------------------- */
static const int CNV_DIMS[6][4] = { /* ifm_ch, ifm_dim, ofm_ch, ofm_dim */
    {3, 32, 64, 30}, {64, 30, 64, 28}, {64, 14, 128, 12},
    {128, 12, 128, 10}, {128, 5, 256, 3}, {256, 3, 256, 1}};
static const int CNV_FC[3][2] = {{256, 512}, {512, 512}, {512, 64}}; /* MW, MH */
/* pe, simd, wmem, tmem */
static const int CNV_FOLD_W1[9][4] = {
    {16, 3, 36, 4},    {32, 32, 36, 2},    {16, 32, 144, 8},
    {16, 32, 288, 8},  {4, 32, 2304, 64},  {1, 32, 18432, 256},
    {1, 4, 32768, 512}, {1, 8, 32768, 512}, {4, 1, 8192, 16}};
static const int CNV_FOLD_W2[9][4] = {
    {8, 3, 72, 8},     {16, 16, 144, 4},   {8, 16, 576, 16},
    {8, 16, 1152, 16}, {4, 8, 9216, 64},   {1, 8, 73728, 256},
    {1, 2, 65536, 512}, {2, 2, 65536, 256}, {4, 1, 8192, 16}};
static const int LFC_DIMS[4][2] = {{832, 1024}, {1024, 1024}, {1024, 1024}, {1024, 64}};
static const int LFC_FOLD[4][4] = {
    {32, 64, 416, 32}, {64, 32, 512, 16}, {32, 64, 512, 32}, {16, 8, 512, 4}};

static int setup_net(bnn_oracle *o, const char *network) {
  int wbits, abits;
  memset(o, 0, sizeof(*o));
  snprintf(o->name, sizeof(o->name), "%s", network);
  if (!strcmp(network, "cnvW1A1")) { o->is_cnv = 1; wbits = 1; abits = 1; }
  else if (!strcmp(network, "cnvW1A2")) { o->is_cnv = 1; wbits = 1; abits = 2; }
  else if (!strcmp(network, "cnvW2A2")) { o->is_cnv = 1; wbits = 2; abits = 2; }
  else if (!strcmp(network, "lfcW1A1")) { o->is_cnv = 0; wbits = 1; abits = 1; }
  else if (!strcmp(network, "lfcW1A2")) { o->is_cnv = 0; wbits = 1; abits = 2; }
  else return -1;
  o->abits = abits;
  if (o->is_cnv) {
    const int (*fold)[4] = (wbits == 2) ? CNV_FOLD_W2 : CNV_FOLD_W1;
    o->nl = 9;
    for (int l = 0; l < 9; l++) {
      lcfg *L = &o->L[l];
      L->pe = fold[l][0]; L->simd = fold[l][1]; L->wmem = fold[l][2]; L->tmem = fold[l][3];
      L->wbits = wbits;
      if (l < 6) {
        L->is_conv = 1;
        L->ifm_ch = CNV_DIMS[l][0]; L->ifm_dim = CNV_DIMS[l][1];
        L->ofm_ch = CNV_DIMS[l][2]; L->ofm_dim = CNV_DIMS[l][3];
      } else {
        L->ifm_ch = CNV_FC[l - 6][0]; L->ifm_dim = 1;
        L->ofm_ch = CNV_FC[l - 6][1]; L->ofm_dim = 1;
      }
      /* top.cpp:214-235: L0 Slice<ap_fixed<8,1>>; W1A1 L1..8 Recast<XnorMul>;
       * A2 nets Slice<ap_int<2>> inputs. */
      L->in_mode = (l == 0) ? IN_INT8 : (abits == 1 ? IN_BIN : IN_TER);
      L->xnor = (l > 0 && wbits == 1 && abits == 1);
      L->pool = (l == 1 || l == 3);
      L->thr24 = (l == 0);
      /* main_python.cpp:73-81: Lx_API thresholds, layer 8 loads none. */
      L->nthr = (l == 8) ? 0 : abits;
    }
  } else {
    o->nl = 4;
    for (int l = 0; l < 4; l++) {
      lcfg *L = &o->L[l];
      L->pe = LFC_FOLD[l][0]; L->simd = LFC_FOLD[l][1];
      L->wmem = LFC_FOLD[l][2]; L->tmem = LFC_FOLD[l][3];
      L->wbits = 1;
      L->ifm_ch = LFC_DIMS[l][0]; L->ifm_dim = 1;
      L->ofm_ch = LFC_DIMS[l][1]; L->ofm_dim = 1;
      /* lfcW1A1/hw/top.cpp:156-163 all Recast<XnorMul>; lfcW1A2/hw/top.cpp:
       * 156-163 L0 Recast<Binary> x Recast<Binary>, L1..3 Slice<ap_int<2>>. */
      L->in_mode = (l == 0 || abits == 1) ? IN_BIN : IN_TER;
      L->xnor = (abits == 1);
      L->nthr = (abits == 2 && l < 3) ? 2 : 1;
    }
  }
  for (int l = 0; l < o->nl; l++) {
    lcfg *L = &o->L[l];
    L->mh = L->tmem * L->pe;
    L->mw = (L->wmem / L->tmem) * L->simd;
  }
  return 0;
}

/* --------------------------------------------------------------------------
 * A1. Param loader.  Reader: FoldedMVLoadLayerMem / WeightPE / ThreshPE
 * (foldedmv-offload.cpp:281-336): per PE one file of WMEM (TMEM*nThr)
 * little-endian 64-bit words, a short read leaves the word 0.  Sink: DoMemInit
 * (top.cpp:78-135).  Writer: finnthesizer.py:557-588 (row n -> PE n%PE, slot
 * n/PE), :461-463 + :616-637 (column s of a SIMD group at bit s*WPI).
 * -------------------------------------------------------------------------- */
static int read_words(const char *path, uint64_t *dst, size_t n) {
  FILE *f = fopen(path, "rb");
  if (!f) return -1;
  memset(dst, 0, n * sizeof(uint64_t));
  for (size_t i = 0; i < n; i++) {
    unsigned char b[8];
    memset(b, 0, 8);
    size_t got = fread(b, 1, 8, f);
    /* `ExtMemWord e = 0; wf.read((char*)&e, 8)` (foldedmv-offload.cpp:283-284):
     * a short read stores the bytes it got and sets failbit, every later word
     * stays 0 */
    uint64_t e = 0;
    for (int k = 7; k >= 0; k--) e = (e << 8) | b[k];
    dst[i] = e;
    if (got != 8) break;
  }
  fclose(f);
  return 0;
}

/* decode row n of layer l from the raw PE memories into W / T / bit planes */
static void derive_row(bnn_oracle *o, int l, int n) {
  lcfg *L = &o->L[l];
  const int SF = L->wmem / L->tmem, kw = (L->mw + 63) / 64;
  const int p = n % L->pe, nf = n / L->pe;
  const uint64_t *wf = o->wraw[l] + (size_t)p * L->wmem;
  for (int sf = 0; sf < SF; sf++) {
    const uint64_t word = wf[nf * SF + sf];
    for (int s = 0; s < L->simd; s++) {
      int v;
      if (L->wbits == 1) {
        /* BinaryWeights + Recast<Binary>/XnorMul: bit 1 <=> +1 */
        v = ((word >> s) & 1) ? 1 : -1;
      } else {
        /* FixedPointWeights<.., ap_int<2>, ..> (cnvW2A2/hw/top.cpp:52-60) */
        const int f = (int)((word >> (2 * s)) & 3);
        v = (f >= 2) ? f - 4 : f;
      }
      o->W[l][(size_t)n * L->mw + sf * L->simd + s] = (int8_t)v;
    }
  }
  for (int i = 0; i < L->nthr; i++) {
    const uint64_t e = o->traw[l][(size_t)p * L->tmem * L->nthr + nf * L->nthr + i];
    int32_t t;
    if (L->thr24) {
      /* top.cpp:84: reinterpret as ap_fixed<64,56>, assign to
       * ap_fixed<24,16> (AP_TRN, AP_WRAP): low 24 bits, units of 2^-8 */
      t = (int32_t)(e & 0xFFFFFF);
      if (t & 0x800000) t -= 0x1000000;
    } else {
      /* top.cpp:90: ap_uint<64> -> ap_int<16>: low 16 bits */
      t = (int16_t)(e & 0xFFFF);
    }
    o->T[l][n * 2 + i] = t;
  }
  for (int k = 0; k < kw; k++) o->Wp[l][(size_t)n * kw + k] = o->Wn[l][(size_t)n * kw + k] = o->Wt[l][(size_t)n * kw + k] = 0;
  for (int j = 0; j < L->mw; j++) {
    const int v = o->W[l][(size_t)n * L->mw + j];
    const uint64_t b = (uint64_t)1 << (j & 63);
    if (L->xnor) {
      if (v > 0) o->Wp[l][(size_t)n * kw + j / 64] |= b;
    } else {
      if (v < 0) o->Wp[l][(size_t)n * kw + j / 64] |= b; /* sign plane */
      if (v != 0) o->Wn[l][(size_t)n * kw + j / 64] |= b;
      if (v == -2) o->Wt[l][(size_t)n * kw + j / 64] |= b;
    }
  }
}

static int load_layer(bnn_oracle *o, const char *dir, int l) {
  lcfg *L = &o->L[l];
  const int kw = (L->mw + 63) / 64;
  char path[4096];
  o->wraw[l] = (uint64_t *)calloc((size_t)L->pe * L->wmem, 8);
  o->traw[l] = (uint64_t *)calloc((size_t)L->pe * L->tmem * 2 + 1, 8);
  o->W[l] = (int8_t *)calloc((size_t)L->mh * L->mw, 1);
  o->T[l] = (int32_t *)calloc((size_t)L->mh * 2, sizeof(int32_t));
  o->Wp[l] = (uint64_t *)calloc((size_t)L->mh * kw, 8);
  o->Wn[l] = (uint64_t *)calloc((size_t)L->mh * kw, 8);
  o->Wt[l] = (uint64_t *)calloc((size_t)L->mh * kw, 8);
  for (int p = 0; p < L->pe; p++) {
    snprintf(path, sizeof(path), "%s/%d-%d-weights.bin", dir, l, p);
    if (read_words(path, o->wraw[l] + (size_t)p * L->wmem, (size_t)L->wmem)) {
      fprintf(stderr, "bnn_oracle: Could not open file %s\n", path);
      return -1;
    }
    if (L->nthr > 0) {
      snprintf(path, sizeof(path), "%s/%d-%d-thres.bin", dir, l, p);
      if (read_words(path, o->traw[l] + (size_t)p * L->tmem * L->nthr, (size_t)L->tmem * L->nthr)) {
        fprintf(stderr, "bnn_oracle: Could not open file %s\n", path);
        return -1;
      }
    }
  }
  for (int n = 0; n < L->mh; n++) derive_row(o, l, n);
  return 0;
}

/* --------------------------------------------------------------------------
 * Fault injection on one memory word: inject_fault_impl (foldedmv-offload.h:146-163) through
 * FoldedMVMemRead / FoldedMVMemSet -> DoMemRead / DoMemInit (top.cpp:78-178).
 * target 0: weight memory word (layer, mem = PE, ind); 1: threshold (layer, mem, ind, thresh).
 * bit_pos is aligned down to a multiple of word_size, word_size adjacent bits are flipped.
 * -------------------------------------------------------------------------- */
int bnn_oracle_apply_fault(bnn_oracle *o, int target, int layer, int mem, int ind, int thresh, int bit_pos,
                           int word_size) {
  if (layer < 0 || layer >= o->nl || word_size < 1 || word_size > 64) return -1;
  lcfg *L = &o->L[layer];
  uint64_t flip = (word_size >= 64) ? ~(uint64_t)0 : (((uint64_t)1 << word_size) - 1);
  flip <<= (bit_pos / word_size) * word_size;
  int n;
  if (target == 0) {
    if (mem >= L->pe || ind >= L->wmem) return -1;
    const int ebits = L->simd * L->wbits; /* m_weights is ap_uint<SIMD*WPI> */
    const uint64_t emask = (ebits >= 64) ? ~(uint64_t)0 : (((uint64_t)1 << ebits) - 1);
    uint64_t *w = &o->wraw[layer][(size_t)mem * L->wmem + ind];
    *w = ((*w & emask) ^ flip) & emask;
    n = (ind / (L->wmem / L->tmem)) * L->pe + mem;
  } else {
    if (L->nthr == 0 || mem >= L->pe || ind >= L->tmem || thresh >= L->nthr) return -1;
    uint64_t *t = &o->traw[layer][(size_t)mem * L->tmem * L->nthr + (size_t)ind * L->nthr + thresh];
    int64_t v;
    if (L->thr24) {
      /* DoMemRead: static_cast<ap_uint<64>>(ap_fixed<24,16>) = the INTEGER part (top.cpp:143);
       * DoMemInit then reinterprets the word as ap_fixed<64,56> (top.cpp:84) */
      int32_t t24 = (int32_t)(*t & 0xFFFFFF);
      if (t24 & 0x800000) t24 -= 0x1000000;
      v = (int64_t)(t24 >> 8);
    } else {
      v = (int64_t)(int16_t)(*t & 0xFFFF);
    }
    *t = (uint64_t)v ^ flip;
    n = ind * L->pe + mem;
  }
  derive_row(o, layer, n);
  return n;
}

bnn_oracle *bnn_oracle_create(const char *network, const char *param_dir) {
  bnn_oracle *o = (bnn_oracle *)malloc(sizeof(bnn_oracle));
  if (setup_net(o, network)) {
    fprintf(stderr, "bnn_oracle: unknown network %s\n", network);
    free(o);
    return NULL;
  }
  for (int l = 0; l < o->nl; l++) {
    if (load_layer(o, param_dir, l)) {
      bnn_oracle_destroy(o);
      return NULL;
    }
  }
  return o;
}

void bnn_oracle_destroy(bnn_oracle *o) {
  if (!o) return;
  for (int l = 0; l < 9; l++) {
    free(o->W[l]); free(o->T[l]); free(o->Wp[l]); free(o->Wn[l]); free(o->Wt[l]); free(o->wraw[l]); free(o->traw[l]);
  }
  free(o);
}

int bnn_oracle_is_cnv(const bnn_oracle *o) { return o->is_cnv; }
int bnn_oracle_num_layers(const bnn_oracle *o) { return o->nl; }
int bnn_oracle_layer_mw(const bnn_oracle *o, int l) { return o->L[l].mw; }
int bnn_oracle_layer_mh(const bnn_oracle *o, int l) { return o->L[l].mh; }
int bnn_oracle_weight(const bnn_oracle *o, int l, int n, int j) {
  return o->W[l][(size_t)n * o->L[l].mw + j];
}
int bnn_oracle_threshold(const bnn_oracle *o, int l, int n, int i) {
  return o->T[l][n * 2 + i];
}

/* --------------------------------------------------------------------------
 * A2. Input conversion.  tiny-cnn parse_cifar10(path,..,-1.0,1.0,0,0) scales a
 * byte p to float x = -1 + 2*p/255 (float_t = float); quantiseAndPack<8,1>
 * (foldedmv-offload.h:129-144) converts through ap_fixed<8,1,AP_RND,AP_SAT>:
 * 7 fraction bits, round half towards +inf, saturate -> int8 q = x*128.
 * -------------------------------------------------------------------------- */
int bnn_oracle_quantise_u8(int p) {
  const float x = -1.0f + (1.0f - (-1.0f)) * (float)p / 255.0f;
  double r = floor((double)x * 128.0 + 0.5);
  if (r > 127.0) r = 127.0;
  if (r < -128.0) r = -128.0;
  return (int)r;
}

/* --------------------------------------------------------------------------
 * A5/A6. One MVAU + activation on one input vector (faithful form).
 * Matrix_Vector_Activate_Batch<...> (finn-hlslib mvau.hpp, un-vendored;
 * instantiations top.cpp:214-235) + ThresholdsActivation::activate
 * (activations.hpp; std::less<TA>(thr, acc) => strict thr < acc).
 * in: mw values; out: mh values (value domain) or raw accumulators.
 * -------------------------------------------------------------------------- */
static void mvau_ref(const bnn_oracle *o, int l, const int8_t *in, int8_t *out,
                     int32_t *raw) {
  const lcfg *L = &o->L[l];
  const int8_t *W = o->W[l];
  const int32_t *T = o->T[l];
  for (int n = 0; n < L->mh; n++) {
    int32_t acc = 0;
    const int8_t *w = W + (size_t)n * L->mw;
    if (L->xnor) {
      for (int j = 0; j < L->mw; j++) acc += (w[j] == in[j]); /* XNOR popcount */
    } else {
      for (int j = 0; j < L->mw; j++) acc += (int32_t)w[j] * (int32_t)in[j];
      if (L->in_mode == IN_INT8) acc *= 2; /* q/128 held with 8 fraction bits */
    }
    if (raw) raw[n] = acc;
    if (out) {
      if (L->nthr == 1) out[n] = (T[n * 2] < acc) ? 1 : -1;
      else if (L->nthr == 2) out[n] = (int8_t)(-1 + (T[n * 2] < acc) + (T[n * 2 + 1] < acc));
    }
  }
}

/* A4 + A8: ConvolutionInputGenerator<3,..> + MVAU per output pixel, row-major
 * pixels, window order (ky, kx, c) (slidingwindow.h; finnthesizer.py:310-316). */
static void conv_ref(const bnn_oracle *o, int l, const int8_t *in, int8_t *out) {
  const lcfg *L = &o->L[l];
  const int C = L->ifm_ch, D = L->ifm_dim, OD = L->ofm_dim;
  int8_t *col = (int8_t *)malloc((size_t)L->mw);
  for (int oy = 0; oy < OD; oy++)
    for (int ox = 0; ox < OD; ox++) {
      for (int ky = 0; ky < 3; ky++)
        for (int kx = 0; kx < 3; kx++)
          for (int c = 0; c < C; c++)
            col[(ky * 3 + kx) * C + c] = in[((oy + ky) * D + ox + kx) * C + c];
      mvau_ref(o, l, col, out + (size_t)(oy * OD + ox) * L->ofm_ch, NULL);
    }
  free(col);
}

/* A7: StreamingMaxPool_Batch / _Precision_Batch: 2x2 stride-2 max per channel
 * on the thresholded maps (max of +-1 == OR of the bits). */
static void pool_ref(const int8_t *in, int8_t *out, int D, int C) {
  const int OD = D / 2;
  for (int y = 0; y < OD; y++)
    for (int x = 0; x < OD; x++)
      for (int c = 0; c < C; c++) {
        int8_t m = in[((2 * y) * D + 2 * x) * C + c];
        const int8_t b = in[((2 * y) * D + 2 * x + 1) * C + c];
        const int8_t d = in[((2 * y + 1) * D + 2 * x) * C + c];
        const int8_t e = in[((2 * y + 1) * D + 2 * x + 1) * C + c];
        if (b > m) m = b;
        if (d > m) m = d;
        if (e > m) m = e;
        out[(y * OD + x) * C + c] = m;
      }
}

/* runs layers 0..upto (inclusive, with pooling); returns element count in buf */
static int cnv_forward_ref(const bnn_oracle *o, const uint8_t *img, int upto,
                           int8_t *bufA, int8_t *bufB, int16_t *scores) {
  /* chaninterleave_layer (foldedmv-offload.h:381-386): CHW -> HWC, then int8 */
  for (int y = 0; y < 32; y++)
    for (int x = 0; x < 32; x++)
      for (int c = 0; c < 3; c++)
        bufA[(y * 32 + x) * 3 + c] =
            (int8_t)bnn_oracle_quantise_u8(img[c * 1024 + y * 32 + x]);
  int8_t *cur = bufA, *nxt = bufB;
  int count = 0;
  for (int l = 0; l <= upto && l < 9; l++) {
    const lcfg *L = &o->L[l];
    if (L->is_conv) {
      conv_ref(o, l, cur, nxt);
      count = L->ofm_dim * L->ofm_dim * L->ofm_ch;
      { int8_t *t = cur; cur = nxt; nxt = t; }
      if (L->pool) {
        pool_ref(cur, nxt, L->ofm_dim, L->ofm_ch);
        count /= 4;
        { int8_t *t = cur; cur = nxt; nxt = t; }
      }
    } else if (L->nthr > 0) {
      mvau_ref(o, l, cur, nxt, NULL);
      count = L->mh;
      { int8_t *t = cur; cur = nxt; nxt = t; }
    } else {
      /* PassThroughActivation<ap_uint<16>> (top.cpp:232-235) */
      int32_t raw[64];
      mvau_ref(o, l, cur, NULL, raw);
      if (scores)
        for (int n = 0; n < 64; n++) scores[n] = (int16_t)(uint16_t)(raw[n] & 0xFFFF);
      count = 0;
    }
  }
  if (cur != bufA && count > 0) memcpy(bufA, cur, (size_t)count);
  return count;
}

#define CNV_BUF (30 * 30 * 64)

void bnn_oracle_cnv_scores_ref(const bnn_oracle *o, const uint8_t *img, int16_t scores[64]) {
  int8_t *a = (int8_t *)malloc(CNV_BUF), *b = (int8_t *)malloc(CNV_BUF);
  cnv_forward_ref(o, img, 8, a, b, scores);
  free(a); free(b);
}

/* binarizeAndPack (foldedmv-offload.cpp:82-98): bit i = (x[i] >= 0) with
 * x = -1 + 2*p/255 (parse_mnist_images(path,..,-1.0,1.0,0,0)), padding 0. */
static int lfc_forward_ref(const bnn_oracle *o, const uint8_t *px, int upto, int8_t *out) {
  int8_t a[1024], b[1024];
  for (int i = 0; i < 832; i++) {
    if (i < 784) {
      const float x = ((float)px[i] / 255.0f) * (1.0f - (-1.0f)) + (-1.0f);
      a[i] = (x >= 0) ? 1 : -1;
    } else {
      a[i] = -1; /* FOLDEDMV_INPUT_PADCHAR 0 -> bit 0 -> -1 */
    }
  }
  int8_t *cur = a, *nxt = b;
  int count = 0;
  for (int l = 0; l <= upto && l < 4; l++) {
    mvau_ref(o, l, cur, nxt, NULL);
    count = o->L[l].mh;
    { int8_t *t = cur; cur = nxt; nxt = t; }
  }
  memcpy(out, cur, (size_t)count);
  return count;
}

/* binarizeAndPack itself (foldedmv-offload.cpp:82-98), the words the reference's host ships to the accelerator
 * (:186-194): memset to FOLDEDMV_INPUT_PADCHAR (0), then bit i of word i / 64 set where in[i] >= 0, in[i] being
 * tiny-cnn's float scaling of pixel i to [-1, 1]. */
void bnn_oracle_lfc_binarize(const uint8_t *px, uint64_t words[13]) {
  memset(words, 0, 13 * sizeof(uint64_t));
  for (int i = 0; i < 784; i++) {
    const float x = ((float)px[i] / 255.0f) * (1.0f - (-1.0f)) + (-1.0f);
    if (x >= 0) words[i / 64] |= (uint64_t)1 << (i % 64);
    else words[i / 64] &= ~((uint64_t)1 << (i % 64));
  }
}

uint64_t bnn_oracle_lfc_word_ref(const bnn_oracle *o, const uint8_t *px) {
  int8_t out[1024];
  lfc_forward_ref(o, px, 3, out);
  uint64_t w = 0;
  for (int n = 0; n < 64; n++)
    if (out[n] > 0) w |= (uint64_t)1 << n;
  return w;
}

int bnn_oracle_layer_ref(const bnn_oracle *o, const uint8_t *img, int layer,
                         int8_t *out, int cap) {
  int count;
  if (layer < 0 || layer >= o->nl) return -1;
  if (o->is_cnv) {
    if (layer == 8) return -1;
    int8_t *a = (int8_t *)malloc(CNV_BUF), *b = (int8_t *)malloc(CNV_BUF);
    count = cnv_forward_ref(o, img, layer, a, b, NULL);
    if (count > cap) count = -1; else memcpy(out, a, (size_t)count);
    free(a); free(b);
  } else {
    int8_t tmp[1024];
    count = lfc_forward_ref(o, img, layer, tmp);
    if (count > cap) count = -1; else memcpy(out, tmp, (size_t)count);
  }
  return count;
}

/* --------------------------------------------------------------------------
 * Fast path.  Activations bit-packed pixel-major, channel c of a pixel at bit
 * c of that pixel's words (the reference's own inter-layer stream layout,
 * SURVEY A3).  1-bit activations: plane P (bit=1 <=> +1).  2-bit: sign plane S
 * (bit=1 <=> -1) and non-zero plane Z.
 * -------------------------------------------------------------------------- */
static inline int pc64(uint64_t x) { return __builtin_popcountll(x); }

typedef struct { uint64_t *s, *z; } planes; /* xnor nets use only s (= P) */

/* one neuron over kw words of (gathered) input */
static inline int32_t dot_fast(const lcfg *L, const uint64_t *wp, const uint64_t *wn, const uint64_t *wt,
                               const uint64_t *as, const uint64_t *az, int kw) {
  int32_t acc = 0;
  if (L->xnor) {
    for (int k = 0; k < kw; k++) acc += pc64(~(wp[k] ^ as[k]));
  } else {
    int32_t nzc = 0, neg = 0;
    for (int k = 0; k < kw; k++) {
      const uint64_t nz = wn[k] & az[k];
      nzc += pc64(nz);
      neg += pc64(nz & (wp[k] ^ as[k]));
    }
    acc = nzc - 2 * neg;
    /* a weight of -2 sits in both planes and has been counted as -1: the other -a_j */
    for (int k = 0; k < kw; k++) {
      const uint64_t x = wt[k] & az[k];
      acc += 2 * pc64(x & as[k]) - pc64(x);
    }
  }
  return acc;
}

static inline void act_store(const bnn_oracle *o, int l, int n, int32_t acc,
                             uint64_t *os, uint64_t *oz) {
  const lcfg *L = &o->L[l];
  const int32_t *T = o->T[l];
  const uint64_t b = (uint64_t)1 << (n & 63);
  if (o->abits == 1 || L->nthr == 1) {
    if (T[n * 2] < acc) os[n / 64] |= b; /* P plane */
  } else {
    const int v = -1 + (T[n * 2] < acc) + (T[n * 2 + 1] < acc);
    if (v < 0) os[n / 64] |= b;
    if (v != 0) oz[n / 64] |= b;
  }
}

/* max over two ternary values given as (s,z) bits, bitwise for 64 channels */
static inline void tmax(uint64_t s1, uint64_t z1, uint64_t s2, uint64_t z2,
                        uint64_t *s, uint64_t *z) {
  /* value: +1 (z&~s), 0 (~z), -1 (z&s) */
  const uint64_t p1 = z1 & ~s1, p2 = z2 & ~s2;
  const uint64_t n1 = z1 & s1, n2 = z2 & s2;
  const uint64_t pos = p1 | p2;
  const uint64_t neg = n1 & n2;
  *s = neg;
  *z = pos | neg;
}

static void cnv_one_fast(const bnn_oracle *o, const uint8_t *img, int16_t *scores) {
  /* layer 0: int8 x {-1,0,+1}, direct */
  static const int MAXW = 30 * 30 * 2; /* words per plane, generous */
  uint64_t *As = (uint64_t *)calloc(4 * (size_t)MAXW, 8);
  uint64_t *Az = As + MAXW, *Bs = Az + MAXW, *Bz = Bs + MAXW;
  int8_t q[32 * 32 * 3];
  int lut[256];
  for (int p = 0; p < 256; p++) lut[p] = bnn_oracle_quantise_u8(p);
  for (int y = 0; y < 32; y++)
    for (int x = 0; x < 32; x++)
      for (int c = 0; c < 3; c++)
        q[(y * 32 + x) * 3 + c] = (int8_t)lut[img[c * 1024 + y * 32 + x]];
  {
    const lcfg *L = &o->L[0];
    for (int oy = 0; oy < 30; oy++)
      for (int ox = 0; ox < 30; ox++) {
        uint64_t *os = As + (oy * 30 + ox), *oz = Az + (oy * 30 + ox);
        for (int n = 0; n < 64; n++) {
          const int8_t *w = o->W[0] + (size_t)n * L->mw;
          int32_t acc = 0;
          for (int ky = 0; ky < 3; ky++)
            for (int kx = 0; kx < 3; kx++)
              for (int c = 0; c < 3; c++)
                acc += w[(ky * 3 + kx) * 3 + c] * q[((oy + ky) * 32 + ox + kx) * 3 + c];
          act_store(o, 0, n, 2 * acc, os, oz);
        }
      }
  }
  uint64_t *cs = As, *cz = Az, *ns = Bs, *nz = Bz;
  uint64_t cols[36], colz[36];
  for (int l = 1; l < 9; l++) {
    const lcfg *L = &o->L[l];
    const int kw = L->mw / 64;
    if (L->is_conv) {
      const int cw = L->ifm_ch / 64, ow = L->ofm_ch / 64, D = L->ifm_dim, OD = L->ofm_dim;
      memset(ns, 0, (size_t)OD * OD * ow * 8);
      memset(nz, 0, (size_t)OD * OD * ow * 8);
      for (int oy = 0; oy < OD; oy++)
        for (int ox = 0; ox < OD; ox++) {
          for (int ky = 0; ky < 3; ky++)
            for (int kx = 0; kx < 3; kx++)
              for (int k = 0; k < cw; k++) {
                cols[(ky * 3 + kx) * cw + k] = cs[((oy + ky) * D + ox + kx) * cw + k];
                colz[(ky * 3 + kx) * cw + k] = cz[((oy + ky) * D + ox + kx) * cw + k];
              }
          for (int n = 0; n < L->mh; n++) {
            const int32_t acc = dot_fast(L, o->Wp[l] + (size_t)n * kw, o->Wn[l] + (size_t)n * kw, o->Wt[l] + (size_t)n * kw,
                                         cols, colz, kw);
            act_store(o, l, n, acc, ns + (oy * OD + ox) * ow, nz + (oy * OD + ox) * ow);
          }
        }
      { uint64_t *t = cs; cs = ns; ns = t; t = cz; cz = nz; nz = t; }
      if (L->pool) {
        const int PD = OD / 2;
        for (int y = 0; y < PD; y++)
          for (int x = 0; x < PD; x++)
            for (int k = 0; k < ow; k++) {
              const int i00 = ((2 * y) * OD + 2 * x) * ow + k, i01 = i00 + ow;
              const int i10 = ((2 * y + 1) * OD + 2 * x) * ow + k, i11 = i10 + ow;
              if (o->abits == 1) {
                ns[(y * PD + x) * ow + k] = cs[i00] | cs[i01] | cs[i10] | cs[i11];
              } else {
                uint64_t s, z, s2, z2;
                tmax(cs[i00], cz[i00], cs[i01], cz[i01], &s, &z);
                tmax(cs[i10], cz[i10], cs[i11], cz[i11], &s2, &z2);
                tmax(s, z, s2, z2, &ns[(y * PD + x) * ow + k], &nz[(y * PD + x) * ow + k]);
              }
            }
        { uint64_t *t = cs; cs = ns; ns = t; t = cz; cz = nz; nz = t; }
      }
    } else if (L->nthr > 0) {
      memset(ns, 0, (size_t)(L->mh / 64) * 8);
      memset(nz, 0, (size_t)(L->mh / 64) * 8);
      for (int n = 0; n < L->mh; n++) {
        const int32_t acc = dot_fast(L, o->Wp[l] + (size_t)n * kw, o->Wn[l] + (size_t)n * kw, o->Wt[l] + (size_t)n * kw, cs, cz, kw);
        act_store(o, l, n, acc, ns, nz);
      }
      { uint64_t *t = cs; cs = ns; ns = t; t = cz; cz = nz; nz = t; }
    } else {
      for (int n = 0; n < 64; n++) {
        const int32_t acc = dot_fast(L, o->Wp[l] + (size_t)n * kw, o->Wn[l] + (size_t)n * kw, o->Wt[l] + (size_t)n * kw, cs, cz, kw);
        scores[n] = (int16_t)(uint16_t)(acc & 0xFFFF);
      }
    }
  }
  free(As);
}

void bnn_oracle_cnv_scores_fast(const bnn_oracle *o, const uint8_t *imgs, int n,
                                int16_t *scores, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 4)
  for (int i = 0; i < n; i++)
    cnv_one_fast(o, imgs + (size_t)i * 3072, scores + (size_t)i * 64);
}

static uint64_t lfc_one_fast(const bnn_oracle *o, const uint8_t *px) {
  uint64_t as[16], az[16], bs[16], bz[16];
  memset(as, 0, sizeof(as));
  for (int i = 0; i < 784; i++)
    if (px[i] >= 128) as[i / 64] |= (uint64_t)1 << (i & 63);
  if (!o->L[0].xnor) {
    /* signed form: P plane -> (sign, nonzero): sign = ~P, all 832 columns non-zero */
    for (int k = 0; k < 13; k++) { as[k] = ~as[k]; az[k] = ~(uint64_t)0; }
  }
  uint64_t *cs = as, *cz = az, *ns = bs, *nz = bz;
  for (int l = 0; l < 4; l++) {
    const lcfg *L = &o->L[l];
    const int kw = L->mw / 64;
    memset(ns, 0, 16 * 8);
    memset(nz, 0, 16 * 8);
    for (int n = 0; n < L->mh; n++) {
      const int32_t acc = dot_fast(L, o->Wp[l] + (size_t)n * kw, o->Wn[l] + (size_t)n * kw, o->Wt[l] + (size_t)n * kw, cs, cz, kw);
      act_store(o, l, n, acc, ns, nz);
    }
    { uint64_t *t = cs; cs = ns; ns = t; t = cz; cz = nz; nz = t; }
  }
  return cs[0];
}

void bnn_oracle_lfc_words_fast(const bnn_oracle *o, const uint8_t *imgs, int n,
                               uint64_t *words, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) words[i] = lfc_one_fast(o, imgs + (size_t)i * 784);
}

/* --------------------------------------------------------------------------
 * A9/A10. Output decode.
 * -------------------------------------------------------------------------- */
/* testPrebuiltCIFAR10_multiple_images, foldedmv-offload.h:396-408:
 * maxInd=0, maxVal=0, strict '>' => first strict max, floored at 0. */
int bnn_oracle_decode_cnv_batched(const int16_t *s, int ncls) {
  int maxInd = 0;
  int16_t maxVal = 0;
  for (int j = 0; j < ncls; j++)
    if (s[j] > maxVal) { maxVal = s[j]; maxInd = j; }
  return maxInd;
}
/* inference(): std::max_element over number_class scores, main_python.cpp:138 */
int bnn_oracle_decode_cnv_single(const int16_t *s, int ncls) {
  int best = 0;
  for (int j = 1; j < ncls; j++)
    if (s[j] > s[best]) best = j;
  return best;
}
/* testPrebinarized_nolabel_multiple_images, foldedmv-offload.cpp:202-220 */
int bnn_oracle_decode_lfc_batched(uint64_t word, int ncls) {
  uint64_t mask = 0xFFFFFFFFFFFFFFFFull >> (64 - ncls);
  word &= mask;
  if (word == 0) return 0;
  return (int)(unsigned int)log2((double)word);
}
/* testPrebinarized_nolabel, foldedmv-offload.cpp:144-166 + argmax of the
 * 64-entry one-hot vector (lfcW1A1/sw/main_python.cpp:132) */
int bnn_oracle_lfc_single_hot(uint64_t word, int ncls) {
  uint64_t mask = 0xFFFFFFFFFFFFFFFFull >> (64 - ncls);
  word &= mask;
  double r = (word == 0) ? 0.0 : round(log2((double)word));
  return (int)(unsigned int)r; /* index of the one-hot entry; 64 => none of the 64 is set */
}
int bnn_oracle_decode_lfc_single(uint64_t word, int ncls) {
  const int idx = bnn_oracle_lfc_single_hot(word, ncls);
  return (idx < 64) ? idx : 0; /* no entry set => max_element returns 0 */
}

/* --------------------------------------------------------------------------
 * File parsers (tiny-cnn behaviours).
 * -------------------------------------------------------------------------- */
int bnn_oracle_parse_cifar10(const char *path, uint8_t **out) {
  FILE *f = fopen(path, "rb");
  if (!f) return -1;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  int n = (int)(sz / 3073);
  uint8_t *buf = (uint8_t *)malloc((size_t)(n > 0 ? n : 1) * 3072);
  for (int i = 0; i < n; i++) {
    uint8_t label;
    if (fread(&label, 1, 1, f) != 1 || fread(buf + (size_t)i * 3072, 1, 3072, f) != 3072) {
      n = i;
      break;
    }
  }
  fclose(f);
  *out = buf;
  return n;
}

int bnn_oracle_parse_mnist(const char *path, uint8_t **out) {
  FILE *f = fopen(path, "rb");
  if (!f) return -1;
  unsigned char h[16];
  if (fread(h, 1, 16, f) != 16) { fclose(f); return -1; }
  const uint32_t magic = (h[0] << 24) | (h[1] << 16) | (h[2] << 8) | h[3];
  const uint32_t n = ((uint32_t)h[4] << 24) | (h[5] << 16) | (h[6] << 8) | h[7];
  const uint32_t rows = ((uint32_t)h[8] << 24) | (h[9] << 16) | (h[10] << 8) | h[11];
  const uint32_t cols = ((uint32_t)h[12] << 24) | (h[13] << 16) | (h[14] << 8) | h[15];
  if (magic != 0x00000803 || rows != 28 || cols != 28) { fclose(f); return -1; }
  uint8_t *buf = (uint8_t *)malloc((size_t)(n > 0 ? n : 1) * 784);
  uint32_t got = 0;
  for (; got < n; got++)
    if (fread(buf + (size_t)got * 784, 1, 784, f) != 784) break;
  fclose(f);
  if (got != n) { free(buf); return -1; }
  *out = buf;
  return (int)n;
}

void bnn_oracle_free(void *p) { free(p); }
