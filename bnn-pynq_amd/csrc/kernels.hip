// kernels.hip -- CDNA4 (gfx950) kernels of the FINN QNN hot path.
//
// What they replace (all of it un-vendored finn-hlslib code instantiated from
// bnn/src/network/<net>/hw/top.cpp:210-236):
//   ConvolutionInputGenerator (SWU / im2col)            -> the window gather at the top of each conv kernel
//   Matrix_Vector_Activate_Batch (MVAU)                 -> the per-neuron XNOR/AND + popcount loops
//   ThresholdsActivation::activate                      -> the compare against the row's thresholds
//   StreamingMaxPool_Batch / _Precision_Batch           -> min/max over the 2x2 quad before the compare
//   StreamingDataWidthConverter / Mem2Stream / Stream2Mem -> nothing: activations stay bit-packed HWC words in HBM
//
// Execution model (why it looks like this on MI355X):
//   * One LANE owns one work item (an output pixel, a 2x2 quad of output
//     pixels, or an image) and keeps that item's input window in VGPRs for the
//     whole kernel: activations are read from HBM/L2 exactly once per stage.
//   * All 64 lanes of a wave (and all 4 waves of a block) work on the SAME 32
//     output neurons (blockIdx.y selects the group).  The weight row of the
//     neuron being evaluated is therefore wave-uniform: it is fetched with
//     wide SCALAR loads (s_load_dwordx8/x16 through the constant address
//     space) and used directly as the SGPR operand of v_xor_b32 / v_and_b32 /
//     v_dot4c_i32_i8.  Weights cost no VGPRs, no LDS traffic and no vector
//     memory instructions; the VALU does nothing but xor + v_bcnt (which folds
//     the accumulate).  That is 4 VALU ops per 64-bit word of 1-bit MACs.
//   * The 32 results of a lane are assembled into one 32-bit word in a VGPR and
//     stored once: outputs are already in the bit-packed layout the next stage
//     reads.  Max-pool is a min/max over the quad's 4 accumulators before the
//     threshold compare (thresholding is monotone), so it is free.
//   * No MFMA in the bitwise layers.  No LDS in the throughput kernels: there is no data shared
//     between lanes that the scalar path does not already broadcast for free.  The exceptions,
//     each argued where it is defined: the int8 first layer runs on the matrix pipe
//     (k_conv0_mfma); small batches, where a lane per item leaves the chip empty, use a lane
//     per output pixel, 8-neuron blocks, a wave per image (k_fclast_wave) and the one-launch
//     block-per-image kernels with LDS + wave ballots (k_lfc_fused*, k_cnv_tail*).
//
// Data layout in HBM: see DESIGN.md ("Data layout").
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "kernels.h"
#include "packed_params.h"
static_assert(bnn::kL0TileOffset == (int)bnn::kL0MfmaTileOffset && bnn::kL0BigBytes == (int)bnn::kL0MfmaBigBytes,
              "k_conv0_tile's operand offsets must be the ones packed_params.cpp writes the layer-0 table by");

namespace bnn {
namespace {

// scalar (constant) address space: uniform loads through it become s_load_*
typedef const uint32_t __attribute__((address_space(4))) *kptr32;
typedef const uint64_t __attribute__((address_space(4))) *kptr64;

constexpr int kBlock = 256;
constexpr int kLfcFusedMaxA2 = 2048;  // images: up to here lfcW1A2's one-launch kernel beats the six staged ones (tools/batch_sweep.py)

// Block -> (work-item block, neuron group), XCD-aware.  The `groups` blocks that evaluate
// different 32-neuron groups for the SAME 256 work items read the same input windows and write
// the interleaved dwords of the same output lines.  Workgroups are dealt round-robin over the 8
// XCDs (each with its own L2), so those blocks are given ids that are congruent mod 8 and
// adjacent in dispatch order: the re-reads then hit that XCD's L2 and the partial-line writes
// merge there instead of going to HBM once per group (measured: profiles/r01_pmc_*).
// Placement only affects speed/traffic, never results.
struct BlockMap { int cg, item; bool valid; };
__device__ __forceinline__ BlockMap map_block(int groups, int n_items) {
  const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
  const int sg = slot / groups;
  BlockMap m;
  m.cg = slot - sg * groups;
  m.item = (sg * 8 + xcd) * kBlock + threadIdx.x;
  m.valid = m.item < n_items;
  return m;
}

__device__ __forceinline__ int pc64(uint64_t x) { return __builtin_popcountll(x); }

// A single-image call may ask for a completion mark in pinned host memory (runtime.hip, infer_direct): the wave that
// wrote the results drains its stores (system scope) and then stores `seq` -- the host spins on that word instead of
// waiting on the runtime.  Only meaningful for one-block launches.
__device__ __forceinline__ void mark_done(unsigned *done, unsigned seq, int lane) {
  if (!done) return;
  __threadfence_system();
  if (lane == 0) *reinterpret_cast<volatile unsigned *>(done) = seq;
}

// ---------------------------------------------------------------------------
// activation outputs
// ---------------------------------------------------------------------------
// 1-bit maps:  [pixel][Cout/32] dwords, bit c of dword g = channel 32g+c fired (+1).
// 2-bit maps:  [pixel][Cout/64][plane] u64, plane 0 = sign (1 <=> -1), plane 1 = non-zero.
// NPB = neurons per block: 32 (a dword of the word, the throughput form) or 8 (a byte of it, the
// small-batch form: four times as many, four times shorter blocks); g counts groups of NPB neurons.
template <bool OUT2, int NPB = 32>
__device__ __forceinline__ void store_bits(uint32_t *__restrict__ out, size_t pix, int groups, int g,
                                           uint32_t b0, uint32_t b1) {
  constexpr int PER64 = 64 / NPB;  // groups per 64 channels
  if constexpr (!OUT2) {
    if constexpr (NPB == 32) out[pix * groups + g] = b0;
    else reinterpret_cast<uint8_t *>(out)[pix * groups + g] = (uint8_t)b0;
  } else {
    const size_t w64 = pix * (groups / PER64) + g / PER64;
    const int sub = g % PER64;
    if constexpr (NPB == 32) {
      out[(w64 * 2 + 0) * 2 + sub] = b0;
      out[(w64 * 2 + 1) * 2 + sub] = b1;
    } else {
      uint8_t *o8 = reinterpret_cast<uint8_t *>(out);
      o8[(w64 * 2 + 0) * 8 + sub] = (uint8_t)b0;
      o8[(w64 * 2 + 1) * 8 + sub] = (uint8_t)b1;
    }
  }
}

// ---------------------------------------------------------------------------
// one neuron x one window: the MVAU inner product on bit-packed words
// ---------------------------------------------------------------------------
// ARITH        per 64 synapses                                   acc / compare
// AR_XNOR      m += popc(w ^ a)                                  fire = m < t
// AR_TB        m += popc(za & (sa ^ w))                          fire = T < nz - 2m   (nz = popc(za) summed once per window)
// AR_TT        z += popc(za & zw); m += popc(za & zw & (sa^sw))  fire = T < z - 2m
template <int ARITH>
__device__ __forceinline__ void mac(int &m, int &z, uint64_t as, uint64_t az, kptr64 w) {
  if constexpr (ARITH == AR_XNOR) {
    m += pc64(w[0] ^ as);
  } else if constexpr (ARITH == AR_TB) {
    m += pc64(az & (as ^ w[0]));
  } else {
    const uint64_t zz = az & w[1];
    z += pc64(zz);
    m += pc64(zz & (as ^ w[0]));
  }
}

template <int ARITH>
constexpr int planes_in() { return ARITH == AR_XNOR ? 1 : 2; }
template <int ARITH>
constexpr int wplanes() { return ARITH == AR_TT ? 2 : 1; }
// dwords per packed row (packed_params.cpp): {t0, t1, KW x weight planes}; AR_TT rows carry, behind the two
// planes, a third one marking the columns whose weight is -2, then a flag dword ("any such column") and a pad
// dword
template <int ARITH, int KW>
constexpr int row_dw() { return ARITH == AR_TT ? 4 + 6 * KW : 2 + 2 * KW * wplanes<ARITH>(); }
// ap_int<2> weights can be -2 (field 0b10): never in trained parameters, but one bit flip away from 0 and
// from -1, and the reference then multiplies by -2.  The kernels evaluate such a column as -1 (it is set in
// the sign and non-zero planes) and, for rows whose flag is set, add the missing -a_j:
// z += [a_j = -1] - [a_j = +1].  That code lives in separate instantiations (template parameter TWO) which
// the host selects only while some row of the network has its flag set (runtime.hip): inside the normal
// kernels it would cost registers -- occupancy is decided by the worst path of a kernel -- and scalar-load
// waits that the fault-free path must not pay.
__device__ __forceinline__ int two_extra(uint32_t two, uint32_t as, uint32_t az) {
  const uint32_t x = two & az;
  return 2 * __builtin_popcount(x & as) - __builtin_popcount(x);
}
__device__ __forceinline__ int two_extra64(uint64_t two, uint64_t as, uint64_t az) {
  const uint64_t x = two & az;
  return 2 * __builtin_popcountll(x & as) - __builtin_popcountll(x);
}

// (m, z, nz_total) -> the signed accumulator of the ternary forms (used by the last CNV layer)
template <int ARITH>
__device__ __forceinline__ int finish(int m, int z, int nzt) {
  if constexpr (ARITH == AR_XNOR) return m;
  else if constexpr (ARITH == AR_TB) return nzt - 2 * m;
  else return z - 2 * m;
}

// ---------------------------------------------------------------------------
// Stage 0 of the CNV nets: 32x32x3 uint8 (planar CHW, as in a CIFAR-10 record)
// -> 30x30x64 thresholded map.
// Replaces: chaninterleave + quantiseAndPack<8,1> (foldedmv-offload.h:129-144,
// 381-386), the 64->192->24 width converters and ConvLayer_Batch<L0..>
// (top.cpp:210-214).  One lane = one output pixel; the 27 int8 taps are gathered
// as 9 x 3 bytes, compacted into 7 dwords with v_perm_b32, and each neuron
// costs 7 v_dot4 against SGPR weight dwords + a subtract + a v_alignbit.
// ---------------------------------------------------------------------------
// uint8 p -> int8 q = clamp(floor(256*p/255 - 128 + 0.5)) = p - 128 + (p >= 128) - (p == 255),
// four bytes at a time (no carry can cross a byte: see DESIGN.md).
__device__ __forceinline__ uint32_t quantise4(uint32_t p) {
  const uint32_t t = p ^ 0x80808080u;                     // p - 128 per byte
  const uint32_t s = (t & 0x7F7F7F7Fu) + 0x01010101u;     // bit 7 of a byte: its low seven bits are all ones (p = 127 or 255)
  const uint32_t inc = p & ~s & 0x80808080u;              // 128 <= p < 255, at bit 7 (one v_bitop3)
  return t + (inc >> 7);
}

// bits = (bits << 1) | (v < 0)
// (the empty asm hides v's origin: otherwise LLVM turns "sign of a difference" back into
// v_cmp + v_cndmask + v_lshl_or, three integer-pipe slots instead of one)
__device__ __forceinline__ uint32_t shift_in_sign(uint32_t bits, int v) {
  asm("" : "+v"(v));
  return __builtin_amdgcn_alignbit(bits, (uint32_t)v, 31);
}

// The 27 int8 taps of output pixel `item` (3 channels x 3 rows x 3 columns), quantised, compacted
// into 7 dwords: tap 3*(c*3+ky) + kx at byte tau%4 of dword tau/4; byte 27 is a don't-care byte.
__device__ __forceinline__ void gather_taps(const uint8_t *__restrict__ imgs, int item, uint32_t (&a)[7]) {
  const int img = item / 900, p = item - img * 900;
  const int oy = p / 30, ox = p - oy * 30;
  const uint32_t *__restrict__ im32 = reinterpret_cast<const uint32_t *>(imgs + (size_t)img * 3072);
  const uint32_t *__restrict__ row0 = im32 + oy * 8 + (ox >> 2);
  const int last = 767 - (oy * 8 + (ox >> 2));  // the clamp below only ever feeds the don't-care byte
  const int sh = ox & 3;
  // g[c*3+r] = bytes {x, x+1, x+2, don't-care} of channel c, row oy+r
  uint32_t g[9];
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const int off = c * 256 + r * 8;
      const uint32_t d0 = row0[off];
      // (only the last dword read of the image's last rows can leave the image: off + 1 <= 529 and the row base is
      // at most 239, so the clamp is needed for c = r = 2 alone -- the others keep their immediate offsets)
      const uint32_t d1 = (c == 2 && r == 2) ? row0[off + 1 <= last ? off + 1 : last] : row0[off + 1];
      g[c * 3 + r] = __builtin_amdgcn_alignbyte(d1, d0, sh);
    }
#pragma unroll
  for (int h = 0; h < 2; h++) {
    a[3 * h + 0] = __builtin_amdgcn_perm(g[4 * h + 1], g[4 * h + 0], 0x04020100u);
    a[3 * h + 1] = __builtin_amdgcn_perm(g[4 * h + 2], g[4 * h + 1], 0x05040201u);
    a[3 * h + 2] = __builtin_amdgcn_perm(g[4 * h + 3], g[4 * h + 2], 0x06050402u);
  }
  a[6] = g[8];  // taps 24..26 + one don't-care byte (its weight byte is 0)
#pragma unroll
  for (int j = 0; j < 7; j++) a[j] = quantise4(a[j]);
}

template <bool OUT2>
__global__ __launch_bounds__(kBlock) void k_conv0(const uint8_t *__restrict__ imgs, uint32_t *__restrict__ out,
                                                   const uint32_t *__restrict__ rows, int n_items, int groups, int gpb) {
  const BlockMap bm = map_block(groups / gpb, n_items);
  if (!bm.valid) return;
  const int item = bm.item;
  uint32_t a[7];
  gather_taps(imgs, item, a);
  // neuron groups handled by this block: gpb = all of them for large batches (one lane then
  // writes whole output words), 1 for small ones (more blocks in flight)
  for (int cg = bm.cg * gpb, cg_end = cg + gpb; cg < cg_end; cg++) {
    kptr32 w = (kptr32)(uintptr_t)(rows + (size_t)cg * 32 * 12);
    uint32_t b0 = 0, b1 = 0;
    // Two neurons per iteration (two independent v_dot4c chains, strictly alternating).  Each
    // accumulator starts at ~t = -t-1, so that afterwards  acc < 0  <=>  dot <= t  <=>  !fire:
    // the threshold test costs no instruction, its result is the accumulator's sign bit.
    for (int c = 31; c >= 0; c -= 2) {
      kptr32 rA = w + c * 12, rB = rA - 12;
      int accA = ~(int)rA[0], accB = ~(int)rB[0];
  #pragma unroll
      for (int k = 0; k < 7; k++) {
        // (builtin, not inline asm: a VALU read of a v_dot4c result needs wait states that only
        // the compiler's hazard recogniser inserts; the empty asm just pins the issue order)
        accA = __builtin_amdgcn_sdot4((int)a[k], (int)rA[2 + k], accA, false);
        asm("" : "+v"(accA));
        accB = __builtin_amdgcn_sdot4((int)a[k], (int)rB[2 + k], accB, false);
        asm("" : "+v"(accB));
      }
      if constexpr (!OUT2) {
        b0 = shift_in_sign(b0, accA);
        b0 = shift_in_sign(b0, accB);
      } else {
        // second threshold: dot - t1 - 1 = acc + (t0 - t1)
        const int dA = accA + ((int)rA[0] - (int)rA[1]), dB = accB + ((int)rB[0] - (int)rB[1]);
        b0 = shift_in_sign(b0, accA & dA);  // sign plane: !f0 & !f1
        b1 = shift_in_sign(b1, accA ^ dA);  // (f0 != f1): inverted below
        b0 = shift_in_sign(b0, accB & dB);
        b1 = shift_in_sign(b1, accB ^ dB);
      }
    }
    if constexpr (!OUT2) b0 = ~b0;  // collected !fire
    else b1 = ~b1;
    store_bits<OUT2>(out, (size_t)item, 2, cg, b0, b1);
  }
}

// ---------------------------------------------------------------------------
// Stage 0 on the matrix pipe.  Layer 0 is the one layer of these networks that is NOT bitwise: an
// int8 x {-1,0,+1} contraction over 27 taps -- a small dense GEMM [64 neurons x 32] x [32 x pixels].
// On the integer pipe it costs 8 issue slots per pixel and neuron (k_conv0, 18 % of the whole
// cnvW1A1 run); as v_mfma_i32_32x32x32_i8 it runs on the otherwise idle matrix cores, and
// concurrently with the popcount stages' integer work of other waves.
//   D[neuron][pixel] = A[neuron][k] * B[k][pixel]      one wave = 64 pixels x 64 neurons = 4 MFMAs
//   A: rows of the blob's layer-0 MFMA table {27 taps, a0, a1, 0, 0, 0},  a0 + 64*a1 = -t0 - 1
//   B: the pixel's 27 int8 taps, then the constants 1 and 64   =>   D = dot - t0 - 1, sign bit = !fire
// Lane l = (r = l & 31, h = l >> 5) supplies k = 16h .. 16h+15 of pixel r (B) / of neuron r (A); the
// same (h, byte) -> k assignment on both operands is all the product needs.  The result has its
// pixel on the lane and 16 neurons 8g + 4h + q (reg 4g + q) in registers: their sign bits are shifted
// into nibbles, the partner half (l ^ 32) contributes the other nibbles (v_permlane32_swap), and
// every lane ends up with the 32-neuron dword of its pixel, already in the activation layout.
// ---------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// sign bits of the 16 accumulators of one lane -> bits 8g + 4h + q of a dword (other bits 0)
__device__ __forceinline__ uint32_t sign_nibbles(const int (&v)[16], int h) {
  uint32_t x = 0;
#pragma unroll
  for (int g = 3; g >= 0; g--) {
#pragma unroll
    for (int q = 3; q >= 0; q--) x = shift_in_sign(x, v[4 * g + q]);
    if (g) x <<= 4;
  }
  return x << (4 * h);
}
// OR with the partner lane l ^ 32
__device__ __forceinline__ uint32_t or_halves(uint32_t x) {
  const auto sw = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return sw[0] | sw[1];
}

template <bool OUT2>
__global__ __launch_bounds__(kBlock, 2) void k_conv0_mfma(const uint8_t *__restrict__ imgs, uint32_t *__restrict__ out,
                                                        const uint8_t *__restrict__ l0tab, int n_items) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int pix = blockIdx.x * kBlock + threadIdx.x;   // lane = output pixel, like k_conv0
  if (pix - lane >= n_items) return;                   // whole wave out of range (wave-uniform)
  uint32_t t[7];
  gather_taps(imgs, pix < n_items ? pix : n_items - 1, t);  // ragged tail: duplicate, store guarded
  // K slots 27 and 28 carry the constants 1 and 64 (threshold folded into the product), 29..31 zero
  uint32_t lo[4] = {t[0], t[1], t[2], t[3]};
  uint32_t hi[4] = {t[4], t[5], (t[6] & 0x00FFFFFFu) | 0x01000000u, 0x00000040u};
  // B operands of the two pixel tiles: lane (r, h) must hold k = 16h .. 16h+15 of pixel r.  Swapping
  // the upper half of `lo` with the lower half of `hi` does exactly that for pixels 0..31 (left in lo)
  // and for pixels 32..63 (left in hi): 4 v_permlane32_swap, no LDS.
  v4i bp[2];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const auto sw = __builtin_amdgcn_permlane32_swap(lo[k], hi[k], false, false);
    bp[0][k] = (int)sw[0];
    bp[1][k] = (int)sw[1];
  }
  const v4i *__restrict__ atab = reinterpret_cast<const v4i *>(l0tab);
  uint32_t w0[2] = {0, 0}, w1[2] = {0, 0};  // this lane's pixel: [neuron tile], plane 0 / plane 1
  const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int ct = 0; ct < 2; ct++) {
    const v4i a = atab[(32 * ct + r) * 2 + h];
    v4i a2 = a;
    if constexpr (OUT2) a2 = atab[(64 + 32 * ct + r) * 2 + h];  // the same rows with t1 folded in
#pragma unroll
    for (int pt = 0; pt < 2; pt++) {
      const v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bp[pt], zero, 0, 0, 0);
      int v[16];
#pragma unroll
      for (int i = 0; i < 16; i++) v[i] = acc[i];
      uint32_t y0, y1 = 0;
      if constexpr (!OUT2) {
        y0 = ~or_halves(sign_nibbles(v, h));  // collected !fire
      } else {
        // second threshold: one more MFMA (the matrix pipe is idle anyway) instead of 16 adds; the
        // planes are formed on the packed sign words, not per accumulator
        const v16i acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2, bp[pt], zero, 0, 0, 0);
        int u[16];
#pragma unroll
        for (int i = 0; i < 16; i++) u[i] = acc2[i];
        const uint32_t n0 = sign_nibbles(v, h), n1 = sign_nibbles(u, h);  // !f0, !f1 of this half's 16 neurons
        y0 = or_halves(n0 & n1);                                          // sign plane: neither fired
        y1 = ~or_halves(n0 ^ n1);                                         // non-zero plane: f0 == f1
      }
      // both halves now hold the words of pixel r of tile pt; a lane keeps those of its own pixel
      const uint32_t mine = (pt == h) ? 0xFFFFFFFFu : 0u;
      w0[ct] |= y0 & mine;
      w1[ct] |= y1 & mine;
    }
  }
  if (pix < n_items) {
    if constexpr (!OUT2) {
      *reinterpret_cast<uint2 *>(out + (size_t)pix * 2) = make_uint2(w0[0], w0[1]);
    } else {  // [pixel][C/64 = 1][plane][half]
      *reinterpret_cast<uint4 *>(out + (size_t)pix * 4) = make_uint4(w0[0], w0[1], w1[0], w1[1]);
    }
  }
}

// ---------------------------------------------------------------------------
// Stage 0 on the matrix pipe, throughput form (round 3).  k_conv0_mfma above spends 209 VALU instructions per
// 64-pixel wave, 105 of them on the window gather: every input byte is fetched by up to 9 lanes and quantised in
// up to 9 windows, and the 3-byte runs are compacted into the K = 32 operand with v_alignbyte / v_perm.  Here a
// block stages kL0Imgs images in LDS ONCE -- coalesced 16-byte loads, every byte quantised once -- and a run
// (c, ky) of a window is one 4-byte fetch from LDS (two aligned dwords + v_alignbyte): its three taps plus a don't-care byte whose weight is 0.
// No compaction: K grows from 32 to 48 (9 runs + the threshold constants, spread over v_mfma_i32_32x32x32_i8 and
// v_mfma_i32_32x32x16_i8, operand layout in packed_params.h) -- the matrix pipe was 12 % busy.  A wave takes a
// tile of 32 pixels x 64 neurons at a time: lane (r, h) fetches the half of pixel r's runs its K slots hold (no
// v_permlane swaps of the operands), the A operands stay in registers for all the block's tiles, the table's
// neuron order makes the 16 accumulators of a lane half 16 consecutive neurons (no shifts between nibbles), and
// ONE v_permlane32_swap + OR leaves lane (r, h) with output dword h of pixel r.
// ---------------------------------------------------------------------------
constexpr int kL0Imgs = 8;
constexpr int kL0Plane = 1024 + 64, kL0Image = 3 * kL0Plane;  // LDS bytes: channel planes padded so that the two lane halves
                                                              // (a plane apart) read different banks

// sign bits of 16 accumulators -> bits 0..15 (register i -> bit i)
__device__ __forceinline__ uint32_t sign_bits16(const v16i &acc) {
  uint32_t x = 0;
#pragma unroll
  for (int i = 15; i >= 0; i--) x = shift_in_sign(x, acc[i]);
  return x;
}
// lanes 0..31 contribute `a`, lanes 32..63 `b`; lane (r, h): (a of lane r | a of lane r + 32) for h = 0, the same of b for h = 1
__device__ __forceinline__ uint32_t merge_halves(uint32_t a, uint32_t b) {
  const auto sw = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  return sw[0] | sw[1];
}
// the 4 bytes at LDS byte address base + OFF (OFF a multiple of 4, base any alignment): two aligned dwords (one
// ds_read2_b32) and a v_alignbyte -- an unaligned ds_read_b32 is legal on gfx950 but executes lane by lane
// (measured: the first version of this kernel, one unaligned read per run, took 1.98 ms against 0.78 ms)
template <int OFF>
__device__ __forceinline__ uint32_t lds_run(const uint32_t *aligned, uint32_t shift) {
  return __builtin_amdgcn_alignbyte(aligned[OFF / 4 + 1], aligned[OFF / 4], shift);
}

template <bool OUT2>
__global__ __launch_bounds__(kBlock, 2) void k_conv0_tile(const uint8_t *__restrict__ imgs, uint32_t *__restrict__ out,
                                                        const uint8_t *__restrict__ l0tab, int n_images) {
  __shared__ uint4 q4[kL0Imgs * kL0Image / 16];  // quantised images, planar CHW int8
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img0 = blockIdx.x * kL0Imgs;
  const int cnt = __builtin_amdgcn_readfirstlane(min(kL0Imgs, n_images - img0));  // >= 1 by the grid size
  const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(imgs + (size_t)img0 * 3072);
  for (int i = tid; i < cnt * 192; i += kBlock) {
    uint4 v = src[i];
    v.x = quantise4(v.x); v.y = quantise4(v.y); v.z = quantise4(v.z); v.w = quantise4(v.w);
    const int g = i / 192, j = i - g * 192;                       // image, 16-byte piece of it
    q4[(g * kL0Image + (j >> 6) * kL0Plane + (j & 63) * 16) >> 4] = v;  // 64 pieces per channel plane
  }
  // A operands (loop-invariant): 16 + 8 (+ 8) bytes per lane and neuron tile
  const uint8_t *__restrict__ tab = l0tab + kL0TileOffset;
  v4i abig[2];
  long asm0[2], asm1[2];
#pragma unroll
  for (int ct = 0; ct < 2; ct++) {
    abig[ct] = *reinterpret_cast<const v4i *>(tab + ((ct * 32 + r) * 2 + h) * 16);
    asm0[ct] = *reinterpret_cast<const long *>(tab + kL0BigBytes + (((0 * 2 + ct) * 32 + r) * 2 + h) * 8);
    asm1[ct] = OUT2 ? *reinterpret_cast<const long *>(tab + kL0BigBytes + (((1 * 2 + ct) * 32 + r) * 2 + h) * 8) : 0;
  }
  __syncthreads();
  const uint8_t *qb = reinterpret_cast<const uint8_t *>(q4);
  const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // A tile is one output ROW: 30 of the 32 pixel lanes carry a pixel (the other two compute on the bytes behind the row
  // and store nothing), 30 tiles per image instead of 29 -- and everything a lane needs to find its bytes and its output
  // word is a wave-uniform term plus a constant of the lane, where tiles of 32 consecutive pixels of the flattened map
  // (29 per image, the first form) cost thirteen VALU instructions of divide-by-30 arithmetic per tile against six now.
  // Measured gain: 2 % of the stage only (0.502 -> 0.491 ms, profiles/r03_layer0_row_tiles_ab.txt): the instructions
  // saved are the cheap full-rate kind and there are 3.4 % more tiles
  const uint32_t lane_byte = (uint32_t)(r + h * kL0Plane);
  const bool live = r < 30;
  for (int T = wave; T < cnt * 30; T += kBlock / 64) {
    const int gi = T / 30, oy = T - gi * 30;            // wave-uniform
    const size_t pix = (size_t)(img0 + gi) * 900 + (size_t)(oy * 30) + r;
    const uint32_t byte = (uint32_t)(gi * kL0Image + 32 * oy) + lane_byte;
    // this lane's runs: h = 0: (c,ky) = (0,0) (0,1) (0,2) (2,0) and, for the K = 16 product, (2,1);
    //                   h = 1: (1,0) (1,1) (1,2) (2,2) and the constants 1, 64
    const uint32_t *al = reinterpret_cast<const uint32_t *>(qb + (byte & ~3u));  // shift: (byte & 3) in both halves (kL0Plane % 4 == 0)
    const uint32_t *al2 = reinterpret_cast<const uint32_t *>(qb + ((byte & ~3u) - h * (kL0Plane - 64)));  // run (2,0) / (2,2): two rows apart
    v4i bb;
    bb[0] = (int)lds_run<0>(al, byte);
    bb[1] = (int)lds_run<32>(al, byte);
    bb[2] = (int)lds_run<64>(al, byte);
    bb[3] = (int)lds_run<2 * kL0Plane>(al2, byte);
    const uint32_t mid = lds_run<2 * kL0Plane + 32>(al2, byte);  // run (2,1) where h = 0
    const long bs = (long)(uint64_t)(h ? 0x00004001u : mid);
    uint32_t x[2], y[2];
    // both neuron tiles' products are issued before either is read: the second chain covers the first one's latency
    v16i big[2], d0[2], d1[2];
#pragma unroll
    for (int ct = 0; ct < 2; ct++) big[ct] = __builtin_amdgcn_mfma_i32_32x32x32_i8(abig[ct], bb, zero, 0, 0, 0);
#pragma unroll
    for (int ct = 0; ct < 2; ct++) {
      d0[ct] = __builtin_amdgcn_mfma_i32_32x32x16_i8(asm0[ct], bs, big[ct], 0, 0, 0);  // dot - t0 - 1: sign = !fire
      if constexpr (OUT2) d1[ct] = __builtin_amdgcn_mfma_i32_32x32x16_i8(asm1[ct], bs, big[ct], 0, 0, 0);  // second threshold
    }
#pragma unroll
    for (int ct = 0; ct < 2; ct++) {
      if constexpr (!OUT2) {
        x[ct] = sign_bits16(d0[ct]) << (16 * h);
      } else {
        const uint32_t n0 = sign_bits16(d0[ct]), n1 = sign_bits16(d1[ct]);
        x[ct] = (n0 & n1) << (16 * h);  // sign plane: neither fired
        y[ct] = (n0 ^ n1) << (16 * h);  // inverted non-zero plane
      }
    }
    if constexpr (!OUT2) {
      const uint32_t w = ~merge_halves(x[0], x[1]);  // collected !fire
      if (live) out[pix * 2 + h] = w;
    } else {  // [pixel][C/64 = 1][plane][half]
      const uint32_t sg = merge_halves(x[0], x[1]), nz = ~merge_halves(y[0], y[1]);
      if (live) {
        out[pix * 4 + h] = sg;
        out[pix * 4 + 2 + h] = nz;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// COMPARISON FIGURE, not the product path (BNN_MI355X_L1=lds): cnvW1A1 layer 1 written the way the north-star
// words it -- weight tile staged in LDS with coalesced loads, 64-bit XNOR + __popcll, the threshold as a compare,
// max-pool as a wavefront-shuffle reduction (lane = output pixel, the four pixels of a pooling quad on four
// consecutive lanes).  Straightforward code, no instruction-level tuning: it is here so that the bench line can
// carry the measured price of the two places where the product departs from that wording (weights from SGPRs
// through the scalar cache instead of LDS; one lane per 2x2 quad with min-before-compare instead of shuffles).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_l1_literal(const uint64_t *__restrict__ in, uint64_t *__restrict__ out,
                                                        const uint32_t *__restrict__ rows, int n_items) {
  __shared__ uint64_t w[64][9];
  __shared__ int thr[64];
  for (int i = threadIdx.x; i < 64 * 20; i += kBlock) {  // a row = {t0, t1, 9 x u64}
    const int n = i / 20, j = i - n * 20;
    const uint32_t v = rows[i];
    if (j == 0) thr[n] = (int)v;
    else if (j >= 2) reinterpret_cast<uint32_t *>(&w[n][0])[j - 2] = v;
  }
  __syncthreads();
  const int item = blockIdx.x * kBlock + threadIdx.x;            // (n_items is a multiple of 4: whole quads)
  const int it = item < n_items ? item : n_items - 1;
  const int quad = it >> 2, sub = it & 3, img = quad / 196, q = quad - img * 196, qy = q / 14, qx = q - qy * 14;
  const int oy = 2 * qy + (sub >> 1), ox = 2 * qx + (sub & 1);
  const uint64_t *__restrict__ base = in + (size_t)img * 900 + (size_t)oy * 30 + ox;
  uint64_t a[9];
#pragma unroll
  for (int ky = 0; ky < 3; ky++)
#pragma unroll
    for (int kx = 0; kx < 3; kx++) a[ky * 3 + kx] = base[ky * 30 + kx];
  uint64_t bits = 0;
  for (int n = 0; n < 64; n++) {
    int m = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) m += __popcll(w[n][j] ^ a[j]);
    bits |= (uint64_t)(m < thr[n]) << n;
  }
  bits |= __shfl_xor(bits, 1, 64);  // max-pool of a thresholded map = OR over the quad
  bits |= __shfl_xor(bits, 2, 64);
  if (sub == 0 && item < n_items) out[quad] = bits;
}

// ---------------------------------------------------------------------------
// SIDE EXPERIMENT, not the product path (BNN_MI355X_L1=mfma; DESIGN.md 5 "Pricing the rule"): cnvW1A1 layer 1
// -- 46 % of the network's time on the integer pipe -- as an implicit GEMM on the matrix cores.  The north-star
// rules the matrix pipe out for the bitwise layers; this kernel exists to put a measured number next to that
// rule, bit-exact like everything else.
//   D[neuron][pixel] = sum over 9 taps x 64 channels of w * a,  w, a in {-1, +1} as FP4 E2M1 (0x2 / 0xA): exact
//   in f32 (|D| <= 576).  v_mfma_scale_f32_32x32x64_f8f6f4: 32 neurons x 32 pixels x one tap (64 channels) per
//   instruction, 65 536 MACs in 32 SIMD-cycles -- 6.3x the 64 lanes x 32 synapses / 6.3 cycles of xor + bcnt.
// A block (256 threads) takes two images at a time: their bit-packed 30x30x64 maps are expanded once into FP4
// planes in LDS ([image][channel half h][pixel] x 16 bytes: the B operand of lane (pixel column c, h) is one
// ds_read_b128, conflict-free); the 18 KB of FP4 weights stay in 72 VGPRs of every wave.  A wave owns a PAIR of
// output rows (2r, 2r+1) x 28 columns (lanes c = 28..31 idle: 87.5 % of the tile) x all 64 neurons: 36 MFMAs,
// accumulators seeded with -(theta + 1) so that the result's sign bit is !fire; vertical max-pool = AND of the
// two rows' !fire words in the lane, horizontal = AND with lane c ^ 1 (DPP), 14 pooled pixels stored per pair.
// ---------------------------------------------------------------------------
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int kL1Pix = 912;  // pixels per LDS plane: 900 + what the idle lanes' windows overrun

// 8 bits -> 8 FP4 nibbles, bit i -> nibble i: 1 -> 0x2 (+1.0), 0 -> 0xA (-1.0)
__device__ __forceinline__ uint32_t fp4_pm1(uint32_t byte) {
  uint32_t t = ~byte & 0xFFu;                 // 1 where the activation is -1
  t = (t | (t << 12)) & 0x000F000Fu;
  t = (t | (t << 6)) & 0x03030303u;
  t = (t | (t << 3)) & 0x11111111u;
  return (t << 3) + 0x22222222u;              // 0x2 + 8 * [-1]
}

__global__ __launch_bounds__(256, 2) void k_l1_mfma(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                     const uint8_t *__restrict__ tab, int n_images) {
  __shared__ uint4 plane[2][2][kL1Pix];  // [image of the pair][h][pixel]
  __shared__ uint32_t lut[256];          // 8 channel bits -> 8 FP4 nibbles (the expansion is 4 lookups per source dword
                                         // instead of 4 x 7 integer instructions)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, h = lane >> 5;
  lut[tid] = fp4_pm1((uint32_t)tid);
  // weights: 9 taps x 2 neuron tiles, 16 bytes each, for the whole kernel
  const uint4 *__restrict__ wt = reinterpret_cast<const uint4 *>(tab);
  v8i wreg[9][2];
#pragma unroll
  for (int tap = 0; tap < 9; tap++)
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
      const uint4 v = wt[(tap * 2 + mt) * 64 + lane];
      wreg[tap][mt] = v8i{(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0};
    }
  // accumulator seeds -(theta + 1): [neuron tile][h][16] floats, read from LDS at the start of every row pair (32
  // VGPRs less than keeping them: with the prefetch registers the kernel would spill)
  __shared__ v16f seed_lds[2][2];
  if (tid < 64) reinterpret_cast<float *>(seed_lds)[tid] = reinterpret_cast<const float *>(tab + kL1MfmaWeights)[tid];
  const int cc = c < 28 ? c : 27;  // idle lanes repeat column 27 (their results are dropped)
  // The bit maps of the NEXT pair of images are requested (15 dwords per thread, into registers) before the MFMAs of
  // the current pair are issued: measured without it the block alternated ~3 us of load latency with ~3.6 us of
  // matrix work and the matrix pipe sat at 46 % (profiles/r02_sq_l1_matrix_pipe.json).
  uint32_t pre[15];
  auto fetch = [&](int pair) {
    const int img0 = pair * 2, lim = min(2, n_images - img0) * 1800;
#pragma unroll
    for (int j = 0; j < 15; j++) {
      const int d = tid + 256 * j;
      pre[j] = d < lim ? in[(size_t)img0 * 1800 + d] : 0u;
    }
  };
  if (blockIdx.x * 2 < n_images) fetch(blockIdx.x);
  for (int pair = blockIdx.x; pair * 2 < n_images; pair += gridDim.x) {
    const int img0 = pair * 2, nimg = min(2, n_images - img0);
    __syncthreads();  // the previous pair's planes are no longer read
#pragma unroll
    for (int j = 0; j < 15; j++) {  // one source dword (32 channels of a pixel) per iteration
      const int d = tid + 256 * j;
      if (d < nimg * 1800) {
        const int i = d >= 1800, e = d - i * 1800, pix = e >> 1, hh = e & 1;
        const uint32_t bits = pre[j];
        plane[i][hh][pix] = make_uint4(lut[bits & 255], lut[(bits >> 8) & 255], lut[(bits >> 16) & 255], lut[bits >> 24]);
      }
    }
    if ((pair + gridDim.x) * 2 < n_images) fetch(pair + gridDim.x);
    __syncthreads();
    for (int rp = wave; rp < nimg * 14; rp += 4) {  // row pair (2r, 2r+1) of image i
      const int i = rp / 14, r = rp - i * 14;
      const uint4 *__restrict__ P = &plane[i][h][0];
      v16f acc[2][2] = {{seed_lds[0][h], seed_lds[0][h]}, {seed_lds[1][h], seed_lds[1][h]}};  // [neuron tile][row of the pair]
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
        v8i b[4];  // input rows 2r .. 2r+3 at column c + kx
#pragma unroll
        for (int y = 0; y < 4; y++) {
          const uint4 v = P[(2 * r + y) * 30 + cc + kx];
          b[y] = v8i{(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0};
        }
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
          for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int pt = 0; pt < 2; pt++)
              acc[mt][pt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wreg[ky * 3 + kx][mt], b[ky + pt], acc[mt][pt], 4, 4, 0,
                                                                              0x7F7F7F7F, 0, 0x7F7F7F7F);
      }
      uint32_t word[2];
#pragma unroll
      for (int mt = 0; mt < 2; mt++) {
        // vertical max-pool on the accumulators: both rows' results negative (neither fires) <=> their maximum is
        // negative -- 16 v_max_f32 (FP pipe) instead of 16 more sign extractions on the integer pipe
        int v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = __float_as_int(fmaxf(acc[mt][0][k], acc[mt][1][k]));
        const uint32_t nf = or_halves(sign_nibbles(v, h));  // !fire of the row pair, 32 neurons
        const uint32_t pooled = nf & (uint32_t)__builtin_amdgcn_mov_dpp((int)nf, 0xB1, 0xF, 0xF, true);  // & lane c ^ 1
        word[mt] = ~pooled;
      }
      if (h == 0 && c < 28 && !(c & 1))
        *reinterpret_cast<uint2 *>(out + ((size_t)(img0 + i) * 196 + r * 14 + (c >> 1)) * 2) = make_uint2(word[0], word[1]);
    }
  }
}

// ---------------------------------------------------------------------------
// AR_XNOR, 1-bit out: the headline path (cnvW1A1 layers 1..7, lfcW1A1).
// Measured on MI355X (profiles/r01_microbench*.txt): v_xor_b32 and
// v_bcnt_u32_b32 both go down the integer pipe, ~4.2 cycles per wave64
// instruction per SIMD for v_bcnt and for anything with an SGPR operand; an
// alternating xor/bcnt stream sustains 6.3 cycles per 32-bit pair when every pair is followed by one
// s_nop 0 (which the compiler puts behind each of these asm statements; profiles/r01_microbench9).  That pair
// rate IS the roofline of this path, so the kernels below spend integer-pipe
// slots on nothing else:
//   * every accumulator is one chain of v_bcnt (which adds for free); the empty
//     asm pins the chain so that LLVM does not re-associate it into a tree of
//     v_add3_u32;
//   * the threshold test is a subtract whose SIGN BIT is shifted into the
//     result word with one v_alignbit_b32 (no v_cmp / v_cndmask / v_or);
//     channels are walked 31..0 so that channel c lands on bit c.
// ---------------------------------------------------------------------------
// One (v_xor, v_bcnt) pair per asm statement.  `t` is ONE scratch VGPR threaded through every statement of a
// kernel as an in/out operand: that (false) dependency keeps the statements in source order -- the pairs of
// the four accumulators strictly alternate -- and makes each statement touch a VGPR the previous one defined,
// which is exactly the condition under which the compiler puts one `s_nop 0` between them.  (op, bcnt, nop)
// issues in 6.3 cycles, the same pairs back to back in 8.0 (profiles/r01_microbench9_nop_cadence.txt); the
// built code object is checked for this stream by tools/check_cadence.py (tests/test_kernel_cadence.py).
__device__ __forceinline__ uint32_t chain_temp() {
  uint32_t t;
  asm("" : "=v"(t));  // any value: every statement overwrites it before reading it
  return t;
}
// The neuron loops that step by one: the compiler otherwise counts them in a VGPR (v_add_co + a vcc branch), one
// more VALU slot per neuron next to the pairs.  Pinning the counter to an SGPR once per iteration emits nothing.
#ifdef BNN_VGPR_COUNTER  // A/B build only
#define SCALAR_COUNTER(c) do { } while (0)
#else
#define SCALAR_COUNTER(c) asm volatile("" : "+s"(c))
#endif
__device__ __forceinline__ void xpop(int &acc, uint32_t w, uint32_t a, uint32_t &t) {
  asm("v_xor_b32 %0, %2, %3\n\tv_bcnt_u32_b32 %1, %0, %1" : "+v"(t), "+v"(acc) : "s"(w), "v"(a));
}
// four accumulators against one weight dword, strictly alternating xor / bcnt
__device__ __forceinline__ void xpop4(int &m0, int &m1, int &m2, int &m3, uint32_t w, uint32_t a0, uint32_t a1,
                                      uint32_t a2, uint32_t a3, uint32_t &t) {
  xpop(m0, w, a0, t); xpop(m1, w, a1, t); xpop(m2, w, a2, t); xpop(m3, w, a3, t);
}
// first pair of a chain: the accumulator starts at `seed` (an SGPR: -threshold, so that the chain ends on
// m - t and the compare costs neither a v_mov 0 nor a subtract)
__device__ __forceinline__ int xpop_seed(uint32_t w, uint32_t a, int seed, uint32_t &t) {
  int acc;
  asm("v_xor_b32 %0, %2, %3\n\tv_bcnt_u32_b32 %1, %0, %4" : "+v"(t), "=v"(acc) : "s"(w), "v"(a), "s"(seed));
  return acc;
}
// result of one neuron group of one pixel/vector: a whole dword (32 neurons per block, the
// throughput form) or one byte of it (8 neurons per block: four times as many, four times shorter
// blocks -- small batches, where the 32-neuron form leaves most of the chip idle)
template <int NPB>
__device__ __forceinline__ void store_group(uint32_t *__restrict__ out, size_t index, uint32_t bits) {
  if constexpr (NPB == 32) out[index] = bits;
  else reinterpret_cast<uint8_t *>(out)[index] = (uint8_t)bits;
}

template <int CW, int ID, bool POOL, int NPB = 32>
__global__ __launch_bounds__(kBlock) void k_quad_x(const uint64_t *__restrict__ in, uint32_t *__restrict__ out,
                                                    const uint32_t *__restrict__ rows, int n_items, int groups, int gpb) {
  constexpr int OD = ID - 2, QD = OD / 2, NQ = QD * QD, KW = 9 * CW, ROW_DW = 2 + 2 * KW;
  const BlockMap bm = map_block(groups / gpb, n_items);
  if (!bm.valid) return;
  const int item = bm.item;
  const int img = item / NQ, q = item - img * NQ;
  const int qy = q / QD, qx = q - qy * QD;
  const uint64_t *__restrict__ base = in + ((size_t)img * ID * ID + (size_t)(2 * qy) * ID + 2 * qx) * CW;
  uint32_t wl[4][4][CW], wh[4][4][CW];
#pragma unroll
  for (int y = 0; y < 4; y++)
#pragma unroll
    for (int x = 0; x < 4; x++)
#pragma unroll
      for (int k = 0; k < CW; k++) {
        const uint64_t v = base[(y * ID + x) * CW + k];
        wl[y][x][k] = (uint32_t)v;
        wh[y][x][k] = (uint32_t)(v >> 32);
      }
  uint32_t t = chain_temp();
  for (int cg = bm.cg * gpb, cg_end = cg + gpb; cg < cg_end; cg++) {  // see k_conv0
    kptr32 w = (kptr32)(uintptr_t)(rows + (size_t)cg * NPB * ROW_DW);
    uint32_t b[4] = {0, 0, 0, 0};
    for (int c = NPB - 1; c >= 0; c--) {
      kptr32 r = w + c * ROW_DW;
      SCALAR_COUNTER(c);
      const int nt = -(int)r[0];  // (scalar) every chain starts at -t: it ends on m - t, whose sign is the decision
      int m[2][2];
  #pragma unroll
      for (int j = 0; j < KW; j++) {
        const int ky = j / (3 * CW), kx = (j / CW) % 3, k = j % CW;
        const uint32_t w0 = r[2 + 2 * j], w1 = r[3 + 2 * j];
        if (j == 0) {
  #pragma unroll
          for (int i = 0; i < 4; i++) m[i >> 1][i & 1] = xpop_seed(w0, wl[ky + (i >> 1)][kx + (i & 1)][k], nt, t);
        } else {
          xpop4(m[0][0], m[0][1], m[1][0], m[1][1], w0, wl[ky][kx][k], wl[ky][kx + 1][k], wl[ky + 1][kx][k], wl[ky + 1][kx + 1][k], t);
        }
        xpop4(m[0][0], m[0][1], m[1][0], m[1][1], w1, wh[ky][kx][k], wh[ky][kx + 1][k], wh[ky + 1][kx][k], wh[ky + 1][kx + 1][k], t);
      }
      if constexpr (POOL) {  // OR of the four fire bits == (min (m - t)) < 0
        b[0] = shift_in_sign(b[0], min(min(m[0][0], m[0][1]), min(m[1][0], m[1][1])));
      } else {
  #pragma unroll
        for (int i = 0; i < 4; i++) b[i] = shift_in_sign(b[i], m[i >> 1][i & 1]);
      }
    }
    if constexpr (POOL) {
      store_group<NPB>(out, (size_t)item * groups + cg, b[0]);
    } else {
  #pragma unroll
      for (int i = 0; i < 4; i++) {
        const size_t pix = (size_t)img * OD * OD + (size_t)(2 * qy + (i >> 1)) * OD + 2 * qx + (i & 1);
        store_group<NPB>(out, pix * groups + cg, b[i]);
      }
    }
  }
}

// one lane = one vector of KW words (FC layers, CNV layer 5; SINGLE: CNV layer 4 window gather).
// Two neurons per iteration: two independent v_bcnt chains per lane.
// POOL (with SINGLE): lane = one output pixel of a 2x2-pooled conv layer, the four pixels of a pooling
// quad on four consecutive lanes (item = 4 * quad + 2 dy + dx); the pooled bits are the OR over those
// lanes.  Small batches only: four times the lanes of k_quad_x, a quarter of the work per lane.
template <int KW, bool SINGLE, int CW, int ID, int NPB = 32, bool POOL = false>
__global__ __launch_bounds__(kBlock) void k_vec_x(const uint64_t *__restrict__ in, uint32_t *__restrict__ out,
                                                   const uint32_t *__restrict__ rows, int n_items, int groups, int gpb) {
  constexpr int ROW_DW = 2 + 2 * KW;
  static_assert(!POOL || SINGLE, "pooling needs the window form");
  const BlockMap bm = map_block(groups / gpb, n_items);
  if (!bm.valid) return;  // (POOL: n_items is a multiple of 4, so a quad is valid or invalid as a whole)
  const int item = bm.item;
  uint32_t al[KW], ah[KW];
  if constexpr (SINGLE) {
    constexpr int OD = ID - 2;
    static_assert(KW == 9 * CW, "window size");
    int img, oy, ox;
    if constexpr (POOL) {
      constexpr int QD = OD / 2, NQ = QD * QD;
      const int quad = item >> 2, sub = item & 3;
      img = quad / NQ;
      const int q = quad - img * NQ, qy = q / QD, qx = q - qy * QD;
      oy = 2 * qy + (sub >> 1);
      ox = 2 * qx + (sub & 1);
    } else {
      img = item / (OD * OD);
      const int p = item - img * (OD * OD);
      oy = p / OD;
      ox = p - oy * OD;
    }
    const uint64_t *__restrict__ base = in + ((size_t)img * ID * ID + (size_t)oy * ID + ox) * CW;
#pragma unroll
    for (int ky = 0; ky < 3; ky++)
#pragma unroll
      for (int kx = 0; kx < 3; kx++)
#pragma unroll
        for (int k = 0; k < CW; k++) {
          const uint64_t v = base[(ky * ID + kx) * CW + k];
          al[(ky * 3 + kx) * CW + k] = (uint32_t)v;
          ah[(ky * 3 + kx) * CW + k] = (uint32_t)(v >> 32);
        }
  } else {
    const uint64_t *__restrict__ base = in + (size_t)item * KW;
#pragma unroll
    for (int k = 0; k < KW; k++) {
      const uint64_t v = base[k];
      al[k] = (uint32_t)v;
      ah[k] = (uint32_t)(v >> 32);
    }
  }
  uint32_t t = chain_temp();
  for (int cg = bm.cg * gpb, cg_end = cg + gpb; cg < cg_end; cg++) {  // see k_conv0
    kptr32 w = (kptr32)(uintptr_t)(rows + (size_t)cg * NPB * ROW_DW);
    uint32_t b = 0;
    for (int c = NPB - 1; c >= 0; c -= 2) {
      kptr32 r1 = w + c * ROW_DW, r0 = r1 - ROW_DW;
      int m1 = xpop_seed(r1[2], al[0], -(int)r1[0], t), m0 = xpop_seed(r0[2], al[0], -(int)r0[0], t);
      xpop(m1, r1[3], ah[0], t);
      xpop(m0, r0[3], ah[0], t);
  #pragma unroll
      for (int k = 1; k < KW; k++) {
        xpop(m1, r1[2 + 2 * k], al[k], t);
        xpop(m0, r0[2 + 2 * k], al[k], t);
        xpop(m1, r1[3 + 2 * k], ah[k], t);
        xpop(m0, r0[3 + 2 * k], ah[k], t);
      }
      b = shift_in_sign(b, m1);
      b = shift_in_sign(b, m0);
    }
    if constexpr (POOL) {
      b |= __shfl_xor(b, 1, 64);
      b |= __shfl_xor(b, 2, 64);
      if ((item & 3) == 0) store_group<NPB>(out, (size_t)(item >> 2) * groups + cg, b);
    } else {
      store_group<NPB>(out, (size_t)item * groups + cg, b);
    }
  }
}


// ---------------------------------------------------------------------------
// Generic stages: the 2-bit-activation networks (cnvW1A2, cnvW2A2, lfcW1A2) and
// the one XNOR stage with a 2-bit output (lfcW1A2 layer 0).  Same structure and
// the same integer-pipe discipline as the XNOR kernels above; per 32 synapses
//   AR_TB  v_bitop3 (za & (sa ^ w)) + v_bcnt                       2 slots, like XNOR
//   AR_TT  v_and (za & zw) + v_bcnt + v_bitop3 (.. & (sa ^ sw)) + v_bcnt   4 slots
// and the activation is decided on the sign of
//   g0 = q + c0,  g1 = g0 + (t1 - t0),   fire_i  <=>  g_i < 0
//   AR_XNOR q = m            c0 = -t0        (m < t)
//   AR_TB   q = 2m - nz(a)   c0 = t0         (t < nz - 2m)
//   AR_TT   q = 2m - z       c0 = t0         (t < z - 2m)
// Max-pool = min of q over the quad (thresholding is monotone).
// ---------------------------------------------------------------------------
// truth table index = src0<<2 | src1<<1 | src2; f = src2 & (src0 ^ src1) -> rows 3 and 5
#define BNN_BITOP_AND_XOR "0x28"

// one 32-bit half of a weight word against one 32-bit half of an activation word.
// wq: this word's weight dwords {lo, hi} (XNOR, TB) or {sign lo, sign hi, nz lo, nz hi} (TT)
// FIRST: the first statement of a chain -- the accumulator is WRITTEN, not read: v_bcnt's addend is the seed
// (XNOR: -t0 from an SGPR, so that the chain ends on m - t0; the ternary forms: the inline constant 0), which
// saves the v_mov 0 per accumulator and neuron that an initialised C variable costs.
// Every (logic op, v_bcnt) pair is its own asm statement: the s_nop 0 that ends up behind each of them is what
// lets the SIMD alternate between waves at the right cadence -- (op, bcnt, s_nop 0) issues in 6.3 cycles, the
// same pairs back to back in 8.0 (profiles/r01_microbench9_nop_cadence.txt).  The compiler puts that nop behind
// an asm statement whenever the next instruction touches one of its VGPR results: the scratch register `t`
// (chain_temp) is threaded through all statements for that reason, and to pin their order.
// tools/check_cadence.py verifies the result in the built code object (tests/test_kernel_cadence.py): a
// toolchain that changes the habit fails the build check instead of silently costing 20 %.
template <int ARITH, bool FIRST = false>
__device__ __forceinline__ void mac32(int &m, int &z, uint32_t as, uint32_t az, kptr32 wq, int half, uint32_t &t, int seed = 0) {
  if constexpr (ARITH == AR_XNOR) {
    if constexpr (FIRST) asm("v_xor_b32 %0, %2, %3\n\tv_bcnt_u32_b32 %1, %0, %4" : "+v"(t), "=v"(m) : "s"(wq[half]), "v"(as), "s"(seed));
    else asm("v_xor_b32 %0, %2, %3\n\tv_bcnt_u32_b32 %1, %0, %1" : "+v"(t), "+v"(m) : "s"(wq[half]), "v"(as));
  } else if constexpr (ARITH == AR_TB) {
    if constexpr (FIRST)
      asm("v_bitop3_b32 %0, %2, %3, %4 bitop3:" BNN_BITOP_AND_XOR "\n\tv_bcnt_u32_b32 %1, %0, 0"
          : "+v"(t), "=v"(m)
          : "s"(wq[half]), "v"(as), "v"(az));
    else
      asm("v_bitop3_b32 %0, %2, %3, %4 bitop3:" BNN_BITOP_AND_XOR "\n\tv_bcnt_u32_b32 %1, %0, %1"
          : "+v"(t), "+v"(m)
          : "s"(wq[half]), "v"(as), "v"(az));
  } else {
    // z += popc(za & zw), then m += popc((za & zw) & (sa ^ sw)): the second statement works on t in place
    if constexpr (FIRST) {
      asm("v_and_b32 %0, %2, %3\n\tv_bcnt_u32_b32 %1, %0, 0" : "+v"(t), "=v"(z) : "s"(wq[2 + half]), "v"(az));
      asm("v_bitop3_b32 %0, %2, %3, %0 bitop3:" BNN_BITOP_AND_XOR "\n\tv_bcnt_u32_b32 %1, %0, 0" : "+v"(t), "=v"(m) : "s"(wq[half]), "v"(as));
    } else {
      asm("v_and_b32 %0, %2, %3\n\tv_bcnt_u32_b32 %1, %0, %1" : "+v"(t), "+v"(z) : "s"(wq[2 + half]), "v"(az));
      asm("v_bitop3_b32 %0, %2, %3, %0 bitop3:" BNN_BITOP_AND_XOR "\n\tv_bcnt_u32_b32 %1, %0, %1" : "+v"(t), "+v"(m) : "s"(wq[half]), "v"(as));
    }
  }
}

// The quantity the thresholds are compared with, d = sum of w*a over the window (the reference's accumulator):
//   AR_XNOR: the chain was seeded with -t0, so m IS the decision value g0 = m - t0 (fire0 <=> g0 < 0)
//   AR_TB:   d = nz - 2m     AR_TT:   d = z - 2m        one v_mad_i32_i24 (m * -2 + zz), fire_i <=> t_i - d < 0
template <int ARITH>
__device__ __forceinline__ int d_of(int m, int zz) {
  int d;  // (written out: LLVM would turn m * -2 + zz into a shift and a subtract, two issue slots)
  asm("v_mad_i32_i24 %0, %1, -2, %2" : "=v"(d) : "v"(m), "v"(zz));
  return d;
}

// shift the decision(s) into the result word(s).  1-bit out: fire0.  2-bit out: b0 collects fire0
// and b1 fire1, one v_alignbit each; the planes (sign = neither fired, non-zero = both or neither)
// are formed from the two words once per group of neurons (finish_bits), not per neuron.
template <bool OUT2>
__device__ __forceinline__ void decide(uint32_t &b0, uint32_t &b1, int g0, int dt) {
  b0 = shift_in_sign(b0, g0);
  if constexpr (OUT2) b1 = shift_in_sign(b1, g0 + dt);
}
template <bool OUT2>
__device__ __forceinline__ void finish_bits(uint32_t &b0, uint32_t &b1) {
  if constexpr (OUT2) {
    const uint32_t f0 = b0, f1 = b1;
    b0 = ~(f0 | f1);
    b1 = ~(f0 ^ f1);
  }
}

// 3x3 valid conv, one lane = a 2x2 quad of output pixels (4x4 window in VGPRs), optional pool.
// Replaces ConvolutionInputGenerator + Matrix_Vector_Activate_Batch + ThresholdsActivation
// (+ StreamingMaxPool_Precision_Batch) for CNV layers 1..3 of the A2 networks.
template <int ARITH, int CW, int ID, bool POOL, bool OUT2, int NPB = 32, bool TWO = false>
__global__ __launch_bounds__(kBlock) void k_quad(const uint64_t *__restrict__ in, uint32_t *__restrict__ out,
                                                  const uint32_t *__restrict__ rows, int n_items, int groups, int gpb) {
  constexpr int OD = ID - 2, QD = OD / 2, NQ = QD * QD, PL = planes_in<ARITH>(), WPL = wplanes<ARITH>();
  constexpr int KW = 9 * CW, ROW_DW = row_dw<ARITH, KW>(), ZW = (PL == 2) ? CW : 1;
  const BlockMap bm = map_block(groups / gpb, n_items);
  if (!bm.valid) return;
  const int item = bm.item;
  const int img = item / NQ, q = item - img * NQ;
  const int qy = q / QD, qx = q - qy * QD;
  const uint64_t *__restrict__ base = in + ((size_t)img * ID * ID + (size_t)(2 * qy) * ID + 2 * qx) * CW * PL;
  uint32_t ws[4][4][CW][2], wz[4][4][ZW][2];
#pragma unroll
  for (int y = 0; y < 4; y++)
#pragma unroll
    for (int x = 0; x < 4; x++)
#pragma unroll
      for (int k = 0; k < CW; k++) {
        const uint64_t v = base[((y * ID + x) * CW + k) * PL];
        ws[y][x][k][0] = (uint32_t)v; ws[y][x][k][1] = (uint32_t)(v >> 32);
        if constexpr (PL == 2) {
          const uint64_t u = base[((y * ID + x) * CW + k) * PL + 1];
          wz[y][x][k][0] = (uint32_t)u; wz[y][x][k][1] = (uint32_t)(u >> 32);
        }
      }
  // AR_TB: the number of non-zero activations in each of the 4 windows (weight independent)
  int nn[2][2] = {{0, 0}, {0, 0}};
  if constexpr (ARITH == AR_TB) {
#pragma unroll
    for (int dy = 0; dy < 2; dy++)
#pragma unroll
      for (int dx = 0; dx < 2; dx++)
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
          for (int kx = 0; kx < 3; kx++)
#pragma unroll
            for (int k = 0; k < CW; k++)
              nn[dy][dx] += __builtin_popcount(wz[dy + ky][dx + kx][k][0]) + __builtin_popcount(wz[dy + ky][dx + kx][k][1]);
  }
  uint32_t t = chain_temp();
  for (int cg = bm.cg * gpb, cg_end = cg + gpb; cg < cg_end; cg++) {  // see k_conv0
    kptr32 w = (kptr32)(uintptr_t)(rows + (size_t)cg * NPB * ROW_DW);
    uint32_t b0[4] = {0, 0, 0, 0}, b1[4] = {0, 0, 0, 0};
    for (int c = NPB - 1; c >= 0; c--) {
      kptr32 r = w + c * ROW_DW;
      SCALAR_COUNTER(c);
      const int t0 = (int)r[0], t1 = (int)r[1];
      // (the zeros are never materialised: the first statement of each chain writes its accumulator)
      int m[2][2] = {{0, 0}, {0, 0}}, z[2][2] = {{0, 0}, {0, 0}};
  #pragma unroll
      for (int j = 0; j < KW; j++) {
        const int ky = j / (3 * CW), kx = (j / CW) % 3, k = j % CW;
        kptr32 wq = r + 2 + 2 * WPL * j;
  #pragma unroll
        for (int h = 0; h < 2; h++)
  #pragma unroll
          for (int dy = 0; dy < 2; dy++)
  #pragma unroll
            for (int dx = 0; dx < 2; dx++) {
              const uint32_t as = ws[dy + ky][dx + kx][k][h], az = wz[dy + ky][dx + kx][PL == 2 ? k : 0][h];
              if (j == 0 && h == 0) mac32<ARITH, true>(m[dy][dx], z[dy][dx], as, az, wq, h, t, -t0);
              else mac32<ARITH>(m[dy][dx], z[dy][dx], as, az, wq, h, t);
            }
      }
      if constexpr (TWO) {
        if (r[2 + 6 * KW]) {  // this neuron has weights of -2 (fault injection only)
  #pragma unroll
          for (int j = 0; j < KW; j++) {
            const int ky = j / (3 * CW), kx = (j / CW) % 3, k = j % CW;
  #pragma unroll
            for (int h = 0; h < 2; h++) {
              const uint32_t two = r[2 + 4 * KW + 2 * j + h];
  #pragma unroll
              for (int dy = 0; dy < 2; dy++)
  #pragma unroll
                for (int dx = 0; dx < 2; dx++) z[dy][dx] += two_extra(two, ws[dy + ky][dx + kx][k][h], wz[dy + ky][dx + kx][k][h]);
            }
          }
        }
      }
      // decision values: g0 < 0 <=> the first threshold fires, g1 = g0 + dt likewise for the second
      if constexpr (ARITH == AR_XNOR) {  // chains seeded with -t0: m is g0 already; max-pool = min over the quad
        if constexpr (POOL) {
          decide<OUT2>(b0[0], b1[0], min(min(m[0][0], m[0][1]), min(m[1][0], m[1][1])), t0 - t1);
        } else {
  #pragma unroll
          for (int i = 0; i < 4; i++) decide<OUT2>(b0[i], b1[i], m[i >> 1][i & 1], t0 - t1);
        }
      } else {  // d = sum of w*a; fire_i <=> t_i - d < 0; max-pool = max of d over the quad
        int d[4];
  #pragma unroll
        for (int i = 0; i < 4; i++) d[i] = d_of<ARITH>(m[i >> 1][i & 1], ARITH == AR_TB ? nn[i >> 1][i & 1] : z[i >> 1][i & 1]);
        if constexpr (POOL) {
          decide<OUT2>(b0[0], b1[0], t0 - max(max(d[0], d[1]), max(d[2], d[3])), t1 - t0);
        } else {
  #pragma unroll
          for (int i = 0; i < 4; i++) decide<OUT2>(b0[i], b1[i], t0 - d[i], t1 - t0);
        }
      }
    }
    if constexpr (POOL) {
      finish_bits<OUT2>(b0[0], b1[0]);
      store_bits<OUT2, NPB>(out, (size_t)item, groups, cg, b0[0], b1[0]);
    } else {
  #pragma unroll
      for (int i = 0; i < 4; i++) {
        const size_t pix = (size_t)img * OD * OD + (size_t)(2 * qy + (i >> 1)) * OD + 2 * qx + (i & 1);
        finish_bits<OUT2>(b0[i], b1[i]);
        store_bits<OUT2, NPB>(out, pix, groups, cg, b0[i], b1[i]);
      }
    }
  }
}

// Generic "KW words in, thresholded bits out": one lane = one vector (FC layers, CNV layer 5,
// and with SINGLE the 3x3 window gather of CNV layer 4).  Two neurons per iteration.
// POOL: as in k_vec_x (lane = output pixel, quad on four consecutive lanes, OR of the fire words).
template <int ARITH, int KW, bool OUT2, bool SINGLE, int CW, int ID, int NPB = 32, bool POOL = false, bool TWO = false>
__global__ __launch_bounds__(kBlock) void k_vec(const uint64_t *__restrict__ in, uint32_t *__restrict__ out,
                                                 const uint32_t *__restrict__ rows, int n_items, int groups, int gpb) {
  constexpr int PL = planes_in<ARITH>(), WPL = wplanes<ARITH>();
  constexpr int ROW_DW = row_dw<ARITH, KW>(), ZW = (PL == 2) ? KW : 1;
  static_assert(!POOL || SINGLE, "pooling needs the window form");
  const BlockMap bm = map_block(groups / gpb, n_items);
  if (!bm.valid) return;
  const int item = bm.item;
  uint32_t as[KW][2], az[ZW][2];
  auto load_word = [&](int dst, const uint64_t *__restrict__ src) {
    const uint64_t v = src[0];
    as[dst][0] = (uint32_t)v; as[dst][1] = (uint32_t)(v >> 32);
    if constexpr (PL == 2) {
      const uint64_t u = src[1];
      az[dst][0] = (uint32_t)u; az[dst][1] = (uint32_t)(u >> 32);
    }
  };
  if constexpr (SINGLE) {
    constexpr int OD = ID - 2;
    static_assert(KW == 9 * CW, "window size");
    int img, oy, ox;
    if constexpr (POOL) {
      constexpr int QD = OD / 2, NQ = QD * QD;
      const int quad = item >> 2, sub = item & 3;
      img = quad / NQ;
      const int q = quad - img * NQ, qy = q / QD, qx = q - qy * QD;
      oy = 2 * qy + (sub >> 1);
      ox = 2 * qx + (sub & 1);
    } else {
      img = item / (OD * OD);
      const int p = item - img * (OD * OD);
      oy = p / OD;
      ox = p - oy * OD;
    }
    const uint64_t *__restrict__ base = in + ((size_t)img * ID * ID + (size_t)oy * ID + ox) * CW * PL;
#pragma unroll
    for (int ky = 0; ky < 3; ky++)
#pragma unroll
      for (int kx = 0; kx < 3; kx++)
#pragma unroll
        for (int k = 0; k < CW; k++) load_word((ky * 3 + kx) * CW + k, base + ((ky * ID + kx) * CW + k) * PL);
  } else {
    const uint64_t *__restrict__ base = in + (size_t)item * KW * PL;
#pragma unroll
    for (int k = 0; k < KW; k++) load_word(k, base + k * PL);
  }
  int nn = 0;  // AR_TB: non-zero activations of the vector
  if constexpr (ARITH == AR_TB) {
#pragma unroll
    for (int k = 0; k < KW; k++) nn += __builtin_popcount(az[k][0]) + __builtin_popcount(az[k][1]);
  }
  uint32_t t = chain_temp();
  for (int cg = bm.cg * gpb, cg_end = cg + gpb; cg < cg_end; cg++) {  // see k_conv0
    kptr32 w = (kptr32)(uintptr_t)(rows + (size_t)cg * NPB * ROW_DW);
    uint32_t b0 = 0, b1 = 0;
    for (int c = NPB - 1; c >= 0; c -= 2) {
      kptr32 rA = w + c * ROW_DW, rB = rA - ROW_DW;
      const int tA0 = (int)rA[0], tA1 = (int)rA[1], tB0 = (int)rB[0], tB1 = (int)rB[1];
      int mA = 0, zA = 0, mB = 0, zB = 0;  // (never materialised: the first statement of a chain writes them)
  #pragma unroll
      for (int k = 0; k < KW; k++)
  #pragma unroll
        for (int h = 0; h < 2; h++) {
          const uint32_t s_ = as[k][h], z_ = az[PL == 2 ? k : 0][h];
          if (k == 0 && h == 0) {
            mac32<ARITH, true>(mA, zA, s_, z_, rA + 2 + 2 * WPL * k, h, t, -tA0);
            mac32<ARITH, true>(mB, zB, s_, z_, rB + 2 + 2 * WPL * k, h, t, -tB0);
          } else {
            mac32<ARITH>(mA, zA, s_, z_, rA + 2 + 2 * WPL * k, h, t);
            mac32<ARITH>(mB, zB, s_, z_, rB + 2 + 2 * WPL * k, h, t);
          }
        }
      if constexpr (TWO) {
        if (rA[2 + 6 * KW] | rB[2 + 6 * KW]) {  // weights of -2 in one of the two rows (fault injection only)
  #pragma unroll
          for (int k = 0; k < KW; k++)
  #pragma unroll
            for (int h = 0; h < 2; h++) {
              zA += two_extra(rA[2 + 4 * KW + 2 * k + h], as[k][h], az[k][h]);
              zB += two_extra(rB[2 + 4 * KW + 2 * k + h], as[k][h], az[k][h]);
            }
        }
      }
      if constexpr (ARITH == AR_XNOR) {  // seeded with -t0: the chain's end is the decision value
        decide<OUT2>(b0, b1, mA, tA0 - tA1);
        decide<OUT2>(b0, b1, mB, tB0 - tB1);
      } else {
        decide<OUT2>(b0, b1, tA0 - d_of<ARITH>(mA, ARITH == AR_TB ? nn : zA), tA1 - tA0);
        decide<OUT2>(b0, b1, tB0 - d_of<ARITH>(mB, ARITH == AR_TB ? nn : zB), tB1 - tB0);
      }
    }
    if constexpr (POOL) {  // max-pool of a thresholded map = OR of each threshold's fire bits over the quad
      b0 |= __shfl_xor(b0, 1, 64);
      b0 |= __shfl_xor(b0, 2, 64);
      if constexpr (OUT2) {
        b1 |= __shfl_xor(b1, 1, 64);
        b1 |= __shfl_xor(b1, 2, 64);
      }
    }
    finish_bits<OUT2>(b0, b1);
    if (!POOL || (item & 3) == 0) store_bits<OUT2, NPB>(out, (size_t)(POOL ? item >> 2 : item), groups, cg, b0, b1);
  }
}

// ---------------------------------------------------------------------------
// CNV layer 8: 512 -> 64 raw accumulators (PassThroughActivation<ap_uint<16>>,
// top.cpp:232-235) + the batched class decode of
// testPrebuiltCIFAR10_multiple_images (foldedmv-offload.h:396-408): first
// strict maximum over the first number_class scores, floored at 0.
// AR_XNOR score = popcount of matches = MW - m; ternary nets: the signed sum.
// ---------------------------------------------------------------------------
template <int ARITH, int KW, bool TWO = false>
__global__ __launch_bounds__(kBlock) void k_fclast(const uint64_t *__restrict__ in, int16_t *__restrict__ scores,
                                                    int32_t *__restrict__ classes, const uint32_t *__restrict__ rows,
                                                    int n_items, int number_class) {
  constexpr int PL = planes_in<ARITH>(), WPL = wplanes<ARITH>();
  constexpr int ROW_DW = row_dw<ARITH, KW>();
  const int item = blockIdx.x * kBlock + threadIdx.x;
  if (item >= n_items) return;
  uint64_t as[KW], az[PL == 2 ? KW : 1];
  const uint64_t *__restrict__ base = in + (size_t)item * KW * PL;
#pragma unroll
  for (int k = 0; k < KW; k++) {
    as[k] = base[k * PL];
    if constexpr (PL == 2) az[k] = base[k * PL + 1];
  }
  int nzt = 0;
  if constexpr (ARITH == AR_TB) {
#pragma unroll
    for (int k = 0; k < KW; k++) nzt += pc64(az[k]);
  }
  kptr32 w = (kptr32)(uintptr_t)rows;
  uint32_t *__restrict__ s32 = reinterpret_cast<uint32_t *>(scores) + (size_t)item * 32;
  int best = 0, bestv = 0;
  uint32_t lo = 0;
  for (int n = 0; n < 64; n++) {
    kptr64 rw = (kptr64)(w + n * ROW_DW + 2);
    int m = 0, z = 0;
#pragma unroll
    for (int k = 0; k < KW; k++) mac<ARITH>(m, z, as[k], az[PL == 2 ? k : 0], rw + k * WPL);
    if constexpr (TWO) {
      if (w[n * ROW_DW + 2 + 6 * KW]) {  // weights of -2 in this row (fault injection only)
#pragma unroll
        for (int k = 0; k < KW; k++) z += two_extra64(rw[2 * KW + k], as[k], az[k]);
      }
    }
    int s = (ARITH == AR_XNOR) ? (KW * 64 - m) : finish<ARITH>(m, z, nzt);
    s = (int)(int16_t)s;  // 16-bit output word read back as ap_int<16>
    if (n < number_class && s > bestv) { bestv = s; best = n; }
    if (scores) {
      if (n & 1) s32[n >> 1] = lo | ((uint32_t)(uint16_t)s << 16);
      else lo = (uint32_t)(uint16_t)s;
    }
  }
  if (classes) classes[item] = best;
}

// The same layer for small batches: one WAVE per image, lane = neuron (the layer has exactly 64).  With a
// lane per image a single image is a serial chain of 64 neurons x 8 words on one lane (13 us); here it is
// 8 words per lane and a wave reduction for the decode.  key = score * 64 + (63 - neuron) orders by score,
// then by lowest index; scores <= 0 never beat the initial (class 0, value 0) of the reference's loop.
template <int ARITH, int KW, bool TWO = false>
__global__ __launch_bounds__(kBlock) void k_fclast_wave(const uint64_t *__restrict__ in, int16_t *__restrict__ scores,
                                                         int32_t *__restrict__ classes, const uint32_t *__restrict__ rows,
                                                         int n_items, int number_class) {
  constexpr int PL = planes_in<ARITH>();
  constexpr int ROW_DW = row_dw<ARITH, KW>();
  const int lane = threadIdx.x & 63, item = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (item >= n_items) return;  // wave-uniform
  const uint64_t *__restrict__ base = in + (size_t)item * KW * PL;
  const uint64_t *__restrict__ rw = reinterpret_cast<const uint64_t *>(rows + (size_t)lane * ROW_DW + 2);
  int m = 0, z = 0, nzt = 0;
#pragma unroll
  for (int k = 0; k < KW; k++) {
    const uint64_t as = base[k * PL], az = (PL == 2) ? base[k * PL + 1] : 0;
    if constexpr (ARITH == AR_XNOR) {
      m += pc64(rw[k] ^ as);
    } else if constexpr (ARITH == AR_TB) {
      m += pc64(az & (as ^ rw[k]));
      nzt += pc64(az);
    } else {
      const uint64_t zz = az & rw[k * 2 + 1];
      z += pc64(zz);
      m += pc64(zz & (as ^ rw[k * 2]));
    }
  }
  if constexpr (TWO) {
    if (rows[(size_t)lane * ROW_DW + 2 + 6 * KW]) {  // per lane here: each lane is its own neuron
#pragma unroll
      for (int k = 0; k < KW; k++) z += two_extra64(rw[2 * KW + k], base[k * PL], base[k * PL + 1]);
    }
  }
  int sc = (ARITH == AR_XNOR) ? (KW * 64 - m) : finish<ARITH>(m, z, nzt);
  sc = (int)(int16_t)sc;  // 16-bit output word read back as ap_int<16>
  if (scores) scores[(size_t)item * 64 + lane] = (int16_t)sc;
  if (classes) {
    int key = (lane < number_class && sc > 0) ? sc * 64 + (63 - lane) : -1;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) key = max(key, __shfl_xor(key, off, 64));
    if (lane == 0) classes[item] = key < 0 ? 0 : 63 - (key & 63);
  }
}

// cnvW1A1, small batches: layers 4..8 in ONE launch, a 512-thread block per image, thread = neuron
// (the form of k_lfc_fused).  From the 5x5x128 map on, one image is too little work for a launch per
// layer: five launches cost ~18 us of a 40 us single-image run, this kernel ~4 us.  Activations live
// in LDS (400 B in, 288 / 32 / 64 / 64 B between the layers), a wave's ballot is the next layer's
// input word, weight rows come from L2 per thread.
template <int KW>
__device__ __forceinline__ int xnor_row(const uint32_t *__restrict__ rows, int n, const uint64_t *act, int &thr) {
  constexpr int ROW_DW = 2 + 2 * KW;
  const uint32_t *__restrict__ r = rows + (size_t)n * ROW_DW;
  const uint64_t *__restrict__ w = reinterpret_cast<const uint64_t *>(r + 2);
  thr = (int)r[0];
  int m = 0;
#pragma unroll
  for (int k = 0; k < KW; k++) m += pc64(w[k] ^ act[k]);
  return m;
}

// (Requesting all five layers' rows up front, as k_lfc_fused does for the next layer, was measured: no
// gain for one image, slower from 512 images on -- 169 VGPRs instead of 60.)
__global__ __launch_bounds__(512) void k_cnv_tail(const uint64_t *__restrict__ in, int16_t *__restrict__ scores,
                                                   int32_t *__restrict__ classes, const uint32_t *__restrict__ r4,
                                                   const uint32_t *__restrict__ r5, const uint32_t *__restrict__ r6,
                                                   const uint32_t *__restrict__ r7, const uint32_t *__restrict__ r8, int number_class,
                                                   unsigned *done, unsigned done_seq) {
  __shared__ uint64_t x3[50], x4[36], x5[4], x6[8], x7[8];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, img = blockIdx.x;
  if (t < 50) x3[t] = in[(size_t)img * 50 + t];
  __syncthreads();
  {  // layer 4: 3x3 conv 5x5x128 -> 3x3x256; neuron = t & 255, the two halves of the block share the 9 pixels
    constexpr int ROW_DW = 2 + 2 * 18;
    const int n = t & 255;
    const uint32_t *__restrict__ r = r4 + (size_t)n * ROW_DW;
    const uint64_t *__restrict__ wp = reinterpret_cast<const uint64_t *>(r + 2);
    const int thr = (int)r[0];
    uint64_t w[18];
#pragma unroll
    for (int k = 0; k < 18; k++) w[k] = wp[k];
    for (int p = (t >> 8); p < 9; p += 2) {  // wave-uniform: waves 0..3 take the even pixels, 4..7 the odd ones
      const int oy = p / 3, ox = p - oy * 3;
      int m = 0;
#pragma unroll
      for (int ky = 0; ky < 3; ky++)
#pragma unroll
        for (int kx = 0; kx < 3; kx++)
#pragma unroll
          for (int k = 0; k < 2; k++) m += pc64(w[(ky * 3 + kx) * 2 + k] ^ x3[((oy + ky) * 5 + ox + kx) * 2 + k]);
      const uint64_t word = __ballot(m < thr);
      if (lane == 0) x4[p * 4 + (wave & 3)] = word;
    }
  }
  __syncthreads();
  if (t < 256) {  // layer 5: the 3x3x256 map is one window: 36 words -> 256 neurons
    int thr;
    const int m = xnor_row<36>(r5, t, x4, thr);
    const uint64_t word = __ballot(m < thr);
    if (lane == 0) x5[wave] = word;
  }
  __syncthreads();
  {  // layer 6: 256 -> 512
    int thr;
    const int m = xnor_row<4>(r6, t, x5, thr);
    const uint64_t word = __ballot(m < thr);
    if (lane == 0) x6[wave] = word;
  }
  __syncthreads();
  {  // layer 7: 512 -> 512
    int thr;
    const int m = xnor_row<8>(r7, t, x6, thr);
    const uint64_t word = __ballot(m < thr);
    if (lane == 0) x7[wave] = word;
  }
  __syncthreads();
  if (wave == 0) {  // layer 8: 512 -> 64 raw scores (popcount of matches) + the batched decode, as k_fclast_wave
    int thr;
    const int m = xnor_row<8>(r8, lane, x7, thr);
    const int sc = (int)(int16_t)(8 * 64 - m);
    if (scores) scores[(size_t)img * 64 + lane] = (int16_t)sc;
    if (classes) {
      int key = (lane < number_class && sc > 0) ? sc * 64 + (63 - lane) : -1;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) key = max(key, __shfl_xor(key, off, 64));
      if (lane == 0) classes[img] = key < 0 ? 0 : 63 - (key & 63);
    }
    mark_done(done, done_seq, lane);
  }
}

// (Not used while cnvW2A2 holds weights of -2: those runs take the staged, -2-aware kernels.)
// The same for the 2-bit nets (cnvW1A2: AR_TB, cnvW2A2: AR_TT): activations are (sign, non-zero) plane
// pairs, a layer's output planes come from the two ballots of a wave (fire_i <=> q + t_i < 0 with
// q = 2m - nz resp. 2m - z, see k_quad), layer 8 yields the signed sums.
template <int ARITH, int KW>
__device__ __forceinline__ int ternary_q(const uint32_t *__restrict__ rows, int n, const uint64_t *sa, const uint64_t *za, int &t0, int &t1) {
  constexpr int ROW_DW = row_dw<ARITH, KW>();
  const uint32_t *__restrict__ r = rows + (size_t)n * ROW_DW;
  const uint64_t *__restrict__ w = reinterpret_cast<const uint64_t *>(r + 2);
  t0 = (int)r[0];
  t1 = (int)r[1];
  int m = 0, z = 0;
#pragma unroll
  for (int k = 0; k < KW; k++) {
    if constexpr (ARITH == AR_TB) {
      m += pc64(za[k] & (sa[k] ^ w[k]));
      z += pc64(za[k]);
    } else {
      const uint64_t zz = za[k] & w[2 * k + 1];
      z += pc64(zz);
      m += pc64(zz & (sa[k] ^ w[2 * k]));
    }
  }
  return 2 * m - z;
}

template <int ARITH>
__global__ __launch_bounds__(512) void k_cnv_tail_a2(const uint64_t *__restrict__ in, int16_t *__restrict__ scores,
                                                      int32_t *__restrict__ classes, const uint32_t *__restrict__ r4,
                                                      const uint32_t *__restrict__ r5, const uint32_t *__restrict__ r6,
                                                      const uint32_t *__restrict__ r7, const uint32_t *__restrict__ r8, int number_class,
                                                      unsigned *done, unsigned done_seq) {
  // [plane 0 = sign, 1 = non-zero][word]
  __shared__ uint64_t x3[2][50], x4[2][36], x5[2][4], x6[2][8], x7[2][8];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, img = blockIdx.x;
  if (t < 100) x3[t & 1][t >> 1] = in[(size_t)img * 100 + t];  // stored [pixel][word][plane]
  __syncthreads();
  {  // layer 4
    constexpr int ROW_DW = row_dw<ARITH, 18>();
    const int n = t & 255;
    const uint32_t *__restrict__ r = r4 + (size_t)n * ROW_DW;
    const uint64_t *__restrict__ w = reinterpret_cast<const uint64_t *>(r + 2);
    const int t0 = (int)r[0], t1 = (int)r[1];
    for (int p = (t >> 8); p < 9; p += 2) {
      const int oy = p / 3, ox = p - oy * 3;
      int m = 0, z = 0;
#pragma unroll
      for (int ky = 0; ky < 3; ky++)
#pragma unroll
        for (int kx = 0; kx < 3; kx++)
#pragma unroll
          for (int k = 0; k < 2; k++) {
            const int j = (ky * 3 + kx) * 2 + k, a = ((oy + ky) * 5 + ox + kx) * 2 + k;
            const uint64_t sa = x3[0][a], za = x3[1][a];
            if constexpr (ARITH == AR_TB) {
              m += pc64(za & (sa ^ w[j]));
              z += pc64(za);
            } else {
              const uint64_t zz = za & w[2 * j + 1];
              z += pc64(zz);
              m += pc64(zz & (sa ^ w[2 * j]));
            }
          }
      const int q = 2 * m - z;
      const uint64_t f0 = __ballot(q + t0 < 0), f1 = __ballot(q + t1 < 0);
      if (lane == 0) { x4[0][p * 4 + (wave & 3)] = ~(f0 | f1); x4[1][p * 4 + (wave & 3)] = ~(f0 ^ f1); }
    }
  }
  __syncthreads();
  if (t < 256) {  // layer 5
    int t0, t1;
    const int q = ternary_q<ARITH, 36>(r5, t, x4[0], x4[1], t0, t1);
    const uint64_t f0 = __ballot(q + t0 < 0), f1 = __ballot(q + t1 < 0);
    if (lane == 0) { x5[0][wave] = ~(f0 | f1); x5[1][wave] = ~(f0 ^ f1); }
  }
  __syncthreads();
  {  // layer 6
    int t0, t1;
    const int q = ternary_q<ARITH, 4>(r6, t, x5[0], x5[1], t0, t1);
    const uint64_t f0 = __ballot(q + t0 < 0), f1 = __ballot(q + t1 < 0);
    if (lane == 0) { x6[0][wave] = ~(f0 | f1); x6[1][wave] = ~(f0 ^ f1); }
  }
  __syncthreads();
  {  // layer 7
    int t0, t1;
    const int q = ternary_q<ARITH, 8>(r7, t, x6[0], x6[1], t0, t1);
    const uint64_t f0 = __ballot(q + t0 < 0), f1 = __ballot(q + t1 < 0);
    if (lane == 0) { x7[0][wave] = ~(f0 | f1); x7[1][wave] = ~(f0 ^ f1); }
  }
  __syncthreads();
  if (wave == 0) {  // layer 8: signed sums  nz - 2m  resp.  z - 2m  = -q
    int t0, t1;
    const int sc = (int)(int16_t)(-ternary_q<ARITH, 8>(r8, lane, x7[0], x7[1], t0, t1));
    if (scores) scores[(size_t)img * 64 + lane] = (int16_t)sc;
    if (classes) {
      int key = (lane < number_class && sc > 0) ? sc * 64 + (63 - lane) : -1;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) key = max(key, __shfl_xor(key, off, 64));
      if (lane == 0) classes[img] = key < 0 ? 0 : 63 - (key & 63);
    }
    mark_done(done, done_seq, lane);
  }
}

// ---------------------------------------------------------------------------
// LFC input: binarizeAndPack (foldedmv-offload.cpp:82-98) on the GPU.
// 784 uint8 -> 13 words, bit i = (p >= 128), bits 784..831 zero.  One lane per
// output word: 4 x 16-byte loads, coalesced across the wave.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t msb4(uint32_t d) { return (((d >> 7) & 0x01010101u) * 0x01020408u) >> 24; }

__global__ __launch_bounds__(kBlock) void k_lfc_binarize(const uint8_t *__restrict__ imgs, uint64_t *__restrict__ out,
                                                          int n_words) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= n_words) return;
  const int img = t / 13, k = t - img * 13;
  const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(imgs + (size_t)img * 784 + k * 64);
  const int nq = (k == 12) ? 1 : 4;  // word 12 holds pixels 768..783 only
  uint64_t word = 0;
  for (int q = 0; q < nq; q++) {
    const uint4 v = src[q];
    const uint32_t bits = msb4(v.x) | (msb4(v.y) << 4) | (msb4(v.z) << 8) | (msb4(v.w) << 12);
    word |= (uint64_t)bits << (16 * q);
  }
  out[t] = word;
}

// ---------------------------------------------------------------------------
// Latency form of lfcW1A1: ONE launch, one 1024-thread block per image, the whole 784 -> 1024 ->
// 1024 -> 1024 -> 64 network.  Here the parallelism of a single image is its NEURONS: thread n of the
// block owns neuron n of the layer (its weight row comes straight from L2 into VGPRs), the layer's
// input vector is the same for all threads (16 words in LDS, read as broadcasts), and the 64
// decisions of a wave are packed by the wave itself -- __ballot IS the wavefront reduction here:
// v_cmp writes the 64-bit lane mask, which is exactly the next layer's input word.  Three
// __syncthreads() separate the layers.  Used for small batches, where the throughput kernels
// (one lane per IMAGE) would leave the chip empty: 1 image 44 us -> see profiles/.
// The same one-launch form for lfcW1A2 (2-bit activations): a layer's output is two words per wave,
// the sign plane (neither threshold exceeded: -1) and the non-zero plane (both or neither: +-1), formed
// from the two ballots of the wave; layers 1..3 use the ternary inner product m = popc(za & (sa ^ w)),
// fire_i  <=>  t_i < nz - 2m  (nz = non-zero inputs of the image, the same for every neuron).
template <int KW>
__device__ __forceinline__ void lfc_load_row2(const uint32_t *__restrict__ rows, int n, uint64_t (&w)[KW], int &t0, int &t1) {
  constexpr int ROW_DW = 2 + 2 * KW;
  const uint32_t *__restrict__ r = rows + (size_t)n * ROW_DW;
  const uint64_t *__restrict__ p = reinterpret_cast<const uint64_t *>(r + 2);
  t0 = (int)r[0];
  t1 = (int)r[1];
#pragma unroll
  for (int k = 0; k < KW; k++) w[k] = p[k];
}
// 2m - nz of one neuron against one image's (sign, non-zero) planes
template <int KW>
__device__ __forceinline__ int lfc_q_tb(const uint64_t (&w)[KW], const uint64_t *sa, const uint64_t *za) {
  int m = 0, nz = 0;
#pragma unroll
  for (int k = 0; k < KW; k++) {
    const uint64_t z = za[k];
    m += pc64(z & (sa[k] ^ w[k]));
    nz += pc64(z);
  }
  return 2 * m - nz;
}

// PACKED: `imgs` holds 13 binarised words per image (the host entry points binarise like the reference does,
// csrc/pack_inputs.h) instead of 784 pixels
template <int IPB, bool PACKED = false>
__global__ __launch_bounds__(1024) void k_lfc_fused_a2(const uint8_t *__restrict__ imgs, uint64_t *__restrict__ words,
                                                        int32_t *__restrict__ classes, const uint32_t *__restrict__ r0,
                                                        const uint32_t *__restrict__ r1, const uint32_t *__restrict__ r2,
                                                        const uint32_t *__restrict__ r3, int n_images, int number_class, unsigned *done, unsigned done_seq) {
  __shared__ uint64_t in0[IPB][16];        // binarised input
  __shared__ uint64_t sg[2][IPB][16], nzp[2][IPB][16];  // ping-pong (sign, non-zero) planes
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, img0 = blockIdx.x * IPB;
  uint64_t w0[13], w1[16], w2[16], w3[16];
  int a0, b0, a1, b1, a2, b2, a3 = 0, b3 = 0;
  // (pixels first: see k_lfc_fused)
  uint8_t px[IPB];
  uint64_t pw[IPB];
#pragma unroll
  for (int i = 0; i < IPB; i++) {
    const int img = img0 + i < n_images ? img0 + i : n_images - 1;
    if constexpr (PACKED) pw[i] = reinterpret_cast<const uint64_t *>(imgs)[(size_t)img * 13 + (t < 13 ? t : 12)];
    else px[i] = imgs[(size_t)img * 784 + (t < 784 ? t : 783)];
  }
  lfc_load_row2<13>(r0, t, w0, a0, b0);
#pragma unroll
  for (int i = 0; i < IPB; i++) {
    if constexpr (PACKED) {
      if (t < 16) in0[i][t] = t < 13 ? pw[i] : 0;
    } else {
      const uint64_t word = __ballot(px[i] >= 128 && t < 784);
      if (lane == 0) in0[i][wave] = word;
    }
  }
  lfc_load_row2<16>(r1, t, w1, a1, b1);
  __syncthreads();
  // layer 0: XNOR inner product, two thresholds (pre-transformed: fire_i <=> m < t_i)
#pragma unroll
  for (int i = 0; i < IPB; i++) {
    int m = 0;
#pragma unroll
    for (int k = 0; k < 13; k++) m += pc64(w0[k] ^ in0[i][k]);
    const uint64_t f0 = __ballot(m < a0), f1 = __ballot(m < b0);
    if (lane == 0) { sg[0][i][wave] = ~(f0 | f1); nzp[0][i][wave] = ~(f0 ^ f1); }
  }
  lfc_load_row2<16>(r2, t, w2, a2, b2);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < IPB; i++) {
    const int q = lfc_q_tb<16>(w1, sg[0][i], nzp[0][i]);
    const uint64_t f0 = __ballot(q + a1 < 0), f1 = __ballot(q + b1 < 0);
    if (lane == 0) { sg[1][i][wave] = ~(f0 | f1); nzp[1][i][wave] = ~(f0 ^ f1); }
  }
  if (wave == 0) lfc_load_row2<16>(r3, lane, w3, a3, b3);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < IPB; i++) {
    const int q = lfc_q_tb<16>(w2, sg[1][i], nzp[1][i]);
    const uint64_t f0 = __ballot(q + a2 < 0), f1 = __ballot(q + b2 < 0);
    if (lane == 0) { sg[0][i][wave] = ~(f0 | f1); nzp[0][i][wave] = ~(f0 ^ f1); }
  }
  __syncthreads();
  if (wave == 0) {  // last layer: 64 neurons, one threshold, one wave
#pragma unroll
    for (int i = 0; i < IPB; i++) {
      const uint64_t word = __ballot(lfc_q_tb<16>(w3, sg[0][i], nzp[0][i]) + a3 < 0);
      if (lane == 0 && img0 + i < n_images) {
        words[img0 + i] = word;
        if (classes) {
          const uint64_t w = word & (~0ull >> (64 - number_class));
          classes[img0 + i] = w ? 63 - __builtin_clzll(w) : 0;
        }
      }
    }
    mark_done(done, done_seq, lane);
  }
}

// ---------------------------------------------------------------------------
template <int KW>
__device__ __forceinline__ void lfc_load_row(const uint32_t *__restrict__ rows, int n, uint64_t (&w)[KW], int &t) {
  constexpr int ROW_DW = 2 + 2 * KW;
  const uint32_t *__restrict__ r = rows + (size_t)n * ROW_DW;
  const uint64_t *__restrict__ p = reinterpret_cast<const uint64_t *>(r + 2);
  t = (int)r[0];
#pragma unroll
  for (int k = 0; k < KW; k++) w[k] = p[k];
}
template <int KW>
__device__ __forceinline__ bool lfc_fires(const uint64_t (&w)[KW], int t, const uint64_t *act) {
  int m = 0;
#pragma unroll
  for (int k = 0; k < KW; k++) m += pc64(w[k] ^ act[k]);
  return m < t;
}

template <int IPB, bool PACKED = false>
__global__ __launch_bounds__(1024) void k_lfc_fused(const uint8_t *__restrict__ imgs, uint64_t *__restrict__ words,
                                                     int32_t *__restrict__ classes, const uint32_t *__restrict__ r0,
                                                     const uint32_t *__restrict__ r1, const uint32_t *__restrict__ r2,
                                                     const uint32_t *__restrict__ r3, int n_images, int number_class, unsigned *done, unsigned done_seq) {
  __shared__ uint64_t act[2][IPB][16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, img0 = blockIdx.x * IPB;
  // the layers are a dependent chain, their weight traffic need not be: the row of layer L+1 is
  // requested before layer L is evaluated (two rows in flight: more would spill at 1024 threads).
  // A row, once in VGPRs, serves all IPB images of the block.
  uint64_t w0[13], w1[16], w2[16], w3[16];
  int t0, t1, t2, t3 = 0;
  // Two rows are in flight at a time.  (For a single image per block, requesting the rows of ALL three big layers at
  // entry -- 90 dwords per thread, 118 VGPRs -- was built and measured in round 3: the kernel trace shows 10.1 us
  // against 8.0 and inference() reports 9.5 against 9.0: one CU takes 377 KB through its 64 B/clk port in 2.4 us
  // either way, and all at once nothing of it hides behind a layer's evaluation.  profiles/r03_latency_all_rows_ab.txt)
  // the pixels are requested first: vector loads return in order, so waiting for them does not mean waiting for rows
  // (against the round-2 order -- row 0, then the pixels behind a branch -- no measurable difference: 8.7-9.3 vs 8.8-9.1 us)
  uint8_t px[IPB];
  uint64_t pw[IPB];
#pragma unroll
  for (int i = 0; i < IPB; i++) {
    const int img = img0 + i < n_images ? img0 + i : n_images - 1;  // ragged tail: duplicate, store guarded
    // (unconditional: a branch would put the wait in front of the row loads)
    if constexpr (PACKED) pw[i] = reinterpret_cast<const uint64_t *>(imgs)[(size_t)img * 13 + (t < 13 ? t : 12)];
    else px[i] = imgs[(size_t)img * 784 + (t < 784 ? t : 783)];
  }
  lfc_load_row<13>(r0, t, w0, t0);
  // binarizeAndPack: bit i = (pixel i >= 128); pixels 784..831 are padding (0)
#pragma unroll
  for (int i = 0; i < IPB; i++) {
    if constexpr (PACKED) {
      if (t < 16) act[0][i][t] = t < 13 ? pw[i] : 0;  // already binarised on the host: words 13..15 are padding
    } else {
      const uint64_t word = __ballot(px[i] >= 128 && t < 784);
      if (lane == 0) act[0][i][wave] = word;  // waves 13..15 write zeros
    }
  }
  lfc_load_row<16>(r1, t, w1, t1);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < IPB; i++) {
    const uint64_t word = __ballot(lfc_fires<13>(w0, t0, act[0][i]));
    if (lane == 0) act[1][i][wave] = word;
  }
  lfc_load_row<16>(r2, t, w2, t2);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < IPB; i++) {
    const uint64_t word = __ballot(lfc_fires<16>(w1, t1, act[1][i]));
    if (lane == 0) act[0][i][wave] = word;
  }
  if (wave == 0) lfc_load_row<16>(r3, lane, w3, t3);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < IPB; i++) {
    const uint64_t word = __ballot(lfc_fires<16>(w2, t2, act[0][i]));
    if (lane == 0) act[1][i][wave] = word;
  }
  __syncthreads();
  if (wave == 0) {  // last layer: 64 neurons, one wave
#pragma unroll
    for (int i = 0; i < IPB; i++) {
      const uint64_t word = __ballot(lfc_fires<16>(w3, t3, act[1][i]));
      if (lane == 0 && img0 + i < n_images) {
        words[img0 + i] = word;
        if (classes) {
          const uint64_t w = word & (~0ull >> (64 - number_class));
          classes[img0 + i] = w ? 63 - __builtin_clzll(w) : 0;
        }
      }
    }
    mark_done(done, done_seq, lane);
  }
}

// ---------------------------------------------------------------------------
// lfcW1A1, mid-size batches (BASELINE config 2: 10 000 MNIST images): the whole network in ONE launch at the
// throughput kernels' instruction rate.  Six staged launches cost ~5 us each in gaps, prologues and tails --
// at 10 000 images that is a third of the run.  A 1024-thread block owns ceil(n / 512) images and walks the
// four layers over ALL of them before it moves on: thread = neuron, its weight row sits in VGPRs for the whole
// layer, the wave's 64 decisions for an image are one v_cmp mask = one word of the next layer's input, parked
// in lane (i mod 64) of a VGPR pair by two v_writelane and stored once per 64 images.  Four hand-offs per
// launch whatever the batch.
// First form (round 2, replaced): the image's 16 words broadcast from LDS into 32 VGPRs of every lane -- 128 KB
// of LDS reads per image and layer per CU, and the counters said that is what the waves waited for
// (profiles/r02_sq_lfc_block.json: SQ_WAIT_ANY 48 % of wave-cycles, 3.9 SIMD-cycles per VALU instruction against
// 3.1 in the throughput kernels; 65 us for 10 000 images).  This form keeps a wave-uniform operand where it
// belongs, in SGPRs: the maps between the layers live in the two global workspace buffers (128 B per image,
// L2-resident), a layer's output is stored with vector stores, and after the block's barrier + s_dcache_inv
// the next layer reads it with two s_load_dwordx16 per image through the scalar cache: the pair becomes
// v_xor(s, v) + v_bcnt exactly as in the throughput kernels, no LDS at all, 47 VGPRs and 68 SGPRs => two
// 1024-thread blocks per CU (8 waves per SIMD), each covering the other's barriers, weight loads and
// scalar-load latencies (profiles/r02_lfc_block_sweep.txt: 10 000 images 74 -> 61 us).
// ---------------------------------------------------------------------------
template <int KW>
__device__ __forceinline__ void lfc_row_regs(const uint32_t *__restrict__ rows, int n, uint32_t (&wl)[KW], uint32_t (&wh)[KW], int &nt) {
  constexpr int ROW_DW = 2 + 2 * KW;
  const uint32_t *__restrict__ r = rows + (size_t)n * ROW_DW;
  nt = -(int)r[0];
  const uint2 *__restrict__ p = reinterpret_cast<const uint2 *>(r + 2);
#pragma unroll
  for (int k = 0; k < KW; k++) {
    const uint2 v = p[k];
    wl[k] = v.x;
    wh[k] = v.y;
  }
}

#ifdef BNN_LFC_STAMPS
// diagnostic build only (tools/build_variant.sh ... -DBNN_LFC_STAMPS): wave 0 of every block of k_lfc_block_s writes the
// 100 MHz wall clock at entry, behind each of the five hand-offs / layers, and at exit: where a launch's time goes
__device__ unsigned long long g_lfc_stamps[1024 * 8];
#define LFC_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_lfc_stamps[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
// ... and every wave its own: [block][wave][16]
__device__ unsigned long long g_lfc_wstamps[1024 * 16 * 16];
#define LFC_WSTAMP(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) g_lfc_wstamps[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define LFC_STAMP(i) do { } while (0)
#define LFC_WSTAMP(i) do { } while (0)
#endif
#ifdef BNN_LFC_NO_PRIO  // A/B build only
#define LFC_PRIO(p) do { } while (0)
#else
#define LFC_PRIO(p) __builtin_amdgcn_s_setprio(p)
#endif
typedef uint32_t v16u __attribute__((ext_vector_type(16)));
// row `n` of a layer with KW input words into the first KW entries of (wl, wh)
template <int KW>
__device__ __forceinline__ void lfc_row_regs16(const uint32_t *__restrict__ rows, int n, uint32_t (&wl)[16], uint32_t (&wh)[16], int &nt) {
  constexpr int ROW_DW = 2 + 2 * KW;
  const uint32_t *__restrict__ r = rows + (size_t)n * ROW_DW;
  nt = -(int)r[0];
  const uint2 *__restrict__ p = reinterpret_cast<const uint2 *>(r + 2);
#pragma unroll
  for (int k = 0; k < KW; k++) {
    const uint2 v = p[k];
    wl[k] = v.x;
    wh[k] = v.y;
  }
}
// Both loads AND their wait in one statement, early-clobber outputs: the compiler can neither place `lo` over the
// address pair the second load still reads, nor copy / spill the tuples between the loads and the wait.
__device__ __forceinline__ void sload_image(const uint64_t *p, v16u &lo, v16u &hi) {  // p: wave-uniform
  asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(lo), "=&s"(hi) : "s"(p));
}
// m - t of this thread's neuron for one image (its 32 dwords in SGPRs).  (Two neurons per thread -- 4 * KW
// pairs per scalar-load wait, 512-thread blocks -- was measured too: 3 % faster at 131 072 images, 10 % slower at
// 10 000, where this kernel is used.)
template <int KW, int N = KW>
__device__ __forceinline__ int lfc_neuron_s(const uint32_t (&wl)[N], const uint32_t (&wh)[N], int nt, const v16u &lo, const v16u &hi,
                                            uint32_t &t) {
  auto word = [&](int d) { return d < 16 ? lo[d] : hi[d - 16]; };
  int m;
  asm("v_xor_b32 %0, %2, %3\n\tv_bcnt_u32_b32 %1, %0, %4" : "+v"(t), "=v"(m) : "s"(word(0)), "v"(wl[0]), "v"(nt));
  asm("v_xor_b32 %0, %2, %3\n\tv_bcnt_u32_b32 %1, %0, %1" : "+v"(t), "+v"(m) : "s"(word(1)), "v"(wh[0]));
#pragma unroll
  for (int k = 1; k < KW; k++) {
    asm("v_xor_b32 %0, %2, %3\n\tv_bcnt_u32_b32 %1, %0, %1" : "+v"(t), "+v"(m) : "s"(word(2 * k)), "v"(wl[k]));
    asm("v_xor_b32 %0, %2, %3\n\tv_bcnt_u32_b32 %1, %0, %1" : "+v"(t), "+v"(m) : "s"(word(2 * k + 1)), "v"(wh[k]));
  }
  return m;
}

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"  // m0 is named as a clobber on purpose (nothing else in this kernel uses it)
// decisions of this wave's 64 neurons for image i of the chunk -> lane i of (lo, hi)
__device__ __forceinline__ void park_word(int &lo, int &hi, uint64_t word, int i) {
  asm("s_mov_b32 m0, %4\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
      : "+v"(lo), "+v"(hi)
      : "s"((uint32_t)word), "s"((uint32_t)(word >> 32)), "s"(i)
      : "m0");
}
#pragma clang diagnostic pop

// one layer over the block's cnt images: in / out = global maps [image][16] words (in: wave-uniform pointer)
// The weight row of the layer is in (wl, wh, nt) on entry; on exit the row of the NEXT layer (NKW words, 0: none) has
// been requested into the same registers.  (Written behind this layer's last pair, in front of the hand-off; the
// compiler sinks the loads behind the barrier all the same -- and rightly so: the wave that arrives last, the one
// the barrier waits for, could not have issued them any earlier, its registers hold the current row until then.  Only
// layer 0's row, requested at kernel entry, has its trip to L2 -- 1.3-2 us, per-wave clock stamps -- hidden.)
template <int KW, int NKW>
__device__ __forceinline__ void lfc_block_layer_s(const uint32_t *__restrict__ next_rows, int next_neuron, uint32_t (&wl)[16], uint32_t (&wh)[16],
                                                  int &nt, const uint64_t *in, uint64_t *out, int cnt, int wave, int lane, uint32_t &t,
                                                  int stamp = 0) {
  LFC_WSTAMP(stamp);
  v16u a_lo, a_hi;
  for (int base = 0; base < cnt; base += 64) {
    const int m = __builtin_amdgcn_readfirstlane(min(64, cnt - base));
    int lo = 0, hi = 0;
    // One image in flight per wave: a second SGPR buffer would push the kernel past 80 SGPRs and cost the second
    // block per CU; with 8 waves on the SIMD the other seven cover this wave's scalar load.  (Touching image i + 2
    // in the scalar cache ahead of time was measured: no gain -- and it would have been the only access outside
    // the block's own images.)
    // (Walking the images in an order rotated per wave -- so that an image's scalar-cache miss is paid by one wave
    // instead of all sixteen at once -- was measured: no change, 61.9 vs 61.6 us for 10 000 images.)
    for (int i = 0; i < m; i++) {
      sload_image(in + (size_t)(base + i) * 16, a_lo, a_hi);
      park_word(lo, hi, __ballot(lfc_neuron_s<KW, 16>(wl, wh, nt, a_lo, a_hi, t) < 0), i);
#ifdef BNN_LFC_STAMPS
      if (base == 0 && i == 0) LFC_WSTAMP(stamp + 1);
#endif
    }
    if (lane < m) out[(size_t)(base + lane) * 16 + wave] = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
  }
  if constexpr (NKW > 0) lfc_row_regs16<NKW>(next_rows, next_neuron, wl, wh, nt);
  LFC_WSTAMP(stamp + 2);
}

// all waves' stores of a layer are in L2, and no stale line of the map is left in the scalar cache
__device__ __forceinline__ void lfc_block_handoff() {
  // this wave's stores have reached L2 (the vector L1 writes through; the scalar cache reads from L2, not from
  // the vector L1 -- __syncthreads() alone orders only what the CU's vector L1 serves)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

template <bool PACKED>
__global__ __launch_bounds__(1024, 2) void k_lfc_block_s(const uint8_t *__restrict__ imgs, uint64_t *__restrict__ words,
                                                          int32_t *__restrict__ classes, uint64_t *__restrict__ gA, uint64_t *__restrict__ gB,
                                                          const uint32_t *__restrict__ r0, const uint32_t *__restrict__ r1,
                                                          const uint32_t *__restrict__ r2, const uint32_t *__restrict__ r3, int n_images,
                                                          int number_class, int ipb) {
  const int tid = threadIdx.x, lane = tid & 63, img0 = blockIdx.x * ipb;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                  // in an SGPR: it indexes scalar loads
  const int cnt = __builtin_amdgcn_readfirstlane(min(ipb, n_images - img0));  // block-uniform, >= 1 by the grid size
  uint64_t *A = gA + (size_t)img0 * 16, *B = gB + (size_t)img0 * 16;
  uint32_t t = chain_temp();
  LFC_STAMP(0);
  LFC_WSTAMP(0);
  uint32_t wl[16], wh[16];
  int nt;
  lfc_row_regs16<13>(r0, tid, wl, wh, nt);  // layer 0's row: its trip to L2 runs behind the binarisation below
  // binarizeAndPack into A: one lane per output word (words 13..15 of an image are never read by layer 0)
  for (int idx = tid; idx < cnt * 16; idx += 1024) {
    const int i = idx >> 4, k = idx & 15;
    uint64_t word = 0;
    if constexpr (PACKED) {  // binarised on the host (csrc/pack_inputs.h): 13 words per image, re-pitched to 16 for the scalar loads
      if (k < 13) word = reinterpret_cast<const uint64_t *>(imgs)[(size_t)(img0 + i) * 13 + k];
    } else if (k < 13) {
      const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(imgs + (size_t)(img0 + i) * 784 + k * 64);
      const int nq = (k == 12) ? 1 : 4;  // word 12 holds pixels 768..783 only
      for (int q = 0; q < nq; q++) {
        const uint4 v = src[q];
        const uint32_t bits = msb4(v.x) | (msb4(v.y) << 4) | (msb4(v.z) << 8) | (msb4(v.w) << 12);
        word |= (uint64_t)bits << (16 * q);
      }
    }
    A[idx] = word;
  }
  LFC_WSTAMP(1);
  lfc_block_handoff();
  LFC_STAMP(1);
  LFC_WSTAMP(2);
  // The SIMD's arbiter serves the OLDEST wave first.  With two blocks per CU that made the older block run three
  // layers in 40 us while the younger one was still in its first, which then ran the rest alone at half the issue
  // rate: 244 of 500 blocks finished at 40 us, the other 244 at 61 (per-wave clock stamps, profiles/r03_lfc_block_stamps*.txt).
  // So a wave's priority falls with the layer it is in: whichever block is behind goes first, the two stay within a
  // layer of each other and the SIMDs have eight waves to choose from until the end (10 000 images 62.0 -> 57.0 us).
  LFC_PRIO(3);
  lfc_block_layer_s<13, 16>(r1, tid, wl, wh, nt, A, B, cnt, wave, lane, t, 3);
  LFC_STAMP(2);
  lfc_block_handoff();
  LFC_STAMP(3);
  LFC_WSTAMP(6);
  LFC_PRIO(2);
  lfc_block_layer_s<16, 16>(r2, tid, wl, wh, nt, B, A, cnt, wave, lane, t, 7);
  lfc_block_handoff();
  LFC_STAMP(4);
  LFC_WSTAMP(10);
  LFC_PRIO(1);
  lfc_block_layer_s<16, 16>(r3, lane, wl, wh, nt, A, B, cnt, wave, lane, t, 11);  // (layer 3: 64 neurons, neuron = lane in every wave)
  lfc_block_handoff();
  LFC_STAMP(5);
  LFC_WSTAMP(14);
  LFC_PRIO(0);
  {  // layer 3 + decode: the waves share out the images
    v16u lo, hi;
    for (int i = wave; i < cnt; i += 16) {
      sload_image(B + (size_t)i * 16, lo, hi);
      const uint64_t word = __ballot(lfc_neuron_s<16, 16>(wl, wh, nt, lo, hi, t) < 0);
      if (lane == 0) {
        words[img0 + i] = word;
        if (classes) {
          const uint64_t w = word & (~0ull >> (64 - number_class));
          classes[img0 + i] = w ? 63 - __builtin_clzll(w) : 0;
        }
      }
    }
  }
  LFC_STAMP(6);
  LFC_WSTAMP(15);
}

#ifdef BNN_LFC_STAMPS
}  // namespace
hipError_t lfc_stamps_read(unsigned long long *dst) { return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_lfc_stamps), sizeof(g_lfc_stamps)); }
hipError_t lfc_wstamps_read(unsigned long long *dst) { return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_lfc_wstamps), sizeof(g_lfc_wstamps)); }
namespace {
#endif

// LFC output decode, batched form (testPrebinarized_nolabel_multiple_images,
// foldedmv-offload.cpp:202-220): mask to number_class bits, class = index of
// the highest set bit, 0 when none.  (unsigned)log2((double)w) equals that
// index exactly while w < 2^47; the host decodes larger label sets itself.
__global__ __launch_bounds__(kBlock) void k_lfc_decode(const uint64_t *__restrict__ words, int32_t *__restrict__ classes,
                                                        int n, int number_class) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const uint64_t mask = ~0ull >> (64 - number_class);
  const uint64_t w = words[i] & mask;
  classes[i] = w ? 63 - __builtin_clzll(w) : 0;
}

// images: up to here lfcW1A1 runs as one k_lfc_block_s launch.  Round 2 (oldest-wave-first arbitration left to itself)
// it lost to the staged kernels beyond ~40 000 images; with the per-layer wave priorities it wins over the whole range
// of one pass (profiles/r03_lfc_block_priorities.txt, us per batch, block vs staged): 32 768 images 161 vs 175,
// 49 152 219 vs 239, 65 536 286 vs 304, 98 304 425 vs 438, 131 072 565 vs 572.  BNN_MI355X_LFC_BLOCK_MAX overrides
// (tools/batch_sweep.py); the staged kernels remain what per-stage profiling and the stage-output test hook run.
// images: up to here lfcW1A1 runs as k_lfc_fused<IPB> (a block per 1/2/4 images), beyond as k_lfc_block_s -- measured
// (profiles/r03_lfc_fused_vs_block.txt, us per batch, fused vs block): 513 images 15.5 vs 16.5, 1 024 16.1 vs 17.9,
// 1 025 22.8 vs 18.2, 2 048 22.9 vs 19.1, 4 096 39.5 vs 25.7.  BNN_MI355X_LFC_FUSED_MAX overrides (tools/batch_sweep.py)
inline long long lfc_fused_max() {
  static const long long v = [] {
    const char *e = std::getenv("BNN_MI355X_LFC_FUSED_MAX");
    return e ? std::atoll(e) : 1024LL;
  }();
  return v;
}
inline long long lfc_block_max() {
  static const long long v = [] {
    const char *e = std::getenv("BNN_MI355X_LFC_BLOCK_MAX");
    return e ? std::atoll(e) : 131072LL;
  }();
  return v;
}

// images: from here on layer 0 runs in its LDS-staged tile form (k_conv0_tile) -- measured (tools/stage_times.py): 2 048
// images 28.6 vs 18.2 us for the lane-per-pixel form, 8 192 44.7 vs 47.0, 131 072 516 vs 785; BNN_MI355X_L0_TILE_MIN overrides
inline long long l0_tile_min() {
  static const long long v = [] {
    const char *e = std::getenv("BNN_MI355X_L0_TILE_MIN");
    return e ? std::atoll(e) : 8192LL;
  }();
  return v;
}

// neuron groups per block: all of them once the work items alone fill the chip (256 CUs x 8 blocks),
// so that a lane writes whole output words and reads its window once; otherwise one (parallelism first)
inline int gpb_for(long long items, int groups) { return (items + kBlock - 1) / kBlock >= 2048 ? groups : 1; }

// small batches: 8 neurons per block instead of 32 while the 32-neuron grid is short of blocks (four
// times as many, four times shorter blocks balance better over the 1024 SIMDs)
inline bool narrow_for(long long items, int groups32, long long limit) { return ((items + kBlock - 1) / kBlock) * groups32 < limit; }
// measured (tools/batch_sweep.py): the conv stages gain from the 8-neuron form up to ~8192 blocks of the
// 32-neuron grid (1024 images +10 %, 4096 +9 %); the FC stacks of the LFC nets (16-word inputs, 1024
// neurons) lose beyond 512
constexpr long long kNarrowLimitCnv = 8192, kNarrowLimitLfc = 512;
constexpr long long kFcLastWaveMax = 32768;
constexpr long long kPixelLaneMax = 512;    // images: cnvW1A1 layers 1..3 with a lane per output pixel
constexpr long long kCnvTailMax = 1024;  // images: cnvW1A1 layers 4..8 as one block-per-image launch (k_cnv_tail)  // images: CNV layer 8 with a wave per image instead of a lane per image

inline dim3 grid_for(long long items, int groups) {  // matches map_block()
  const long long item_blocks = (items + kBlock - 1) / kBlock;
  return dim3((unsigned)(((item_blocks + 7) / 8) * 8 * groups));
}

#define BNN_LAUNCH(kern, grid, stream, ...)                                   \
  do {                                                                        \
    if ((grid).x > 0) hipLaunchKernelGGL(kern, grid, dim3(kBlock), 0, stream, __VA_ARGS__); \
  } while (0)
// one thresholded stage over `items` work items and `groups32` groups of 32 neurons: the 8-neuron
// form for small batches (narrow_for), else the 32-neuron form with gpb_for() groups per block
#define BNN_STAGE(kern32, kern8, items, groups32, in, out, rows)                                                                   \
  do {                                                                                                                             \
    const long long it_ = (items);                                                                                                 \
    if (narrow_for(it_, (groups32), narrow_limit)) BNN_LAUNCH(kern8, grid_for(it_, (groups32) * 4), s, in, out, rows, (int)it_, (groups32) * 4, 1); \
    else BNN_LAUNCH(kern32, grid_for(it_, (groups32) / gpb_for(it_, (groups32))), s, in, out, rows, (int)it_, (groups32), gpb_for(it_, (groups32))); \
  } while (0)
// stage boundary: with profiling on, an event separates consecutive stages
#define BNN_MARK(ev, i, stream)                          \
  do {                                                   \
    if (ev) (void)hipEventRecord((ev)[i], stream);       \
  } while (0)

// TWO: the -2-aware instantiations of the 2-bit-weight kernels (see two_extra)
template <int ARITH, bool OUT2, bool TWO = false>
void run_cnv_t(const CnvLaunch &a) {
  const long long n = a.n, narrow_limit = kNarrowLimitCnv;
  uint32_t *A = reinterpret_cast<uint32_t *>(a.buf0), *B = reinterpret_cast<uint32_t *>(a.buf1);
  const uint64_t *A64 = reinterpret_cast<const uint64_t *>(a.buf0), *B64 = reinterpret_cast<const uint64_t *>(a.buf1);
  hipStream_t s = a.stream;
  BNN_MARK(a.events, 0, s);
  if (a.l0_mfma) {
    // throughput form (images staged in LDS, a block per kL0Imgs images) once its grid fills the chip; below
    // that the lane-per-pixel form, whose 15 waves per image spread over the CUs
    const dim3 g0((unsigned)((n * 900 + kBlock - 1) / kBlock));  // lane = pixel
    if (a.last_stage >= 0) {
      if (n >= l0_tile_min()) BNN_LAUNCH((k_conv0_tile<OUT2>), dim3((unsigned)((n + kL0Imgs - 1) / kL0Imgs)), s, a.images, A, a.l0_mfma, (int)n);
      else BNN_LAUNCH((k_conv0_mfma<OUT2>), g0, s, a.images, A, a.l0_mfma, (int)(n * 900));
    }
  } else {
    if (a.last_stage >= 0) BNN_LAUNCH((k_conv0<OUT2>), grid_for(n * 900, 2 / gpb_for(n * 900, 2)), s, a.images, A, a.rows[0], (int)(n * 900), 2, gpb_for(n * 900, 2));
  }
  BNN_MARK(a.events, 1, s);
  if constexpr (ARITH == AR_XNOR && !OUT2) {
    // tiny batches: a lane per output PIXEL instead of per 2x2 quad (k_vec_x in its window form, 8 neurons
    // per block): four times the lanes, a quarter of the serial work of each
    const bool pix = n <= kPixelLaneMax;
    if (a.last_stage >= 1) {
      if (a.l1_literal) {  // comparison figure (BNN_MI355X_L1=lds): the north-star's wording, untuned
        BNN_LAUNCH(k_l1_literal, dim3((unsigned)((n * 784 + kBlock - 1) / kBlock)), s, A64, reinterpret_cast<uint64_t *>(B), a.rows[1], (int)(n * 784));
      } else if (a.l1_mfma) {  // side experiment (BNN_MI355X_L1=mfma): the layer on the matrix pipe
        const long long pairs = (n + 1) / 2;
        hipLaunchKernelGGL(k_l1_mfma, dim3((unsigned)(pairs < 512 ? pairs : 512)), dim3(256), 0, s, reinterpret_cast<const uint32_t *>(a.buf0), B,
                           a.l1_mfma, (int)n);
      } else if (pix) BNN_LAUNCH((k_vec_x<9, true, 1, 30, 8, true>), grid_for(n * 784, 8), s, A64, B, a.rows[1], (int)(n * 784), 8, 1);
      else BNN_STAGE((k_quad_x<1, 30, true>), (k_quad_x<1, 30, true, 8>), n * 196, 2, A64, B, a.rows[1]);
    }
    BNN_MARK(a.events, 2, s);
    if (a.last_stage >= 2) {
      if (pix) BNN_LAUNCH((k_vec_x<9, true, 1, 14, 8>), grid_for(n * 144, 16), s, B64, A, a.rows[2], (int)(n * 144), 16, 1);
      else BNN_STAGE((k_quad_x<1, 14, false>), (k_quad_x<1, 14, false, 8>), n * 36, 4, B64, A, a.rows[2]);
    }
    BNN_MARK(a.events, 3, s);
    if (a.last_stage >= 3) {
      if (pix) BNN_LAUNCH((k_vec_x<18, true, 2, 12, 8, true>), grid_for(n * 100, 16), s, A64, B, a.rows[3], (int)(n * 100), 16, 1);
      else BNN_STAGE((k_quad_x<2, 12, true>), (k_quad_x<2, 12, true, 8>), n * 25, 4, A64, B, a.rows[3]);
    }
    BNN_MARK(a.events, 4, s);
    if (n <= kCnvTailMax && !a.events && a.last_stage >= kCnvStages - 1) {
      // small batch: layers 4..8 as one launch (no per-stage events there: there are no stages)
      hipLaunchKernelGGL(k_cnv_tail, dim3((unsigned)n), dim3(512), 0, s, B64, a.scores, a.classes, a.rows[4], a.rows[5], a.rows[6],
                         a.rows[7], a.rows[8], a.number_class, n == 1 ? a.done_flag : nullptr, a.done_seq);
      return;
    }
    if (a.last_stage >= 4) BNN_STAGE((k_vec_x<18, true, 2, 5>), (k_vec_x<18, true, 2, 5, 8>), n * 9, 8, B64, A, a.rows[4]);
    BNN_MARK(a.events, 5, s);
    if (a.last_stage >= 5) BNN_STAGE((k_vec_x<36, false, 1, 1>), (k_vec_x<36, false, 1, 1, 8>), n, 8, A64, B, a.rows[5]);
    BNN_MARK(a.events, 6, s);
    if (a.last_stage >= 6) BNN_STAGE((k_vec_x<4, false, 1, 1>), (k_vec_x<4, false, 1, 1, 8>), n, 16, B64, A, a.rows[6]);
    BNN_MARK(a.events, 7, s);
    if (a.last_stage >= 7) BNN_STAGE((k_vec_x<8, false, 1, 1>), (k_vec_x<8, false, 1, 1, 8>), n, 16, A64, B, a.rows[7]);
    BNN_MARK(a.events, 8, s);
  } else {
    const bool pix = n <= kPixelLaneMax;  // tiny batches: a lane per output pixel (see the XNOR branch)
    if (a.last_stage >= 1) {
      if (pix) BNN_LAUNCH((k_vec<ARITH, 9, OUT2, true, 1, 30, 8, true, TWO>), grid_for(n * 784, 8), s, A64, B, a.rows[1], (int)(n * 784), 8, 1);
      else BNN_STAGE((k_quad<ARITH, 1, 30, true, OUT2, 32, TWO>), (k_quad<ARITH, 1, 30, true, OUT2, 8, TWO>), n * 196, 2, A64, B, a.rows[1]);
    }
    BNN_MARK(a.events, 2, s);
    if (a.last_stage >= 2) {
      if (pix) BNN_LAUNCH((k_vec<ARITH, 9, OUT2, true, 1, 14, 8, false, TWO>), grid_for(n * 144, 16), s, B64, A, a.rows[2], (int)(n * 144), 16, 1);
      else BNN_STAGE((k_quad<ARITH, 1, 14, false, OUT2, 32, TWO>), (k_quad<ARITH, 1, 14, false, OUT2, 8, TWO>), n * 36, 4, B64, A, a.rows[2]);
    }
    BNN_MARK(a.events, 3, s);
    if (a.last_stage >= 3) {
      if (pix) BNN_LAUNCH((k_vec<ARITH, 18, OUT2, true, 2, 12, 8, true, TWO>), grid_for(n * 100, 16), s, A64, B, a.rows[3], (int)(n * 100), 16, 1);
      else BNN_STAGE((k_quad<ARITH, 2, 12, true, OUT2, 32, TWO>), (k_quad<ARITH, 2, 12, true, OUT2, 8, TWO>), n * 25, 4, A64, B, a.rows[3]);
    }
    BNN_MARK(a.events, 4, s);
    if constexpr (!TWO) {  // (the -2-aware variant of this kernel would spill: such runs take the staged layers)
      if (n <= kCnvTailMax && !a.events && a.last_stage >= kCnvStages - 1) {
        hipLaunchKernelGGL((k_cnv_tail_a2<ARITH>), dim3((unsigned)n), dim3(512), 0, s, B64, a.scores, a.classes, a.rows[4], a.rows[5],
                           a.rows[6], a.rows[7], a.rows[8], a.number_class, n == 1 ? a.done_flag : nullptr, a.done_seq);
        return;
      }
    }
    if (a.last_stage >= 4) BNN_STAGE((k_vec<ARITH, 18, OUT2, true, 2, 5, 32, false, TWO>), (k_vec<ARITH, 18, OUT2, true, 2, 5, 8, false, TWO>), n * 9, 8, B64, A, a.rows[4]);
    BNN_MARK(a.events, 5, s);
    if (a.last_stage >= 5) BNN_STAGE((k_vec<ARITH, 36, OUT2, false, 1, 1, 32, false, TWO>), (k_vec<ARITH, 36, OUT2, false, 1, 1, 8, false, TWO>), n, 8, A64, B, a.rows[5]);
    BNN_MARK(a.events, 6, s);
    if (a.last_stage >= 6) BNN_STAGE((k_vec<ARITH, 4, OUT2, false, 1, 1, 32, false, TWO>), (k_vec<ARITH, 4, OUT2, false, 1, 1, 8, false, TWO>), n, 16, B64, A, a.rows[6]);
    BNN_MARK(a.events, 7, s);
    if (a.last_stage >= 7) BNN_STAGE((k_vec<ARITH, 8, OUT2, false, 1, 1, 32, false, TWO>), (k_vec<ARITH, 8, OUT2, false, 1, 1, 8, false, TWO>), n, 16, A64, B, a.rows[7]);
    BNN_MARK(a.events, 8, s);
  }
  if (a.last_stage >= 8) {
    if (n <= kFcLastWaveMax) BNN_LAUNCH((k_fclast_wave<ARITH, 8, TWO>), dim3((unsigned)((n + 3) / 4)), s, B64, a.scores, a.classes, a.rows[8], (int)n, a.number_class);
    else BNN_LAUNCH((k_fclast<ARITH, 8, TWO>), grid_for(n, 1), s, B64, a.scores, a.classes, a.rows[8], (int)n, a.number_class);
  }
  BNN_MARK(a.events, 9, s);
}

}  // namespace

const char *stage_name(bool is_cnv, int stage) {
  static const char *cnv[kCnvStages] = {"k_conv0 (L0)", "k_quad L1+pool", "k_quad L2", "k_quad L3+pool", "k_vec L4",
                                        "k_vec L5", "k_vec L6", "k_vec L7", "k_fclast L8+decode"};
  static const char *lfc[kLfcStages] = {"k_lfc_binarize", "k_vec L0", "k_vec L1", "k_vec L2", "k_vec L3", "k_lfc_decode"};
  if (stage < 0 || stage >= (is_cnv ? kCnvStages : kLfcStages)) return "";
  return is_cnv ? cnv[stage] : lfc[stage];
}

// bytes per image of the output of `stage` (what the next stage reads), and which buffer holds it
size_t stage_output_bytes(bool is_cnv, int abits, int stage, int *in_buf1) {
  static const size_t cnv[8] = {7200, 1568, 2304, 400, 288, 32, 64, 64};
  if (is_cnv) {
    if (stage < 0 || stage > 7) return 0;
    *in_buf1 = stage & 1;
    return cnv[stage] * (size_t)abits;
  }
  if (stage < 0 || stage > 3) return 0;
  *in_buf1 = stage & 1;
  return stage == 0 ? 104 : 128 * (size_t)abits;
}

// per-image bytes of the two ping-pong activation buffers
void cnv_workspace_bytes(int abits, size_t *buf0, size_t *buf1) {
  *buf0 = 900 * 8 * (size_t)abits;  // L0 out 30x30x64 (largest of the even stages)
  *buf1 = 196 * 8 * (size_t)abits;  // L1 out 14x14x64 (largest of the odd stages)
}
void lfc_workspace_bytes(int abits, size_t *buf0, size_t *buf1) {
  *buf0 = 128 * (size_t)abits;  // 1024 activations
  *buf1 = 128 * (size_t)abits;
}

void l1_mfma_table(const uint32_t *rows, uint8_t *dst) {
  constexpr int ROW_DW = 2 + 2 * 9;  // cnvW1A1 layer 1: t0, t1, 9 x u64 (bit = 1 <=> weight +1)
  for (int tap = 0; tap < 9; tap++)
    for (int mt = 0; mt < 2; mt++)
      for (int lane = 0; lane < 64; lane++) {
        const int r = lane & 31, h = lane >> 5, n = 32 * mt + r;
        const uint32_t bits = rows[n * ROW_DW + 2 + 2 * tap + h];  // channels 32h .. 32h+31 of this tap
        uint8_t *o = dst + ((size_t)(tap * 2 + mt) * 64 + lane) * 16;
        for (int j = 0; j < 16; j++) {
          const uint32_t lo = (bits >> (2 * j)) & 1, hi = (bits >> (2 * j + 1)) & 1;
          o[j] = (uint8_t)((lo ? 0x2 : 0xA) | ((hi ? 0x2 : 0xA) << 4));
        }
      }
  float *seeds = reinterpret_cast<float *>(dst + kL1MfmaWeights);
  for (int mt = 0; mt < 2; mt++)
    for (int h = 0; h < 2; h++)
      for (int reg = 0; reg < 16; reg++) {
        const int n = 32 * mt + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        // fire <=> mismatches < t <=> (576 - dot) / 2 < t <=> dot > 576 - 2t =: theta.  dot and theta are even, so
        // fire <=> dot - theta - 1 > 0; the clamp keeps never / always firing rows inside the exact f32 range
        long long t = (int)rows[n * ROW_DW];
        t = t < -1 ? -1 : (t > 578 ? 578 : t);
        seeds[(mt * 2 + h) * 16 + reg] = (float)(-(576 - 2 * t) - 1);
      }
}

hipError_t run_cnv(NetId net, const CnvLaunch &a) {
  if (a.n <= 0) return hipSuccess;
  const bool marked = a.done_flag && a.n == 1;  // (the call is timed and waited for by the completion mark: no event packets around it)
  if (a.t0 && !marked) (void)hipEventRecord(a.t0, a.stream);
  switch (net) {
    case NET_CNVW1A1: run_cnv_t<AR_XNOR, false>(a); break;
    case NET_CNVW1A2: run_cnv_t<AR_TB, true>(a); break;
    case NET_CNVW2A2:
      if (a.has_two) run_cnv_t<AR_TT, true, true>(a);
      else run_cnv_t<AR_TT, true>(a);
      break;
    default: return hipErrorInvalidValue;
  }
  if (a.t1 && !marked) (void)hipEventRecord(a.t1, a.stream);
  return hipGetLastError();
}

hipError_t run_lfc(NetId net, const LfcLaunch &a) {
  if (a.n <= 0) return hipSuccess;
  const long long n = a.n, narrow_limit = kNarrowLimitLfc;
  uint32_t *A = reinterpret_cast<uint32_t *>(a.buf0), *B = reinterpret_cast<uint32_t *>(a.buf1);
  uint64_t *A64 = reinterpret_cast<uint64_t *>(a.buf0), *B64 = reinterpret_cast<uint64_t *>(a.buf1);
  hipStream_t s = a.stream;
  if (n <= (net == NET_LFCW1A1 ? lfc_fused_max() : kLfcFusedMaxA2) && !a.events && a.last_stage >= kLfcStages - 1) {
    // small batch: the one-launch form, a block per group of IPB images (no per-stage events: there are no
    // stages).  A block costs ~8 us + ~1.6 us per further image whatever the batch, so the group is the
    // smallest that still fits the batch in one round of 256 blocks (profiles/r01_lfc_forms.txt).
    const int ipb = n <= 256 ? 1 : n <= 512 ? 2 : n <= 1024 ? 4 : 8;
    const dim3 g((unsigned)((n + ipb - 1) / ipb)), b(1024);
#define BNN_FUSED_P(K, I, P)                                                                                                              \
  do {                                                                                                                                    \
    unsigned *const done_ = (n == 1) ? a.done_flag : nullptr; /* (a one-block launch: the completion mark is meaningful) */            \
    if (a.t0 && a.t_dispatch && !done_) {                                                                                                 \
      hipExtLaunchKernelGGL((K<I, P>), g, b, 0, s, a.t0, a.t1, 0, a.images, a.words, a.classes, a.rows[0], a.rows[1], a.rows[2],          \
                            a.rows[3], (int)n, a.number_class, done_, a.done_seq);                                                        \
    } else {                                                                                                                              \
      if (a.t0 && !done_) (void)hipEventRecord(a.t0, s);                                                                                  \
      hipLaunchKernelGGL((K<I, P>), g, b, 0, s, a.images, a.words, a.classes, a.rows[0], a.rows[1], a.rows[2], a.rows[3], (int)n,         \
                         a.number_class, done_, a.done_seq);                                                                              \
      if (a.t1 && !done_) (void)hipEventRecord(a.t1, s);                                                                                  \
    }                                                                                                                                     \
  } while (0)
#define BNN_FUSED(K, I)                      \
  do {                                       \
    if (a.packed) BNN_FUSED_P(K, I, true);   \
    else BNN_FUSED_P(K, I, false);           \
  } while (0)
    if (net == NET_LFCW1A1) {
      if (ipb == 1) BNN_FUSED(k_lfc_fused, 1);
      else if (ipb == 2) BNN_FUSED(k_lfc_fused, 2);
      else if (ipb == 4) BNN_FUSED(k_lfc_fused, 4);
      else BNN_FUSED(k_lfc_fused, 8);
    } else if (net == NET_LFCW1A2) {
      if (ipb == 1) BNN_FUSED(k_lfc_fused_a2, 1);
      else if (ipb == 2) BNN_FUSED(k_lfc_fused_a2, 2);
      else if (ipb == 4) BNN_FUSED(k_lfc_fused_a2, 4);
      else BNN_FUSED(k_lfc_fused_a2, 8);
    } else {
      return hipErrorInvalidValue;
    }
#undef BNN_FUSED
#undef BNN_FUSED_P
    return hipGetLastError();
  }
  if (net == NET_LFCW1A1 && n <= lfc_block_max() && !a.events && a.last_stage >= kLfcStages - 1) {
    // mid-size batch: two 1024-thread blocks per CU, each walking all four layers over its share of the images
    const int ipb = (int)((n + 511) / 512);
    const dim3 g((unsigned)((n + ipb - 1) / ipb)), b(1024);
    auto *const kern = a.packed ? k_lfc_block_s<true> : k_lfc_block_s<false>;
    if (a.t0 && a.t_dispatch) {
      hipExtLaunchKernelGGL(kern, g, b, 0, s, a.t0, a.t1, 0, a.images, a.words, a.classes, A64, B64, a.rows[0], a.rows[1],
                            a.rows[2], a.rows[3], (int)n, a.number_class, ipb);
    } else {
      if (a.t0) (void)hipEventRecord(a.t0, s);
      hipLaunchKernelGGL(kern, g, b, 0, s, a.images, a.words, a.classes, A64, B64, a.rows[0], a.rows[1], a.rows[2], a.rows[3],
                         (int)n, a.number_class, ipb);
      if (a.t1) (void)hipEventRecord(a.t1, s);
    }
    return hipGetLastError();
  }
  if (a.t0) (void)hipEventRecord(a.t0, s);
  BNN_MARK(a.events, 0, s);
  // host-binarised input IS stage 0's output (13 words per image): layer 0 reads it where it lies; only the stage-output
  // hook asking for stage 0 itself has it copied into the workspace
  const uint64_t *L0in = A64;
  if (a.packed) {
    if (a.last_stage == 0) (void)hipMemcpyAsync(A64, a.images, (size_t)n * 13 * 8, hipMemcpyDeviceToDevice, s);
    else L0in = reinterpret_cast<const uint64_t *>(a.images);
  } else if (a.last_stage >= 0) {
    BNN_LAUNCH(k_lfc_binarize, grid_for(n * 13, 1), s, a.images, A64, (int)(n * 13));
  }
  BNN_MARK(a.events, 1, s);
  uint32_t *W32 = reinterpret_cast<uint32_t *>(a.words);
  if (net == NET_LFCW1A1) {
    if (a.last_stage >= 1) BNN_STAGE((k_vec_x<13, false, 1, 1>), (k_vec_x<13, false, 1, 1, 8>), n, 32, L0in, B, a.rows[0]);
    BNN_MARK(a.events, 2, s);
    if (a.last_stage >= 2) BNN_STAGE((k_vec_x<16, false, 1, 1>), (k_vec_x<16, false, 1, 1, 8>), n, 32, B64, A, a.rows[1]);
    BNN_MARK(a.events, 3, s);
    if (a.last_stage >= 3) BNN_STAGE((k_vec_x<16, false, 1, 1>), (k_vec_x<16, false, 1, 1, 8>), n, 32, A64, B, a.rows[2]);
    BNN_MARK(a.events, 4, s);
    if (a.last_stage >= 4) BNN_STAGE((k_vec_x<16, false, 1, 1>), (k_vec_x<16, false, 1, 1, 8>), n, 2, B64, W32, a.rows[3]);
    BNN_MARK(a.events, 5, s);
  } else if (net == NET_LFCW1A2) {
    if (a.last_stage >= 1) BNN_STAGE((k_vec<AR_XNOR, 13, true, false, 1, 1>), (k_vec<AR_XNOR, 13, true, false, 1, 1, 8>), n, 32, L0in, B, a.rows[0]);
    BNN_MARK(a.events, 2, s);
    if (a.last_stage >= 2) BNN_STAGE((k_vec<AR_TB, 16, true, false, 1, 1>), (k_vec<AR_TB, 16, true, false, 1, 1, 8>), n, 32, B64, A, a.rows[1]);
    BNN_MARK(a.events, 3, s);
    if (a.last_stage >= 3) BNN_STAGE((k_vec<AR_TB, 16, true, false, 1, 1>), (k_vec<AR_TB, 16, true, false, 1, 1, 8>), n, 32, A64, B, a.rows[2]);
    BNN_MARK(a.events, 4, s);
    if (a.last_stage >= 4) BNN_STAGE((k_vec<AR_TB, 16, false, false, 1, 1>), (k_vec<AR_TB, 16, false, false, 1, 1, 8>), n, 2, B64, W32, a.rows[3]);
    BNN_MARK(a.events, 5, s);
  } else {
    return hipErrorInvalidValue;
  }
  if (a.classes && a.last_stage >= 5) BNN_LAUNCH(k_lfc_decode, grid_for(n, 1), s, a.words, a.classes, (int)n, a.number_class);
  BNN_MARK(a.events, 6, s);
  if (a.t1) (void)hipEventRecord(a.t1, s);
  return hipGetLastError();
}

}  // namespace bnn
