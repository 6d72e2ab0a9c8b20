// packed_params.cpp -- bnn/params files -> device blob (host only, no HIP).
#include "packed_params.h"

#include <cstdio>
#include <cstring>

namespace bnn {
namespace {

// One PE memory file: `n` little-endian 64-bit words, zero-filled when short.
bool read_pe_file(const std::string &path, std::vector<uint64_t> &dst, size_t n) {
  dst.assign(n, 0);
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  std::vector<unsigned char> raw(n * 8, 0);
  const size_t got = std::fread(raw.data(), 1, raw.size(), f);
  std::fclose(f);
  // istream::read on a short file stores the bytes it got (rest of `e` stays 0) and
  // fails every later read: identical to decoding the zero-padded buffer.
  const size_t full = (got + 7) / 8;
  for (size_t i = 0; i < full; i++) {
    uint64_t e = 0;
    for (int k = 7; k >= 0; k--) e = (e << 8) | raw[i * 8 + k];
    dst[i] = e;
  }
  return true;
}

inline int32_t floor_div2(int32_t v) { return v >> 1; }  // arithmetic shift == floor for negatives

// logical weight (value domain) of neuron n, column j, and threshold i of neuron n, from the PE memories
struct LayerView {
  const LayerSpec *L;
  const std::vector<std::vector<uint64_t>> *w, *t;  // per PE
  int weight(int n, int j) const {
    const int pe = n % L->fold.pe, nf = n / L->fold.pe;
    const int sf_count = L->fold.wmem / L->fold.tmem;
    const int sf = j / L->fold.simd, s = j % L->fold.simd;
    const uint64_t word = (*w)[pe][(size_t)nf * sf_count + sf];
    if (L->wbits == 1) return ((word >> s) & 1) ? 1 : -1;
    const int f = (int)((word >> (2 * s)) & 3);  // ap_int<2>
    return f >= 2 ? f - 4 : f;
  }
  int32_t threshold(int n, int i) const {
    const int pe = n % L->fold.pe, nf = n / L->fold.pe;
    const uint64_t e = (*t)[pe][(size_t)nf * L->nthr + i];
    if (L->thr24) {  // ap_fixed<64,56> bits assigned to ap_fixed<24,16>: low 24 bits, sign-extended
      int32_t v = (int32_t)(e & 0xFFFFFF);
      return (v & 0x800000) ? v - 0x1000000 : v;
    }
    return (int16_t)(e & 0xFFFF);  // ap_uint<64> assigned to ap_int<16>
  }
};

void fill_row(const LayerSpec &L, const LayerView &F, int n, uint32_t *row, uint32_t rd) {
  const int MW = L.mw();
  for (uint32_t i = 0; i < rd; i++) row[i] = 0;
  int32_t T[2] = {0, 0};
  for (int i = 0; i < L.nthr && i < 2; i++) T[i] = F.threshold(n, i);
  if (L.nthr == 1) T[1] = T[0];
  for (int i = 0; i < 2; i++) {
    int32_t t;
    if (L.arith == AR_INT8) t = floor_div2(T[i]);
    else if (L.arith == AR_XNOR) t = L.signed_bb ? floor_div2(MW - T[i] + 1) : (MW - T[i]);
    else t = T[i];
    row[i] = (uint32_t)t;
  }
  if (L.arith == AR_INT8) {
    // tap tau = 3*(c*3+ky) + kx  <->  reference column (ky*3+kx)*3 + c
    for (int c = 0; c < 3; c++)
      for (int ky = 0; ky < 3; ky++)
        for (int kx = 0; kx < 3; kx++) {
          const int tau = 3 * (c * 3 + ky) + kx;
          const int wv = F.weight(n, (ky * 3 + kx) * 3 + c);
          row[2 + tau / 4] |= (uint32_t)(uint8_t)(int8_t)wv << (8 * (tau % 4));
        }
  } else if (L.wbits == 1) {
    // 1-bit weights: the row's MW raw bits (1 <=> +1) are the low SIMD bits of its SF memory words, in
    // order -- concatenate them instead of asking for one weight at a time
    uint64_t *wq = reinterpret_cast<uint64_t *>(row + 2);
    const int pe = n % L.fold.pe, nf = n / L.fold.pe, simd = L.fold.simd;
    const int sf_count = L.fold.wmem / L.fold.tmem;
    const uint64_t mask = simd >= 64 ? ~0ull : ((1ull << simd) - 1);
    const uint64_t *words = (*F.w)[pe].data() + (size_t)nf * sf_count;
    int bit = 0;
    for (int sf = 0; sf < sf_count; sf++, bit += simd) {
      const uint64_t v = words[sf] & mask;
      wq[bit >> 6] |= v << (bit & 63);
      if ((bit & 63) + simd > 64) wq[(bit >> 6) + 1] |= v >> (64 - (bit & 63));
    }
    if (L.arith == AR_TB)  // stored as the "weight is -1" plane
      for (int k = 0; k < MW / 64; k++) wq[k] = ~wq[k];
  } else {
    uint64_t *wq = reinterpret_cast<uint64_t *>(row + 2);
    const int kw = MW / 64;
    for (int k = 0; k < kw; k++) {
      uint64_t pos = 0, neg = 0, nz = 0;
      for (int b = 0; b < 64; b++) {
        const int wv = F.weight(n, k * 64 + b);
        if (wv > 0) pos |= 1ull << b;
        if (wv < 0) neg |= 1ull << b;
        if (wv != 0) nz |= 1ull << b;
      }
      if (L.arith == AR_XNOR) wq[k] = pos;
      else if (L.arith == AR_TB) wq[k] = neg;
      else { wq[2 * k] = neg; wq[2 * k + 1] = nz; }
    }
    if (L.arith == AR_TT) {
      // ap_int<2> also has the value -2 (field 0b10).  No shipped parameter set uses it, but a bit flip
      // makes one out of a 0 or a -1, and the reference then multiplies by -2.  Such columns are marked in
      // a third plane behind the two others; the kernels evaluate the row as if the weight were -1 and,
      // only when the row's flag is set, add the missing -a_j of every marked column.
      uint64_t *two = wq + 2 * kw;
      uint32_t any = 0;
      for (int k = 0; k < kw; k++) {
        uint64_t t = 0;
        for (int b = 0; b < 64; b++)
          if (F.weight(n, k * 64 + b) == -2) t |= 1ull << b;
        two[k] = t;
        any |= t != 0;
      }
      row[2 + 6 * kw] = any;
    }
  }
}

void layout_header(const NetSpec &net, PackedHeader &h) {
  h = PackedHeader{};
  h.magic0 = kBlobMagic0; h.magic1 = kBlobMagic1; h.version = kBlobVersion;
  h.net_id = (uint32_t)net.id; h.nlayers = (uint32_t)net.nlayers;
  uint32_t off = (sizeof(PackedHeader) + 255u) & ~255u;
  for (int l = 0; l < net.nlayers; l++) {
    const LayerSpec &L = net.L[l];
    h.layer[l].offset = off;
    h.layer[l].row_dwords = row_dwords_for(L);
    h.layer[l].rows = (uint32_t)L.mh();
    h.layer[l].kw = (L.arith == AR_INT8) ? 0 : (uint32_t)L.mw() / 64;
    off += h.layer[l].row_dwords * 4 * h.layer[l].rows;
    off = (off + 255u) & ~255u;
  }
  if (net.is_cnv) {
    h.l0_mfma_offset = off;
    off = (off + kL0MfmaBytes + 255u) & ~255u;
  }
  h.total_bytes = off + 256;  // tail slack: wide scalar loads may run past the last row
}

// layer 0 of the CNV nets as an MFMA operand (packed_params.h)
void fill_l0_mfma(const NetSpec &net, const RawParams &raw, uint8_t *dst) {
  const LayerSpec &L = net.L[0];
  const LayerView F{&L, &raw.w[0], &raw.t[0]};
  for (int which = 0; which < 2; which++) {
    int8_t *a = reinterpret_cast<int8_t *>(dst) + which * 64 * 32;
    for (int n = 0; n < 64; n++) {
      for (int k = 0; k < 32; k++) a[n * 32 + k] = 0;
      for (int c = 0; c < 3; c++)
        for (int ky = 0; ky < 3; ky++)
          for (int kx = 0; kx < 3; kx++) a[n * 32 + 3 * (c * 3 + ky) + kx] = (int8_t)F.weight(n, (ky * 3 + kx) * 3 + c);
      const int32_t T = F.threshold(n, (which && L.nthr > 1) ? 1 : 0);
      const int32_t t = floor_div2(T);
      // clamp to the reachable range of the dot product (same decisions): |dot| <= 27*128 for weights in
      // {-1,0,+1}; ap_int<2> weights can be -2 after a bit flip (cnvW2A2), which doubles the range.  With
      // |v| <= 6914, a1 = floor((v + 32) / 64) still fits int8.
      const int32_t lim = (L.wbits == 2) ? 2 * 3456 : 3456;
      const int32_t tc = t > lim ? lim : (t < -lim - 1 ? -lim - 1 : t);
      const int32_t v = -tc - 1;
      const int32_t a1 = (v + 32 + 64 * 128) / 64 - 128;  // floor((v + 32) / 64)
      a[n * 32 + 27] = (int8_t)(v - 64 * a1);
      a[n * 32 + 28] = (int8_t)a1;
    }
  }
  // the same numbers in the operand form of k_conv0_tile (packed_params.h)
  const int8_t *pix = reinterpret_cast<const int8_t *>(dst);
  int8_t *big = reinterpret_cast<int8_t *>(dst) + kL0MfmaTileOffset, *small = big + kL0MfmaBigBytes;
  static const int run_of[2][4] = {{0, 1, 2, 6}, {3, 4, 5, 8}};  // run = c*3 + ky; run 7 = (2,1) goes to the K = 16 product
  for (int ct = 0; ct < 2; ct++)
    for (int i = 0; i < 32; i++) {
      const int g = i >> 3, hh = (i >> 2) & 1, q = i & 3, n = 32 * ct + 16 * hh + 4 * g + q;
      for (int h = 0; h < 2; h++) {
        int8_t *b = big + ((ct * 32 + i) * 2 + h) * 16;
        for (int s = 0; s < 4; s++)
          for (int k = 0; k < 4; k++) b[4 * s + k] = k < 3 ? pix[n * 32 + 3 * run_of[h][s] + k] : 0;
        for (int which = 0; which < 2; which++) {
          int8_t *sm = small + (((which * 2 + ct) * 32 + i) * 2 + h) * 8;
          const int8_t *row = pix + which * 64 * 32 + n * 32;
          for (int k = 0; k < 8; k++) sm[k] = 0;
          if (h == 0) { sm[0] = row[21]; sm[1] = row[22]; sm[2] = row[23]; }
          else { sm[0] = row[27]; sm[1] = row[28]; }
        }
      }
    }
}

}  // namespace

uint32_t row_dwords_for(const LayerSpec &L) {
  const uint32_t kw = (uint32_t)L.mw() / 64;
  switch (L.arith) {
    case AR_INT8: return 12;
    case AR_XNOR: return 2 + 2 * kw;
    case AR_TB: return 2 + 2 * kw;
    case AR_TT: return 4 + 6 * kw;  // t0, t1, (sign, non-zero) planes, "weight is -2" plane, flag, pad
  }
  return 0;
}

std::string read_raw_params(const NetSpec &net, const std::string &dir, RawParams &raw) {
  for (int l = 0; l < net.nlayers; l++) {
    const LayerSpec &L = net.L[l];
    raw.w[l].assign(L.fold.pe, {});
    raw.t[l].assign(L.fold.pe, {});
    for (int pe = 0; pe < L.fold.pe; pe++) {
      const std::string stem = dir + "/" + std::to_string(l) + "-" + std::to_string(pe);
      if (!read_pe_file(stem + "-weights.bin", raw.w[l][pe], (size_t)L.fold.wmem))
        return "Could not open file " + stem + "-weights.bin";
      if (L.nthr > 0 && !read_pe_file(stem + "-thres.bin", raw.t[l][pe], (size_t)L.fold.tmem * L.nthr))
        return "Could not open file " + stem + "-thres.bin";
    }
  }
  return "";
}

// rows of the 2-bit-weight layers that hold a weight of -2 (their flag dword, see fill_row): the runtime
// launches the -2-aware kernel variants only while this is non-zero
int count_two_rows(const NetSpec &net, const std::vector<uint8_t> &blob) {
  PackedHeader h;
  std::memcpy(&h, blob.data(), sizeof(h));
  int count = 0;
  for (int l = 0; l < net.nlayers; l++) {
    if (net.L[l].arith != AR_TT) continue;
    const uint32_t rd = h.layer[l].row_dwords, kw = h.layer[l].kw;
    const uint32_t *rows = reinterpret_cast<const uint32_t *>(blob.data() + h.layer[l].offset);
    for (uint32_t n = 0; n < h.layer[l].rows; n++) count += rows[(size_t)n * rd + 2 + 6 * kw] != 0;
  }
  return count;
}

void pack_blob(const NetSpec &net, const RawParams &raw, std::vector<uint8_t> &blob) {
  PackedHeader h;
  layout_header(net, h);
  blob.assign(h.total_bytes, 0);
  std::memcpy(blob.data(), &h, sizeof(h));
  for (int l = 0; l < net.nlayers; l++) {
    const LayerSpec &L = net.L[l];
    const LayerView F{&L, &raw.w[l], &raw.t[l]};
    const uint32_t rd = h.layer[l].row_dwords;
    uint32_t *rows = reinterpret_cast<uint32_t *>(blob.data() + h.layer[l].offset);
    for (int n = 0; n < L.mh(); n++) fill_row(L, F, n, rows + (size_t)n * rd, rd);
  }
  if (h.l0_mfma_offset) fill_l0_mfma(net, raw, blob.data() + h.l0_mfma_offset);
}

void repack_row(const NetSpec &net, const RawParams &raw, int l, int n, std::vector<uint8_t> &blob, size_t *offset,
                size_t *bytes) {
  PackedHeader h;
  std::memcpy(&h, blob.data(), sizeof(h));
  const LayerSpec &L = net.L[l];
  const LayerView F{&L, &raw.w[l], &raw.t[l]};
  const uint32_t rd = h.layer[l].row_dwords;
  const size_t off = h.layer[l].offset + (size_t)n * rd * 4;
  fill_row(L, F, n, reinterpret_cast<uint32_t *>(blob.data() + off), rd);
  *offset = off;
  *bytes = (size_t)rd * 4;
  if (l == 0 && h.l0_mfma_offset) {  // keep the matrix-pipe copy of layer 0 in step: report one span covering both
    fill_l0_mfma(net, raw, blob.data() + h.l0_mfma_offset);
    *offset = h.layer[0].offset;
    *bytes = (size_t)h.l0_mfma_offset + kL0MfmaBytes - h.layer[0].offset;
  }
}

std::string pack_params_from_dir(const NetSpec &net, const std::string &dir, std::vector<uint8_t> &blob) {
  RawParams raw;
  const std::string e = read_raw_params(net, dir, raw);
  if (!e.empty()) return e;
  pack_blob(net, raw, blob);
  return "";
}

size_t blob_bytes(const NetSpec &net) {
  PackedHeader h;
  layout_header(net, h);
  return h.total_bytes;
}

std::string validate_blob(const NetSpec &net, const void *blob, size_t bytes) {
  if (bytes < sizeof(PackedHeader)) return "packed params: blob shorter than its header";
  PackedHeader h, want;
  std::memcpy(&h, blob, sizeof(h));
  if (h.magic0 != kBlobMagic0 || h.magic1 != kBlobMagic1) return "packed params: bad magic";
  if (h.version != kBlobVersion) return "packed params: version mismatch";
  if (h.net_id != (uint32_t)net.id) return std::string("packed params: blob is not for network ") + net.name;
  // The layout is a function of the network alone: the bytes come from outside (broadcast, file), and every
  // field of the header is later used as an offset or a stride on the host and on the device -- so each one
  // must be exactly what this library would have written, not merely plausible.
  layout_header(net, want);
  if (h.nlayers != want.nlayers || h.total_bytes != want.total_bytes || bytes != want.total_bytes) return "packed params: size mismatch";
  if (h.l0_mfma_offset != want.l0_mfma_offset || h.reserved1 != 0) return "packed params: layer-0 table offset mismatch";
  for (int l = 0; l < 9; l++) {
    const PackedLayer &p = h.layer[l], &q = want.layer[l];
    if (p.row_dwords != q.row_dwords || p.rows != q.rows || p.kw != q.kw) return "packed params: layer shape mismatch";
    if (p.offset != q.offset) return "packed params: layer offset mismatch";
  }
  return "";
}

}  // namespace bnn
