// AddressSanitizer / UBSan run of the product's HOST code (the GPU pool has no device sanitizers):
// parameter reading and packing, row re-packing under thousands of faults, blob validation on
// corrupted input, the thumbnail rule and the Lanczos coefficient tables.  Built and run by
// tests/test_host_sanitize.py with g++ -fsanitize=address,undefined; any finding aborts.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "faults.h"
#include "pack_inputs.h"
#include "packed_params.h"
#include "resample.h"
#include "topology.h"

using namespace bnn;

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const std::string root = argv[1];  // .../bnn/params
  const struct { NetId id; const char *dir; } sets[] = {
      {NET_CNVW1A1, "cifar10/cnvW1A1"}, {NET_CNVW1A2, "cifar10/cnvW1A2"}, {NET_CNVW2A2, "cifar10/cnvW2A2"},
      {NET_LFCW1A1, "mnist/lfcW1A1"},   {NET_LFCW1A2, "mnist/lfcW1A2"},   {NET_CNVW1A1, "road-signs/cnvW1A1"}};
  long checked = 0;
  for (const auto &s : sets) {
    const NetSpec &net = net_spec(s.id);
    RawParams raw;
    const std::string e = read_raw_params(net, root + "/" + s.dir, raw);
    if (!e.empty()) { std::fprintf(stderr, "%s\n", e.c_str()); return 1; }
    std::vector<uint8_t> blob;
    pack_blob(net, raw, blob);
    if (!validate_blob(net, blob.data(), blob.size()).empty()) return 1;
    // every fault mode, many faults: apply + re-pack the touched row (in-bounds reads/writes of raw and blob)
    for (int target = -1; target <= 1; target++)
      for (int ws : {1, 3, 8, 16}) {
        std::vector<Fault> plan;
        if (!plan_faults(net, 17 + ws, 1000, 400, ws, target, nullptr, 0, plan).empty()) return 1;
        for (const Fault &f : plan) {
          const int row = apply_fault(net, raw, f);
          if (row < 0) continue;
          size_t off = 0, bytes = 0;
          repack_row(net, raw, f.layer, row, blob, &off, &bytes);
          if (off + bytes > blob.size()) return 1;
          checked++;
        }
      }
    // out-of-range fault records are rejected, not applied
    if (apply_fault(net, raw, Fault{0, 0, 99, 0, 0, 0, 0, 1}) >= 0) return 1;
    if (apply_fault(net, raw, Fault{0, 1, 0, 1 << 20, 0, 0, 0, 1}) >= 0) return 1;
    // corrupted / truncated blobs are refused
    std::vector<uint8_t> bad(blob);
    bad[0] ^= 0xFF;
    if (validate_blob(net, bad.data(), bad.size()).empty()) return 1;
    if (validate_blob(net, blob.data(), blob.size() / 2).empty()) return 1;
    if (validate_blob(net, blob.data(), 8).empty()) return 1;
  }
  // binarizeAndPack on the host (what the LFC host paths run on their worker threads): exact-size heap buffers at every
  // byte alignment -- a read past the 784th pixel of the last image or a write past its 13th word is a finding --, both
  // forms against the definition
  {
    for (size_t n : {(size_t)0, (size_t)1, (size_t)2, (size_t)7, (size_t)80, (size_t)81}) {
      for (size_t soff = 0; soff < 4; soff++)
        for (size_t doff : {(size_t)0, (size_t)8, (size_t)3}) {
          std::vector<uint8_t> src(soff + n * kLfcPixels + (n == 0)), dst(doff + n * kLfcWords * 8 + (n == 0)), dst2(dst.size());  // (n = 0: a non-null pointer to nothing)
          for (size_t i = 0; i < src.size(); i++) src[i] = (uint8_t)((i * 131 + (i >> 3) * 29 + soff) & 0xFF);
          binarize_pack(src.data() + soff, n, reinterpret_cast<uint64_t *>(dst.data() + doff));
          binarize_pack_portable(src.data() + soff, n, reinterpret_cast<uint64_t *>(dst2.data() + doff));
          if (std::memcmp(dst.data() + doff, dst2.data() + doff, n * kLfcWords * 8) != 0) return 1;
          for (size_t i = 0; i < n; i++)
            for (int b = 0; b < kLfcWords * 64; b++) {
              uint64_t w;
              std::memcpy(&w, dst.data() + doff + (i * kLfcWords + b / 64) * 8, 8);
              const int want = b < kLfcPixels ? (src[soff + i * kLfcPixels + b] >= 128) : 0;
              if ((int)((w >> (b % 64)) & 1) != want) return 1;
            }
          checked += (long)n;
        }
    }
    std::printf("binarize_pack: %s form\n", binarize_pack_isa());
  }
  // a missing directory is an error string, not a crash
  {
    std::vector<uint8_t> blob;
    if (pack_params_from_dir(net_spec(NET_LFCW1A1), root + "/nowhere", blob).empty()) return 1;
  }
  // resampling tables: every bound within the axis, every row sums to ~2^22
  for (int in = 1; in <= 5000; in += (in < 200 ? 1 : 97)) {
    int ow = in, oh = in;
    (void)thumbnail_size(in, 2 * in + 1, 32, &ow, &oh);
    if (ow < 1 || oh < 1 || ow > 32 || (oh > 32 && 2 * in + 1 > 32)) return 1;
    for (int out : {1, 7, 32}) {
      if (out > in) continue;
      std::vector<int32_t> kk, b;
      const int ks = lanczos_coeffs(in, out, kk, b);
      for (int x = 0; x < out; x++) {
        if (b[2 * x] < 0 || b[2 * x + 1] < 1 || b[2 * x] + b[2 * x + 1] > in || b[2 * x + 1] > ks) return 1;
        long sum = 0;
        for (int k = 0; k < b[2 * x + 1]; k++) sum += kk[(size_t)x * ks + k];
        if (sum < (1 << 22) - 2048 || sum > (1 << 22) + 2048) return 1;
        checked++;
      }
    }
  }
  std::printf("host sanitize run ok: %ld checks\n", checked);
  return 0;
}
