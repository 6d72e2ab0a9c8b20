// runtime.hip -- host side of the MI355X runtime: device state, workspace,
// batch chunking, timing, output decode, and the C ABI of include/bnn_mi355x.h.
//
// Host-side reference code this replaces (the RAWHLS half of
// bnn/src/library/host/foldedmv-offload.{h,cpp} and rawhls-offload.cpp, plus
// the per-network bnn/src/network/<net>/sw/main_python.cpp):
//   FoldedMVInit / FoldedMVDeinit            -> Workspace (HBM buffers, lazily sized)
//   FoldedMVLoadLayerMem / DoMemInit         -> packed_params.cpp + one hipMemcpy
//   parse_cifar10 / parse_mnist_images       -> open_image_file + k_strip_records (file streamed to HBM as it lies on disk)
//   quantiseAndPack / binarizeAndPack        -> done on the GPU (k_conv0 / k_lfc_binarize)
//   BlackBoxJam(.., numReps)                 -> run_cnv / run_lfc (kernels.hip)
//   copyFromLowPrecBuffer + argmax / log2    -> k_fclast / host decode below
//
// There is no CPU compute path in this library: without a usable HIP device
// every inference entry point fails loudly (message on stderr, NULL / -1).
#include <hip/hip_runtime.h>

#include <fcntl.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <sys/stat.h>
#include <sys/uio.h>
#include <sched.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <memory>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/bnn_mi355x.h"
#include "faults.h"
#include "kernels.h"
#include "preprocess.h"
#include "resample.h"
#include "packed_params.h"
#include "pack_inputs.h"
#include "topology.h"

#ifndef BNN_NETWORK
#error "compile with -DBNN_NETWORK=NET_CNVW1A1 (or another bnn::NetId)"
#endif
// BNN_VARIANT: the library is built under the name of one of the fork's hardened overlays
// (cnvW1A1-TMR, ...-interleaved, ...): same compute as BNN_NETWORK, same parameter files; their
// fault model (replicated / bit-interleaved parameter memories) is not modelled.

namespace bnn {
namespace {

constexpr int kMaxChunk = 131072;  // images per pass through the stages
constexpr int kForkMin = 16384;     // images: a device-pointer pass of a CNV net of this size and more forks over the two compute lanes
constexpr int kStageSlots = 4;      // HBM staging buffers of the host paths: two on one compute lane, four on two
// Events that only TIME device work: a device-scope release at the record point instead of a flush to system scope
// ("useful to obtain more precise timings of commands between events", hip_runtime_api.h).  Nothing on the host reads
// results on the strength of these events: every entry point fetches them with a copy + stream synchronisation.
constexpr unsigned kTimeEventFlags = hipEventReleaseToDevice;
constexpr int kHostChunk = 32768;  // host-buffer / file path: H2D of chunk i+1 overlaps the stages of chunk i (largest chunk; LFC nets)
constexpr int kHostChunkCnv = 16384;  // ... the CNV nets since the chunks run on two compute lanes (below)
constexpr int kHeadChunk = 512;    // ... the first chunk: what the stages wait for before anything runs
constexpr int kFastGrowthBelow = 4096;  // ... chunks double up to here, then grow by half
// Pinned, device-mapped I/O block (ensure_io): small calls cross the PCIe link by the kernels' own loads and stores --
// no hipMemcpy, no staging, one wait.  Input area: a single CIFAR record (body 16-byte aligned) or up to kDirectMaxLfc
// host-binarised MNIST images; result area: classes / raw words of up to kMappedResMax images, scores of kMappedScoresMax.
constexpr int kDirectMaxLfc = 1024;   // images: an LFC call up to here is binarised by the calling thread and read in place
constexpr int kDirectMaxCnv = 32;     // CIFAR images from a host buffer (classify_image(s) of a few pictures: whole call 38-46 us against 46-59 through a copy,
                                      // profiles/r04_small_n_latency.txt; beyond ~64 the kernels' reads over the link cost more than the copy); from a FILE: one
constexpr int kMappedResMax = 32768, kMappedScoresMax = 256;
constexpr size_t kIoInBytes = 256u << 10;
constexpr size_t kIoClassesOff = kIoInBytes, kIoWordsOff = kIoClassesOff + (size_t)kMappedResMax * 4,
                 kIoScoresOff = kIoWordsOff + (size_t)kMappedResMax * 8, kIoDoneOff = kIoScoresOff + (size_t)kMappedScoresMax * 128,
                 kIoBytes = kIoDoneOff + 64;  // (the last 64 bytes: the completion word of single-image calls timed by the host)
static_assert((size_t)kDirectMaxLfc * kLfcWords * 8 <= kIoInBytes && 16 + 3073 <= kIoInBytes && 16 + (size_t)kDirectMaxCnv * 3072 <= kIoInBytes,
              "input area too small");

// Chunk boundaries of a host-buffer / file call: base[c] .. base[c+1] are the images of chunk c.
// The pipeline is copy | stages, double buffered per compute lane.  With equal chunks the first copy (32 768 CIFAR records
// = 100 MB: ~3 ms of pread + H2D) runs with the GPU idle, so the chunks ramp up -- by a factor that keeps the NEXT chunk's
// transfer about as long as THIS chunk's stages, or the stages starve at every step of the ramp (measured with a device
// timeline, profiles/r03_host_path_timelines.txt: doubling 2 048 -> 32 768 left 0.3-0.8 ms holes).  Both sources arrive
// faster than the stages consume them (a buffer in host memory at 54 GB/s = 17.6 M CIFAR images/s, a file through the
// pinned ring and two DMA queues at ~46 GB/s = 15 M/s, against 12.5 M/s of stages at full-size chunks), so a call is
// compute-bound: what ends it is the last chunk's stages whatever their size -- no ramp down.  Round 4, on the reference's
// own call size (a 10 000-record test-set file: 30 MB, 0.94 ms of stages): the first chunk's transfer is the one thing
// nothing hides, and small chunks run their stages at 7-10 M images/s -- half the link's rate -- so the head of the plan is
// 512 images and DOUBLES up to 4 096 (the next transfer still fits behind the current stages), then grows by half up to
// 16 384 (CNV; the LFC nets: four times these sizes in images, an MNIST image is a quarter of a CIFAR one):
// 10 000 records from a file 1.59 -> 1.25 ms, from a buffer 1.16 -> 1.14 ms on the plan alone
// (profiles/r04_small_call_plan_sweep.txt; r03_chunk_plan_sweep.txt and r03_two_lanes_plan_sweep.txt for the large calls).
// The chunks alternate over two compute lanes ("Two compute lanes" below): one chunk's launch gaps and tails are filled by
// the other's kernels.
// BNN_MI355X_CHUNKS=head:tail:max[:growth%] overrides the sizes (0 = no ramp at that end; a growth given here applies
// to every step; tuning / A-B runs).
std::vector<int> plan_chunks(int n, bool single, bool /*from_file*/) {
  // (sizes are tuned in bytes on CIFAR records: an MNIST image is a quarter of one)
  // The LFC nets' host paths ship binarised words (104 bytes per image, "binarizeAndPack on the host" below): what a chunk
  // waits for is its share of the binarising cores' work, not the link, and the one-launch kernel is at its best on large
  // chunks -- 8 192 images first, doubling (131 072 images: 102 M images/s with 2 048 first, 115-119 M with 8 192;
  // 10 000 images as ONE chunk 62 M/s against 53 M/s in three; profiles/r04_small_call_plan_sweep.txt).
  const bool cnv = net_spec(BNN_NETWORK).is_cnv;
  const int scale = cnv ? 1 : 4;
  // ... and a CNV call of 8 192 ... 32 767 images ramps DOWN as well: at that size the bytes arrive about as fast as small
  // chunks' stages consume them, and what follows the last byte is the last chunk's stages.  Interleaved A/B
  // (tools/plan_ab.py, profiles/r04_plan_ab_interleaved.txt, medians of 20-40 rounds): 10 000 images from a file 1.243 ->
  // 1.228 ms, from a buffer 1.137 -> 1.132; 20 000 images 2.297 -> 2.251 / 2.065 -> 2.059; 5 000 images LOSE 3-4 % to the two
  // extra chunks, 131 072 images 3 %: those keep the one-sided ramp.
  int head = cnv ? kHeadChunk : 8192, tail = (cnv && n >= 8192 && n < 32768) ? kHeadChunk : 0, big = cnv ? kHostChunkCnv : kHostChunk, growth = cnv ? 0 : 200;
  if (const char *e = std::getenv("BNN_MI355X_CHUNKS")) {
    int h = 0, t = 0, b = 0, g = 150;
    const int got = std::sscanf(e, "%d:%d:%d:%d", &h, &t, &b, &g);
    // (head and tail are clamped to `max`: no field of the override can ask for a chunk above the activation workspace)
    if (got >= 3 && b >= 256 && b <= kMaxChunk && h >= 0 && t >= 0 && g > 100 && g <= 400) { head = h < b ? h : b; tail = t < b ? t : b; big = b; growth = g; }
  }
  std::vector<int> front, back;
  int rem = n;
  // nothing worth overlapping: one chunk -- but never one above the largest chunk of the plan (`single`, the stage-output
  // hook, is limited to kHostChunk images by its caller)
  if ((single && n <= kMaxChunk) || (n <= 2LL * head && n <= big)) {
    front.push_back(n);
    rem = 0;
  }
  auto grow = [&](int s) {
    const int pct = growth ? growth : (s < kFastGrowthBelow * scale ? 200 : 150);
    const long long g = ((long long)s * pct / 100 + 255) & ~255LL;  // whole 256-image blocks
    return (int)(g < big ? g : big);
  };
  int sf = head > 0 ? head : big, sb = tail > 0 ? tail : big;
  if (sf > big) sf = big;
  while (rem > 0) {
    int t = sf < rem ? sf : rem;
    front.push_back(t);
    rem -= t;
    sf = grow(sf);
    if (rem > 0 && sb < big) {
      t = sb < rem ? sb : rem;
      back.push_back(t);
      rem -= t;
      sb = grow(sb);
    }
  }
  // a small remainder joins the chunk in front of it (a chunk of a few hundred images costs nine launches all the same)
  // (with a ramp at both ends the remainder is the chunk in the middle)
  if (front.size() >= 2 && front.back() * 2 < front[front.size() - 2] && front.back() + front[front.size() - 2] <= big) {
    front[front.size() - 2] += front.back();
    front.pop_back();
  }
  std::vector<int> base;
  base.push_back(0);
  for (int t : front) base.push_back(base.back() + t);
  for (size_t i = back.size(); i-- > 0;) base.push_back(base.back() + back[i]);
  return base;
}
int largest_chunk(const std::vector<int> &base) {
  int m = 0;
  for (size_t c = 0; c + 1 < base.size(); c++) m = base[c + 1] - base[c] > m ? base[c + 1] - base[c] : m;
  return m;
}

struct Runtime {
  const NetSpec &spec = net_spec(BNN_NETWORK);
  int device = -1;  // -1: whatever HIP's current device is
  std::string err;
  // parameters
  RawParams raw;  // the reference-layout memories (empty when the blob was imported)
  std::vector<uint8_t> blob;
  void *d_blob = nullptr;
  size_t d_blob_bytes = 0;
  uint64_t fault_seed = 0;  // 0: std::random_device, like the reference
  int debug_last_stage = -1;  // >= 0: stop after this stage (bnn_mi355x_debug_stage_output)
  std::vector<Fault> last_faults;
  const uint32_t *rows[9] = {};
  const uint8_t *l0_mfma = nullptr;  // layer 0 as an MFMA operand (CNV nets), unless BNN_MI355X_L0=valu
  uint8_t *d_l1_mfma = nullptr;      // cnvW1A1, BNN_MI355X_L1=mfma only: layer 1 as FP4 MFMA operands (side experiment)
  bool l1_mfma = false, l1_literal = false;  // BNN_MI355X_L1=mfma / =lds (comparison figures, never the default)
  bool warmed = false;  // warm_up() has run since the last deinit()
  int two_rows = 0;  // rows holding a weight of -2 (2-bit-weight net under fault injection): kernels.hip, two_extra
  // workspace
  int cap = 0;
  void *buf0 = nullptr, *buf1 = nullptr;
  // second compute lane of the host paths ("Two compute lanes" below): its own workspace and stream
  int cap2 = 0;
  void *buf0b = nullptr, *buf1b = nullptr;
  hipStream_t stream2 = nullptr;
  hipEvent_t lane2_done = nullptr;
  hipStream_t feed_aux = nullptr;  // the pinned ring's second DMA queue (Feeder::aux), created with the streams above
  // host-buffer path: two image staging buffers in HBM (ping-pong) + results for the whole call
  int stage_cap = 0;
  // (kStageSlots buffers exist once a call has run on two lanes: each lane consumes one while the next chunk of each arrives)
  int stage_slots = 0;
  uint8_t *d_images[kStageSlots] = {};
  size_t res_cap = 0;
  int16_t *d_scores = nullptr;
  int32_t *d_classes = nullptr;
  uint64_t *d_words = nullptr;
  hipStream_t stream = nullptr, copy_stream = nullptr;
  // pinned host memory the GPU addresses directly (kIo* above): h_io as the CPU sees it, d_io as the kernels do
  uint8_t *h_io = nullptr, *d_io = nullptr;
  hipEvent_t io_t0 = nullptr, io_t1 = nullptr;  // device time of a direct call (system-scope release: the host reads what the kernels wrote)
  unsigned io_seq = 0;                           // completion marks handed out so far (BNN_MI355X_DIRECT_TIMING=host)
  std::vector<uint8_t> h_scratch;               // direct LFC calls from a file: the pixels on their way to the binariser
  // results of calls above kMappedResMax images: ONE D2H at the end of the call into pinned memory (a pageable destination
  // would be staged by the runtime), handed to the caller / decoded from there
  int32_t *h_classes = nullptr;
  uint64_t *h_words = nullptr;
  size_t h_classes_cap = 0, h_words_cap = 0;
  // The activation workspace (buf0/buf1, d_words) is shared by every call.  Host-path calls drain r.stream
  // before they return; bnn_mi355x_inference_device leaves work in flight on the CALLER's stream, so it marks
  // the end of that work with ws_event and the next call on any other stream waits for it first.
  hipEvent_t ws_event = nullptr;
  hipStream_t ws_last = nullptr;  // the caller's stream of that call (may be the null stream: hence the flag)
  bool ws_pending = false;
  // (both workspaces since such a call may fork over the two lanes: the library's own two streams each remember the last
  // hand-over they have waited for, a caller's stream waits whenever one is pending)
  uint64_t ws_gen = 0, ws_seen[2] = {0, 0};
  hipEvent_t fork_ev = nullptr;
  hipEvent_t copied[kStageSlots] = {}, consumed[kStageSlots] = {};
  std::vector<hipEvent_t> time_events;
  // file path: records as they lie on disk, two host chunks (filled by reader threads) and two HBM chunks
  size_t file_cap = 0;
  std::unique_ptr<uint8_t[]> h_file[2];
  uint8_t *d_file[2] = {nullptr, nullptr};
  hipEvent_t file_sent[2] = {nullptr, nullptr};
  uint8_t *d_all = nullptr;  // a whole input file's images, resident (fault campaigns)
  size_t all_cap = 0;
  // picture -> CIFAR record (bnn_mi355x_images_to_cifar): source picture, horizontal-pass output,
  // coefficient tables, records
  size_t pp_src_cap = 0, pp_tmp_cap = 0, pp_coef_cap = 0, pp_rec_cap = 0;
  uint8_t *d_pp_src = nullptr, *d_pp_tmp = nullptr, *d_pp_rec = nullptr;
  int32_t *d_pp_coef = nullptr;
  // optional per-stage profiling (bnn_mi355x_profile): one event set per enqueued chunk
  bool profiling = false;
  std::vector<std::vector<hipEvent_t>> prof_sets;
  size_t prof_used = 0;
};

Runtime &rt() {
  static Runtime r;
  return r;
}

int fail(const std::string &msg) {
  rt().err = msg;
  std::fprintf(stderr, "bnn-mi355x[%s]: %s\n", rt().spec.name, msg.c_str());
  return -1;
}

// BNN_MI355X_TRACE=1: host-side time stamps of a host-data call (microseconds since its start) on stderr -- where a call's
// wall time goes that no device timeline shows (tools/small_call_sweep.py prints them next to the rates)
struct CallTrace {
  const bool on = std::getenv("BNN_MI355X_TRACE") != nullptr;
  std::chrono::steady_clock::time_point t0;
  std::vector<std::pair<std::string, double>> marks;
  double now() const { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
  void start() {
    if (!on) return;
    t0 = std::chrono::steady_clock::now();
    marks.clear();
  }
  void mark(const char *what, int k = -1) {
    if (!on) return;
    marks.emplace_back(k < 0 ? std::string(what) : std::string(what) + "[" + std::to_string(k) + "]", now());
  }
  void dump(const char *title) {
    if (!on) return;
    std::string line = std::string("bnn-mi355x trace ") + title + ":";
    char buf[64];
    for (auto &m : marks) {
      std::snprintf(buf, sizeof buf, " %s=%.0f", m.first.c_str(), m.second);
      line += buf;
    }
    std::fprintf(stderr, "%s\n", line.c_str());
    marks.clear();
  }
};
CallTrace &trace() {
  static CallTrace t;
  return t;
}

#define HIP_OK(expr)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

int bind_device() {
  Runtime &r = rt();
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail("no HIP device available: this runtime has no CPU fallback");
  if (r.device >= 0) HIP_OK(hipSetDevice(r.device));
  else HIP_OK(hipGetDevice(&r.device));  // first use: the calling thread's current device is this library's from now on
  if (!r.stream) {
    // Four streams in all with the feeder's second DMA queue (Feeder::aux) -- and no more: the runtime maps a process's
    // streams onto four hardware queues, a fifth stream shares one with an earlier stream and the two then serialise.
    // Found the hard way in round 4: a third compute lane (its stream created here) put Feeder::aux on this stream's
    // queue and cost the file path 20 % (131 072 records 11.3 -> 13.6 ms) before a single chunk ran on it
    // (profiles/r04_third_lane_experiment.txt).
    HIP_OK(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
    HIP_OK(hipStreamCreateWithFlags(&r.copy_stream, hipStreamNonBlocking));
    HIP_OK(hipStreamCreateWithFlags(&r.stream2, hipStreamNonBlocking));
    // (the feeder's second DMA queue is created HERE, right behind the other three, although the feeder itself comes into
    // being only with the first call that needs it: streams the application creates in between would otherwise decide
    // which of this library's streams it shares a hardware queue with.  BNN_MI355X_FEEDER_STREAMS=1: one DMA queue.)
    {
      const char *es = std::getenv("BNN_MI355X_FEEDER_STREAMS");
      if (!es || std::atoi(es) == 2) HIP_OK(hipStreamCreateWithFlags(&r.feed_aux, hipStreamNonBlocking));
    }
    HIP_OK(hipEventCreateWithFlags(&r.lane2_done, hipEventDisableTiming));
    for (int i = 0; i < kStageSlots; i++) {
      HIP_OK(hipEventCreateWithFlags(&r.copied[i], hipEventDisableTiming));
      HIP_OK(hipEventCreateWithFlags(&r.consumed[i], hipEventDisableTiming));
    }
    HIP_OK(hipEventCreateWithFlags(&r.ws_event, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&r.fork_ev, hipEventDisableTiming));
  }
  return 0;
}

// On a failing exit nothing may still be queued that reads the caller's buffers, the host chunks or the patch
// images, or that writes the caller's result arrays: the caller is free to release them as soon as it sees
// the error.  (A successful exit has waited for its streams anyway.)
void drain_feeder();  // (the pinned ring's second DMA queue, below)
struct DrainOnFailure {
  bool armed = true;
  void ok() { armed = false; }
  ~DrainOnFailure() {
    if (!armed) return;
    Runtime &r = rt();
    if (r.stream) (void)hipStreamSynchronize(r.stream);
    if (r.stream2) (void)hipStreamSynchronize(r.stream2);
    if (r.copy_stream) (void)hipStreamSynchronize(r.copy_stream);
    drain_feeder();  // DMAs still reading the pinned ring: the next job's workers would refill it under them
  }
};

int upload_blob() {
  Runtime &r = rt();
  if (bind_device()) return -1;
  if (r.d_blob && r.d_blob_bytes != r.blob.size()) { HIP_OK(hipFree(r.d_blob)); r.d_blob = nullptr; }
  if (!r.d_blob) {
    HIP_OK(hipMalloc(&r.d_blob, r.blob.size()));
    r.d_blob_bytes = r.blob.size();
  }
  HIP_OK(hipDeviceSynchronize());  // a reload (fault campaigns reload per run): nothing may still read the old rows
  HIP_OK(hipMemcpy(r.d_blob, r.blob.data(), r.blob.size(), hipMemcpyHostToDevice));
  r.two_rows = count_two_rows(r.spec, r.blob);
  const PackedHeader *h = reinterpret_cast<const PackedHeader *>(r.blob.data());
  for (int l = 0; l < r.spec.nlayers; l++)
    r.rows[l] = reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(r.d_blob) + h->layer[l].offset);
  const char *l0 = std::getenv("BNN_MI355X_L0");
  const bool valu = l0 && std::strcmp(l0, "valu") == 0;
  r.l0_mfma = (h->l0_mfma_offset && !valu) ? static_cast<const uint8_t *>(r.d_blob) + h->l0_mfma_offset : nullptr;
  // side experiment, never the default: layer 1 of cnvW1A1 on the matrix pipe (kernels.hip, k_l1_mfma)
  const char *l1 = std::getenv("BNN_MI355X_L1");
  r.l1_mfma = r.spec.id == NET_CNVW1A1 && l1 && std::strcmp(l1, "mfma") == 0;
  r.l1_literal = r.spec.id == NET_CNVW1A1 && l1 && std::strcmp(l1, "lds") == 0;
  if (r.l1_mfma) {
    std::vector<uint8_t> tab(kL1MfmaBytes);
    l1_mfma_table(reinterpret_cast<const uint32_t *>(r.blob.data() + h->layer[1].offset), tab.data());
    if (!r.d_l1_mfma) HIP_OK(hipMalloc(reinterpret_cast<void **>(&r.d_l1_mfma), kL1MfmaBytes));
    HIP_OK(hipMemcpy(r.d_l1_mfma, tab.data(), kL1MfmaBytes, hipMemcpyHostToDevice));
  }
  return 0;
}

void free_workspace() {
  Runtime &r = rt();
  if (r.cap == 0 && r.cap2 == 0 && r.stage_cap == 0 && r.res_cap == 0 && !r.d_pp_src && !r.d_pp_rec && !r.file_cap && !r.all_cap && !r.h_io &&
      !r.h_classes && !r.h_words)
    return;
  if (r.device >= 0) (void)hipSetDevice(r.device);
  (void)hipDeviceSynchronize();
  if (r.h_io) (void)hipHostFree(r.h_io);
  r.h_io = r.d_io = nullptr;
  if (r.h_classes) (void)hipHostFree(r.h_classes);
  if (r.h_words) (void)hipHostFree(r.h_words);
  r.h_classes = nullptr; r.h_words = nullptr;
  r.h_classes_cap = r.h_words_cap = 0;
  r.h_scratch = std::vector<uint8_t>();
  (void)hipFree(r.buf0); (void)hipFree(r.buf1);
  for (auto &b : r.d_images) { (void)hipFree(b); b = nullptr; }
  (void)hipFree(r.d_scores); (void)hipFree(r.d_classes); (void)hipFree(r.d_words);
  r.buf0 = r.buf1 = nullptr;
  r.d_scores = nullptr; r.d_classes = nullptr; r.d_words = nullptr;
  r.cap = r.stage_cap = r.stage_slots = 0;
  r.res_cap = 0;
  (void)hipFree(r.buf0b); (void)hipFree(r.buf1b);
  r.buf0b = r.buf1b = nullptr;
  r.cap2 = 0;
  (void)hipFree(r.d_all);
  r.d_all = nullptr;
  r.all_cap = 0;
  (void)hipFree(r.d_file[0]); (void)hipFree(r.d_file[1]);
  r.d_file[0] = r.d_file[1] = nullptr;
  r.h_file[0].reset(); r.h_file[1].reset();
  r.file_cap = 0;
  (void)hipFree(r.d_pp_src); (void)hipFree(r.d_pp_tmp); (void)hipFree(r.d_pp_rec); (void)hipFree(r.d_pp_coef);
  r.d_pp_src = r.d_pp_tmp = r.d_pp_rec = nullptr;
  r.d_pp_coef = nullptr;
  r.pp_src_cap = r.pp_tmp_cap = r.pp_coef_cap = r.pp_rec_cap = 0;
}

// grow-only device buffer (contents are not preserved)
template <typename T>
int grow(T *&ptr, size_t &cap, size_t need) {
  if (need <= cap) return 0;
  HIP_OK(hipStreamSynchronize(rt().stream));
  (void)hipFree(ptr);
  ptr = nullptr;
  cap = 0;
  const size_t n = need + need / 4 + 256;
  HIP_OK(hipMalloc(reinterpret_cast<void **>(&ptr), n * sizeof(T)));
  cap = n;
  return 0;
}

// grow-only pinned host buffer (contents are not preserved)
template <typename T>
int grow_pinned(T *&ptr, size_t &cap, size_t need) {
  if (need <= cap) return 0;
  if (ptr) {
    HIP_OK(hipStreamSynchronize(rt().stream));
    (void)hipHostFree(ptr);
  }
  ptr = nullptr;
  cap = 0;
  const size_t n = need + need / 4 + 256;
  HIP_OK(hipHostMalloc(reinterpret_cast<void **>(&ptr), n * sizeof(T), hipHostMallocDefault));
  cap = n;
  return 0;
}

// activation workspace for `n` images per pass (at most kMaxChunk)
int reserve(int n) {
  Runtime &r = rt();
  if (n > kMaxChunk) n = kMaxChunk;
  if (n <= r.cap) return 0;
  if (bind_device()) return -1;
  HIP_OK(hipDeviceSynchronize());
  (void)hipFree(r.buf0); (void)hipFree(r.buf1);
  r.buf0 = r.buf1 = nullptr;
  r.cap = 0;
  size_t b0, b1;
  if (r.spec.is_cnv) cnv_workspace_bytes(r.spec.abits, &b0, &b1);
  else lfc_workspace_bytes(r.spec.abits, &b0, &b1);
  HIP_OK(hipMalloc(&r.buf0, (size_t)n * b0 + 256));
  HIP_OK(hipMalloc(&r.buf1, (size_t)n * b1 + 256));
  r.cap = n;
  return 0;
}

// Two compute lanes.  A host-path call cuts its batch into chunks so that copies and kernels overlap (plan_chunks); on ONE
// stream the nine launches of every chunk follow each other with the dispatcher's gap between them and each with its
// own tail, and the small chunks at the head of the plan pay that at 2 048-8 192 images a time: the plan of 131 072
// CIFAR images takes 11.12 ms of kernels against 10.40 ms for the batch in one pass.  With the chunks alternating over
// two streams -- a second activation workspace, the staging buffer of slot s feeds lane s -- one lane's gaps and tails
// are filled by the other's kernels: 10.61 ms (tools/two_lane_probe.py, profiles/r03_two_lane_probe.txt).
// Each lane needs its own pair of staging buffers (slots_for: four in all): with two, both are held by the two chunks
// in flight, the next copy cannot start before one of them ends, and a call from a host buffer got SLOWER (1 048 576
// images 87.7 -> 100.7 ms); with four: 131 072 images from a host buffer 11.55 -> 11.3 ms, from a file 12.6-13.3 ->
// 11.7-12.0 ms, 524 288 from a file 46.1-48.3 -> 44.6-44.9 ms (profiles/r03_two_lanes_ab.txt).
// BNN_MI355X_LANES=1 forces one lane (A/B); stage profiling and the stage-output hook always run on one lane.
int lanes_for(int nchunks) {
  static const bool one = [] { const char *e = std::getenv("BNN_MI355X_LANES"); return e && std::atoi(e) == 1; }();
  const Runtime &r = rt();
  return (nchunks >= 3 && !one && !r.profiling && r.debug_last_stage < 0) ? 2 : 1;
}
// the second lane's activation workspace for `n` images per pass
int reserve2(int n) {
  Runtime &r = rt();
  if (n > kMaxChunk) n = kMaxChunk;
  if (n <= r.cap2) return 0;
  if (bind_device()) return -1;
  HIP_OK(hipDeviceSynchronize());
  (void)hipFree(r.buf0b); (void)hipFree(r.buf1b);
  r.buf0b = r.buf1b = nullptr;
  r.cap2 = 0;
  size_t b0, b1;
  if (r.spec.is_cnv) cnv_workspace_bytes(r.spec.abits, &b0, &b1);
  else lfc_workspace_bytes(r.spec.abits, &b0, &b1);
  HIP_OK(hipMalloc(&r.buf0b, (size_t)n * b0 + 256));
  HIP_OK(hipMalloc(&r.buf1b, (size_t)n * b1 + 256));
  r.cap2 = n;
  return 0;
}
// HBM staging buffers of a call on `lanes` lanes: chunk c lands in slot c % slots and runs on lane c & 1 -- with two
// buffers both would be held by the two chunks in flight and the next copy could not start before one of them ends
int slots_for(int lanes) { return lanes == 2 ? kStageSlots : 2; }
// chunk c of a call that runs on `lanes` lanes: its stream
hipStream_t lane_stream(int c, int lanes) { return (lanes == 2 && (c & 1)) ? rt().stream2 : rt().stream; }
// all chunks are enqueued: whatever follows on r.stream (the results' way back) comes behind the second lane too
int join_lanes(int lanes) {
  Runtime &r = rt();
  if (lanes == 2) {
    HIP_OK(hipEventRecord(r.lane2_done, r.stream2));
    HIP_OK(hipStreamWaitEvent(r.stream, r.lane2_done, 0));
  }
  return 0;
}
// device time of a call's chunks, ms: the union of their [t0, t1] intervals (on one lane they do not overlap: the sum)
int chunks_device_ms(int nchunks, double *out) {
  Runtime &r = rt();
  std::vector<std::pair<float, float>> iv((size_t)nchunks);
  for (int c = 0; c < nchunks; c++) {
    float a = 0.f, d = 0.f;
    if (c) HIP_OK(hipEventElapsedTime(&a, r.time_events[0], r.time_events[2 * c]));
    HIP_OK(hipEventElapsedTime(&d, r.time_events[2 * c], r.time_events[2 * c + 1]));
    iv[(size_t)c] = {a, a + d};
  }
  if (trace().on) {  // when each chunk's stages ran on the device, microseconds from the first chunk's start
    std::string line = "bnn-mi355x trace device intervals of the chunks (start-end us):";
    char buf[48];
    for (auto &x : iv) {
      std::snprintf(buf, sizeof buf, " %.0f-%.0f", x.first * 1e3, x.second * 1e3);
      line += buf;
    }
    std::fprintf(stderr, "%s\n", line.c_str());
  }
  std::sort(iv.begin(), iv.end());
  double total = 0.0;
  float lo = iv[0].first, hi = iv[0].second;
  for (int c = 1; c < nchunks; c++) {
    if (iv[(size_t)c].first > hi) { total += hi - lo; lo = iv[(size_t)c].first; hi = iv[(size_t)c].second; }
    else if (iv[(size_t)c].second > hi) hi = iv[(size_t)c].second;
  }
  *out = total + (hi - lo);
  return 0;
}

// host-buffer path: staging for `chunk` images x `slots` and result buffers for `n_total` images
int reserve_host(int chunk, size_t n_total, int slots = 2) {
  Runtime &r = rt();
  if (bind_device()) return -1;
  if (chunk > r.stage_cap || slots > r.stage_slots) {
    HIP_OK(hipDeviceSynchronize());
    if (chunk < r.stage_cap) chunk = r.stage_cap;
    if (slots < r.stage_slots) slots = r.stage_slots;
    for (auto &b : r.d_images) {
      (void)hipFree(b);
      b = nullptr;
    }
    r.stage_cap = r.stage_slots = 0;
    for (int i = 0; i < slots; i++)
      HIP_OK(hipMalloc(reinterpret_cast<void **>(&r.d_images[i]), (size_t)chunk * r.spec.image_bytes() + 256));
    r.stage_cap = chunk;
    r.stage_slots = slots;
  }
  if (n_total > r.res_cap) {
    HIP_OK(hipDeviceSynchronize());
    (void)hipFree(r.d_scores); (void)hipFree(r.d_classes); (void)hipFree(r.d_words);
    r.d_scores = nullptr; r.d_classes = nullptr; r.d_words = nullptr;
    r.res_cap = 0;
    const size_t N = n_total < 1024 ? 1024 : n_total;
    if (r.spec.is_cnv) HIP_OK(hipMalloc(reinterpret_cast<void **>(&r.d_scores), N * 64 * sizeof(int16_t)));
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&r.d_classes), N * sizeof(int32_t)));
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&r.d_words), N * sizeof(uint64_t)));
    r.res_cap = N;
  }
  return 0;
}

// the pinned, device-mapped I/O block (kIo*); once per load (warm_up) or at first use
int ensure_io() {
  Runtime &r = rt();
  if (r.h_io) return 0;
  if (bind_device()) return -1;
  void *h = nullptr, *d = nullptr;
  HIP_OK(hipHostMalloc(&h, kIoBytes, hipHostMallocMapped));
  if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess || !d) {
    (void)hipHostFree(h);
    return fail("pinned I/O block: no device address for host memory");
  }
  std::memset(h, 0, kIoBytes);
  r.h_io = static_cast<uint8_t *>(h);
  r.d_io = static_cast<uint8_t *>(d);
  if (!r.io_t0) {
    HIP_OK(hipEventCreate(&r.io_t0));
    HIP_OK(hipEventCreate(&r.io_t1));
  }
  return 0;
}

// mapped result slots of a call of n images (null: that result goes through HBM and a copy)
struct ResultSlots {
  int32_t *h_classes = nullptr, *d_classes = nullptr;
  uint64_t *h_words = nullptr, *d_words = nullptr;
  int16_t *h_scores = nullptr, *d_scores = nullptr;
};
ResultSlots mapped_results(int n, bool scores, bool direct = false) {
  Runtime &r = rt();
  ResultSlots m;
  static const bool off = std::getenv("BNN_MI355X_NO_MAPPED_RESULTS") != nullptr;  // A/B of the chunked calls (a direct call has no other way)
  if ((off && !direct) || n > kMappedResMax || (scores && n > kMappedScoresMax) || ensure_io()) return m;
  m.h_classes = reinterpret_cast<int32_t *>(r.h_io + kIoClassesOff);
  m.d_classes = reinterpret_cast<int32_t *>(r.d_io + kIoClassesOff);
  m.h_words = reinterpret_cast<uint64_t *>(r.h_io + kIoWordsOff);
  m.d_words = reinterpret_cast<uint64_t *>(r.d_io + kIoWordsOff);
  if (scores) {
    m.h_scores = reinterpret_cast<int16_t *>(r.h_io + kIoScoresOff);
    m.d_scores = reinterpret_cast<int16_t *>(r.d_io + kIoScoresOff);
  }
  return m;
}

// An earlier device-pointer call on another stream may still own the activation workspaces: work about to be queued on
// `s` waits for its end first.
int settle_handover(hipStream_t s) {
  Runtime &r = rt();
  const int own = s == r.stream ? 0 : (s == r.stream2 ? 1 : -1);
  if (own >= 0 ? (r.ws_seen[own] == r.ws_gen || r.ws_last == s) : (!r.ws_pending || r.ws_last == s)) return 0;
  {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
    // A capturing stream must not wait on an event recorded outside its capture (the capture would be invalidated, or
    // the dependency silently dropped on replay): settle the hand-over on the host before anything is recorded.
    if (cap != hipStreamCaptureStatusNone) {
      // (a host-side wait is "unsafe" under the global capture mode frameworks use: switch this thread to the relaxed
      // mode around it -- the wait is on work submitted before the capture began, which is exactly the safe case)
      hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
      HIP_OK(hipThreadExchangeStreamCaptureMode(&mode));
      const hipError_t we = hipEventSynchronize(r.ws_event);
      (void)hipThreadExchangeStreamCaptureMode(&mode);
      if (we != hipSuccess)
        return fail("inference_device: an earlier call on another stream still owns the workspace and cannot be waited for during "
                    "stream capture; synchronise that stream before capturing");
    } else {
      HIP_OK(hipStreamWaitEvent(s, r.ws_event, 0));
    }
  }
  if (own >= 0) r.ws_seen[own] = r.ws_gen;
  else r.ws_pending = false;
  return 0;
}

// enqueue one chunk (n <= cap) whose images are already in HBM
// t0 / t1 (optional): this chunk's device time is t0 -> t1 (kernels.h)
int enqueue(const uint8_t *d_imgs, int n, int ncls, int32_t *d_classes, int16_t *d_scores, uint64_t *d_words,
            hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr, int lane = 0, bool t_dispatch = false, bool packed = false,
            unsigned *done_flag = nullptr, unsigned done_seq = 0) {
  Runtime &r = rt();
  hipError_t e;
  hipEvent_t *evs = nullptr;
  void *const ws0 = lane ? r.buf0b : r.buf0, *const ws1 = lane ? r.buf1b : r.buf1;
  // (the stages index the workspace by image: a chunk above its capacity would write past it on the device)
  if (n > (lane ? r.cap2 : r.cap)) return fail("internal: a chunk of " + std::to_string(n) + " images exceeds the activation workspace");
  if (settle_handover(s)) return -1;
  if (r.profiling) {
    const int need = (r.spec.is_cnv ? kCnvStages : kLfcStages) + 1;
    if (r.prof_used == r.prof_sets.size()) {
      std::vector<hipEvent_t> set(need);
      for (auto &ev : set) HIP_OK(hipEventCreate(&ev));
      r.prof_sets.push_back(set);
    }
    evs = r.prof_sets[r.prof_used++].data();
  }
  if (r.spec.is_cnv) {
    CnvLaunch a{};
    a.images = d_imgs; a.n = n; a.buf0 = ws0; a.buf1 = ws1;
    for (int l = 0; l < 9; l++) a.rows[l] = r.rows[l];
    a.l0_mfma = r.l0_mfma;
    a.l1_mfma = r.l1_mfma ? r.d_l1_mfma : nullptr;
    a.l1_literal = r.l1_literal;
    a.has_two = r.two_rows > 0;
    a.scores = d_scores; a.classes = d_classes; a.number_class = ncls; a.stream = s; a.events = evs;
    a.last_stage = r.debug_last_stage >= 0 ? r.debug_last_stage : kCnvStages - 1;
    a.t0 = t0; a.t1 = t1; a.done_flag = done_flag; a.done_seq = done_seq;
    e = run_cnv(r.spec.id, a);
  } else {
    LfcLaunch a{};
    a.images = d_imgs; a.packed = packed; a.n = n; a.buf0 = ws0; a.buf1 = ws1;
    for (int l = 0; l < 4; l++) a.rows[l] = r.rows[l];
    a.words = d_words;
    a.classes = d_classes; a.number_class = ncls; a.stream = s; a.events = evs;
    a.last_stage = r.debug_last_stage >= 0 ? r.debug_last_stage : kLfcStages - 1;
    a.t0 = t0; a.t1 = t1; a.t_dispatch = t_dispatch; a.done_flag = done_flag; a.done_seq = done_seq;
    e = run_lfc(r.spec.id, a);
  }
  if (e != hipSuccess) return fail(std::string("kernel launch: ") + hipGetErrorString(e));
  return 0;
}

bool ready() {
  if (!rt().d_blob) { fail("load_parameters has not been called (or failed)"); return false; }
  return true;
}

// ---- host data -> HBM through a ring of pinned pieces --------------------------------------------------
// hipMemcpyAsync from PAGEABLE memory is a single-threaded copy into the runtime's own pinned staging plus the DMA
// (measured: 34 GB/s, 11.8 ms for 131 072 CIFAR images -- more than the 10.7 ms their stages take), and a file read
// with pread() into a pageable chunk first pays a second CPU copy (27 GB/s with 8 reader threads).  For calls of
// kFeederMinBytes and more, worker threads fill 4 MB pinned pieces -- memcpy from the caller's buffer, or pread()
// straight from the page cache -- and the calling thread, the only one that talks to HIP, sends each piece with an
// asynchronous DMA as soon as it is full: one CPU copy per byte, spread over the cores this process may use.
constexpr size_t kFeederMinBytes = 2u << 20;  // (24 MB until round 4: with the small first pieces and the caller's own first piece a 5 000-record file takes 0.80 ms through the ring, 1.03 ms through a pageable chunk)
int usable_cpus() {
  int n = 0;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
  if (n <= 0) n = (int)std::thread::hardware_concurrency();
  if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // a container's CPU quota
    long long quota = 0, period = 0;
    char q[32] = {0};
    if (std::fscanf(f, "%31s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) {
      quota = std::atoll(q);
      const int lim = (int)(quota / period);
      if (lim >= 1 && lim < n) n = lim;
    }
    std::fclose(f);
  }
  return n > 0 ? n : 1;
}

#if defined(__x86_64__)
// write the lines of [p, p + n) back to memory and drop them from the caches (clflushopt: unordered, one fence at the end)
__attribute__((target("clflushopt"))) void flush_lines(const uint8_t *p, size_t n) {
  for (size_t i = 0; i < n; i += 64) _mm_clflushopt(const_cast<uint8_t *>(p) + i);
  _mm_sfence();
}
#endif

struct Feeder {
  static constexpr int kSlots = 12;
  static constexpr int kMaxWorkers = 14;
  size_t kSlotBytes = 4u << 20;  // BNN_MI355X_FEEDER_PIECE_MB overrides (tuning): alone, 4 MB pieces move at 49 GB/s, 8 MB at 52, 16 MB at 54
                                 // (profiles/r03_h2d_probe.txt); behind the readers 2-4 MB pieces gave the shortest calls
  // A PIECE is what one worker fills in one go (256-512 KB of source: a pread of that size takes 20-40 us, so the pieces of
  // a group are read side by side); a GROUP is what goes to HBM with one DMA: the pieces that share ring slot
  // (group % kSlots), up to kSlotBytes -- DMAs of 1-4 MB move at the link's rate, smaller ones cost ~10 us each whatever
  // their size (round 4's first form had piece = DMA: 4 MB pieces took one reader 0.3 ms each and the last bytes of a
  // 30 MB file arrived at 1.16 ms, 256 KB pieces moved at a third of the link's rate;
  // profiles/r04_timeline_cnvW1A1_10000_file_*.txt).  RAW job (CNV records, images as they are): source bytes are copied.
  // PACKED job (LFC: binarizeAndPack on the host, csrc/pack_inputs.h): `bytes` = whole 784-byte images, their 13-word
  // forms are what lands in the ring, and a chunk is one group (104 bytes per image: 32 768 images are 3.4 MB).
  struct Piece {
    int chunk, group;
    size_t ring_off;   // where in the group's ring slot this piece's output goes
    size_t src_off;    // of its first item (record / image) in the source
    size_t items;
    bool last_of_group, last_of_chunk;
    size_t group_dst_off, group_bytes;  // (on the last piece of a group) the group's place in the chunk's HBM buffer and its size
  };
  // the items of the job in flight: `rec` source bytes each, of which the first `skip` are dropped (the label byte of a
  // CIFAR-10 record: the readers scatter the records with preadv() so that only the image bodies land in the ring -- no
  // label bytes cross the link and no kernel has to remove them)
  size_t rec = 0, skip = 0;
  uint8_t *ring = nullptr;  // kSlots x kSlotBytes, pinned
  bool ready = false;       // init() went through: ring, events, streams and workers exist
  hipEvent_t sent[kSlots] = {};
  // Two things make the pipe faster than "pread, then one DMA queue" (tools/file_pipe_probe.cpp, profiles/r03_file_pipe_probe.txt:
  // 33-37 GB/s): the pieces alternate over TWO streams, i.e. two DMA queues (46 GB/s), and the readers write their lines
  // back behind the pread (clflushopt: a DMA that finds the lines dirty in a core's cache runs at two thirds of its rate;
  // 43 GB/s on one queue).  BNN_MI355X_FEEDER_STREAMS / BNN_MI355X_FEEDER_FLUSH override.
  hipStream_t aux = nullptr;
  hipEvent_t aux_done = nullptr;
  bool flush = false;
  std::vector<std::thread> workers;
  int raw_workers = 1;  // a RAW job keeps the first raw_workers threads busy (more readers slow the DMA down), a PACKED job all of them
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  bool quit = false;
  uint64_t job_id = 0;
  // the job in flight
  const std::vector<Piece> *pieces = nullptr;
  const uint8_t *mem = nullptr;  // source: host memory ...
  int fd = -1;                   // ... or a file
  bool packed = false;
  int job_workers = 0;
  std::atomic<size_t> next{0}, released{0};  // released: groups whose DMA is done (their ring slot may be refilled)
  std::unique_ptr<std::atomic<uint8_t>[]> filled;  // per piece: 0 not yet, 1 filled, 2 failed
  size_t filled_cap = 0;
  std::atomic<bool> abort{false};
  int active = 0;  // workers that have not yet left the current job (under mu)
  // BNN_MI355X_TRACE: per worker, microseconds from begin() to its first claim, microseconds spent filling, pieces filled
  struct Stat { double first = -1, busy = 0; int pieces = 0; };
  Stat stats[kMaxWorkers + 1];
  std::chrono::steady_clock::time_point job_t0;

  uint8_t *ring_at(const Piece &pc) const { return ring + (size_t)(pc.group % kSlots) * kSlotBytes + pc.ring_off; }
  bool read_all(uint8_t *dst, size_t bytes, size_t off) const {
    size_t done = 0;
    while (done < bytes) {
      const ssize_t got = ::pread(fd, dst + done, bytes - done, (off_t)(off + done));
      if (got <= 0) return false;
      done += (size_t)got;
    }
    return true;
  }
  bool fill(const Piece &pc, uint8_t *dst) const {
    if (packed) {
      if (mem) { binarize_pack(mem + pc.src_off, pc.items, reinterpret_cast<uint64_t *>(dst)); return true; }
      // from a file: the pixels pass through a cache-sized scratch block on their way to the binariser
      constexpr size_t kBlock = 80;  // images: 62 720 bytes
      uint8_t scratch[kBlock * kLfcPixels];
      for (size_t i = 0; i < pc.items; i += kBlock) {
        const size_t m = pc.items - i < kBlock ? pc.items - i : kBlock;
        if (!read_all(scratch, m * kLfcPixels, pc.src_off + i * kLfcPixels)) return false;
        binarize_pack(scratch, m, reinterpret_cast<uint64_t *>(dst) + i * kLfcWords);
      }
      return true;
    }
    const size_t body = rec - skip;
    if (skip == 0) {
      if (mem) { std::memcpy(dst, mem + pc.src_off, pc.items * rec); return true; }
      if (!read_all(dst, pc.items * rec, pc.src_off)) return false;
    } else if (mem) {
      for (size_t i = 0; i < pc.items; i++) std::memcpy(dst + i * body, mem + pc.src_off + i * rec + skip, body);
      return true;
    } else {
      // label bytes into a bin, bodies side by side: two iovecs per record, IOV_MAX (1024) per call
      constexpr size_t kBatch = 512;
      static_assert(2 * kBatch <= 1024, "IOV_MAX");
      struct iovec iov[2 * kBatch];
      uint8_t bin[64];
      if (skip > sizeof bin) return false;
      for (size_t i = 0; i < pc.items; i += kBatch) {
        const size_t m = pc.items - i < kBatch ? pc.items - i : kBatch;
        for (size_t k = 0; k < m; k++) {
          iov[2 * k] = {bin, skip};
          iov[2 * k + 1] = {dst + (i + k) * body, body};
        }
        size_t want = m * rec, done = 0;
        off_t off = (off_t)(pc.src_off + i * rec);
        int first_iov = 0;
        while (done < want) {
          const ssize_t got = ::preadv(fd, iov + first_iov, (int)(2 * m) - first_iov, off + (off_t)done);
          if (got <= 0) return false;
          done += (size_t)got;
          if (done < want) {  // a short read: step over the iovecs that are full, trim the one that is not
            size_t g = (size_t)got;
            while (g >= iov[first_iov].iov_len) g -= iov[first_iov++].iov_len;
            iov[first_iov].iov_base = static_cast<uint8_t *>(iov[first_iov].iov_base) + g;
            iov[first_iov].iov_len -= g;
          }
        }
      }
    }
#if defined(__x86_64__)
    if (flush) flush_lines(dst, pc.items * body);
#endif
    return true;
  }
  // claim the next piece and fill it; false when none is left.  (Also called once by the thread that started the job:
  // the workers need some tens of microseconds to wake up, and the first piece is what everything waits for.)
  bool work_one(int who = kMaxWorkers) {
    const size_t np = pieces->size();
    const size_t p = next.fetch_add(1, std::memory_order_relaxed);
    if (p >= np) return false;
    const Piece &pc = (*pieces)[p];
    // (its ring slot is free once the DMA of the group that had it before is done: `released` counts finished DMAs)
    while (!abort.load(std::memory_order_relaxed) && (size_t)pc.group >= released.load(std::memory_order_acquire) + kSlots) std::this_thread::yield();
    const bool tr = trace().on;
    std::chrono::steady_clock::time_point a;
    if (tr) {
      a = std::chrono::steady_clock::now();
      if (stats[who].first < 0) stats[who].first = std::chrono::duration<double, std::micro>(a - job_t0).count();
    }
    const bool ok = !abort.load(std::memory_order_relaxed) && fill(pc, ring_at(pc));
    if (tr) {
      stats[who].busy += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
      stats[who].pieces++;
    }
    filled[p].store(ok ? 1 : 2, std::memory_order_release);
    return true;
  }
  void worker(int index) {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_job.wait(lk, [&] { return quit || job_id != seen; });
        if (quit) return;
        seen = job_id;
        if (index >= job_workers) continue;  // not needed for this job (and not counted in `active`)
      }
      while (work_one(index)) {}
      std::lock_guard<std::mutex> lk(mu);
      if (--active == 0) cv_done.notify_all();
    }
  }
  // pinned ring, events and worker threads: once per process (the threads sleep between jobs)
  int init() {
    if (ready) return 0;
    if (ring) return -1;  // an earlier attempt got the ring but not its events or streams: stays refused (the caller reports it)
    if (const char *e = std::getenv("BNN_MI355X_FEEDER_PIECE_MB")) {
      const int mb = std::atoi(e);
      if (mb >= 1 && mb <= 32) kSlotBytes = (size_t)mb << 20;
    }
    if (hipHostMalloc(reinterpret_cast<void **>(&ring), kSlots * kSlotBytes, hipHostMallocDefault) != hipSuccess) { ring = nullptr; return -1; }
    for (auto &e : sent)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return -1;
    const char *ef = std::getenv("BNN_MI355X_FEEDER_FLUSH");
    aux = rt().feed_aux;  // (created in bind_device, next to the library's other streams: see there)
    if (aux && hipEventCreateWithFlags(&aux_done, hipEventDisableTiming) != hipSuccess) return -1;
#if defined(__x86_64__)
    flush = (ef ? std::atoi(ef) != 0 : false) && __builtin_cpu_supports("clflushopt");
#endif
    // Readers of a RAW job: 4 pread() threads already move 45 GB/s out of the page cache (8: 74 GB/s), more than the link
    // takes; beyond ~6 the DMA, which reads the lines they have just written, slows down more than they speed up
    // (profiles/r03_file_path_sweep.txt: 2 threads 20.4 ms per 131 072-record file, 4: 12.9-14.7, 6: 14.1, 14: 14.2).
    // A PACKED job is the other way round -- 7.5 source bytes per byte that crosses the link, the binarising cores are the
    // bound -- and takes every core this process may use but two (one for the calling thread, one for the driver's).
    const int avail = usable_cpus() - 2;
    int nr = avail > 6 ? 6 : avail, nt = avail > kMaxWorkers ? kMaxWorkers : avail;
    if (const char *e = std::getenv("BNN_MI355X_FEEDER_THREADS")) nr = std::atoi(e);
    if (const char *e = std::getenv("BNN_MI355X_PACK_THREADS")) nt = std::atoi(e);
    nr = nr < 1 ? 1 : (nr > kMaxWorkers ? kMaxWorkers : nr);
    nt = nt < nr ? nr : (nt > kMaxWorkers ? kMaxWorkers : nt);
    raw_workers = nr;
    for (int i = 0; i < nt; i++) workers.emplace_back([this, i] { worker(i); });
    ready = true;
    return 0;
  }
  void begin(const std::vector<Piece> &pcs, const uint8_t *m, int f, bool pack, size_t item_bytes, size_t drop) {
    if (pcs.size() > filled_cap) {
      filled.reset(new std::atomic<uint8_t>[pcs.size()]);
      filled_cap = pcs.size();
    }
    for (size_t i = 0; i < pcs.size(); i++) filled[i].store(0, std::memory_order_relaxed);
    next = 0; released = 0; abort = false;
    if (trace().on) {
      for (auto &st : stats) st = Stat{};
      job_t0 = std::chrono::steady_clock::now();
    }
    {
      std::lock_guard<std::mutex> lk(mu);
      pieces = &pcs; mem = m; fd = f; packed = pack; rec = item_bytes; skip = drop;
      const int want = pack ? (int)workers.size() : raw_workers;
      // (no more threads than pieces: a thread woken for nothing still has to be waited for at the end)
      job_workers = (size_t)want < pcs.size() ? want : (int)pcs.size();
      active = job_workers;
      job_id++;
      cv_job.notify_all();
    }
    (void)work_one();  // the first piece by this thread, while the workers wake up
  }
  // every worker has left the job: its description may go out of scope
  void end() {
    abort = true;  // (a no-op after a complete run: nothing is left to claim)
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return active == 0; });
    if (trace().on) {
      std::string line = "bnn-mi355x trace feeder (worker: first claim us / busy us / pieces; last = the calling thread):";
      char buf[64];
      for (int i = 0; i <= kMaxWorkers; i++) {
        if (stats[i].pieces == 0) continue;
        std::snprintf(buf, sizeof buf, " %d:%.0f/%.0f/%d", i, stats[i].first, stats[i].busy, stats[i].pieces);
        line += buf;
      }
      std::fprintf(stderr, "%s\n", line.c_str());
    }
  }
};
// One per process, never destroyed: its threads sleep on a condition variable between jobs and end with the process
// (a static object's destructor would have to join them at exit, behind the HIP runtime's own teardown).
Feeder &feeder() {
  static Feeder *f = new Feeder;
  return *f;
}
void drain_feeder() {
  Feeder &f = feeder();
  if (f.aux) (void)hipStreamSynchronize(f.aux);
}
// calls of this many source bytes and more go through the ring (BNN_MI355X_FEEDER_MIN_MB overrides; tuning)
bool use_feeder(size_t bytes) {
  static const bool off = std::getenv("BNN_MI355X_NO_FEEDER") != nullptr;
  static const size_t min_bytes = [] {
    const char *e = std::getenv("BNN_MI355X_FEEDER_MIN_MB");
    return e ? (size_t)std::atoi(e) << 20 : kFeederMinBytes;
  }();
  return !off && bytes >= min_bytes;
}

// n images from host memory (fd < 0) or from an open file, cut by `plan`, through the pinned ring into the HBM chunk
// buffers; consume(c, base, m, slot) enqueues chunk c's stages on its lane's stream (lane_stream) once its bytes
// (RAW: the image bodies, the readers having dropped the `skip` label bytes of every record; PACKED: 13 words per image)
// are in r.d_images[slot] and that stream has been made to wait for them.
template <typename Consume>
int feed_chunks(const uint8_t *mem, int fd, size_t first, size_t rec, size_t skip, const std::vector<int> &plan, int lanes, bool packed, Consume consume) {
  Runtime &r = rt();
  Feeder &F = feeder();
  if (F.init()) return fail("pinned staging ring: allocation failed");
  const int nchunks = (int)plan.size() - 1, nslots = slots_for(lanes);
  if (packed && (skip || rec != (size_t)kLfcPixels || (size_t)largest_chunk(plan) * kLfcWords * 8 > F.kSlotBytes))
    return fail("internal: packed feed of a chunk that does not fit a ring slot");
  // Groups (DMAs) start small and double -- the first one is what everything else waits for --, the pieces inside them are
  // small throughout; the first chunk is one group in the smallest pieces, so that all workers share it.
  std::vector<Feeder::Piece> pieces;
  const size_t out = packed ? (size_t)kLfcWords * 8 : rec - skip;          // ring / HBM bytes per item
  const size_t piece0 = packed ? 256 : (256u << 10) / rec, piece1 = packed ? 1024 : (512u << 10) / rec;  // items per piece
  const size_t group_max = F.kSlotBytes / out;                              // items per group at most
  size_t group_want = (1u << 20) / out;                                     // RAW: 1, 2, 4 MB ...; PACKED: a chunk is a group
  int ngroups = 0;
  for (int c = 0; c < nchunks; c++) {
    const size_t items = (size_t)(plan[c + 1] - plan[c]), src0 = first + (size_t)plan[c] * rec;
    for (size_t g0 = 0; g0 < items;) {
      size_t gi = items - g0;
      if (!packed) {
        const size_t cap = c == 0 ? group_max : group_want;
        if (gi > cap) gi = cap;
        if (items - g0 - gi < gi / 4 && items - g0 <= group_max) gi = items - g0;  // (no crumb of a group at the end of a chunk)
        if (c > 0) group_want = group_want * 2 < group_max ? group_want * 2 : group_max;
      }
      const size_t psz = c == 0 ? piece0 : piece1;
      for (size_t o = 0; o < gi;) {
        const size_t b = gi - o < psz ? gi - o : psz;
        Feeder::Piece pc{};
        pc.chunk = c; pc.group = ngroups;
        pc.ring_off = o * out;
        pc.src_off = src0 + (g0 + o) * rec;
        pc.items = b;
        pc.last_of_group = o + b == gi;
        pc.last_of_chunk = pc.last_of_group && g0 + gi == items;
        pc.group_dst_off = g0 * out;
        pc.group_bytes = gi * out;
        // (what the ring and the HBM buffers rely on, checked where it is decided: a piece ends inside its group's ring slot,
        // a group inside its chunk's buffer)
        if (pc.ring_off + b * out > F.kSlotBytes || pc.group_bytes > F.kSlotBytes || pc.group_dst_off + pc.group_bytes > items * out)
          return fail("internal: feed plan outside the ring slot / chunk buffer");
        pieces.push_back(pc);
        o += b;
      }
      g0 += gi;
      ngroups++;
    }
  }
  F.begin(pieces, mem, fd, packed, rec, skip);
  struct End {
    Feeder &f;
    ~End() { f.end(); }
  } end_guard{F};
  size_t issued = 0, released = 0;  // in groups
  auto release_done = [&]() {  // DMAs that have finished: their ring slots may be refilled
    while (released < issued && hipEventQuery(F.sent[released % Feeder::kSlots]) == hipSuccess) released++;
    F.released.store(released, std::memory_order_release);
  };
  bool chunk_open = false;  // a group of the current chunk has been sent already
  for (size_t p = 0; p < pieces.size(); p++) {
    const Feeder::Piece &pc = pieces[p];
    const int c = pc.chunk, slot = c % nslots;
    uint8_t st;
    while ((st = F.filled[p].load(std::memory_order_acquire)) == 0) {
      release_done();
      std::this_thread::yield();
    }
    if (st != 1) {
      F.abort = true;
      return fail("input file: read error");
    }
    if (!pc.last_of_group) continue;
    const int base = plan[c], m = plan[c + 1] - plan[c];
    uint8_t *chunk_dst = r.d_images[slot];
    // the chunk that had this HBM buffer before: its stages are done (both DMA queues write the buffer)
    if (!chunk_open && c >= nslots) {
      HIP_OK(hipStreamWaitEvent(r.copy_stream, r.consumed[slot], 0));
      if (F.aux) HIP_OK(hipStreamWaitEvent(F.aux, r.consumed[slot], 0));
    }
    chunk_open = !pc.last_of_chunk;
    hipStream_t ds = (F.aux && (pc.group & 1)) ? F.aux : r.copy_stream;
    HIP_OK(hipMemcpyAsync(chunk_dst + pc.group_dst_off, F.ring + (size_t)(pc.group % Feeder::kSlots) * F.kSlotBytes, pc.group_bytes, hipMemcpyHostToDevice, ds));
    HIP_OK(hipEventRecord(F.sent[pc.group % Feeder::kSlots], ds));
    issued = (size_t)pc.group + 1;
    release_done();
    if (!pc.last_of_chunk) continue;
    if (F.aux) {  // the chunk is complete when both queues have delivered their groups
      HIP_OK(hipEventRecord(F.aux_done, F.aux));
      HIP_OK(hipStreamWaitEvent(r.copy_stream, F.aux_done, 0));
    }
    HIP_OK(hipEventRecord(r.copied[slot], r.copy_stream));
    HIP_OK(hipStreamWaitEvent(lane_stream(c, lanes), r.copied[slot], 0));
    if (consume(c, base, m, slot)) return -1;
    // (only where a later chunk will reuse this buffer: a marker packet on the lane's stream is not free, BNN_MI355X_TRACE)
    if (c + nslots < nchunks) HIP_OK(hipEventRecord(r.consumed[slot], lane_stream(c, lanes)));
  }
  return 0;
}

// First use of a process: the first launch of a kernel pays for loading its code object and for the
// allocation of workspace and events, and all of that would land in the `usecPerImage` of the caller's first
// inference() (measured: 2.4 ms instead of ~10 us for one lfcW1A1 image; the reference's FPGA reports 7-8 us
// from the first call on).  So the load pays it: one image through the small-batch kernels and a few
// thousand through the throughput forms, zeros in, results dropped.  Once per process.
int warm_up() {
  Runtime &r = rt();
  if (r.warmed || std::getenv("BNN_MI355X_NO_WARMUP")) return 0;
  const int big = 8192;
  const size_t isz = (size_t)r.spec.image_bytes();
  if (reserve(big) || reserve_host(big, (size_t)big)) return -1;
  // (the pinned ring and the reader threads of the file path are NOT created here: 48 MB of pinned memory and six threads
  // per loaded network would be a poor default for users who never classify a large file; the first such call pays ~10 ms)
  HIP_OK(hipMemsetAsync(r.d_images[0], 0, (size_t)big * isz, r.stream));
  for (int n : {1, 2, 300, 600, 1100, 2500, 5000, big})  // one batch size inside every band of the dispatch policy
    if (enqueue(r.d_images[0], n, 10, r.d_classes, r.spec.is_cnv ? r.d_scores : nullptr, r.d_words, r.stream)) return -1;
  if (r.spec.is_cnv) {  // the file path's label-stripping kernel lives in another code object
    const hipError_t e = launch_strip_records(r.d_images[0], 3073, 1, r.d_images[1], 2, r.stream);
    if (e != hipSuccess) return fail(std::string("kernel launch: ") + hipGetErrorString(e));
  } else {  // the forms that start from host-binarised words (the zeros read as 13-word images just as well)
    for (int n : {1, 300, 600, 1100, 2500, big})
      if (enqueue(r.d_images[0], n, 10, r.d_classes, nullptr, r.d_words, r.stream, nullptr, nullptr, 0, false, true)) return -1;
  }
  // the pinned I/O block of the direct calls, and one image through it
  if (ensure_io()) return -1;
  {
    const ResultSlots m = mapped_results(1, r.spec.is_cnv, true);
    if (m.d_classes && enqueue(r.spec.is_cnv ? r.d_io + 16 : r.d_io, 1, 10, m.d_classes, m.d_scores, m.d_words, r.stream, r.io_t0, r.io_t1, 0, true, !r.spec.is_cnv))
      return -1;
  }
  while (r.time_events.size() < 2) {
    hipEvent_t e;
    HIP_OK(hipEventCreateWithFlags(&e, kTimeEventFlags));
    r.time_events.push_back(e);
  }
  HIP_OK(hipEventRecord(r.time_events[0], r.stream));
  HIP_OK(hipEventRecord(r.time_events[1], r.stream));
  HIP_OK(hipStreamSynchronize(r.stream));
  r.warmed = true;
  return 0;
}

// ---- input files, streamed ---------------------------------------------------
// CIFAR-10 binary: records of [label u8][R 1024][G 1024][B 1024] (tiny-cnn parse_cifar10, call site
// main_python.cpp:129,152); bodies go to the GPU as they are, planar CHW uint8.  MNIST idx3: 16-byte
// big-endian header (magic 0x803, count, rows, cols), then count x 784 uint8 (tiny-cnn
// parse_mnist_images, lfcW1A1/sw/main_python.cpp:122,144).
// The entry points do not parse the file on the host: the records go to HBM as they lie on
// disk (reader threads pread() chunk c+1 while chunk c is copied and classified) and the label
// bytes are dropped by a kernel (k_strip_records).  An MNIST body needs no kernel at all.
struct ImageFile {
  int fd = -1;
  size_t n = 0;            // images
  size_t rec = 0, skip = 0;  // bytes per record on disk, bytes to drop at the start of each
  size_t first = 0;        // file offset of record 0
  ~ImageFile() { if (fd >= 0) ::close(fd); }
};

int open_image_file(const char *path, ImageFile &f) {
  Runtime &r = rt();
  f.fd = ::open(path ? path : "", O_RDONLY);
  struct stat st;
  if (f.fd < 0 || ::fstat(f.fd, &st) != 0) return fail(std::string("Could not open file ") + (path ? path : ""));
  const size_t size = (size_t)st.st_size;
  if (r.spec.is_cnv) {  // whole records only, a trailing partial record is ignored (the reference reads record by record)
    f.rec = 3073; f.skip = 1; f.first = 0;
    f.n = size / 3073;
  } else {
    unsigned char h[16];
    if (::pread(f.fd, h, 16, 0) != 16) return fail("MNIST image file: short header");
    auto be = [&](int o) { return ((uint32_t)h[o] << 24) | ((uint32_t)h[o + 1] << 16) | ((uint32_t)h[o + 2] << 8) | h[o + 3]; };
    if (be(0) != 0x803u || be(8) != 28 || be(12) != 28) return fail("MNIST image file: bad header");
    f.rec = 784; f.skip = 0; f.first = 16;
    f.n = be(4);
    if (size < 16 + f.n * 784) return fail("MNIST image file: truncated");
  }
  if (f.n > 0x7FFFFFFFu) return fail("input file: too many images for the int-sized ABI");
  return 0;
}

// pread() [offset, offset + bytes) into dst on up to 8 threads (a single thread moves page-cache
// data at a fraction of what the PCIe link takes)
struct ChunkReader {
  std::vector<std::thread> threads;
  std::atomic<bool> ok{true};
  void start(int fd, size_t offset, size_t bytes, uint8_t *dst) {
    ok = true;
    const size_t hw = std::thread::hardware_concurrency();
    size_t nt = bytes / (4u << 20) + 1;
    if (nt > 8) nt = 8;
    if (hw && nt > hw) nt = hw;
    if (nt == 1) {  // small read: not worth a thread
      size_t done = 0;
      while (done < bytes) {
        const ssize_t got = ::pread(fd, dst + done, bytes - done, (off_t)(offset + done));
        if (got <= 0) { ok = false; return; }
        done += (size_t)got;
      }
      return;
    }
    const size_t part = ((bytes + nt - 1) / nt + 4095) & ~(size_t)4095;
    for (size_t t = 0; t < nt; t++) {
      const size_t lo = t * part, hi = (lo + part < bytes) ? lo + part : bytes;
      if (lo >= hi) break;
      threads.emplace_back([this, fd, offset, dst, lo, hi] {
        size_t done = lo;
        while (done < hi) {
          const ssize_t got = ::pread(fd, dst + done, hi - done, (off_t)(offset + done));
          if (got <= 0) { ok = false; return; }
          done += (size_t)got;
        }
      });
    }
  }
  bool wait() {
    for (auto &t : threads) t.join();
    threads.clear();
    return ok;
  }
  ~ChunkReader() { (void)wait(); }
};

// Streams images [0, n) of an open file into HBM, chunk by chunk (plan_chunks; reader
// threads one chunk ahead of the copy).  Chunk c lands packed (labels stripped) at dst(base, slot) and
// `consume(c, base, m, slot)` is called once its copy and strip are enqueued on copy_stream and
// the chunk's lane (lane_stream(c, lanes)) has been made to wait for them.  reuse_slots: the destinations alternate between two
// buffers, so a chunk's copy must wait until the stages of the chunk two before have consumed it.
template <typename Dst, typename Consume>
int stream_file(const ImageFile &f, int n, bool reuse_slots, int lanes, Dst dst, Consume consume) {
  Runtime &r = rt();
  const std::vector<int> plan = plan_chunks(n, false, true);
  const int chunk = largest_chunk(plan);
  const int nchunks = (int)plan.size() - 1;
  const size_t chunk_bytes = (size_t)chunk * f.rec;
  if (chunk_bytes > r.file_cap) {
    HIP_OK(hipDeviceSynchronize());
    for (int i = 0; i < 2; i++) {
      (void)hipFree(r.d_file[i]);
      r.d_file[i] = nullptr;
      r.h_file[i].reset();
    }
    r.file_cap = 0;
    for (int i = 0; i < 2; i++) {
      r.h_file[i].reset(new (std::nothrow) uint8_t[chunk_bytes + 8]);
      if (!r.h_file[i]) return fail("out of memory");
      if (f.skip) HIP_OK(hipMalloc(reinterpret_cast<void **>(&r.d_file[i]), chunk_bytes + 256));
      if (!r.file_sent[i]) HIP_OK(hipEventCreateWithFlags(&r.file_sent[i], hipEventDisableTiming));
    }
    r.file_cap = chunk_bytes;
  }
  ChunkReader reader;
  auto span = [&](int c, size_t *off, int *m) {
    *m = plan[c + 1] - plan[c];
    *off = f.first + (size_t)plan[c] * f.rec;
  };
  size_t off;
  int m;
  span(0, &off, &m);
  reader.start(f.fd, off, (size_t)m * f.rec, r.h_file[0].get());
  for (int c = 0; c < nchunks; c++) {
    const int base = plan[c], slot = c & 1;
    span(c, &off, &m);
    if (!reader.wait()) return fail("input file: read error");
    if (c + 1 < nchunks) {  // the other host chunk is free once its copy (chunk c-1) has left it
      if (c >= 1) HIP_OK(hipEventSynchronize(r.file_sent[slot ^ 1]));
      size_t off2;
      int m2;
      span(c + 1, &off2, &m2);
      reader.start(f.fd, off2, (size_t)m2 * f.rec, r.h_file[slot ^ 1].get());
    }
    // one chunk: nothing to overlap, stay on the compute stream (fewer driver round trips for small calls)
    hipStream_t cs = (nchunks == 1 && reuse_slots) ? r.stream : r.copy_stream;
    if (c >= 2) HIP_OK(hipStreamWaitEvent(cs, r.consumed[slot], 0));  // d_file[slot] / dst free again
    uint8_t *packed = dst(base, slot);
    uint8_t *to = f.skip ? r.d_file[slot] : packed;
    HIP_OK(hipMemcpyAsync(to, r.h_file[slot].get(), (size_t)m * f.rec, hipMemcpyHostToDevice, cs));
    if (nchunks > 1) HIP_OK(hipEventRecord(r.file_sent[slot], cs));
    if (f.skip) {
      const hipError_t e = launch_strip_records(r.d_file[slot], (int)f.rec, (int)f.skip, packed, m, cs);
      if (e != hipSuccess) return fail(std::string("kernel launch: ") + hipGetErrorString(e));
    }
    if (cs != r.stream) {
      HIP_OK(hipEventRecord(r.copied[slot], cs));
      HIP_OK(hipStreamWaitEvent(lane_stream(c, lanes), r.copied[slot], 0));
    }
    if (consume(c, base, m, slot)) return -1;
    // (with reuse_slots the consumer's stages read `packed`; without, only the strip kernel reads d_file[slot])
    if (c + 2 < nchunks) HIP_OK(hipEventRecord(r.consumed[slot], reuse_slots ? lane_stream(c, lanes) : r.copy_stream));
  }
  return 0;
}

// ---- host buffers: the copies on a thread of their own -------------------------------------------------------
// hipMemcpyAsync from PAGEABLE memory returns when the bytes have left (the runtime pins or stages the pages and waits for
// its DMA): with copies and launches on one thread the link idles while the nine launches of every chunk are enqueued --
// 20-45 us a chunk, 150-250 us of a 1.2 ms call of 10 000 CIFAR images (BNN_MI355X_TRACE: the copies alone ran at
// 28-52 GB/s, the call's bytes arrived at 37).  So the copies of chunks 1.. are issued by a helper thread, back to back
// on the copy stream, while the calling thread enqueues stages; chunk 0's copy stays with the caller (the helper needs
// 30-60 us to wake up).  Two counters tie them together: `copied` (helper -> caller: chunks whose `copied` event is
// recorded -- a stream may only be made to wait for an event that has been recorded) and `consumed` (caller -> helper:
// chunks whose stages are enqueued and whose `consumed` event is recorded -- their HBM slot may be overwritten).
// One per process, never destroyed (like the feeder's threads).
struct Copier {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv, cv_idle;
  bool started = false, busy = false;
  uint64_t job_id = 0;
  // the job
  const uint8_t *src = nullptr;
  const std::vector<int> *plan = nullptr;
  size_t isz = 0;
  int nslots = 2, device = 0;
  std::atomic<int> copied{0}, consumed{0};
  std::atomic<bool> abort{false}, failed{false};
  std::string err;

  void run() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return job_id != seen; });
        seen = job_id;
      }
      Runtime &r = rt();
      const int nchunks = (int)plan->size() - 1;
      hipError_t e = hipSetDevice(device);
      for (int c = 1; c < nchunks && e == hipSuccess; c++) {
        const int slot = c % nslots, base = (*plan)[c], m = (*plan)[c + 1] - base;
        // chunk c-1's copy is in the stream (the caller's for c = 1), and the chunk that had this slot is consumed
        while (!abort.load(std::memory_order_relaxed) &&
               (copied.load(std::memory_order_acquire) < c || (c >= nslots && consumed.load(std::memory_order_acquire) < c - nslots + 1)))
          std::this_thread::yield();
        if (abort.load(std::memory_order_relaxed)) break;
        if (c >= nslots) e = hipStreamWaitEvent(r.copy_stream, r.consumed[slot], 0);
        if (e == hipSuccess) e = hipMemcpyAsync(r.d_images[slot], src + (size_t)base * isz, (size_t)m * isz, hipMemcpyHostToDevice, r.copy_stream);
        if (e == hipSuccess) e = hipEventRecord(r.copied[slot], r.copy_stream);
        if (e == hipSuccess) copied.store(c + 1, std::memory_order_release);
      }
      if (e != hipSuccess) {
        err = std::string("host buffer copy: ") + hipGetErrorString(e);
        failed.store(true, std::memory_order_release);
      }
      std::lock_guard<std::mutex> lk(mu);
      busy = false;
      cv_idle.notify_all();
    }
  }
  void begin(const uint8_t *s, const std::vector<int> &p, size_t image_bytes, int slots, int dev) {
    std::lock_guard<std::mutex> lk(mu);
    if (!started) {
      th = std::thread([this] { run(); });
      th.detach();
      started = true;
    }
    src = s; plan = &p; isz = image_bytes; nslots = slots; device = dev;
    copied = 0; consumed = 0; abort = false; failed = false;
    busy = true;
    job_id++;
    cv.notify_one();
  }
  // the helper has left the job: its description may go out of scope
  void end() {
    abort = true;  // (a no-op after a complete run)
    std::unique_lock<std::mutex> lk(mu);
    cv_idle.wait(lk, [&] { return !busy; });
  }
};
Copier &copier() {
  static Copier *c = new Copier;
  return *c;
}

// ---- the entry points that take HOST data --------------------------------------------------------------------
// n images from host memory (mem) or from an open input file -> any of classes / scores / words (host arrays).
// usec: device time of the compute stages only, per image (the reference times the accelerator call alone,
// foldedmv-offload.h:389-392, and excludes the transfers: TRANSFER_EXCL, foldedmv-offload.cpp:53-61).
struct Source {
  const uint8_t *mem = nullptr;     // n x image_bytes, as in the bodies of the file formats, or
  const ImageFile *file = nullptr;  // an open CIFAR-10 binary / MNIST idx3 file
};

// Small calls -- one CIFAR image (classify_image and the webcam loops call inference(path) per frame,
// bnn/bnn.py:131-137, main_python.cpp:120-139), up to kDirectMaxLfc MNIST images -- with no transfer at all: the image
// (LFC: binarised here, 13 words each, like the reference's host does) is placed in pinned memory the GPU addresses,
// the one-launch kernels read it over the link and write their results into pinned memory: launch, one wait.
// words_view (optional, LFC): receives a pointer to the raw output words where the call left them in pinned memory,
// valid until the next call into the library -- the batched decode reads them there instead of from a copy
int infer_direct(const Source &src, int n, int ncls, int32_t *classes, int16_t *scores, uint64_t *words, float *usec, const uint64_t **words_view) {
  Runtime &r = rt();
  if (ensure_io() || reserve(n)) return -1;
  trace().mark("reserved");
  const uint8_t *d_in = r.d_io;
  if (r.spec.is_cnv) {
    // the kernels read an image with 128-bit loads: the body (3072 bytes behind the label byte) starts at +16
    if (src.file) {  // (n == 1: the records of a file are a label byte apart)
      if (::pread(src.file->fd, r.h_io + 15, 3073, (off_t)src.file->first) != 3073) return fail("input file: read error");
    } else {
      std::memcpy(r.h_io + 16, src.mem, (size_t)n * 3072);
    }
    d_in = r.d_io + 16;
  } else if (src.file) {
    const size_t bytes = (size_t)n * kLfcPixels;
    if (r.h_scratch.size() < bytes) r.h_scratch.resize(bytes);
    size_t done = 0;
    while (done < bytes) {
      const ssize_t got = ::pread(src.file->fd, r.h_scratch.data() + done, bytes - done, (off_t)(src.file->first + done));
      if (got <= 0) return fail("input file: read error");
      done += (size_t)got;
    }
    binarize_pack(r.h_scratch.data(), (size_t)n, reinterpret_cast<uint64_t *>(r.h_io));
  } else {
    binarize_pack(src.mem, (size_t)n, reinterpret_cast<uint64_t *>(r.h_io));
  }
  const bool want_scores = scores && r.spec.is_cnv;
  const ResultSlots m = mapped_results(n, want_scores, true);
  if (!m.d_classes) return fail("pinned I/O block unavailable");
  DrainOnFailure drain;
  trace().mark("input_placed");
  // BNN_MI355X_DIRECT_TIMING=host (opt-in, one image): the call is timed and waited for WITHOUT the runtime -- no event
  // packets around the launch, the last kernel stores a completion word into the pinned block behind its results, the host
  // spins on that word, and usecPerImage is the host's clock from the launch to the word, i.e. the reference's own
  // definition (wall clock around the accelerator call, foldedmv-offload.cpp:138-140, foldedmv-offload.h:342-345) instead of this runtime's (device
  // time by events).  tools/launch_latency_probe.hip, profiles/r04_launch_latency_probe.txt: the two timing events cost
  // 4.3 us in the launch call and ~2 us in a later kernel start, the event's readiness is seen ~2 us after the word:
  // 21.5 -> 13.9 us around an 8 us kernel.  The default keeps the events: usecPerImage keeps its meaning.
  static const bool host_timing = [] { const char *e = std::getenv("BNN_MI355X_DIRECT_TIMING"); return e && std::strcmp(e, "host") == 0; }();
  if (host_timing && n == 1) {
    volatile unsigned *done = reinterpret_cast<volatile unsigned *>(r.h_io + kIoDoneOff);
    const unsigned seq = ++r.io_seq;
    const auto t_launch = std::chrono::steady_clock::now();
    if (enqueue(d_in, n, ncls, classes ? m.d_classes : nullptr, want_scores ? m.d_scores : nullptr, m.d_words, r.stream, nullptr, nullptr, 0, false,
                !r.spec.is_cnv, reinterpret_cast<unsigned *>(r.d_io + kIoDoneOff), seq))
      return -1;
    trace().mark("launched");
    const auto until = t_launch + std::chrono::microseconds(500);
    while (*done != seq && std::chrono::steady_clock::now() < until) {}
    auto t_done = std::chrono::steady_clock::now();
    if (*done != seq) {
      // not within 500 us: something else holds the GPU -- or this pass has no one-block last kernel to set the mark (a
      // cnvW2A2 under fault injection runs its -2-aware staged layers): the blocking wait, the results are there after it
      HIP_OK(hipStreamSynchronize(r.stream));
      t_done = std::chrono::steady_clock::now();
    }
    trace().mark("synced");
    if (classes) std::memcpy(classes, m.h_classes, (size_t)n * 4);
    if (want_scores) std::memcpy(scores, m.h_scores, (size_t)n * 128);
    if (words && !r.spec.is_cnv) std::memcpy(words, m.h_words, (size_t)n * 8);
    if (words_view) *words_view = m.h_words;
    if (usec) *usec = std::chrono::duration<float, std::micro>(t_done - t_launch).count();
    drain.ok();
    trace().mark("done");
    trace().dump("direct, host-timed");
    return 0;
  }
  if (enqueue(d_in, n, ncls, classes ? m.d_classes : nullptr, want_scores ? m.d_scores : nullptr, m.d_words, r.stream, r.io_t0, r.io_t1, 0, true,
              !r.spec.is_cnv))
    return -1;
  trace().mark("launched");
  // The wait: a blocking wait sleeps until the completion interrupt has made its way to this thread -- 15-25 us behind a
  // 9 us kernel (BNN_MI355X_TRACE).  For calls this small the thread polls the stop event instead, for at most 300 us;
  // whatever is not done by then gets the blocking wait.
  {
    const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(300);
    hipError_t q;
    while ((q = hipEventQuery(r.io_t1)) == hipErrorNotReady && std::chrono::steady_clock::now() < until) {}
    if (q != hipSuccess) HIP_OK(hipStreamSynchronize(r.stream));
  }
  trace().mark("synced");
  if (classes) std::memcpy(classes, m.h_classes, (size_t)n * 4);
  if (want_scores) std::memcpy(scores, m.h_scores, (size_t)n * 128);
  if (words && !r.spec.is_cnv) std::memcpy(words, m.h_words, (size_t)n * 8);
  if (words_view) *words_view = m.h_words;
  float ms = 0.f;
  HIP_OK(hipEventElapsedTime(&ms, r.io_t0, r.io_t1));
  if (usec) *usec = ms * 1000.f / (float)n;
  drain.ok();
  trace().mark("done");
  trace().dump("direct");
  return 0;
}

int infer_any(const Source &src, int n, int ncls, int32_t *classes, int16_t *scores, uint64_t *words, float *usec, const uint64_t **words_view = nullptr) {
  Runtime &r = rt();
  if (!ready()) return -1;
  if (ncls < 1 || ncls > 64) return fail("number_class must be in 1..64");
  if (usec) *usec = 0.f;
  if (n <= 0) return 0;
  if (!src.file) trace().start();  // (a file call started the clock before it opened the file)
  const bool from_file = src.file != nullptr;
  const bool hook = r.debug_last_stage >= 0;  // (the stage-output test hook reads the workspace after the call: all images in it, one chunk, device binarisation)
  const bool want_scores = scores && r.spec.is_cnv;
  static const bool no_direct = std::getenv("BNN_MI355X_NO_DIRECT") != nullptr;  // A/B
  if (!hook && !r.profiling && !no_direct && n <= (r.spec.is_cnv ? (from_file ? 1 : kDirectMaxCnv) : kDirectMaxLfc) && (!want_scores || n <= kMappedScoresMax))
    return infer_direct(src, n, ncls, classes, scores, words, usec, words_view);
  const size_t isz = (size_t)r.spec.image_bytes();
  const size_t rec = from_file ? src.file->rec : isz, skip = from_file ? src.file->skip : 0, first = from_file ? src.file->first : 0;
  const std::vector<int> plan = plan_chunks(n, hook, from_file);
  const int chunk = largest_chunk(plan);
  const int nchunks = (int)plan.size() - 1;
  // LFC: binarizeAndPack on the host side of the copy, as the reference does (foldedmv-offload.cpp:82-98,186-194): worker
  // threads turn 784 pixels into 13 words straight into pinned memory and 104 bytes per image cross the link (the raw pixels
  // made every host-path call of the LFC nets PCIe-bound at 34-37 GB/s = 43 M images/s, against 230 M/s of stages).
  static const bool no_pack = std::getenv("BNN_MI355X_NO_HOST_PACK") != nullptr;  // A/B: raw pixels to HBM, binarised there
  const bool packed = !r.spec.is_cnv && !hook && !no_pack && feeder().init() == 0 && (size_t)chunk * kLfcWords * 8 <= feeder().kSlotBytes;
  // The pinned ring is for FILES (and for the LFC nets' binarised words).  A CIFAR buffer in host memory goes faster without
  // it: the runtime's own pageable path moves a 100 MB chunk at 54 GB/s (profiles/r03_h2d_probe.txt), the ring's 4-8 MB
  // pieces reach 49-52 and add a copy (measured, same box: 13.8 ms through the ring, 12.7 ms without, 131 072 CIFAR
  // images).  BNN_MI355X_FEED_HOST=1 forces it.
  static const bool feed_host = std::getenv("BNN_MI355X_FEED_HOST") != nullptr;
  // large file: worker threads pread() it into the pinned ring piece by piece (feed_chunks); small: one or a few chunks
  // through a pageable host chunk (stream_file) -- also where the ring cannot be had (no pinned memory to spare): slower, same result
  const bool ring = packed || ((from_file || feed_host) && nchunks > 1 && use_feeder((size_t)n * rec) && feeder().init() == 0);
  const int lanes = lanes_for(nchunks);
  const int nslots = slots_for(lanes);
  if (reserve(chunk) || (lanes == 2 && reserve2(chunk)) || reserve_host(chunk, (size_t)n, nslots)) return -1;
  while ((int)r.time_events.size() < 2 * nchunks) {
    hipEvent_t e;
    HIP_OK(hipEventCreateWithFlags(&e, kTimeEventFlags));
    r.time_events.push_back(e);
  }
  // results: small calls have the last stage write them into pinned memory (no copy back); larger ones keep them in HBM
  // and fetch them with ONE transfer at the end (a D2H into pageable memory per chunk would make the host wait for each
  // chunk's kernels before it can queue the next copy)
  const ResultSlots m = hook ? ResultSlots{} : mapped_results(n, want_scores);
  int32_t *const dc = m.d_classes ? m.d_classes : r.d_classes;
  uint64_t *const dw = m.d_words ? m.d_words : r.d_words;
  int16_t *const ds = m.d_scores ? m.d_scores : r.d_scores;
  const bool want_words = (words || words_view) && !r.spec.is_cnv;
  if (classes && !m.d_classes && grow_pinned(r.h_classes, r.h_classes_cap, (size_t)n)) return -1;
  if (want_words && !m.d_words && grow_pinned(r.h_words, r.h_words_cap, (size_t)n)) return -1;
  DrainOnFailure drain;
  trace().mark("reserved");
  // (experiment switch, read per call so that tools/plan_ab.py can interleave it: no timing events around the chunks,
  // usecPerImage reported as 0 -- what the two marker packets per chunk cost a device-bound call)
  const bool timed = std::getenv("BNN_MI355X_NO_CHUNK_TIMING") == nullptr;
  auto stages = [&](int c, int base, int mm, int slot) {
    trace().mark("chunk_in", c);
    const int rc = enqueue(r.d_images[slot], mm, ncls, classes ? dc + base : nullptr, want_scores ? ds + (size_t)base * 64 : nullptr, dw + base,
                           lane_stream(c, lanes), timed ? r.time_events[2 * c] : nullptr, timed ? r.time_events[2 * c + 1] : nullptr,
                           lanes == 2 ? (c & 1) : 0, nchunks == 1, packed);
    trace().mark("queued", c);
    return rc;
  };
  if (ring) {
    if (feed_chunks(src.mem, from_file ? src.file->fd : -1, first, rec, skip, plan, lanes, packed, stages)) return -1;
  } else if (from_file) {
    if (stream_file(*src.file, n, true, lanes, [&](int, int slot) { return r.d_images[slot]; }, stages)) return -1;
  } else if (nchunks == 1) {  // nothing to overlap: stay on one stream (fewer driver round trips for small calls)
    HIP_OK(hipMemcpyAsync(r.d_images[0], src.mem, (size_t)n * isz, hipMemcpyHostToDevice, r.stream));
    if (stages(0, 0, n, 0)) return -1;
  } else {
    static const bool one_thread = std::getenv("BNN_MI355X_NO_COPIER") != nullptr;  // A/B: copies and launches on the calling thread
    Copier &K = copier();
    struct End {
      Copier *k;
      ~End() { if (k) k->end(); }
    } end_guard{one_thread ? nullptr : &K};
    if (!one_thread) K.begin(src.mem, plan, isz, nslots, r.device);  // (copies of chunks 1..: "the copies on a thread of their own" above)
    for (int c = 0; c < nchunks; c++) {
      const int base = plan[c], mm = plan[c + 1] - plan[c], slot = c % nslots;
      if (one_thread || c == 0) {
        if (c >= nslots) HIP_OK(hipStreamWaitEvent(r.copy_stream, r.consumed[slot], 0));
        HIP_OK(hipMemcpyAsync(r.d_images[slot], src.mem + (size_t)base * isz, (size_t)mm * isz, hipMemcpyHostToDevice, r.copy_stream));
        HIP_OK(hipEventRecord(r.copied[slot], r.copy_stream));
        if (!one_thread) K.copied.store(1, std::memory_order_release);
      } else {
        while (K.copied.load(std::memory_order_acquire) <= c) {
          if (K.failed.load(std::memory_order_acquire)) return fail(K.err);
          std::this_thread::yield();
        }
      }
      HIP_OK(hipStreamWaitEvent(lane_stream(c, lanes), r.copied[slot], 0));
      if (stages(c, base, mm, slot)) return -1;
      if (c + nslots < nchunks) HIP_OK(hipEventRecord(r.consumed[slot], lane_stream(c, lanes)));  // (a later chunk reuses the slot)
      if (!one_thread) K.consumed.store(c + 1, std::memory_order_release);
    }
  }
  if (join_lanes(lanes)) return -1;
  trace().mark("all_queued");
  if (classes && !m.d_classes) HIP_OK(hipMemcpyAsync(r.h_classes, r.d_classes, (size_t)n * 4, hipMemcpyDeviceToHost, r.stream));
  if (want_scores && !m.d_scores) HIP_OK(hipMemcpyAsync(scores, r.d_scores, (size_t)n * 128, hipMemcpyDeviceToHost, r.stream));
  if (want_words && !m.d_words) HIP_OK(hipMemcpyAsync(r.h_words, r.d_words, (size_t)n * 8, hipMemcpyDeviceToHost, r.stream));
  HIP_OK(hipStreamSynchronize(r.stream));
  if (from_file) HIP_OK(hipStreamSynchronize(r.copy_stream));
  trace().mark("synced");
  if (classes) std::memcpy(classes, m.d_classes ? m.h_classes : r.h_classes, (size_t)n * 4);
  if (want_scores && m.d_scores) std::memcpy(scores, m.h_scores, (size_t)n * 128);
  if (words && !r.spec.is_cnv) std::memcpy(words, m.d_words ? m.h_words : r.h_words, (size_t)n * 8);
  if (words_view) *words_view = m.d_words ? m.h_words : r.h_words;
  double total_ms = 0.0;
  if (timed && chunks_device_ms(nchunks, &total_ms)) return -1;
  if (usec) *usec = (float)(total_ms * 1000.0 / n);
  drain.ok();
  trace().mark("done");
  trace().dump(ring ? (packed ? "ring, host-binarised" : "ring") : (from_file ? "pageable file chunks" : "pageable buffer"));
  return 0;
}
int infer_host(const uint8_t *imgs, int n, int ncls, int32_t *classes, int16_t *scores, uint64_t *words, float *usec, const uint64_t **words_view = nullptr) {
  Source s;
  s.mem = imgs;
  return infer_any(s, n, ncls, classes, scores, words, usec, words_view);
}
int infer_file(const ImageFile &f, int n, int ncls, int32_t *classes, int16_t *scores, uint64_t *words, float *usec, const uint64_t **words_view = nullptr) {
  Source s;
  s.file = &f;
  return infer_any(s, n, ncls, classes, scores, words, usec, words_view);
}

// all images of a file, packed, resident in HBM at r.d_all (fault campaigns classify them in runs)
int load_file_resident(const ImageFile &f, int n) {
  Runtime &r = rt();
  const size_t isz = (size_t)r.spec.image_bytes();
  if (bind_device() || grow(r.d_all, r.all_cap, (size_t)n * isz + 256)) return -1;
  DrainOnFailure drain;
  if (stream_file(
          f, n, false, 1, [&](int base, int) { return r.d_all + (size_t)base * isz; }, [&](int, int, int, int) { return 0; }))
    return -1;
  HIP_OK(hipStreamSynchronize(r.copy_stream));
  drain.ok();
  return 0;
}

// ---- LFC host decode (libm, like the reference) -------------------------------
uint64_t label_mask(int ncls) { return 0xFFFFFFFFFFFFFFFFull >> (64 - ncls); }
// batched: (unsigned) log2((double) word), 0 when no bit is set (foldedmv-offload.cpp:213-220)
int lfc_class_batched(uint64_t w, int ncls) {
  w &= label_mask(ncls);
  // below 2^47 the double-precision log2 cannot round up to the next integer (2^k - 1 lies 1.44 * 2^-k below k, an ulp
  // of k is 2^-47 for k >= 32): its truncation IS the index of the highest set bit -- without 131 072 calls into libm
  // per LFC batch (0.8 ms of a 1.9 ms call, BNN_MI355X_TRACE).  Wider words go through libm like the reference's.
  if ((w >> 47) == 0) return w ? 63 - __builtin_clzll(w) : 0;
  return (int)(unsigned int)std::log2((double)w);
}
// single: index of the one-hot entry = round(log2(word)) (foldedmv-offload.cpp:152-165)
int lfc_hot_single(uint64_t w, int ncls) {
  w &= label_mask(ncls);
  return w ? (int)(unsigned int)std::round(std::log2((double)w)) : 0;
}

// the result array of inference_multiple: class indices, or (CNV, enable_detail) number_class scores
// per image.  `infer(classes, scores, words)` runs the batch from wherever the images are.
template <typename Infer>
int *classify_with(Infer infer, int n, int ncls, int enable_detail) {
  Runtime &r = rt();
  int *result = nullptr;
  if (r.spec.is_cnv && enable_detail) {
    std::vector<int16_t> s((size_t)n * 64);
    if (infer(nullptr, s.data(), nullptr, nullptr)) return nullptr;
    result = new (std::nothrow) int[(size_t)(n > 0 ? n : 1) * ncls];
    if (!result) { fail("out of memory"); return nullptr; }
    for (int i = 0; i < n; i++)
      for (int j = 0; j < ncls; j++) result[(size_t)i * ncls + j] = s[(size_t)i * 64 + j];
  } else if (r.spec.is_cnv) {
    result = new (std::nothrow) int[(size_t)(n > 0 ? n : 1)];
    if (!result) { fail("out of memory"); return nullptr; }
    if (infer(result, nullptr, nullptr, nullptr)) { delete[] result; return nullptr; }
  } else if (ncls <= 47) {
    // the LFC libraries ignore enable_detail (lfcW1A1/sw/main_python.cpp:135-156).  Up to 47 classes the reference's
    // (unsigned) log2((double) word) IS the index of the highest set bit (lfc_class_batched above), which the last stage
    // computes on the device: the classes come back as they are, no pass over 131 072 words on the host after the wait
    result = new (std::nothrow) int[(size_t)(n > 0 ? n : 1)];
    if (!result) { fail("out of memory"); return nullptr; }
    if (infer(result, nullptr, nullptr, nullptr)) { delete[] result; return nullptr; }
  } else {
    const uint64_t *w = nullptr;  // the raw output words, where the call left them (pinned memory)
    if (infer(nullptr, nullptr, nullptr, &w)) return nullptr;
    result = new (std::nothrow) int[(size_t)(n > 0 ? n : 1)];
    if (!result) { fail("out of memory"); return nullptr; }
    for (int i = 0; i < n; i++) result[i] = lfc_class_batched(w[i], ncls);
  }
  return result;
}

int *classify_host(const uint8_t *imgs, int n, int ncls, float *usec, int enable_detail) {
  return classify_with([&](int32_t *c, int16_t *s, uint64_t *w, const uint64_t **v) { return infer_host(imgs, n, ncls, c, s, w, usec, v); }, n, ncls,
                       enable_detail);
}

}  // namespace
}  // namespace bnn

#ifdef BNN_LFC_STAMPS
namespace bnn { hipError_t lfc_stamps_read(unsigned long long *dst); hipError_t lfc_wstamps_read(unsigned long long *dst); }
#endif
using namespace bnn;

// ============================================================================ C ABI
extern "C" {

void load_parameters(const char *path) {
  Runtime &r = rt();
  std::printf("Setting network weights and thresholds in accelerator...\n");
  RawParams raw;
  const std::string e = read_raw_params(r.spec, path ? path : "", raw);
  if (!e.empty()) { fail(e); return; }
  r.raw = std::move(raw);
  pack_blob(r.spec, r.raw, r.blob);
  if (upload_blob()) { r.blob.clear(); return; }
  r.err.clear();
  (void)warm_up();  // failure is reported (stderr, last_error) but the parameters are loaded
}

int inference(const char *path, int results[64], int number_class, float *usecPerImage) {
  Runtime &r = rt();
  if (!ready()) return -1;
  ImageFile f;
  trace().start();
  if (open_image_file(path, f)) return -1;
  trace().mark("opened");
  if (f.n == 0) return fail("no image in input file");
  float usec = 0.f;
  int cls;
  if (r.spec.is_cnv) {
    // testPrebuiltCIFAR10_from_image: count = 1 (foldedmv-offload.h:318): the first record of the file
    int16_t s[64];
    if (infer_file(f, 1, number_class, nullptr, s, nullptr, &usec)) return -1;
    if (results)
      for (int j = 0; j < number_class; j++) results[j] = s[j];
    cls = 0;
    for (int j = 1; j < number_class; j++)
      if (s[j] > s[cls]) cls = j;  // std::max_element: first maximum
  } else {
    uint64_t w = 0;
    if (infer_file(f, 1, number_class, nullptr, nullptr, &w, &usec)) return -1;
    const int hot = lfc_hot_single(w, number_class);
    if (results)
      for (int i = 0; i < 64; i++) results[i] = (i == hot) ? 1 : 0;
    cls = hot < 64 ? hot : 0;
  }
  std::printf("Inference took %.0f microseconds, %g usec per image\n", usec, usec);
  std::printf("Classification rate: %g images per second\n", 1000000.0 / usec);
  if (usecPerImage) *usecPerImage = usec;
  return cls;
}

int *inference_multiple(const char *path, int number_class, int *image_number, float *usecPerImage,
                        int enable_detail) {
  if (!ready()) return nullptr;
  ImageFile f;
  trace().start();
  if (open_image_file(path, f)) return nullptr;
  trace().mark("opened");
  const int n = (int)f.n;
  float usec = 0.f;
  int *res = classify_with([&](int32_t *c, int16_t *s, uint64_t *w, const uint64_t **v) { return infer_file(f, n, number_class, c, s, w, &usec, v); }, n,
                           number_class, enable_detail);
  if (!res) return nullptr;
  std::printf("Inference took %.0f microseconds, %g usec per image\n", usec * n, usec);
  std::printf("Classification rate: %g images per second\n", 1000000.0 / usec);
  if (image_number) *image_number = n;
  if (usecPerImage) *usecPerImage = usec;
  return res;
}

int *inference_multiple_with_faults(const char *path, int number_class, int *image_number,
                                    float *usecPerImage, unsigned int flip_count, int word_size,
                                    int target, int *target_layers, unsigned int num_targets) {
  Runtime &r = rt();
  if (flip_count == 0) return inference_multiple(path, number_class, image_number, usecPerImage, 0);
#ifdef BNN_VARIANT
  // TMRBinaryWeights / the interleaved memories live in the un-vendored finn-hlslib fork: their voting
  // and de-interleaving are not restated here, so a fault campaign would not mean what the caller expects
  fail("fault injection is not modelled for " BNN_VARIANT " (replicated / interleaved parameter memories); use the base network");
  return nullptr;
#endif
  if (!ready()) return nullptr;
  if (r.raw.empty()) {
    fail("fault injection needs the parameter files (load_parameters), not an imported blob");
    return nullptr;
  }
  if (r.l1_mfma || r.l1_literal) {
    fail("fault injection is not wired to the BNN_MI355X_L1 comparison forms (the matrix-pipe table is not patched)");
    return nullptr;
  }
  ImageFile f;
  if (open_image_file(path, f)) return nullptr;
  const int n = (int)f.n;
  // The reference classifies image by image and injects each fault just before the image it was
  // drawn for (faults.h:115-148).  Same result, fewer launches: the images go to HBM once, the run
  // of images between two fault times is classified as one batch, and a fault patches the one
  // affected row of the blob in HBM -- all of it queued on one stream, one wait at the end.
  {
    const std::string pe = plan_faults(r.spec, r.fault_seed, n, flip_count, word_size, target, target_layers, num_targets, r.last_faults);
    if (!pe.empty()) { fail(pe); return nullptr; }
  }
  int *result = new (std::nothrow) int[(size_t)(n > 0 ? n : 1)];
  if (!result) { fail("out of memory"); return nullptr; }
  auto run = [&]() -> int {
    if (n == 0) return 0;
    const size_t isz = (size_t)r.spec.image_bytes();
    if (load_file_resident(f, n)) return -1;
    if (reserve(n) || reserve_host(1, (size_t)n)) return -1;
    std::vector<std::vector<uint8_t>> patches;  // row images must outlive their asynchronous upload
    patches.reserve(r.last_faults.size());
    DrainOnFailure drain;  // (declared after `patches`: a failing return drains the stream before they are released)
    // one pair of events around the whole campaign: the row patches between the runs are a few hundred
    // bytes each, and an event pair per run would cost more than they do
    while (r.time_events.size() < 2) {
      hipEvent_t e;
      HIP_OK(hipEventCreateWithFlags(&e, kTimeEventFlags));
      r.time_events.push_back(e);
    }
    HIP_OK(hipEventRecord(r.time_events[0], r.stream));
    size_t k = 0;
    int start = 0;
    while (start < n) {
      while (k < r.last_faults.size() && r.last_faults[k].image <= start) {
        const Fault &flt = r.last_faults[k++];
        const int row = apply_fault(r.spec, r.raw, flt);
        if (row < 0) continue;
        size_t off = 0, bytes = 0;
        repack_row(r.spec, r.raw, flt.layer, row, r.blob, &off, &bytes);
        if (r.spec.L[flt.layer].arith == AR_TT) r.two_rows = count_two_rows(r.spec, r.blob);
        patches.emplace_back(r.blob.begin() + off, r.blob.begin() + off + bytes);
        HIP_OK(hipMemcpyAsync(static_cast<uint8_t *>(r.d_blob) + off, patches.back().data(), bytes, hipMemcpyHostToDevice, r.stream));
      }
      const int end = (k < r.last_faults.size()) ? r.last_faults[k].image : n;
      for (int base = start; base < end; base += kMaxChunk) {
        const int m = (end - base < kMaxChunk) ? end - base : kMaxChunk;
        if (enqueue(r.d_all + (size_t)base * isz, m, number_class, r.d_classes + base, nullptr, r.d_words + base, r.stream)) return -1;
      }
      start = end;
    }
    HIP_OK(hipEventRecord(r.time_events[1], r.stream));
    std::vector<uint64_t> w;
    if (r.spec.is_cnv) {
      HIP_OK(hipMemcpyAsync(result, r.d_classes, (size_t)n * 4, hipMemcpyDeviceToHost, r.stream));
    } else {  // host decode, like the batched LFC entry point
      w.resize((size_t)n);
      HIP_OK(hipMemcpyAsync(w.data(), r.d_words, (size_t)n * 8, hipMemcpyDeviceToHost, r.stream));
    }
    HIP_OK(hipStreamSynchronize(r.stream));
    for (int i = 0; !r.spec.is_cnv && i < n; i++) result[i] = lfc_class_batched(w[i], number_class);
    float ms_total = 0.f;
    HIP_OK(hipEventElapsedTime(&ms_total, r.time_events[0], r.time_events[1]));
    drain.ok();
    return (int)(ms_total * 1000.0 + 0.5);  // device microseconds of the campaign
  };
  const int total_us = run();
  if (total_us < 0) {
    delete[] result;
    return nullptr;
  }
  const float usec = n > 0 ? (float)total_us / (float)n : 0.f;
  std::printf("Inference took %.0f microseconds, %g usec per image\n", (double)total_us, usec);
  std::printf("Classification rate: %g images per second\n", 1000000.0 / usec);
  if (image_number) *image_number = n;
  if (usecPerImage) *usecPerImage = usec;
  return result;
}

void free_results(int *result) { delete[] result; }

void deinit(void) {
  free_workspace();
  rt().warmed = false;  // the buffers the warm-up allocated are gone: the next load warms again
}

const char *bnn_mi355x_network(void) {
#ifdef BNN_VARIANT
  return BNN_VARIANT;
#else
  return rt().spec.name;
#endif
}
int bnn_mi355x_image_bytes(void) { return rt().spec.image_bytes(); }
const char *bnn_mi355x_last_error(void) { return rt().err.c_str(); }

int bnn_mi355x_set_device(int ordinal) {
  Runtime &r = rt();
  if (r.d_blob || r.cap || r.stage_cap || r.stream) {
    if (ordinal == r.device) return 0;  // already bound to exactly this device: nothing to do
    return fail("set_device must be called before load_parameters (this library is bound to device " + std::to_string(r.device) + ")");
  }
  if (ordinal < 0) return fail("set_device: negative ordinal");
  r.device = ordinal;
  return 0;
}

size_t bnn_mi355x_pack_params(const char *path, void *dst, size_t cap) {
  std::vector<uint8_t> blob;
  const std::string e = pack_params_from_dir(rt().spec, path ? path : "", blob);
  if (!e.empty()) { fail(e); return 0; }
  if (dst) {
    if (cap < blob.size()) { fail("pack_params: destination too small"); return 0; }
    std::memcpy(dst, blob.data(), blob.size());
  }
  return blob.size();
}

size_t bnn_mi355x_export_params(void *dst, size_t cap) {
  Runtime &r = rt();
  if (r.blob.empty()) { fail("export_params: nothing loaded"); return 0; }
  if (dst) {
    if (cap < r.blob.size()) { fail("export_params: destination too small"); return 0; }
    std::memcpy(dst, r.blob.data(), r.blob.size());
  }
  return r.blob.size();
}

int bnn_mi355x_import_params(const void *src, size_t bytes) {
  Runtime &r = rt();
  const std::string e = validate_blob(r.spec, src, bytes);
  if (!e.empty()) return fail(e);
  r.blob.assign(static_cast<const uint8_t *>(src), static_cast<const uint8_t *>(src) + bytes);
  r.raw = RawParams{};
  if (upload_blob()) { r.blob.clear(); return -1; }
  r.err.clear();
  (void)warm_up();  // like load_parameters: a failing warm-up is reported (stderr, last_error), the parameters are loaded
  return 0;
}

size_t bnn_mi355x_params_bytes(void) { return blob_bytes(rt().spec); }

int bnn_mi355x_import_params_device(const void *d_src, size_t bytes, void *hip_stream) {
  Runtime &r = rt();
  if (!d_src || bytes != blob_bytes(r.spec)) return fail("import_params_device: size is not this network's blob size");
  if (bind_device()) return -1;
  // The host keeps a copy as well: it validates the header before anything on the device trusts it, and
  // export_params / the fault machinery read it.  ~0.2-0.8 MB once per load.
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  std::vector<uint8_t> host(bytes);
  HIP_OK(hipMemcpyAsync(host.data(), d_src, bytes, hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  const std::string e = validate_blob(r.spec, host.data(), bytes);
  if (!e.empty()) return fail(e);
  r.blob.swap(host);
  r.raw = RawParams{};
  if (upload_blob()) { r.blob.clear(); return -1; }
  r.err.clear();
  (void)warm_up();  // (as above)
  return 0;
}

unsigned int bnn_mi355x_params_crc(void) {
  Runtime &r = rt();
  if (r.blob.empty() || !r.d_blob) { fail("params_crc: nothing loaded"); return 0; }
  // of the bytes the GPU actually holds, read back -- not of the host copy
  std::vector<uint8_t> dev(r.d_blob_bytes);
  if (bind_device() || hipStreamSynchronize(r.stream) != hipSuccess ||
      hipMemcpy(dev.data(), r.d_blob, dev.size(), hipMemcpyDeviceToHost) != hipSuccess) {
    fail("params_crc: read-back failed");
    return 0;
  }
  uint32_t crc = 0xFFFFFFFFu;  // CRC-32 (IEEE 802.3, reflected), bitwise: 200 KB once
  for (uint8_t b : dev) {
    crc ^= b;
    for (int k = 0; k < 8; k++) crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1u)));
  }
  return ~crc;
}

int *bnn_mi355x_inference_buffer(const uint8_t *images, int n_images, int number_class, float *usecPerImage,
                                 int enable_detail) {
  if (!ready()) return nullptr;
  if (n_images < 0 || (n_images > 0 && !images)) { fail("inference_buffer: bad arguments"); return nullptr; }
  return classify_host(images, n_images, number_class, usecPerImage, enable_detail);
}

int bnn_mi355x_binarize_pack(const uint8_t *images, int n_images, uint64_t *words) {
  if (rt().spec.is_cnv) return fail("binarize_pack: the CNV networks take 8-bit inputs (quantised on the GPU)");
  if (n_images < 0 || (n_images > 0 && (!images || !words))) return fail("binarize_pack: bad arguments");
  binarize_pack(images, (size_t)n_images, words);
  return 0;
}

int bnn_mi355x_inference_raw(const uint8_t *images, int n_images, int16_t *scores, uint64_t *words,
                             float *usecPerImage) {
  if (!ready()) return -1;
  if (n_images < 0 || (n_images > 0 && !images)) return fail("inference_raw: bad arguments");
  return infer_host(images, n_images, 64, nullptr, scores, words, usecPerImage);
}

int bnn_mi355x_reserve(int max_images) {
  if (!ready() || reserve(max_images)) return -1;
  // (a CNV pass of kForkMin images and more runs its second half on the second lane: size that workspace now as well, so
  // that the call itself stays free of allocations)
  const int m = max_images < kMaxChunk ? max_images : kMaxChunk;
  if (rt().spec.is_cnv && m >= kForkMin && lanes_for(3) == 2) return reserve2((m + 1) / 2);  // (>= the second half of any pass up to m)
  return 0;
}

#ifdef BNN_LFC_STAMPS
// diagnostic build only: 1024 blocks x 8 wall-clock stamps (100 MHz) of the last k_lfc_block_s launch
int bnn_mi355x_debug_lfc_stamps(unsigned long long *dst) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return bnn::lfc_stamps_read(dst) == hipSuccess ? 0 : -1;
}
// ... and 1024 blocks x 16 waves x 16 stamps
int bnn_mi355x_debug_lfc_wstamps(unsigned long long *dst) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return bnn::lfc_wstamps_read(dst) == hipSuccess ? 0 : -1;
}
#endif

int bnn_mi355x_chunk_plan(int n_images, int from_file, int *bases, int cap) {
  if (n_images < 0) return fail("chunk_plan: bad arguments");
  const std::vector<int> plan = plan_chunks(n_images, false, from_file != 0);
  for (size_t i = 0; i < plan.size() && (int)i < cap && bases; i++) bases[i] = plan[i];
  return (int)plan.size();
}

long bnn_mi355x_debug_stage_output(const uint8_t *images, int n_images, int stage, void *dst, size_t cap) {
  Runtime &r = rt();
  if (!ready()) return -1;
  int in_buf1 = 0;
  const size_t per = stage_output_bytes(r.spec.is_cnv, r.spec.abits, stage, &in_buf1);
  if (per == 0 || n_images <= 0 || n_images > kHostChunk || !images || !dst || cap < per * (size_t)n_images)
    return fail("debug_stage_output: bad arguments");
  r.debug_last_stage = stage;
  std::vector<uint64_t> w((size_t)n_images);
  const int rc = infer_host(images, n_images, 10, nullptr, nullptr, r.spec.is_cnv ? nullptr : w.data(), nullptr);
  r.debug_last_stage = -1;
  if (rc) return -1;
  HIP_OK(hipMemcpy(dst, in_buf1 ? r.buf1 : r.buf0, per * (size_t)n_images, hipMemcpyDeviceToHost));
  return (long)per;
}

int bnn_mi355x_plan_faults(unsigned long long seed, int num_images, unsigned int flip_count, int word_size, int target,
                           const int *target_layers, unsigned int num_targets, int *records, int cap_records) {
  std::vector<Fault> plan;
  const std::string pe = plan_faults(rt().spec, seed, num_images, flip_count, word_size, target, target_layers, num_targets, plan);
  if (!pe.empty()) return fail(pe);
  for (size_t i = 0; i < plan.size() && (int)i < cap_records; i++) {
    const Fault &f = plan[i];
    const int v[8] = {f.image, f.target, f.layer, f.mem, f.ind, f.thresh, f.bit, f.word_size};
    for (int j = 0; j < 8; j++) records[i * 8 + j] = v[j];
  }
  return (int)plan.size();
}

size_t bnn_mi355x_pack_params_faulty(const char *path, const int *records, int n_faults, void *dst, size_t cap) {
  RawParams raw;
  const std::string e = read_raw_params(rt().spec, path ? path : "", raw);
  if (!e.empty()) { fail(e); return 0; }
  for (int i = 0; i < n_faults; i++) {
    const int *v = records + i * 8;
    const Fault f{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
    if (apply_fault(rt().spec, raw, f) < 0) { fail("pack_params_faulty: fault record out of range"); return 0; }
  }
  std::vector<uint8_t> blob;
  pack_blob(rt().spec, raw, blob);
  if (dst) {
    if (cap < blob.size()) { fail("pack_params_faulty: destination too small"); return 0; }
    std::memcpy(dst, blob.data(), blob.size());
  }
  return blob.size();
}

int bnn_mi355x_set_fault_seed(unsigned long long seed) {
  rt().fault_seed = seed;
  return 0;
}

int bnn_mi355x_last_faults(int *records, int cap_records) {
  Runtime &r = rt();
  const int n = (int)r.last_faults.size();
  for (int i = 0; i < n && i < cap_records; i++) {
    const Fault &f = r.last_faults[i];
    const int v[8] = {f.image, f.target, f.layer, f.mem, f.ind, f.thresh, f.bit, f.word_size};
    for (int j = 0; j < 8; j++) records[i * 8 + j] = v[j];
  }
  return n;
}

int bnn_mi355x_thumbnail_size(int width, int height, int *out_w, int *out_h) {
  int ow = width, oh = height;
  const bool resized = width > 0 && height > 0 && thumbnail_size(width, height, 32, &ow, &oh);
  if (out_w) *out_w = ow;
  if (out_h) *out_h = oh;
  return resized ? 1 : 0;
}

int bnn_mi355x_images_to_cifar(const uint8_t *const *pixels, const int *widths, const int *heights, const int *bands,
                               const long *row_strides, int n_images, uint8_t *records) {
  Runtime &r = rt();
  if (!r.spec.is_cnv) return fail("images_to_cifar: CIFAR-10 records are the input of the CNV networks only");
  if (n_images < 0 || (n_images > 0 && (!pixels || !widths || !heights || !bands || !records)))
    return fail("images_to_cifar: bad arguments");
  if (n_images == 0) return 0;
  if (bind_device()) return -1;
  if (grow(r.d_pp_rec, r.pp_rec_cap, (size_t)n_images * 3073)) return -1;
  // host copies of the coefficient tables must outlive their (asynchronous) upload: kept until the next sync
  std::vector<std::vector<int32_t>> keep;
  // declared after `keep`, so destroyed (= streams drained on a failing exit) before it: uploads from the caller's
  // pictures and from `keep` may still be queued when an error returns
  DrainOnFailure drain;
  int last_w = -1, last_h = -1;
  size_t off_bh = 0, off_kv = 0, off_bv = 0;
  int ksize_h = 0, ksize_v = 0, ow = 0, oh = 0;
  for (int i = 0; i < n_images; i++) {
    const int w = widths[i], h = heights[i], nb = bands[i];
    if (!pixels[i] || w < 1 || h < 1 || w > 65535 || h > 65535 || (nb != 1 && nb != 3 && nb != 4))
      return fail("images_to_cifar: image " + std::to_string(i) + ": need 1..65535 x 1..65535 pixels of 1 (L), 3 (RGB) or 4 (RGBA) bytes");
    const size_t row_bytes = (size_t)w * nb;
    const long stride = row_strides ? row_strides[i] : (long)row_bytes;
    if (stride < (long)row_bytes) return fail("images_to_cifar: row stride shorter than a row");
    if (w != last_w || h != last_h) {
      ow = w; oh = h;
      (void)thumbnail_size(w, h, 32, &ow, &oh);
      std::vector<int32_t> kh, bh, kv, bv, all;
      ksize_h = lanczos_coeffs(w, ow, kh, bh);
      ksize_v = lanczos_coeffs(h, oh, kv, bv);
      off_bh = kh.size(); off_kv = off_bh + bh.size(); off_bv = off_kv + kv.size();
      all.reserve(off_bv + bv.size());
      all.insert(all.end(), kh.begin(), kh.end()); all.insert(all.end(), bh.begin(), bh.end());
      all.insert(all.end(), kv.begin(), kv.end()); all.insert(all.end(), bv.begin(), bv.end());
      if (grow(r.d_pp_coef, r.pp_coef_cap, all.size())) return -1;
      if (keep.size() >= 16) { HIP_OK(hipStreamSynchronize(r.stream)); keep.clear(); }
      keep.push_back(std::move(all));
      HIP_OK(hipMemcpyAsync(r.d_pp_coef, keep.back().data(), keep.back().size() * sizeof(int32_t), hipMemcpyHostToDevice, r.stream));
      last_w = w; last_h = h;
    }
    if (grow(r.d_pp_src, r.pp_src_cap, row_bytes * h) || grow(r.d_pp_tmp, r.pp_tmp_cap, (size_t)(h > w ? h : w) * 32 * 4 + 4096)) return -1;
    if ((size_t)stride == row_bytes)
      HIP_OK(hipMemcpyAsync(r.d_pp_src, pixels[i], row_bytes * h, hipMemcpyHostToDevice, r.stream));
    else
      HIP_OK(hipMemcpy2DAsync(r.d_pp_src, row_bytes, pixels[i], (size_t)stride, row_bytes, (size_t)h, hipMemcpyHostToDevice, r.stream));
    ResampleJob j{};
    j.src = r.d_pp_src; j.w = w; j.h = h; j.bands = nb; j.stride = (long)row_bytes;
    j.out_w = ow; j.out_h = oh;
    j.vertical_first = vertical_pass_first(w, h, oh);
    j.kh = r.d_pp_coef; j.bh = r.d_pp_coef + off_bh; j.ksize_h = ksize_h;
    j.kv = r.d_pp_coef + off_kv; j.bv = r.d_pp_coef + off_bv; j.ksize_v = ksize_v;
    j.tmp = r.d_pp_tmp;
    j.record = r.d_pp_rec + (size_t)i * 3073;
    const hipError_t e = launch_image_to_cifar(j, r.stream);
    if (e != hipSuccess) return fail(std::string("images_to_cifar: kernel launch: ") + hipGetErrorString(e));
  }
  HIP_OK(hipMemcpyAsync(records, r.d_pp_rec, (size_t)n_images * 3073, hipMemcpyDeviceToHost, r.stream));
  HIP_OK(hipStreamSynchronize(r.stream));
  drain.ok();
  return 0;
}

int bnn_mi355x_profile(int enable) {
  Runtime &r = rt();
  r.profiling = enable != 0;
  r.prof_used = 0;
  return 0;
}

int bnn_mi355x_profile_read(float *ms_per_stage, int cap, int *n_chunks) {
  Runtime &r = rt();
  const int stages = r.spec.is_cnv ? kCnvStages : kLfcStages;
  if (!ms_per_stage || cap < stages) return fail("profile_read: need room for all stages");
  for (int i = 0; i < stages; i++) ms_per_stage[i] = 0.f;
  for (size_t k = 0; k < r.prof_used; k++) {
    std::vector<hipEvent_t> &set = r.prof_sets[k];
    HIP_OK(hipEventSynchronize(set[stages]));
    for (int i = 0; i < stages; i++) {
      float ms = 0.f;
      HIP_OK(hipEventElapsedTime(&ms, set[i], set[i + 1]));
      ms_per_stage[i] += ms;
    }
  }
  if (n_chunks) *n_chunks = (int)r.prof_used;
  r.prof_used = 0;
  return stages;
}

const char *bnn_mi355x_stage_name(int stage) { return stage_name(rt().spec.is_cnv, stage); }

int bnn_mi355x_inference_device(const void *d_images, int n_images, int number_class, int32_t *d_classes,
                                int16_t *d_scores, uint64_t *d_words, void *hip_stream) {
  Runtime &r = rt();
  if (!ready()) return -1;
  if (n_images < 0 || (n_images > 0 && !d_images)) return fail("inference_device: bad arguments");
  if (number_class < 1 || number_class > 64) return fail("number_class must be in 1..64");
  // the kernels read images with 128-bit loads and write scores as dwords (include/bnn_mi355x.h states the alignment)
  auto misaligned = [](const void *p, uintptr_t a) { return p && (reinterpret_cast<uintptr_t>(p) & (a - 1)) != 0; };
  if (misaligned(d_images, 16)) return fail("inference_device: d_images must be 16-byte aligned");
  if (misaligned(d_classes, 4) || misaligned(d_scores, 4) || misaligned(d_words, 8))
    return fail("inference_device: d_classes / d_scores must be 4-byte aligned, d_words 8-byte aligned");
  // the device-side LFC decode is an exact integer floor(log2); the reference's (unsigned) log2((double) word)
  // rounds UP for some words of 48 bits and more, which only the host decode (libm) reproduces
  if (!r.spec.is_cnv && d_classes && number_class > 47)
    return fail("inference_device: device-side LFC classes need number_class <= 47 (take d_words and decode on the host)");
  if (bind_device()) return -1;  // the calling thread's current device may not be this library's
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const size_t isz = (size_t)r.spec.image_bytes();
  if (!r.spec.is_cnv && !d_words) {  // the LFC decode stage reads the raw words: give it somewhere to put them
    if (reserve_host(1, (size_t)n_images)) return -1;
    d_words = r.d_words;
  }
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &capturing) != hipSuccess) capturing = hipStreamCaptureStatusNone;
  // one pass of at most kMaxChunk images; returns -1 with the error in last_error
  auto pass = [&](int base, int m) -> int {
    const uint8_t *img = static_cast<const uint8_t *>(d_images) + (size_t)base * isz;
    int32_t *cls = d_classes ? d_classes + base : nullptr;
    int16_t *sc = d_scores ? d_scores + (size_t)base * 64 : nullptr;
    uint64_t *wd = d_words ? d_words + base : nullptr;
    // A pass of a CNV net forks over the two compute lanes: first half on the caller's stream, second half on the
    // library's second stream (its own activation workspace), joined before the call returns -- one half's launch gaps
    // and kernel tails are filled by the other's kernels.  tools/two_lane_probe.py (HALVES=2), profiles/
    // r03_device_call_fork_probe.txt: 131 072 images 10.28 -> 10.16 ms, 65 536 5.26 -> 5.13, 32 768 2.69 -> 2.60, 20 000
    // 1.66 -> 1.63; 10 000 images and the LFC nets (one launch per pass) lose, four or eight pieces gain less.  Not while the
    // caller's stream is being captured into a graph, nor with stage profiling on (the stage events would time overlapping
    // kernels), nor under BNN_MI355X_LANES=1.
    const bool fork = r.spec.is_cnv && m >= kForkMin && lanes_for(3) == 2 && capturing == hipStreamCaptureStatusNone;
    if (!fork) return (reserve(m) || enqueue(img, m, number_class, cls, sc, wd, s)) ? -1 : 0;
    const int h = ((m / 2) + 255) & ~255;  // whole blocks of 256 images in the first half
    // (the first workspace is sized for the WHOLE pass although this call uses half of it: the usual "warm-up call, then
    // capture a call of the same size" sequence runs the captured call unforked, and it must not allocate while capturing)
    if (reserve(m) || reserve2(m - h)) return -1;
    if (settle_handover(s)) return -1;  // (before the fork: the second lane inherits the wait through the fork event)
    HIP_OK(hipEventRecord(r.fork_ev, s));
    HIP_OK(hipStreamWaitEvent(r.stream2, r.fork_ev, 0));
    r.ws_seen[1] = r.ws_gen;
    // From here on the second lane is tied to the caller's stream: whatever fails, the join still happens -- work already
    // queued on the library's own stream must not outlive the call unseen (the next call on another stream waits for
    // ws_event, which is recorded on `s` only).
    const bool ok = enqueue(img, h, number_class, cls, sc, wd, s) == 0 &&
                    enqueue(img + (size_t)h * isz, m - h, number_class, cls ? cls + h : nullptr, sc ? sc + (size_t)h * 64 : nullptr,
                            wd ? wd + h : nullptr, r.stream2, nullptr, nullptr, 1) == 0;
    const std::string why = r.err;
    const bool joined = hipEventRecord(r.lane2_done, r.stream2) == hipSuccess && hipStreamWaitEvent(s, r.lane2_done, 0) == hipSuccess;
    if (!joined) (void)hipStreamSynchronize(r.stream2);  // the join itself failed: settle it on the host
    if (!ok) return fail(why);
    return joined ? 0 : fail("inference_device: joining the second compute lane failed");
  };
  int rc = 0;
  for (int base = 0; base < n_images && rc == 0; base += kMaxChunk) rc = pass(base, (n_images - base < kMaxChunk) ? n_images - base : kMaxChunk);
  // mark the end of this call's use of the shared workspace -- on a failing exit too: passes queued before the failure
  // still run (not while the stream is being captured into a graph: a replayed graph is serialised by its own stream, and
  // the caller must not replay it concurrently with other calls into this library -- include/bnn_mi355x.h)
  if (n_images > 0 && s != r.stream && capturing == hipStreamCaptureStatusNone) {
    const std::string why = r.err;
    if (hipEventRecord(r.ws_event, s) == hipSuccess) {
      r.ws_last = s;
      r.ws_pending = true;
      r.ws_gen++;
    } else if (rc == 0) {
      return fail("inference_device: recording the workspace hand-over failed");
    }
    if (rc) r.err = why;
  }
  return rc;
}

}  // extern "C"
