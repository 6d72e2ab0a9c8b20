/*
 * bnn_mi355x.h -- C ABI of the MI355X-native BNN-PYNQ runtime.
 *
 * One shared object per network, named like the reference's build does
 * (bnn/src/network/make-sw.sh:106-119):
 *     <runtime>-<network>-<platform>.so      e.g. python_sw-cnvW1A1-mi355x.so
 * It drops into bnn/libraries/<platform>/ and is dlopen'ed by the reference's
 * cffi shim (bnn/bnn.py:67-79,108-111) unchanged.
 *
 * PART 1 is exactly the reference's cdef (bnn/bnn.py:69-77); the functions
 * they replace are the extern "C" entry points of
 * bnn/src/network/<net>/sw/main_python.cpp.  PART 2 are extensions for
 * callers that already hold images in memory (host or HBM) and for multi-GPU
 * parameter broadcast; they use plain pointers and sizes only.
 *
 * Error behaviour: the reference throws C++ string literals across the ABI
 * (=> std::terminate).  This library never throws: it prints the same text to
 * stderr and returns NULL / -1 (a behavioural superset).
 * Thread-safety: like the reference, one classifier per loaded .so, calls must
 * not overlap (file-static weights: top.cpp:51-68, rawhls-offload.cpp:52).
 */
#ifndef BNN_MI355X_H
#define BNN_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ PART 1 */

/* Replaces load_parameters (cnvW1A1/sw/main_python.cpp:67-82,
 * lfcW1A1/sw/main_python.cpp:65-75): reads <path>/L-P-weights.bin and
 * L-P-thres.bin (format unchanged), repacks once, uploads to HBM. */
void load_parameters(const char *path);

/* Replaces inference (main_python.cpp:120-139 / lfc 113-133).  `path` is a
 * CIFAR-10 binary file (records of 1+3072 bytes) or an MNIST idx3 file; only
 * the first image is classified.  CNV: results (may be NULL) receives
 * number_class 16-bit scores, return = first maximum.  LFC: results receives a
 * 64-entry one-hot vector at round(log2(output word)), return = that index.
 * usecPerImage (may be NULL): device compute time, inputs resident in HBM. */
int inference(const char *path, int results[64], int number_class, float *usecPerImage);

/* Replaces inference_multiple (main_python.cpp:141-169 / lfc 135-156).
 * Returns a new int[n] of class indices (CNV: first strict maximum floored at
 * 0, foldedmv-offload.h:396-408; LFC: floor(log2(word)), foldedmv-offload.cpp:
 * 202-220) or, CNV with enable_detail != 0, a new int[n*number_class] of
 * scores.  No 10 000-image cap (the reference's INPUT_BUF_ENTRIES limit,
 * foldedmv-offload.h:56-58): the batch is chunked internally.
 * Free with free_results(). */
int *inference_multiple(const char *path, int number_class, int *image_number, float *usecPerImage,
                        int enable_detail);

/* Replaces inference_multiple_with_faults (main_python.cpp:171-223).
 * flip_count == 0: classifies like inference_multiple(enable_detail = 0).
 * flip_count > 0: flip_count faults (word_size adjacent bits of one weight or
 * threshold memory word; target < 0 any, 0 weights, > 0 thresholds; optional
 * list of target layers) at uniformly drawn image indices and memory positions,
 * selection weighted by memory size, addressed and applied exactly like
 * inject_fault / inject_fault_impl (foldedmv-offload.h:146-214).  Faults stay in
 * the loaded parameters until the next load_parameters.  The reference seeds
 * from std::random_device; see bnn_mi355x_set_fault_seed. */
int *inference_multiple_with_faults(const char *path, int number_class, int *image_number,
                                    float *usecPerImage, unsigned int flip_count, int word_size,
                                    int target, int *target_layers, unsigned int num_targets);

/* Replaces free_results (main_python.cpp:225-227). */
void free_results(int *result);

/* Replaces deinit (main_python.cpp:229-231, FoldedMVDeinit): frees the I/O
 * workspace; the loaded parameters stay, as in the reference. */
void deinit(void);

/* ------------------------------------------------------------------ PART 2 */

/* network compiled into this .so ("cnvW1A1", ...), and bytes per input image
 * (3072 planar CHW uint8 for cnv*, 784 for lfc*) */
const char *bnn_mi355x_network(void);
int bnn_mi355x_image_bytes(void);

/* last error text of this library ("" if none) */
const char *bnn_mi355x_last_error(void);

/* select the GPU (HIP ordinal) used by this library instance; call before
 * load_parameters.  Returns 0 on success.  Without the call the library binds to the calling thread's
 * current device at its first use.  Once bound, set_device(the bound ordinal) is a no-op returning 0 and any
 * other ordinal is an error (-1 + last_error). */
int bnn_mi355x_set_device(int ordinal);

/* Packed parameter blob (position-independent bytes, see csrc/packed_params.h).
 * pack: param directory -> blob, host only, touches no GPU (dst may be NULL to
 *       query the size); returns bytes or 0 on error.
 * export: the blob currently loaded; import: upload a blob obtained elsewhere,
 *       e.g. received through an RCCL broadcast from rank 0. */
size_t bnn_mi355x_pack_params(const char *path, void *dst, size_t cap);
size_t bnn_mi355x_export_params(void *dst, size_t cap);
int bnn_mi355x_import_params(const void *src, size_t bytes);
/* params_bytes: the size of this network's blob -- a function of the topology alone, so every rank of a
 *       multi-GPU job can allocate the receive buffer of the ONE broadcast without a size exchange.
 * import_params_device: like import_params, but the blob lies in HBM (e.g. the tensor RCCL broadcast into);
 *       d_src must be on this library's device, hip_stream is the stream that produced it (NULL = default).
 *       The header is validated on the host before anything uses it.
 * params_crc: CRC-32 of the parameter bytes the GPU holds (read back from HBM; 0 + last_error when nothing
 *       is loaded): lets the ranks of a job prove that they classify with identical parameters. */
size_t bnn_mi355x_params_bytes(void);
int bnn_mi355x_import_params_device(const void *d_src, size_t bytes, void *hip_stream);
unsigned int bnn_mi355x_params_crc(void);

/* Classify n images held in HOST memory (n x image_bytes, same byte layout as
 * the bodies of the file formats).  Same return convention and ownership as
 * inference_multiple. */
int *bnn_mi355x_inference_buffer(const uint8_t *images, int n_images, int number_class,
                                 float *usecPerImage, int enable_detail);

/* The LFC networks' input hand-over as the reference's host performs it (binarizeAndPack,
 * bnn/src/library/host/foldedmv-offload.cpp:82-98, called per image at :186-188): n images of 784 uint8 pixels ->
 * n x 13 little-endian 64-bit words, bit i = (pixel i >= 128), bits 784..831 zero.  This is what the entry points
 * that take HOST data (inference_multiple, inference_buffer, inference_raw) run on worker threads so that 104 bytes
 * per image cross the PCIe link instead of 784 (images already in HBM -- inference_device -- are binarised by the
 * kernels).  Host only, touches no GPU; LFC libraries only (-1 + last_error on a CNV library).  Returns 0. */
int bnn_mi355x_binarize_pack(const uint8_t *images, int n_images, uint64_t *words);

/* Raw outputs for n host images: CNV scores[n*64] (16-bit, all 64 neurons of
 * layer 8), LFC words[n] (raw 64-bit output word).  Either may be NULL.
 * Returns 0 on success. */
int bnn_mi355x_inference_raw(const uint8_t *images, int n_images, int16_t *scores, uint64_t *words,
                             float *usecPerImage);

/* Classify n images already resident in HBM, asynchronously on `hip_stream`
 * (a hipStream_t, NULL = the default stream).  d_classes: int32[n] (batched
 * decode); d_scores (CNV, optional): int16[n*64]; d_words (LFC, optional):
 * uint64[n].  All device pointers.  Alignment: d_images 16 bytes (every image then is: 3072 and 784 are
 * multiples of 16; the kernels read them with 128-bit loads), d_classes and d_scores 4 bytes, d_words 8 bytes;
 * a misaligned pointer is refused (-1 + last_error), nothing is launched.  The workspace grows on demand (which
 * synchronises); call bnn_mi355x_reserve first to keep the call fully
 * asynchronous / graph-capturable.  Returns 0 on success.
 * A pass of a CNV net of 16 384 images and more runs its second half on a stream of the library's own (second activation
 * workspace) and joins it into `hip_stream` before anything queued after the call can run: to the caller the call is still
 * one in-order piece of work on `hip_stream` (not while that stream is being captured, then everything stays on it).
 * The activation workspaces belong to the library instance: calls are serialised on the device -- a call on another
 * stream than the previous one first waits (hipStreamWaitEvent) for that call's kernels -- so they never
 * race, but they do not overlap either (a stream handed in here must stay alive until its work is done: the
 * previous call's stream is recognised by its handle).  While `hip_stream` is being captured into a graph the
 * hand-over from an earlier call on another stream is settled on the host (the call blocks until that call's
 * kernels are done) and nothing from outside the capture is recorded into it; a captured graph must not be
 * replayed concurrently with other calls into the same library.  The calling thread's current device is switched to this library's.
 * LFC with d_classes: number_class <= 47 (the device decode is an exact floor(log2); the reference's
 * (unsigned) log2((double) word) differs from it for some words of 48 and more bits, which only the host
 * decode of inference_multiple / inference_buffer reproduces); take d_words beyond that. */
int bnn_mi355x_inference_device(const void *d_images, int n_images, int number_class, int32_t *d_classes,
                                int16_t *d_scores, uint64_t *d_words, void *hip_stream);
int bnn_mi355x_reserve(int max_images);

/* How the entry points that take HOST data (inference_multiple, inference_buffer, inference_raw) move it:
 *  - a single CIFAR image from a file, up to 32 from a host buffer, or up to 1 024 MNIST images: no transfer at all -- the image (LFC: binarised by the calling thread)
 *    is placed in pinned memory the GPU addresses, the one-launch kernels read it over the link and write their results into
 *    pinned memory; one launch, one wait.
 *  - the LFC networks otherwise: worker threads binarise (bnn_mi355x_binarize_pack) into pinned memory, 104 bytes per image
 *    cross the link; chunks of 8 192 images first, doubling up to 32 768.
 *  - the CNV networks otherwise: chunks whose transfer overlaps the previous chunks' stages: 512 images first (the first
 *    transfer is what nothing overlaps), doubling up to 4 096, then growing by half up to 16 384; a call of 8 192 ... 32 767
 *    images ramps down again at its end (512 last).  A file is read by worker threads straight into a ring of pinned pieces, the label
 *    byte of every record dropped on the way (preadv); a host buffer goes through the runtime's pageable path, its copies
 *    issued by a helper thread while the calling thread enqueues stages.
 * A call of three or more chunks runs them alternately on two internal streams ("compute lanes"), each with its own activation
 * workspace; usecPerImage is then the union of the chunks' device intervals over the images.  Results of calls up to 32 768
 * images are written by the last stage into pinned memory, larger ones come back in one transfer at the end.
 * chunk_plan writes the chunk boundaries base[0] = 0 < base[1] < ... < base[k] = n (at most cap of them) and returns k + 1
 * (from_file is accepted for compatibility: both entry points use the same plan since round 4).  Host only; results never
 * depend on the plan (tests/test_gpu_parity.py and tests/test_gpu_host_paths.py walk its edges). */
int bnn_mi355x_chunk_plan(int n_images, int from_file, int *bases, int cap);

/* Fault campaigns: fix the seed of the fault planner (0 = std::random_device like the
 * reference, the default) and read back the faults of the last
 * inference_multiple_with_faults call as records of 8 ints
 * {image, target (0 weights / 1 thresholds), layer, mem (PE), ind, thresh, bit, word_size};
 * returns the number of faults. */
int bnn_mi355x_set_fault_seed(unsigned long long seed);
int bnn_mi355x_last_faults(int *records, int cap_records);
/* Host-only helpers of the same machinery (no GPU touched): draw a fault plan; pack a parameter
 * directory with a list of fault records applied (what the GPU holds after those faults). */
int bnn_mi355x_plan_faults(unsigned long long seed, int num_images, unsigned int flip_count, int word_size, int target,
                           const int *target_layers, unsigned int num_targets, int *records, int cap_records);
size_t bnn_mi355x_pack_params_faulty(const char *path, const int *records, int n_faults, void *dst, size_t cap);

/* The step before the path: CnvClassifier.image_to_cifar (bnn/bnn.py:226-242) on the
 * device.  The reference shrinks a picture with PIL's Image.thumbnail((32, 32), ANTIALIAS) -- Lanczos-3,
 * Pillow's 8-bit fixed-point two-pass resampler -- pastes it centred on a white 32x32 canvas and writes
 * a CIFAR-10 record: label byte 1, then the R, G, B planes (3073 bytes).  These two entry points do the
 * same for decoded pictures in host memory; the records are bit-identical to Pillow's
 * (tests/test_image_to_cifar.py).  CNV libraries only.
 *   pixels[i]: heights[i] rows of widths[i] pixels of bands[i] bytes (1 = mode "L", 3 = "RGB", 4 = "RGBA":
 *              resampled with premultiplied alpha like Image.resize does, the alpha channel then dropped),
 *              rows row_strides[i] bytes apart (row_strides NULL: packed rows);
 *   records:   n_images x 3073 bytes, host.
 * Pictures of other modes (palette, LA, CMYK, ...) stay with PIL on the host, like in the reference.
 * thumbnail_size: the size Image.thumbnail((32, 32)) gives a width x height picture; returns 1 when
 * the picture is resampled, 0 when it already fits (out = in). */
int bnn_mi355x_thumbnail_size(int width, int height, int *out_w, int *out_h);
int bnn_mi355x_images_to_cifar(const uint8_t *const *pixels, const int *widths, const int *heights, const int *bands,
                               const long *row_strides, int n_images, uint8_t *records);

/* Test hook: run the stages 0..stage on n host images (n <= 32768) and copy that stage's output,
 * exactly as it sits in HBM (bit-packed activation layout), to dst.  CNV: stage L =
 * layer L (0..7, after the max-pool where there is one); LFC: stage 0 = binarised input, stage
 * L+1 = layer L (L <= 2).  Returns the bytes per image, or -1. */
long bnn_mi355x_debug_stage_output(const uint8_t *images, int n_images, int stage, void *dst, size_t cap);

/* Per-stage device timing with HIP events on the stream the kernels run on
 * (used by bench.py for the roofline line).  profile(1) makes every later
 * inference call bracket each stage with events; profile_read waits for them,
 * writes the SUM of each stage's milliseconds over all chunks enqueued since
 * the last read (n_chunks of them) and returns the number of stages, or -1. */
int bnn_mi355x_profile(int enable);
int bnn_mi355x_profile_read(float *ms_per_stage, int cap, int *n_chunks);
const char *bnn_mi355x_stage_name(int stage);

#ifdef __cplusplus
}
#endif
#endif
