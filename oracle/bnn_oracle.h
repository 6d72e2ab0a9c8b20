/*
 * bnn_oracle.h -- CPU restatement of the BNN-PYNQ SW-runtime hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under bnn-pynq_amd/ (the product) may
 * include, link or dlopen this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement
 * against the 40 class scores and the class indices recorded in the
 * reference's notebooks / tests; the fixtures are under
 * tests/golden/.
 *
 * The arithmetic of the path lives in the un-vendored submodule
 * cbrl/finn-hlslib (branch master, unpinned; .gitmodules:1-4) and in
 * xilinx-tiny-cnn (HEAD, unpinned; make-sw.sh:69-73): neither is present in
 * /root/reference, so every function below cites the reference call site
 * that fixes its behaviour.
 */
#ifndef BNN_ORACLE_H
#define BNN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bnn_oracle bnn_oracle;

/* network = "cnvW1A1" | "cnvW1A2" | "cnvW2A2" | "lfcW1A1" | "lfcW1A2".
 * param_dir = directory of L-P-weights.bin / L-P-thres.bin files.
 * Returns NULL (message on stderr) when the network is unknown or a file is
 * missing (the reference throws "Could not open file",
 * foldedmv-offload.cpp:321-323). */
bnn_oracle *bnn_oracle_create(const char *network, const char *param_dir);
void bnn_oracle_destroy(bnn_oracle *o);
int bnn_oracle_is_cnv(const bnn_oracle *o);
int bnn_oracle_num_layers(const bnn_oracle *o);

/* ---- faithful scalar path (one MAC at a time, unpacked integers) -------- */
/* img: one CIFAR-10 record body, planar CHW uint8[3*32*32]. scores: the 64
 * 16-bit outputs of layer 8 (top.cpp:232-235, read back as ap_int<16>,
 * foldedmv-offload.h:394-408). */
void bnn_oracle_cnv_scores_ref(const bnn_oracle *o, const uint8_t *img,
                               int16_t scores[64]);
/* px: 784 MNIST pixels row-major. Returns the raw 64-bit output word of the
 * last layer (lfcW1A1/hw/top.cpp:155-164), not yet masked to labelBits. */
uint64_t bnn_oracle_lfc_word_ref(const bnn_oracle *o, const uint8_t *px);
/* binarizeAndPack (foldedmv-offload.cpp:82-98): 784 pixels -> the 13 words the host ships to the accelerator */
void bnn_oracle_lfc_binarize(const uint8_t *px, uint64_t words[13]);
/* Activations after layer `layer` (after the max-pool that follows layers 1
 * and 3 of the CNV nets), value domain (+1/-1 for 1-bit activations,
 * -1/0/+1 for 2-bit), pixel-major HWC / neuron order.  For the last CNV layer
 * the 64 raw accumulators are not available here (use *_scores_ref).
 * Returns the element count written, or -1. */
int bnn_oracle_layer_ref(const bnn_oracle *o, const uint8_t *img, int layer,
                         int8_t *out, int cap);

/* ---- fast path: same results, 64-bit popcount words + OpenMP over images - */
void bnn_oracle_cnv_scores_fast(const bnn_oracle *o, const uint8_t *imgs, int n,
                                int16_t *scores /* n*64 */, int nthreads);
void bnn_oracle_lfc_words_fast(const bnn_oracle *o, const uint8_t *imgs, int n,
                               uint64_t *words /* n */, int nthreads);

/* ---- output decode (foldedmv-offload.h:347-354,394-408; .cpp:144-166,202-220) */
int bnn_oracle_decode_cnv_batched(const int16_t *scores, int number_class);
int bnn_oracle_decode_cnv_single(const int16_t *scores, int number_class);
int bnn_oracle_decode_lfc_batched(uint64_t word, int number_class);
int bnn_oracle_decode_lfc_single(uint64_t word, int number_class);
/* index of the entry set in the 64-entry one-hot results[] of the single-image
 * LFC path (64 => none) */
int bnn_oracle_lfc_single_hot(uint64_t word, int number_class);

/* ---- input conversion (foldedmv-offload.h:129-144 / tiny-cnn parse_cifar10) */
int bnn_oracle_quantise_u8(int p); /* literal float formula, returns int8 value */

/* ---- file parsers (tiny-cnn parse_cifar10 / parse_mnist_images call sites:
 * main_python.cpp:129,152; lfcW1A1/sw/main_python.cpp:122,144).
 * Return image count (>=0) and a malloc'd buffer of count*3072 (CHW) or
 * count*784 bytes in *out (caller frees with bnn_oracle_free), or -1. */
int bnn_oracle_parse_cifar10(const char *path, uint8_t **out);
int bnn_oracle_parse_mnist(const char *path, uint8_t **out);
void bnn_oracle_free(void *p);

/* ---- fault injection on one memory word (inject_fault_impl, foldedmv-offload.h:146-163):
 * target 0 = weight word (layer, mem = PE, ind), 1 = threshold (layer, mem, ind, thresh); flips
 * word_size adjacent bits at bit_pos aligned down to word_size.  Returns the matrix row changed,
 * or -1.  The fault stays until the oracle is destroyed, like the reference's memories. */
int bnn_oracle_apply_fault(bnn_oracle *o, int target, int layer, int mem, int ind, int thresh, int bit_pos,
                           int word_size);

/* ---- unpacked parameters, for tests that cross-check the GPU repacker ---- */
/* weight of (layer, row n, column j) in value domain; threshold i of row n. */
int bnn_oracle_weight(const bnn_oracle *o, int layer, int n, int j);
int bnn_oracle_threshold(const bnn_oracle *o, int layer, int n, int i);
int bnn_oracle_layer_mw(const bnn_oracle *o, int layer);
int bnn_oracle_layer_mh(const bnn_oracle *o, int layer);

#ifdef __cplusplus
}
#endif
#endif
