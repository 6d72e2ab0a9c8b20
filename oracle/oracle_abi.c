/*
 * oracle_abi.c -- the reference's six-symbol C ABI (bnn/bnn.py:69-77) on top
 * of the CPU restatement, one shared object per network like make-sw.sh:106-119
 * builds them ("python_sw-<network>-<platform>.so").
 *
 * TEST INFRASTRUCTURE ONLY: this is BASELINE.json configs[0] ("LFC-W1A1 MNIST
 * single-image classify via the SW-runtime .so on CPU") and the CPU side of the
 * ABI parity tests.  The product library under bnn-pynq_amd/ never loads it.
 *
 * Entry-point behaviour follows bnn/src/network/cnvW1A1/sw/main_python.cpp:
 * 67-82,120-169,225-231 and bnn/src/network/lfcW1A1/sw/main_python.cpp:
 * 65-75,113-156,210-216.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "bnn_oracle.h"

#ifndef ORACLE_NETWORK
#error "compile with -DORACLE_NETWORK=\"cnvW1A1\" (or another network name)"
#endif

static bnn_oracle *g_net;

static double now_us(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

void load_parameters(const char *path) {
  if (g_net) bnn_oracle_destroy(g_net);
  g_net = bnn_oracle_create(ORACLE_NETWORK, path);
}

static int load_images(const char *path, uint8_t **imgs) {
  if (!g_net) {
    fprintf(stderr, "oracle_abi: load_parameters was not called\n");
    return -1;
  }
  int n = bnn_oracle_is_cnv(g_net) ? bnn_oracle_parse_cifar10(path, imgs)
                                   : bnn_oracle_parse_mnist(path, imgs);
  if (n < 0) fprintf(stderr, "oracle_abi: cannot parse %s\n", path);
  return n;
}

int inference(const char *path, int results[64], int number_class, float *usecPerImage) {
  uint8_t *imgs = NULL;
  int n = load_images(path, &imgs);
  if (n <= 0) { bnn_oracle_free(imgs); return -1; }
  int cls;
  double t1 = now_us(), t2;
  if (bnn_oracle_is_cnv(g_net)) {
    /* testPrebuiltCIFAR10_from_image: count = 1, first record only */
    int16_t s[64];
    bnn_oracle_cnv_scores_ref(g_net, imgs, s);
    t2 = now_us();
    if (results)
      for (int j = 0; j < number_class; j++) results[j] = s[j];
    cls = bnn_oracle_decode_cnv_single(s, number_class);
  } else {
    uint64_t w = bnn_oracle_lfc_word_ref(g_net, imgs);
    t2 = now_us();
    cls = bnn_oracle_decode_lfc_single(w, number_class);
    if (results) {
      /* one-hot over 64 entries at round(log2(word)); word==0 sets entry 0 */
      const int hot = bnn_oracle_lfc_single_hot(w, number_class);
      for (int i = 0; i < 64; i++) results[i] = (i == hot) ? 1 : 0;
    }
  }
  if (usecPerImage) *usecPerImage = (float)(t2 - t1);
  bnn_oracle_free(imgs);
  return cls;
}

static int *run_multiple(const char *path, int number_class, int *image_number,
                         float *usecPerImage, int enable_detail) {
  uint8_t *imgs = NULL;
  int n = load_images(path, &imgs);
  if (n < 0) return NULL;
  int *result;
  double t1, t2;
  if (bnn_oracle_is_cnv(g_net)) {
    int16_t *s = (int16_t *)malloc((size_t)(n > 0 ? n : 1) * 64 * sizeof(int16_t));
    t1 = now_us();
    bnn_oracle_cnv_scores_fast(g_net, imgs, n, s, 0);
    t2 = now_us();
    if (enable_detail) {
      result = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1) * number_class);
      for (int i = 0; i < n; i++)
        for (int j = 0; j < number_class; j++) result[i * number_class + j] = s[i * 64 + j];
    } else {
      result = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
      for (int i = 0; i < n; i++)
        result[i] = bnn_oracle_decode_cnv_batched(s + (size_t)i * 64, number_class);
    }
    free(s);
  } else {
    uint64_t *w = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n > 0 ? n : 1));
    t1 = now_us();
    bnn_oracle_lfc_words_fast(g_net, imgs, n, w, 0);
    t2 = now_us();
    result = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) result[i] = bnn_oracle_decode_lfc_batched(w[i], number_class);
    free(w);
  }
  if (image_number) *image_number = n;
  if (usecPerImage) *usecPerImage = n > 0 ? (float)((t2 - t1) / n) : 0.0f;
  bnn_oracle_free(imgs);
  return result;
}

int *inference_multiple(const char *path, int number_class, int *image_number,
                        float *usecPerImage, int enable_detail) {
  return run_multiple(path, number_class, image_number, usecPerImage, enable_detail);
}

int *inference_multiple_with_faults(const char *path, int number_class, int *image_number,
                                    float *usecPerImage, unsigned int flip_count,
                                    int word_size, int target, int *target_layers,
                                    unsigned int num_targets) {
  (void)word_size; (void)target; (void)target_layers; (void)num_targets;
  if (flip_count != 0) {
    /* the reference draws fault positions from std::random_device
     * (faults.h:115-148): there is nothing deterministic to restate */
    fprintf(stderr, "oracle_abi: fault injection (flip_count=%u) is not restated\n", flip_count);
    return NULL;
  }
  /* flip_count == 0: one image per call, class indices only */
  return run_multiple(path, number_class, image_number, usecPerImage, 0);
}

void free_results(int *result) { free(result); }

void deinit(void) { /* FoldedMVDeinit frees I/O buffers only; weights stay (rawhls-offload.cpp:69-74) */ }
