/*
 * ia_oracle.h -- CPU restatement of imageAnalysisLib's per-pixel path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing that ships (the HIP library, the C++ API
 * layer, the pipeline driver) may link or call this file; only tests/, the
 * smoke check and the cpu_baseline leg of bench.py do, and only as the checker.
 *
 * PARITY UNPINNED for everything that lives in a CUDA kernel or a CUDA-calling
 * wrapper: the reference holds no golden vectors, known-answer tests or
 * fixtures, and its kernels cannot be built here (CUDA, cuRAND, Thrust, OpenCV,
 * jsoncpp are absent).  REFERENCE-PINNED are the functions the reference
 * keeps as plain host code, which g++ compiles unmodified from its sources
 * (oracle/ref_host/): iao_eq_map, the rand()-based centroid draws,
 * iao_centroids_changed, iao_array_sum, the four histogram features and the
 * cross-validation metrics are checked against vectors made by running that
 * code (tests/golden/ref_host.npz, tests/test_ref_host_golden.py).
 * Every function below restates the reference source
 * text (cited per function as imageAnalysis.cu:line) as one scalar "logical
 * thread per pixel" loop.  Where the reference is racy, reads out of bounds or
 * relies on undefined casts, the resolution used here is stated at the
 * function.  The known-answer vectors under tests/golden/ were produced by
 * this file and hand-checked against the source text.  Two third-party
 * pieces have a second source in this image and are checked against it by
 * the tests: the XORWOW generator (jump, step, output) against rocRAND's
 * device generator, and rand() against the C library.
 *
 * Float rules: compile with -ffp-contract=off; every float expression keeps
 * the reference's operand types and evaluation order.
 */
#ifndef IA_ORACLE_H
#define IA_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- contraction mode ---------------------------------------------------- */
/* on != 0: the luminance (imageAnalysis.cu:22) and the filter accumulation (:521) are evaluated
 * with fused multiply-adds, as nvcc's default -fmad=true would compile them; 0 (default): every operation
 * rounds on its own.  Process-wide; not for use while other threads run oracle functions. */
void iao_set_contraction(int on);
int iao_get_contraction(void);

/* ---- colour ------------------------------------------------------------ */
/* imageAnalysis.cu:17-37,43-70.  rgb is planar R|G|B, 3*W*H bytes.
 * method 0: (uchar)((.21*R)+(.72*G)+(.07*B)) in double; 1: (R+G+B)/3.
 * Returns 0, or -1 for an unknown method (output untouched). */
int iao_gray(const uint8_t *rgb, uint8_t *gray, int W, int H, int method);

/* ---- histogram / equalisation / LUT family ------------------------------ */
/* imageAnalysis.cu:130-152.  256 bins, value 0 is never counted. */
void iao_hist(const uint8_t *img, long N, int *hist);
/* imageAnalysis.cu:155-176.  Sequential float32 pdf/cdf. */
void iao_eq_map(const int *hist, int *map, int numPixels);
/* imageAnalysis.cu:178-186.  out = (uchar)map[in]. */
void iao_apply_map(const uint8_t *in, uint8_t *out, long N, const int *map);
/* imageAnalysis.cu:764-778,814-828.  levels = numLevels x {min,max,val}.
 * numLevels == 0 leaves the reference's output unwritten; here: identity. */
void iao_quantize(const uint8_t *in, uint8_t *out, long N, const int *levels, int numLevels);
/* imageAnalysis.cu:207-226. */
void iao_threshold(const uint8_t *in, uint8_t *out, long N, int threshold);
/* imageAnalysis.cu:237-279.  Returns the threshold; out may be NULL. */
int iao_otsu(const uint8_t *in, uint8_t *out, int W, int H);
/* imageAnalysis.cu:106-128.  Thrust's float tree sums are order-unspecified;
 * here: exact integer pixel sum, mean=(float)sum/(float)N,
 * var=(float)(sum_v cnt[v]*((double)((float)v-mean))^2)/(float)(N-1). */
void iao_mean_std(const uint8_t *in, long N, float *mean, float *stddev);
/* Same statistics from a full 256-bin histogram (bin 0 included). */
void iao_mean_std_from_hist(const long long *cnt, long N, float *mean, float *stddev);
/* The same two with the pixel sum handed in, so that tests can ask what another value of the
 * reference's order-unspecified float sum (imageAnalysis.cu:111-115,250-254) would change. */
void iao_mean_std_from_hist_sum(const long long *cnt, long N, long long sum, float *mean, float *stddev);
int iao_otsu_from_hist(const int *hist, long long sum, int numPixels);

/* ---- random numbers ------------------------------------------------------ */
/* XORWOW as initialised by curand_init(seed, subsequence=tid, offset=0)
 * (imageAnalysis.cu:709-715): first two 32-bit outputs of subsequence tid.
 * The states are never written back (imageAnalysis.cu:723,836), so these two
 * words are all the noise kernels ever see. */
void iao_xorwow_pair(unsigned long long seed, unsigned int tid, uint32_t *x0, uint32_t *x1);
void iao_xorwow_planes(unsigned long long seed, long N, uint32_t *x0, uint32_t *x1);
/* Plain generator access for self-checks: state init + n sequential outputs. */
void iao_xorwow_stream(unsigned long long seed, unsigned int subsequence, int n, uint32_t *out);
/* The 160x160 GF(2) matrix of 2^log2steps generator steps, as 160 rows of 5
 * words (row i*32+j = image of bit j of state word i). */
void iao_xorwow_jump_matrix(int log2steps, uint32_t *m800);
/* curand_uniform's mapping of a 32-bit output to (0,1]. */
float iao_uniform(uint32_t x);

/* imageAnalysis.cu:717-737,753-762. */
void iao_salt_pepper(const uint8_t *in, uint8_t *out, long N,
                     const uint32_t *x0, const uint32_t *x1, int level);
/* imageAnalysis.cu:831-863.  mean == -1 or stddev == -1 => both from the image.
 * The 256-entry amplitude table is exposed so its construction can be checked
 * on its own. use_libm != 0 swaps the portable exp for libm's expf. */
void iao_gauss_table(float mean, float variance, uint8_t *table256, int use_libm);
void iao_gauss_noise(const uint8_t *in, uint8_t *out, long N,
                     const uint32_t *x0, float mean, float stddev);

/* ---- stencils ---------------------------------------------------------------
 * Border rule shared by all (imageAnalysis.cu:389,429,476,514,649): with
 * a=(kH-1)/2, s=(kW-1)/2 a pixel is border (output 0) iff
 * row<a || row>H-a || col<s || col>W-s.  Taps are read at flat index i*W+j, so
 * column W wraps to the next row; flat indices >= W*H (the reference reads past
 * its allocation there) read as 0. */
/* imageAnalysis.cu:507-547. */
void iao_linear(const uint8_t *in, uint8_t *out, int W, int H, const float *kernel, int kW, int kH);
/* imageAnalysis.cu:469-505. */
void iao_average(const uint8_t *in, uint8_t *out, int W, int H, const float *kernel, int kW, int kH);
/* imageAnalysis.cu:641-706.  Race-free private window, weights replicate, result
 * sorted[kernSum/2].  Returns -1 (nothing written) for even sizes or negative
 * weights, where the reference reads uninitialised scratch. */
int iao_median(const uint8_t *in, uint8_t *out, int W, int H, const int *weights, int kW, int kH);
/* imageAnalysis.cu:549-590. */
void iao_sobel(const uint8_t *in, uint8_t *out, int W, int H);
/* imageAnalysis.cu:592-639.  k restarts per mask; sum carries across masks. */
void iao_compass(const uint8_t *in, uint8_t *out, int W, int H);
/* imageAnalysis.cu:381-419.  Racy scatter; resolution "255 wins". */
void iao_dilate(const uint8_t *in, uint8_t *out, int W, int H, const int *se, int kW, int kH);
/* Same kernel executed as one sequential loop in ascending tid (another legal
 * outcome of the race); used for the property seq <= maxwins. */
void iao_dilate_seq(const uint8_t *in, uint8_t *out, int W, int H, const int *se, int kW, int kH);
/* imageAnalysis.cu:421-467. */
void iao_erode(const uint8_t *in, uint8_t *out, int W, int H, const int *se, int kW, int kH);

/* ---- segmentation / features (callers of the path) ------------------------- */
/* imageAnalysis.cu:281-379 with explicit initial centroids (the reference draws
 * them from unseeded rand()).  Class sums/counts start at zero (the reference
 * never zeroes them).  Returns iterations run. */
int iao_kmeans2(const uint8_t *in, uint8_t *out, long N, int c0, int c1);
/* imageMachineLearning.cu:341-422: mean,std,area,perimeter,mode,rangeMin,rangeMax */
/* The reference's small host-only functions, kept as functions of their own so that the
 * vectors made by running the reference's own code (tests/golden/ref_host.npz) pin them:
 * centroidsHaveChanged (imageAnalysis.cu:281-293), random (:39-41) on one rand() value, arraySum
 * (:682-688), calcMode / calculateCellArea / calculateCellPerimeter / calcMinAndMaxOfRange
 * (imageMachineLearning.cu:341-391), and the confusion-matrix metrics (:214-255; 6 classes,
 * rows whose class falls outside are left out instead of written out of bounds). */
int iao_centroids_changed(uint8_t *cent, int k);
int iao_random_from_rand(int lo, int hi, int rand_value);
int iao_array_sum(const int *a, int n);
int iao_hist_mode(const int *hist);
int iao_hist_area(const int *hist, int numPixels);
int iao_hist_perimeter(const int *hist);
void iao_hist_range(const int *hist, int *rmin, int *rmax);
void iao_confusion_metrics(const int *pred, const float *rows, int numAttributes, int n, int *confusion36,
                           float *accuracy, float *precision, float *recall);
void iao_features(const uint8_t *in, int W, int H, float *feat7);

/* ---- whole pipelines, for the CPU baseline and executor parity ------------- */
enum {
    IAO_OP_EQUALIZE = 1, IAO_OP_GAUSS = 2, IAO_OP_SALT_PEPPER = 3, IAO_OP_LINEAR = 4,
    IAO_OP_AVERAGE = 5, IAO_OP_MEDIAN = 6, IAO_OP_SOBEL = 7, IAO_OP_COMPASS = 8,
    IAO_OP_DILATE = 9, IAO_OP_ERODE = 10, IAO_OP_THRESHOLD = 11, IAO_OP_OTSU = 12,
    IAO_OP_QUANTIZE = 13, IAO_OP_KMEANS = 14, IAO_OP_FEATURES = 15
};
typedef struct {
    int type;
    int kW, kH;           /* stencils */
    const float *fkernel; /* linear / average */
    const int *ikernel;   /* median weights, structuring element, quantisation levels */
    int iparam;           /* S&P level, threshold, numLevels */
    float mean, stddev;   /* gaussian */
} iao_op;
/* rgb: B planar images; out: B gray images; hist_sum: 256 x int64 dataset
 * histogram of the grayscale images (bin 0 empty, as iao_hist).  x0/x1 planes
 * may be NULL when no noise op is present.  threads<=1: serial. */
void iao_pipeline(const uint8_t *rgb, uint8_t *out, int W, int H, int B,
                  const iao_op *ops, int nops, const uint32_t *x0, const uint32_t *x1,
                  long long *hist_sum, int threads);
/* As iao_pipeline, with the two operations that need more than the image:
 * IAO_OP_KMEANS (test/test.cu:519-532) takes its two initial centroids from
 * centroids[(b * nk + k) * 2 + {0,1}], the random(0,256) draws (imageAnalysis.cu:39-41,
 * 359-361) of image b's k-th k-means operation, nk = number of such operations;
 * IAO_OP_FEATURES (test/test.cu:533-555) leaves the image alone and writes
 * features[b * 7 ..] (imageMachineLearning.cu:393-415).
 * input: what the driver makes of the RGB image first (test/test.cu:276-322): 0 luminance
 * grayscale, 1 (R+G+B)/3, 2 / 3 / 4 the red / green / blue plane. */
void iao_pipeline_ex(const uint8_t *rgb, uint8_t *out, int W, int H, int B,
                     const iao_op *ops, int nops, const uint32_t *x0, const uint32_t *x1,
                     long long *hist_sum, int threads, const int *centroids, float *features,
                     int input);

#ifdef __cplusplus
}
#endif
/* k-nearest neighbours (imageMachineLearning.cu:29-339); see the notes at the definitions:
 * parity with the reference is unpinned here more than anywhere (its kernels race). */
void iao_knn_predict(const float *train, const float *test, int numTrain, int numTest,
                     int numAttributes, int k, int *predictions);
float iao_knn_crossval(const float *dataset, int numInstances, int numAttributes, int k,
                       float *precision, float *recall, float *fold_accuracy10);

#endif
