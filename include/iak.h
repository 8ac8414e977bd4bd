/*
 * iak.h -- C ABI of the MI355X (gfx950) image-analysis kernel library, libiak.so.
 *
 * This is the boundary a host binds instead of the reference's CUDA wrappers in
 * imageAnalysisLib/imageAnalysis.cu.  Every entry point is a launcher: it enqueues
 * work on the given HIP stream (void* = hipStream_t, NULL = default stream) and
 * returns without synchronising.  Unless stated otherwise an entry point allocates
 * nothing and frees nothing; all image pointers are DEVICE pointers.
 *
 * Batches: an image batch is B images of W*H bytes (row-major, no row padding, as
 * the reference's Image struct, imageAnalysis.h:19-23), image b starting at
 * base + b*stride bytes.  RGB batches are planar R|G|B per image
 * (imageAnalysis.h:11-17, test/test.cu:53-59).
 *
 * Return value: 0 on success; a positive value is a hipError_t; negative values
 * are the IAK_E_* argument errors below.
 *
 * Float semantics follow the reference source text evaluated without
 * multiply-add contraction; undefined float->uchar casts are defined as
 * truncate-to-int then modulo 256.
 */
#ifndef IAK_H
#define IAK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IAK_E_ARG (-1)         /* null pointer, non-positive size, bad enum */
#define IAK_E_UNSUPPORTED (-2) /* well-formed but outside what the kernels implement */
#define IAK_E_TOO_LARGE (-3)   /* exceeds a fixed capacity (kernel taps, scratch) */

#define IAK_MAX_TAPS 625       /* kW*kH limit for kernels passed by value (25x25) */

int iak_version(void);
/* Static string for a return code of this library (never NULL). */
const char *iak_error_string(int code);

/* Contraction mode (process-wide, default off).  The reference has two expressions of the form a * b + c in
 * floating point whose product is not exact: the luminance (imageAnalysis.cu:22, double) and the filter
 * accumulation `sum += pixel * weight` (:484, :521, float); cuRAND's uniform, x * 2^-32 + 2^-33, multiplies by a
 * power of two and comes out the same either way.  Built by nvcc with its default
 * -fmad=true each is one fused multiply-add with a single rounding; built with -fmad=false each operation
 * rounds on its own.  Off reproduces the second build, on the first (fused the way LLVM's combiner fuses
 * them: fma(.07, B, fma(.21, R, .72 * G)); fma(pixel, weight, sum)).  The grayscale conversion differs in about
 * 0.17 % of all (R, G, B) triples, always by one grey level; filters differ only for weights whose products are
 * not exact in float (none of the reference's configurations).  Nothing the reference ships pins either build.
 * Read by iak_gray*, iak_linear at every call and by iak_pipeline_create once (a
 * pipeline keeps the setting it was created under).  In contraction mode linear / average filters larger than
 * 3x3 take the one-pixel-per-lane kernel. */
int iak_set_contraction(int enabled);
int iak_get_contraction(void);

/* ---- memory / stream helpers, so a host never has to name hip* itself ----------
 * Replace cudaMalloc/cudaFree/cudaMemcpy in imageAnalysis.cu:866-907. These DO
 * allocate / free / block as their names say. */
int iak_malloc(void **dev_ptr, size_t bytes);
int iak_free(void *dev_ptr);
int iak_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes, void *stream);
int iak_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes, void *stream);
int iak_memset(void *dst_dev, int value, size_t bytes, void *stream);
int iak_stream_sync(void *stream);
int iak_device_count(int *count);
int iak_set_device(int device);

/* ---- colour ----------------------------------------------------------------------
 * Replaces convertRGBToGrayscale + kernels (imageAnalysis.cu:17-37,43-70).
 * method 0: (uchar)((.21*R)+(.72*G)+(.07*B)) in double; 1: (R+G+B)/3.
 * Any other method returns IAK_E_ARG (the reference prints and leaves the image unset). */
int iak_gray(const uint8_t *rgb, uint8_t *gray, int W, int H, int B,
             long long rgb_stride, long long gray_stride, int method, void *stream);
/* Same pass, also producing the 256-bin histogram of every grayscale image
 * (hist[b*256+v], fully overwritten).  Fuses convertRGBToGrayscale with
 * calculateHistogram (imageAnalysis.cu:130-152): bin 0 is always 0. */
int iak_gray_hist(const uint8_t *rgb, uint8_t *gray, uint32_t *hist, int W, int H, int B,
                  long long rgb_stride, long long gray_stride, int method, void *stream);

/* One colour plane of every image copied out, with its histogram (hist may be NULL): the batched
 * form of extractSingleColorChannel (imageAnalysis.cu:72-92; channel 0 = R, 1 = G, 2 = B, planar
 * layout) followed by copyHostImageToDevice and calculateHistogram (test/test.cu:296-327). */
int iak_channel_hist(const uint8_t *rgb, uint8_t *out, uint32_t *hist, int W, int H, int B,
                     long long rgb_stride, long long out_stride, int channel, void *stream);

/* ---- histogram / equalisation / look-up family -------------------------------------
 * Replaces calculateHistogram + calcHistogram (imageAnalysis.cu:130-152): value 0 is
 * never counted.  hist[B][256] is fully overwritten. */
int iak_hist(const uint8_t *img, uint32_t *hist, long long N, int B, long long stride, void *stream);
/* Replaces equalizeHistogram (imageAnalysis.cu:155-176), on the device, one image per
 * lane: the float32 pdf/cdf is accumulated sequentially in bin order. */
int iak_eq_map(const uint32_t *hist, int32_t *map, int numPixels, int B, void *stream);
/* Replaces equalizeImageWithHist + equalizeImage (imageAnalysis.cu:178-201):
 * out = (uchar)map[in].  map is a DEVICE array [B][256]; map_stride is in elements
 * (256 for one map per image, 0 to share one map). */
int iak_apply_map(const uint8_t *in, uint8_t *out, long long N, int B, long long in_stride,
                  long long out_stride, const int32_t *map, long long map_stride, void *stream);
/* Replaces imageQuantization (imageAnalysis.cu:764-778,814-828).  levels is a HOST array
 * of numLevels x {min,max,val}; numLevels == 0 copies the image. */
int iak_quantize(const uint8_t *in, uint8_t *out, long long N, int B, long long in_stride,
                 long long out_stride, const int *levels, int numLevels, void *stream);
/* Replaces thresholdImage (imageAnalysis.cu:207-226): in > threshold ? 255 : 0. */
int iak_threshold(const uint8_t *in, uint8_t *out, long long N, int B, long long in_stride,
                  long long out_stride, int threshold, void *stream);
/* Same with one threshold per image, read from the DEVICE array thr[B]. */
int iak_threshold_per_image(const uint8_t *in, uint8_t *out, long long N, int B, long long in_stride,
                            long long out_stride, const int32_t *thr, void *stream);
/* The host loop of otsuThresholdImage (imageAnalysis.cu:255-277) on the device, from
 * histograms with bin 0 empty (as iak_hist gives them): thr[b] for every image. */
int iak_otsu_threshold(const uint32_t *hist, int32_t *thr, int numPixels, int B, void *stream);
/* calculateMeanAndStdDev (imageAnalysis.cu:106-128) from histograms with bin 0 empty:
 * mean_std[b*2] = mean, mean_std[b*2+1] = standard deviation.  The pixel sum is
 * exact (the reference's float tree sum has no defined order). */
int iak_mean_std(const uint32_t *hist, float *mean_std, int numPixels, int B, void *stream);

/* The per-pixel part of calcMSQE (imageAnalysis.cu:780-812): for every image
 * sums[b] = sum_t (uchar)round((in-out)^2 * (hist[in]/255.0)), with hist the histogram of
 * `in` as iak_hist gives it (DEVICE [B][256]).  The reference then adds the bytes with
 * plus<unsigned char>, i.e. modulo 256 -- the caller applies that.  (The reference reads its
 * histogram from a buffer it never fills; the intended histogram is used here.) */
int iak_msqe(const uint8_t *in, const uint8_t *out, const uint32_t *hist, uint32_t *sums, long long N, int B,
             long long in_stride, long long out_stride, void *stream);

/* ---- random numbers --------------------------------------------------------------------
 * Replaces setupRandomness + setup_kernel (imageAnalysis.cu:709-715,740-746).
 * curand_init(seed, tid, 0) followed by two curand() draws, for tid in [0,N): the only
 * two outputs the noise kernels ever consume, because they never store their state
 * back (imageAnalysis.cu:723,836).  x0/x1: DEVICE arrays of N words. */
int iak_rng_planes(uint32_t *x0, uint32_t *x1, long long N, unsigned long long seed, void *stream);
/* The same planes from an explicit start state (HOST array: x[0..4], then the Weyl counter d)
 * instead of a seed: subsequence tid = 2^67 * tid steps from that state, two draws.
 * iak_rng_planes is this with cuRAND's seed scrambling in front.  rocRAND's XORWOW is the same
 * generator with other scrambling constants, so through this entry the jump, the step and the
 * output function are checked against rocRAND bit for bit (tests/cpp/rocrand_check.hip). */
int iak_rng_planes_from_state(uint32_t *x0, uint32_t *x1, long long N, const uint32_t *state6, void *stream);
/* rv[t] = (int)(uniform(x0[t]) * 255.999999), the draw of imageAnalysis.cu:837. */
int iak_rv_plane(const uint32_t *x0, uint8_t *rv, long long N, void *stream);
/* sp[t] = 0 keep, 1 pepper (0), 2 salt (255) for a given level (imageAnalysis.cu:724-731). */
int iak_sp_plane(const uint32_t *x0, const uint32_t *x1, uint8_t *sp, long long N, int level, void *stream);

/* ---- noise ---------------------------------------------------------------------------
 * Replaces saltAndPepperNoise + generateSaltAndPepper (imageAnalysis.cu:717-737,753-762)
 * given the plane of iak_sp_plane (shared by all images of the batch, as the
 * reference's per-tid states are). */
int iak_salt_pepper(const uint8_t *in, uint8_t *out, long long N, int B, long long in_stride,
                    long long out_stride, const uint8_t *sp, void *stream);
/* The per-image amplitude table of addGaussianNoiseToImage (imageAnalysis.cu:841):
 * table[b*256+rv] = round(255*exp(-(rv-mean)^2/(2*std^2))) in the low byte, 0x100 set
 * where that value is NaN.  If mean == -1 or stddev == -1 both come from image b's
 * histogram (imageAnalysis.cu:858-860); hist may be NULL otherwise. */
int iak_gauss_table(const uint32_t *hist, int numPixels, int B, float mean, float stddev,
                    uint16_t *table, void *stream);
/* Replaces addGaussianNoise's kernel (imageAnalysis.cu:831-847): out = (uchar)(in + table[rv[t]]).
 * table_stride in elements (256 per image, 0 shared). */
int iak_gauss_noise(const uint8_t *in, uint8_t *out, long long N, int B, long long in_stride,
                    long long out_stride, const uint8_t *rv, const uint16_t *table,
                    long long table_stride, void *stream);

/* ---- stencils --------------------------------------------------------------------------
 * All share the reference's border rule (imageAnalysis.cu:514): with a=(kH-1)/2,
 * s=(kW-1)/2 the output is 0 where row<a || row>H-a || col<s || col>W-s; taps are read
 * at flat index i*W+j (column W is the next row's first pixel) and read 0 at or past
 * W*H.  kernel / weights / se are HOST arrays of kW*kH values (<= IAK_MAX_TAPS). */
#define IAK_LINEAR_CLAMP 0   /* linearFilter  (imageAnalysis.cu:507-547) */
#define IAK_LINEAR_AVERAGE 1 /* averageFilter (imageAnalysis.cu:469-505) */
int iak_linear(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride,
               long long out_stride, const float *kernel, int kW, int kH, int mode, void *stream);
/* Replaces medianFilter (imageAnalysis.cu:641-706) without its global scratch: each pixel
 * keeps its window privately.  Even sizes and negative weights return IAK_E_UNSUPPORTED. */
int iak_median(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride,
               long long out_stride, const int *weights, int kW, int kH, void *stream);
/* Replaces sobelFilter (imageAnalysis.cu:549-590): both clamped 3x3 responses and the
 * magnitude in one pass; setupEdgeDetection's device masks are not needed. */
int iak_sobel(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride,
              long long out_stride, void *stream);
/* Replaces compassFilter (imageAnalysis.cu:592-639). */
int iak_compass(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride,
                long long out_stride, void *stream);
/* Replaces imageDilation (imageAnalysis.cu:381-419) as a gather: out = 255 where any
 * interior foreground pixel's window covers the pixel through a non-zero element. */
int iak_dilate(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride,
               long long out_stride, const int *se, int kW, int kH, void *stream);
/* Replaces imageErosion (imageAnalysis.cu:421-467). */
int iak_erode(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride,
              long long out_stride, const int *se, int kW, int kH, void *stream);


/* ---- features and k-means thresholding for batches ------------------------------------------- */
#define IAK_N_FEATURES 7
/* featureExtraction (imageMachineLearning.cu:393-415) for B images: features[b] = {mean, stddev
 * (calculateMeanAndStdDev, imageAnalysis.cu:106-128), area = min(#mask, N - #mask) of the Otsu
 * mask (:353-360), perimeter = number of 255s in the compass filter of that mask (:362-366),
 * mode (:341-351), rangeMin, rangeMax (:379-390; -1 for an all-zero image)}.  DEVICE float[B][7].
 * hist: the images' histograms [B][256] (bin 0 empty) if the caller has them, else NULL.
 * workspace: DEVICE memory of iak_features_workspace_bytes(W, H, B) bytes, 256-byte aligned.
 * W % 16 == 0 reads each image once and stores no image; it needs 4-byte aligned images and
 * stride (IAK_E_UNSUPPORTED otherwise). */
long long iak_features_workspace_bytes(int W, int H, int B);
int iak_features(const uint8_t *img, int W, int H, int B, long long stride, const uint32_t *hist,
                 float *features, void *workspace, void *stream);
/* iak_features without a workspace argument and without any allocation.  It has no counterpart in
 * the reference: it is for callers that cannot allocate device memory or carry a workspace --
 * code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same
 * results as iak_features.  Its scratch (per image: a 256-bin histogram, an edge counter, a
 * threshold; once: the edge count's look-up tables; about 1 MiB in all) is a fixed array inside the library, one copy per GPU, held until the
 * process ends.  Limits: B <= iak_features_static_max_images() (1024; more returns
 * IAK_E_TOO_LARGE), W % 16 == 0 and 4-byte aligned images and stride (IAK_E_UNSUPPORTED otherwise:
 * the other shapes need image-sized scratch).  Every call on a GPU uses that GPU's one array, so
 * calls that overlap on one GPU (different streams, different host threads) race; calls on one
 * stream, or on streams ordered after one another, are fine.  Run it with the device current
 * whose memory img / features live in.  The first call on a device looks up the array's address and
 * builds host-side tables, which are not stream operations: make one ordinary (eager) call on each
 * device before a call under graph capture. */
int iak_features_static_max_images(void);
int iak_features_static(const uint8_t *img, int W, int H, int B, long long stride, const uint32_t *hist,
                        float *features, void *stream);
/* The dataset-level sums of one batch, packed into one array so that ONE collective (or one host
 * merge) carries them: packed[0..256) = hist_sum (the dataset histogram iak_pipeline_run
 * accumulates, test/test.cu:407-411; zeros if NULL), then C per-class image counts, C x 7
 * per-class feature sums, C x 7 sums of squares, as doubles (DEVICE double[256 + 15 * C], fully
 * overwritten).  features: DEVICE float[B][7] as iak_features / IAK_OP_FEATURES write them;
 * labels: DEVICE int32[B], class of image b in [0, C) (test/test.cu:541-550 keeps the same rows
 * and labels on the host); other labels are not counted.  Every sum is formed in a fixed order:
 * results do not depend on scheduling, and every integer stays exact (< 2^53). */
int iak_dataset_sums(const unsigned long long *hist_sum, const float *features, const int32_t *labels, int B,
                     int C, double *packed, void *stream);
/* kMeansThresholding (imageAnalysis.cu:281-379; k = 2) on histograms: map[b][v] = 0 or 255, the
 * value applyLabels (:339-349) gives a pixel of value v; apply it with iak_apply_map.
 * init_centroids: DEVICE int32[B][2], the two random(0,256) draws of each image (:359-361).
 * iterations: DEVICE int32[B] or NULL. */
int iak_kmeans2_map(const uint32_t *hist, int numPixels, int B, const int32_t *init_centroids,
                    int32_t *map, int32_t *iterations, void *stream);
/* The reference draws those centroids from libc rand() and never seeds it, so they are a fixed
 * sequence.  state: DEVICE uint32[35].  _seed writes the state srand(seed) leaves (1 = the
 * unseeded program) and synchronises; _centroids draws the next images * ops_per_image * 2
 * values of random(0,256) = rand() / (RAND_MAX / 257 + 1) in the reference's order (image by
 * image, within an image one pair per k-means operation), stores them operation-major,
 * out[(op * images + image) * 2 + j] (DEVICE int32), and advances the state. */
#define IAK_GLIBC_RAND_WORDS 35
int iak_glibc_rand_seed(uint32_t *state, unsigned seed, void *stream);
int iak_glibc_rand_centroids(uint32_t *state, int images, int ops_per_image, int32_t *out, void *stream);

/* k-nearest-neighbour prediction, replaces knn (imageMachineLearning.cu:29-197): rows of
 * numAttributes floats, the last one the class (an integer in [0, 32)); DEVICE pointers.
 * Distance = sqrt of the float sum of squared differences over the first numAttributes-1
 * values; the k nearest are taken in (distance, index) order (the reference's tournament
 * races); the vote is the reference's, whose tie test always fires and retries with k-1, so
 * the prediction is the nearest row's class.  k is cut to numTrain; k <= 64. */
int iak_knn_predict(const float *train, const float *test, int numTrain, int numTest, int numAttributes,
                    int k, int32_t *predictions, void *stream);

/* ---- batched pipeline executor ------------------------------------------------------------
 * Replaces the per-image / per-operation loop of executeOperations (test/test.cu:266-564)
 * for a whole batch that is resident on the device: grayscale (luminance) + histogram,
 * then the configured operations in order, with no per-operation allocation, copy or
 * synchronisation.  Results are identical to calling the entry points above one after
 * the other on every image. */
enum {
    IAK_OP_EQUALIZE = 1,    /* test.cu:404-419  histogram-equalization */
    IAK_OP_GAUSS = 2,       /* test.cu:375-389  gaussian-noise: mean, stddev (-1: from image) */
    IAK_OP_SALT_PEPPER = 3, /* test.cu:390-403  salt-and-pepper: iparam = intensity */
    IAK_OP_LINEAR = 4,      /* test.cu:330-344  linear-filter: fkernel, kW, kH */
    IAK_OP_AVERAGE = 5,     /* test.cu:345-359  average-filter */
    IAK_OP_MEDIAN = 6,      /* test.cu:360-374  median-filter: ikernel = weights */
    IAK_OP_SOBEL = 7,       /* test.cu:444-456 */
    IAK_OP_COMPASS = 8,     /* test.cu:457-469 */
    IAK_OP_DILATE = 9,      /* test.cu:470-484  ikernel = structuring element */
    IAK_OP_ERODE = 10,      /* test.cu:485-499 */
    IAK_OP_THRESHOLD = 11,  /* test.cu:500-509  iparam = threshold */
    IAK_OP_OTSU = 12,       /* test.cu:510-518 */
    IAK_OP_QUANTIZE = 13,   /* test.cu:420-443  ikernel = levels {min,max,val}, iparam = numLevels */
    IAK_OP_KMEANS = 14,     /* test.cu:519-532  kMeans-thresh (k = 2; centroids from the unseeded rand() sequence,
                             * two draws per image in image order, continuing from run to run) */
    IAK_OP_FEATURES = 15    /* test.cu:533-555  feature-extract: leaves the image as it is and writes the 7 features
                             * of every image to the array bound with iak_pipeline_bind_features */
};
typedef struct {
    int type;
    int kW, kH;
    const float *fkernel; /* HOST */
    const int *ikernel;   /* HOST */
    int iparam;
    float mean, stddev;
} iak_op;

#define IAK_PIPE_STAGEWISE 0 /* one launch (or more) per operation, every intermediate image written */
#define IAK_PIPE_FUSED 1     /* runs of per-pixel operations share one pass, and that pass also feeds the 3x3 stencil behind it */
#define IAK_PIPE_FUSED_POINTWISE 2 /* as FUSED, but every stencil stays a pass of its own */
#define IAK_PIPE_MAX_STAGES 64

typedef struct iak_pipeline iak_pipeline;
/* Plans the pipeline for batches of up to maxB images of W x H and allocates everything a
 * run needs (two image buffers, histograms, tables, the random planes for `seed`).
 * This call allocates and synchronises; iak_pipeline_run does neither.  A pipeline belongs to
 * the device that is current when it is created (run it with that device current) and to one
 * host thread at a time; runs on one pipeline share its buffers, so they must not overlap
 * (enqueue them on one stream, or on streams ordered after one another). */
int iak_pipeline_create(iak_pipeline **p, int W, int H, int maxB, const iak_op *ops, int nops,
                        unsigned long long seed, int mode);
int iak_pipeline_destroy(iak_pipeline *p);
/* What the first pass makes of the RGB input (the driver's extract_channel, test/test.cu:276-322):
 * luminance grayscale (default), the (R+G+B)/3 grayscale, or one colour plane. */
#define IAK_INPUT_LUMINANCE 0
#define IAK_INPUT_AVERAGE 1
#define IAK_INPUT_RED 2
#define IAK_INPUT_GREEN 3
#define IAK_INPUT_BLUE 4
int iak_pipeline_set_input(iak_pipeline *p, int input);
/* rgb: B planar RGB images (stride 3*W*H); out: B result images (stride W*H).
 * hist_sum: DEVICE array of 256 uint64, or NULL; the batch's grayscale-image histogram
 * (bin 0 empty) is ADDED to it -- the dataset histogram a multi-GPU job all-reduces.
 * gray_hist: DEVICE [B][256] uint32 per-image grayscale histograms, or NULL.
 * rgb and out may have any alignment.  The fused passes work on 16-byte aligned images; when
 * `out` is not (W % 16 == 0 plans), the last pass writes the executor's own buffer and one
 * device-to-device copy to `out` follows -- allocations of hipMalloc / torch are aligned and
 * never pay it.
 * Runs of one pipeline are expected on one stream, or on streams ordered after one another (as
 * any use of its buffers requires): the executor's histogram tables are cleared eight runs'
 * worth at a time on the stream of the run that finds none clean, and a run on a different
 * stream than the previous one clears them again there.  Which tables a run uses, and whether it
 * clears them, therefore depends on the runs before it: do NOT record a run into a HIP graph and
 * replay it (the replays would add onto tables that nothing clears).  Under capture pass a gray_hist
 * array of your own, which every run clears itself, or use the per-operation entry points. */
int iak_pipeline_run(iak_pipeline *p, const uint8_t *rgb, uint8_t *out, int B,
                     unsigned long long *hist_sum, uint32_t *gray_hist, void *stream);
/* Cuts every batch into `lanes` (1..8) consecutive sub-batches that run concurrently on
 * library-owned streams forked from / joined to the caller's stream, so that bandwidth-bound
 * and issue-bound kernels of different sub-batches overlap.  Results are unchanged.  This call
 * may create streams and events; runs do not.  Default 1. */
int iak_pipeline_set_lanes(iak_pipeline *p, int lanes);
/* Where IAK_OP_FEATURES writes: DEVICE float[maxB][7], image b of a run at features + 7 * b.
 * Required before running a pipeline that has that operation. */
int iak_pipeline_bind_features(iak_pipeline *p, float *features);
/* Tells a new pipeline that `images` images of the data set come before the first one it will
 * see: the k-means centroid draws (two libc rand() values per image per k-means operation, in
 * image order, test/test.cu:266 + imageAnalysis.cu:359-361) continue from there.  This is what
 * lets a shard of a folder that starts at image i0 produce what the serial loop produces for
 * those images.  No effect on pipelines without IAK_OP_KMEANS. */
int iak_pipeline_skip_images(iak_pipeline *p, long long images, void *stream);
/* Number of images the timed sub-batch of the most recent run held (all of them with 1 lane). */
int iak_pipeline_timed_images(iak_pipeline *p);
/* Per-launch-group timing.  set_timing(p, slots) (re)creates `slots` sets of HIP events
 * (0 disables; this call allocates events, runs do not); run number r then records one event
 * between launch groups into set r % slots, counting runs from the set_timing call.
 * timings(p, slot, ...) waits for that set's last event and fills up to cap entries:
 * ms[] (the whole group), main_ms[] (the group's image kernel alone, without the small helper
 * launch some groups have: the table prologue of a per-pixel run, the dataset-histogram
 * accumulation after the grayscale pass), names[], alg_bytes_per_px[] (the stage-wise
 * compulsory traffic of the operations in the group).  ms and main_ms may both be NULL to
 * query names only.  Returns the number of groups, or a negative error. */
int iak_pipeline_set_timing(iak_pipeline *p, int slots);
/* Records events only on every `every`-th run after set_timing (default 1: all of them), so that
 * a long measurement pays for the event records on a sample of its runs only; slot numbers
 * count the timed runs. */
int iak_pipeline_set_timing_every(iak_pipeline *p, int every);
int iak_pipeline_timings(iak_pipeline *p, int slot, float *ms, float *main_ms, const char **names,
                         long long *alg_bytes_per_px, int cap);

/* Bytes per pixel each launch group moves between HBM-sized arrays as the plan executes it
 * (image read + image written; intermediates a fused group keeps on chip and the batch-shared
 * noise planes do not count) -- the denominator of a roofline fraction for the executed plan,
 * as opposed to alg_bytes_per_px above, which adds up the stage-wise traffic of the reference's
 * one-kernel-per-operation flow (test/test.cu:326-564).  Returns the number of groups. */
int iak_pipeline_group_bytes(iak_pipeline *p, long long *moved_bytes_per_px, int cap);

#ifdef __cplusplus
}
#endif
#endif
