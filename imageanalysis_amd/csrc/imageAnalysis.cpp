// The reference's imageAnalysis.h function API (imageAnalysisLib/imageAnalysis.h:46-111)
// on top of the C ABI of libiak.so.  Each function keeps the reference wrapper's contract:
// the callee allocates output->image on the device and sets width/height, the caller frees
// the previous image; failures of the runtime print and exit(1).  Single images are batches
// of one.  Nothing here computes on pixels or histograms: that is all behind iak.h; what stays
// on the host is what the reference keeps on the host (equalizeHistogram's 256-entry loop, the
// rand() draws, the cross-validation bookkeeping).
#include "../../include/imageAnalysis.h"
#include "../../include/imageMachineLearning.h"
#include "../../include/iak.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace {

void check(int rc, const char *what, const char *file, int line) {
    if (rc == 0) return;
    std::cerr << what << " returned " << iak_error_string(rc) << "(" << rc << ") at " << file << ":" << line << std::endl;
    exit(1);
}
#define IAK_CHECK(call) check((call), #call, __FILE__, __LINE__)

long long pixels(const Image *im) { return (long long)im->width * im->height; }

void alloc_like(const Image *image, Image *output) {
    output->width = image->width;
    output->height = image->height;
    IAK_CHECK(iak_malloc((void **)&output->image, (size_t)pixels(image)));
}

// Small device scratch shared by the wrappers (histogram, map, tables, thresholds), the
// counterpart of the per-call cudaMalloc/cudaFree pairs in the reference wrappers.
struct Scratch {
    uint32_t *hist = nullptr;   // 256
    int32_t *map = nullptr;     // 256
    uint16_t *table = nullptr;  // 256
    int32_t *thr = nullptr;     // 1
    int32_t *cent = nullptr;    // 2: initial k-means centroids
    float *ms = nullptr;        // 2
    uint32_t *sum = nullptr;    // 1
    float *feat = nullptr;      // 7
    void *feat_ws = nullptr;    // iak_features workspace, grown on demand
    long long feat_ws_bytes = 0;
    void ensure() {
        if (hist) return;
        IAK_CHECK(iak_malloc((void **)&hist, 256 * sizeof(uint32_t)));
        IAK_CHECK(iak_malloc((void **)&map, 256 * sizeof(int32_t)));
        IAK_CHECK(iak_malloc((void **)&table, 256 * sizeof(uint16_t)));
        IAK_CHECK(iak_malloc((void **)&thr, sizeof(int32_t)));
        IAK_CHECK(iak_malloc((void **)&cent, 2 * sizeof(int32_t)));
        IAK_CHECK(iak_malloc((void **)&ms, 2 * sizeof(float)));
        IAK_CHECK(iak_malloc((void **)&sum, sizeof(uint32_t)));
        IAK_CHECK(iak_malloc((void **)&feat, IAK_N_FEATURES * sizeof(float)));
    }
} g_scratch;

// setupRandomness state (the reference's global d_states, imageAnalysis.cu:15)
struct Randomness {
    long long n = 0;
    uint32_t *x0 = nullptr, *x1 = nullptr;
    uint8_t *rv = nullptr, *sp = nullptr;
    int sp_level = -1;
    bool sp_valid = false;
} g_rng;

void require_rng(const Image *image, const char *who) {
    if (!g_rng.x0 || g_rng.n < pixels(image)) {
        // the reference would read d_states out of bounds here; refuse instead
        std::cerr << who << ": setupRandomness() has not been called for an image this large" << std::endl;
        exit(1);
    }
}

}  // namespace

// ---- copies (imageAnalysis.cu:866-907) --------------------------------------------------------
void copyHostImageToDevice(Image *host, Image *device) {
    const size_t n = (size_t)pixels(host);
    IAK_CHECK(iak_malloc((void **)&device->image, n));
    IAK_CHECK(iak_memcpy_h2d(device->image, host->image, n, nullptr));
    device->height = host->height;
    device->width = host->width;
}

void copyDeviceImageToHost(Image *device, Image *host) {
    host->height = device->height;
    host->width = device->width;
    const size_t n = (size_t)pixels(device);
    host->image = (unsigned char *)malloc(n);
    IAK_CHECK(iak_memcpy_d2h(host->image, device->image, n, nullptr));
}

void copyDeviceRGBImageToHost(RGBImage *device, RGBImage *host) {
    // the reference copies one plane and reports one channel (imageAnalysis.cu:884-891)
    host->height = device->height;
    host->width = device->width;
    host->channels = 1;
    const size_t n = (size_t)device->width * device->height;
    host->image = (unsigned char *)malloc(n);
    IAK_CHECK(iak_memcpy_d2h(host->image, device->image, n, nullptr));
}

void copyHostRGBImageToDevice(RGBImage *host, RGBImage *device) {
    const size_t n = (size_t)host->width * host->height * host->channels;
    IAK_CHECK(iak_malloc((void **)&device->image, n));
    IAK_CHECK(iak_memcpy_h2d(device->image, host->image, n, nullptr));
    device->height = host->height;
    device->width = host->width;
    device->channels = host->channels;
}

// ---- colour (imageAnalysis.cu:43-92) ------------------------------------------------------------
void convertRGBToGrayscale(RGBImage *d_rgb, Image *d_gray, int method) {
    const long long n = (long long)d_rgb->width * d_rgb->height;
    IAK_CHECK(iak_malloc((void **)&d_gray->image, (size_t)n));
    if (method != 0 && method != 1) {
        printf("WTF why are we defaulting?\n");  // imageAnalysis.cu:66-68: dims stay unset
        return;
    }
    IAK_CHECK(iak_gray(d_rgb->image, d_gray->image, d_rgb->width, d_rgb->height, 1, 3 * n, n, method, nullptr));
    d_gray->width = d_rgb->width;
    d_gray->height = d_rgb->height;
}

void extractSingleColorChannel(RGBImage *rgb, Image *out, int color) {
    out->width = rgb->width;
    out->height = rgb->height;
    const long long n = (long long)rgb->width * rgb->height;
    switch (color) {
    case 0: out->image = rgb->image; break;
    case 1: out->image = rgb->image + n; break;
    case 2: out->image = rgb->image + 2 * n; break;
    default: printf("invalid option\n"); break;
    }
}

// ---- histogram / equalisation (imageAnalysis.cu:130-201) -----------------------------------------
void calculateHistogram(Image *image, int *h_histogram, int *d_histogram) {
    (void)d_histogram;  // dead in the reference too: the by-value pointer is overwritten (:146)
    g_scratch.ensure();
    IAK_CHECK(iak_hist(image->image, g_scratch.hist, pixels(image), 1, pixels(image), nullptr));
    IAK_CHECK(iak_memcpy_d2h(h_histogram, g_scratch.hist, 256 * sizeof(int), nullptr));
}

void equalizeHistogram(int *original, int *mappings, int numPixels) {
    // host code in the reference as well (imageAnalysis.cu:155-176): sequential float32 cdf
    float cdf = 0.0f;
    for (int i = 0; i < 256; i++) {
        const float pdf = original[i] / (float)numPixels;
        cdf = (i > 0) ? pdf + cdf : pdf;
        mappings[i] = (int)(cdf * 255);
    }
}

void equalizeImageWithHist(Image *image, Image *d_equalizedImage, int *h_mappings) {
    g_scratch.ensure();
    alloc_like(image, d_equalizedImage);
    IAK_CHECK(iak_memcpy_h2d(g_scratch.map, h_mappings, 256 * sizeof(int), nullptr));
    IAK_CHECK(iak_apply_map(image->image, d_equalizedImage->image, pixels(image), 1, pixels(image), pixels(image),
                            g_scratch.map, 0, nullptr));
}

void imageQuantization(Image *image, Image *output, int *levels, int numLevels) {
    alloc_like(image, output);
    IAK_CHECK(iak_quantize(image->image, output->image, pixels(image), 1, pixels(image), pixels(image), levels,
                           numLevels, nullptr));
}

int calcMSQE(Image *image, Image *output) {
    g_scratch.ensure();
    const long long n = pixels(image);
    IAK_CHECK(iak_hist(image->image, g_scratch.hist, n, 1, n, nullptr));
    IAK_CHECK(iak_msqe(image->image, output->image, g_scratch.hist, g_scratch.sum, n, 1, n, n, nullptr));
    uint32_t s = 0;
    IAK_CHECK(iak_memcpy_d2h(&s, g_scratch.sum, sizeof s, nullptr));
    return (int)(s & 0xffu);  // thrust::plus<unsigned char> (imageAnalysis.cu:802-806)
}

// ---- thresholds (imageAnalysis.cu:207-379) ---------------------------------------------------------
void thresholdImage(Image *image, Image *output, int threshold) {
    alloc_like(image, output);
    IAK_CHECK(iak_threshold(image->image, output->image, pixels(image), 1, pixels(image), pixels(image), threshold,
                            nullptr));
}

void otsuThresholdImage(Image *image, Image *output) {
    g_scratch.ensure();
    alloc_like(image, output);
    const long long n = pixels(image);
    IAK_CHECK(iak_hist(image->image, g_scratch.hist, n, 1, n, nullptr));
    IAK_CHECK(iak_otsu_threshold(g_scratch.hist, g_scratch.thr, (int)n, 1, nullptr));
    IAK_CHECK(iak_threshold_per_image(image->image, output->image, n, 1, n, n, g_scratch.thr, nullptr));
}

static int reference_random(int min, int max) {  // imageAnalysis.cu:39-41
    return min + rand() / (RAND_MAX / (max - min + 1) + 1);
}

void kMeansThresholding(Image *image, Image *output) {
    // imageAnalysis.cu:351-379.  A pixel's label depends on its value only, so the iterations run
    // on the 256-bin histogram (iak_kmeans2_map) and the image is touched by one histogram pass
    // and one table pass.  The two initial centroids are this process's next two rand() draws,
    // as in the reference (:359-361).
    g_scratch.ensure();
    alloc_like(image, output);
    const long long n = pixels(image);
    const int32_t cent[2] = {reference_random(0, 256), reference_random(0, 256)};
    IAK_CHECK(iak_memcpy_h2d(g_scratch.cent, cent, sizeof cent, nullptr));
    IAK_CHECK(iak_hist(image->image, g_scratch.hist, n, 1, n, nullptr));
    IAK_CHECK(iak_kmeans2_map(g_scratch.hist, (int)n, 1, g_scratch.cent, g_scratch.map, nullptr, nullptr));
    IAK_CHECK(iak_apply_map(image->image, output->image, n, 1, n, n, g_scratch.map, 256, nullptr));
}

void kMeansThresholding(Image *image, Image *output, int k) {
    // declared but never defined in the reference (imageAnalysis.h:90); k is ignored by its
    // driver as well (test/test.cu:521-525)
    (void)k;
    kMeansThresholding(image, output);
}

void calculateMeanAndStdDev(Image *image, float *mean, float *stdDev) {
    g_scratch.ensure();
    const long long n = pixels(image);
    IAK_CHECK(iak_hist(image->image, g_scratch.hist, n, 1, n, nullptr));
    IAK_CHECK(iak_mean_std(g_scratch.hist, g_scratch.ms, (int)n, 1, nullptr));
    float ms[2];
    IAK_CHECK(iak_memcpy_d2h(ms, g_scratch.ms, sizeof ms, nullptr));
    *mean = ms[0];
    *stdDev = ms[1];
}

// ---- noise (imageAnalysis.cu:709-863) -----------------------------------------------------------------
void setupRandomness(Image *image) {
    cleanupRandomness();
    const long long n = pixels(image);
    IAK_CHECK(iak_malloc((void **)&g_rng.x0, (size_t)n * sizeof(uint32_t)));
    IAK_CHECK(iak_malloc((void **)&g_rng.x1, (size_t)n * sizeof(uint32_t)));
    IAK_CHECK(iak_malloc((void **)&g_rng.rv, (size_t)n));
    IAK_CHECK(iak_malloc((void **)&g_rng.sp, (size_t)n));
    IAK_CHECK(iak_rng_planes(g_rng.x0, g_rng.x1, n, 1234ULL, nullptr));  // curand_init(1234, tid, 0)
    IAK_CHECK(iak_rv_plane(g_rng.x0, g_rng.rv, n, nullptr));
    g_rng.n = n;
    g_rng.sp_valid = false;
}

void cleanupRandomness() {
    if (g_rng.x0) {
        IAK_CHECK(iak_free(g_rng.x0));
        IAK_CHECK(iak_free(g_rng.x1));
        IAK_CHECK(iak_free(g_rng.rv));
        IAK_CHECK(iak_free(g_rng.sp));
    }
    g_rng = Randomness();
}

void saltAndPepperNoise(Image *image, Image *output, int level) {
    require_rng(image, "saltAndPepperNoise");
    alloc_like(image, output);
    if (!g_rng.sp_valid || g_rng.sp_level != level) {
        IAK_CHECK(iak_sp_plane(g_rng.x0, g_rng.x1, g_rng.sp, g_rng.n, level, nullptr));
        g_rng.sp_level = level;
        g_rng.sp_valid = true;
    }
    IAK_CHECK(iak_salt_pepper(image->image, output->image, pixels(image), 1, pixels(image), pixels(image), g_rng.sp,
                              nullptr));
}

void addGaussianNoise(Image *image, Image *output, float meanPar, float stdDevPar) {
    require_rng(image, "addGaussianNoise");
    g_scratch.ensure();
    alloc_like(image, output);
    const long long n = pixels(image);
    if (meanPar == -1 || stdDevPar == -1) IAK_CHECK(iak_hist(image->image, g_scratch.hist, n, 1, n, nullptr));
    IAK_CHECK(iak_gauss_table(g_scratch.hist, (int)n, 1, meanPar, stdDevPar, g_scratch.table, nullptr));
    IAK_CHECK(iak_gauss_noise(image->image, output->image, n, 1, n, n, g_rng.rv, g_scratch.table, 0, nullptr));
}

// ---- filters, edges, morphology (imageAnalysis.cu:381-706) ---------------------------------------------
void linearFilter(Image *image, Image *output, float *kernel, int kWidth, int kHeight) {
    alloc_like(image, output);
    IAK_CHECK(iak_linear(image->image, output->image, image->width, image->height, 1, pixels(image), pixels(image),
                         kernel, kWidth, kHeight, IAK_LINEAR_CLAMP, nullptr));
}

void averageFilter(Image *image, Image *output, float *kernel, int kWidth, int kHeight) {
    alloc_like(image, output);
    IAK_CHECK(iak_linear(image->image, output->image, image->width, image->height, 1, pixels(image), pixels(image),
                         kernel, kWidth, kHeight, IAK_LINEAR_AVERAGE, nullptr));
}

void medianFilter(Image *image, Image *output, int *kernel, int kWidth, int kHeight) {
    alloc_like(image, output);
    IAK_CHECK(iak_median(image->image, output->image, image->width, image->height, 1, pixels(image), pixels(image),
                         kernel, kWidth, kHeight, nullptr));
}

// The Sobel masks are compiled into the kernel: nothing to upload or free.
void setupEdgeDetection() {}
void cleanupEdgeDetection() {}

void sobelFilter(Image *image, Image *output) {
    alloc_like(image, output);
    IAK_CHECK(iak_sobel(image->image, output->image, image->width, image->height, 1, pixels(image), pixels(image),
                        nullptr));
}

void compassFilter(Image *image, Image *output) {
    alloc_like(image, output);
    IAK_CHECK(iak_compass(image->image, output->image, image->width, image->height, 1, pixels(image), pixels(image),
                          nullptr));
}

void imageDilation(Image *image, Image *output, int *structuringElement, int kWidth, int kHeight) {
    alloc_like(image, output);
    IAK_CHECK(iak_dilate(image->image, output->image, image->width, image->height, 1, pixels(image), pixels(image),
                         structuringElement, kWidth, kHeight, nullptr));
}

void imageErosion(Image *image, Image *output, int *structuringElement, int kWidth, int kHeight) {
    alloc_like(image, output);
    IAK_CHECK(iak_erode(image->image, output->image, image->width, image->height, 1, pixels(image), pixels(image),
                        structuringElement, kWidth, kHeight, nullptr));
}

// ---- feature extraction (imageMachineLearning.cu:341-422) -----------------------------------------------
std::vector<float> featureExtraction(Image *image, Image *tempImage, int *h_histogram, int *d_histogram) {
    // imageMachineLearning.cu:393-415.  The reference makes six image passes (statistics, Otsu
    // mask, its histogram, compass filter, its histogram, the image's histogram); here the
    // histogram is taken once and iak_features derives everything from it plus one more read
    // of the image.  Observable leftovers are kept: h_histogram ends up holding the image's
    // histogram, tempImage carries the image's size and no buffer (the reference frees it).
    (void)d_histogram;
    g_scratch.ensure();
    const long long n = pixels(image);
    const long long need = iak_features_workspace_bytes(image->width, image->height, 1);
    if (need < 0) IAK_CHECK((int)need);
    if (g_scratch.feat_ws_bytes < need) {
        if (g_scratch.feat_ws) IAK_CHECK(iak_free(g_scratch.feat_ws));
        IAK_CHECK(iak_malloc(&g_scratch.feat_ws, (size_t)need));
        g_scratch.feat_ws_bytes = need;
    }
    IAK_CHECK(iak_hist(image->image, g_scratch.hist, n, 1, n, nullptr));
    IAK_CHECK(iak_features(image->image, image->width, image->height, 1, n, g_scratch.hist, g_scratch.feat,
                           g_scratch.feat_ws, nullptr));
    float f[IAK_N_FEATURES];
    IAK_CHECK(iak_memcpy_d2h(f, g_scratch.feat, sizeof f, nullptr));
    IAK_CHECK(iak_memcpy_d2h(h_histogram, g_scratch.hist, 256 * sizeof(int), nullptr));
    tempImage->width = image->width;
    tempImage->height = image->height;
    tempImage->image = nullptr;
    return std::vector<float>(f, f + IAK_N_FEATURES);
}

std::vector<float> extractAllPixelsAsFeatures(Image *image) {
    std::vector<float> features;
    const size_t n = (size_t)pixels(image);
    std::vector<unsigned char> vals(n);
    IAK_CHECK(iak_memcpy_d2h(vals.data(), image->image, n, nullptr));
    features.reserve(n);
    for (size_t i = 0; i < n; i++) features.push_back(vals[i]);
    return features;
}


// ---- KNN (imageMachineLearning.cu:158-339) -------------------------------------------------------------
void knn(int numTrain, int numTest, float *train, float *test, int numAttributes, int *h_predictions, int k) {
    float *d_train = nullptr, *d_test = nullptr;
    int32_t *d_pred = nullptr;
    const size_t row = sizeof(float) * (size_t)numAttributes;
    IAK_CHECK(iak_malloc((void **)&d_train, row * (size_t)numTrain));
    IAK_CHECK(iak_malloc((void **)&d_test, row * (size_t)numTest));
    IAK_CHECK(iak_malloc((void **)&d_pred, sizeof(int32_t) * (size_t)numTest));
    IAK_CHECK(iak_memcpy_h2d(d_train, train, row * (size_t)numTrain, nullptr));
    IAK_CHECK(iak_memcpy_h2d(d_test, test, row * (size_t)numTest, nullptr));
    IAK_CHECK(iak_knn_predict(d_train, d_test, numTrain, numTest, numAttributes, k, d_pred, nullptr));
    IAK_CHECK(iak_memcpy_d2h(h_predictions, d_pred, sizeof(int32_t) * (size_t)numTest, nullptr));
    IAK_CHECK(iak_free(d_train));
    IAK_CHECK(iak_free(d_test));
    IAK_CHECK(iak_free(d_pred));
}

float knnTenfoldCrossVal(float *dataset, int numInstances, int numAttributes, int k, float *precision, float *recall) {
    const int NUM_CLASSES = 6;                          // imageMachineLearning.cu:212
    const int numToRotate = (int)(numInstances * .1);  // :262
    const int trainSize = numInstances - numToRotate;
    const size_t total = (size_t)numInstances * (size_t)numAttributes;
    std::vector<float> rot(total);
    std::vector<int> predictions((size_t)std::max(numToRotate, 1));
    double totalAccuracy = 0.0;
    float totalPrecision = 0, totalRecall = 0;
    for (int fold = 0; fold < 10; fold++) {
        const size_t shift = (size_t)numToRotate * (size_t)numAttributes * (size_t)fold;  // std::rotate, :264
        for (size_t i = 0; i < total; i++) rot[i] = dataset[(i + shift) % total];
        float accuracy = 0, prec = 0, rec = 0;
        if (numToRotate > 0) {
            float *test = rot.data() + (size_t)trainSize * (size_t)numAttributes;
            knn(trainSize, numToRotate, rot.data(), test, numAttributes, predictions.data(), k);
            int confusion[NUM_CLASSES * NUM_CLASSES] = {0};  // :214-227; classes outside the matrix are left out
            for (int i = 0; i < numToRotate; i++) {
                const int tc = (int)test[(size_t)i * numAttributes + numAttributes - 1], pc = predictions[i];
                if (tc >= 0 && tc < NUM_CLASSES && pc >= 0 && pc < NUM_CLASSES) confusion[tc * NUM_CLASSES + pc]++;
            }
            int ok = 0;
            for (int i = 0; i < NUM_CLASSES; i++) ok += confusion[i * NUM_CLASSES + i];
            accuracy = ok / (float)numToRotate;  // :229-236
            float tp = 0, tr = 0;
            for (int i = 0; i < NUM_CLASSES; ++i) {  // :239-255
                int rowSum = 0, colSum = 0;
                for (int j = 0; j < NUM_CLASSES; ++j) {
                    colSum += confusion[i + j * NUM_CLASSES];
                    rowSum += confusion[i * NUM_CLASSES + j];
                }
                const int t = confusion[i * NUM_CLASSES + i];
                tp += (rowSum > 0) ? t / (float)rowSum : 0;
                tr += (colSum > 0) ? t / (float)colSum : 0;
            }
            prec = tp / NUM_CLASSES;
            rec = tr / NUM_CLASSES;
        }
        printf("thread %i had accuracy of %f, precision of %f, and recall of %f\n", fold, (double)accuracy, prec, rec);
        totalAccuracy += (double)accuracy;
        totalPrecision += prec;
        totalRecall += rec;
    }
    printf("Average accuracy was %f\n", totalAccuracy / 10.0);
    *precision = totalPrecision / 10.0;
    *recall = totalRecall / 10.0;
    return totalAccuracy / 10.0;
}
