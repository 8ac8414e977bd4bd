// Exercises the C++ imageAnalysis.h API (libimageAnalysis.so) the way the reference's driver
// does -- host image in, per-operation device images out, caller frees -- and dumps every
// result so the Python test can compare it with the CPU oracle.
//
// usage: api_check <rgb.raw> <W> <H> <outdir>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "imageAnalysis.h"
#include "imageMachineLearning.h"

static void dump(const std::string &dir, const char *name, Image *d_img) {
    Image h;
    copyDeviceImageToHost(d_img, &h);
    std::string path = dir + "/" + name + ".u8";
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(2); }
    fwrite(h.image, 1, (size_t)h.width * h.height, f);
    fclose(f);
    free(h.image);
}

int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: api_check rgb.raw W H outdir\n"); return 2; }
    const int W = atoi(argv[2]), H = atoi(argv[3]);
    const std::string dir = argv[4];
    const size_t n = (size_t)W * H;
    std::vector<unsigned char> rgb(3 * n);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(rgb.data(), 1, 3 * n, f) != 3 * n) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(f);

    RGBImage h_rgb{rgb.data(), 3, W, H}, d_rgb;
    Image d_img, d_tmp, h_chan, d_chan;
    copyHostRGBImageToDevice(&h_rgb, &d_rgb);
    convertRGBToGrayscale(&d_rgb, &d_img, 0);
    dump(dir, "gray", &d_img);
    convertRGBToGrayscale(&d_rgb, &d_tmp, 1);
    dump(dir, "gray_avg", &d_tmp);
    CUDA_CHECK_RETURN(hipFree(d_tmp.image));
    extractSingleColorChannel(&h_rgb, &h_chan, 1);  // host-side alias, as test.cu:300-301
    copyHostImageToDevice(&h_chan, &d_chan);
    dump(dir, "green", &d_chan);
    CUDA_CHECK_RETURN(hipFree(d_chan.image));

    FILE *js = fopen((dir + "/scalars.txt").c_str(), "w");
    int hist[256], map[256];
    calculateHistogram(&d_img, hist, nullptr);
    fprintf(js, "hist");
    for (int i = 0; i < 256; i++) fprintf(js, " %d", hist[i]);
    fprintf(js, "\n");
    equalizeHistogram(hist, map, W * H);
    equalizeImageWithHist(&d_img, &d_tmp, map);
    dump(dir, "equalize", &d_tmp);
    CUDA_CHECK_RETURN(hipFree(d_tmp.image));

    float mean, sd;
    calculateMeanAndStdDev(&d_img, &mean, &sd);
    fprintf(js, "mean_std %.9g %.9g\n", mean, sd);

    int levels[6] = {10, 20, 15, 30, 250, 77};
    imageQuantization(&d_img, &d_tmp, levels, 2);
    dump(dir, "quantize", &d_tmp);
    fprintf(js, "msqe %d\n", calcMSQE(&d_img, &d_tmp));
    CUDA_CHECK_RETURN(hipFree(d_tmp.image));

    setupRandomness(&d_img);
    addGaussianNoise(&d_img, &d_tmp, -1, -1);
    dump(dir, "gauss", &d_tmp);
    CUDA_CHECK_RETURN(hipFree(d_tmp.image));
    addGaussianNoise(&d_img, &d_tmp, 100.0f, 25.0f);
    dump(dir, "gauss_fixed", &d_tmp);
    CUDA_CHECK_RETURN(hipFree(d_tmp.image));
    saltAndPepperNoise(&d_img, &d_tmp, 50);
    dump(dir, "salt_pepper", &d_tmp);
    CUDA_CHECK_RETURN(hipFree(d_tmp.image));
    saltAndPepperNoise(&d_img, &d_tmp, 5);
    dump(dir, "salt_pepper5", &d_tmp);
    CUDA_CHECK_RETURN(hipFree(d_tmp.image));
    cleanupRandomness();

    float k1[9] = {1, 1, 1, 10, 1, 1, 1, 1, 1};
    float k2[9] = {1 / 16.f, 2 / 16.f, 1 / 16.f, 2 / 16.f, 4 / 16.f, 2 / 16.f, 1 / 16.f, 2 / 16.f, 1 / 16.f};
    linearFilter(&d_img, &d_tmp, k1, 3, 3);
    dump(dir, "linear_int", &d_tmp);
    CUDA_CHECK_RETURN(hipFree(d_tmp.image));
    linearFilter(&d_img, &d_tmp, k2, 3, 3);
    dump(dir, "linear_g16", &d_tmp);
    CUDA_CHECK_RETURN(hipFree(d_tmp.image));
    averageFilter(&d_img, &d_tmp, k1, 3, 3);
    dump(dir, "average", &d_tmp);
    CUDA_CHECK_RETURN(hipFree(d_tmp.image));
    int ones[9] = {1, 1, 1, 1, 1, 1, 1, 1, 1};
    int cross[9] = {0, 1, 0, 1, 1, 1, 0, 1, 0};
    medianFilter(&d_img, &d_tmp, ones, 3, 3);
    dump(dir, "median", &d_tmp);
    CUDA_CHECK_RETURN(hipFree(d_tmp.image));
    setupEdgeDetection();
    sobelFilter(&d_img, &d_tmp);
    dump(dir, "sobel", &d_tmp);
    CUDA_CHECK_RETURN(hipFree(d_tmp.image));
    compassFilter(&d_img, &d_tmp);
    dump(dir, "compass", &d_tmp);
    CUDA_CHECK_RETURN(hipFree(d_tmp.image));
    cleanupEdgeDetection();
    thresholdImage(&d_img, &d_tmp, 40);
    dump(dir, "threshold", &d_tmp);
    CUDA_CHECK_RETURN(hipFree(d_tmp.image));
    Image d_otsu;
    otsuThresholdImage(&d_img, &d_otsu);
    dump(dir, "otsu", &d_otsu);
    imageDilation(&d_otsu, &d_tmp, cross, 3, 3);
    dump(dir, "dilate", &d_tmp);
    CUDA_CHECK_RETURN(hipFree(d_tmp.image));
    imageErosion(&d_otsu, &d_tmp, cross, 3, 3);
    dump(dir, "erode", &d_tmp);
    CUDA_CHECK_RETURN(hipFree(d_tmp.image));
    CUDA_CHECK_RETURN(hipFree(d_otsu.image));

    srand(1);  // glibc's unseeded sequence, which the reference gets (imageAnalysis.cu:39-41,359-361)
    kMeansThresholding(&d_img, &d_tmp);
    dump(dir, "kmeans", &d_tmp);
    CUDA_CHECK_RETURN(hipFree(d_tmp.image));

    std::vector<float> feat = featureExtraction(&d_img, &d_tmp, hist, nullptr);
    fprintf(js, "features");
    for (float v : feat) fprintf(js, " %.9g", v);
    fprintf(js, "\n");
    fclose(js);
    CUDA_CHECK_RETURN(hipFree(d_img.image));
    CUDA_CHECK_RETURN(hipFree(d_rgb.image));
    printf("api_check ok\n");
    return 0;
}
