/*
 * imageAnalysis.h -- drop-in for the reference's imageAnalysisLib/imageAnalysis.h
 * (same structs, same 29 free functions, same ownership rules), implemented on the
 * MI355X kernel library behind include/iak.h.  Link with libimageAnalysis.so.
 *
 * Differences a caller can see, all at points where the reference is undefined:
 *   - device memory comes from the HIP runtime: free an image with hipFree() (or
 *     iak_free()) where the reference's driver calls cudaFree();
 *   - CUDA_CHECK_RETURN keeps its name (driver code uses it, test/test.cu:338) and
 *     checks a hipError_t;
 *   - races / out-of-bounds reads of the reference kernels are resolved as the
 *     operator comments in iak.h state.
 */
#ifndef IMAGEANALYSIS_LIBRARY_H
#define IMAGEANALYSIS_LIBRARY_H

#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <iostream>

/* imageAnalysis.h:11-23 of the reference: planar 1-D byte buffers; the same struct
 * describes host and device images, only the pointer's address space differs. */
typedef struct {
    unsigned char *image;
    short channels;
    int width;
    int height;
} RGBImage;

typedef struct {
    unsigned char *image;
    int width;
    int height;
} Image;

/* imageAnalysis.h:29-36 of the reference: report and exit(1). */
static inline void CheckHipErrorAux(const char *file, unsigned line, const char *statement, hipError_t err) {
    if (err == hipSuccess) return;
    std::cerr << statement << " returned " << hipGetErrorString(err) << "(" << err << ") at " << file << ":" << line
              << std::endl;
    exit(1);
}
#define CUDA_CHECK_RETURN(value) CheckHipErrorAux(__FILE__, __LINE__, #value, value);

/* Prototypes: imageAnalysis.h:46-111 of the reference, unchanged. */
void copyHostImageToDevice(Image *host, Image *device);
void copyDeviceRGBImageToHost(RGBImage *device, RGBImage *host);
void convertRGBToGrayscale(RGBImage *d_rgb, Image *d_gray, int method);
void copyHostRGBImageToDevice(RGBImage *host, RGBImage *device);
void copyDeviceImageToHost(Image *device, Image *host);
void calculateHistogram(Image *image, int *h_histogram, int *d_histogram);
void equalizeHistogram(int *original, int *mappings, int numPixels);
void equalizeImageWithHist(Image *image, Image *d_equalizedImage, int *h_mappings);
void linearFilter(Image *image, Image *output, float *kernel, int kWidth, int kHeight);
void averageFilter(Image *image, Image *output, float *kernel, int kWidth, int kHeight);
void medianFilter(Image *image, Image *output, int *kernel, int kWidth, int kHeight);
void sobelFilter(Image *image, Image *output);
void compassFilter(Image *image, Image *output);
void saltAndPepperNoise(Image *image, Image *output, int level);
void addGaussianNoise(Image *image, Image *output, float mean, float stdDev);
void cleanupEdgeDetection();
void setupEdgeDetection();
void cleanupRandomness();
void imageDilation(Image *image, Image *output, int *structuringElement, int kWidth, int kHeight);
void imageErosion(Image *image, Image *output, int *structuringElement, int kWidth, int kHeight);
void thresholdImage(Image *image, Image *output, int threshold);
void otsuThresholdImage(Image *image, Image *output);
void kMeansThresholding(Image *image, Image *output, int k);
void kMeansThresholding(Image *image, Image *output);
void extractSingleColorChannel(RGBImage *rgb, Image *out, int color);
void setupRandomness(Image *image);
void imageQuantization(Image *image, Image *output, int *levels, int numLevels);
int calcMSQE(Image *d_image, Image *d_tempImage);
void calculateMeanAndStdDev(Image *image, float *mean, float *stdDev);

#endif
