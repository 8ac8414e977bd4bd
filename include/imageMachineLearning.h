/*
 * imageMachineLearning.h -- the feature-extraction entry point of the reference's
 * imageAnalysisLib/imageMachineLearningLib/imageMachineLearning.h:16-25, which is the
 * caller of the per-pixel path in the segmentation configs.  The KNN classifier and its
 * cross-validation harness in that header are consumers of a few hundred feature rows
 * and are not part of this library.
 */
#ifndef IMAGE_MACHINE_LEARNING_H
#define IMAGE_MACHINE_LEARNING_H

#include <vector>

#include "imageAnalysis.h"

/* imageMachineLearning.cu:393-422: {mean, stdDev, area, perimeter, mode, rangeMin, rangeMax}.
 * tempImage is scratch (allocated and freed inside), h_histogram receives the image's
 * histogram, d_histogram is unused as in the reference. */
std::vector<float> featureExtraction(Image *image, Image *tempImage, int *h_histogram, int *d_histogram);
/* imageMachineLearning.cu:368-377 */
std::vector<float> extractAllPixelsAsFeatures(Image *image);

#endif
