/*
 * imageMachineLearning.h -- the feature-extraction entry point of the reference's
 * imageAnalysisLib/imageMachineLearningLib/imageMachineLearning.h:16-25, which is the
 * caller of the per-pixel path in the segmentation configs, and the KNN classifier with its
 * ten-fold cross-validation, the consumer of the feature rows.
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

/* imageMachineLearning.cu:158-197.  Rows of numAttributes floats, the last one the class; HOST
 * pointers; h_predictions receives numTest ints.  The k nearest training rows are taken in
 * (distance, index) order and put to the reference's vote (see iak_knn_predict in iak.h). */
void knn(int numTrain, int numTest, float *train, float *test, int numAttributes, int *h_predictions, int k);
/* imageMachineLearning.cu:257-339.  Ten folds: fold f rotates the rows left by f tenths and
 * tests on the last tenth (the reference passes `dataset + trainSize` as the test rows, a
 * pointer that is numAttributes times too small; the rows meant are used here).  The folds run
 * one after the other instead of on ten threads; the per-fold lines and the average are printed
 * as the reference prints them.  Returns the mean accuracy. */
float knnTenfoldCrossVal(float *dataset, int numInstances, int numAttributes, int k, float *precision, float *recall);

#endif
