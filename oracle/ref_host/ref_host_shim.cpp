// C entry points over the reference's own host-only functions (cut out of its .cu files by
// slice_reference.py and included below unmodified).  TEST INFRASTRUCTURE: built only where
// /root/reference exists, used only by tests/golden/make_ref_host_golden.py to write the
// reference-made vectors under tests/golden/.  Nothing here restates the reference: the
// headers below are the standard ones those functions were written against.
#include <algorithm>
#include <cstdlib>
#include <cstring>

using namespace std;  // the reference's files say `using namespace std` (min in calculateCellArea)

#include REF_SLICE

extern "C" {
// imageAnalysis.cu:39-41; the state of libc rand() is the caller's business (srand)
int ref_random(int lo, int hi) { return random(lo, hi); }
// imageAnalysis.cu:155-176
void ref_equalizeHistogram(int *original, int *mappings, int numPixels) { equalizeHistogram(original, mappings, numPixels); }
// imageAnalysis.cu:281-293: centroids = k current values followed by k new ones; updated in place
int ref_centroidsHaveChanged(unsigned char *centroids, int k) {
    int count = 0;
    return centroidsHaveChanged(centroids, &count, k) ? 1 : 0;
}
// imageAnalysis.cu:682-688
int ref_arraySum(int *array, int n) { return arraySum(array, n); }
// imageMachineLearning.cu:341-391
int ref_calcMode(int *hist) { return calcMode(hist); }
int ref_calculateCellArea(int *hist, int numPixels) { return calculateCellArea(hist, numPixels); }
int ref_calculateCellPerimeter(int *hist) { return calculateCellPerimeter(hist); }
void ref_calcMinAndMaxOfRange(int *hist, int *lo, int *hi) {
    *lo = -1;  // featureExtraction starts both at -1 (imageMachineLearning.cu:406-407)
    *hi = -1;
    calcMinAndMaxOfRange(lo, hi, hist);
}
// imageMachineLearning.cu:214-255
int ref_num_classes(void) { return NUM_CLASSES; }
void ref_confusion_metrics(int *predictions, float *dataset, int numAttributes, int numInstances, int *confusion_out,
                           float *accuracy, float *precision, float *recall) {
    int *cm = computeConfusionMatrix(predictions, dataset, numAttributes, numInstances);
    memcpy(confusion_out, cm, sizeof(int) * NUM_CLASSES * NUM_CLASSES);
    *accuracy = computeAccuracy(cm, numInstances);
    computePrecisionAndRecall(cm, precision, recall);
    free(cm);
}
}
