// Device functions for the 256-entry-per-image work (equalisation map, Otsu threshold,
// mean / std-dev, Gaussian amplitude), shared by the stand-alone entry points and the
// pipeline executor.  All follow the reference's arithmetic order; see each citation.
#pragma once
#include "iak_common.hpp"

namespace iak {

// ---- equalizeHistogram (imageAnalysis.cu:155-176): sequential float32 cdf ------------------
// h: 256 counts with bin 0 empty.  m: 256 ints.
template <class HistPtr, class MapPtr>
__device__ inline void eq_map_from_hist(HistPtr h, MapPtr m, int numPixels) {
    float cdf = 0.0f;
    const float n = (float)numPixels;
    for (int i = 0; i < 256; i++) {
        float pdf = (float)(int)h[i] / n;
        cdf = (i > 0) ? (pdf + cdf) : pdf;
        m[i] = (int)(cdf * 255);
    }
}

// ---- Otsu's loop (imageAnalysis.cu:255-277); hist has bin 0 empty ----------------------------
__device__ inline int otsu_threshold_from_hist(const uint32_t *h, int totalPixels) {
    // sum of pixel values: zeros add nothing, so the zero-less histogram gives it exactly
    long long sum64 = 0;
    for (int i = 1; i < 256; i++) sum64 += (long long)h[i] * i;
    int sum = (int)sum64;
    float varMax = 0;
    int threshold = 0;
    int backgroundSum = 0;
    int nB = 0;
    for (int i = 0; i < 256; i++) {
        int hi = (i == 0) ? 0 : (int)h[i];
        nB += hi;
        if (nB == 0) continue;
        int nF = totalPixels - nB;
        if (nF == 0) continue;
        backgroundSum += i * hi;
        float bMean = (float)(backgroundSum / nB);
        float fMean = (float)((sum - backgroundSum) / nF);
        float bcv = (float)nB * (float)nF * (bMean - fMean) * (bMean - fMean);
        if (bcv > varMax) {
            varMax = bcv;
            threshold = i;
        }
    }
    return threshold;
}

// ---- calculateMeanAndStdDev (imageAnalysis.cu:106-128) from a zero-less histogram -------------
__device__ inline void mean_std_from_hist(const uint32_t *h, int numPixels, float *mean, float *stddev) {
    long long sum = 0, nonzero = 0;
    for (int v = 1; v < 256; v++) {
        sum += (long long)h[v] * v;
        nonzero += h[v];
    }
    const long long zeros = (long long)numPixels - nonzero;
    const float m = (float)sum / (float)numPixels;
    double acc = 0.0;
    for (int v = 0; v < 256; v++) {
        double d = (double)((float)v - m);
        long long c = (v == 0) ? zeros : (long long)h[v];
        acc += (double)c * (d * d);
    }
    float variance = (float)acc / (float)((long long)numPixels - 1);
    *mean = m;
    *stddev = sqrtf(variance);
}

// exp in double from + - * only (no fused multiply-add, no library call), so the bits do
// not depend on a maths library.  |error| ~1e-16 relative.
__device__ inline double portable_exp(double x) {
    if (x != x) return x;
    if (x > 709.0) return __longlong_as_double(0x7ff0000000000000LL);
    if (x < -708.0) return 0.0;
    const double INV_LN2 = 1.4426950408889634074;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double kf = x * INV_LN2;
    long long k = (long long)(kf + (kf < 0 ? -0.5 : 0.5));
    double dk = (double)k;
    double r = (x - dk * LN2_HI) - dk * LN2_LO;
    double p = 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    if (k < -1022) return 0.0;
    if (k > 1023) return __longlong_as_double(0x7ff0000000000000LL);
    return p * __longlong_as_double((long long)(k + 1023) << 52);
}

// imageAnalysis.cu:841 for one draw value
__device__ inline uint16_t gauss_entry(int rv, float mean, float variance) {
    float d = (float)rv - mean;
    float q = (d * d) / (2 * variance);
    float x = -1.0f * q;
    float e = (float)portable_exp((double)x);
    float noise = roundf(255 * e);
    if (noise != noise) return 0x100;
    return (uint16_t)(((uint32_t)(int)noise) & 0xffu);
}


// curand_uniform: x * 2^-32 + 2^-33 as two float operations
// (2.3283064e-10f is 2^-32 exactly, so the product is exact and a compiler that fuses the two operations --
// nvcc by default -- gets the same bits)
__device__ __forceinline__ float xw_uniform(uint32_t x) {
    float f = (float)x * 2.3283064e-10f;
    return f + (2.3283064e-10f / 2.0f);
}

// 16 pixels through a 256-byte table in LDS
__device__ __forceinline__ u32x4 lut16(const uint8_t *lut, u32x4 px) {
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        uint32_t x = px[w];
        o[w] = pack4(lut[x & 0xff], lut[(x >> 8) & 0xff], lut[(x >> 16) & 0xff], lut[x >> 24]);
    }
    return o;
}


// salt-and-pepper code plane (0 keep, 1 pepper, 2 salt) applied to 4 packed pixels
__device__ __forceinline__ uint32_t sp4(uint32_t px, uint32_t code) {
    // per byte: any code bit set -> drop the pixel; code bit 1 set -> 255
    uint32_t hit = (code | (code >> 1)) & 0x01010101u;
    uint32_t salt = (code >> 1) & 0x01010101u;
    uint32_t hitmask = (hit << 8) - hit;
    uint32_t saltmask = (salt << 8) - salt;
    return (px & ~hitmask) | saltmask;
}


}  // namespace iak
