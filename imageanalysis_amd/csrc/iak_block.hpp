// Workgroup-wide helpers for the per-image 256-bin work: 256 lanes (4 waves), lane v holds the
// count of grey value v.  Shared by the executor's table prologue and feature extraction.
#pragma once
#include "iak_common.hpp"

namespace iak {

struct BlockScratch {
    long long red[4];
    int ibuf[256];
    int ibuf2[256];
    float fbuf[256];
    double dbuf[256];
    float ms[2];
    int thr;
};

__device__ __forceinline__ long long block_sum_ll(long long v, long long *scratch4) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch4[threadIdx.x >> 6] = v;
    __syncthreads();
    return scratch4[0] + scratch4[1] + scratch4[2] + scratch4[3];
}

__device__ __forceinline__ long long block_max_ll(long long v, long long *scratch4) {
    for (int o = 32; o > 0; o >>= 1) {
        const long long w = __shfl_down(v, o);
        v = w > v ? w : v;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch4[threadIdx.x >> 6] = v;
    __syncthreads();
    long long m = scratch4[0];
    for (int i = 1; i < 4; i++) m = scratch4[i] > m ? scratch4[i] : m;
    return m;
}

// inclusive prefix sum over the 256 lanes, wrapping 32-bit arithmetic (as `int` in the reference)
__device__ __forceinline__ int block_scan_i32(int v, int *buf) {
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        int t = (threadIdx.x >= (unsigned)o) ? buf[threadIdx.x - o] : 0;
        __syncthreads();
        buf[threadIdx.x] = (int)((unsigned)buf[threadIdx.x] + (unsigned)t);
        __syncthreads();
    }
    return buf[threadIdx.x];
}

// calculateMeanAndStdDev (imageAnalysis.cu:106-128) from the zero-less histogram: hv = count of
// value v (lane 0 passes 0).  The variance terms are summed by one lane in bin order, the order
// the serial form uses.  Result in s.ms[0] (mean), s.ms[1] (standard deviation); ends in a barrier.
__device__ __forceinline__ void block_mean_std(uint32_t hv, int numPixels, BlockScratch &s) {
    const int v = threadIdx.x;
    const long long sum = block_sum_ll((long long)hv * v, s.red);
    const long long nonzero = block_sum_ll((long long)hv, s.red);
    const float m = (float)sum / (float)numPixels;
    const double dv = (double)((float)v - m);
    const long long c = (v == 0) ? ((long long)numPixels - nonzero) : (long long)hv;
    s.dbuf[v] = (double)c * (dv * dv);
    __syncthreads();
    if (v == 0) {
        double acc = 0.0;
        for (int i = 0; i < 256; i++) acc += s.dbuf[i];
        const float variance = (float)acc / (float)((long long)numPixels - 1);
        s.ms[0] = m;
        s.ms[1] = sqrtf(variance);
    }
    __syncthreads();
}

// Otsu's loop (imageAnalysis.cu:255-277): running sums by prefix scan, the between-class
// variance per lane, then the first bin that attains the maximum.  Result in s.thr; ends in
// a barrier.
__device__ __forceinline__ void block_otsu(uint32_t hv, int numPixels, BlockScratch &s) {
    const int v = threadIdx.x;
    const long long sum64 = block_sum_ll((long long)hv * v, s.red);
    const int sum = (int)sum64;
    const int nB = block_scan_i32((int)hv, s.ibuf);
    const int bSum = block_scan_i32((int)((unsigned)v * hv), s.ibuf2);
    const int nF = numPixels - nB;
    float bcv = -1.0f;
    if (nB != 0 && nF != 0) {
        const float bMean = (float)(bSum / nB);
        const float fMean = (float)((sum - bSum) / nF);
        bcv = (float)nB * (float)nF * (bMean - fMean) * (bMean - fMean);
    }
    __syncthreads();
    s.fbuf[v] = bcv;
    __syncthreads();
    if (v == 0) {
        float varMax = 0;
        int t = 0;
        for (int i = 0; i < 256; i++)
            if (s.fbuf[i] > varMax) {
                varMax = s.fbuf[i];
                t = i;
            }
        s.thr = t;
    }
    __syncthreads();
}

}  // namespace iak
