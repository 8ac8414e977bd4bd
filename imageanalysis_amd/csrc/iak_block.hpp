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

// A 64-bit value from the lane that a data-parallel-primitive pattern pairs this lane with
// (quad_perm / row_mirror / row_half_mirror: register-to-register, no LDS round trip).
template <int CTRL>
__device__ __forceinline__ long long dpp_ll(long long v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(v >> 32), CTRL, 0xf, 0xf, false);
    return ((long long)hi << 32) | (long long)(unsigned)lo;
}
// lanes 0, 16, 32, 48 of a value whose rows of 16 lanes each hold their row's result
__device__ __forceinline__ long long row_result(long long v, int row) {
    const int lo = __builtin_amdgcn_readlane((int)v, 16 * row), hi = __builtin_amdgcn_readlane((int)(v >> 32), 16 * row);
    return ((long long)hi << 32) | (long long)(unsigned)lo;
}
// Sum / maximum over the wave, the same in every lane: a butterfly inside each row of 16 lanes
// (pairs 1 apart, pairs 2 apart, the two quads of a half row mirrored, the two half rows
// mirrored), then the four rows.
__device__ __forceinline__ long long wave_sum_ll(long long v) {
    v += dpp_ll<0xb1>(v);   // quad_perm [1,0,3,2]
    v += dpp_ll<0x4e>(v);   // quad_perm [2,3,0,1]
    v += dpp_ll<0x141>(v);  // row_half_mirror
    v += dpp_ll<0x140>(v);  // row_mirror
    return row_result(v, 0) + row_result(v, 1) + row_result(v, 2) + row_result(v, 3);
}
__device__ __forceinline__ long long max_ll(long long a, long long b) { return a > b ? a : b; }
__device__ __forceinline__ long long wave_max_ll(long long v) {
    v = max_ll(v, dpp_ll<0xb1>(v));
    v = max_ll(v, dpp_ll<0x4e>(v));
    v = max_ll(v, dpp_ll<0x141>(v));
    v = max_ll(v, dpp_ll<0x140>(v));
    return max_ll(max_ll(row_result(v, 0), row_result(v, 1)), max_ll(row_result(v, 2), row_result(v, 3)));
}

__device__ __forceinline__ long long block_sum_ll(long long v, long long *scratch4) {
    v = wave_sum_ll(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch4[threadIdx.x >> 6] = v;
    __syncthreads();
    return scratch4[0] + scratch4[1] + scratch4[2] + scratch4[3];
}

__device__ __forceinline__ long long block_max_ll(long long v, long long *scratch4) {
    v = wave_max_ll(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch4[threadIdx.x >> 6] = v;
    __syncthreads();
    return max_ll(max_ll(scratch4[0], scratch4[1]), max_ll(scratch4[2], scratch4[3]));
}

// Inclusive prefix sums of two values over the 256 lanes, wrapping 32-bit arithmetic (as `int` in
// the reference): a shuffle scan inside each wave, then the totals of the waves before.
// buf: 8 ints.
__device__ __forceinline__ void block_scan2_i32(int &a, int &b, int *buf) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const int ta = __shfl_up(a, o), tb = __shfl_up(b, o);
        if (lane >= o) {
            a = (int)((unsigned)a + (unsigned)ta);
            b = (int)((unsigned)b + (unsigned)tb);
        }
    }
    __syncthreads();
    if (lane == 63) {
        buf[w] = a;
        buf[4 + w] = b;
    }
    __syncthreads();
    for (int i = 0; i < w; i++) {
        a = (int)((unsigned)a + (unsigned)buf[i]);
        b = (int)((unsigned)b + (unsigned)buf[4 + i]);
    }
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
    int nB = (int)hv, bSum = (int)((unsigned)v * hv);
    block_scan2_i32(nB, bSum, s.ibuf);
    const int nF = numPixels - nB;
    float bcv = -1.0f;
    if (nB != 0 && nF != 0) {
        const float bMean = (float)(bSum / nB);
        const float fMean = (float)((sum - bSum) / nF);
        bcv = (float)nB * (float)nF * (bMean - fMean) * (bMean - fMean);
    }
    // the first bin that attains the largest variance above 0, else bin 0 (:271-275: a strict
    // comparison against a maximum that starts at 0); positive floats order as their bit patterns
    const long long key = bcv > 0.0f ? ((long long)__float_as_uint(bcv) << 8) | (long long)(255 - v) : -1;
    const long long best = block_max_ll(key, s.red);
    if (v == 0) s.thr = best < 0 ? 0 : 255 - (int)(best & 255);
    __syncthreads();
}

}  // namespace iak
