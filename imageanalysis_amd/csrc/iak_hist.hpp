// Histogram privatisation shared by every kernel that counts grey values: one 256x32 table of
// 32-bit counters per workgroup in LDS (32 KiB), counter (bin, lane & 31) at word
// bin*32 + (lane & 31).  A lane only ever touches the LDS bank equal to its lane number mod 32,
// so an LDS atomic wave-instruction has no bank conflict whatever the pixel values are -- a
// constant image costs the same as a uniform-random one.  Lanes l and l+32 can meet on one
// counter; they are served in different halves of the instruction.  The 32 columns are summed
// at the end and each non-empty bin is merged into the image's global histogram with one
// atomic per workgroup, so a launch wants few, large workgroups per image.
#pragma once
#include "iak_common.hpp"

namespace iak {

#define HIST_COLS 32

__device__ __forceinline__ void hist_zero(uint32_t *sh) {
    for (int i = threadIdx.x; i < 256 * HIST_COLS; i += blockDim.x) sh[i] = 0;
    __syncthreads();
}

__device__ __forceinline__ void hist_add4(uint32_t *sh, uint32_t w, int col) {
    atomicAdd(&sh[(w & 0xff) * HIST_COLS + col], 1u);
    atomicAdd(&sh[((w >> 8) & 0xff) * HIST_COLS + col], 1u);
    atomicAdd(&sh[((w >> 16) & 0xff) * HIST_COLS + col], 1u);
    atomicAdd(&sh[(w >> 24) * HIST_COLS + col], 1u);
}

// 256 lanes: lane v sums the 32 columns of bin v (rotated start: conflict-free) and merges.
// Value 0 is never reported (imageAnalysis.cu:136).
__device__ __forceinline__ void hist_flush(const uint32_t *sh, uint32_t *hist_image) {
    __syncthreads();
    int v = threadIdx.x;
    if (v < 256) {
        uint32_t s = 0;
#pragma unroll 8
        for (int i = 0; i < HIST_COLS; i++) s += sh[v * HIST_COLS + ((i + v) & (HIST_COLS - 1))];
        if (v != 0 && s != 0) atomicAdd(&hist_image[v], s);
    }
}

// The same two steps for workgroups of THREADS = 512 or 1024 lanes with every lane taking part: the clear
// as 16-byte stores, the merge with THREADS / 256 lanes per bin (a share of the columns each, rotated
// start, partial sums added across neighbouring lanes).  sh must be 16-byte aligned.
template <int THREADS>
__device__ __forceinline__ void hist_zero_wide(uint32_t *sh) {
    for (int i = threadIdx.x; i < 256 * HIST_COLS / 4; i += THREADS) reinterpret_cast<u32x4 *>(sh)[i] = u32x4{0, 0, 0, 0};
    __syncthreads();
}
template <int THREADS>
__device__ __forceinline__ void hist_flush_wide(const uint32_t *sh, uint32_t *hist_image) {
    __syncthreads();
    constexpr int PER = THREADS / 256, CPL = HIST_COLS / PER;
    const int v = threadIdx.x / PER, part = threadIdx.x % PER;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < CPL; i++) s += sh[v * HIST_COLS + ((part * CPL + i + v) & (HIST_COLS - 1))];
#pragma unroll
    for (int m = 1; m < PER; m <<= 1) s += __shfl_xor(s, m);
    if (part == 0 && v != 0 && s != 0) atomicAdd(&hist_image[v], s);  // value 0 is never reported (imageAnalysis.cu:136)
}

// A workgroup's 32 KiB table and its 256-atomic merge are amortised over the pixels it
// streams: give each (1024 lanes) at least ~256K pixels, and keep the whole launch near 512
// workgroups (two per CU) -- for a batch of 500 images that is one workgroup per image.
static int hist_blocks_x(long long N, int B) {
    long long max_by_work = (N + 262143) / 262144;
    long long want = (512 + B - 1) / B;
    long long bx = want < max_by_work ? want : max_by_work;
    return (int)(bx < 1 ? 1 : bx);
}


}  // namespace iak
