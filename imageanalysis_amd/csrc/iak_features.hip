// Feature extraction and k-means thresholding for batches, as histogram arithmetic.
//
// featureExtraction (imageMachineLearning.cu:393-415) makes five image passes per image
// (statistics, Otsu threshold, histogram of the mask, compass filter, histogram of the edges,
// histogram of the image).  Everything but the edge count is a function of the image's
// 256-bin histogram; the edge count needs the compass filter of the Otsu mask, which is
// computed here straight from the image on bit masks (one bit per mask pixel, the filter as a
// 512-entry table) and counted without being stored: one histogram pass (skipped when the
// caller has the histogram) plus one read of the image.
//
// kMeansThresholding (imageAnalysis.cu:281-379): a pixel's label depends on its value only,
// so each iteration's class sums and counts come from the histogram; the image is touched by
// one table pass at the end.
#include "iak_block.hpp"
#include "iak_bitmask.hpp"
#include "iak_chain_stencil.hpp"
#include "iak_device_fns.hpp"
#include "iak_common.hpp"

#include <cstddef>
#include <cstring>
#include <mutex>
#include <vector>

namespace iak {

#define FEAT_N 7

// Scratch of iak_features_static: per image a 256-bin histogram, an edge counter and a
// threshold.  One copy per GPU, held until the process ends; host code reaches it only through
// the per-device address that features_static_scratch() looks up.
#define FEAT_STATIC_MAX_IMAGES 1024
// Look-up tables of the edge count (see perimeter_bits_kernel): built by the statistics kernel,
// copied into LDS by every workgroup of the edge count.
struct PerimeterTables {
    uint8_t two[4096];      // responses of two adjacent pixels, summed
    uint32_t spread[1024];  // bit i -> bit 3 i
    uint8_t one[512];       // response of one pixel
};
struct FeatStaticScratch {
    uint32_t hist[FEAT_STATIC_MAX_IMAGES * 256];
    unsigned long long counts[FEAT_STATIC_MAX_IMAGES];
    int32_t thr[FEAT_STATIC_MAX_IMAGES];
    __attribute__((aligned(16))) PerimeterTables tables;
};
static __device__ __align__(256) unsigned char scratch[sizeof(FeatStaticScratch)];

// ---- the edge count on bit masks -----------------------------------------------------------
// The compass filter runs on the Otsu mask, whose pixels are 0 or 255, so its output at a pixel
// is a function of the nine mask bits around it.  is255[idx] says whether that output is 255,
// evaluated once on the host with the filter's own arithmetic (imageAnalysis.cu:592-639: four
// masks, the running sum carried from mask to mask, the largest |sum| seen, low byte).  The index
// is column-major: bit 3 * j + i = mask pixel at column offset j - 1, row offset i - 1, so that the
// indices of neighbouring pixels are overlapping bit fields of one word (see perimeter_bits_kernel).
struct CompassLut {
    uint8_t is255[512];
};
static CompassLut make_compass_lut() {
    CompassLut t;
    for (int idx = 0; idx < 512; idx++) {
        int p[9];  // row-major, as the filter's masks are written
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) p[3 * i + j] = ((idx >> (3 * j + i)) & 1) * 255;
        const int s0 = -p[0] + p[2] - 2 * p[3] + 2 * p[5] - p[6] + p[8];
        const int s1 = -2 * p[0] - p[1] - p[3] + p[5] + p[7] + 2 * p[8];
        const int s2 = -p[0] - 2 * p[1] - p[2] + p[6] + 2 * p[7] + p[8];
        const int s3 = -p[1] - 2 * p[2] + p[3] - p[5] + 2 * p[6] + p[7];
        int sum = s0 < 0 ? -s0 : s0, mx = sum;
        sum += s1; sum = sum < 0 ? -sum : sum; mx = sum > mx ? sum : mx;
        sum += s2; sum = sum < 0 ? -sum : sum; mx = sum > mx ? sum : mx;
        sum += s3; sum = sum < 0 ? -sum : sum; mx = sum > mx ? sum : mx;
        t.is255[idx] = (mx & 0xff) == 255 ? 1 : 0;
    }
    return t;
}

// One workgroup of 256 lanes per image, lane v = grey value v.
// features[b] = {mean, std, area, (perimeter: later), mode, rangeMin, rangeMax};
// counts[b] = 0; thr_out[b] = the Otsu threshold (the mask is pixel > threshold, imageAnalysis.cu:207-216), or
// FEAT_EMPTY_MASK when no pixel lies above it: the mask is then all zero, so is its compass image, the perimeter
// feature is 0 (written here) and the edge count has nothing to read (every two-valued image with both values present ends here: Otsu's threshold on
// its one-bin histogram is 255).
// The workgroup past the last image writes the edge count's tables (tables != nullptr).
#define FEAT_EMPTY_MASK 256
__global__ __launch_bounds__(256) void features_stats_kernel(const uint32_t *__restrict__ hist, int numPixels, int B,
                                                             float *__restrict__ features, unsigned long long *__restrict__ counts,
                                                             int32_t *__restrict__ thr_out, PerimeterTables *__restrict__ tables,
                                                             CompassLut lut, int mark_empty) {
    __shared__ BlockScratch bs;
    const int b = blockIdx.x, v = threadIdx.x;
    if (b == B) {  // uniform over the workgroup
        uint8_t *one = reinterpret_cast<uint8_t *>(bs.ibuf);
        for (int i = v; i < 512; i += 256) one[i] = tables->one[i] = lut.is255[i];
        for (int i = v; i < 1024; i += 256) {
            uint32_t x = (uint32_t)i;
            x = (x | (x << 16)) & 0x030000ffu;
            x = (x | (x << 8)) & 0x0300f00fu;
            x = (x | (x << 4)) & 0x030c30c3u;
            x = (x | (x << 2)) & 0x09249249u;
            tables->spread[i] = x;
        }
        __syncthreads();
        for (int i = v; i < 4096; i += 256) tables->two[i] = one[i & 0x1ff] + one[i >> 3];
        return;
    }
    const uint32_t hv = v == 0 ? 0u : hist[(long long)b * 256 + v];  // bin 0 is always empty
    block_mean_std(hv, numPixels, bs);                            // imageMachineLearning.cu:396
    const float mean = bs.ms[0], sd = bs.ms[1];
    block_otsu(hv, numPixels, bs);                                // :399 -> imageAnalysis.cu:255-277
    const int thr = bs.thr;
    const long long above = block_sum_ll(v > thr ? (long long)hv : 0, bs.red);  // pixels the mask sets to 255
    const long long rest = (long long)numPixels - above;
    const long long area = above < rest ? above : rest;           // :353-360
    // mode (:341-351): the first bin with the largest count, -1 when every count is 0;
    // key = count, then the smaller value wins
    const long long key = block_max_ll(((long long)hv << 8) | (long long)(255 - v), bs.red);
    const int mode = (key >> 8) == 0 ? -1 : 255 - (int)(key & 255);
    // range (:379-390): first and last non-empty bin, -1 when there is none
    const long long lo_key = block_max_ll(hv ? (long long)(255 - v) : -1, bs.red);
    const long long hi_key = block_max_ll(hv ? (long long)v : -1, bs.red);
    if (v == 0) {
        float *f = features + (long long)b * FEAT_N;
        f[0] = mean;
        f[1] = sd;
        f[2] = (float)(int)area;
        f[4] = (float)mode;
        f[5] = lo_key < 0 ? -1.0f : (float)(255 - (int)lo_key);
        f[6] = hi_key < 0 ? -1.0f : (float)(int)hi_key;
        counts[b] = 0;
        thr_out[b] = (above == 0 && mark_empty) ? FEAT_EMPTY_MASK : thr;
        if (above == 0 && mark_empty) f[3] = 0.0f;  // the edge count's workgroups for this image leave at once
    }
}

// perimeter = number of 255s in the compass image of the mask (:362-366), from that image's histogram
__global__ void features_perimeter_kernel(const uint32_t *__restrict__ edge_hist, float *__restrict__ features, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) features[(long long)b * FEAT_N + 3] = (float)(int)edge_hist[(long long)b * 256 + 255];
}

// The top bit of byte b of the result = byte b of x is above the threshold t: the carry out of the
// byte-wise sum x + (255 - t), which is the majority of the two top bits and the carry into the top
// bit.  k7f = (255 - t) & 0x7f and kw = 255 - t in every byte; the other bits of the result are 0.
__device__ __forceinline__ uint32_t above4_top(uint32_t x, uint32_t k7f, uint32_t kw) {
    const uint32_t lo = (x & 0x7f7f7f7fu) + k7f;  // top bit of each byte: the carry into it
    const uint32_t differ = x ^ lo;
    const uint32_t maj = (differ & kw) | (~differ & x);  // top bits: x and lo agree ? theirs : kw's
    return maj & 0x80808080u;
}

// counts[b] += number of pixels whose compass response on the mask (pixel > thr[b]) is 255.
// A lane owns 8 pixels x 8 rows: ten 16-byte loads, each turned into the 10 mask bits of
// columns col0-1 .. col0+8 and spread to every third bit (a 1024-entry table), so that three
// consecutive rows OR-ed at shifts 0, 1, 2 put the nine bits around pixel p at bits 3p .. 3p+8:
// the twelve bits from 3p on index a table of the summed responses of pixels p and p + 1 -- one
// bit-field extract and one LDS byte per two pixels.
// Flat view of the image as everywhere: the pixel right of a row's last one is the next row's
// first, pixels outside the image are 0 (never above a threshold >= 0), and the filter's border
// (row 0, column 0) is not counted.  W % 8 == 0, images 4-byte aligned.
__global__ __launch_bounds__(256) void perimeter_bits_kernel(const uint8_t *__restrict__ in, int W, int H, long long stride,
                                                             const int32_t *__restrict__ thr,
                                                             unsigned long long *__restrict__ counts,
                                                             const PerimeterTables *__restrict__ tables,
                                                             float *__restrict__ features) {
    __shared__ __attribute__((aligned(16))) PerimeterTables t;
    __shared__ uint32_t part[4];
    const int b = blockIdx.y;
    // an empty mask has an all-zero compass image: nothing to read, and the statistics kernel has written the
    // feature (0) already (uniform over the workgroup)
    if (thr[b] >= FEAT_EMPTY_MASK) return;
    {
        const u32x4 *from = reinterpret_cast<const u32x4 *>(tables);
        u32x4 *to = reinterpret_cast<u32x4 *>(&t);
        for (int i = threadIdx.x; i < (int)(sizeof(PerimeterTables) / 16); i += 256) to[i] = from[i];
    }
    __syncthreads();
    const uint8_t *src = in + (long long)b * stride;
    const int N = W * H;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, N, 0x00020000);
    const uint32_t kw = (uint32_t)(255 - thr[b]) * 0x01010101u, k7f = kw & 0x7f7f7f7fu;
    const int chunks = W >> 3;
    const int items = chunks * ((H + 7) >> 3);
    uint32_t n255 = 0;
    for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < items; it += gridDim.x * blockDim.x) {
        const int strip = it / chunks;
        const int col0 = (it - strip * chunks) << 3;
        const int y0 = strip << 3;
        u32x4 v[10];
#pragma unroll
        for (int r = 0; r < 10; r++) v[r] = load16_flat(rsrc, (y0 - 1 + r) * W + col0 - 4, N);
        uint32_t s[10];  // bit 3 i of row r = mask pixel (y0 - 1 + r, col0 - 1 + i)
#pragma unroll
        for (int r = 0; r < 10; r++) {
            // pixels col0-1 .. col0+8 are bytes 3 .. 12 of the load: bytes 3 and 12, then the two words between
            const uint32_t fe = above4_top(__builtin_amdgcn_perm(v[r][3], v[r][0], 0x0c0c0403u), k7f, kw);
            const uint32_t f1 = above4_top(v[r][1], k7f, kw), f2 = above4_top(v[r][2], k7f, kw);
            // the flags are worth 128 each: mask bits 0 .. 4 at bits 7 .. 11 of lo, bits 5 .. 9 at bits 7 .. 11 of hi
            const uint32_t lo = __builtin_amdgcn_udot4(f1, 0x10080402u, __builtin_amdgcn_udot4(fe, 0x00000001u, 0u, false), false);
            const uint32_t hi = __builtin_amdgcn_udot4(f2, 0x08040201u, __builtin_amdgcn_udot4(fe, 0x00001000u, 0u, false), false);
            const uint32_t byte_off = (lo >> 5) | hi;  // 4 * mask
            s[r] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(t.spread) + byte_off);
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int row = y0 + r;
            uint32_t c = s[r] | (s[r + 1] << 1) | (s[r + 2] << 2);
            if (row < 1 || row >= H) c = 0;  // index 0: no response
            n255 += (uint32_t)t.two[c & 0xfffu] + t.two[(c >> 6) & 0xfffu] + t.two[(c >> 12) & 0xfffu] + t.two[(c >> 18) & 0xfffu];
            if (col0 == 0) n255 -= t.one[c & 0x1ffu];  // column 0 is border
        }
    }
    for (int o = 32; o > 0; o >>= 1) n255 += __shfl_down(n255, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n255;
    __syncthreads();
    // One 64-bit atomic per workgroup carries its count (low 40 bits; N < 2^30) and a ticket (the
    // bits above): the value that comes back tells the image's last workgroup that it is the last,
    // and what the others had counted -- it writes the perimeter feature (:362-366).  No fence, no
    // second look at memory.
    if (threadIdx.x == 0) {
        const unsigned long long mine = (unsigned long long)part[0] + part[1] + part[2] + part[3];
        const unsigned long long before = atomicAdd(&counts[b], mine | (1ull << 40));
        if ((before >> 40) == gridDim.x - 1)
            features[(long long)b * FEAT_N + 3] = (float)(int)((before & ((1ull << 40) - 1)) + mine);
    }
}

// ---- dataset-level sums of one batch, packed for one collective -------------------------------
// packed = [256 histogram bins | C counts | C x 7 feature sums | C x 7 sums of squares], doubles.
// The reference's driver keeps the per-image feature rows and labels on the host
// (test/test.cu:541-550) and a dataset histogram (:407-411); a multi-GPU job reduces these sums
// instead.  One lane per output slot walks the images in order, so the sums do not depend on
// scheduling; the integers stay exact in a double (far below 2^53).
__global__ __launch_bounds__(256) void dataset_sums_kernel(const unsigned long long *__restrict__ hist_sum,
                                                           const float *__restrict__ features,
                                                           const int32_t *__restrict__ labels, int B, int C,
                                                           double *__restrict__ packed) {
    if (blockIdx.x == 0) {  // the histogram, as doubles
        packed[threadIdx.x] = hist_sum ? (double)hist_sum[threadIdx.x] : 0.0;
        return;
    }
    // one wave per output slot: a lane adds up its images (lane, lane + 64, ...) in image order,
    // then the 64 partial sums are folded in a fixed butterfly -- the same order on every run
    const int lane = threadIdx.x & 63;
    const int j = (int)(blockIdx.x - 1) * 4 + (int)(threadIdx.x >> 6);  // slot behind the histogram
    if (j >= C * (1 + 2 * FEAT_N)) return;
    double acc = 0.0;
    if (j < C) {
        for (int b = lane; b < B; b += 64) acc += labels[b] == j ? 1.0 : 0.0;
    } else {
        const int k = j - C;
        const bool sq = k >= C * FEAT_N;
        const int c = (sq ? k - C * FEAT_N : k) / FEAT_N, f = (sq ? k - C * FEAT_N : k) % FEAT_N;
        for (int b = lane; b < B; b += 64) {
            const double v = (double)features[(long long)b * FEAT_N + f];
            acc += labels[b] == c ? (sq ? v * v : v) : 0.0;
        }
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) packed[256 + j] = acc;
}

// ---- kMeansThresholding (imageAnalysis.cu:281-379), k = 2, on the histogram ------------------
// One wave per image, four bins per lane.  Labels: 0 = nearer to centroid 1 (the reference's
// inverted test, :298-302); class sums accumulate as unsigned int (wrapping like the
// reference's, :323), counts as int; new centroid = (uchar)floorf(sum / (float)count), NaN
// (empty class) -> 0; stop when no centroid changes or after 100 rounds (:376).  The map is
// that of the LAST labelling (:339-349).
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__global__ __launch_bounds__(64) void kmeans2_kernel(const uint32_t *__restrict__ hist, int numPixels, int B,
                                                      const int32_t *__restrict__ init, int32_t *__restrict__ map,
                                                      int32_t *__restrict__ iterations) {
    const int b = blockIdx.x, lane = threadIdx.x;
    uint32_t h[4];
    uint32_t nonzero = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int v = lane * 4 + j;
        h[j] = v == 0 ? 0u : hist[(long long)b * 256 + v];
        nonzero += h[j];
    }
    nonzero = wave_sum(nonzero);
    if (lane == 0) h[0] = (uint32_t)numPixels - nonzero;  // the histogram leaves the zeros out
    int c0 = (uint8_t)init[2 * b], c1 = (uint8_t)init[2 * b + 1];
    int count = 0;
    bool changed;
    uint32_t lab = 0;  // bit j = label of bin lane*4+j
    do {
        uint32_t s0 = 0, s1 = 0, n0 = 0, n1 = 0;
        lab = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int v = lane * 4 + j;
            const bool one = !(abs(c0 - v) > abs(c1 - v));
            lab |= (one ? 1u : 0u) << j;
            if (one) { s1 += (uint32_t)v * h[j]; n1 += h[j]; }
            else { s0 += (uint32_t)v * h[j]; n0 += h[j]; }
        }
        s0 = wave_sum(s0); s1 = wave_sum(s1); n0 = wave_sum(n0); n1 = wave_sum(n1);
        const int m0 = (int)f2u8(floorf((float)s0 / (float)(int)n0));
        const int m1 = (int)f2u8(floorf((float)s1 / (float)(int)n1));
        count++;
        changed = (c0 != m0) || (c1 != m1);
        c0 = m0;
        c1 = m1;
    } while (changed && count < 100);
#pragma unroll
    for (int j = 0; j < 4; j++) map[(long long)b * 256 + lane * 4 + j] = ((lab >> j) & 1u) ? 255 : 0;
    if (iterations && lane == 0) iterations[b] = count;
}

// ---- glibc's rand() (TYPE_3 additive feedback, x[i] = x[i-3] + x[i-31], output x >> 1) --------
// The reference seeds its k-means centroids from libc rand() without ever calling srand
// (imageAnalysis.cu:39-41,359-361), so the centroids of the n-th image are a fixed function
// of n.  state: 34 words of history + a position; one lane draws sequentially.
#define GR_WORDS 35
// Draws are consumed image by image, and within an image operation by operation, two each;
// they are stored operation-major: out[(op * images + image) * 2 + j].
__global__ void glibc_rand_centroids_kernel(uint32_t *__restrict__ state, int images, int ops_per_image,
                                            int32_t *__restrict__ out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const int count = images * ops_per_image * 2;
    uint32_t r[34];
    for (int i = 0; i < 34; i++) r[i] = state[i];
    uint32_t pos = state[34];  // index of the oldest word: history is r[pos], r[pos+1], ... (mod 34)
    for (int i = 0; i < count; i++) {
        // newest = x[n-3] + x[n-31]; with 34 words kept, x[n-34] sits at pos
        const uint32_t x = r[(pos + 31) % 34] + r[(pos + 3) % 34];
        r[pos] = x;
        pos = (pos + 1) % 34;
        const int draw = (int)(x >> 1);
        const int image = i / (2 * ops_per_image), op = (i >> 1) % ops_per_image;
        out[((long long)op * images + image) * 2 + (i & 1)] = draw / (2147483647 / 257 + 1);  // random(0, 256), imageAnalysis.cu:39-41
    }
    for (int i = 0; i < 34; i++) state[i] = r[i];
    state[34] = pos;
}

// ---- k-nearest-neighbour prediction (imageMachineLearning.cu:29-156) -------------------------
// One wave per test instance.  Distance as computeDistances (:29-43): float sum of squared
// differences over the first A-1 attributes in order, then sqrt; the last attribute is the
// class.  The reference picks its k candidates with a shared-memory tournament that races and
// reads one distance per thread slot (:103-140); here the k nearest are taken in (distance,
// index) order, which is what that code is after.  The vote (:54-90) is taken in closed form, below.
#define KNN_MAX_K 64
__device__ __forceinline__ float knn_distance(const float *a, const float *b, int A) {
    float sum = 0;
    for (int i = 0; i < A - 1; i++) sum += (a[i] - b[i]) * (a[i] - b[i]);
    return sqrtf(sum);
}
// nearer first; equal distances by index; NaN distances last
__device__ __forceinline__ bool knn_before(float d1, int i1, float d2, int i2) {
    const bool n1 = d1 != d1, n2 = d2 != d2;
    if (n1 != n2) return n2;
    if (!n1 && d1 != d2) return d1 < d2;
    return i1 < i2;
}
// The reference's vote (:54-90) counts the classes of the k nearest rows, takes the first class with the most
// votes, then asks whether any class has that many votes -- which the winner itself always has -- and if so
// starts over with k - 1 as long as k - 1 > 0.  Every call therefore ends at k = 1, and the answer is the class of
// the nearest row alone (class 0 when that label lies outside the 32 counters, where the reference would write
// out of bounds, and for k < 1, where nothing is counted).
__device__ inline int knn_vote(const int *nearest, const float *train, int k, int A) {
    if (k < 1) return 0;
    const int label = (int)train[(long long)nearest[0] * A + A - 1];
    return (label >= 0 && label < 32) ? label : 0;
}
__global__ __launch_bounds__(64) void knn_kernel(const float *__restrict__ train, const float *__restrict__ test, int numTrain,
                                                 int numTest, int A, int k, int32_t *__restrict__ predictions) {
    __shared__ int chosen[KNN_MAX_K];
    const int t = blockIdx.x, lane = threadIdx.x;
    const float *q = test + (long long)t * A;
    for (int r = 0; r < k; r++) {
        float best = 0;
        int best_i = -1;
        for (int j = lane; j < numTrain; j += 64) {
            bool taken = false;
            for (int c = 0; c < r; c++) taken = taken || chosen[c] == j;
            if (taken) continue;
            const float d = knn_distance(q, train + (long long)j * A, A);
            if (best_i < 0 || knn_before(d, j, best, best_i)) {
                best = d;
                best_i = j;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float od = __shfl_xor(best, o);
            const int oi = __shfl_xor(best_i, o);
            if (oi >= 0 && (best_i < 0 || knn_before(od, oi, best, best_i))) {
                best = od;
                best_i = oi;
            }
        }
        if (lane == 0) chosen[r] = best_i;
        __syncthreads();
    }
    if (lane == 0) predictions[t] = knn_vote(chosen, train, k, A);
}

}  // namespace iak

using namespace iak;

static void glibc_rand_initial_state(unsigned seed, uint32_t st[GR_WORDS]) {
    // srandom_r for TYPE_3: 31 words by the Lehmer recurrence, three wrapped copies, 310 discards
    std::vector<uint32_t> x(344);
    int32_t w = seed == 0 ? 1 : (int32_t)seed;
    x[0] = (uint32_t)w;
    for (int i = 1; i < 31; i++) {
        const long long hi = w / 127773, lo = w % 127773;
        long long word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        w = (int32_t)word;
        x[i] = (uint32_t)w;
    }
    for (int i = 31; i < 34; i++) x[i] = x[i - 31];
    for (int i = 34; i < 344; i++) x[i] = x[i - 31] + x[i - 3];
    for (int i = 0; i < 34; i++) st[i] = x[310 + i];  // the last 34 words, oldest first
    st[34] = 0;
}

extern "C" {

long long iak_features_workspace_bytes(int W, int H, int B) {
    if (W <= 0 || H <= 0 || B <= 0) return IAK_E_ARG;
    const long long N = (long long)W * H;
    long long bytes = 0;
    bytes += (long long)B * 256 * 4;              // histogram of the image
    bytes += (long long)B * CHAIN_MAX * 256 * 2;  // (reserved)
    bytes += (long long)B * (8 + 4);              // edge counts, thresholds
    bytes = (bytes + 255) & ~255LL;
    bytes += (sizeof(PerimeterTables) + 255) & ~255LL;
    if (W % 16 != 0 || N >= (1LL << 30)) bytes += 2 * (((long long)B * N + 255) & ~255LL) + (long long)B * 256 * 4;  // mask, edges, their histogram
    return bytes;
}

// The one-pass form (W % 16 == 0): statistics from the histogram, the perimeter from one read of
// the image on bit masks.  own_hist / counts / thr: scratch for B images.
static int features_one_pass(const uint8_t *img, int W, int H, int B, long long stride, const uint32_t *hist, float *features,
                             uint32_t *own_hist, unsigned long long *counts, int32_t *thr, PerimeterTables *tables, void *stream) {
    const long long N = (long long)W * H;
    if ((reinterpret_cast<uintptr_t>(img) & 3u) || stride % 4 != 0) return IAK_E_UNSUPPORTED;
    hipStream_t s = as_stream(stream);
    int rc;
    if (!hist) {
        if ((rc = iak_hist(img, own_hist, N, B, stride, stream))) return rc;
        hist = own_hist;
    }
    // also clears counts[0..B)
    static const CompassLut lut = make_compass_lut();
    hipLaunchKernelGGL(features_stats_kernel, dim3(B + 1), dim3(256), 0, s, hist, (int)N, B, features, counts, thr, tables, lut,
                       getenv("IAK_NO_EMPTY_MASK_SKIP") ? 0 : 1);
    IAK_LAUNCH_CHECK();
    const long long items = (long long)(W >> 3) * ((H + 7) / 8);
    // two work items per lane: a workgroup's copy of the tables costs about a fifth of an item per lane
    hipLaunchKernelGGL(perimeter_bits_kernel, dim3((unsigned)((items + 511) / 512), B), dim3(256), 0, s, img, W, H,
                       stride, thr, counts, tables, features);
    IAK_LAUNCH_CHECK();
    return 0;
}

int iak_features(const uint8_t *img, int W, int H, int B, long long stride, const uint32_t *hist, float *features,
                 void *workspace, void *stream) {
    if (!img || !features || !workspace || W <= 0 || H <= 0 || B <= 0) return IAK_E_ARG;
    const long long N = (long long)W * H;
    if (stride < N || N >= (1LL << 31)) return IAK_E_ARG;
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) return IAK_E_ARG;
    hipStream_t s = as_stream(stream);
    uint8_t *ws = static_cast<uint8_t *>(workspace);
    uint32_t *own_hist = reinterpret_cast<uint32_t *>(ws);
    unsigned long long *counts = reinterpret_cast<unsigned long long *>(ws + (size_t)B * 256 * 4 + (size_t)B * CHAIN_MAX * 256 * 2);
    int32_t *thr = reinterpret_cast<int32_t *>(counts + B);
    size_t head = (size_t)B * 256 * 4 + (size_t)B * CHAIN_MAX * 256 * 2 + (size_t)B * 12;
    head = (head + 255) & ~(size_t)255;
    PerimeterTables *tables = reinterpret_cast<PerimeterTables *>(ws + head);
    head += (sizeof(PerimeterTables) + 255) & ~(size_t)255;
    // the workspace for a W % 16 == 0 shape holds no image buffers: the one-pass form is the only one
    if (W % 16 == 0 && N < (1LL << 30))
        return features_one_pass(img, W, H, B, stride, hist, features, own_hist, counts, thr, tables, stream);
    int rc;
    if (!hist) {
        if ((rc = iak_hist(img, own_hist, N, B, stride, stream))) return rc;
        hist = own_hist;
    }
    hipLaunchKernelGGL(features_stats_kernel, dim3(B), dim3(256), 0, s, hist, (int)N, B, features, counts, thr,
                       (PerimeterTables *)nullptr, CompassLut(), 0);
    IAK_LAUNCH_CHECK();
    // any other shape: the three passes of the reference, on workspace images
    const size_t img_bytes = ((size_t)B * (size_t)N + 255) & ~(size_t)255;
    uint8_t *mask = ws + head, *edges = mask + img_bytes;
    uint32_t *edge_hist = reinterpret_cast<uint32_t *>(edges + img_bytes);
    if ((rc = iak_threshold_per_image(img, mask, N, B, stride, N, thr, stream))) return rc;
    if ((rc = iak_compass(mask, edges, W, H, B, N, N, stream))) return rc;
    if ((rc = iak_hist(edges, edge_hist, N, B, N, stream))) return rc;
    hipLaunchKernelGGL(features_perimeter_kernel, dim3((B + 255) / 256), dim3(256), 0, s, edge_hist, features, B);
    IAK_LAUNCH_CHECK();
    return 0;
}

// Address of this device's copy of `scratch`.  The array's name is only a host-side handle for
// hipGetSymbolAddress; every GPU has its own copy at its own address, looked up once per device
// with that device current.
#define FEAT_STATIC_MAX_DEVICES 64
static int features_static_scratch(FeatStaticScratch **out) {
    static std::mutex mu;
    static FeatStaticScratch *ptr_of_device[FEAT_STATIC_MAX_DEVICES];
    int dev = -1;
    IAK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= FEAT_STATIC_MAX_DEVICES) return IAK_E_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(mu);
    if (!ptr_of_device[dev]) {
        void *p = nullptr;
        IAK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return IAK_E_UNSUPPORTED;
        ptr_of_device[dev] = static_cast<FeatStaticScratch *>(p);
    }
    *out = ptr_of_device[dev];
    return 0;
}

int iak_features_static_max_images(void) { return FEAT_STATIC_MAX_IMAGES; }

int iak_features_static(const uint8_t *img, int W, int H, int B, long long stride, const uint32_t *hist, float *features,
                        void *stream) {
    if (!img || !features || W <= 0 || H <= 0 || B <= 0) return IAK_E_ARG;
    const long long N = (long long)W * H;
    if (stride < N || N >= (1LL << 31)) return IAK_E_ARG;
    if (W % 16 != 0 || N >= (1LL << 30)) return IAK_E_UNSUPPORTED;  // the other shapes need image-sized scratch
    if (B > FEAT_STATIC_MAX_IMAGES) return IAK_E_TOO_LARGE;
    FeatStaticScratch *sc = nullptr;
    const int rc = features_static_scratch(&sc);
    if (rc) return rc;
    return features_one_pass(img, W, H, B, stride, hist, features, sc->hist, sc->counts, sc->thr, &sc->tables, stream);
}

int iak_dataset_sums(const unsigned long long *hist_sum, const float *features, const int32_t *labels, int B, int C,
                     double *packed, void *stream) {
    if (!packed || B < 0 || C < 0 || C > 4096 || (B > 0 && C > 0 && (!features || !labels))) return IAK_E_ARG;
    const int slots = C * (1 + 2 * FEAT_N);  // behind the 256 histogram bins
    hipLaunchKernelGGL(dataset_sums_kernel, dim3(1 + (slots + 3) / 4), dim3(256), 0, as_stream(stream), hist_sum, features,
                       labels, B, C, packed);
    IAK_LAUNCH_CHECK();
    return 0;
}

int iak_kmeans2_map(const uint32_t *hist, int numPixels, int B, const int32_t *init_centroids, int32_t *map,
                    int32_t *iterations, void *stream) {
    if (!hist || !init_centroids || !map || numPixels <= 0 || B <= 0) return IAK_E_ARG;
    hipLaunchKernelGGL(kmeans2_kernel, dim3(B), dim3(64), 0, as_stream(stream), hist, numPixels, B, init_centroids, map,
                       iterations);
    IAK_LAUNCH_CHECK();
    return 0;
}

int iak_knn_predict(const float *train, const float *test, int numTrain, int numTest, int numAttributes, int k,
                    int32_t *predictions, void *stream) {
    if (!train || !test || !predictions || numTrain <= 0 || numTest <= 0 || numAttributes < 1 || k < 1) return IAK_E_ARG;
    if (k > numTrain) k = numTrain;
    if (k > KNN_MAX_K) return IAK_E_TOO_LARGE;
    hipLaunchKernelGGL(knn_kernel, dim3(numTest), dim3(64), 0, as_stream(stream), train, test, numTrain, numTest,
                       numAttributes, k, predictions);
    IAK_LAUNCH_CHECK();
    return 0;
}

int iak_glibc_rand_seed(uint32_t *state, unsigned seed, void *stream) {
    if (!state) return IAK_E_ARG;
    uint32_t st[GR_WORDS];
    glibc_rand_initial_state(seed, st);
    hipStream_t s = as_stream(stream);
    IAK_HIP(hipMemcpyAsync(state, st, sizeof st, hipMemcpyHostToDevice, s));
    IAK_HIP(hipStreamSynchronize(s));  // st is a stack array
    return 0;
}

int iak_glibc_rand_centroids(uint32_t *state, int images, int ops_per_image, int32_t *out, void *stream) {
    if (!state || !out || images < 0 || ops_per_image < 0) return IAK_E_ARG;
    if (images == 0 || ops_per_image == 0) return 0;
    if ((long long)images * ops_per_image > (1 << 28)) return IAK_E_TOO_LARGE;
    hipLaunchKernelGGL(glibc_rand_centroids_kernel, dim3(1), dim3(64), 0, as_stream(stream), state, images, ops_per_image,
                       out);
    IAK_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
