// Neighbourhood kernels: linear / average filter, median, Sobel, compass, erosion, dilation.
//
// Addressing model (from the reference, imageAnalysis.cu:514-521 and siblings): a pixel's
// taps are read at FLAT index i*W+j of the image, so a tap one past the right edge is the
// first pixel of the next row, and taps at or past W*H read 0.  Everything here is written
// on that flat view; (row, col) only decide the border mask.
//
// Two implementations per operation:
//   * a 3x3 path for widths that are a multiple of 16: each lane owns a 16-pixel column
//     chunk and walks down a strip of rows, keeping a three-row window in registers, so
//     every input byte is fetched from memory once per strip (+2 halo rows) with 16-byte
//     loads and every output byte leaves in a 16-byte store;
//   * a generic path (any width, any kW x kH up to IAK_MAX_TAPS), one pixel per lane.
#include "iak_common.hpp"

namespace iak {

struct Taps {  // by-value kernel argument
    float f[IAK_MAX_TAPS];
};
struct ITaps {
    int v[IAK_MAX_TAPS];
};

__device__ __forceinline__ bool is_border(int row, int col, int W, int H, int a, int s) {
    return row < a || row > (H - a) || col < s || col > (W - s);
}

__device__ __forceinline__ uint32_t tap_at(const uint8_t *img, long long N, long long idx) {
    return (idx >= 0 && idx < N) ? img[idx] : 0u;
}

// =======================================================================================
// Generic path
// =======================================================================================
enum { G_LINEAR = 0, G_AVERAGE = 1, G_ERODE = 2, G_DILATE = 3, G_COMPASS = 4, G_SOBEL = 5 };

__device__ __forceinline__ int accumulate(const uint8_t *img, long long N, int W, int row, int col, int a, int s,
                                          const float *k) {
    // imageAnalysis.cu:517-524: `int sum; sum += uchar * float` truncates at every tap
    int sum = 0;
    int n = 0;
    for (int i = row - a; i <= row + a; i++)
        for (int j = col - s; j <= col + s; j++) {
            float prod = (float)tap_at(img, N, (long long)i * W + j) * k[n];
            float acc = (float)sum + prod;
            sum = (int)acc;
            n++;
        }
    return sum;
}

template <int MODE>
__global__ __launch_bounds__(256) void generic_float_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                            int W, int H, long long in_stride, long long out_stride,
                                                            int kW, int kH, Taps taps) {
    const long long N = (long long)W * H;
    const uint8_t *src = in + (long long)blockIdx.y * in_stride;
    uint8_t *dst = out + (long long)blockIdx.y * out_stride;
    const int a = (kH - 1) / 2, s = (kW - 1) / 2;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < N; t += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(t / W), col = (int)(t - (long long)row * W);
        uint32_t o = 0;
        if (!is_border(row, col, W, H, a, s)) {
            int sum = accumulate(src, N, W, row, col, a, s, taps.f);
            if (MODE == G_LINEAR) {
                sum = sum > 255 ? 255 : (sum < 0 ? 0 : sum);
                o = (uint32_t)sum;
            } else {
                o = (uint32_t)(sum / (kW * kH)) & 0xffu;
            }
        }
        dst[t] = (uint8_t)o;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void generic_morph_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                            int W, int H, long long in_stride, long long out_stride,
                                                            int kW, int kH, ITaps se) {
    const long long N = (long long)W * H;
    const uint8_t *src = in + (long long)blockIdx.y * in_stride;
    uint8_t *dst = out + (long long)blockIdx.y * out_stride;
    const int a = (kH - 1) / 2, s = (kW - 1) / 2;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < N; t += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(t / W), col = (int)(t - (long long)row * W);
        uint32_t o = 0;
        if (MODE == G_ERODE) {
            // imageAnalysis.cu:429-447
            const uint32_t p = src[t];
            if (!is_border(row, col, W, H, a, s) && p > 0) {
                bool cond = true;
                int n = 0;
                for (int i = row - a; i <= row + a; i++)
                    for (int j = col - s; j <= col + s; j++) {
                        if (se.v[n] != 0) cond = cond && (tap_at(src, N, (long long)i * W + j) != 0);
                        n++;
                    }
                o = cond ? p : 0u;
            }
        } else {
            // imageAnalysis.cu:389-399 turned into a gather: pixel x is set iff some interior
            // foreground pixel u = x - (di*W + dj) reaches it through a non-zero element.
            int n = 0;
            for (int di = -a; di <= a; di++)
                for (int dj = -s; dj <= s; dj++) {
                    if (se.v[n] != 0) {
                        const long long u = t - ((long long)di * W + dj);
                        if (u >= 0 && u < N && src[u] > 0) {
                            const int ur = (int)(u / W), uc = (int)(u - (long long)ur * W);
                            if (!is_border(ur, uc, W, H, a, s)) o = 255u;
                        }
                    }
                    n++;
                }
        }
        dst[t] = (uint8_t)o;
    }
}

// Weighted median by bisection on the value: the answer is the smallest v with
// #(window values <= v, counted with their weights) > kernSum/2.  No scratch memory.
__global__ __launch_bounds__(256) void generic_median_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                             int W, int H, long long in_stride, long long out_stride,
                                                             int kW, int kH, int rank, ITaps wts) {
    const long long N = (long long)W * H;
    const uint8_t *src = in + (long long)blockIdx.y * in_stride;
    uint8_t *dst = out + (long long)blockIdx.y * out_stride;
    const int a = (kH - 1) / 2, s = (kW - 1) / 2;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < N; t += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(t / W), col = (int)(t - (long long)row * W);
        uint32_t o = 0;
        if (!is_border(row, col, W, H, a, s)) {
            int lo = 0, hi = 255;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                int cnt = 0, n = 0;
                for (int i = row - a; i <= row + a; i++)
                    for (int j = col - s; j <= col + s; j++) {
                        cnt += ((int)tap_at(src, N, (long long)i * W + j) <= mid) ? wts.v[n] : 0;
                        n++;
                    }
                if (cnt > rank) hi = mid; else lo = mid + 1;
            }
            o = (uint32_t)lo;
        }
        dst[t] = (uint8_t)o;
    }
}

// Sobel / compass for widths the 3x3 path does not take
template <int MODE>
__global__ __launch_bounds__(256) void generic_edge_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                           int W, int H, long long in_stride, long long out_stride) {
    const long long N = (long long)W * H;
    const uint8_t *src = in + (long long)blockIdx.y * in_stride;
    uint8_t *dst = out + (long long)blockIdx.y * out_stride;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < N; t += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(t / W), col = (int)(t - (long long)row * W);
        uint32_t o = 0;
        if (!is_border(row, col, W, H, 1, 1)) {
            int p[9];
            int n = 0;
            for (int i = row - 1; i <= row + 1; i++)
                for (int j = col - 1; j <= col + 1; j++) p[n++] = (int)tap_at(src, N, (long long)i * W + j);
            if (MODE == G_SOBEL) {
                int gx = -p[0] + p[2] - 2 * p[3] + 2 * p[5] - p[6] + p[8];
                int gy = -p[0] - 2 * p[1] - p[2] + p[6] + 2 * p[7] + p[8];
                gx = gx > 255 ? 255 : (gx < 0 ? 0 : gx);
                gy = gy > 255 ? 255 : (gy < 0 ? 0 : gy);
                o = f2u8(sqrtf((float)(gx * gx + gy * gy)));
            } else {
                int s0 = -p[0] + p[2] - 2 * p[3] + 2 * p[5] - p[6] + p[8];
                int s1 = -2 * p[0] - p[1] - p[3] + p[5] + p[7] + 2 * p[8];
                int s2 = -p[0] - 2 * p[1] - p[2] + p[6] + 2 * p[7] + p[8];
                int s3 = -p[1] - 2 * p[2] + p[3] - p[5] + 2 * p[6] + p[7];
                int sum = abs(s0);
                int mx = sum;
                sum = abs(sum + s1); mx = max(mx, sum);
                sum = abs(sum + s2); mx = max(mx, sum);
                sum = abs(sum + s3); mx = max(mx, sum);
                o = (uint32_t)mx & 0xffu;
            }
        }
        dst[t] = (uint8_t)o;
    }
}

// =======================================================================================
// 3x3 path, W % 16 == 0
// =======================================================================================
#define ROWS_PER_LANE 8

struct Row18 {        // pixels col0-1 .. col0+16 of one image row (flat addressing)
    uint32_t c[4];    // col0 .. col0+15
    uint32_t l, r;    // col0-1, col0+16
};

// J in [-1, 16]
template <int J>
__device__ __forceinline__ uint32_t px(const Row18 &q) {
    if (J < 0) return q.l;
    if (J > 15) return q.r;
    return (q.c[(J & 15) >> 2] >> (8 * (J & 3))) & 0xffu;
}

__device__ __forceinline__ Row18 load_row(const uint8_t *img, long long N, long long off, bool valid_row) {
    Row18 q;
    if (valid_row && off < N) {  // N % 16 == 0: a chunk is wholly inside or wholly outside
        u32x4 v = ld16(img + off);
        q.c[0] = v[0]; q.c[1] = v[1]; q.c[2] = v[2]; q.c[3] = v[3];
    } else {
        q.c[0] = q.c[1] = q.c[2] = q.c[3] = 0;
    }
    q.l = (valid_row && off >= 1 && off - 1 < N) ? img[off - 1] : 0u;
    q.r = (valid_row && off + 16 < N) ? img[off + 16] : 0u;
    return q;
}

// Op interface: template<int J> uint32_t at(const Row18& up, const Row18& mid, const Row18& dn) -> 0..255
// plus `static const bool kBorderZero` (false only for dilation, whose border pixels can
// be reached by a neighbour's window) and `prep(Row18&, row, col0, W, H)` applied to each
// loaded row (dilation masks non-source pixels there).
template <class Op, int J>
struct Unroll16 {
    __device__ __forceinline__ static void run(const Op &op, const Row18 &u, const Row18 &m, const Row18 &d,
                                               uint32_t *o) {
        Unroll16<Op, J - 1>::run(op, u, m, d, o);
        o[J >> 2] |= (op.template at<J>(u, m, d) & 0xffu) << (8 * (J & 3));
    }
};
template <class Op>
struct Unroll16<Op, -1> {
    __device__ __forceinline__ static void run(const Op &, const Row18 &, const Row18 &, const Row18 &, uint32_t *) {}
};

template <class Op>
__global__ __launch_bounds__(256) void stencil3_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int W,
                                                       int H, long long in_stride, long long out_stride, Op op) {
    const long long N = (long long)W * H;
    const uint8_t *src = in + (long long)blockIdx.y * in_stride;
    uint8_t *dst = out + (long long)blockIdx.y * out_stride;
    const int chunks = W >> 4;
    const int strips = (H + ROWS_PER_LANE - 1) / ROWS_PER_LANE;
    const long long items = (long long)chunks * strips;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < items;
         it += (long long)gridDim.x * blockDim.x) {
        const int strip = (int)(it / chunks);
        const int cx = (int)(it - (long long)strip * chunks);
        const int col0 = cx << 4;
        const int y0 = strip * ROWS_PER_LANE;
        Row18 up = load_row(src, N, (long long)(y0 - 1) * W + col0, y0 - 1 >= 0);
        op.prep(up, y0 - 1, col0, W, H);
        Row18 mid = load_row(src, N, (long long)y0 * W + col0, true);
        op.prep(mid, y0, col0, W, H);
#pragma unroll
        for (int r = 0; r < ROWS_PER_LANE; r++) {
            const int row = y0 + r;
            if (row >= H) break;
            Row18 dn = load_row(src, N, (long long)(row + 1) * W + col0, true);
            op.prep(dn, row + 1, col0, W, H);
            uint32_t o[4] = {0, 0, 0, 0};
            Unroll16<Op, 15>::run(op, up, mid, dn, o);
            if (Op::kBorderZero) {
                if (row < 1) o[0] = o[1] = o[2] = o[3] = 0;  // row > H-1 cannot happen
                if (col0 == 0) o[0] &= 0xffffff00u;          // col < 1; col > W-1 cannot happen
            }
            u32x4 v = {o[0], o[1], o[2], o[3]};
            st16(dst + (long long)row * W + col0, v);
            up = mid;
            mid = dn;
        }
    }
}

// median of three unsigned values: one v_med3_u32
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

struct NoPrep {
    __device__ __forceinline__ void prep(Row18 &, int, int, int, int) const {}
};

// linear / average with float weights: exact restatement of the truncating accumulate
struct Linear3Float : NoPrep {
    static const bool kBorderZero = true;
    float k[9];
    int average;
    template <int J>
    __device__ __forceinline__ uint32_t at(const Row18 &u, const Row18 &m, const Row18 &d) const {
        // (int)x for |x| < 2^31 equals truncf(x); keeping the running sum as an integral float
        // skips the int<->float round trips.  Partial sums stay far below 2^24 for sane weights,
        // where float holds every integer exactly.
        float sum = 0.0f;
        sum = truncf(sum + (float)px<J - 1>(u) * k[0]);
        sum = truncf(sum + (float)px<J>(u) * k[1]);
        sum = truncf(sum + (float)px<J + 1>(u) * k[2]);
        sum = truncf(sum + (float)px<J - 1>(m) * k[3]);
        sum = truncf(sum + (float)px<J>(m) * k[4]);
        sum = truncf(sum + (float)px<J + 1>(m) * k[5]);
        sum = truncf(sum + (float)px<J - 1>(d) * k[6]);
        sum = truncf(sum + (float)px<J>(d) * k[7]);
        sum = truncf(sum + (float)px<J + 1>(d) * k[8]);
        int s = (int)sum;
        if (average) return (uint32_t)(s / 9) & 0xffu;
        return (uint32_t)(s > 255 ? 255 : (s < 0 ? 0 : s));
    }
};

// same with integral weights: pure integer arithmetic, identical results
struct Linear3Int : NoPrep {
    static const bool kBorderZero = true;
    int k[9];
    int average;
    template <int J>
    __device__ __forceinline__ uint32_t at(const Row18 &u, const Row18 &m, const Row18 &d) const {
        int s = (int)px<J - 1>(u) * k[0] + (int)px<J>(u) * k[1] + (int)px<J + 1>(u) * k[2] +
                (int)px<J - 1>(m) * k[3] + (int)px<J>(m) * k[4] + (int)px<J + 1>(m) * k[5] +
                (int)px<J - 1>(d) * k[6] + (int)px<J>(d) * k[7] + (int)px<J + 1>(d) * k[8];
        if (average) return (uint32_t)(s / 9) & 0xffu;
        return (uint32_t)(s > 255 ? 255 : (s < 0 ? 0 : s));
    }
};

struct Sobel3 : NoPrep {
    static const bool kBorderZero = true;
    template <int J>
    __device__ __forceinline__ uint32_t at(const Row18 &u, const Row18 &m, const Row18 &d) const {
        // imageAnalysis.cu:560-561 masks; each response clamped to [0,255] by linearFilter
        int a = px<J - 1>(u), b = px<J>(u), c = px<J + 1>(u);
        int e = px<J - 1>(m), f = px<J + 1>(m);
        int g = px<J - 1>(d), h = px<J>(d), i = px<J + 1>(d);
        int gx = (c - a) + 2 * (f - e) + (i - g);
        int gy = (g - a) + 2 * (h - b) + (i - c);
        gx = gx > 255 ? 255 : (gx < 0 ? 0 : gx);
        gy = gy > 255 ? 255 : (gy < 0 ? 0 : gy);
        return f2u8(sqrtf((float)(gx * gx + gy * gy)));  // imageAnalysis.cu:552
    }
};

struct Compass3 : NoPrep {
    static const bool kBorderZero = true;
    template <int J>
    __device__ __forceinline__ uint32_t at(const Row18 &u, const Row18 &m, const Row18 &d) const {
        int a = px<J - 1>(u), b = px<J>(u), c = px<J + 1>(u);
        int e = px<J - 1>(m), f = px<J + 1>(m);
        int g = px<J - 1>(d), h = px<J>(d), i = px<J + 1>(d);
        int s0 = (c - a) + 2 * (f - e) + (i - g);
        int s1 = 2 * (i - a) + (f - b) + (h - e);
        int s2 = (g - a) + 2 * (h - b) + (i - c);
        int s3 = 2 * (g - c) + (e - b) + (h - f);
        // the running sum is carried from mask to mask (imageAnalysis.cu:607-618)
        int sum = abs(s0);
        int mx = sum;
        sum = abs(sum + s1); mx = max(mx, sum);
        sum = abs(sum + s2); mx = max(mx, sum);
        sum = abs(sum + s3); mx = max(mx, sum);
        return (uint32_t)mx & 0xffu;
    }
};

// 3x3 median of 9, all weights 1: sort the three columns (shared by neighbouring outputs
// after inlining), then max of lows / median of mids / min of highs, then their median.
struct Median3 : NoPrep {
    static const bool kBorderZero = true;
    template <int J>
    __device__ __forceinline__ void col(const Row18 &u, const Row18 &m, const Row18 &d, uint32_t &lo, uint32_t &md,
                                        uint32_t &hi) const {
        uint32_t a = px<J>(u), b = px<J>(m), c = px<J>(d);
        lo = min(a, min(b, c));
        hi = max(a, max(b, c));
        md = umed3(a, b, c);
    }
    template <int J>
    __device__ __forceinline__ uint32_t at(const Row18 &u, const Row18 &m, const Row18 &d) const {
        uint32_t l0, m0, h0, l1, m1, h1, l2, m2, h2;
        col<J - 1>(u, m, d, l0, m0, h0);
        col<J>(u, m, d, l1, m1, h1);
        col<J + 1>(u, m, d, l2, m2, h2);
        uint32_t maxlo = max(l0, max(l1, l2));
        uint32_t minhi = min(h0, min(h1, h2));
        uint32_t medmd = umed3(m0, m1, m2);
        return umed3(maxlo, medmd, minhi);
    }
};

struct Erode3 : NoPrep {
    static const bool kBorderZero = true;
    uint32_t mask;  // bit n set: structuring element n is non-zero
    template <int J>
    __device__ __forceinline__ uint32_t at(const Row18 &u, const Row18 &m, const Row18 &d) const {
        const uint32_t p = px<J>(m);
        bool ok = p > 0;
        if (mask & 1u) ok = ok && px<J - 1>(u) != 0;
        if (mask & 2u) ok = ok && px<J>(u) != 0;
        if (mask & 4u) ok = ok && px<J + 1>(u) != 0;
        if (mask & 8u) ok = ok && px<J - 1>(m) != 0;
        if (mask & 16u) ok = ok && px<J>(m) != 0;
        if (mask & 32u) ok = ok && px<J + 1>(m) != 0;
        if (mask & 64u) ok = ok && px<J - 1>(d) != 0;
        if (mask & 128u) ok = ok && px<J>(d) != 0;
        if (mask & 256u) ok = ok && px<J + 1>(d) != 0;
        return ok ? p : 0u;
    }
};

struct Dilate3 {
    static const bool kBorderZero = false;
    uint32_t mask;
    // Keep only pixels that scatter in the reference: foreground and not on the border
    // (row >= 1, col >= 1 for a 3x3 window).  The left halo byte of chunk 0 is the previous
    // row's last pixel, the right halo byte of the last chunk the next row's first.
    __device__ __forceinline__ void prep(Row18 &q, int row, int col0, int W, int H) const {
        (void)H;
        if (row < 1) q.c[0] = q.c[1] = q.c[2] = q.c[3] = 0;
        if (col0 == 0) {
            q.c[0] &= 0xffffff00u;
            if (row - 1 < 1) q.l = 0;
        } else if (row < 1) {
            q.l = 0;
        }
        if (col0 + 16 >= W || row < 1) q.r = 0;
    }
    template <int J>
    __device__ __forceinline__ uint32_t at(const Row18 &u, const Row18 &m, const Row18 &d) const {
        // output x takes element n=(di+1)*3+(dj+1) from source x-(di,dj): the window is mirrored
        bool hit = false;
        if (mask & 1u) hit = hit || px<J + 1>(d) != 0;    // di=-1,dj=-1 -> source (row+1, col+1)
        if (mask & 2u) hit = hit || px<J>(d) != 0;        // di=-1,dj=0
        if (mask & 4u) hit = hit || px<J - 1>(d) != 0;    // di=-1,dj=+1
        if (mask & 8u) hit = hit || px<J + 1>(m) != 0;    // di=0,dj=-1
        if (mask & 16u) hit = hit || px<J>(m) != 0;
        if (mask & 32u) hit = hit || px<J - 1>(m) != 0;
        if (mask & 64u) hit = hit || px<J + 1>(u) != 0;   // di=+1,dj=-1 -> source (row-1, col+1)
        if (mask & 128u) hit = hit || px<J>(u) != 0;
        if (mask & 256u) hit = hit || px<J - 1>(u) != 0;
        return hit ? 255u : 0u;
    }
};

template <class Op>
static int launch3(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                   const Op &op, void *stream) {
    const long long items = (long long)(W >> 4) * ((H + ROWS_PER_LANE - 1) / ROWS_PER_LANE);
    dim3 grid(blocks_x(items, 256, B, 8192), B);
    hipLaunchKernelGGL((stencil3_kernel<Op>), grid, dim3(256), 0, as_stream(stream), in, out, W, H, in_stride,
                       out_stride, op);
    IAK_LAUNCH_CHECK();
    return 0;
}

static bool fast3_ok(const uint8_t *in, const uint8_t *out, int W, long long in_stride, long long out_stride) {
    return (W % 16 == 0) && aligned16(in) && aligned16(out) && in_stride % 16 == 0 && out_stride % 16 == 0;
}

static int check_image_args(const uint8_t *in, const uint8_t *out, int W, int H, int B, long long in_stride,
                            long long out_stride) {
    if (!in || !out || W <= 0 || H <= 0 || B <= 0) return IAK_E_ARG;
    const long long N = (long long)W * H;
    if (in_stride < N || out_stride < N) return IAK_E_ARG;
    if (in == out) return IAK_E_ARG;  // neighbourhood reads: never in place
    return 0;
}

static dim3 generic_grid(int W, int H, int B) { return dim3(blocks_x((long long)W * H, 256, B, 16384), B); }

}  // namespace iak

using namespace iak;

extern "C" {

int iak_linear(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
               const float *kernel, int kW, int kH, int mode, void *stream) {
    int rc = check_image_args(in, out, W, H, B, in_stride, out_stride);
    if (rc) return rc;
    if (!kernel || kW < 1 || kH < 1) return IAK_E_ARG;
    if (mode != IAK_LINEAR_CLAMP && mode != IAK_LINEAR_AVERAGE) return IAK_E_ARG;
    if ((long long)kW * kH > IAK_MAX_TAPS) return IAK_E_TOO_LARGE;
    if (kW == 3 && kH == 3 && fast3_ok(in, out, W, in_stride, out_stride)) {
        bool integral = true;
        double mag = 0;
        for (int i = 0; i < 9; i++) {
            integral = integral && (kernel[i] == (float)(int)kernel[i]) && kernel[i] > -65536.f && kernel[i] < 65536.f;
            mag += 255.0 * (kernel[i] < 0 ? -kernel[i] : kernel[i]);
        }
        if (integral && mag < 16777216.0) {  // every partial sum exact in float: integer path is identical
            Linear3Int op;
            for (int i = 0; i < 9; i++) op.k[i] = (int)kernel[i];
            op.average = (mode == IAK_LINEAR_AVERAGE);
            return launch3(in, out, W, H, B, in_stride, out_stride, op, stream);
        }
        Linear3Float op;
        for (int i = 0; i < 9; i++) op.k[i] = kernel[i];
        op.average = (mode == IAK_LINEAR_AVERAGE);
        return launch3(in, out, W, H, B, in_stride, out_stride, op, stream);
    }
    // taps actually visited are (2a+1) x (2s+1), indexed consecutively (imageAnalysis.cu:518-523)
    Taps t;
    const int a = (kH - 1) / 2, s = (kW - 1) / 2;
    const int visited = (2 * a + 1) * (2 * s + 1);
    for (int i = 0; i < visited; i++) t.f[i] = kernel[i];
    if (mode == IAK_LINEAR_CLAMP)
        hipLaunchKernelGGL((generic_float_kernel<G_LINEAR>), generic_grid(W, H, B), dim3(256), 0, as_stream(stream), in,
                           out, W, H, in_stride, out_stride, kW, kH, t);
    else
        hipLaunchKernelGGL((generic_float_kernel<G_AVERAGE>), generic_grid(W, H, B), dim3(256), 0, as_stream(stream), in,
                           out, W, H, in_stride, out_stride, kW, kH, t);
    IAK_LAUNCH_CHECK();
    return 0;
}

int iak_median(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
               const int *weights, int kW, int kH, void *stream) {
    int rc = check_image_args(in, out, W, H, B, in_stride, out_stride);
    if (rc) return rc;
    if (!weights || kW < 1 || kH < 1) return IAK_E_ARG;
    if (!(kW & 1) || !(kH & 1)) return IAK_E_UNSUPPORTED;
    if ((long long)kW * kH > IAK_MAX_TAPS) return IAK_E_TOO_LARGE;
    long long kernSum = 0;
    bool ones = true;
    for (int i = 0; i < kW * kH; i++) {
        if (weights[i] < 0) return IAK_E_UNSUPPORTED;
        kernSum += weights[i];
        ones = ones && weights[i] == 1;
    }
    if (kernSum <= 0 || kernSum > 0x7fffffff) return IAK_E_UNSUPPORTED;
    if (kW == 3 && kH == 3 && ones && fast3_ok(in, out, W, in_stride, out_stride)) {
        Median3 op;
        return launch3(in, out, W, H, B, in_stride, out_stride, op, stream);
    }
    ITaps t;
    for (int i = 0; i < kW * kH; i++) t.v[i] = weights[i];
    hipLaunchKernelGGL(generic_median_kernel, generic_grid(W, H, B), dim3(256), 0, as_stream(stream), in, out, W, H,
                       in_stride, out_stride, kW, kH, (int)(kernSum / 2), t);
    IAK_LAUNCH_CHECK();
    return 0;
}

int iak_sobel(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
              void *stream) {
    int rc = check_image_args(in, out, W, H, B, in_stride, out_stride);
    if (rc) return rc;
    if (fast3_ok(in, out, W, in_stride, out_stride)) {
        Sobel3 op;
        return launch3(in, out, W, H, B, in_stride, out_stride, op, stream);
    }
    hipLaunchKernelGGL((generic_edge_kernel<G_SOBEL>), generic_grid(W, H, B), dim3(256), 0, as_stream(stream), in, out, W,
                       H, in_stride, out_stride);
    IAK_LAUNCH_CHECK();
    return 0;
}

int iak_compass(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                void *stream) {
    int rc = check_image_args(in, out, W, H, B, in_stride, out_stride);
    if (rc) return rc;
    if (fast3_ok(in, out, W, in_stride, out_stride)) {
        Compass3 op;
        return launch3(in, out, W, H, B, in_stride, out_stride, op, stream);
    }
    hipLaunchKernelGGL((generic_edge_kernel<G_COMPASS>), generic_grid(W, H, B), dim3(256), 0, as_stream(stream), in, out,
                       W, H, in_stride, out_stride);
    IAK_LAUNCH_CHECK();
    return 0;
}

static int morph(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                 const int *se, int kW, int kH, bool erode, void *stream) {
    int rc = check_image_args(in, out, W, H, B, in_stride, out_stride);
    if (rc) return rc;
    if (!se || kW < 1 || kH < 1) return IAK_E_ARG;
    if ((long long)kW * kH > IAK_MAX_TAPS) return IAK_E_TOO_LARGE;
    if (kW == 3 && kH == 3 && fast3_ok(in, out, W, in_stride, out_stride)) {
        uint32_t mask = 0;
        for (int i = 0; i < 9; i++)
            if (se[i] != 0) mask |= 1u << i;
        if (erode) {
            Erode3 op;
            op.mask = mask;
            return launch3(in, out, W, H, B, in_stride, out_stride, op, stream);
        }
        Dilate3 op;
        op.mask = mask;
        return launch3(in, out, W, H, B, in_stride, out_stride, op, stream);
    }
    ITaps t;
    const int a = (kH - 1) / 2, s = (kW - 1) / 2;
    const int visited = (2 * a + 1) * (2 * s + 1);
    for (int i = 0; i < visited; i++) t.v[i] = se[i];
    if (erode)
        hipLaunchKernelGGL((generic_morph_kernel<G_ERODE>), generic_grid(W, H, B), dim3(256), 0, as_stream(stream), in,
                           out, W, H, in_stride, out_stride, kW, kH, t);
    else
        hipLaunchKernelGGL((generic_morph_kernel<G_DILATE>), generic_grid(W, H, B), dim3(256), 0, as_stream(stream), in,
                           out, W, H, in_stride, out_stride, kW, kH, t);
    IAK_LAUNCH_CHECK();
    return 0;
}

int iak_dilate(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
               const int *se, int kW, int kH, void *stream) {
    return morph(in, out, W, H, B, in_stride, out_stride, se, kW, kH, false, stream);
}

int iak_erode(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
              const int *se, int kW, int kH, void *stream) {
    return morph(in, out, W, H, B, in_stride, out_stride, se, kW, kH, true, stream);
}

}  // extern "C"
