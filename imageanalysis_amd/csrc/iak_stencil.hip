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

#include <cstdlib>

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

struct Row18 {        // pixels col0-1 .. col0+16 of one image row (flat addressing), packed
    uint32_t c[4];    // col0 .. col0+15
    uint32_t l, r;    // col0-1, col0+16
};

// Branch-free: the address is clamped into the image and the result zeroed where the flat
// index falls outside it, so all the loads of a strip can be in flight together.
__device__ __forceinline__ Row18 load_row(const uint8_t *img, long long N, long long off, bool valid_row) {
    Row18 q;
    const bool okc = valid_row && off < N;  // N % 16 == 0: a chunk is wholly inside or wholly outside
    const bool okl = valid_row && off >= 1 && off - 1 < N;
    const bool okr = valid_row && off + 16 < N;
    const u32x4 v = ld16(img + (okc ? off : 0));
    const uint32_t l = img[okl ? off - 1 : 0];
    const uint32_t r = img[okr ? off + 16 : 0];
    q.c[0] = okc ? v[0] : 0u; q.c[1] = okc ? v[1] : 0u; q.c[2] = okc ? v[2] : 0u; q.c[3] = okc ? v[3] : 0u;
    q.l = okl ? l : 0u;
    q.r = okr ? r : 0u;
    return q;
}

// three-input integer ops are single instructions on gfx950; the asm pins them
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t umax3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int imed3(int a, int b, int c) {
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// low bytes of four values -> one packed word, three v_perm_b32 (no masking needed)
__device__ __forceinline__ uint32_t pack4_lo(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3) {
    uint32_t p01 = __builtin_amdgcn_perm(b1, b0, 0x0c0c0400u);
    uint32_t p23 = __builtin_amdgcn_perm(b3, b2, 0x0c0c0400u);
    return __builtin_amdgcn_perm(p23, p01, 0x05040100u);
}

// A row with one pixel per register: p[i] = pixel col0-1+i, i in [0,18)
struct URow {
    uint32_t p[18];
};
__device__ __forceinline__ void unpack(const Row18 &q, URow &u) {
    u.p[0] = q.l;
    u.p[17] = q.r;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        u.p[1 + 4 * w] = q.c[w] & 0xffu;
        u.p[2 + 4 * w] = (q.c[w] >> 8) & 0xffu;
        u.p[3 + 4 * w] = (q.c[w] >> 16) & 0xffu;
        u.p[4 + 4 * w] = q.c[w] >> 24;
    }
}

// Op interface:
//   typedef ... Row;                         a prepared input row kept in registers
//   void prepare(const Row18&, row, col0, W, H, Row&)   once per loaded row
//   void compute(const Row& up, const Row& mid, const Row& dn, uint32_t o[4])   16 outputs
//   static const bool kBorderZero            false only for dilation (its border pixels can
//                                            be reached by a neighbour's window)
template <class Op, int ROWS_PER_LANE, int MIN_WAVES>
__global__ __launch_bounds__(256, MIN_WAVES) void stencil3_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int W,
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
        // all rows of the strip (and its two halo rows) are requested before any is used
        Row18 raw[ROWS_PER_LANE + 2];
#pragma unroll
        for (int r = 0; r < ROWS_PER_LANE + 2; r++)
            raw[r] = load_row(src, N, (long long)(y0 - 1 + r) * W + col0, y0 - 1 + r >= 0);
        typename Op::Row up, mid, dn;
        op.prepare(raw[0], y0 - 1, col0, W, H, up);
        op.prepare(raw[1], y0, col0, W, H, mid);
#pragma unroll
        for (int r = 0; r < ROWS_PER_LANE; r++) {
            const int row = y0 + r;
            op.prepare(raw[r + 2], row + 1, col0, W, H, dn);
            uint32_t o[4];
            op.compute(up, mid, dn, o);
            if (Op::kBorderZero) {
                if (row < 1) o[0] = o[1] = o[2] = o[3] = 0;  // row > H-1 cannot happen
                if (col0 == 0) o[0] &= 0xffffff00u;          // col < 1; col > W-1 cannot happen
            }
            u32x4 v = {o[0], o[1], o[2], o[3]};
            if (row < H) st16(dst + (long long)row * W + col0, v);  // straight-line body: loads stay hoisted
            up = mid;
            mid = dn;
        }
    }
}

struct UnpackedRows {
    typedef URow Row;
    __device__ __forceinline__ void prepare(const Row18 &q, int, int, int, int, URow &u) const { unpack(q, u); }
};

// ---- linear / average, float weights ---------------------------------------------------
// The reference accumulates `int sum; sum += uchar * float` (imageAnalysis.cu:517-524): the
// running sum is truncated toward zero after every tap.  For |x| < 2^31, (int)x equals
// truncf(x), so the sum is kept as an integral float and the int<->float round trips go.
struct FRow {
    float f[18];  // f[i] = pixel col0-1+i
};

template <bool AVERAGE>
struct Linear3Float {
    static const bool kBorderZero = true;
    typedef FRow Row;
    float k[9];
    __device__ __forceinline__ void prepare(const Row18 &q, int, int, int, int, FRow &f) const {
        f.f[0] = (float)q.l;
        f.f[17] = (float)q.r;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            f.f[1 + 4 * w] = (float)(q.c[w] & 0xffu);
            f.f[2 + 4 * w] = (float)((q.c[w] >> 8) & 0xffu);
            f.f[3 + 4 * w] = (float)((q.c[w] >> 16) & 0xffu);
            f.f[4 + 4 * w] = (float)(q.c[w] >> 24);
        }
    }
    __device__ __forceinline__ void compute(const FRow &u, const FRow &m, const FRow &d, uint32_t o[4]) const {
        int res[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            float sum = 0.0f;
            sum = truncf(sum + u.f[j] * k[0]);
            sum = truncf(sum + u.f[j + 1] * k[1]);
            sum = truncf(sum + u.f[j + 2] * k[2]);
            sum = truncf(sum + m.f[j] * k[3]);
            sum = truncf(sum + m.f[j + 1] * k[4]);
            sum = truncf(sum + m.f[j + 2] * k[5]);
            sum = truncf(sum + d.f[j] * k[6]);
            sum = truncf(sum + d.f[j + 1] * k[7]);
            sum = truncf(sum + d.f[j + 2] * k[8]);
            const int s = (int)sum;
            res[j] = AVERAGE ? s / 9 : imed3(s, 0, 255);
        }
#pragma unroll
        for (int w = 0; w < 4; w++)
            o[w] = pack4_lo((uint32_t)res[4 * w], (uint32_t)res[4 * w + 1], (uint32_t)res[4 * w + 2],
                            (uint32_t)res[4 * w + 3]);
    }
};

// Non-negative weights that are multiples of 2^-q (box / binomial masks over a power of two):
// with the sum held as an integer scaled by 2^q, "add the product, truncate" is one 24-bit
// multiply-add and one AND.  Chosen by the launcher only where every float operation of the
// path above would have been exact, so the results are identical.
template <bool AVERAGE>
struct Linear3Fixed : UnpackedRows {
    static const bool kBorderZero = true;
    uint32_t k[9];  // weight * 2^q
    int q;
    __device__ __forceinline__ void compute(const URow &u, const URow &m, const URow &d, uint32_t o[4]) const {
        const uint32_t keep = ~((1u << q) - 1u);
        int res[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            uint32_t s = __umul24(u.p[j], k[0]) & keep;
            s = (s + __umul24(u.p[j + 1], k[1])) & keep;
            s = (s + __umul24(u.p[j + 2], k[2])) & keep;
            s = (s + __umul24(m.p[j], k[3])) & keep;
            s = (s + __umul24(m.p[j + 1], k[4])) & keep;
            s = (s + __umul24(m.p[j + 2], k[5])) & keep;
            s = (s + __umul24(d.p[j], k[6])) & keep;
            s = (s + __umul24(d.p[j + 1], k[7])) & keep;
            s = (s + __umul24(d.p[j + 2], k[8])) & keep;
            const int v = (int)(s >> q);
            res[j] = AVERAGE ? v / 9 : min(v, 255);
        }
#pragma unroll
        for (int w = 0; w < 4; w++)
            o[w] = pack4_lo((uint32_t)res[4 * w], (uint32_t)res[4 * w + 1], (uint32_t)res[4 * w + 2],
                            (uint32_t)res[4 * w + 3]);
    }
};

// same with integral weights (every partial sum exact in float): 24-bit integer multiply-adds
template <bool AVERAGE>
struct Linear3Int : UnpackedRows {
    static const bool kBorderZero = true;
    int k[9];
    __device__ __forceinline__ void compute(const URow &u, const URow &m, const URow &d, uint32_t o[4]) const {
        int res[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            int s = __mul24((int)u.p[j], k[0]);
            s += __mul24((int)u.p[j + 1], k[1]);
            s += __mul24((int)u.p[j + 2], k[2]);
            s += __mul24((int)m.p[j], k[3]);
            s += __mul24((int)m.p[j + 1], k[4]);
            s += __mul24((int)m.p[j + 2], k[5]);
            s += __mul24((int)d.p[j], k[6]);
            s += __mul24((int)d.p[j + 1], k[7]);
            s += __mul24((int)d.p[j + 2], k[8]);
            res[j] = AVERAGE ? s / 9 : imed3(s, 0, 255);
        }
#pragma unroll
        for (int w = 0; w < 4; w++)
            o[w] = pack4_lo((uint32_t)res[4 * w], (uint32_t)res[4 * w + 1], (uint32_t)res[4 * w + 2],
                            (uint32_t)res[4 * w + 3]);
    }
};

// floor(sqrt(q)) & 255 for 0 <= q <= 2*255^2, as (uchar)sqrtf((float)q) gives it
// (imageAnalysis.cu:552).  The hardware square root is good to 1 ulp (3e-5 at 360); the
// true root of a non-square q stays at least 1/(2*361) below the next integer, so adding
// 0.001 before truncating repairs a perfect square that came out one ulp low and never
// lifts a non-square over an integer.
__device__ __forceinline__ uint32_t isqrt_small(uint32_t q) {
    return (uint32_t)(int)(__builtin_amdgcn_sqrtf((float)q) + 0.001f);
}

struct Sobel3 : UnpackedRows {
    static const bool kBorderZero = true;
    __device__ __forceinline__ void compute(const URow &u, const URow &m, const URow &d, uint32_t o[4]) const {
        // imageAnalysis.cu:560-561 masks, separated into column sums shared by neighbours:
        //   s = up + 2*mid + down, t = down - up;  gx = s[j+1] - s[j-1], gy = t[j-1] + 2 t[j] + t[j+1]
        int s[18], t[18];
#pragma unroll
        for (int i = 0; i < 18; i++) {
            s[i] = (int)(u.p[i] + 2 * m.p[i] + d.p[i]);
            t[i] = (int)d.p[i] - (int)u.p[i];
        }
        uint32_t res[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            int gx = s[j + 2] - s[j];
            int gy = t[j] + 2 * t[j + 1] + t[j + 2];
            gx = imed3(gx, 0, 255);  // each response is clamped by linearFilter before the combine
            gy = imed3(gy, 0, 255);
            res[j] = isqrt_small((uint32_t)(__mul24(gx, gx) + __mul24(gy, gy)));
        }
#pragma unroll
        for (int w = 0; w < 4; w++) o[w] = pack4_lo(res[4 * w], res[4 * w + 1], res[4 * w + 2], res[4 * w + 3]);
    }
};

struct Compass3 : UnpackedRows {
    static const bool kBorderZero = true;
    __device__ __forceinline__ void compute(const URow &u, const URow &m, const URow &d, uint32_t o[4]) const {
        uint32_t res[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            int a = u.p[j], b = u.p[j + 1], c = u.p[j + 2];
            int e = m.p[j], f = m.p[j + 2];
            int g = d.p[j], h = d.p[j + 1], i = d.p[j + 2];
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
            res[j] = (uint32_t)mx;
        }
#pragma unroll
        for (int w = 0; w < 4; w++) o[w] = pack4_lo(res[4 * w], res[4 * w + 1], res[4 * w + 2], res[4 * w + 3]);
    }
};

// 3x3 median, all weights 1: sort each of the 18 columns once (min3 / med3 / max3), then per
// output the max of the three lows, the median of the three middles, the min of the three
// highs, and the median of those -- 7 three-input instructions per pixel.
struct Median3 : UnpackedRows {
    static const bool kBorderZero = true;
    __device__ __forceinline__ void compute(const URow &u, const URow &m, const URow &d, uint32_t o[4]) const {
        uint32_t lo[18], md[18], hi[18];
#pragma unroll
        for (int i = 0; i < 18; i++) {
            lo[i] = umin3(u.p[i], m.p[i], d.p[i]);
            md[i] = umed3(u.p[i], m.p[i], d.p[i]);
            hi[i] = umax3(u.p[i], m.p[i], d.p[i]);
        }
        uint32_t res[16];
#pragma unroll
        for (int j = 0; j < 16; j++)
            res[j] = umed3(umax3(lo[j], lo[j + 1], lo[j + 2]), umed3(md[j], md[j + 1], md[j + 2]),
                           umin3(hi[j], hi[j + 1], hi[j + 2]));
#pragma unroll
        for (int w = 0; w < 4; w++) o[w] = pack4_lo(res[4 * w], res[4 * w + 1], res[4 * w + 2], res[4 * w + 3]);
    }
};

// ---- 3x3 morphology on bit masks -----------------------------------------------------------
// A row is reduced to one bit per pixel (bit i = pixel col0-1+i is non-zero / is a source);
// a structuring element then costs a handful of shifts and ANDs/ORs for all 16 outputs.
struct MRow {
    uint32_t bits;   // 18 bits
    uint32_t c[4];   // the packed pixels (erosion keeps the grey value)
};

// bit k of the result = byte k of x is non-zero
__device__ __forceinline__ uint32_t nonzero4(uint32_t x) {
    uint32_t t = ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x;  // bit 7 of each byte = byte != 0
    t = (t >> 7) & 0x01010101u;
    return (t * 0x10204080u) >> 28;
}
// 4 bits -> 4 bytes of 0x00 / 0xff
__device__ __forceinline__ uint32_t spread4(uint32_t bits4) {
    uint32_t t = (bits4 * 0x00204081u) & 0x01010101u;
    return (t << 8) - t;
}
__device__ __forceinline__ uint32_t row_bits(const Row18 &q) {
    return (q.l != 0 ? 1u : 0u) | (nonzero4(q.c[0]) << 1) | (nonzero4(q.c[1]) << 5) | (nonzero4(q.c[2]) << 9) |
           (nonzero4(q.c[3]) << 13) | (q.r != 0 ? (1u << 17) : 0u);
}

struct Erode3 {
    static const bool kBorderZero = true;
    typedef MRow Row;
    uint32_t mask;  // bit n set: structuring element n is non-zero
    __device__ __forceinline__ void prepare(const Row18 &q, int, int, int, int, MRow &r) const {
        r.bits = row_bits(q);
        r.c[0] = q.c[0]; r.c[1] = q.c[1]; r.c[2] = q.c[2]; r.c[3] = q.c[3];
    }
    __device__ __forceinline__ void compute(const MRow &u, const MRow &m, const MRow &d, uint32_t o[4]) const {
        // output j looks at bits j, j+1, j+2 of each row (columns j-1, j, j+1)
        uint32_t ok = 0xffffu;
        if (mask & 1u) ok &= u.bits;
        if (mask & 2u) ok &= u.bits >> 1;
        if (mask & 4u) ok &= u.bits >> 2;
        if (mask & 8u) ok &= m.bits;
        if (mask & 16u) ok &= m.bits >> 1;
        if (mask & 32u) ok &= m.bits >> 2;
        if (mask & 64u) ok &= d.bits;
        if (mask & 128u) ok &= d.bits >> 1;
        if (mask & 256u) ok &= d.bits >> 2;
        // a zero pixel stays zero whatever ok says: the AND with the pixel itself does it
#pragma unroll
        for (int w = 0; w < 4; w++) o[w] = m.c[w] & spread4((ok >> (4 * w)) & 0xfu);
    }
};

struct Dilate3 {
    static const bool kBorderZero = false;
    typedef MRow Row;
    uint32_t mask;
    // Source pixels are those that scatter in the reference: foreground and not on the border
    // (row >= 1, col >= 1 for a 3x3 window).  The left halo bit of chunk 0 is the previous
    // row's last pixel, the right halo bit of the last chunk the next row's first.
    __device__ __forceinline__ void prepare(const Row18 &q, int row, int col0, int W, int, MRow &r) const {
        uint32_t b = row_bits(q);
        if (row < 1) b = 0;
        if (col0 == 0) {
            b &= ~2u;                  // column 0
            if (row - 1 < 1) b &= ~1u;  // left halo = (row-1, W-1)
        }
        if (col0 + 16 >= W) b &= ~(1u << 17);  // right halo = (row+1, 0): column 0
        r.bits = b;
    }
    __device__ __forceinline__ void compute(const MRow &u, const MRow &m, const MRow &d, uint32_t o[4]) const {
        // output x takes element n=(di+1)*3+(dj+1) from source x-(di,dj): the window is mirrored
        uint32_t hit = 0;
        if (mask & 1u) hit |= d.bits >> 2;    // di=-1,dj=-1 -> source (row+1, col+1)
        if (mask & 2u) hit |= d.bits >> 1;
        if (mask & 4u) hit |= d.bits;
        if (mask & 8u) hit |= m.bits >> 2;
        if (mask & 16u) hit |= m.bits >> 1;
        if (mask & 32u) hit |= m.bits;
        if (mask & 64u) hit |= u.bits >> 2;   // di=+1,dj=-1 -> source (row-1, col+1)
        if (mask & 128u) hit |= u.bits >> 1;
        if (mask & 256u) hit |= u.bits;
#pragma unroll
        for (int w = 0; w < 4; w++) o[w] = spread4((hit >> (4 * w)) & 0xfu);
    }
};

// =======================================================================================
// 3x3 path, W % 8 == 0: 8 pixels per lane, one 16-byte buffer load per row
// =======================================================================================
// A lane owns the 8-pixel chunk [col0, col0+8) of a strip of rows.  For every input row it
// issues ONE buffer_load_dwordx4 at flat offset row*W + col0 - 4: bytes col0-4 .. col0+11,
// i.e. the chunk plus four halo pixels on either side, so no separate halo loads and no
// cross-lane traffic.  Neighbouring lanes overlap by half a load; the overlap is served by
// the vector L1.  The buffer descriptor covers exactly the image (num_records = W*H): dwords
// at or past the end read as 0, which is the reference's "taps past the image read 0" rule
// for free, and a negative offset (row -1) is out of range as well.  Everything is on the
// flat view of the image, so pixel col0+8 of the last chunk of a row is the next row's
// first pixel, as in the reference.
//
// With 8 pixels per lane a three-row window of unpacked pixels is 30 registers, which keeps
// the kernels near 64 VGPRs (8 waves per SIMD): these kernels are issue-bound, not
// bandwidth-bound, and a gfx950 SIMD only reaches its vector issue rate with two or more
// waves ready.
struct Raw16 {
    uint32_t d[4];  // bytes col0-4 .. col0+11
};

__device__ __forceinline__ Raw16 load_raw(__amdgpu_buffer_rsrc_t rsrc, int voffset) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, 0, 0);
    Raw16 r;
    r.d[0] = v[0]; r.d[1] = v[1]; r.d[2] = v[2]; r.d[3] = v[3];
    return r;
}

// pixel j in [-4, 11] relative to col0
template <int J>
__device__ __forceinline__ uint32_t rpx(const Raw16 &q) {
    return (q.d[(J + 4) >> 2] >> (8 * ((J + 4) & 3))) & 0xffu;
}

// unpacked pixels col0-1 .. col0+8
struct PRow {
    uint32_t p[10];
};
__device__ __forceinline__ void unpack10(const Raw16 &q, PRow &u) {
    u.p[0] = q.d[0] >> 24;
    u.p[1] = q.d[1] & 0xffu;
    u.p[2] = (q.d[1] >> 8) & 0xffu;
    u.p[3] = (q.d[1] >> 16) & 0xffu;
    u.p[4] = q.d[1] >> 24;
    u.p[5] = q.d[2] & 0xffu;
    u.p[6] = (q.d[2] >> 8) & 0xffu;
    u.p[7] = (q.d[2] >> 16) & 0xffu;
    u.p[8] = q.d[2] >> 24;
    u.p[9] = q.d[3] & 0xffu;
}

// Op interface (8-pixel form):
//   typedef ... Row;
//   void prepare(const Raw16&, row, col0, W, H, Row&)
//   void compute(const Row& up, const Row& mid, const Row& dn, uint32_t o[2])
//   static const bool kBorderZero
template <class Op, int RPL>
__global__ __launch_bounds__(256) void stencil8_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int W,
                                                       int H, long long in_stride, long long out_stride, Op op) {
    const uint8_t *src = in + (long long)blockIdx.y * in_stride;
    uint8_t *dst = out + (long long)blockIdx.y * out_stride;
    const int N = W * H;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, N, 0x00020000);
    const int chunks = W >> 3;
    const int strips = (H + RPL - 1) / RPL;
    const int items = chunks * strips;
    for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < items; it += gridDim.x * blockDim.x) {
        const int strip = it / chunks;
        const int cx = it - strip * chunks;
        const int col0 = cx << 3;
        const int y0 = strip * RPL;
        const int voff = (y0 - 1) * W + col0 - 4;
        Raw16 raw[RPL + 2];
#pragma unroll
        for (int r = 0; r < RPL + 2; r++) raw[r] = load_raw(rsrc, voff + r * W);
        if (y0 == 0 && col0 == 0) {
            // the very first chunk of the image: offset -4 is out of range as a whole on some
            // range-check implementations, so fetch from 0 and shift the dwords up by one
            Raw16 f = load_raw(rsrc, 0);
            raw[1].d[0] = 0; raw[1].d[1] = f.d[0]; raw[1].d[2] = f.d[1]; raw[1].d[3] = f.d[2];
        }
        typename Op::Row up, mid, dn;
        op.prepare(raw[0], y0 - 1, col0, W, H, up);
        op.prepare(raw[1], y0, col0, W, H, mid);
#pragma unroll
        for (int r = 0; r < RPL; r++) {
            const int row = y0 + r;
            op.prepare(raw[r + 2], row + 1, col0, W, H, dn);
            uint32_t o[2];
            op.compute(up, mid, dn, o);
            if (Op::kBorderZero) {
                if (row < 1) o[0] = o[1] = 0;        // row > H-1 cannot happen
                if (col0 == 0) o[0] &= 0xffffff00u;  // col < 1; col > W-1 cannot happen
            }
            if (row < H) {
                uint2 v = make_uint2(o[0], o[1]);
                *reinterpret_cast<uint2 *>(dst + (long long)row * W + col0) = v;
            }
            up = mid;
            mid = dn;
        }
    }
}

struct Unpacked8 {
    typedef PRow Row;
    __device__ __forceinline__ void prepare(const Raw16 &q, int, int, int, int, PRow &u) const { unpack10(q, u); }
};

template <class T>
__device__ __forceinline__ void pack8(const T *res, uint32_t o[2]) {
    o[0] = pack4_lo((uint32_t)res[0], (uint32_t)res[1], (uint32_t)res[2], (uint32_t)res[3]);
    o[1] = pack4_lo((uint32_t)res[4], (uint32_t)res[5], (uint32_t)res[6], (uint32_t)res[7]);
}

struct FRow8 {
    float f[10];
};

// linear / average, float weights: `int sum; sum += uchar * float` of the reference
// (imageAnalysis.cu:517-524) truncates after every tap; (int)x == truncf(x) for |x| < 2^31,
// so the sum stays an integral float.
template <bool AVERAGE>
struct Lin8Float {
    static const bool kBorderZero = true;
    typedef FRow8 Row;
    float k[9];
    __device__ __forceinline__ void prepare(const Raw16 &q, int, int, int, int, FRow8 &f) const {
        f.f[0] = (float)(q.d[0] >> 24);
#pragma unroll
        for (int w = 0; w < 2; w++) {
            f.f[1 + 4 * w] = (float)(q.d[1 + w] & 0xffu);
            f.f[2 + 4 * w] = (float)((q.d[1 + w] >> 8) & 0xffu);
            f.f[3 + 4 * w] = (float)((q.d[1 + w] >> 16) & 0xffu);
            f.f[4 + 4 * w] = (float)(q.d[1 + w] >> 24);
        }
        f.f[9] = (float)(q.d[3] & 0xffu);
    }
    __device__ __forceinline__ void compute(const FRow8 &u, const FRow8 &m, const FRow8 &d, uint32_t o[2]) const {
        int res[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float sum = 0.0f;
            sum = truncf(sum + u.f[j] * k[0]);
            sum = truncf(sum + u.f[j + 1] * k[1]);
            sum = truncf(sum + u.f[j + 2] * k[2]);
            sum = truncf(sum + m.f[j] * k[3]);
            sum = truncf(sum + m.f[j + 1] * k[4]);
            sum = truncf(sum + m.f[j + 2] * k[5]);
            sum = truncf(sum + d.f[j] * k[6]);
            sum = truncf(sum + d.f[j + 1] * k[7]);
            sum = truncf(sum + d.f[j + 2] * k[8]);
            const int s = (int)sum;
            res[j] = AVERAGE ? s / 9 : imed3(s, 0, 255);
        }
        pack8(res, o);
    }
};

// integral weights with every partial sum below 2^23: 24-bit integer multiply-adds, identical results
template <bool AVERAGE>
struct Lin8Int : Unpacked8 {
    static const bool kBorderZero = true;
    int k[9];
    __device__ __forceinline__ void compute(const PRow &u, const PRow &m, const PRow &d, uint32_t o[2]) const {
        int res[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int s = __mul24((int)u.p[j], k[0]);
            s += __mul24((int)u.p[j + 1], k[1]);
            s += __mul24((int)u.p[j + 2], k[2]);
            s += __mul24((int)m.p[j], k[3]);
            s += __mul24((int)m.p[j + 1], k[4]);
            s += __mul24((int)m.p[j + 2], k[5]);
            s += __mul24((int)d.p[j], k[6]);
            s += __mul24((int)d.p[j + 1], k[7]);
            s += __mul24((int)d.p[j + 2], k[8]);
            res[j] = AVERAGE ? s / 9 : imed3(s, 0, 255);
        }
        pack8(res, o);
    }
};

// Non-negative weights that are multiples of 2^-q.  With the running sum an integer, adding
// a non-negative product and truncating adds floor(product): every tap contributes
// (pixel * weight*2^q) >> q on its own.  The launcher takes this path only where each float
// operation of Lin8Float would have been exact, so the results are identical.
template <bool AVERAGE>
struct Lin8Fixed : Unpacked8 {
    static const bool kBorderZero = true;
    uint32_t k[9];  // weight * 2^q
    int q;
    __device__ __forceinline__ void compute(const PRow &u, const PRow &m, const PRow &d, uint32_t o[2]) const {
        int res[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t s = (__umul24(u.p[j], k[0]) >> q) + (__umul24(u.p[j + 1], k[1]) >> q) +
                         (__umul24(u.p[j + 2], k[2]) >> q) + (__umul24(m.p[j], k[3]) >> q) +
                         (__umul24(m.p[j + 1], k[4]) >> q) + (__umul24(m.p[j + 2], k[5]) >> q) +
                         (__umul24(d.p[j], k[6]) >> q) + (__umul24(d.p[j + 1], k[7]) >> q) +
                         (__umul24(d.p[j + 2], k[8]) >> q);
            res[j] = AVERAGE ? (int)s / 9 : (int)min(s, 255u);
        }
        pack8(res, o);
    }
};

// Symmetric smoothing kernels [a b a; b c b; a b a] whose weights are 0 or 2^-s and sum to at
// most 1 (binomial / Gaussian 3x3, small box filters): every tap contributes pixel >> s
// (see Lin8Fixed) and the nine terms can never exceed 255, so the whole filter is byte-wise
// SWAR on packed words -- shift+mask per weight, plain 32-bit adds (no carry can cross a
// byte), byte-aligned windows for the +-1 column taps -- about 4 instructions per pixel and
// no unpacking, clamping or re-packing.  Chosen by the launcher; results are identical to
// the float path because every float operation there is exact for such weights.
struct SwarRow {
    uint32_t a[4], b[4];  // (pixel >> sa), (pixel >> sb) for bytes col0-4 .. col0+11
    uint32_t c[2];        // (pixel >> sc) for the chunk itself
};

struct Lin8Swar {
    static const bool kBorderZero = true;
    typedef SwarRow Row;
    int sa, sb, sc;            // shifts; 8 means the weight is zero
    __device__ __forceinline__ static uint32_t plane(uint32_t x, int s) {
        return (x >> s) & ((0xffu >> s) * 0x01010101u);  // s == 8 gives 0
    }
    __device__ __forceinline__ void prepare(const Raw16 &q, int, int, int, int, SwarRow &r) const {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            r.a[i] = plane(q.d[i], sa);
            r.b[i] = plane(q.d[i], sb);
        }
        r.c[0] = plane(q.d[1], sc);
        r.c[1] = plane(q.d[2], sc);
    }
    __device__ __forceinline__ void compute(const SwarRow &u, const SwarRow &m, const SwarRow &d, uint32_t o[2]) const {
        uint32_t s1[4];  // terms read at columns j-1 and j+1: corners of the outer rows, edges of the middle row
#pragma unroll
        for (int i = 0; i < 4; i++) s1[i] = u.a[i] + d.a[i] + m.b[i];
        // terms read at column j: edges of the outer rows, centre
        const uint32_t s0a = u.b[1] + d.b[1] + m.c[0];
        const uint32_t s0b = u.b[2] + d.b[2] + m.c[1];
        o[0] = __builtin_amdgcn_alignbyte(s1[1], s1[0], 3) + __builtin_amdgcn_alignbyte(s1[2], s1[1], 1) + s0a;
        o[1] = __builtin_amdgcn_alignbyte(s1[2], s1[1], 3) + __builtin_amdgcn_alignbyte(s1[3], s1[2], 1) + s0b;
    }
};

struct Sobel8 : Unpacked8 {
    static const bool kBorderZero = true;
    __device__ __forceinline__ void compute(const PRow &u, const PRow &m, const PRow &d, uint32_t o[2]) const {
        // imageAnalysis.cu:560-561 masks through column sums shared by neighbouring outputs:
        //   s = up + 2*mid + down, t = down - up;  gx = s[j+1] - s[j-1], gy = t[j-1] + 2 t[j] + t[j+1]
        int s[10], t[10];
#pragma unroll
        for (int i = 0; i < 10; i++) {
            s[i] = (int)(u.p[i] + 2 * m.p[i] + d.p[i]);
            t[i] = (int)d.p[i] - (int)u.p[i];
        }
        uint32_t res[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int gx = s[j + 2] - s[j];
            int gy = t[j] + 2 * t[j + 1] + t[j + 2];
            gx = imed3(gx, 0, 255);  // each response is clamped by linearFilter before the combine
            gy = imed3(gy, 0, 255);
            res[j] = isqrt_small((uint32_t)(__mul24(gx, gx) + __mul24(gy, gy)));
        }
        pack8(res, o);
    }
};

struct Compass8 : Unpacked8 {
    static const bool kBorderZero = true;
    __device__ __forceinline__ void compute(const PRow &u, const PRow &m, const PRow &d, uint32_t o[2]) const {
        // the four masks (imageAnalysis.cu:599-602) from shared column terms
        int s[10], t[10];
#pragma unroll
        for (int i = 0; i < 10; i++) {
            s[i] = (int)(u.p[i] + 2 * m.p[i] + d.p[i]);
            t[i] = (int)d.p[i] - (int)u.p[i];
        }
        uint32_t res[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int a = u.p[j], b = u.p[j + 1], c = u.p[j + 2];
            const int e = m.p[j], f = m.p[j + 2];
            const int g = d.p[j], h = d.p[j + 1], i = d.p[j + 2];
            const int s0 = s[j + 2] - s[j];
            const int s2 = t[j] + 2 * t[j + 1] + t[j + 2];
            const int s1 = 2 * (i - a) + (f - b) + (h - e);
            const int s3 = 2 * (g - c) + (e - b) + (h - f);
            // the running sum is carried from mask to mask (imageAnalysis.cu:607-618)
            int sum = abs(s0);
            int mx = sum;
            sum = abs(sum + s1); mx = max(mx, sum);
            sum = abs(sum + s2); mx = max(mx, sum);
            sum = abs(sum + s3); mx = max(mx, sum);
            res[j] = (uint32_t)mx;
        }
        pack8(res, o);
    }
};

// 3x3 median, all weights 1: sort each of the 10 columns once (min3 / med3 / max3), then per
// output the max of the three lows, the median of the three middles, the min of the three
// highs, and the median of those.
struct Median8 : Unpacked8 {
    static const bool kBorderZero = true;
    __device__ __forceinline__ void compute(const PRow &u, const PRow &m, const PRow &d, uint32_t o[2]) const {
        uint32_t lo[10], md[10], hi[10];
#pragma unroll
        for (int i = 0; i < 10; i++) {
            lo[i] = umin3(u.p[i], m.p[i], d.p[i]);
            md[i] = umed3(u.p[i], m.p[i], d.p[i]);
            hi[i] = umax3(u.p[i], m.p[i], d.p[i]);
        }
        uint32_t res[8];
#pragma unroll
        for (int j = 0; j < 8; j++)
            res[j] = umed3(umax3(lo[j], lo[j + 1], lo[j + 2]), umed3(md[j], md[j + 1], md[j + 2]),
                           umin3(hi[j], hi[j + 1], hi[j + 2]));
        pack8(res, o);
    }
};

// morphology on bit masks: bit i of a row mask = pixel col0-1+i (i in [0,10)) is non-zero / is a source
struct MRow8 {
    uint32_t bits;
    uint32_t c[2];  // the packed pixels of the chunk (erosion keeps the grey value)
};
__device__ __forceinline__ uint32_t row_bits8(const Raw16 &q) {
    return ((q.d[0] >> 24) != 0 ? 1u : 0u) | (nonzero4(q.d[1]) << 1) | (nonzero4(q.d[2]) << 5) |
           ((q.d[3] & 0xffu) != 0 ? (1u << 9) : 0u);
}

struct Erode8 {
    static const bool kBorderZero = true;
    typedef MRow8 Row;
    uint32_t mask;  // bit n set: structuring element n is non-zero
    __device__ __forceinline__ void prepare(const Raw16 &q, int, int, int, int, MRow8 &r) const {
        r.bits = row_bits8(q);
        r.c[0] = q.d[1];
        r.c[1] = q.d[2];
    }
    __device__ __forceinline__ void compute(const MRow8 &u, const MRow8 &m, const MRow8 &d, uint32_t o[2]) const {
        uint32_t ok = 0xffu;  // output j looks at bits j, j+1, j+2 of each row
        if (mask & 1u) ok &= u.bits;
        if (mask & 2u) ok &= u.bits >> 1;
        if (mask & 4u) ok &= u.bits >> 2;
        if (mask & 8u) ok &= m.bits;
        if (mask & 16u) ok &= m.bits >> 1;
        if (mask & 32u) ok &= m.bits >> 2;
        if (mask & 64u) ok &= d.bits;
        if (mask & 128u) ok &= d.bits >> 1;
        if (mask & 256u) ok &= d.bits >> 2;
        // a zero pixel stays zero whatever ok says: the AND with the pixel itself does it
        o[0] = m.c[0] & spread4(ok & 0xfu);
        o[1] = m.c[1] & spread4((ok >> 4) & 0xfu);
    }
};

struct Dilate8 {
    static const bool kBorderZero = false;
    typedef MRow8 Row;
    uint32_t mask;
    // Source pixels are those that scatter in the reference: foreground and not on the border
    // (row >= 1, col >= 1 for a 3x3 window).  The left halo bit of chunk 0 is the previous
    // row's last pixel, the right halo bit of the last chunk the next row's first.
    __device__ __forceinline__ void prepare(const Raw16 &q, int row, int col0, int W, int, MRow8 &r) const {
        uint32_t b = row_bits8(q);
        if (row < 1) b = 0;
        if (col0 == 0) {
            b &= ~2u;                   // column 0
            if (row - 1 < 1) b &= ~1u;  // left halo = (row-1, W-1)
        }
        if (col0 + 8 >= W) b &= ~(1u << 9);  // right halo = (row+1, 0): column 0
        r.bits = b;
    }
    __device__ __forceinline__ void compute(const MRow8 &u, const MRow8 &m, const MRow8 &d, uint32_t o[2]) const {
        // output x takes element n=(di+1)*3+(dj+1) from source x-(di,dj): the window is mirrored
        uint32_t hit = 0;
        if (mask & 1u) hit |= d.bits >> 2;   // di=-1,dj=-1 -> source (row+1, col+1)
        if (mask & 2u) hit |= d.bits >> 1;
        if (mask & 4u) hit |= d.bits;
        if (mask & 8u) hit |= m.bits >> 2;
        if (mask & 16u) hit |= m.bits >> 1;
        if (mask & 32u) hit |= m.bits;
        if (mask & 64u) hit |= u.bits >> 2;  // di=+1,dj=-1 -> source (row-1, col+1)
        if (mask & 128u) hit |= u.bits >> 1;
        if (mask & 256u) hit |= u.bits;
        o[0] = spread4(hit & 0xfu);
        o[1] = spread4((hit >> 4) & 0xfu);
    }
};

template <class Op, int RPL>
static int launch8v(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                    const Op &op, void *stream) {
    const long long items = (long long)(W >> 3) * ((H + RPL - 1) / RPL);
    dim3 grid(blocks_x(items, 256, B, 1 << 20), B);
    hipLaunchKernelGGL((stencil8_kernel<Op, RPL>), grid, dim3(256), 0, as_stream(stream), in, out, W, H, in_stride,
                       out_stride, op);
    IAK_LAUNCH_CHECK();
    return 0;
}

template <class Op>
static int launch8(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                   const Op &op, void *stream) {
    static int variant = -1;
    if (variant < 0) {
        const char *e = getenv("IAK_STENCIL_RPL");
        variant = e ? atoi(e) : 8;
    }
    switch (variant) {
    case 4: return launch8v<Op, 4>(in, out, W, H, B, in_stride, out_stride, op, stream);
    case 16: return launch8v<Op, 16>(in, out, W, H, B, in_stride, out_stride, op, stream);
    default: return launch8v<Op, 8>(in, out, W, H, B, in_stride, out_stride, op, stream);
    }
}

// the 8-pixel path needs 4-byte aligned rows and chunks, and 32-bit flat offsets
static bool fast8_ok(const uint8_t *in, const uint8_t *out, int W, int H, long long in_stride, long long out_stride) {
    return (W % 8 == 0) && (reinterpret_cast<uintptr_t>(in) % 4 == 0) && (reinterpret_cast<uintptr_t>(out) % 8 == 0) &&
           in_stride % 4 == 0 && out_stride % 8 == 0 && (long long)W * H < (1LL << 30);
}

template <class Op, int RPL, int MINW>
static int launch3v(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                    const Op &op, void *stream) {
    const long long items = (long long)(W >> 4) * ((H + RPL - 1) / RPL);
    dim3 grid(blocks_x(items, 256, B, 1 << 20), B);
    hipLaunchKernelGGL((stencil3_kernel<Op, RPL, MINW>), grid, dim3(256), 0, as_stream(stream), in, out, W, H,
                       in_stride, out_stride, op);
    IAK_LAUNCH_CHECK();
    return 0;
}

template <class Op>
static int launch3(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                   const Op &op, void *stream) {
    static int variant = -1;
    if (variant < 0) {
        const char *e = getenv("IAK_STENCIL_VARIANT");
        variant = e ? atoi(e) : 0;
    }
    switch (variant) {
    case 1: return launch3v<Op, 8, 4>(in, out, W, H, B, in_stride, out_stride, op, stream);
    case 2: return launch3v<Op, 8, 6>(in, out, W, H, B, in_stride, out_stride, op, stream);
    case 3: return launch3v<Op, 4, 1>(in, out, W, H, B, in_stride, out_stride, op, stream);
    case 4: return launch3v<Op, 4, 6>(in, out, W, H, B, in_stride, out_stride, op, stream);
    case 5: return launch3v<Op, 16, 1>(in, out, W, H, B, in_stride, out_stride, op, stream);
    case 6: return launch3v<Op, 16, 4>(in, out, W, H, B, in_stride, out_stride, op, stream);
    case 7: return launch3v<Op, 4, 8>(in, out, W, H, B, in_stride, out_stride, op, stream);
    default: return launch3v<Op, 8, 1>(in, out, W, H, B, in_stride, out_stride, op, stream);
    }
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

namespace {

// IAK_STENCIL_PATH=16 selects the 16-pixel-per-lane kernels for the 3x3 operators (for A/B
// timing); the default is the 8-pixel single-load path.
bool prefer8() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("IAK_STENCIL_PATH");
        v = (e && atoi(e) == 16) ? 0 : 1;
    }
    return v == 1;
}

enum LinKind { LIN_INT, LIN_FIXED, LIN_FLOAT, LIN_SWAR };

struct LinPlan {
    LinKind kind;
    int ik[9];
    uint32_t fk[9];
    int q;
    int sa, sb, sc;  // LIN_SWAR
};

// weight -> shift if the weight is 0 (8) or 2^-s with s in [0,7]; -1 otherwise
int pow2_shift(float w) {
    if (w == 0.0f) return 8;
    for (int s = 0; s <= 7; s++)
        if (w == 1.0f / (float)(1 << s)) return s;
    return -1;
}

// [a b a; b c b; a b a] with weights 0 or 2^-s and 4a + 4b + c <= 1
bool plan_swar(const float *k, LinPlan &p) {
    if (!(k[0] == k[2] && k[0] == k[6] && k[0] == k[8] && k[1] == k[3] && k[1] == k[5] && k[1] == k[7])) return false;
    const int sa = pow2_shift(k[0]), sb = pow2_shift(k[1]), sc = pow2_shift(k[4]);
    if (sa < 0 || sb < 0 || sc < 0) return false;
    if (4.0 * k[0] + 4.0 * k[1] + k[4] > 1.0) return false;
    p.kind = LIN_SWAR;
    p.sa = sa; p.sb = sb; p.sc = sc;
    return true;
}

// Which exact-arithmetic path a 3x3 kernel admits (see Lin8Int / Lin8Fixed / Lin8Float).
LinPlan plan_linear3(const float *kernel) {
    LinPlan p{};
    bool integral = true;
    double mag = 0;
    for (int i = 0; i < 9; i++) {
        integral = integral && (kernel[i] == (float)(int)kernel[i]) && kernel[i] > -65536.f && kernel[i] < 65536.f;
        mag += 255.0 * (kernel[i] < 0 ? -kernel[i] : kernel[i]);
    }
    if (integral && mag < 8388608.0) {  // every partial sum exact in float and within 24 bits
        p.kind = LIN_INT;
        for (int i = 0; i < 9; i++) p.ik[i] = (int)kernel[i];
        return p;
    }
    // weights >= 0, all multiples of 2^-q (q <= 12), and 255 * (sum of weights) * 2^q < 2^24:
    // every product and partial sum of the float path is then exact
    bool ok = true;
    int q = 0;
    double total = 0;
    for (int i = 0; i < 9 && ok; i++) {
        const float w = kernel[i];
        if (!(w >= 0.0f) || w > 4096.0f) { ok = false; break; }
        total += w;
        int qi = 0;
        float t = w;
        while (t != (float)(int)t && qi <= 12) { t *= 2.0f; qi++; }
        if (qi > 12) ok = false;
        if (qi > q) q = qi;
    }
    if (ok && q > 0 && total * 255.0 * (double)(1 << q) < 16777216.0) {
        p.kind = LIN_FIXED;
        p.q = q;
        for (int i = 0; i < 9; i++) p.fk[i] = (uint32_t)(kernel[i] * (float)(1 << q));
        return p;
    }
    p.kind = LIN_FLOAT;
    return p;
}

template <template <bool> class IntOp, template <bool> class FixedOp, template <bool> class FloatOp, bool AVG, class L>
int run_linear3(const LinPlan &p, const float *kernel, L launch) {
    if (p.kind == LIN_INT) {
        IntOp<AVG> op;
        for (int i = 0; i < 9; i++) op.k[i] = p.ik[i];
        return launch(op);
    }
    if (p.kind == LIN_FIXED) {
        FixedOp<AVG> op;
        for (int i = 0; i < 9; i++) op.k[i] = p.fk[i];
        op.q = p.q;
        return launch(op);
    }
    FloatOp<AVG> op;
    for (int i = 0; i < 9; i++) op.k[i] = kernel[i];
    return launch(op);
}

}  // namespace

extern "C" {

int iak_linear(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
               const float *kernel, int kW, int kH, int mode, void *stream) {
    int rc = check_image_args(in, out, W, H, B, in_stride, out_stride);
    if (rc) return rc;
    if (!kernel || kW < 1 || kH < 1) return IAK_E_ARG;
    if (mode != IAK_LINEAR_CLAMP && mode != IAK_LINEAR_AVERAGE) return IAK_E_ARG;
    if ((long long)kW * kH > IAK_MAX_TAPS) return IAK_E_TOO_LARGE;
    const bool avg = (mode == IAK_LINEAR_AVERAGE);
    if (kW == 3 && kH == 3 && prefer8() && fast8_ok(in, out, W, H, in_stride, out_stride)) {
        LinPlan p = plan_linear3(kernel);
        if (!avg && p.kind != LIN_INT && plan_swar(kernel, p)) {
            Lin8Swar op;
            op.sa = p.sa; op.sb = p.sb; op.sc = p.sc;
            return launch8(in, out, W, H, B, in_stride, out_stride, op, stream);
        }
        auto go = [&](const auto &op) { return launch8(in, out, W, H, B, in_stride, out_stride, op, stream); };
        return avg ? run_linear3<Lin8Int, Lin8Fixed, Lin8Float, true>(p, kernel, go)
                   : run_linear3<Lin8Int, Lin8Fixed, Lin8Float, false>(p, kernel, go);
    }
    if (kW == 3 && kH == 3 && fast3_ok(in, out, W, in_stride, out_stride)) {
        const LinPlan p = plan_linear3(kernel);
        auto go = [&](const auto &op) { return launch3(in, out, W, H, B, in_stride, out_stride, op, stream); };
        return avg ? run_linear3<Linear3Int, Linear3Fixed, Linear3Float, true>(p, kernel, go)
                   : run_linear3<Linear3Int, Linear3Fixed, Linear3Float, false>(p, kernel, go);
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
    if (kW == 3 && kH == 3 && ones && prefer8() && fast8_ok(in, out, W, H, in_stride, out_stride))
        return launch8(in, out, W, H, B, in_stride, out_stride, Median8(), stream);
    if (kW == 3 && kH == 3 && ones && fast3_ok(in, out, W, in_stride, out_stride))
        return launch3(in, out, W, H, B, in_stride, out_stride, Median3(), stream);
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
    if (prefer8() && fast8_ok(in, out, W, H, in_stride, out_stride))
        return launch8(in, out, W, H, B, in_stride, out_stride, Sobel8(), stream);
    if (fast3_ok(in, out, W, in_stride, out_stride))
        return launch3(in, out, W, H, B, in_stride, out_stride, Sobel3(), stream);
    hipLaunchKernelGGL((generic_edge_kernel<G_SOBEL>), generic_grid(W, H, B), dim3(256), 0, as_stream(stream), in, out, W,
                       H, in_stride, out_stride);
    IAK_LAUNCH_CHECK();
    return 0;
}

int iak_compass(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                void *stream) {
    int rc = check_image_args(in, out, W, H, B, in_stride, out_stride);
    if (rc) return rc;
    if (prefer8() && fast8_ok(in, out, W, H, in_stride, out_stride))
        return launch8(in, out, W, H, B, in_stride, out_stride, Compass8(), stream);
    if (fast3_ok(in, out, W, in_stride, out_stride))
        return launch3(in, out, W, H, B, in_stride, out_stride, Compass3(), stream);
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
    if (kW == 3 && kH == 3) {
        uint32_t mask = 0;
        for (int i = 0; i < 9; i++)
            if (se[i] != 0) mask |= 1u << i;
        if (prefer8() && fast8_ok(in, out, W, H, in_stride, out_stride)) {
            if (erode) {
                Erode8 op;
                op.mask = mask;
                return launch8(in, out, W, H, B, in_stride, out_stride, op, stream);
            }
            Dilate8 op;
            op.mask = mask;
            return launch8(in, out, W, H, B, in_stride, out_stride, op, stream);
        }
        if (fast3_ok(in, out, W, in_stride, out_stride)) {
            if (erode) {
                Erode3 op;
                op.mask = mask;
                return launch3(in, out, W, H, B, in_stride, out_stride, op, stream);
            }
            Dilate3 op;
            op.mask = mask;
            return launch3(in, out, W, H, B, in_stride, out_stride, op, stream);
        }
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
