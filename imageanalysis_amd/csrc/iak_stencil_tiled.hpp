// Neighbourhoods larger than 3x3: an LDS-tiled linear / average filter and a sliding-histogram
// median.  Both work on the flat view of the image (a tap one past the right edge is the next
// row's first pixel, taps outside [0, W*H) read 0), exactly like the one-pixel-per-lane
// kernels they take over from.
#pragma once
#include "iak_box.hpp"
#include "iak_common.hpp"
#include "iak_stencil_ops.hpp"

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

}  // namespace iak
#include "iak_linear_rows.hpp"
namespace iak {

// ---------------------------------------------------------------------------------------
// Linear / average filter, KW x kH taps (KW odd, compile time; kH odd, run time).
// A workgroup of 256 lanes produces a 64 x 32 tile of outputs from a (32+kH-1) x (64+KW-1)
// tile of the image staged in LDS; a lane owns 8 adjacent outputs of one row and walks the
// taps in the reference's order (rows outer, columns inner), adding each product to a running
// sum that is truncated toward zero after every tap (imageAnalysis.cu:517-524: `int sum`
// receives `sum + uchar * float`).  The sum is kept as a float holding an integer: with
// sum(|k|) * 255 < 2^24 every intermediate is exact, so trunc(float) equals the reference's
// float -> int -> float round trip.  Per tap and output: multiply, add, truncate.
// ---------------------------------------------------------------------------------------
#define TL_TW 64
#define TL_TH 32
// UNIFORM: all taps equal (a box filter).  The product pixel * tap then depends on the pixel
// only, so the tile is staged as those products (floats) and a tap costs an add and a
// truncation.  (Packed two-float multiplies and adds were measured: slower than the plain
// ones here.)
template <int KW, bool AVG, bool UNIFORM>
__global__ __launch_bounds__(256) void tiled_linear_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int W,
                                                           int H, long long in_stride, long long out_stride, int kH,
                                                           int divisor, int tiles_x, Taps taps) {
    constexpr int COLS = TL_TW + KW - 1;
    // bytes: an odd number of dwords per row; floats: a multiple of 4 plus 4 (16-byte rows, skewed banks)
    constexpr int PITCH = UNIFORM ? ((COLS + 3) / 4) * 4 + 4 : ((COLS + 3) / 4) * 4 + ((((COLS + 3) / 4) & 1) ? 0 : 4);
    extern __shared__ __attribute__((aligned(16))) uint8_t tile[];
    float *tilef = reinterpret_cast<float *>(tile);
    const long long N = (long long)W * H;
    const uint8_t *src = in + (long long)blockIdx.y * in_stride;
    uint8_t *dst = out + (long long)blockIdx.y * out_stride;
    const int a = (kH - 1) / 2, s = (KW - 1) / 2;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int x0 = tx * TL_TW, y0 = ty * TL_TH;
    const int rows = TL_TH + kH - 1;
    const float k0 = taps.f[0];
    for (int e = threadIdx.x; e < rows * COLS; e += 256) {
        const int r = e / COLS, c = e - r * COLS;
        const long long idx = (long long)(y0 - a + r) * W + (x0 - s + c);
        const uint32_t v = tap_at(src, N, idx);
        if (UNIFORM) tilef[r * PITCH + c] = (float)v * k0;
        else tile[r * PITCH + c] = (uint8_t)v;
    }
    __syncthreads();
    const int lx = threadIdx.x & 7, ly = threadIdx.x >> 3;
    const int row = y0 + ly, col0 = x0 + lx * 8;
    if (row >= H || col0 >= W) return;
    float sum[8];
#pragma unroll
    for (int i = 0; i < 8; i++) sum[i] = 0.0f;
    constexpr int NPX = 8 + KW - 1;
    constexpr int NW = (NPX + 3) / 4;
    for (int r = 0; r < kH; r++) {
        float px[NW * 4];
        if (UNIFORM) {
            const float4 *rp = reinterpret_cast<const float4 *>(tilef + (ly + r) * PITCH + lx * 8);
#pragma unroll
            for (int w = 0; w < NW; w++) {
                const float4 v = rp[w];
                px[4 * w + 0] = v.x; px[4 * w + 1] = v.y; px[4 * w + 2] = v.z; px[4 * w + 3] = v.w;
            }
        } else {
            const uint32_t *rp = reinterpret_cast<const uint32_t *>(tile + (ly + r) * PITCH + lx * 8);
#pragma unroll
            for (int w = 0; w < NW; w++) {
                const uint32_t v = rp[w];
                px[4 * w + 0] = (float)(v & 0xffu);
                px[4 * w + 1] = (float)((v >> 8) & 0xffu);
                px[4 * w + 2] = (float)((v >> 16) & 0xffu);
                px[4 * w + 3] = (float)(v >> 24);
            }
        }
#pragma unroll
        for (int j = 0; j < KW; j++) {
            const float k = taps.f[r * KW + j];
#pragma unroll
            for (int i = 0; i < 8; i++) sum[i] = __builtin_truncf(UNIFORM ? sum[i] + px[i + j] : sum[i] + px[i + j] * k);
        }
    }
    uint32_t o[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int v = (int)sum[i];
        uint32_t r8;
        if (AVG) r8 = (uint32_t)(v / divisor) & 0xffu;  // imageAnalysis.cu:487: no clamp, wraps
        else r8 = (uint32_t)(v > 255 ? 255 : (v < 0 ? 0 : v));
        o[i] = is_border(row, col0 + i, W, H, a, s) ? 0u : r8;
    }
    uint8_t *q = dst + (long long)row * W + col0;
    if (col0 + 8 <= W) {
        st8_unaligned(q, o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24), o[4] | (o[5] << 8) | (o[6] << 16) | (o[7] << 24));
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (col0 + i < W) q[i] = (uint8_t)o[i];
    }
}

template <int KW>
static size_t tiled_linear_lds(int kH, bool uniform) {
    constexpr int COLS = TL_TW + KW - 1;
    const int pitch = uniform ? ((COLS + 3) / 4) * 4 + 4 : ((COLS + 3) / 4) * 4 + ((((COLS + 3) / 4) & 1) ? 0 : 4);
    // the last row's word reads may run a few elements past its pixels
    return ((size_t)(TL_TH + kH - 1) * pitch + 16) * (uniform ? sizeof(float) : 1);
}

template <int KW>
static int launch_tiled_linear_kw(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride,
                                  long long out_stride, int kH, int divisor, bool avg, bool uniform, const Taps &t,
                                  hipStream_t s) {
    const int tiles_x = (W + TL_TW - 1) / TL_TW, tiles_y = (H + TL_TH - 1) / TL_TH;
    const dim3 grid((unsigned)(tiles_x * tiles_y), B);
    const size_t lds = tiled_linear_lds<KW>(kH, uniform);
#define TL_LAUNCH(A, U)                                                                                              \
    hipLaunchKernelGGL((tiled_linear_kernel<KW, A, U>), grid, dim3(256), lds, s, in, out, W, H, in_stride, out_stride, kH, \
                       divisor, tiles_x, t)
    if (avg) {
        if (uniform) TL_LAUNCH(true, true); else TL_LAUNCH(true, false);
    } else {
        if (uniform) TL_LAUNCH(false, true); else TL_LAUNCH(false, false);
    }
#undef TL_LAUNCH
    IAK_LAUNCH_CHECK();
    return 0;
}

// kWe, kHe: the odd window actually visited.  Returns IAK_E_UNSUPPORTED when this path does not
// apply (the caller then uses the one-pixel-per-lane kernel).
static int launch_tiled_linear(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride,
                               long long out_stride, int kWe, int kHe, int divisor, bool avg, const Taps &t, hipStream_t s) {
    if (kWe > 25 || kHe > 25 || (long long)W * H >= (1LL << 31)) return IAK_E_UNSUPPORTED;
    double mag = 0;
    for (int i = 0; i < kWe * kHe; i++) {
        if (!(t.f[i] == t.f[i]) || t.f[i] > 1e30f || t.f[i] < -1e30f) return IAK_E_UNSUPPORTED;
        mag += t.f[i] < 0 ? -(double)t.f[i] : (double)t.f[i];
    }
    if (mag * 255.0 >= 16777216.0) return IAK_E_UNSUPPORTED;  // sums must stay exact in float
    bool uniform = true;
    for (int i = 1; i < kWe * kHe; i++) uniform = uniform && t.f[i] == t.f[0];
    if (uniform) {  // box filters whose per-tap truncation provably commutes with the sum: integer box sums
        const int rc = launch_box_sum(in, out, W, H, B, in_stride, out_stride, kWe, kHe, t.f[0], divisor, avg, s);
        if (rc != IAK_E_UNSUPPORTED) return rc;
    }
    {  // windows of up to 7 rows x 9 columns: in registers
        const int rc = launch_linear_rows(in, out, W, H, B, in_stride, out_stride, kWe, kHe, divisor, avg, t, s);
        if (rc != IAK_E_UNSUPPORTED) return rc;
    }
#define TL_CASE(K) case K: return launch_tiled_linear_kw<K>(in, out, W, H, B, in_stride, out_stride, kHe, divisor, avg, uniform, t, s);
    switch (kWe) {
        TL_CASE(1) TL_CASE(3) TL_CASE(5) TL_CASE(7) TL_CASE(9) TL_CASE(11) TL_CASE(13) TL_CASE(15) TL_CASE(17)
        TL_CASE(19) TL_CASE(21) TL_CASE(23) TL_CASE(25)
    default: return IAK_E_UNSUPPORTED;
    }
#undef TL_CASE
}

// ---------------------------------------------------------------------------------------
// Erosion / dilation with structuring elements above 3x3 (odd window up to 25x25) on bit masks.
// A workgroup of 256 lanes makes a 64 x 32 tile of outputs.  Staging turns the
// (32+kH-1) x (64+kW-1) input tile straight into one bit per pixel: a wave takes a tile row,
// its lanes the columns, and a ballot of "this pixel counts" is the row's mask (128 bits per row
// in LDS).  A lane then owns 8 adjacent outputs of a row: per row of the structuring element it
// cuts its 8+kW-1 bits out of the row mask and shifts them once per set element -- one AND
// (erosion) or OR (dilation) covers all 8 outputs.
//   erosion (imageAnalysis.cu:421-451): out = pixel if every element's tap is non-zero, else 0;
//     border pixels 0.  A pixel counts if it is non-zero.
//   dilation (imageAnalysis.cu:381-404, a scatter from every non-border foreground pixel, the
//     write race resolved as "255 wins"): out = 255 iff some element reaches the pixel from such
//     a source, i.e. a gather through the mirrored element.  A pixel counts if it is a source:
//     foreground and not on ITS border, judged at its true row / column on the flat view.
// Taps come through a buffer descriptor over exactly the image: offsets outside [0, N),
// negative ones included as unsigned, read 0 (clamped in a register so that the compiler cannot
// split them into a negative register part and a positive immediate).
// ---------------------------------------------------------------------------------------
struct SeRows {
    uint32_t m[25];  // bit j of m[r]: element (r, j) is set (dilation: already mirrored)
};

template <bool ERODE>
__global__ __launch_bounds__(256) void tiled_morph_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int W,
                                                          int H, long long in_stride, long long out_stride, int kW, int kH,
                                                          int tiles_x, SeRows se) {
    __shared__ uint32_t brow[(TL_TH + 24) * 4];  // [tile row][4 words]: bit c = tile column c counts
    const int Ni = W * H;  // the launcher keeps W * (H + 32) below 2^31
    const uint8_t *src = in + (long long)blockIdx.y * in_stride;
    uint8_t *dst = out + (long long)blockIdx.y * out_stride;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, Ni, 0x00020000);
    const int a = (kH - 1) / 2, s = (kW - 1) / 2;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int x0 = tx * TL_TW, y0 = ty * TL_TH;
    const int rows = TL_TH + kH - 1, cols = TL_TW + kW - 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // dilation: where tile column c lies on the flat view -- how many rows further down (negative:
    // up) and whether that column is off the border; the same for every tile row
    int wraps[2] = {0, 0};
    bool col_ok[2] = {true, true};
    if (!ERODE) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int xc = x0 - s + lane + 64 * k;
            int q = xc / W, ucol = xc - q * W;
            if (ucol < 0) { ucol += W; q--; }
            wraps[k] = q;
            col_ok[k] = ucol >= s && ucol <= W - s;
        }
    }
    {
        // all of a wave's loads (up to 14 rows: the tile has at most 56) are issued before the first
        // is used, so their latencies overlap
        uint32_t v[14][2];
#pragma unroll
        for (int u = 0; u < 14; u++) {
            const int r = wave + 4 * u;
            const int base = (y0 - a + r) * W + (x0 - s);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int c = lane + 64 * k;
                const uint32_t o = (r < rows && c < cols) ? min((uint32_t)(base + c), (uint32_t)Ni) : (uint32_t)Ni;
                v[u][k] = __builtin_amdgcn_raw_buffer_load_b8(rsrc, (int)o, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < 14; u++) {
            const int r = wave + 4 * u;
            if (r < rows) {  // uniform over the wave
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    bool counts = v[u][k] != 0;  // columns past the tile were loaded as 0
                    if (!ERODE) {  // a source is not on ITS border: its true row is the tile row's plus the wraps
                        const int urow = y0 - a + r + wraps[k];
                        counts = counts && col_ok[k] && urow >= a && urow <= H - a;
                    }
                    const unsigned long long m = __ballot(counts);
                    if (lane == 0) {
                        brow[r * 4 + 2 * k] = (uint32_t)m;
                        brow[r * 4 + 2 * k + 1] = (uint32_t)(m >> 32);
                    }
                }
            }
        }
    }
    __syncthreads();
    const int lx = threadIdx.x & 7, ly = threadIdx.x >> 3;
    const int row = y0 + ly, col0 = x0 + lx * 8;
    if (row >= H || col0 >= W) return;
    const int wi = (lx * 8) >> 5, sh = (lx * 8) & 31;
    uint32_t acc = ERODE ? 0xffu : 0u;
    for (int r = 0; r < kH; r++) {
        // the 32 row bits from tile column 8 * lx on: the lane's 8 + kW - 1 columns and more
        const uint32_t bits = __builtin_amdgcn_alignbit(brow[(ly + r) * 4 + wi + 1], brow[(ly + r) * 4 + wi], (uint32_t)sh);
        uint32_t m = se.m[r];
        for (int j = 0; m != 0; j++, m >>= 1)
            if (m & 1u) acc = ERODE ? (acc & (bits >> j)) : (acc | (bits >> j));
    }
    acc &= 0xffu;
    uint32_t o[2];
    if (ERODE) {
        // border: rows < a or > H - a, columns < s or > W - s (imageAnalysis.cu:429)
        if (row < a || row > H - a) acc = 0;
        const int lo = s - col0, hi = W - s - col0;  // output i survives iff lo <= i <= hi
        if (lo > 0) acc &= lo >= 8 ? 0u : (0xffu << lo);
        if (hi < 7) acc &= hi < 0 ? 0u : ((1u << (hi + 1)) - 1u);
        // the pixels themselves: 8 bytes at the output's own position (byte by byte at the image's end)
        const int off = row * W + col0;
        uint32_t c0 = 0, c1 = 0;
        if (off + 8 <= Ni) {
            c0 = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0);
            c1 = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off + 4, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                c0 |= (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rsrc, min(off + i, Ni), 0, 0) << (8 * i);
                c1 |= (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rsrc, min(off + 4 + i, Ni), 0, 0) << (8 * i);
            }
        }
        o[0] = c0 & spread4(acc & 0xfu);
        o[1] = c1 & spread4(acc >> 4);
    } else {
        o[0] = spread4(acc & 0xfu);
        o[1] = spread4(acc >> 4);
    }
    uint8_t *q = dst + (long long)row * W + col0;
    if (col0 + 8 <= W) {
        st8_unaligned(q, o[0], o[1]);
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (col0 + i < W) q[i] = (uint8_t)(o[i >> 2] >> (8 * (i & 3)));
    }
}

// se: the (2a+1) x (2s+1) elements actually visited, row-major.  IAK_E_UNSUPPORTED when this
// path does not apply.
static int launch_tiled_morph(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride,
                              long long out_stride, const int *se, int kWe, int kHe, bool erode, hipStream_t s) {
    if (kWe > 25 || kHe > 25 || (long long)W * ((long long)H + 32) >= (1LL << 31)) return IAK_E_UNSUPPORTED;
    SeRows rows;
    for (int r = 0; r < 25; r++) rows.m[r] = 0;
    for (int r = 0; r < kHe; r++)
        for (int j = 0; j < kWe; j++) {
            const int n = erode ? r * kWe + j : (kHe - 1 - r) * kWe + (kWe - 1 - j);  // dilation gathers through the mirror image
            if (se[n] != 0) rows.m[r] |= 1u << j;
        }
    const int tiles_x = (W + TL_TW - 1) / TL_TW, tiles_y = (H + TL_TH - 1) / TL_TH;
    const dim3 grid((unsigned)(tiles_x * tiles_y), B);
    if (erode)
        hipLaunchKernelGGL((tiled_morph_kernel<true>), grid, dim3(256), 0, s, in, out, W, H, in_stride, out_stride, kWe, kHe,
                           tiles_x, rows);
    else
        hipLaunchKernelGGL((tiled_morph_kernel<false>), grid, dim3(256), 0, s, in, out, W, H, in_stride, out_stride, kWe, kHe,
                           tiles_x, rows);
    IAK_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------
// Median, all weights 1, kW x kH <= 255 taps: column histograms (after Perreault and Hebert's
// constant-time median).  One lane per image column, one wave per workgroup, strips of
// CM_STRIP rows.  A lane keeps the histogram of ITS column's kH rows in LDS -- 256 byte
// counters packed four to a word plus 16 block counters (16 values each) -- so moving down a
// row costs one pixel out and one in per lane, whatever kW is.  The window of output column x
// is columns x-s .. x+s: its block counters and, once the block holding the sought position
// is known, its 16 fine counters are the byte-wise sums of the neighbouring lanes' words, read
// straight from LDS (word w of lane l sits at w * CM_LANES + l, so lane l +- k is an immediate
// offset and every access of a wave instruction goes to bank l % 32: no conflicts, whatever
// the pixel values are).  The outer s lanes on either side only serve as neighbours, so
// adjacent workgroups overlap by 2s columns.
// The result is the smallest v with #(window <= v) > taps/2, i.e. sorted[taps/2]
// (imageAnalysis.cu:641-680 with a race-free private window).
// ---------------------------------------------------------------------------------------
#define CM_LANES 64
#define CM_STRIP 64
// Counters are bytes, four to a word, in REVERSED order inside the word (value v of a fine word
// sits in byte 3 - (v & 3), block m = v >> 4 in byte 3 - (m & 3) of block word m >> 2): the
// one-hot increment is then a single byte-align of a constant by the low two bits of its
// operand.  LDS add/sub instructions without a return value are fire-and-forget.
// NIB: the fine counters are 4 bits wide (a column holds at most 15 pixels), eight to a word in
// natural order: half the table, twice the waves per CU.
template <bool ADD, bool NIB>
__device__ __forceinline__ void cm_update(uint32_t *fine, uint32_t *blocks, uint32_t v) {
    const uint32_t f = NIB ? (1u << ((v & 7u) << 2)) : __builtin_amdgcn_alignbyte(0u, 0x01000000u, v);  // 1 << 8 * (3 - (v & 3))
    const uint32_t m = __builtin_amdgcn_alignbyte(0u, 0x01000000u, v >> 4);  // 1 << 8 * (3 - ((v >> 4) & 3))
    uint32_t *fw = fine + (NIB ? (v >> 3) : (v >> 2)) * CM_LANES;
    uint32_t *bw = blocks + (v >> 6) * CM_LANES;
    if (ADD) {
        atomicAdd(fw, f);
        atomicAdd(bw, m);
    } else {
        atomicSub(fw, f);
        atomicSub(bw, m);
    }
}
// the counters of a word in ascending order of value
__device__ __forceinline__ uint32_t hm_ascending(uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x00010203u); }
// byte-wise helpers on four packed counters (totals never exceed 255, so no byte overflows)
__device__ __forceinline__ uint32_t hm_prefix4(uint32_t x) {  // inclusive prefix sums, byte by byte
    // two shift-and-adds; written as asm because the compiler folds them into one multiplication by
    // 0x01010101, and a 32-bit integer multiply issues at a quarter of the rate
    uint32_t y, p;
    asm("v_lshl_add_u32 %0, %1, 8, %1" : "=v"(y) : "v"(x));
    asm("v_lshl_add_u32 %0, %1, 16, %1" : "=v"(p) : "v"(y));
    return p;
}
__device__ __forceinline__ uint32_t hm_sum4(uint32_t x) { return __builtin_amdgcn_sad_u8(x, 0u, 0u); }
// how many of the first three bytes of p are below need
__device__ __forceinline__ uint32_t hm_below3(uint32_t p, uint32_t need) {
    return ((p & 0xffu) < need ? 1u : 0u) + (((p >> 8) & 0xffu) < need ? 1u : 0u) + (((p >> 16) & 0xffu) < need ? 1u : 0u);
}
// x[i] for i in [0, 4), as two levels of selects on values already in registers
__device__ __forceinline__ uint32_t hm_pick(uint32_t i, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
    const bool b0 = (i & 1u) != 0, b1 = (i & 2u) != 0;
    const uint32_t lo = b0 ? x1 : x0, hi = b0 ? x3 : x2;
    return b1 ? hi : lo;
}
// 16 byte counters packed in w[0..3], ascending; `need` (>= 1) is how many more pixels, counted
// from the first counter, the sought position lies: returns the index of the counter that
// holds it, and in `rest` how far into that counter it lies (need minus everything before the
// counter).  The caller guarantees that the 16 counters together hold at least `need`.
__device__ __forceinline__ uint32_t hm_find16(const uint32_t w[4], uint32_t need, uint32_t &rest) {
    // totals through word 0, 1, 2 (the byte-sum instruction adds onto its third operand)
    const uint32_t t0 = __builtin_amdgcn_sad_u8(w[0], 0u, 0u), t1 = __builtin_amdgcn_sad_u8(w[1], 0u, t0),
                   t2 = __builtin_amdgcn_sad_u8(w[2], 0u, t1);
    // the totals ascend, so the three comparisons select directly (no index arithmetic in between)
    const bool c0 = t0 < need, c1 = t1 < need, c2 = t2 < need;
    const uint32_t word = c2 ? w[3] : (c1 ? w[2] : (c0 ? w[1] : w[0]));
    const uint32_t base = c2 ? t2 : (c1 ? t1 : (c0 ? t0 : 0u));
    const uint32_t k4 = c2 ? 12u : (c1 ? 8u : (c0 ? 4u : 0u));
    const uint32_t need1 = need - base;
    const uint32_t p = hm_prefix4(word);
    const uint32_t p0 = p & 0xffu, p1 = (p >> 8) & 0xffu, p2 = (p >> 16) & 0xffu;
    const bool d0 = p0 < need1, d1 = p1 < need1, d2 = p2 < need1;
    rest = need1 - (d2 ? p2 : (d1 ? p1 : (d0 ? p0 : 0u)));  // minus the prefix through the counter before
    return k4 + (d2 ? 3u : (d1 ? 2u : (d0 ? 1u : 0u)));
}

// One tap through the image's buffer descriptor: flat offsets outside [0, N) read 0.  The
// offset is clamped to N in a register first so that the compiler cannot split it into a
// negative register part and a positive immediate: the range check is done on the unwrapped
// sum, and such a pair would read 0 for a tap that is inside the image.
__device__ __forceinline__ uint32_t hm_tap(__amdgpu_buffer_rsrc_t rsrc, int offset, int N) {
    const uint32_t o = min((uint32_t)offset, (uint32_t)N);
    return __builtin_amdgcn_raw_buffer_load_b8(rsrc, (int)o, 0, 0);
}

// MODE 0: byte counters; 1: 4-bit fine counters (kH <= 15); 2: the same with kH <= 7, where the
// counters of two columns can be added before the nibbles are taken apart (7 + 7 < 16).
template <int KW, int MODE>
__global__ __launch_bounds__(CM_LANES) void column_hist_median_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                                      int W, int H, long long in_stride, long long out_stride,
                                                                      int kH, int strips_x) {
    constexpr bool NIB = MODE != 0;
    constexpr int FINE_WORDS = NIB ? 32 : 64;
    // 16-byte aligned: pairs of columns are read as 8-byte LDS loads
    __shared__ __attribute__((aligned(16))) uint32_t fine_all[FINE_WORDS * CM_LANES];   // [fine word][lane]
    __shared__ __attribute__((aligned(16))) uint32_t block_all[4 * CM_LANES];   // [block word][lane]
    constexpr int s = (KW - 1) / 2;
    constexpr int OUT_W = CM_LANES - 2 * s;  // columns a workgroup produces
    const int lane = threadIdx.x;
    uint32_t *fine = fine_all + lane, *blocks = block_all + lane;
    const int N = W * H;  // the launcher keeps W * (H + CM_STRIP + 32) below 2^31
    const uint8_t *src = in + (long long)blockIdx.y * in_stride;
    uint8_t *dst = out + (long long)blockIdx.y * out_stride;
    // Taps come through a buffer descriptor over exactly the image: a flat offset outside
    // [0, N) -- negative ones included, as unsigned -- reads 0 without a branch.  A column index
    // below 0 or from W up is a pixel of the neighbouring row on the flat view (imageAnalysis.cu:657).
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, N, 0x00020000);
    const int a = (kH - 1) / 2;
    const int sy = blockIdx.x / strips_x, sx = blockIdx.x - sy * strips_x;
    const int col = sx * OUT_W - s + lane;  // this lane's column
    const int y0 = sy * CM_STRIP;
    const int y1 = (y0 + CM_STRIP < H) ? y0 + CM_STRIP : H;
    const bool emits = lane >= s && lane < CM_LANES - s && col < W;  // this lane writes output pixels
    const bool live = emits && !(col < s || col > W - s);            // border columns produce zeros
#pragma unroll 8
    for (int w = 0; w < FINE_WORDS; w++) fine[w * CM_LANES] = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) blocks[w * CM_LANES] = 0;
    const uint32_t need = (uint32_t)(KW * kH) / 2 + 1;  // position of sorted[taps / 2], counted from 1
    // the column's rows y0-a .. y0+a
    {
        int off = (y0 - a) * W + col;
        for (int i = 0; i < kH; i++, off += W) cm_update<true, NIB>(fine, blocks, hm_tap(rsrc, off, N));
    }
    // All lanes are one wave and the LDS executes a wave's instructions in the order they were
    // issued, so a lane sees its neighbours' updates without a barrier as long as the compiler
    // keeps the order of the LDS accesses too: that is all cm_order() asks for.
    auto cm_order = []() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // The window's counters: byte-wise sums of word w of the kW columns lane-s .. lane+s.  They are
    // read as (kW-1)/2 aligned pairs of columns (8-byte LDS reads move twice the data of 4-byte ones
    // per cycle, and an aligned pair of columns never straddles a bank pair) and one single column:
    // the window's first column when that one is odd, its last one otherwise.
    constexpr int NP = (KW - 1) / 2;
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    struct Window {
        u32x2_t pair[4][NP > 0 ? NP : 1];
        uint32_t one[4];
    };
    const int first_col = min(max(lane - s, 0), CM_LANES - KW);  // lanes that emit nothing read a valid window too
    const int pair0 = first_col + (first_col & 1);
    const int single = (first_col & 1) ? first_col : first_col + KW - 1;
    auto window_words = [&](const uint32_t *base, Window &r) {
#pragma unroll
        for (int w = 0; w < 4; w++) {
#pragma unroll
            for (int k = 0; k < NP; k++) r.pair[w][k] = *reinterpret_cast<const u32x2_t *>(base + w * CM_LANES + pair0 + 2 * k);
            r.one[w] = base[w * CM_LANES + single];
        }
    };
    auto window_words2 = [&](const uint32_t *base, Window &r) {  // two word rows only
#pragma unroll
        for (int w = 0; w < 2; w++) {
#pragma unroll
            for (int k = 0; k < NP; k++) r.pair[w][k] = *reinterpret_cast<const u32x2_t *>(base + w * CM_LANES + pair0 + 2 * k);
            r.one[w] = base[w * CM_LANES + single];
        }
    };
    // nibble counters: even and odd nibbles are added up apart, as bytes, and interleaved again --
    // out[0..3] = the counters of values 0-3, 4-7, 8-11, 12-15 of the block, ascending
    auto sum_nibble_words = [&](const Window &r, uint32_t (&out)[4]) {
#pragma unroll
        for (int w = 0; w < 2; w++) {
            uint32_t even = r.one[w] & 0x0f0f0f0fu, odd = (r.one[w] >> 4) & 0x0f0f0f0fu;
#pragma unroll
            for (int k = 0; k < NP; k++) {
                const uint32_t x = r.pair[w][k].x, y = r.pair[w][k].y;
                if (MODE == 2) {
                    const uint32_t xy = x + y;
                    even += xy & 0x0f0f0f0fu;
                    odd += (xy >> 4) & 0x0f0f0f0fu;
                } else {
                    even += (x & 0x0f0f0f0fu) + (y & 0x0f0f0f0fu);
                    odd += ((x >> 4) & 0x0f0f0f0fu) + ((y >> 4) & 0x0f0f0f0fu);
                }
            }
            out[2 * w] = __builtin_amdgcn_perm(odd, even, 0x05010400u);      // values 0, 1, 2, 3
            out[2 * w + 1] = __builtin_amdgcn_perm(odd, even, 0x07030602u);  // values 4, 5, 6, 7
        }
    };
    auto sum_words = [&](const Window &r, uint32_t (&out)[4]) {
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint32_t sum = r.one[w];
#pragma unroll
            for (int k = 0; k < NP; k++) sum += r.pair[w][k].x + r.pair[w][k].y;
            out[w] = hm_ascending(sum);
        }
    };
    int top = (y0 - a) * W + col, bot = (y0 + a + 1) * W + col;  // the pixel that leaves / enters at the first slide
    uint32_t t0 = hm_tap(rsrc, top, N), b0 = hm_tap(rsrc, bot, N);
    uint8_t *q = dst + (long long)y0 * W + col;
    // The loop is software-pipelined: the block counters of row y+1 are requested right behind the
    // updates that make them, and travel while the fine counters of row y are searched.
    // No branches inside the loop (the compiler would wait for every outstanding LDS access at
    // each of them): lanes that emit nothing read a valid window all the same and drop the result,
    // and the slide behind the strip's last row is done although nobody uses it.
    const uint32_t *nb_blocks = block_all;
    const uint32_t *nb_fine = fine_all;
    Window braw, fraw;
    cm_order();
    window_words(nb_blocks, braw);
    for (int y = y0; y < y1; y++, q += W) {
        // the next slide's pixels are fetched now and used a step later
        top += W;
        bot += W;
        const uint32_t t1 = hm_tap(rsrc, top, N), b1 = hm_tap(rsrc, bot, N);
        uint32_t mw[4], need2;
        sum_words(braw, mw);
        const uint32_t blk = hm_find16(mw, need, need2);
        // the 16 fine counters of that block: four words of bytes, or two of nibbles
        if (NIB) window_words2(nb_fine + blk * (2 * CM_LANES), fraw);
        else window_words(nb_fine + blk * (4 * CM_LANES), fraw);
        cm_order();
        cm_update<false, NIB>(fine, blocks, t0);
        cm_update<true, NIB>(fine, blocks, b0);
        cm_order();
        window_words(nb_blocks, braw);  // row y+1's block counters
        uint32_t fw[4], need3;
        if (NIB) sum_nibble_words(fraw, fw);
        else sum_words(fraw, fw);
        const uint32_t o = blk * 16 + hm_find16(fw, need2, need3);
        const bool row_live = live && !(y < a || y > H - a);
        if (emits) *q = (uint8_t)(row_live ? o : 0u);
        t0 = t1;
        b0 = b1;
    }
}

static int launch_column_hist_median(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride,
                                     long long out_stride, int kW, int kH, hipStream_t s) {
    if (kW * kH > 255 || kW > 25 || (long long)W * ((long long)H + CM_STRIP + 32) >= (1LL << 31)) return IAK_E_UNSUPPORTED;
    const int mode = kH <= 7 ? 2 : (kH <= 15 ? 1 : 0);  // a column's counters fit four bits up to 15 rows
    const int out_w = CM_LANES - (kW - 1);
    const int strips_x = (W + out_w - 1) / out_w, strips_y = (H + CM_STRIP - 1) / CM_STRIP;
    const dim3 grid((unsigned)(strips_x * strips_y), B);
#define CM_CASE(K)                                                                                                       \
    case K:                                                                                                              \
        if (mode == 2)                                                                                                   \
            hipLaunchKernelGGL((column_hist_median_kernel<K, 2>), grid, dim3(CM_LANES), 0, s, in, out, W, H, in_stride,         \
                               out_stride, kH, strips_x);                                                                \
        else if (mode == 1)                                                                                              \
            hipLaunchKernelGGL((column_hist_median_kernel<K, 1>), grid, dim3(CM_LANES), 0, s, in, out, W, H, in_stride,         \
                               out_stride, kH, strips_x);                                                                \
        else                                                                                                             \
            hipLaunchKernelGGL((column_hist_median_kernel<K, 0>), grid, dim3(CM_LANES), 0, s, in, out, W, H, in_stride,         \
                               out_stride, kH, strips_x);                                                                \
        break;
    switch (kW) {
        CM_CASE(1) CM_CASE(3) CM_CASE(5) CM_CASE(7) CM_CASE(9) CM_CASE(11) CM_CASE(13) CM_CASE(15) CM_CASE(17)
        CM_CASE(19) CM_CASE(21) CM_CASE(23) CM_CASE(25)
    default: return IAK_E_UNSUPPORTED;
    }
#undef CM_CASE
    IAK_LAUNCH_CHECK();
    return 0;
}

}  // namespace iak
