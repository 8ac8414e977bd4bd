// Box filters (every tap the same positive weight w) as integer box sums.
//
// The reference adds pixel * w to an int after every tap (imageAnalysis.cu:517-524), i.e.
// sum = (int)((float)sum + p) with p = (float)pixel * w.  With sum a small integer and frac(p)
// not within two float steps (at the sums' magnitude) of 1, the float sum cannot round up to
// the next integer, so the truncation gives exactly sum + floor(p): the
// filter is the plain integer sum of floor(pixel * w) over the window, and integer sums can be
// taken in any order -- rows first, then columns, each as a sliding sum.  The host checks the
// condition for all 256 pixel values before this path is taken; anything else goes through the
// tap-by-tap kernel.
//
// A workgroup of 256 lanes makes a 64 x 64 tile of outputs: (A) the (64+kH-1) x (64+kW-1) input
// tile, on the flat view of the image (taps outside it read 0), is staged in LDS as 16-bit
// floor(pixel * w); (B) sliding vertical sums, two columns per lane as one packed word;
// (C) sliding horizontal sums, 16 outputs per lane, clamp or divide, border rule, store.
#pragma once
#include "iak_common.hpp"
#include "iak_stencil_ops.hpp"

namespace iak {

#define BX_T 64                      // tile edge
#define BX_IN (BX_T + 24)            // input tile edge at most (windows up to 25)
#define BX_PITCH (BX_IN + 2)         // 16-bit elements per LDS row: 45 dwords, odd

// true if the box path is exact for weight w over `taps` taps.  With every partial sum plus tap
// below 2^e the float grid there is ulp = 2^(e-24) or finer; a product whose fraction leaves
// at least 2 ulp to the next integer (or is 0) can never be rounded up to it.
inline bool box_sum_exact(float w, int taps) {
    if (!(w > 0.0f) || !(w < 1e30f)) return false;
    int ipmax = 0;
    for (int px = 0; px < 256; px++) {
        const float p = (float)px * w;
        if (!(p < 65536.0f)) return false;
        if ((int)p > ipmax) ipmax = (int)p;
    }
    const long long bound = (long long)ipmax * taps + ipmax + 2;  // above every sum + tap
    if ((long long)ipmax * taps >= 65536) return false;           // sums are kept in 16 bits
    double ulp = 1.0 / 16777216.0;                                // 2^(e-24) with 2^e >= bound
    for (long long m = 1; m < bound; m <<= 1) ulp *= 2.0;
    for (int px = 0; px < 256; px++) {
        const float p = (float)px * w;
        const double fp = (double)p - (double)(int)p;
        if (fp != 0.0 && 1.0 - fp < 2.0 * ulp) return false;
    }
    return true;
}

template <bool AVG>
__global__ __launch_bounds__(256) void box_sum_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int W, int H,
                                                      long long in_stride, long long out_stride, int kW, int kH, float w,
                                                      int divisor, int tiles_x) {
    __shared__ __attribute__((aligned(16))) uint16_t t1[BX_IN * BX_PITCH];  // floor(pixel * w)
    __shared__ __attribute__((aligned(16))) uint16_t t2[BX_T * BX_PITCH];   // vertical sums
    const int N = W * H;  // the launcher keeps W * (H + 32) below 2^31
    const uint8_t *src = in + (long long)blockIdx.y * in_stride;
    uint8_t *dst = out + (long long)blockIdx.y * out_stride;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, N, 0x00020000);
    const int a = (kH - 1) / 2, s = (kW - 1) / 2;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int x0 = tx * BX_T, y0 = ty * BX_T;
    const int IW = BX_T + kW - 1, IH = BX_T + kH - 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (A) a wave per row, lanes along it (two passes cover up to 88 columns); eight rows' loads are
    // issued before the first is used, so that their latencies overlap
    for (int rb = wave; rb < IH; rb += 32) {
        uint32_t v[8][2];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int r = rb + 4 * u;
            const int base = (y0 - a + r) * W + (x0 - s);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int c = lane + 64 * k;
                // offsets outside [0, N), negative ones included (as unsigned), read 0; clamped in a
                // register so that the compiler cannot split them into register + immediate.  Rows
                // and columns past the tile are not loaded (offset N: out of range).
                const uint32_t o = (r < IH && c < IW) ? min((uint32_t)(base + c), (uint32_t)N) : (uint32_t)N;
                v[u][k] = __builtin_amdgcn_raw_buffer_load_b8(rsrc, (int)o, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int r = rb + 4 * u;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int c = lane + 64 * k;
                if (r < IH && c < IW) t1[r * BX_PITCH + c] = (uint16_t)(uint32_t)((float)v[u][k] * w);
            }
        }
    }
    __syncthreads();
    // (B) column pairs x four segments of 16 output rows: sliding sums of kH rows.  Two 16-bit
    // sums travel in one word; a carry from the low half into the high one during an add is
    // taken back by the subtraction that follows, since every finished sum fits 16 bits.
    {
        const int pairs = (IW + 1) >> 1;
        const int pr = threadIdx.x % 48, seg = threadIdx.x / 48;  // 48 pairs cover 96 columns; segments 0..4 (4 unused)
        if (pr < pairs && seg < 4) {
            const uint32_t *col = reinterpret_cast<const uint32_t *>(t1) + pr;  // BX_PITCH / 2 words per row
            uint32_t *dstc = reinterpret_cast<uint32_t *>(t2) + pr;
            const int r0 = seg * 16;
            uint32_t acc = 0;
            for (int i = 0; i < kH; i++) acc += col[(r0 + i) * (BX_PITCH / 2)];
            dstc[r0 * (BX_PITCH / 2)] = acc;
#pragma unroll 4
            for (int y = 1; y < 16; y++) {
                acc += col[(r0 + y + kH - 1) * (BX_PITCH / 2)];
                acc -= col[(r0 + y - 1) * (BX_PITCH / 2)];
                dstc[(r0 + y) * (BX_PITCH / 2)] = acc;
            }
        }
    }
    __syncthreads();
    // (C) 64 rows x 4 chunks of 16 outputs: sliding sums of kW columns
    const int y = threadIdx.x >> 2, q = threadIdx.x & 3;
    const int row = y0 + y, col0 = x0 + 16 * q;
    if (row >= H || col0 >= W) return;
    const uint16_t *rowp = t2 + y * BX_PITCH + 16 * q;
    uint32_t acc = 0;
    for (int j = 0; j < kW; j++) acc += rowp[j];
    uint32_t o[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (i) acc += (uint32_t)rowp[i + kW - 1] - (uint32_t)rowp[i - 1];
        const uint32_t sum = acc & 0xffffu;
        uint32_t r8;
        if (AVG) r8 = (sum / (uint32_t)divisor) & 0xffu;  // imageAnalysis.cu:487: no clamp, wraps
        else r8 = sum > 255u ? 255u : sum;
        const int col = col0 + i;
        const bool border = row < a || row > H - a || col < s || col > W - s;
        o[i] = border ? 0u : r8;
    }
    uint8_t *qd = dst + (long long)row * W + col0;
    if (col0 + 16 <= W) {
        st8_unaligned(qd, o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24), o[4] | (o[5] << 8) | (o[6] << 16) | (o[7] << 24));
        st8_unaligned(qd + 8, o[8] | (o[9] << 8) | (o[10] << 16) | (o[11] << 24),
                      o[12] | (o[13] << 8) | (o[14] << 16) | (o[15] << 24));
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (col0 + i < W) qd[i] = (uint8_t)o[i];
    }
}

// kW, kH: the odd window actually visited (at most 25 x 25); w its one weight; divisor: for the
// average filter.  IAK_E_UNSUPPORTED when the path does not apply.
static int launch_box_sum(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                          int kW, int kH, float w, int divisor, bool avg, hipStream_t s) {
    if (kW > 25 || kH > 25 || kW < 1 || kH < 1 || (long long)W * ((long long)H + 32) >= (1LL << 31)) return IAK_E_UNSUPPORTED;
    if (avg && divisor <= 0) return IAK_E_UNSUPPORTED;
    if (!box_sum_exact(w, kW * kH)) return IAK_E_UNSUPPORTED;
    const int tiles_x = (W + BX_T - 1) / BX_T, tiles_y = (H + BX_T - 1) / BX_T;
    const dim3 grid((unsigned)(tiles_x * tiles_y), B);
    if (avg)
        hipLaunchKernelGGL(box_sum_kernel<true>, grid, dim3(256), 0, s, in, out, W, H, in_stride, out_stride, kW, kH, w, divisor,
                           tiles_x);
    else
        hipLaunchKernelGGL(box_sum_kernel<false>, grid, dim3(256), 0, s, in, out, W, H, in_stride, out_stride, kW, kH, w, divisor,
                           tiles_x);
    IAK_LAUNCH_CHECK();
    return 0;
}

}  // namespace iak
