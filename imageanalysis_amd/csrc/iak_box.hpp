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
// Weights such as fl(1/121) miss that condition for a few pixel values only: 121 * fl(1/121) lies
// one float step below 1, so that tap adds floor(p) = 0 to a sum of 0 and, the float sum then
// rounding up, 1 to any sum >= 1.  When every pixel value is either exact as above or "special" in
// exactly that way (floor(p) onto an empty sum, floor(p) + 1 onto any other; the host tries every
// pixel value against every possible sum), the filter is still an order-free box sum plus a
// count: a special tap earns its extra 1 iff an earlier tap of the window, in the reference's
// row-major tap order, has contributed at least 1.  Per window that is
//   sum f + (#special - #special before the first tap with f >= 1),   or sum f if no tap has f >= 1,
// (a special tap that is itself the first one with f >= 1 counts as "before": it lands on an
// empty sum), with #special and #(f >= 1) box sums of their own (they ride in spare bits of the same
// elements) and the walk to the first contributing tap only where a window holds both kinds.
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

// How the box path applies to weight w over a kW x kH window: BOX_NO, BOX_PLAIN, or BOX_SPECIAL
// with the (at most BOX_MAX_SPECIAL) pixel values that add one more onto a non-empty sum.  The special form keeps three counts in one 16-bit element (see box_special_kernel), which
// bounds floor(255 * w) by 3 and the window by 15 x 15.
enum { BOX_NO = 0, BOX_PLAIN = 1, BOX_SPECIAL = 2 };
#define BOX_MAX_SPECIAL 4
struct BoxSpecial {
    uint32_t value[BOX_MAX_SPECIAL];  // unused entries hold 256 (no pixel has that value)
};
inline int box_classify(float w, int kW, int kH, BoxSpecial *sp) {
    const int taps = kW * kH;
    if (box_sum_exact(w, taps)) return BOX_PLAIN;
    if (!(w > 0.0f) || !(255.0f * w < 4.0f) || kW > 15 || kH > 15) return BOX_NO;
    for (int i = 0; i < BOX_MAX_SPECIAL; i++) sp->value[i] = 256;
    const int smax = 3 * taps + taps;  // above every running sum
    int any = 0;
    for (int px = 0; px < 256; px++) {
        const float p = (float)px * w;
        const int fl = (int)p;
        int up_from = -1;  // first running sum at which the tap gives floor + 1
        bool ok = true;
        for (int sum = 0; sum <= smax && ok; sum++) {
            const int d = (int)((float)sum + p) - (sum + fl);
            if (d == 0) ok = up_from < 0;            // exact here: must not have deviated at a smaller sum
            else if (d == 1) { if (up_from < 0) up_from = sum; }
            else ok = false;
        }
        if (!ok || (up_from >= 0 && up_from != 1)) return BOX_NO;
        if (up_from == 1) {
            if (any == BOX_MAX_SPECIAL) return BOX_NO;
            sp->value[any++] = (uint32_t)px;
        }
    }
    return any ? BOX_SPECIAL : BOX_NO;
}

// (A) of both kernels: a wave per row, lanes along it (two passes cover up to 88 columns); eight
// rows' loads are issued before the first is used, so that their latencies overlap.
// With `lut` (256 entries in LDS, ready before the call) the element is lut[pixel] instead.
// Returns the OR of the elements this lane staged.
template <bool SPECIAL>
__device__ __forceinline__ uint32_t box_stage_tile(uint16_t *t1, __amdgpu_buffer_rsrc_t rsrc, int N, int W, int x0, int y0, int a,
                                                   int s, int IW, int IH, float w, const uint16_t *lut) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t seen = 0;
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
                if (r < IH && c < IW) {
                    const uint16_t e = SPECIAL ? lut[v[u][k]] : (uint16_t)(uint32_t)((float)v[u][k] * w);
                    t1[r * BX_PITCH + c] = e;
                    seen |= e;
                }
            }
        }
    }
    return seen;
}

// The same tile staged four pixels per load, for images whose rows start at multiples of 4 bytes
// (W % 4 == 0, 4-byte aligned images).  The byte-wise form above spends 40 % of the special
// kernel's time (one load, one offset clamp, one LDS store per pixel).  Here the tile starts
// SH = s rounded up to a multiple of 4 columns left of x0, so that every dword is aligned and
// wholly inside or outside the image (N % 4 == 0); tile column c of the form above is column
// c + SH - s here.  IWs: staged columns, a multiple of 4.  Eight loads per lane are issued before
// the first is used.
template <bool SPECIAL>
__device__ __forceinline__ uint32_t box_stage_tile_wide(uint16_t *t1, __amdgpu_buffer_rsrc_t rsrc, int N, int W, int x0, int y0,
                                                        int a, int SH, int IWs, int IH, float w, const uint16_t *lut) {
    const int DW = IWs >> 2;
    uint32_t seen = 0;
    const int tasks = IH * DW;
    // task t = (row t / DW, dword t % DW); a lane's tasks are 256 apart: one division, then steps
    const int step_r = 256 / DW, step_d = 256 - step_r * DW;
    int r0 = (int)threadIdx.x / DW, d0 = (int)threadIdx.x - r0 * DW;
    for (int t0 = threadIdx.x; t0 < tasks; t0 += 8 * 256) {
        uint32_t v[8];
        int r = r0, d = d0;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int f = (y0 - a + r) * W + x0 - SH + 4 * d;
            // tasks past the tile and offsets outside [0, N) (negative ones as unsigned) read 0
            const uint32_t o = t0 + 256 * u < tasks ? min((uint32_t)f, (uint32_t)N) : (uint32_t)N;
            v[u] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)o, 0, 0);
            r += step_r; d += step_d;
            if (d >= DW) { d -= DW; r++; }
        }
        r = r0; d = d0;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (t0 + 256 * u < tasks) {
                uint32_t e[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t px = (v[u] >> (8 * k)) & 0xffu;
                    e[k] = SPECIAL ? (uint32_t)lut[px] : (uint32_t)(uint16_t)(uint32_t)((float)px * w);
                }
                uint32_t *q = reinterpret_cast<uint32_t *>(t1 + r * BX_PITCH + 4 * d);  // BX_PITCH is even
                q[0] = e[0] | (e[1] << 16);
                q[1] = e[2] | (e[3] << 16);
                seen |= e[0] | e[1] | e[2] | e[3];
            }
            r += step_r; d += step_d;
            if (d >= DW) { d -= DW; r++; }
        }
        r0 = r; d0 = d;
    }
    return seen;
}

// (B) and (C) of the plain box sum on a staged tile: t1 holds the elements (IW x IH, the tile's column 0 at
// staged column sh), t2 receives the vertical sums.  keep: the bits of an element pair that belong to the value
// (the special form's elements carry flags above them, which the vertical sums drop here).
template <bool AVG>
__device__ __forceinline__ void box_plain_passes(const uint16_t *t1, uint16_t *t2, uint8_t *dst, int W, int H, int x0, int y0,
                                                 int IW, int kW, int kH, int a, int s, int sh, int divisor, uint32_t keep) {
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
            dstc[r0 * (BX_PITCH / 2)] = acc & keep;
#pragma unroll 4
            for (int y = 1; y < 16; y++) {
                acc += col[(r0 + y + kH - 1) * (BX_PITCH / 2)];
                acc -= col[(r0 + y - 1) * (BX_PITCH / 2)];
                dstc[(r0 + y) * (BX_PITCH / 2)] = acc & keep;
            }
        }
    }
    __syncthreads();
    // (C) 64 rows x 4 chunks of 16 outputs: sliding sums of kW columns
    const int y = threadIdx.x >> 2, q = threadIdx.x & 3;
    const int row = y0 + y, col0 = x0 + 16 * q;
    if (row >= H || col0 >= W) return;
    const uint16_t *rowp = t2 + y * BX_PITCH + 16 * q + sh;
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

template <bool AVG>
__global__ __launch_bounds__(256) void box_sum_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int W, int H,
                                                      long long in_stride, long long out_stride, int kW, int kH, float w,
                                                      int divisor, int tiles_x, bool wide) {
    __shared__ __attribute__((aligned(16))) uint16_t t1[BX_IN * BX_PITCH];  // floor(pixel * w)
    __shared__ __attribute__((aligned(16))) uint16_t t2[BX_T * BX_PITCH];   // vertical sums
    const int N = W * H;  // the launcher keeps W * (H + 32) below 2^31
    const uint8_t *src = in + (long long)blockIdx.y * in_stride;
    uint8_t *dst = out + (long long)blockIdx.y * out_stride;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, N, 0x00020000);
    const int a = (kH - 1) / 2, s = (kW - 1) / 2;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int x0 = tx * BX_T, y0 = ty * BX_T;
    int IW = BX_T + kW - 1;
    const int IH = BX_T + kH - 1;
    int sh = 0;  // tile column 0 (image column x0 - s) sits at staged column sh
    if (wide) {
        const int SH = (s + 3) & ~3;
        sh = SH - s;
        IW = (BX_T + SH + s + 3) & ~3;
        box_stage_tile_wide<false>(t1, rsrc, N, W, x0, y0, a, SH, IW, IH, w, nullptr);
    } else {
        box_stage_tile<false>(t1, rsrc, N, W, x0, y0, a, s, IW, IH, w, nullptr);
    }
    __syncthreads();
    box_plain_passes<AVG>(t1, t2, dst, W, H, x0, y0, IW, kW, kH, a, s, sh, divisor, 0xffffffffu);
}

// The special form, rows first.  Per window the specials that earn nothing are those in front of
// (and including) the first contributing tap in row-major order: all specials of the window rows
// above the first row that holds a contributing tap, plus, in that row, the specials up to that tap.
// (A) an element of the staged tile is floor(pixel * w) (2 bits) | special << 6 | contributing << 10,
// from a 256-entry table the workgroup makes first.  (B) per input row: sliding sums of kW (<= 15)
// columns -- floor sum, #special and #contributing stay apart in 6 + 4 + 4 bits -- and, by a
// right-to-left pass, for every column the number G of specials from it up to and including the
// next contributing tap (only used where that tap lies inside the window); stored as one word
// per (row, column): floor sum | #special << 10 | #contributing << 18 | G << 26.  (C) one lane per
// output column and 16-row segment, from the bottom row up: sliding sums of kH (<= 15) rows of those
// words (10 + 8 + 8 bits hold the window's counts; what the G fields add up to above them drops
// out of the top), and one running number, "specials that earn nothing if the window started at
// this row": restarted from G at a row with a contributing tap, grown by the row's specials
// otherwise.  No branches on the pixel data.
#define BXS_PITCH (BX_T + 1)
#ifndef BXS_TY
#define BXS_TY 64  // tile height of the special form (its width is BX_T)
#endif
#define BXS_SEG (BXS_TY / 4)
// KW is a template parameter so that the row passes unroll and their LDS reads travel together.
template <bool AVG, int KW>
__global__ __launch_bounds__(256) void box_special_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int W,
                                                          int H, long long in_stride, long long out_stride, int kH,
                                                          float w, int divisor, int tiles_x, BoxSpecial sp, bool wide) {
    constexpr int kW = KW;
    __shared__ __attribute__((aligned(16))) uint16_t t1[(BXS_TY + 14) * BX_PITCH];  // per pixel: floor | special | contributing
    __shared__ uint32_t t2[(BXS_TY + 14) * BXS_PITCH];                        // per input row and column: see above
    __shared__ uint16_t lut[256];
    {
        const uint32_t px = threadIdx.x;
        const uint32_t f = (uint32_t)((float)px * w);
        const uint32_t special = (px == sp.value[0] || px == sp.value[1] || px == sp.value[2] || px == sp.value[3]) ? 1u : 0u;
        lut[px] = (uint16_t)(f | (special << 6) | ((f >= 1u ? 1u : 0u) << 10));
    }
    __syncthreads();
    const int N = W * H;  // the launcher keeps W * (H + 32) below 2^31
    const uint8_t *src = in + (long long)blockIdx.y * in_stride;
    uint8_t *dst = out + (long long)blockIdx.y * out_stride;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, N, 0x00020000);
    const int a = (kH - 1) / 2, s = (kW - 1) / 2;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int x0 = tx * BX_T, y0 = ty * BXS_TY;
    const int IW = BX_T + kW - 1, IH = BXS_TY + kH - 1;
    int sh = 0;  // tile column 0 (image column x0 - s) sits at staged column sh
    int IWs = IW;  // staged columns
    uint32_t seen;
    if (wide) {
        const int SH = (s + 3) & ~3;
        sh = SH - s;
        IWs = (BX_T + SH + s + 3) & ~3;
        seen = box_stage_tile_wide<true>(t1, rsrc, N, W, x0, y0, a, SH, IWs, IH, w, lut);
    } else {
        seen = box_stage_tile<true>(t1, rsrc, N, W, x0, y0, a, s, IW, IH, w, lut);
    }
    // A tile that holds no special value at all is a plain box sum of the floors: the two sliding passes of
    // box_sum_kernel on the same elements (their flag bits dropped after the vertical sums, where they are
    // still apart from the value's six).  Uniform over the workgroup.
    if (!__syncthreads_or((int)(seen & 0x40u))) {
        box_plain_passes<AVG>(t1, reinterpret_cast<uint16_t *>(t2), dst, W, H, x0, y0, IWs, kW, kH, a, s, sh, divisor, 0x003f003fu);
        return;
    }
    // (B) IH rows x 8 chunks of 8 outputs
    for (int task = threadIdx.x; task < IH * 8; task += 256) {
        const int r = task >> 3, c0 = (task & 7) << 3;
        const uint16_t *rowp = t1 + r * BX_PITCH + c0 + sh;
        uint32_t el[8 + KW - 1];  // the chunk's 8 windows
#pragma unroll
        for (int c = 0; c < 8 + KW - 1; c++) el[c] = rowp[c];
        uint32_t g8[8], g = 0;
#pragma unroll
        for (int c = 7 + KW - 1; c >= 0; c--) {
            const uint32_t special = (el[c] >> 6) & 1u;
            g = (el[c] & 0x400u) ? special : g + special;
            if (c < 8) g8[c] = g;
        }
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < KW; j++) acc += el[j];
        uint32_t *d2 = t2 + r * BXS_PITCH + c0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (i) acc += el[i + KW - 1] - el[i - 1];
            d2[i] = (acc & 0x3fu) | ((acc & 0x3c0u) << 4) | ((acc & 0x3c00u) << 8) | (g8[i] << 26);
        }
    }
    __syncthreads();
    // (C) 64 columns x 4 segments of BXS_SEG rows, bottom row first
    const int x = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int col = x0 + x;
    if (col >= W) return;
    const uint32_t *colp = t2 + (seg * BXS_SEG) * BXS_PITCH + x;
    uint32_t acc = 0, before = 0;
    auto step = [&](uint32_t e) {  // one row further up
        acc += e;
        before = (e & 0x03fc0000u) ? ((e >> 26) & 0xfu) : before + ((e >> 10) & 0xffu);
    };
    // the window of the segment's last row, but for its top row, which the loop below adds
#pragma unroll 4
    for (int r = BXS_SEG - 1 + kH - 1; r > BXS_SEG - 1; r--) step(colp[r * BXS_PITCH]);
#pragma unroll
    for (int i = BXS_SEG - 1; i >= 0; i--) {
        step(colp[i * BXS_PITCH]);
        if (i < BXS_SEG - 1) acc -= colp[(i + kH) * BXS_PITCH];
        const int row = y0 + seg * BXS_SEG + i;
        if (row >= H) continue;
        uint32_t sum = acc & 0x3ffu;
        const uint32_t n_special = (acc >> 10) & 0xffu, n_big = (acc >> 18) & 0xffu;
        if (n_big != 0) sum += n_special - before;  // with a contributing tap in the window, `before` counts inside it
        uint32_t r8;
        if (AVG) r8 = (sum / (uint32_t)divisor) & 0xffu;  // imageAnalysis.cu:487: no clamp, wraps
        else r8 = sum > 255u ? 255u : sum;
        const bool border = row < a || row > H - a || col < s || col > W - s;
        dst[(long long)row * W + col] = (uint8_t)(border ? 0u : r8);
    }
}

}  // namespace iak
#include "iak_box_rows.hpp"
#include "iak_box_bits.hpp"
namespace iak {

// kW, kH: the odd window actually visited (at most 25 x 25); w its one weight; divisor: for the
// average filter.  IAK_E_UNSUPPORTED when the path does not apply.
static int launch_box_sum(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                          int kW, int kH, float w, int divisor, bool avg, hipStream_t s) {
    if (kW > 25 || kH > 25 || kW < 1 || kH < 1 || (long long)W * ((long long)H + 32) >= (1LL << 31)) return IAK_E_UNSUPPORTED;
    if (avg && divisor <= 0) return IAK_E_UNSUPPORTED;
    // the classification tries every pixel value against every sum: remembered for the last weight seen
    struct Memo { float w; int kW, kH, kind; BoxSpecial sp; };
    static thread_local Memo memo = {0.0f, 0, 0, BOX_NO, {}};
    if (!(memo.w == w && memo.kW == kW && memo.kH == kH)) {
        Memo m{w, kW, kH, BOX_NO, {}};
        m.kind = box_classify(w, kW, kH, &m.sp);
        memo = m;
    }
    if (memo.kind == BOX_NO) return IAK_E_UNSUPPORTED;
    if (memo.kind == BOX_PLAIN) {  // square windows up to 9 x 9: in registers
        const int rc = launch_box_rows(in, out, W, H, B, in_stride, out_stride, kW, kH, w, divisor, avg, s);
        if (rc != IAK_E_UNSUPPORTED) return rc;
    }
    const int tiles_x = (W + BX_T - 1) / BX_T, tiles_y = (H + BX_T - 1) / BX_T;
    const dim3 grid((unsigned)(tiles_x * tiles_y), B);
    // rows that start at multiples of 4 bytes: the tile is staged a dword at a time
    const bool wide = W % 4 == 0 && reinterpret_cast<uintptr_t>(in) % 4 == 0 && in_stride % 4 == 0;
    if (memo.kind == BOX_SPECIAL) {
        if (!avg && !getenv("IAK_NO_BITSLICE_BOX")) {  // square windows, linear filter: 32 pixels per register
            const int rc = launch_bitslice_box(in, out, W, H, B, in_stride, out_stride, kW, kH, w, memo.sp, s);
            if (rc != IAK_E_UNSUPPORTED) return rc;
        }
        const dim3 grid((unsigned)(tiles_x * ((H + BXS_TY - 1) / BXS_TY)), B);
#define BXS_CASE(K)                                                                                                       \
    case K:                                                                                                               \
        if (avg)                                                                                                          \
            hipLaunchKernelGGL((box_special_kernel<true, K>), grid, dim3(256), 0, s, in, out, W, H, in_stride, out_stride, kH, w, \
                               divisor, tiles_x, memo.sp, wide);                                                          \
        else                                                                                                              \
            hipLaunchKernelGGL((box_special_kernel<false, K>), grid, dim3(256), 0, s, in, out, W, H, in_stride, out_stride, kH, \
                               w, divisor, tiles_x, memo.sp, wide);                                                       \
        break;
        switch (kW) {
            BXS_CASE(1) BXS_CASE(3) BXS_CASE(5) BXS_CASE(7) BXS_CASE(9) BXS_CASE(11) BXS_CASE(13) BXS_CASE(15)
        default: return IAK_E_UNSUPPORTED;
        }
#undef BXS_CASE
    } else if (avg) {
        hipLaunchKernelGGL(box_sum_kernel<true>, grid, dim3(256), 0, s, in, out, W, H, in_stride, out_stride, kW, kH, w, divisor,
                           tiles_x, wide);
    } else {
        hipLaunchKernelGGL(box_sum_kernel<false>, grid, dim3(256), 0, s, in, out, W, H, in_stride, out_stride, kW, kH, w, divisor,
                           tiles_x, wide);
    }
    IAK_LAUNCH_CHECK();
    return 0;
}

}  // namespace iak
