// Erosion / dilation with structuring elements of up to 9 x 9 on bit masks held in registers.
//
// A lane owns 8 pixels x 8 rows of the output, as in the 8-pixel 3x3 kernels: it loads the
// 8 + 2a input rows once (one 16-byte buffer load each: columns col0-4 .. col0+11, which holds
// the +-4 columns a 9-wide element reaches), turns each into 16 bits ("this pixel counts"),
// and an output row is the AND (erosion) or OR (dilation) of shifted row masks -- one shift
// and one logic operation per set element for 8 outputs.  Elements whose rows show at most
// three different patterns (boxes, lines, crosses, diamonds, disks) take less: one horizontal
// pass per input row and pattern -- by doubling when the pattern is one run of ones -- and one
// operation per element row.  No LDS, no barrier, every input row loaded once per 8 output rows.
//   erosion (imageAnalysis.cu:421-451): out = pixel if every element's tap is non-zero, else 0;
//     border pixels (row < a, row > H-a, col < s, col > W-s) 0.  A pixel counts if it is non-zero.
//   dilation (imageAnalysis.cu:381-404, a scatter from every non-border foreground pixel, "255
//     wins"): out = 255 iff some element reaches the pixel from such a source, i.e. a gather
//     through the mirrored element (the launcher mirrors it).  A pixel counts if it is a
//     source: non-zero and not on the border at ITS true row / column on the flat view.
// Flat view of the image as everywhere: bit p of a row's mask is the pixel at flat offset
// row*W + col0-4 + p, so columns past a row's end are the next row's first ones, and pixels
// outside [0, N) read 0 through the buffer descriptor.  W % 8 == 0, images 4-byte and outputs
// 8-byte aligned.
#pragma once
#include "iak_bitmask.hpp"
#include "iak_common.hpp"
#include "iak_stencil_ops.hpp"

namespace iak {

struct SeRows9 {
    uint32_t m[9];  // bit j of m[i]: element (i, j) is set (dilation: already mirrored)
    // the rows by pattern (NPAT > 0): pattern p is pat[p], used by the rows in the bit set rows_of[p];
    // run_len[p] > 0: the pattern is run_len[p] ones from bit run_start[p] on
    uint32_t pat[3], rows_of[3];
    int run_start[3], run_len[3];
};

// AND (erosion) or OR (dilation) of the n bits from each position on, n >= 1, by doubling
template <bool ERODE>
__device__ __forceinline__ uint32_t run_of(uint32_t x, int n) {
    uint32_t acc = x;
    int width = 1;
    while (2 * width <= n) {
        acc = ERODE ? (acc & (acc >> width)) : (acc | (acc >> width));
        width *= 2;
    }
    if (width < n) acc = ERODE ? (acc & (acc >> (n - width))) : (acc | (acc >> (n - width)));
    return acc;
}

// A: half height of the element (kH = 2A + 1).  NPAT: its non-empty rows show NPAT different
// patterns (1 to 3), or 0: element by element.
template <bool ERODE, int A, int NPAT>
__global__ __launch_bounds__(256) void morph_rows_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int W,
                                                         int H, long long in_stride, long long out_stride, int s,
                                                         SeRows9 se) {
    constexpr int KH = 2 * A + 1, ROWS = 8 + 2 * A;
    const uint8_t *src = in + (long long)blockIdx.y * in_stride;
    uint8_t *dst = out + (long long)blockIdx.y * out_stride;
    const int N = W * H;  // the launcher keeps W * (H + 16) below 2^31
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, N, 0x00020000);
    const int chunks = W >> 3;
    const int items = chunks * ((H + 7) >> 3);
    const int base_shift = 4 - s;  // element column j looks at bit x + j + 4 - s for output x
    for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < items; it += gridDim.x * blockDim.x) {
        const int strip = it / chunks;
        const int col0 = (it - strip * chunks) << 3;
        const int y0 = strip << 3;
        u32x4 v[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) v[r] = load16_flat(rsrc, (y0 - A + r) * W + col0 - 4, N);
        // dilation: which of the 16 positions can be sources at all, by column -- in the row itself, and
        // for the first / last chunk of a row the four positions that belong to the row before / after
        uint32_t same = 0xffffu, prev = 0, next = 0;
        if (!ERODE) {
            // in the row itself: positions pl .. pr-1, of which columns s .. W-s
            const int pl = col0 == 0 ? 4 : 0, pr = col0 + 8 == W ? 12 : 16;
            const int lo_p = max(pl, s + 4 - col0), hi_p = min(pr - 1, W - s - col0 + 4);
            same = hi_p < lo_p ? 0u : (((2u << hi_p) - 1u) & ~((1u << lo_p) - 1u));
            if (col0 == 0) prev = (1u << (4 - s + 1 > 4 ? 4 : 4 - s + 1)) - 1u;  // columns W-4+p <= W-s
            if (col0 + 8 == W) next = 0xffffu & ~((1u << (12 + s)) - 1u);     // columns p-12 >= s
        }
        // NPAT == 0: bit p of row r = pixel (y0 - A + r, col0 - 4 + p) counts; else per pattern: bit x = the
        // pattern's taps around output x of that row all (erosion) / any (dilation) count
        uint32_t bits[NPAT ? NPAT : 1][ROWS];
        uint32_t c0[8], c1[8];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            uint32_t b = nonzero16(v[r][0], v[r][1], v[r][2], v[r][3]);
            if (!ERODE) {
                const int yy = y0 - A + r;
                const bool ok = yy >= A && yy <= H - A, okp = yy - 1 >= A && yy - 1 <= H - A, okn = yy + 1 >= A && yy + 1 <= H - A;
                b &= (ok ? same : 0u) | (okp ? prev : 0u) | (okn ? next : 0u);
            }
            if (NPAT == 0) bits[0][r] = b;
#pragma unroll
            for (int p = 0; p < NPAT; p++) {
                uint32_t h;
                if (se.run_len[p] > 0) {  // uniform over the grid
                    h = run_of<ERODE>(b >> (se.run_start[p] + base_shift), se.run_len[p]);
                } else {
                    h = ERODE ? 0xffffffffu : 0u;
                    uint32_t m = se.pat[p];
                    for (int j = 0; m != 0; j++, m >>= 1)
                        if (m & 1u) h = ERODE ? (h & (b >> (j + base_shift))) : (h | (b >> (j + base_shift)));
                }
                bits[p][r] = h;
            }
            if (ERODE && r >= A && r < A + 8) {
                c0[r - A] = v[r][1];
                c1[r - A] = v[r][2];
            }
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int row = y0 + r;
            uint32_t acc = ERODE ? 0xffu : 0u;
#pragma unroll
            for (int i = 0; i < KH; i++) {
                if (NPAT == 0) {
                    uint32_t m = se.m[i];
                    for (int j = 0; m != 0; j++, m >>= 1)
                        if (m & 1u) acc = ERODE ? (acc & (bits[0][r + i] >> (j + base_shift))) : (acc | (bits[0][r + i] >> (j + base_shift)));
                }
#pragma unroll
                for (int p = 0; p < NPAT; p++)
                    if ((se.rows_of[p] >> i) & 1u) acc = ERODE ? (acc & bits[p][r + i]) : (acc | bits[p][r + i]);
            }
            acc &= 0xffu;
            uint32_t o0, o1;
            if (ERODE) {
                if (row < A || row > H - A) acc = 0;
                const int lo = s - col0, hi = W - s - col0;  // output x survives iff lo <= x <= hi
                if (lo > 0) acc &= lo >= 8 ? 0u : (0xffu << lo);
                if (hi < 7) acc &= hi < 0 ? 0u : ((1u << (hi + 1)) - 1u);
                o0 = c0[r] & spread4(acc & 0xfu);
                o1 = c1[r] & spread4(acc >> 4);
            } else {
                o0 = spread4(acc & 0xfu);
                o1 = spread4(acc >> 4);
            }
            if (row < H) st8<IAK_NT_STENCIL_ST>(dst + (long long)row * W + col0, o0, o1);
        }
    }
}

// se: the (2a+1) x (2s+1) elements actually visited, row-major.  IAK_E_UNSUPPORTED when this
// path does not apply (the caller then takes the tiled kernel).
static int launch_morph_rows(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                             const int *se, int kWe, int kHe, bool erode, hipStream_t stream) {
    const int a = (kHe - 1) / 2, s = (kWe - 1) / 2;
    if (a > 4 || s > 4 || W % 8 != 0 || W < 8 || (long long)W * ((long long)H + 16) >= (1LL << 31)) return IAK_E_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(in) % 4 != 0 || reinterpret_cast<uintptr_t>(out) % 8 != 0 || in_stride % 4 != 0 ||
        out_stride % 8 != 0)
        return IAK_E_UNSUPPORTED;
    SeRows9 rows;
    for (int i = 0; i < 9; i++) rows.m[i] = 0;
    for (int i = 0; i < kHe; i++)
        for (int j = 0; j < kWe; j++) {
            const int n = erode ? i * kWe + j : (kHe - 1 - i) * kWe + (kWe - 1 - j);  // dilation gathers through the mirror image
            if (se[n] != 0) rows.m[i] |= 1u << j;
        }
    // the non-empty rows by pattern
    int npat = 0;
    for (int p = 0; p < 3; p++) rows.pat[p] = rows.rows_of[p] = 0, rows.run_start[p] = rows.run_len[p] = 0;
    for (int i = 0; i < kHe && npat >= 0; i++) {
        if (rows.m[i] == 0) continue;
        int p = 0;
        while (p < npat && rows.pat[p] != rows.m[i]) p++;
        if (p == npat) {
            if (npat == 3) { npat = -1; break; }
            rows.pat[npat++] = rows.m[i];
        }
        rows.rows_of[p] |= 1u << i;
    }
    if (npat < 0) npat = 0;  // more than three patterns: element by element
    for (int p = 0; p < npat; p++) {
        const uint32_t m = rows.pat[p];
        int j0 = 0;
        while (!((m >> j0) & 1u)) j0++;
        int n = 0;
        while ((m >> (j0 + n)) & 1u) n++;
        if ((m >> (j0 + n)) == 0) rows.run_start[p] = j0, rows.run_len[p] = n;  // one run of ones
    }
    // (an element without any set entry has npat == 0 and no work in the element-by-element form either)
    const long long items = (long long)(W >> 3) * ((H + 7) / 8);
    const dim3 grid(blocks_x(items, 256, B, 1 << 20), B);
#define MR_LAUNCH(E, AA, NP) \
    hipLaunchKernelGGL((morph_rows_kernel<E, AA, NP>), grid, dim3(256), 0, stream, in, out, W, H, in_stride, out_stride, s, rows)
#define MR_PAT(E, AA)                                  \
    switch (npat) {                                    \
    case 1: MR_LAUNCH(E, AA, 1); break;                \
    case 2: MR_LAUNCH(E, AA, 2); break;                \
    case 3: MR_LAUNCH(E, AA, 3); break;                \
    default: MR_LAUNCH(E, AA, 0); break;               \
    }
#define MR_CASE(AA)                                    \
    case AA:                                           \
        if (erode) { MR_PAT(true, AA) } else { MR_PAT(false, AA) } \
        break;
    switch (a) {
        MR_CASE(0) MR_CASE(1) MR_CASE(2) MR_CASE(3) MR_CASE(4)
    }
#undef MR_CASE
#undef MR_PAT
#undef MR_LAUNCH
    IAK_LAUNCH_CHECK();
    return 0;
}

}  // namespace iak
