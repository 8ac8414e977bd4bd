// Linear / average filters with arbitrary taps and windows of up to 7 rows x 9 columns, in
// registers.  Included by iak_stencil_tiled.hpp (Taps, is_border).
//
// A lane owns 8 (windows up to 5 rows) or 4 (7 rows: the sums of 2a+1 output rows are live at a
// time, and 7 x 8 of them cost too many waves) pixels x 8 rows of the output and walks
// down the 8 + 2a input rows, one
// 16-byte buffer load each (columns col0-4 .. col0+11), fetched four rows ahead.  (At 9x9 the
// register budget leaves three waves per SIMD and the kernel is no faster than the tiled one:
// 1.23 against 1.26 ms on the 218 Mpx bench batch.)  The reference adds the taps in row-major order and
// truncates the sum toward zero after every tap (imageAnalysis.cu:517-524: `int sum` receives
// `sum + uchar * float`), so the order matters: an input row is applied to the up to 2a+1
// output rows whose windows hold it, as their tap row i = input row - output row + a, columns
// left to right -- each output row therefore sees its tap rows in ascending order, as the
// reference does.  The sums are floats holding integers (the launcher checks that
// sum(|k|) * 255 < 2^24, so every intermediate is exact) and cost a multiply, an add and a
// truncation per tap and output, with no tile in LDS and no barrier: the tiled kernel this
// takes over from spent 40 % of its time staging the tile byte by byte.
// Taps outside the image read 0 through the buffer descriptor (flat view: columns past a
// row's end are the next row's first ones); border pixels are 0.  W % 8 == 0, images 4-byte
// and outputs 8-byte aligned.
#pragma once
#include "iak_bitmask.hpp"

namespace iak {

template <bool AVG, int A, int S, int PX>
__global__ __launch_bounds__(256) void linear_rows_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int W, int H,
                                                          long long in_stride, long long out_stride, int divisor, Taps taps) {
    constexpr int KW = 2 * S + 1, KH = 2 * A + 1, ROWS = 8 + 2 * A, AHEAD = ROWS < 4 ? ROWS : 4;
    const uint8_t *src = in + (long long)blockIdx.y * in_stride;
    uint8_t *dst = out + (long long)blockIdx.y * out_stride;
    const int N = W * H;  // the launcher keeps W * (H + 16) below 2^31
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, N, 0x00020000);
    const int chunks = W / PX;
    const int items = chunks * ((H + 7) >> 3);
    for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < items; it += gridDim.x * blockDim.x) {
        const int strip = it / chunks;
        const int col0 = (it - strip * chunks) * PX;
        const int y0 = strip << 3;
        const int voff = (y0 - A) * W + col0 - 4;
        u32x4 ring[AHEAD + 1];
#pragma unroll
        for (int r = 0; r < AHEAD; r++) ring[r] = load16_flat(rsrc, voff + r * W, N);
        float acc[KH][PX];
#pragma unroll
        for (int k = 0; k < KH; k++)
#pragma unroll
            for (int x = 0; x < PX; x++) acc[k][x] = 0.0f;
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            if (r + AHEAD < ROWS) ring[(r + AHEAD) % (AHEAD + 1)] = load16_flat(rsrc, voff + (r + AHEAD) * W, N);
            const u32x4 raw = ring[r % (AHEAD + 1)];
            float px[PX + 2 * S];  // columns col0 - S .. col0 + PX - 1 + S
#pragma unroll
            for (int c = 0; c < PX + 2 * S; c++) {
                const int b = 4 - S + c;  // byte of the load
                px[c] = (float)((raw[b >> 2] >> (8 * (b & 3))) & 0xffu);
            }
            // this input row is tap row i of output row ro = r - i (rows 0 .. 7 of the item)
#pragma unroll
            for (int i = KH - 1; i >= 0; i--) {
                const int ro = r - i;
                if (ro < 0 || ro > 7) continue;
#pragma unroll
                for (int j = 0; j < KW; j++) {
                    const float k = taps.f[i * KW + j];
#pragma unroll
                    for (int x = 0; x < PX; x++) acc[ro % KH][x] = __builtin_truncf(acc[ro % KH][x] + px[x + j] * k);
                }
            }
            if (r >= 2 * A) {  // output row r - 2A has seen its last tap row
                const int ro = r - 2 * A;
                const int row = y0 + ro;
                uint32_t o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int x = 0; x < PX; x++) {
                    const int v = (int)acc[ro % KH][x];
                    uint32_t r8;
                    if (AVG) r8 = (uint32_t)(v / divisor) & 0xffu;  // imageAnalysis.cu:487: no clamp, wraps
                    else r8 = (uint32_t)(v > 255 ? 255 : (v < 0 ? 0 : v));
                    o[x] = is_border(row, col0 + x, W, H, A, S) ? 0u : r8;
                    acc[ro % KH][x] = 0.0f;
                }
                if (row < H) {
                    const uint32_t lo = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
                    if (PX == 8) st8<IAK_NT_STENCIL_ST>(dst + (long long)row * W + col0, lo, o[4] | (o[5] << 8) | (o[6] << 16) | (o[7] << 24));
                    else *reinterpret_cast<uint32_t *>(dst + (long long)row * W + col0) = lo;
                }
            }
        }
    }
}

// kW x kH taps (both odd, kW <= 9, kH <= 7), already checked to keep every sum exact in float.
// IAK_E_UNSUPPORTED when this path does not apply.
static int launch_linear_rows(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                              int kW, int kH, int divisor, bool avg, const Taps &t, hipStream_t stream) {
    const int a = (kH - 1) / 2, s = (kW - 1) / 2;
    if ((kW & 1) == 0 || (kH & 1) == 0 || a > 3 || s > 4 || W % 8 != 0 || W < 8 ||
        (long long)W * ((long long)H + 16) >= (1LL << 31))
        return IAK_E_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(in) % 4 != 0 || reinterpret_cast<uintptr_t>(out) % 8 != 0 || in_stride % 4 != 0 ||
        out_stride % 8 != 0)
        return IAK_E_UNSUPPORTED;
    if (avg && divisor == 0) return IAK_E_UNSUPPORTED;
    const int px = a <= 2 ? 8 : 4;
    const long long items = (long long)(W / px) * ((H + 7) / 8);
    const dim3 grid(blocks_x(items, 256, B, 1 << 20), B);
#define LR_LAUNCH(AA, SS, PX)                                                                                                  \
    do {                                                                                                                       \
        if (avg)                                                                                                               \
            hipLaunchKernelGGL((linear_rows_kernel<true, AA, SS, PX>), grid, dim3(256), 0, stream, in, out, W, H, in_stride,   \
                               out_stride, divisor, t);                                                                        \
        else                                                                                                                   \
            hipLaunchKernelGGL((linear_rows_kernel<false, AA, SS, PX>), grid, dim3(256), 0, stream, in, out, W, H, in_stride,  \
                               out_stride, divisor, t);                                                                        \
    } while (0)
#define LR_ROW(AA, PX)                     \
    case AA:                               \
        switch (s) {                       \
        case 0: LR_LAUNCH(AA, 0, PX); break; \
        case 1: LR_LAUNCH(AA, 1, PX); break; \
        case 2: LR_LAUNCH(AA, 2, PX); break; \
        case 3: LR_LAUNCH(AA, 3, PX); break; \
        default: LR_LAUNCH(AA, 4, PX); break; \
        }                                  \
        break;
    switch (a) {
        LR_ROW(0, 8) LR_ROW(1, 8) LR_ROW(2, 8) LR_ROW(3, 4)
    }
#undef LR_ROW
#undef LR_LAUNCH
    IAK_LAUNCH_CHECK();
    return 0;
}

}  // namespace iak
