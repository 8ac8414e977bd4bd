// A 3x3 operation on a thresholded image as a table look-up.
//
// When the operation's input is the output of a threshold (fixed or Otsu), every input pixel is
// 0 or 255, so the operation's result at a pixel is a function of the nine mask bits around it:
// 512 possible results.  The executor evaluates them once per pipeline with the operation's own
// kernel on an image holding all 512 neighbourhoods, and this pass then reads the UN-thresholded
// image, turns ten rows into bit masks by a byte-wise compare with the threshold, and looks each
// pixel's nine bits up: the mask image is never written and the arithmetic of the operation is
// not repeated per pixel.  For operations that zero their border (row 0, column 0) only; the
// flat view of the image as everywhere (the pixel right of a row's last one is the next row's
// first, pixels outside the image are 0).
#pragma once
#include "iak_bitmask.hpp"
#include "iak_chain.hpp"
#include "iak_stencil_ops.hpp"

namespace iak {

struct BinTable {
    uint8_t v[512];  // index bit 3 * i + j = mask pixel at row offset i - 1, column offset j - 1
};

// tables: [B][CHAIN_MAX][256] as chain_prep_kernel lays them out; slot 0 holds the threshold's
// table (0 up to the threshold, 255 above it; at least one 0).  W % 8 == 0, in 4-byte and out
// 8-byte aligned.
__global__ __launch_bounds__(256) void binary_table_stencil_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                                   int W, int H, long long in_stride, long long out_stride,
                                                                   const uint16_t *__restrict__ tables, BinTable bt) {
    __shared__ uint8_t table[512];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < 512; i += 256) table[i] = bt.v[i];
    // values above the threshold = entries of the threshold table that are not 0
    const int above = __syncthreads_count((tables[(long long)b * CHAIN_MAX * 256 + threadIdx.x] & 0xffu) != 0);
    const uint32_t kw = (uint32_t)above * 0x01010101u;  // 255 - threshold
    const uint8_t *src = in + (long long)b * in_stride;
    uint8_t *dst = out + (long long)b * out_stride;
    const int N = W * H;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, N, 0x00020000);
    const int chunks = W >> 3;
    const int items = chunks * ((H + 7) >> 3);
    for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < items; it += gridDim.x * blockDim.x) {
        const int strip = it / chunks;
        const int col0 = (it - strip * chunks) << 3;
        const int y0 = strip << 3;
        uint32_t bits[10];  // bit i of row r = mask pixel (y0 - 1 + r, col0 - 1 + i)
#pragma unroll
        for (int r = 0; r < 10; r++) bits[r] = mask_row10(rsrc, (y0 - 1 + r) * W + col0 - 4, N, kw);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int row = y0 + r;
            const uint32_t c = bits[r] | (bits[r + 1] << 10) | (bits[r + 2] << 20);
            uint32_t o[8];
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = table[nine_bits(c, j)];
            uint32_t o0 = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
            uint32_t o1 = o[4] | (o[5] << 8) | (o[6] << 16) | (o[7] << 24);
            if (row < 1) o0 = o1 = 0;            // border row
            if (col0 == 0) o0 &= 0xffffff00u;    // border column
            if (row < H) st8<IAK_NT_STENCIL_ST>(dst + (long long)row * W + col0, o0, o1);
        }
    }
}

static int launch_binary_table_stencil(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride,
                                       long long out_stride, const uint16_t *tables, const BinTable &bt, hipStream_t s) {
    if (W % 8 != 0 || (long long)W * ((long long)H + 8) >= (1LL << 30) || reinterpret_cast<uintptr_t>(in) % 4 != 0 ||
        reinterpret_cast<uintptr_t>(out) % 8 != 0 || in_stride % 4 != 0 || out_stride % 8 != 0)
        return IAK_E_UNSUPPORTED;
    const long long items = (long long)(W >> 3) * ((H + 7) / 8);
    hipLaunchKernelGGL(binary_table_stencil_kernel, dim3(blocks_x(items, 256, B, 1 << 20), B), dim3(256), 0, s, in, out, W, H,
                       in_stride, out_stride, tables, bt);
    IAK_LAUNCH_CHECK();
    return 0;
}

// The image that holds every 3x3 pattern of 0 / 255 once: pattern idx in the 3x3 block at block
// row idx / 32, block column idx % 32 of a 96 x 48 image; its result is the output at the
// block's centre (3 * (idx / 32) + 1, 3 * (idx % 32) + 1), which is never on the border.
#define BT_W 96
#define BT_H 48
inline void binary_patterns_image(uint8_t *img /* BT_W * BT_H */) {
    for (int idx = 0; idx < 512; idx++)
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                img[(3 * (idx / 32) + i) * BT_W + 3 * (idx % 32) + j] = ((idx >> (3 * i + j)) & 1) ? 255 : 0;
}
inline void binary_table_from_output(const uint8_t *outimg, BinTable *bt) {
    for (int idx = 0; idx < 512; idx++) bt->v[idx] = outimg[(3 * (idx / 32) + 1) * BT_W + 3 * (idx % 32) + 1];
}

}  // namespace iak
