// Two or three 3x3 erosions / dilations in a row as ONE pass over the image, on bit masks.
//
// Both operations only ask "is the pixel non-zero" of their input (imageAnalysis.cu:381-467), a
// dilation's output is 0 / 255 and an erosion's output is its input where it survives, so a run
// of them can be carried from stage to stage as one bit per pixel: the image is read once, the
// intermediate images are never written, and the output bytes are rebuilt at the end (255 where
// the bit is set once a dilation has been passed, the original pixel otherwise).
//
// A lane owns 8 pixels x 8 rows of the output, as in the 8-pixel 3x3 kernel; with n stages it
// loads 8 + 2n input rows (one 16-byte buffer load each, columns col0-4 .. col0+11) and every
// stage shrinks the window by one row and one column on each side.  Everything is on the flat
// view of the image: bit i of a row window starting at flat offset g0 is pixel g0 + i, so the
// column right of the last one is the next row's first pixel, as the reference's tap indices
// have it.  Per stage, as the stand-alone operators do:
//   erosion : out = in AND (every set element's tap is set); 0 on the border (row < 1, col < 1)
//   dilation: out = OR over set elements of the source at the mirrored offset; sources are the
//             set pixels that are not on the border; nothing outside the image.
#pragma once
#include "iak_bitmask.hpp"
#include "iak_chain.hpp"
#include "iak_common.hpp"
#include "iak_hist.hpp"
#include "iak_stencil_ops.hpp"

namespace iak {

#define MC_MAX_OPS 3
struct MorphChainDesc {
    int n;                     // 1 to 3 stages (1 only behind a threshold)
    uint32_t mask[MC_MAX_OPS];   // bit e: structuring element e (row-major) is non-zero
    uint8_t dilate[MC_MAX_OPS];  // stage is a dilation (else an erosion)
};

// Which of the nbits pixels at flat offsets g0 .. g0+nbits-1 are inside the image and not on its
// border rows / columns that 3x3 operators zero (row >= 1, col >= 1).  W % 8 == 0 and windows
// start at col0 - h with col0 % 8 == 0, h <= 3, so a window holds at most the column-0 pixel of
// its own row (first chunk) and that of the next row (last chunk).
__device__ __forceinline__ uint32_t mc_interior(int g0, int nbits, int h, int col0, int W, int N) {
    uint32_t m = (1u << nbits) - 1u;
    const int t = W - g0;  // pixels before flat offset W are in row 0 (or in front of the image)
    if (t > 0) m = t >= 32 ? 0u : (m & (~0u << t));
    const int u = N - g0;  // pixels from flat offset N on are behind the image
    if (u < nbits) m = u <= 0 ? 0u : (m & ((1u << u) - 1u));
    if (col0 == 0) m &= ~(1u << h);
    if (col0 + 8 == W) m &= ~(1u << (8 + h));
    return m;
}
// the same without the border rule: inside [0, N) only
__device__ __forceinline__ uint32_t mc_inside(int g0, int nbits, int N) {
    uint32_t m = (1u << nbits) - 1u;
    if (g0 < 0) m = -g0 >= 32 ? 0u : (m & (~0u << -g0));
    const int u = N - g0;
    if (u < nbits) m = u <= 0 ? 0u : (m & ((1u << u) - 1u));
    return m;
}

// The stages on a lane's row windows.  bits[r] holds row y0-NOPS+r, bit i = pixel col0-NOPS+i;
// stage s (h = NOPS-1-s) leaves rows y0-h .. y0+7+h and columns col0-h .. col0+7+h in
// bits[0 .. 8+2h).  SAFE: every row any window touches is inside the image and not row 0, so
// the only pixels to mask are the column-0 ones of the first and the last chunk of a row.
template <int NOPS, bool SAFE>
__device__ __forceinline__ void mc_stages(uint32_t (&bits)[8 + 2 * NOPS], const MorphChainDesc &d, int y0, int col0, int W,
                                          int N) {
#pragma unroll
    for (int s = 0; s < NOPS; s++) {
        const int h = NOPS - 1 - s;  // halo of this stage's output
        const int nb = 8 + 2 * h;    // its bits per row
        const int rows_in = nb + 2;
        const uint32_t mask = d.mask[s];
        // column-0 pixels in a window of halo hh: bit hh (first chunk) and bit 8+hh (last chunk)
        const uint32_t col0_in = ~((col0 == 0 ? 1u << (h + 1) : 0u) | (col0 + 8 == W ? 1u << (9 + h) : 0u));
        const uint32_t col0_out = ~((col0 == 0 ? 1u << h : 0u) | (col0 + 8 == W ? 1u << (8 + h) : 0u)) & ((1u << nb) - 1u);
        uint32_t res[8 + 2 * (NOPS - 1)];
        if (d.dilate[s]) {
            // sources: set, inside, not on the border -- at the input window's positions
#pragma unroll
            for (int r = 0; r < rows_in; r++) {
                if (SAFE) bits[r] &= col0_in;
                else bits[r] &= mc_interior((y0 - (h + 1) + r) * W + col0 - (h + 1), nb + 2, h + 1, col0, W, N);
            }
#pragma unroll
            for (int r = 0; r < nb; r++) {
                // output row r (image row y0-h+r) sits between input rows r, r+1, r+2;
                // element (ei, ej) takes the source at row + 1 - ei, column + 1 - ej
                const uint32_t up = bits[r], mid = bits[r + 1], dn = bits[r + 2];
                uint32_t hit = 0;
                if (mask & 1u) hit |= dn >> 2;
                if (mask & 2u) hit |= dn >> 1;
                if (mask & 4u) hit |= dn;
                if (mask & 8u) hit |= mid >> 2;
                if (mask & 16u) hit |= mid >> 1;
                if (mask & 32u) hit |= mid;
                if (mask & 64u) hit |= up >> 2;
                if (mask & 128u) hit |= up >> 1;
                if (mask & 256u) hit |= up;
                res[r] = hit & (SAFE ? (1u << nb) - 1u : mc_inside((y0 - h + r) * W + col0 - h, nb, N));
            }
        } else {
#pragma unroll
            for (int r = 0; r < nb; r++) {
                const uint32_t up = bits[r], mid = bits[r + 1], dn = bits[r + 2];
                uint32_t ok = mid >> 1;  // the pixel itself: a zero pixel stays zero
                if (mask & 1u) ok &= up;
                if (mask & 2u) ok &= up >> 1;
                if (mask & 4u) ok &= up >> 2;
                if (mask & 8u) ok &= mid;
                if (mask & 16u) ok &= mid >> 1;
                if (mask & 32u) ok &= mid >> 2;
                if (mask & 64u) ok &= dn;
                if (mask & 128u) ok &= dn >> 1;
                if (mask & 256u) ok &= dn >> 2;
                res[r] = ok & (SAFE ? col0_out : mc_interior((y0 - h + r) * W + col0 - h, nb, h, col0, W, N));
            }
        }
#pragma unroll
        for (int r = 0; r < nb; r++) bits[r] = res[r];
    }
}

// HIST: 0 no histogram; 1 the output's histogram through the LDS table (runs of erosions: grey
// values); 2 the output is 0 / 255 (a dilation has been passed), so its histogram is the number
// of set bits, added to bin 255 (bin 0 is never counted, imageAnalysis.cu:130-152).
// THR: the run works on the image thresholded first (pixel > t ? 255 : 0, t >= 0): the bit of a
// pixel is "above t" instead of "non-zero", t read off slot 0 of the image's tables (laid out
// [B][CHAIN_MAX][256] by the table prologue: 0 up to t, 255 above), and the output is 0 / 255.
template <int NOPS, int HIST, bool THR>
__global__ __launch_bounds__(256) void morph_chain_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int W,
                                                          int H, long long in_stride, long long out_stride,
                                                          MorphChainDesc d, uint32_t *__restrict__ hist,
                                                          const uint16_t *__restrict__ tables) {
    __shared__ uint32_t sh[HIST == 1 ? 256 * HIST_COLS : 1];
    if (HIST == 1) hist_zero(sh);
    uint32_t n255 = 0;
    const int hcol = threadIdx.x & (HIST_COLS - 1);
    const uint8_t *src = in + (long long)blockIdx.y * in_stride;
    uint8_t *dst = out + (long long)blockIdx.y * out_stride;
    const int N = W * H;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, N, 0x00020000);
    const int chunks = W >> 3;
    const int strips = (H + 7) >> 3;
    const int items = chunks * strips;
    bool binary = THR;  // a threshold or a dilation has been passed: values are 0 / 255
#pragma unroll
    for (int s = 0; s < NOPS; s++) binary = binary || d.dilate[s];
    uint32_t kw = 0;  // 255 - t in every byte
    if (THR) kw = (uint32_t)__syncthreads_count((tables[(long long)blockIdx.y * CHAIN_MAX * 256 + threadIdx.x] & 0xffu) != 0) * 0x01010101u;
    for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < items; it += gridDim.x * blockDim.x) {
        const int strip = it / chunks;
        const int cx = it - strip * chunks;
        const int col0 = cx << 3;
        const int y0 = strip << 3;
        constexpr int R0 = 8 + 2 * NOPS;  // input rows y0-NOPS .. y0+7+NOPS
        uint32_t bits[R0];
        uint32_t keep[8][2];  // the lane's own pixels, for runs of erosions
#pragma unroll
        for (int r = 0; r < R0; r++) {
            const int row = y0 - NOPS + r;
            // rows above the image read 0, but the last chunk of such a row runs on into row 0 on the
            // flat view, and those pixels are taps of the next row's first pixels: load16_flat keeps them
            const u32x4 v = load16_flat(rsrc, row * W + col0 - 4, N);
            const uint32_t m16 = THR ? above16(v[0], v[1], v[2], v[3], kw) : nonzero16(v[0], v[1], v[2], v[3]);
            bits[r] = (m16 >> (4 - NOPS)) & ((1u << (8 + 2 * NOPS)) - 1u);  // bit i = pixel col0 - NOPS + i
            if (r >= NOPS && r < NOPS + 8) {
                keep[r - NOPS][0] = v[1];
                keep[r - NOPS][1] = v[2];
            }
        }
        // rows y0-NOPS-1 .. y0+8+NOPS all inside rows 1 .. H-1: no row or end-of-image masking
        if (y0 - NOPS - 1 >= 1 && y0 + 8 + NOPS + 1 <= H - 1) mc_stages<NOPS, true>(bits, d, y0, col0, W, N);
        else mc_stages<NOPS, false>(bits, d, y0, col0, W, N);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int row = y0 + r;
            uint32_t o0 = spread4(bits[r] & 0xfu), o1 = spread4((bits[r] >> 4) & 0xfu);
            if (!binary) {
                o0 &= keep[r][0];
                o1 &= keep[r][1];
            }
            if (row < H) {
                st8<IAK_NT_STENCIL_ST>(dst + (long long)row * W + col0, o0, o1);
                if (HIST == 1) {
                    hist_add4(sh, o0, hcol);
                    hist_add4(sh, o1, hcol);
                }
                if (HIST == 2) n255 += (uint32_t)__builtin_popcount(bits[r] & 0xffu);
            }
        }
    }
    if (HIST == 1) hist_flush(sh, hist + (long long)blockIdx.y * 256);
    if (HIST == 2) {
        for (int o = 32; o > 0; o >>= 1) n255 += __shfl_down(n255, o);
        if ((threadIdx.x & 63) == 0 && n255) atomicAdd(hist + (long long)blockIdx.y * 256 + 255, n255);
    }
}

// Does the run qualify: 2..3 operations, all 3x3 erosions / dilations; images as the aligned
// 8-pixel path wants them.
static bool morph_chain_ok(const uint8_t *in, const uint8_t *out, int W, int H, long long in_stride, long long out_stride) {
    return W % 8 == 0 && W >= 8 && (long long)W * ((long long)H + 8) < (1LL << 30) && reinterpret_cast<uintptr_t>(in) % 4 == 0 &&
           reinterpret_cast<uintptr_t>(out) % 8 == 0 && in_stride % 4 == 0 && out_stride % 8 == 0;
}

// tables: null, or the per-image tables whose slot 0 is the threshold in front of the run
static int launch_morph_chain(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                              const MorphChainDesc &d, uint32_t *hist, const uint16_t *tables, hipStream_t s,
                              bool hist_cleared = false) {
    if (d.n < (tables ? 1 : 2) || d.n > MC_MAX_OPS || !morph_chain_ok(in, out, W, H, in_stride, out_stride)) return IAK_E_UNSUPPORTED;
    const long long items = (long long)(W >> 3) * ((H + 7) / 8);
    if (hist && !hist_cleared) IAK_HIP(hipMemsetAsync(hist, 0, sizeof(uint32_t) * 256 * (size_t)B, s));
    bool binary = tables != nullptr;
    for (int k = 0; k < d.n; k++) binary = binary || d.dilate[k];
    const int hmode = !hist ? 0 : (binary ? 2 : 1);
    const dim3 grid(blocks_x(items, 256, B, hmode == 1 ? 4096 : 1 << 20), B);
#define MC_LAUNCH(NOPS, HIST, THR)                                                                                      \
    hipLaunchKernelGGL((morph_chain_kernel<NOPS, HIST, THR>), grid, dim3(256), 0, s, in, out, W, H, in_stride, out_stride, d, \
                       hist, tables)
    if (tables) {
        if (d.n == 1) { if (hmode) MC_LAUNCH(1, 2, true); else MC_LAUNCH(1, 0, true); }
        else if (d.n == 2) { if (hmode) MC_LAUNCH(2, 2, true); else MC_LAUNCH(2, 0, true); }
        else { if (hmode) MC_LAUNCH(3, 2, true); else MC_LAUNCH(3, 0, true); }
    } else if (d.n == 2) {
        if (hmode == 0) MC_LAUNCH(2, 0, false); else if (hmode == 1) MC_LAUNCH(2, 1, false); else MC_LAUNCH(2, 2, false);
    } else {
        if (hmode == 0) MC_LAUNCH(3, 0, false); else if (hmode == 1) MC_LAUNCH(3, 1, false); else MC_LAUNCH(3, 2, false);
    }
#undef MC_LAUNCH
    IAK_LAUNCH_CHECK();
    return 0;
}

}  // namespace iak
