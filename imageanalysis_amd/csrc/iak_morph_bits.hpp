// Runs of 3x3 operations on a two-valued image as ONE pass, 32 pixels per register.
//
// Erosion and dilation only ask "is the pixel non-zero" of their input (imageAnalysis.cu:381-467); once a
// threshold or a dilation has been passed every pixel is 0 or 255, and a run of such operations can be carried
// from stage to stage as one bit per pixel (iak_morph_chain.hpp does that on 8-pixel chunks).  Here a lane owns a
// 32-pixel word of the image and walks DOWN a strip of rows: an input row becomes one register of bits, its
// neighbours' words come from the adjacent lanes of the wave, a tap one column to the side is a funnel shift, and
// every stage keeps the last three rows of its input, so that when input row r arrives stage s puts out its row
// r - s.  A 3x3 operation is then a handful of three-input logic instructions for 32 pixels, and the pass is bound
// by the one read and one write of the image.
//
// Stages, all on the flat view of the image (the pixel right of a row's last one is the next row's first, pixels
// outside the image are 0), exactly as the stand-alone operators:
//   erosion : out = in AND (every set element's tap is set); 0 on the border (row < 1, col < 1)
//   dilation: out = OR over set elements of the source at the mirrored offset; sources are the set pixels that
//             are not on the border; nothing outside the image
//   Sobel   : only "is the result non-zero", for a Sobel whose result feeds a dilation or erosion: on a 0 / 255
//             image each clamped gradient (imageAnalysis.cu:559-587) is 0 or 255 -- it is positive iff the
//             weighted count of its positive column / row (1, 2, 1) exceeds that of its negative one -- and the
//             magnitude is 0, 255 or 360 -> 104: non-zero iff one of the two is positive.  0 on the border.
// The output is 0 / 255 (the run must hold a threshold in front or a dilation, and a dilation behind any Sobel).
#pragma once
#include <type_traits>
#include "iak_bitmask.hpp"
#include "iak_bitrows.hpp"
#include "iak_chain.hpp"
#include "iak_median_bits.hpp"
#include "iak_stencil_ops.hpp"

namespace iak {

enum { MBS_ERODE = 0, MBS_DILATE = 1, MBS_SOBEL = 2 };
#define MBS_MAX_OPS 3
#ifndef MBS_AHEAD
#define MBS_AHEAD 3  // rows of loads in flight per lane
#endif
struct MorphBitsDesc {
    int n;
    uint32_t mask[MBS_MAX_OPS];  // bit e: structuring element e (row-major) is non-zero
    uint8_t kind[MBS_MAX_OPS];
    BitRows g;
};

typedef uint32_t mbs_u32x2 __attribute__((ext_vector_type(2)));

// a row of bits with its two sideways views: l bit i = pixel i - 1, r bit i = pixel i + 1
struct MbsRow {
    uint32_t l, c, r;
};

__device__ __forceinline__ uint32_t mbs_erode(uint32_t mask, const MbsRow &up, const MbsRow &mid, const MbsRow &dn) {
    const uint32_t tap[9] = {up.l, up.c, up.r, mid.l, mid.c, mid.r, dn.l, dn.c, dn.r};
    uint32_t ok = mid.c;  // a zero pixel stays zero
#pragma unroll
    for (int e = 0; e < 9; e++) ok = mb_op3<0xd0>(ok, tap[e], 0u - ((mask >> e) & 1u));  // ok & (tap | ~element)
    return ok;
}
__device__ __forceinline__ uint32_t mbs_dilate(uint32_t mask, const MbsRow &up, const MbsRow &mid, const MbsRow &dn) {
    // element (ei, ej) takes the source at row + 1 - ei, column + 1 - ej
    const uint32_t src[9] = {dn.r, dn.c, dn.l, mid.r, mid.c, mid.l, up.r, up.c, up.l};
    uint32_t hit = 0u;
#pragma unroll
    for (int e = 0; e < 9; e++) hit = mb_op3<0xf8>(hit, src[e], 0u - ((mask >> e) & 1u));  // hit | (source & element)
    return hit;
}
// x + 2 y + z > x' + 2 y' + z' on one-bit inputs
__device__ __forceinline__ uint32_t mbs_weighted_gt(uint32_t x, uint32_t y, uint32_t z, uint32_t x2, uint32_t y2, uint32_t z2) {
    const uint32_t a0 = x ^ z, a1 = mb_op3<0x78>(y, x, z), a2 = mb_op3<0x80>(y, x, z);
    const uint32_t b0 = x2 ^ z2, b1 = mb_op3<0x78>(y2, x2, z2), b2 = mb_op3<0x80>(y2, x2, z2);
    uint32_t gt = a0 & ~b0;
    gt = mb_op3<0xb2>(a1, b1, gt);  // (a & ~b) | (~(a ^ b) & gt)
    gt = mb_op3<0xb2>(a2, b2, gt);
    return gt;
}
__device__ __forceinline__ uint32_t mbs_sobel_nonzero(const MbsRow &up, const MbsRow &mid, const MbsRow &dn) {
    return mbs_weighted_gt(up.r, mid.r, dn.r, up.l, mid.l, dn.l) | mbs_weighted_gt(dn.l, dn.c, dn.r, up.l, up.c, up.r);
}

// THR: the run works on the image thresholded first (pixel > t ? 255 : 0, t >= 0), t read off slot 0 of the
// image's tables (laid out [B][CHAIN_MAX][256] by the table prologue: 0 up to t, 255 above).  HIST: the
// output's histogram is its number of set pixels, added to bin 255 (bin 0 is never counted,
// imageAnalysis.cu:130-152).  W % 32 == 0.
template <int NOPS, bool THR, bool HIST>
__global__ __launch_bounds__(256) void morph_bits_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int W, int H,
                                                         long long in_stride, long long out_stride, MorphBitsDesc d,
                                                         uint32_t *__restrict__ hist, const uint16_t *__restrict__ tables) {
    // eight bits -> eight bytes of 0x00 / 0xff, for the output
    __shared__ __attribute__((aligned(8))) uint32_t bytes_of[256][2];
    bytes_of[threadIdx.x][0] = spread4(threadIdx.x & 0xfu);
    bytes_of[threadIdx.x][1] = spread4(threadIdx.x >> 4);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int image, strip, wj;
    bool owns;
    if (!d.g.locate(blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane, &image, &strip, &wj, &owns)) return;
    const int N = W * H;  // the launcher keeps W * (H + 64) below 2^31
    const uint8_t *src = in + (long long)image * in_stride;
    uint8_t *dst = out + (long long)image * out_stride;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, N, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, N, 0x00020000);
    uint32_t kw = 0;  // 255 - t in every byte
    if (THR) {
        const uint16_t *tb = tables + (long long)image * CHAIN_MAX * 256;
        int above = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) above += __builtin_popcountll(__ballot((tb[64 * q + lane] & 0xffu) != 0));
        kw = (uint32_t)above * 0x01010101u;
    }
    // a neighbour lane's word may belong to the row before (word -1) or after (word `words`): its true place
    const int words = d.g.words;
    const int row_adj = wj < 0 ? -1 : (wj >= words ? 1 : 0);
    const int col0 = 32 * (wj - row_adj * words);
    const uint32_t not_col0 = col0 == 0 ? 0xfffffffeu : 0xffffffffu;

    const int ys = min(strip, d.g.strips) * d.g.rows;  // lanes without a strip read below the image: zeros
    const int y_end = min(ys + d.g.rows, H);
    // Rows 1 .. H - 1 only, for every lane of the wave and every row any stage looks at: the border rule is the
    // column-0 bit alone and no row needs a test.  (Lanes without a strip are not safe; their waves take the
    // general path.)
    const bool lane_safe = ys - NOPS - 1 >= 1 && ys + d.g.rows + NOPS + 1 <= H - 1;
    const bool safe = __ballot(lane_safe) == ~0ull;
    MbsRow win[NOPS][2];  // per stage: the two rows of its input before the newest
#pragma unroll
    for (int s = 0; s < NOPS; s++) win[s][0] = win[s][1] = MbsRow{0u, 0u, 0u};
    uint32_t n255 = 0;
    const int steps = d.g.rows + 2 * NOPS;
    int r = ys - NOPS;
    int off_in = r * W + 32 * wj;   // flat offset of the lane's word in row r (negative: in front of the image)
    int off_out = (r - NOPS) * W + 32 * wj;
    auto load = [&](int off, u32x4 &v0, u32x4 &v1) {
        const uint32_t o = min((uint32_t)off, (uint32_t)N);  // negative offsets: out of range as unsigned
        v0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)o, 0, 0);
        v1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)min(o + 16u, (uint32_t)N), 0, 0);
    };
    // MBS_AHEAD rows are in flight per lane: a strip is tall (few rows are read twice) and the waves are few, so
    // each has to keep several rows' loads going
    // (The run-in issues what a step issues, loads then two stores -- dropped ones, offset N -- and the steps are
    // padded to a multiple of MBS_AHEAD: the compiler counts outstanding memory operations in order and takes the
    // safest count where paths join, so only a loop whose entry looks like its steady state waits for the
    // oldest row alone instead of for nearly everything.)
    u32x4 va[MBS_AHEAD], vb[MBS_AHEAD];
#pragma unroll
    for (int j = 0; j < MBS_AHEAD; j++) {
        load(off_in + j * W, va[j], vb[j]);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, rs_out, N, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, rs_out, N + 16, 0, 0);
    }
    off_in += (MBS_AHEAD - 1) * W;
    auto run = [&](auto safe_tag) {
        constexpr bool SAFE = decltype(safe_tag)::value;
        auto interior = [&](int y) {  // not row 0 / column 0, inside the image
            if (SAFE) return not_col0;
            const int t = y + row_adj;
            return (t >= 1 && t < H) ? not_col0 : 0u;
        };
        auto inside = [&](int y) {
            if (SAFE) return 0xffffffffu;
            const int t = y + row_adj;
            return (t >= 0 && t < H) ? 0xffffffffu : 0u;
        };
        auto step = [&](u32x4 &v0, u32x4 &v1) {
            uint32_t cur = THR ? (above16(v0[0], v0[1], v0[2], v0[3], kw) | (above16(v1[0], v1[1], v1[2], v1[3], kw) << 16))
                               : (nonzero16(v0[0], v0[1], v0[2], v0[3]) | (nonzero16(v1[0], v1[1], v1[2], v1[3]) << 16));
            off_in += W;
            load(off_in, v0, v1);
            int rr = r;  // the row `cur` is
#pragma unroll
            for (int s = 0; s < NOPS; s++) {
                const int kind = d.kind[s];
                if (kind == MBS_DILATE) cur &= interior(rr);  // sources
                MbsRow nw;
                nw.c = cur;
                nw.l = __builtin_amdgcn_alignbit(cur, word_from_left(cur), 31);
                nw.r = __builtin_amdgcn_alignbit(word_from_right(cur), cur, 1);
                uint32_t o;
                if (kind == MBS_DILATE) o = mbs_dilate(d.mask[s], win[s][0], win[s][1], nw) & inside(rr - 1);
                else if (kind == MBS_ERODE) o = mbs_erode(d.mask[s], win[s][0], win[s][1], nw) & interior(rr - 1);
                else o = mbs_sobel_nonzero(win[s][0], win[s][1], nw) & interior(rr - 1);
                win[s][0] = win[s][1];
                win[s][1] = nw;
                cur = o;
                rr--;
            }
            // Both stores are issued whether or not there is a row to write (offset N: dropped by the range
            // check): with a fixed number of memory operations behind the rows in flight, the wait for the
            // oldest row's loads does not have to wait for the stores as well.  A store instruction writes whole
            // 64-byte lines: the four lanes of a quad first write the words of its lanes 0 and 1, 16 bytes each,
            // then those of lanes 2 and 3 (bits and offset of the source lane come by quad permutes; lanes that
            // wrote their own 32 bytes as two 16-byte halves would leave every line half written per instruction).
            const int y = r - NOPS;
            const bool emit = owns && y >= ys && y < y_end;
            const int oo = emit ? off_out : N;
            const uint32_t half = (lane & 1) * 16;
            const uint32_t bits01 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cur, 0x50, 0xf, 0xf, true) >> half;  // quad_perm 0,0,1,1
            const uint32_t bits23 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cur, 0xfa, 0xf, 0xf, true) >> half;  // quad_perm 2,2,3,3
            const int oo01 = __builtin_amdgcn_update_dpp(0, oo, 0x50, 0xf, 0xf, true) + (int)half;
            const int oo23 = __builtin_amdgcn_update_dpp(0, oo, 0xfa, 0xf, 0xf, true) + (int)half;
            const mbs_u32x2 b0 = *reinterpret_cast<const mbs_u32x2 *>(bytes_of[bits01 & 0xffu]);
            const mbs_u32x2 b1 = *reinterpret_cast<const mbs_u32x2 *>(bytes_of[(bits01 >> 8) & 0xffu]);
            const mbs_u32x2 b2 = *reinterpret_cast<const mbs_u32x2 *>(bytes_of[bits23 & 0xffu]);
            const mbs_u32x2 b3 = *reinterpret_cast<const mbs_u32x2 *>(bytes_of[(bits23 >> 8) & 0xffu]);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{b0[0], b0[1], b1[0], b1[1]}, rs_out, oo01, 0, IAK_NT_STENCIL_ST ? 2 : 0);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{b2[0], b2[1], b3[0], b3[1]}, rs_out, oo23, 0, IAK_NT_STENCIL_ST ? 2 : 0);
            if (HIST) n255 += emit ? (uint32_t)__builtin_popcount(cur) : 0u;
            off_out += W;
            r++;
        };
        for (int k = 0; k < steps; k += MBS_AHEAD) {  // up to MBS_AHEAD - 1 steps too many: they write nothing
#pragma unroll
            for (int j = 0; j < MBS_AHEAD; j++) step(va[j], vb[j]);
        }
    };
    if (safe) run(std::true_type{});
    else run(std::false_type{});
    if (HIST) {
        for (int o = 32; o > 0; o >>= 1) n255 += __shfl_down(n255, o);
        if (lane == 0 && n255) atomicAdd(hist + (long long)image * 256 + 255, n255);
    }
}

// Does the run qualify: two-valued output (a threshold in front or a dilation in the run), a dilation behind
// every Sobel, a two-valued input for every Sobel; images whose rows are whole 32-pixel words.
static bool morph_bits_ok(const MorphBitsDesc &d, bool thr, const uint8_t *in, const uint8_t *out, int W, int H,
                          long long in_stride, long long out_stride) {
    if (d.n < 1 || d.n > MBS_MAX_OPS || W % 32 != 0 || (long long)W * ((long long)H + 64) >= (1LL << 31)) return false;
    if (reinterpret_cast<uintptr_t>(in) % 4 != 0 || reinterpret_cast<uintptr_t>(out) % 4 != 0 || in_stride % 4 != 0 || out_stride % 4 != 0)
        return false;
    bool two_valued = thr, any_dilate = false;
    for (int k = 0; k < d.n; k++) {
        if (d.kind[k] == MBS_SOBEL) {
            if (!two_valued) return false;
            bool dilate_behind = false;
            for (int q = k + 1; q < d.n; q++) dilate_behind = dilate_behind || d.kind[q] == MBS_DILATE;
            if (!dilate_behind) return false;
        }
        if (d.kind[k] == MBS_DILATE) two_valued = any_dilate = true;
    }
    return thr || any_dilate;
}

// tables: null, or the per-image tables whose slot 0 is the threshold in front of the run
static int launch_morph_bits(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                             MorphBitsDesc d, uint32_t *hist, const uint16_t *tables, hipStream_t s, bool hist_cleared) {
    if (!morph_bits_ok(d, tables != nullptr, in, out, W, H, in_stride, out_stride)) return IAK_E_UNSUPPORTED;
    if (hist && !hist_cleared) IAK_HIP(hipMemsetAsync(hist, 0, sizeof(uint32_t) * 256 * (size_t)B, s));
    // Strips: each reads 2 n rows more than it writes, and the pass runs at the rate the memory system moves its
    // bytes once there are waves enough to fill the chip (eight per SIMD): as tall as that allows.  On the
    // 500 x 768x568 batch: 16 rows 103 us, 32 rows 108 us, 64 rows (2.4 waves per SIMD) 119 us.
    int rows = 64;
    while (rows > 8 && (long long)BitRows::waves_per_image(W, H, rows) * B < 8192) rows /= 2;
    if (const char *e = getenv("IAK_MBS_ROWS")) rows = atoi(e) > 0 && atoi(e) <= 256 ? atoi(e) : rows;  // for measurements
    d.g = BitRows::make(W, H, B, rows);
    const dim3 grid(d.g.grid_blocks());
#define MBS_LAUNCH(NOPS, THR, HIST)                                                                                      \
    hipLaunchKernelGGL((morph_bits_kernel<NOPS, THR, HIST>), grid, dim3(256), 0, s, in, out, W, H, in_stride, out_stride, d, \
                       hist, tables)
#define MBS_N(NOPS)                                                                        \
    if (tables) { if (hist) MBS_LAUNCH(NOPS, true, true); else MBS_LAUNCH(NOPS, true, false); }  \
    else { if (hist) MBS_LAUNCH(NOPS, false, true); else MBS_LAUNCH(NOPS, false, false); }
    if (d.n == 1) { MBS_N(1) } else if (d.n == 2) { MBS_N(2) } else { MBS_N(3) }
#undef MBS_N
#undef MBS_LAUNCH
    IAK_LAUNCH_CHECK();
    return 0;
}

}  // namespace iak
