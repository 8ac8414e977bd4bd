// Neighbourhood kernels: linear / average filter, median, Sobel, compass, erosion, dilation.
//
// Addressing model (from the reference, imageAnalysis.cu:514-521 and siblings): a pixel's
// taps are read at FLAT index i*W+j of the image, so a tap one past the right edge is the
// first pixel of the next row, and taps at or past W*H read 0.  Everything here is written
// on that flat view; (row, col) only decide the border mask.
//
// Two implementations per operation:
//   * a 3x3 path for widths that are a multiple of 8: each lane owns an 8-pixel column chunk
//     and walks down a strip of rows with a three-row window in registers (the operators
//     themselves are in iak_stencil_ops.hpp);
//   * a generic path (any width, any kW x kH up to IAK_MAX_TAPS), one pixel per lane.
#include "iak_common.hpp"
#include "iak_hist.hpp"
#include "iak_stencil_ops.hpp"
#include "iak_stencil_tiled.hpp"
#include "iak_morph_chain.hpp"
#include "iak_morph_bits.hpp"
#include "iak_morph_rows.hpp"
#include "iak_median_bits.hpp"

namespace iak {

// =======================================================================================
// Generic path
// =======================================================================================
enum { G_LINEAR = 0, G_AVERAGE = 1, G_ERODE = 2, G_DILATE = 3, G_COMPASS = 4, G_SOBEL = 5 };

// fused: each tap as one fused multiply-add (contraction mode)
__device__ __forceinline__ int accumulate(const uint8_t *img, long long N, int W, int row, int col, int a, int s,
                                          const float *k, bool fused = false) {
    // imageAnalysis.cu:517-524: `int sum; sum += uchar * float` truncates at every tap
    int sum = 0;
    int n = 0;
    for (int i = row - a; i <= row + a; i++)
        for (int j = col - s; j <= col + s; j++) {
            const float px = (float)tap_at(img, N, (long long)i * W + j);
            float acc;
            if (fused) {
                acc = __builtin_fmaf(px, k[n], (float)sum);
            } else {
                float prod = px * k[n];
                acc = (float)sum + prod;
            }
            sum = (int)acc;
            n++;
        }
    return sum;
}

template <int MODE>
__global__ __launch_bounds__(256) void generic_float_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                            int W, int H, long long in_stride, long long out_stride,
                                                            int kW, int kH, Taps taps, bool fused = false) {
    const long long N = (long long)W * H;
    const uint8_t *src = in + (long long)blockIdx.y * in_stride;
    uint8_t *dst = out + (long long)blockIdx.y * out_stride;
    const int a = (kH - 1) / 2, s = (kW - 1) / 2;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < N; t += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(t / W), col = (int)(t - (long long)row * W);
        uint32_t o = 0;
        if (!is_border(row, col, W, H, a, s)) {
            int sum = accumulate(src, N, W, row, col, a, s, taps.f, fused);
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
// 3x3 path: 8 pixels per lane, one 16-byte buffer load per row
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
// These kernels are bound by vector-instruction issue, not by bandwidth (measured: the SIMDs'
// VALU-active time equals the kernel time), so the operators count instructions: three-input
// min/med/max, 16-bit packed arithmetic, byte-wise SWAR and dot products wherever the
// arithmetic allows, and 8 pixels per lane to keep a three-row window near 30 registers.
__device__ __forceinline__ Raw16 load_raw(__amdgpu_buffer_rsrc_t rsrc, int voffset) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, 0, IAK_NT_STENCIL_LD ? 2 : 0);
    Raw16 r;
    r.d[0] = v[0]; r.d[1] = v[1]; r.d[2] = v[2]; r.d[3] = v[3];
    return r;
}

// The same 16 bytes for a flat offset of any alignment.  The range check is per dword, so a
// load that straddles either end of the image would lose the dword holding the first or the
// last one to three pixels: those few loads are redone byte by byte.
__device__ __forceinline__ Raw16 load_raw_any(__amdgpu_buffer_rsrc_t rsrc, const uint8_t *src, int N, int voffset) {
    Raw16 r = load_raw(rsrc, voffset);
    if ((voffset < 0 && voffset > -16) || (voffset < N && voffset + 16 > N)) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const long long o = (long long)voffset + 4 * i;
            r.d[i] = tap_at(src, N, o) | (tap_at(src, N, o + 1) << 8) | (tap_at(src, N, o + 2) << 16) |
                     (tap_at(src, N, o + 3) << 24);
        }
    }
    return r;
}


// Op interface (8-pixel form):
//   typedef ... Row;
//   void prepare(const Raw16&, row, col0, W, H, Row&)
//   void compute(const Row& up, const Row& mid, const Row& dn, uint32_t o[2])
//   static const bool kBorderZero
// HIST: the histogram of the OUTPUT image is accumulated on the way (hist[b][256], zeroed by the
// launcher; bin 0 left empty as calculateHistogram does), so that a consumer of that histogram
// needs no pass of its own.
// ANYW: any width from 8 up and any alignment of the images.  Rows then start at arbitrary byte
// offsets (unaligned 16-byte loads, unaligned 8-byte stores) and the last chunk of a row may be
// partial: its pixels past the row's end are the next row's first ones, which is what the
// reference's flat tap indices read there, and only its first W - col0 outputs are stored.
template <class Op, int RPL, bool HIST = false, bool ANYW = false>
__global__ __launch_bounds__(256) void stencil8_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int W,
                                                       int H, long long in_stride, long long out_stride, Op op,
                                                       uint32_t *__restrict__ hist = nullptr, uint32_t chunks_magic = 0) {
    __shared__ uint32_t sh[HIST ? 256 * HIST_COLS : 1];
    if (HIST) hist_zero(sh);
    const int hcol = threadIdx.x & (HIST_COLS - 1);
    const uint8_t *src = in + (long long)blockIdx.y * in_stride;
    uint8_t *dst = out + (long long)blockIdx.y * out_stride;
    const int N = W * H;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, N, 0x00020000);
    // the aligned form stores through a descriptor of the output image as well: 32-bit offsets, and rows below
    // the image's last one (the last strip may have some) are dropped by the range check
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, N, 0x00020000);
    const int chunks = ANYW ? (W + 7) >> 3 : W >> 3;
    const int strips = (H + RPL - 1) / RPL;
    const int items = chunks * strips;
    for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < items; it += gridDim.x * blockDim.x) {
        // chunks_magic = ceil(2^32 / chunks) where the launcher has checked that the product gives it / chunks
        const int strip = chunks_magic ? (int)__umulhi((uint32_t)it, chunks_magic) : it / chunks;
        const int cx = it - strip * chunks;
        const int col0 = cx << 3;
        const int y0 = strip * RPL;
        const int voff = (y0 - 1) * W + col0 - 4;
        Raw16 raw[RPL + 2];
        if (ANYW) {
#pragma unroll
            for (int r = 0; r < RPL + 2; r++) raw[r] = load_raw_any(rsrc, src, N, voff + r * W);
        } else {
            int vo = voff;
#pragma unroll
            for (int r = 0; r < RPL + 2; r++) {
                raw[r] = load_raw(rsrc, vo);
                vo += W;
            }
            if (y0 == 0 && col0 == 0) {
                // the very first chunk of the image: offset -4 is out of range as a whole on some
                // range-check implementations, so fetch from 0 and shift the dwords up by one
                Raw16 f = load_raw(rsrc, 0);
                raw[1].d[0] = 0; raw[1].d[1] = f.d[0]; raw[1].d[2] = f.d[1]; raw[1].d[3] = f.d[2];
            }
        }
        const int valid = ANYW ? W - col0 : 8;  // outputs of this chunk that exist
        const uint32_t first_col = col0 == 0 ? 0xffffff00u : 0xffffffffu;  // col < 1; col > W-1 cannot happen
        int ooff = voff + W + 4;  // y0 * W + col0
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
                if (r == 0 && row < 1) o[0] = o[1] = 0;  // row > H-1 cannot happen
                o[0] &= first_col;
            }
            if (!ANYW) {
                const u32x2 ov = {o[0], o[1]};
                __builtin_amdgcn_raw_buffer_store_b64(ov, rs_out, ooff, 0, IAK_NT_STENCIL_ST ? 2 : 0);
                ooff += W;
                if (HIST && row < H) {
                    hist_add4(sh, o[0], hcol);
                    hist_add4(sh, o[1], hcol);
                }
            } else if (row < H) {
                uint8_t *q = dst + (long long)row * W + col0;
                if (valid >= 8) {
                    st8_unaligned(q, o[0], o[1]);
                } else {
                    const uint64_t keep = (1ull << (8 * valid)) - 1;  // valid in [1, 7]
                    o[0] &= (uint32_t)keep;
                    o[1] &= (uint32_t)(keep >> 32);
#pragma unroll
                    for (int i = 0; i < 7; i++)
                        if (i < valid) q[i] = (uint8_t)(o[i >> 2] >> (8 * (i & 3)));
                }
                if (HIST) {
                    hist_add4(sh, o[0], hcol);
                    hist_add4(sh, o[1], hcol);
                }
            }
            up = mid;
            mid = dn;
        }
    }
    if (HIST) hist_flush(sh, hist + (long long)blockIdx.y * 256);
}


static bool aligned8_ok(const uint8_t *in, const uint8_t *out, int W, long long in_stride, long long out_stride) {
    return (W % 8 == 0) && (reinterpret_cast<uintptr_t>(in) % 4 == 0) && (reinterpret_cast<uintptr_t>(out) % 8 == 0) &&
           in_stride % 4 == 0 && out_stride % 8 == 0;
}

// hist: the output's histogram on the way (few, large workgroups per image: each pays a 32 KiB
// table and a 256-atomic merge)
template <class Op>
static int launch8(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                   const Op &op, void *stream, uint32_t *hist = nullptr, bool hist_cleared = false) {
    hipStream_t s = as_stream(stream);
    const bool anyw = !aligned8_ok(in, out, W, in_stride, out_stride);
    const long long items = (long long)((W + 7) >> 3) * ((H + 7) / 8);
    if (hist && !hist_cleared) IAK_HIP(hipMemsetAsync(hist, 0, sizeof(uint32_t) * 256 * (size_t)B, s));
    const dim3 grid(blocks_x(items, 256, B, hist ? 4096 : 1 << 20), B);
    // item / chunks as a multiplication: exact while item * (magic * chunks - 2^32) < 2^32
    const uint32_t chunks = (uint32_t)((W + 7) >> 3);
    uint32_t chunks_magic = (uint32_t)(((1ULL << 32) + chunks - 1) / chunks);
    if ((unsigned long long)items * ((unsigned long long)chunks_magic * chunks - (1ULL << 32)) >= (1ULL << 32) || chunks == 1)
        chunks_magic = 0;
#define IAK_LAUNCH8(HIST, ANYW)                                                                                       \
    hipLaunchKernelGGL((stencil8_kernel<Op, 8, HIST, ANYW>), grid, dim3(256), 0, s, in, out, W, H, in_stride, out_stride, \
                       op, hist, chunks_magic)
    if (hist) {
        if (anyw) IAK_LAUNCH8(true, true); else IAK_LAUNCH8(true, false);
    } else {
        if (anyw) IAK_LAUNCH8(false, true); else IAK_LAUNCH8(false, false);
    }
#undef IAK_LAUNCH8
    IAK_LAUNCH_CHECK();
    return 0;
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

// The 3x3 operations of the 8-pixel path with the output's histogram as a by-product, for the
// executor (not part of the C ABI).  IAK_E_UNSUPPORTED when the operation or the buffers do not
// qualify; the caller then runs the operation and a histogram pass separately.
int linear_with_contraction(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                            const float *kernel, int kW, int kH, int mode, bool fused, void *stream);
int stencil3_with_hist(const iak_op &o, const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride,
                       long long out_stride, uint32_t *hist, void *stream, bool hist_cleared, bool fused) {
    if (check_image_args(in, out, W, H, B, in_stride, out_stride) || !hist) return IAK_E_ARG;
    if (!fast8_ok(in, out, W, H, in_stride, out_stride)) return IAK_E_UNSUPPORTED;
    auto go = [&](const auto &op) { return launch8(in, out, W, H, B, in_stride, out_stride, op, stream, hist, hist_cleared); };
    switch (o.type) {
    case IAK_OP_LINEAR:
        if (o.kW != 3 || o.kH != 3) return IAK_E_UNSUPPORTED;
        return with_linear3_op<false>(o.fkernel, go, fused);
    case IAK_OP_AVERAGE:
        if (o.kW != 3 || o.kH != 3) return IAK_E_UNSUPPORTED;
        return with_linear3_op<true>(o.fkernel, go, fused);
    case IAK_OP_MEDIAN:
        if (o.kW != 3 || o.kH != 3) return IAK_E_UNSUPPORTED;
        for (int i = 0; i < 9; i++)
            if (o.ikernel[i] != 1) return IAK_E_UNSUPPORTED;
        return go(Median8());
    case IAK_OP_SOBEL: return go(Sobel8());
    case IAK_OP_COMPASS: return go(Compass8());
    case IAK_OP_ERODE: case IAK_OP_DILATE: {
        if (o.kW != 3 || o.kH != 3) return IAK_E_UNSUPPORTED;
        if (o.type == IAK_OP_ERODE) {
            Erode8 op;
            op.mask = se_mask3(o.ikernel);
            return go(op);
        }
        Dilate8 op;
        op.mask = se_mask3(o.ikernel);
        return go(op);
    }
    default: return IAK_E_UNSUPPORTED;
    }
}

// A run of 3x3 erosions / dilations as one pass (for the executor; not part of the C ABI): two or
// three of them, or one to three on the image thresholded first when `tables` is given (slot 0 of
// each image's tables = the threshold's table).  hist may be null.  IAK_E_UNSUPPORTED when the
// run or the buffers do not qualify.
int morph_chain3(const iak_op *ops, int n, const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride,
                 long long out_stride, uint32_t *hist, const uint16_t *tables, void *stream, bool hist_cleared) {
    if (check_image_args(in, out, W, H, B, in_stride, out_stride)) return IAK_E_ARG;
    if (n < (tables ? 1 : 2) || n > MC_MAX_OPS) return IAK_E_UNSUPPORTED;
    // 32 pixels per register where the run's output is two-valued and the rows are whole words; the only form
    // that takes a Sobel stage
    MorphBitsDesc db;
    db.n = n;
    bool any_sobel = false;
    for (int k = 0; k < MBS_MAX_OPS; k++) { db.mask[k] = 0; db.kind[k] = MBS_ERODE; }
    for (int k = 0; k < n; k++) {
        const iak_op &o = ops[k];
        if (o.type == IAK_OP_SOBEL) {
            db.kind[k] = MBS_SOBEL;
            any_sobel = true;
            continue;
        }
        if ((o.type != IAK_OP_ERODE && o.type != IAK_OP_DILATE) || o.kW != 3 || o.kH != 3 || !o.ikernel) return IAK_E_UNSUPPORTED;
        db.mask[k] = se_mask3(o.ikernel);
        db.kind[k] = o.type == IAK_OP_DILATE ? MBS_DILATE : MBS_ERODE;
    }
    if (any_sobel || !getenv("IAK_NO_MORPH_BITS")) {
        const int rc = launch_morph_bits(in, out, W, H, B, in_stride, out_stride, db, hist, tables, as_stream(stream), hist_cleared);
        if (rc != IAK_E_UNSUPPORTED || any_sobel) return rc;
    }
    MorphChainDesc d;
    d.n = n;
    for (int k = 0; k < MC_MAX_OPS; k++) { d.mask[k] = 0; d.dilate[k] = 0; }
    for (int k = 0; k < n; k++) {
        const iak_op &o = ops[k];
        if ((o.type != IAK_OP_ERODE && o.type != IAK_OP_DILATE) || o.kW != 3 || o.kH != 3 || !o.ikernel) return IAK_E_UNSUPPORTED;
        d.mask[k] = se_mask3(o.ikernel);
        d.dilate[k] = o.type == IAK_OP_DILATE;
    }
    return launch_morph_chain(in, out, W, H, B, in_stride, out_stride, d, hist, tables, as_stream(stream), hist_cleared);
}

}  // namespace iak

using namespace iak;

extern "C" {

int iak_linear(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
               const float *kernel, int kW, int kH, int mode, void *stream) {
    return iak::linear_with_contraction(in, out, W, H, B, in_stride, out_stride, kernel, kW, kH, mode, contraction(), stream);
}

}  // extern "C"

// iak_linear with the contraction setting as an argument (a pipeline keeps the one it was created under)
int iak::linear_with_contraction(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                                 const float *kernel, int kW, int kH, int mode, bool fused, void *stream) {
    int rc = check_image_args(in, out, W, H, B, in_stride, out_stride);
    if (rc) return rc;
    if (!kernel || kW < 1 || kH < 1) return IAK_E_ARG;
    if (mode != IAK_LINEAR_CLAMP && mode != IAK_LINEAR_AVERAGE) return IAK_E_ARG;
    if ((long long)kW * kH > IAK_MAX_TAPS) return IAK_E_TOO_LARGE;
    if (kW == 3 && kH == 3 && fast8_ok(in, out, W, H, in_stride, out_stride)) {
        auto go = [&](const auto &op) { return launch8(in, out, W, H, B, in_stride, out_stride, op, stream); };
        return mode == IAK_LINEAR_AVERAGE ? with_linear3_op<true>(kernel, go, fused) : with_linear3_op<false>(kernel, go, fused);
    }
    // taps actually visited are (2a+1) x (2s+1), indexed consecutively (imageAnalysis.cu:518-523)
    Taps t;
    const int a = (kH - 1) / 2, s = (kW - 1) / 2;
    const int visited = (2 * a + 1) * (2 * s + 1);
    for (int i = 0; i < visited; i++) t.f[i] = kernel[i];
    // the tiled, row and box kernels are built on the unfused accumulation: contraction mode takes the general kernel
    if (!fused) {
        rc = launch_tiled_linear(in, out, W, H, B, in_stride, out_stride, 2 * s + 1, 2 * a + 1, kW * kH,
                                 mode == IAK_LINEAR_AVERAGE, t, as_stream(stream));
        if (rc != IAK_E_UNSUPPORTED) return rc;
    }
    if (mode == IAK_LINEAR_CLAMP)
        hipLaunchKernelGGL((generic_float_kernel<G_LINEAR>), generic_grid(W, H, B), dim3(256), 0, as_stream(stream), in,
                           out, W, H, in_stride, out_stride, kW, kH, t, fused);
    else
        hipLaunchKernelGGL((generic_float_kernel<G_AVERAGE>), generic_grid(W, H, B), dim3(256), 0, as_stream(stream), in,
                           out, W, H, in_stride, out_stride, kW, kH, t, fused);
    IAK_LAUNCH_CHECK();
    return 0;
}

extern "C" {

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
    if (kW == 3 && kH == 3 && ones && fast8_ok(in, out, W, H, in_stride, out_stride))
        return launch8(in, out, W, H, B, in_stride, out_stride, Median8(), stream);
    if (kW == 3 && kH == 3 && fast8_ok(in, out, W, H, in_stride, out_stride)) {
        bool small = true;
        for (int i = 0; i < 9; i++) small = small && weights[i] <= 255;
        if (small) {
            WMedian8 op;
            for (int i = 0; i < 9; i++) op.w[i] = (uint32_t)weights[i];
            op.rank = (uint32_t)(kernSum / 2);
            return launch8(in, out, W, H, B, in_stride, out_stride, op, stream);
        }
    }
    if (ones && kW * kH <= 49 && !getenv("IAK_NO_BITSLICE_MEDIAN")) {
        rc = launch_bitslice_median(in, out, W, H, B, in_stride, out_stride, kW, kH, as_stream(stream));
        if (rc != IAK_E_UNSUPPORTED) return rc;
    }
    if (ones) {
        rc = launch_column_hist_median(in, out, W, H, B, in_stride, out_stride, kW, kH, as_stream(stream));
        if (rc != IAK_E_UNSUPPORTED) return rc;
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
    if (fast8_ok(in, out, W, H, in_stride, out_stride))
        return launch8(in, out, W, H, B, in_stride, out_stride, Sobel8(), stream);
    hipLaunchKernelGGL((generic_edge_kernel<G_SOBEL>), generic_grid(W, H, B), dim3(256), 0, as_stream(stream), in, out, W,
                       H, in_stride, out_stride);
    IAK_LAUNCH_CHECK();
    return 0;
}

int iak_compass(const uint8_t *in, uint8_t *out, int W, int H, int B, long long in_stride, long long out_stride,
                void *stream) {
    int rc = check_image_args(in, out, W, H, B, in_stride, out_stride);
    if (rc) return rc;
    if (fast8_ok(in, out, W, H, in_stride, out_stride))
        return launch8(in, out, W, H, B, in_stride, out_stride, Compass8(), stream);
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
    if (kW == 3 && kH == 3 && fast8_ok(in, out, W, H, in_stride, out_stride)) {
        if (erode) {
            Erode8 op;
            op.mask = se_mask3(se);
            return launch8(in, out, W, H, B, in_stride, out_stride, op, stream);
        }
        Dilate8 op;
        op.mask = se_mask3(se);
        return launch8(in, out, W, H, B, in_stride, out_stride, op, stream);
    }
    ITaps t;
    const int a = (kH - 1) / 2, s = (kW - 1) / 2;
    const int visited = (2 * a + 1) * (2 * s + 1);
    for (int i = 0; i < visited; i++) t.v[i] = se[i];
    rc = launch_morph_rows(in, out, W, H, B, in_stride, out_stride, t.v, 2 * s + 1, 2 * a + 1, erode, as_stream(stream));
    if (rc != IAK_E_UNSUPPORTED) return rc;
    rc = launch_tiled_morph(in, out, W, H, B, in_stride, out_stride, t.v, 2 * s + 1, 2 * a + 1, erode, as_stream(stream));
    if (rc != IAK_E_UNSUPPORTED) return rc;
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
