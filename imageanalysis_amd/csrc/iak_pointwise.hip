// Per-pixel (no neighbourhood) kernels: colour conversion, histogram, look-up family,
// noise application.  All are HBM-bound byte streams: every lane moves 16 pixels per
// load/store (one dwordx4), a workgroup is 256 lanes (4 waves), the grid is
// (blocks per image, B) with a grid-stride loop inside the image.
#include "iak_common.hpp"
#include "iak_device_fns.hpp"
#include "iak_hist.hpp"

namespace iak {

// ---------------------------------------------------------------------------------------
// Generic 16-pixels-per-lane streaming pass.  F provides
//   void begin(int b)                       per-image set-up (may use LDS + __syncthreads)
//   u32x4 apply16(u32x4 px, long long off)  16 pixels starting at byte offset off
//   uint32_t apply1(uint32_t px, long long off)
// 16-byte accesses need no alignment; the last N % 16 pixels of an image go one by one.
// ---------------------------------------------------------------------------------------
template <class F>
__global__ __launch_bounds__(256) void pointwise_kernel(const uint8_t *__restrict__ in,
                                                        uint8_t *__restrict__ out, long long N,
                                                        long long in_stride, long long out_stride, F f) {
    const int b = blockIdx.y;
    const uint8_t *src = in + (long long)b * in_stride;
    uint8_t *dst = out + (long long)b * out_stride;
    f.begin(b);
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nthreads = (long long)gridDim.x * blockDim.x;
    const long long nchunks = N >> 4;
    // two chunks per iteration: both loads are in flight before the first is used
    for (long long c = tid; c < nchunks; c += 2 * nthreads) {
        const long long c1 = c + nthreads;
        const bool two = c1 < nchunks;
        const u32x4 v0 = ld16<IAK_NT_CHAIN_LD>(src + (c << 4));
        const u32x4 v1 = ld16<IAK_NT_CHAIN_LD>(src + ((two ? c1 : c) << 4));
        st16<IAK_NT_CHAIN_ST>(dst + (c << 4), f.apply16(v0, c << 4));
        if (two) st16<IAK_NT_CHAIN_ST>(dst + (c1 << 4), f.apply16(v1, c1 << 4));
    }
    for (long long t = (nchunks << 4) + tid; t < N; t += nthreads) dst[t] = (uint8_t)f.apply1(src[t], t);
}

template <class F>
static int launch_pointwise(const uint8_t *in, uint8_t *out, long long N, int B, long long in_stride,
                            long long out_stride, F f, void *stream) {
    if (!in || !out || N <= 0 || B <= 0) return IAK_E_ARG;
    dim3 grid(blocks_x((N + 15) >> 4, 256, B), B);
    hipLaunchKernelGGL((pointwise_kernel<F>), grid, dim3(256), 0, as_stream(stream), in, out, N, in_stride, out_stride, f);
    IAK_LAUNCH_CHECK();
    return 0;
}

// ---- look-up through a 256-byte table held in LDS -------------------------------------
struct LutTable {
    uint8_t v[256];
};

struct MapFn {  // out = (uchar)map[in], map int32[256] per image in global memory
    const int32_t *map;
    long long map_stride;
    __device__ void begin(int b) {
        __shared__ uint8_t lut[256];
        lut_ = lut;
        if (threadIdx.x < 256) lut[threadIdx.x] = (uint8_t)map[(long long)b * map_stride + threadIdx.x];
        __syncthreads();
    }
    __device__ __forceinline__ u32x4 apply16(u32x4 px, long long) const { return lut16(lut_, px); }
    __device__ __forceinline__ uint32_t apply1(uint32_t px, long long) const { return lut_[px]; }
    uint8_t *lut_;
};

struct ConstLutFn {  // same table for every image, passed by value
    LutTable t;
    __device__ void begin(int) {
        __shared__ uint8_t lut[256];
        lut_ = lut;
        if (threadIdx.x < 256) lut[threadIdx.x] = t.v[threadIdx.x];
        __syncthreads();
    }
    __device__ __forceinline__ u32x4 apply16(u32x4 px, long long) const { return lut16(lut_, px); }
    __device__ __forceinline__ uint32_t apply1(uint32_t px, long long) const { return lut_[px]; }
    uint8_t *lut_;
};

// ---- threshold: in > t ? 255 : 0 ---------------------------------------------------------
__device__ __forceinline__ uint32_t thr4(uint32_t x, int t) {
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) o |= ((int)byte_of(x, k) > t ? 0xffu : 0u) << (8 * k);
    return o;
}

struct ThresholdFn {
    int t;
    const int32_t *per_image;  // nullptr: use t
    __device__ void begin(int b) {
        if (per_image) t = per_image[b];
    }
    __device__ __forceinline__ u32x4 apply16(u32x4 px, long long) const {
        u32x4 o;
#pragma unroll
        for (int w = 0; w < 4; w++) o[w] = thr4(px[w], t);
        return o;
    }
    __device__ __forceinline__ uint32_t apply1(uint32_t px, long long) const { return (int)px > t ? 255u : 0u; }
};

// ---- salt and pepper from the code plane (0 keep, 1 pepper, 2 salt) ----------------------
struct SaltPepperFn {
    const uint8_t *sp;
    __device__ void begin(int) {}
    __device__ __forceinline__ u32x4 apply16(u32x4 px, long long off) const {
        u32x4 c = ld16(sp + off);
        u32x4 o;
#pragma unroll
        for (int w = 0; w < 4; w++) o[w] = sp4(px[w], c[w]);
        return o;
    }
    __device__ __forceinline__ uint32_t apply1(uint32_t px, long long off) const {
        uint32_t c = sp[off];
        return c == 0 ? px : (c == 2 ? 255u : 0u);
    }
};

// ---- Gaussian noise: out = (uchar)(in + table[rv]) ------------------------------------
struct GaussFn {
    const uint8_t *rv;
    const uint16_t *table;
    long long table_stride;
    __device__ void begin(int b) {
        __shared__ uint16_t tab[256];
        tab_ = tab;
        if (threadIdx.x < 256) tab[threadIdx.x] = table[(long long)b * table_stride + threadIdx.x];
        __syncthreads();
    }
    __device__ __forceinline__ uint32_t one(uint32_t px, uint32_t r) const {
        uint32_t e = tab_[r];
        return (e & 0x100u) ? 0u : ((px + e) & 0xffu);
    }
    __device__ __forceinline__ u32x4 apply16(u32x4 px, long long off) const {
        u32x4 r = ld16(rv + off);
        u32x4 o;
#pragma unroll
        for (int w = 0; w < 4; w++)
            o[w] = pack4(one(byte_of(px[w], 0), byte_of(r[w], 0)), one(byte_of(px[w], 1), byte_of(r[w], 1)),
                         one(byte_of(px[w], 2), byte_of(r[w], 2)), one(byte_of(px[w], 3), byte_of(r[w], 3)));
        return o;
    }
    __device__ __forceinline__ uint32_t apply1(uint32_t px, long long off) const { return one(px, rv[off]); }
    uint16_t *tab_;
};

// ---------------------------------------------------------------------------------------
// Grayscale (+ histogram).  Histogram privatisation: one 256x32 table of 32-bit counters
// per workgroup in LDS (32 KiB), counter (bin, lane & 31) at word bin*32 + (lane & 31).
// A lane only ever touches the LDS bank equal to its lane number mod 32, so an LDS
// atomic wave-instruction has no bank conflict whatever the pixel values are -- a
// constant image costs the same as a uniform-random one.  Lanes l and l+32 can meet on
// one counter; they are served in different halves of the instruction.  The 32 columns
// are summed at the end and each non-empty bin is merged into the image's global
// histogram with one atomic per workgroup.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lum1(uint32_t r, uint32_t g, uint32_t b) {
    // imageAnalysis.cu:22, double literals, left-to-right, no contraction
    double y = (.21 * (double)r + .72 * (double)g) + .07 * (double)b;
    return (uint32_t)(int)y;
}
// the same expression as nvcc's default -fmad=true compiles it: two fused multiply-adds (iak_set_contraction)
__device__ __forceinline__ uint32_t lum1_fused(uint32_t r, uint32_t g, uint32_t b) {
    double y = __builtin_fma(.07, (double)b, __builtin_fma(.21, (double)r, .72 * (double)g));
    return (uint32_t)(int)y;
}
__device__ __forceinline__ uint32_t avg1(uint32_t r, uint32_t g, uint32_t b) { return (r + g + b) / 3u; }

template <int METHOD>
__device__ __forceinline__ uint32_t gray4(uint32_t r, uint32_t g, uint32_t b) {
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t y = METHOD == 0   ? lum1(byte_of(r, k), byte_of(g, k), byte_of(b, k))
                     : METHOD == 3 ? lum1_fused(byte_of(r, k), byte_of(g, k), byte_of(b, k))
                                   : avg1(byte_of(r, k), byte_of(g, k), byte_of(b, k));
        if (METHOD == 2) y = byte_of(r, k);  // one plane as it is (the G and B loads are dead code)
        o |= (y & 0xffu) << (8 * k);
    }
    return o;
}

// 1024 lanes share one 32 KiB histogram table, so two workgroups (32 waves) fit a CU -- which needs the
// pass to stay within 64 registers per lane (waves_per_eu).  A lane's first chunk is requested before the
// table is cleared and every further chunk before the current one is converted, so the clear, the f64
// arithmetic and the counting all run under loads in flight; the launcher sizes the grid for about two
// chunks per lane, so workgroups come and go and are in different phases at any time
// (500 x 768x568: 153 us as three long-lived workgroups per image, 141 us this way; the bare
// three-planes-in, one-plane-out traffic takes 136 us).
#define GRAY_THREADS 1024
template <int METHOD, bool HIST>
__global__ __launch_bounds__(GRAY_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8)))
void gray_kernel(const uint8_t *__restrict__ rgb, uint8_t *__restrict__ gray, uint32_t *__restrict__ hist, long long N,
                 long long rgb_stride, long long gray_stride) {
    __shared__ __attribute__((aligned(16))) uint32_t sh[HIST ? 256 * HIST_COLS : 4];
    const int b = blockIdx.y;
    const uint8_t *R = rgb + (long long)b * rgb_stride;
    const uint8_t *G = R + N;
    const uint8_t *Bp = G + N;
    uint8_t *dst = gray + (long long)b * gray_stride;
    const int col = threadIdx.x & (HIST_COLS - 1);
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nthreads = (long long)gridDim.x * blockDim.x;
    const long long nchunks = N >> 4;
    long long c = tid;
    u32x4 r = {0, 0, 0, 0}, g = r, bl = r;
    if (c < nchunks) {
        // METHOD 2 reads one plane only (the other two pointers may lie past the batch)
        r = ld16<IAK_NT_GRAY_LD>(R + (c << 4));
        if (METHOD != 2) {
            g = ld16<IAK_NT_GRAY_LD>(G + (c << 4));
            bl = ld16<IAK_NT_GRAY_LD>(Bp + (c << 4));
        }
    }
    if (HIST) hist_zero_wide<GRAY_THREADS>(sh);
    while (c < nchunks) {
        const long long cn = c + nthreads;
        u32x4 rn = r, gn = g, bn = bl;
        if (cn < nchunks) {
            rn = ld16<IAK_NT_GRAY_LD>(R + (cn << 4));
            if (METHOD != 2) {
                gn = ld16<IAK_NT_GRAY_LD>(G + (cn << 4));
                bn = ld16<IAK_NT_GRAY_LD>(Bp + (cn << 4));
            }
        }
        u32x4 y;
#pragma unroll
        for (int w = 0; w < 4; w++) y[w] = gray4<METHOD>(r[w], g[w], bl[w]);
        st16<IAK_NT_GRAY_ST>(dst + (c << 4), y);
        if (HIST) {
#pragma unroll
            for (int w = 0; w < 4; w++) hist_add4(sh, y[w], col);
        }
        r = rn;
        g = gn;
        bl = bn;
        c = cn;
    }
    const long long scalar_from = nchunks << 4;
    for (long long t = scalar_from + tid; t < N; t += nthreads) {
        uint32_t y = METHOD == 2 ? R[t] : (METHOD == 0 ? lum1(R[t], G[t], Bp[t]) : (METHOD == 3 ? lum1_fused(R[t], G[t], Bp[t]) : avg1(R[t], G[t], Bp[t])));
        dst[t] = (uint8_t)y;
        if (HIST) atomicAdd(&sh[(y & 0xff) * HIST_COLS + col], 1u);
    }
    if (HIST) hist_flush_wide<GRAY_THREADS>(sh, hist + (long long)b * 256);
}

// 1024 lanes per table, as the grayscale pass: 32 waves per CU instead of the 20 that five
// 256-lane workgroups with a table each would give, and a fifth of the tables to clear and merge.
// HIST_UNROLL chunks per lane are in flight at any time (with one the pass is bound by memory
// latency, 3 TB/s from HBM, not by the counting): the first set is requested before the table is
// cleared, every further set before the current one is counted.
#ifndef HIST_UNROLL
#define HIST_UNROLL 2
#endif
__global__ __launch_bounds__(GRAY_THREADS) void hist_kernel(const uint8_t *__restrict__ img, uint32_t *__restrict__ hist,
                                                   long long N, long long stride) {
    __shared__ __attribute__((aligned(16))) uint32_t sh[256 * HIST_COLS];
    const int b = blockIdx.y;
    const uint8_t *src = img + (long long)b * stride;
    const int col = threadIdx.x & (HIST_COLS - 1);
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nthreads = (long long)gridDim.x * blockDim.x;
    const long long nchunks = N >> 4;
    long long c = tid;
    u32x4 v[HIST_UNROLL];
#pragma unroll
    for (int u = 0; u < HIST_UNROLL; u++) {
        const long long cu = c + u * nthreads;
        v[u] = u32x4{0, 0, 0, 0};
        if (cu < nchunks) v[u] = ld16<true>(src + (cu << 4));
    }
    hist_zero_wide<GRAY_THREADS>(sh);
    while (c < nchunks) {
        const long long cn = c + HIST_UNROLL * nthreads;
        u32x4 nx[HIST_UNROLL];
#pragma unroll
        for (int u = 0; u < HIST_UNROLL; u++) {
            const long long cu = cn + u * nthreads;
            nx[u] = v[u];
            if (cu < nchunks) nx[u] = ld16<true>(src + (cu << 4));
        }
#pragma unroll
        for (int u = 0; u < HIST_UNROLL; u++) {
            if (c + u * nthreads < nchunks) {
#pragma unroll
                for (int w = 0; w < 4; w++) hist_add4(sh, v[u][w], col);
            }
        }
#pragma unroll
        for (int u = 0; u < HIST_UNROLL; u++) v[u] = nx[u];
        c = cn;
    }
    const long long scalar_from = nchunks << 4;
    for (long long t = scalar_from + tid; t < N; t += nthreads) atomicAdd(&sh[src[t] * HIST_COLS + col], 1u);
    hist_flush_wide<GRAY_THREADS>(sh, hist + (long long)b * 256);
}

// calcMSQE's arrayDifference (imageAnalysis.cu:780-786) and its byte sum
__global__ __launch_bounds__(256) void msqe_kernel(const uint8_t *__restrict__ in, const uint8_t *__restrict__ out,
                                                   const uint32_t *__restrict__ hist, uint32_t *__restrict__ sums,
                                                   long long N, long long in_stride, long long out_stride) {
    __shared__ double f[256];
    __shared__ uint32_t part[4];
    const int b = blockIdx.y;
    f[threadIdx.x] = (double)(int)hist[(long long)b * 256 + threadIdx.x] / 255.0;
    __syncthreads();
    const uint8_t *a = in + (long long)b * in_stride;
    const uint8_t *c = out + (long long)b * out_stride;
    uint32_t acc = 0;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < N; t += (long long)gridDim.x * blockDim.x) {
        const int p = a[t];
        const int d = p - (int)c[t];
        const double v = round((double)(d * d) * f[p]);
        acc += ((uint32_t)(int)v) & 0xffu;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&sums[b], part[0] + part[1] + part[2] + part[3]);
}

}  // namespace iak

using namespace iak;

extern "C" {

// method 0 / 1: the two grayscale conversions; 2: the plane at `rgb` copied as it is; 3: method 0 with fused
// multiply-adds (contraction mode)
static int gray_launch(const uint8_t *rgb, uint8_t *gray, uint32_t *hist, int W, int H, int B, long long rgb_stride,
                       long long gray_stride, int method, void *stream, bool hist_cleared = false) {
    const long long N = (long long)W * H;
    hipStream_t s = as_stream(stream);
    dim3 block(GRAY_THREADS);
    // about two 16-pixel chunks per lane (the second is fetched while the first is converted), but no more
    // than 16K workgroups in the launch: beyond that every lane walks further
    long long bx = ((N >> 4) + 2 * GRAY_THREADS - 1) / (2 * GRAY_THREADS);
    const long long cap = (16384 + B - 1) / B;
    if (bx > cap) bx = cap;
    if (bx < 1) bx = 1;
    dim3 grid((unsigned)bx, B);
    if (hist) {
        if (!hist_cleared) IAK_HIP(hipMemsetAsync(hist, 0, sizeof(uint32_t) * 256 * (size_t)B, s));
#define GH(M) hipLaunchKernelGGL((gray_kernel<M, true>), grid, block, 0, s, rgb, gray, hist, N, rgb_stride, gray_stride)
        if (method == 0) GH(0);
        else if (method == 1) GH(1);
        else if (method == 3) GH(3);
        else GH(2);
#undef GH
    } else {
#define GN(M) hipLaunchKernelGGL((gray_kernel<M, false>), grid, block, 0, s, rgb, gray, (uint32_t *)nullptr, N, rgb_stride, gray_stride)
        if (method == 0) GN(0);
        else if (method == 1) GN(1);
        else if (method == 3) GN(3);
        else GN(2);
#undef GN
    }
    IAK_LAUNCH_CHECK();
    return 0;
}

int iak_gray_hist(const uint8_t *rgb, uint8_t *gray, uint32_t *hist, int W, int H, int B, long long rgb_stride,
                  long long gray_stride, int method, void *stream) {
    if (!rgb || !gray || W <= 0 || H <= 0 || B <= 0) return IAK_E_ARG;
    if (method != 0 && method != 1) return IAK_E_ARG;
    const long long N = (long long)W * H;
    if (rgb_stride < 3 * N || gray_stride < N) return IAK_E_ARG;
    return gray_launch(rgb, gray, hist, W, H, B, rgb_stride, gray_stride, (method == 0 && contraction()) ? 3 : method, stream);
}

int iak_channel_hist(const uint8_t *rgb, uint8_t *out, uint32_t *hist, int W, int H, int B, long long rgb_stride,
                     long long out_stride, int channel, void *stream) {
    if (!rgb || !out || W <= 0 || H <= 0 || B <= 0) return IAK_E_ARG;
    if (channel < 0 || channel > 2) return IAK_E_ARG;
    const long long N = (long long)W * H;
    if (rgb_stride < 3 * N || out_stride < N) return IAK_E_ARG;
    return gray_launch(rgb + (long long)channel * N, out, hist, W, H, B, rgb_stride, out_stride, 2, stream);
}

int iak_gray(const uint8_t *rgb, uint8_t *gray, int W, int H, int B, long long rgb_stride, long long gray_stride,
             int method, void *stream) {
    return iak_gray_hist(rgb, gray, nullptr, W, H, B, rgb_stride, gray_stride, method, stream);
}

int iak_hist(const uint8_t *img, uint32_t *hist, long long N, int B, long long stride, void *stream) {
    if (!img || !hist || N <= 0 || B <= 0 || stride < N) return IAK_E_ARG;
    hipStream_t s = as_stream(stream);
    IAK_HIP(hipMemsetAsync(hist, 0, sizeof(uint32_t) * 256 * (size_t)B, s));
    dim3 grid(hist_blocks_x(N, B), B);
    hipLaunchKernelGGL(hist_kernel, grid, dim3(GRAY_THREADS), 0, s, img, hist, N, stride);
    IAK_LAUNCH_CHECK();
    return 0;
}

int iak_msqe(const uint8_t *in, const uint8_t *out, const uint32_t *hist, uint32_t *sums, long long N, int B,
             long long in_stride, long long out_stride, void *stream) {
    if (!in || !out || !hist || !sums || N <= 0 || B <= 0 || in_stride < N || out_stride < N) return IAK_E_ARG;
    hipStream_t s = as_stream(stream);
    IAK_HIP(hipMemsetAsync(sums, 0, sizeof(uint32_t) * (size_t)B, s));
    dim3 grid(blocks_x(N, 256 * 16, B), B);
    hipLaunchKernelGGL(msqe_kernel, grid, dim3(256), 0, s, in, out, hist, sums, N, in_stride, out_stride);
    IAK_LAUNCH_CHECK();
    return 0;
}

int iak_apply_map(const uint8_t *in, uint8_t *out, long long N, int B, long long in_stride, long long out_stride,
                  const int32_t *map, long long map_stride, void *stream) {
    if (!map || (map_stride != 0 && map_stride < 256)) return IAK_E_ARG;
    MapFn f{map, map_stride, nullptr};
    return launch_pointwise(in, out, N, B, in_stride, out_stride, f, stream);
}

int iak_quantize(const uint8_t *in, uint8_t *out, long long N, int B, long long in_stride, long long out_stride,
                 const int *levels, int numLevels, void *stream) {
    if (numLevels < 0 || (numLevels > 0 && !levels)) return IAK_E_ARG;
    // imageAnalysis.cu:767-775 collapses to a table: the first level whose [min,max)
    // holds the value decides, otherwise the value is kept.
    ConstLutFn f;
    for (int v = 0; v < 256; v++) {
        uint8_t o = (uint8_t)v;
        for (int i = 0; i < numLevels; i++)
            if (v >= levels[i * 3] && v < levels[i * 3 + 1]) {
                o = (uint8_t)levels[i * 3 + 2];
                break;
            }
        f.t.v[v] = o;
    }
    f.lut_ = nullptr;
    return launch_pointwise(in, out, N, B, in_stride, out_stride, f, stream);
}

int iak_threshold(const uint8_t *in, uint8_t *out, long long N, int B, long long in_stride, long long out_stride,
                  int threshold, void *stream) {
    ThresholdFn f{threshold, nullptr};
    return launch_pointwise(in, out, N, B, in_stride, out_stride, f, stream);
}

int iak_threshold_per_image(const uint8_t *in, uint8_t *out, long long N, int B, long long in_stride,
                            long long out_stride, const int32_t *thr, void *stream) {
    if (!thr) return IAK_E_ARG;
    ThresholdFn f{0, thr};
    return launch_pointwise(in, out, N, B, in_stride, out_stride, f, stream);
}

int iak_salt_pepper(const uint8_t *in, uint8_t *out, long long N, int B, long long in_stride, long long out_stride,
                    const uint8_t *sp, void *stream) {
    if (!sp) return IAK_E_ARG;
    SaltPepperFn f{sp};
    return launch_pointwise(in, out, N, B, in_stride, out_stride, f, stream);
}

int iak_gauss_noise(const uint8_t *in, uint8_t *out, long long N, int B, long long in_stride, long long out_stride,
                    const uint8_t *rv, const uint16_t *table, long long table_stride, void *stream) {
    if (!rv || !table || (table_stride != 0 && table_stride < 256)) return IAK_E_ARG;
    GaussFn f{rv, table, table_stride, nullptr};
    return launch_pointwise(in, out, N, B, in_stride, out_stride, f, stream);
}

}  // extern "C"

namespace iak {
// The grayscale (method 0 / 1) or plane-copy (2, rgb pointing at the plane) pass for the executor, whose
// histogram table has been cleared ahead of time: no clear here.
int gray_hist_cleared(const uint8_t *rgb, uint8_t *gray, uint32_t *hist, int W, int H, int B, long long rgb_stride,
                      long long gray_stride, int method, void *stream) {
    if (!rgb || !gray || !hist || W <= 0 || H <= 0 || B <= 0 || method < 0 || method > 3) return IAK_E_ARG;
    return gray_launch(rgb, gray, hist, W, H, B, rgb_stride, gray_stride, method, stream, true);
}
}  // namespace iak
