// 256-entry work per image (equalisation map, Otsu threshold, mean / std-dev, Gaussian
// amplitude table) and the XORWOW planes.  These replace host loops of the reference that
// sat between kernel launches and forced a device->host copy + sync per image; here they
// run on the device, one image per lane or per workgroup, so a batch never leaves the GPU.
// Arithmetic is deliberately sequential where the reference's is (float32 cdf, Otsu's
// running sums): a parallel scan would round differently.
#include "iak_common.hpp"
#include "iak_device_fns.hpp"

#include <mutex>

namespace iak {

// ---- equalizeHistogram (imageAnalysis.cu:155-176) -----------------------------------------
__global__ void eq_map_kernel(const uint32_t *__restrict__ hist, int32_t *__restrict__ map, int numPixels, int B) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    eq_map_from_hist(hist + (long long)b * 256, map + (long long)b * 256, numPixels);
}

__global__ void otsu_kernel(const uint32_t *__restrict__ hist, int32_t *__restrict__ thr, int numPixels, int B) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    thr[b] = otsu_threshold_from_hist(hist + (long long)b * 256, numPixels);
}

__global__ void mean_std_kernel(const uint32_t *__restrict__ hist, float *__restrict__ out, int numPixels, int B) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    mean_std_from_hist(hist + (long long)b * 256, numPixels, &out[2 * b], &out[2 * b + 1]);
}

// one workgroup (256 lanes) per image
__global__ __launch_bounds__(256) void gauss_table_kernel(const uint32_t *__restrict__ hist, int numPixels,
                                                          float meanPar, float stdPar, uint16_t *__restrict__ table) {
    __shared__ float ms[2];
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        float mean = meanPar, sd = stdPar;
        if (mean == -1 || sd == -1) mean_std_from_hist(hist + (long long)b * 256, numPixels, &mean, &sd);
        ms[0] = mean;
        ms[1] = sd * sd;  // imageAnalysis.cu:861 passes stdDev * stdDev
    }
    __syncthreads();
    table[(long long)b * 256 + threadIdx.x] = gauss_entry(threadIdx.x, ms[0], ms[1]);
}

// ---------------------------------------------------------------------------------------
// XORWOW, as cuRAND initialises it: the published xorwow recurrence (Marsaglia 2003) on
// v[0..4] plus a Weyl counter d; curand_init(seed, subsequence, 0) salts the start state
// with the seed and jumps subsequence * 2^67 steps.  The jump is linear over GF(2) on the
// 160 state bits: J_k = (one step)^(2^(67+k)) are built once on the host by repeated
// squaring and kept in device memory; lane tid applies J_k for every set bit k of tid.
// ---------------------------------------------------------------------------------------
#define XW_WORDS 800  // 160 rows x 5 words
__device__ uint32_t g_xw_jump[32 * XW_WORDS];

static inline void xw_step_v(uint32_t *v) {
    uint32_t t = v[0] ^ (v[0] >> 2);
    v[0] = v[1];
    v[1] = v[2];
    v[2] = v[3];
    v[3] = v[4];
    v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
}

static void xw_matvec_host(const uint32_t *m, const uint32_t *v, uint32_t *out) {
    uint32_t r[5] = {0, 0, 0, 0, 0};
    for (int bit = 0; bit < 160; bit++)
        if (v[bit >> 5] & (1u << (bit & 31)))
            for (int k = 0; k < 5; k++) r[k] ^= m[bit * 5 + k];
    for (int k = 0; k < 5; k++) out[k] = r[k];
}

static void xw_square_host(const uint32_t *m, uint32_t *out) {
    static thread_local uint32_t tmp[XW_WORDS];
    for (int r = 0; r < 160; r++) xw_matvec_host(m, m + r * 5, tmp + r * 5);
    for (int i = 0; i < XW_WORDS; i++) out[i] = tmp[i];
}

static const uint32_t *xw_host_matrices() {
    static uint32_t *mats = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        mats = new uint32_t[32 * XW_WORDS];
        uint32_t *m = mats;
        for (int bit = 0; bit < 160; bit++) {  // one step
            uint32_t v[5] = {0, 0, 0, 0, 0};
            v[bit >> 5] = 1u << (bit & 31);
            xw_step_v(v);
            for (int k = 0; k < 5; k++) m[bit * 5 + k] = v[k];
        }
        for (int i = 0; i < 67; i++) xw_square_host(m, m);  // 2^67 steps
        for (int k = 1; k < 32; k++) xw_square_host(mats + (k - 1) * XW_WORDS, mats + k * XW_WORDS);
    });
    return mats;
}

struct XwSeed {
    uint32_t v[5];
    uint32_t d;
};

__global__ __launch_bounds__(256) void rng_planes_kernel(uint32_t *__restrict__ x0, uint32_t *__restrict__ x1,
                                                         long long N, XwSeed seed) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    uint32_t v0 = seed.v[0], v1 = seed.v[1], v2 = seed.v[2], v3 = seed.v[3], v4 = seed.v[4];
    const uint32_t tid = (uint32_t)t;
    for (int k = 0; k < 32; k++) {
        if (!((tid >> k) & 1u)) continue;
        const uint32_t *m = g_xw_jump + k * XW_WORDS;
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0;
        const uint32_t vv[5] = {v0, v1, v2, v3, v4};
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const uint32_t w = vv[i];
            for (int j = 0; j < 32; j++) {
                const uint32_t mask = 0u - ((w >> j) & 1u);
                const uint32_t *row = m + (i * 32 + j) * 5;
                r0 ^= row[0] & mask;
                r1 ^= row[1] & mask;
                r2 ^= row[2] & mask;
                r3 ^= row[3] & mask;
                r4 ^= row[4] & mask;
            }
        }
        v0 = r0; v1 = r1; v2 = r2; v3 = r3; v4 = r4;
    }
    uint32_t d = seed.d;
    // two curand() draws
    uint32_t tt = v0 ^ (v0 >> 2);
    uint32_t n4 = (v4 ^ (v4 << 4)) ^ (tt ^ (tt << 1));
    d += 362437u;
    x0[t] = n4 + d;
    // state is now (v1, v2, v3, v4, n4)
    tt = v1 ^ (v1 >> 2);
    uint32_t n5 = (n4 ^ (n4 << 4)) ^ (tt ^ (tt << 1));
    d += 362437u;
    x1[t] = n5 + d;
}

__global__ __launch_bounds__(256) void rv_plane_kernel(const uint32_t *__restrict__ x0, uint8_t *__restrict__ rv, long long N) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    // imageAnalysis.cu:837: float * double, truncated
    rv[t] = (uint8_t)(int)((double)xw_uniform(x0[t]) * (255 - 0 + .999999));
}

__global__ __launch_bounds__(256) void sp_plane_kernel(const uint32_t *__restrict__ x0, const uint32_t *__restrict__ x1,
                                                       uint8_t *__restrict__ sp, long long N, int level) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    int randVal = (int)((double)xw_uniform(x0[t]) * (level - 0 + .999999));  // :724
    uint8_t code = 0;
    if (randVal == level) code = ((double)xw_uniform(x1[t]) > .5) ? 2 : 1;  // :726-731
    sp[t] = code;
}

}  // namespace iak

using namespace iak;

extern "C" {

int iak_eq_map(const uint32_t *hist, int32_t *map, int numPixels, int B, void *stream) {
    if (!hist || !map || numPixels <= 0 || B <= 0) return IAK_E_ARG;
    hipLaunchKernelGGL(eq_map_kernel, dim3((B + 63) / 64), dim3(64), 0, as_stream(stream), hist, map, numPixels, B);
    IAK_LAUNCH_CHECK();
    return 0;
}

int iak_otsu_threshold(const uint32_t *hist, int32_t *thr, int numPixels, int B, void *stream) {
    if (!hist || !thr || numPixels <= 0 || B <= 0) return IAK_E_ARG;
    hipLaunchKernelGGL(otsu_kernel, dim3((B + 63) / 64), dim3(64), 0, as_stream(stream), hist, thr, numPixels, B);
    IAK_LAUNCH_CHECK();
    return 0;
}

int iak_mean_std(const uint32_t *hist, float *mean_std, int numPixels, int B, void *stream) {
    if (!hist || !mean_std || numPixels <= 0 || B <= 0) return IAK_E_ARG;
    hipLaunchKernelGGL(mean_std_kernel, dim3((B + 63) / 64), dim3(64), 0, as_stream(stream), hist, mean_std, numPixels, B);
    IAK_LAUNCH_CHECK();
    return 0;
}

int iak_gauss_table(const uint32_t *hist, int numPixels, int B, float mean, float stddev, uint16_t *table,
                    void *stream) {
    if (!table || B <= 0) return IAK_E_ARG;
    const bool from_image = (mean == -1 || stddev == -1);
    if (from_image && (!hist || numPixels <= 0)) return IAK_E_ARG;
    hipLaunchKernelGGL(gauss_table_kernel, dim3(B), dim3(256), 0, as_stream(stream), hist, numPixels, mean, stddev, table);
    IAK_LAUNCH_CHECK();
    return 0;
}

int iak_rng_planes_from_state(uint32_t *x0, uint32_t *x1, long long N, const uint32_t *state6, void *stream) {
    if (!x0 || !x1 || !state6 || N <= 0 || N > 0xffffffffLL) return IAK_E_ARG;
    hipStream_t s = as_stream(stream);
    // The jump matrices live in this code object's device memory, one copy per GPU.
    static std::mutex mu;
    static bool uploaded[64] = {false};
    int dev = 0;
    IAK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return IAK_E_UNSUPPORTED;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!uploaded[dev]) {
            const uint32_t *mats = xw_host_matrices();
            IAK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_xw_jump), mats, sizeof(uint32_t) * 32 * XW_WORDS, 0,
                                      hipMemcpyHostToDevice));
            uploaded[dev] = true;
        }
    }
    XwSeed sd;
    for (int i = 0; i < 5; i++) sd.v[i] = state6[i];
    sd.d = state6[5];
    hipLaunchKernelGGL(rng_planes_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, x0, x1, N, sd);
    IAK_LAUNCH_CHECK();
    return 0;
}

int iak_rng_planes(uint32_t *x0, uint32_t *x1, long long N, unsigned long long seed, void *stream) {
    // curand_init's seed scrambling as the public curand_kernel.h has it
    const uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u;
    const uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    const uint32_t t0 = 1099087573u * s0;
    const uint32_t t1 = 2591861531u * s1;
    const uint32_t st[6] = {123456789u + t0, 362436069u ^ t0, 521288629u + t1, 88675123u ^ t1, 5783321u + t0,
                            6615241u + t1 + t0};
    return iak_rng_planes_from_state(x0, x1, N, st, stream);
}

int iak_rv_plane(const uint32_t *x0, uint8_t *rv, long long N, void *stream) {
    if (!x0 || !rv || N <= 0) return IAK_E_ARG;
    hipLaunchKernelGGL(rv_plane_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, as_stream(stream), x0, rv, N);
    IAK_LAUNCH_CHECK();
    return 0;
}

int iak_sp_plane(const uint32_t *x0, const uint32_t *x1, uint8_t *sp, long long N, int level, void *stream) {
    if (!x0 || !x1 || !sp || N <= 0) return IAK_E_ARG;
    hipLaunchKernelGGL(sp_plane_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, as_stream(stream), x0, x1, sp, N, level);
    IAK_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
