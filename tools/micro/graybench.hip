// Which part of the grayscale + histogram pass costs what: the pass's own access shape (planar RGB per image,
// gridDim.y = images) with the arithmetic and the histogram switched on one at a time, and a few launch shapes.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o graybench graybench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define COLS 32

__device__ __forceinline__ uint32_t lum1(uint32_t r, uint32_t g, uint32_t b) {
    double y = (.21 * (double)r + .72 * (double)g) + .07 * (double)b;
    return (uint32_t)(int)y;
}
template <int COMPUTE>
__device__ __forceinline__ uint32_t gray4(uint32_t r, uint32_t g, uint32_t b) {
    if (COMPUTE == 0) return r ^ g ^ b;
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) o |= (lum1((r >> 8 * k) & 255, (g >> 8 * k) & 255, (b >> 8 * k) & 255) & 255) << 8 * k;
    return o;
}
__device__ __forceinline__ void hist_add4(uint32_t *sh, uint32_t w, int col) {
    atomicAdd(&sh[(w & 0xff) * COLS + col], 1u);
    atomicAdd(&sh[((w >> 8) & 0xff) * COLS + col], 1u);
    atomicAdd(&sh[((w >> 16) & 0xff) * COLS + col], 1u);
    atomicAdd(&sh[(w >> 24) * COLS + col], 1u);
}

// UNROLL chunks of 16 pixels in flight per lane; CONTIG: a workgroup owns a contiguous range of the image
template <int COMPUTE, bool HIST, int THREADS, int UNROLL, bool NT_LD, bool NT_ST, bool CONTIG>
__global__ __launch_bounds__(THREADS) void gray(const uint8_t *__restrict__ rgb, uint8_t *__restrict__ gray_out,
                                                uint32_t *__restrict__ hist, int N) {
    __shared__ uint32_t sh[HIST ? 256 * COLS : 1];
    const int b = blockIdx.y;
    const uint8_t *R = rgb + (long long)b * 3 * N, *G = R + N, *B = G + N;
    uint8_t *dst = gray_out + (long long)b * N;
    const int col = threadIdx.x & (COLS - 1);
    if (HIST) {
        for (int i = threadIdx.x; i < 256 * COLS; i += THREADS) sh[i] = 0;
        __syncthreads();
    }
    const int nchunks = N >> 4;
    int c0, c1, step;
    if (CONTIG) {
        const int per = (nchunks + gridDim.x - 1) / gridDim.x;
        c0 = per * blockIdx.x + threadIdx.x;
        c1 = min(per * (blockIdx.x + 1), nchunks);
        step = THREADS;
    } else {
        c0 = blockIdx.x * THREADS + threadIdx.x;
        c1 = nchunks;
        step = gridDim.x * THREADS;
    }
    for (int c = c0; c < c1; c += UNROLL * step) {
        u32x4 r[UNROLL], g[UNROLL], bl[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int cu = c + u * step < c1 ? c + u * step : c;
            const long long o = (long long)cu << 4;
            r[u] = NT_LD ? __builtin_nontemporal_load((const u32x4 *)(R + o)) : *(const u32x4 *)(R + o);
            g[u] = NT_LD ? __builtin_nontemporal_load((const u32x4 *)(G + o)) : *(const u32x4 *)(G + o);
            bl[u] = NT_LD ? __builtin_nontemporal_load((const u32x4 *)(B + o)) : *(const u32x4 *)(B + o);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (c + u * step < c1) {
                u32x4 y;
#pragma unroll
                for (int w = 0; w < 4; w++) y[w] = gray4<COMPUTE>(r[u][w], g[u][w], bl[u][w]);
                const long long o = (long long)(c + u * step) << 4;
                if (NT_ST) __builtin_nontemporal_store(y, (u32x4 *)(dst + o)); else *(u32x4 *)(dst + o) = y;
                if (HIST) {
#pragma unroll
                    for (int w = 0; w < 4; w++) hist_add4(sh, y[w], col);
                }
            }
        }
    }
    if (HIST) {
        __syncthreads();
        for (int v = threadIdx.x; v < 256; v += THREADS) {
            uint32_t s = 0;
            for (int i = 0; i < COLS; i++) s += sh[v * COLS + ((i + v) & (COLS - 1))];
            if (s) atomicAdd(&hist[b * 256 + v], s);
        }
    }
}


// Variant: the first chunk's loads are issued before the table is cleared, the next chunk's loads before the
// current chunk is processed (one chunk of look-ahead), every lane takes part in the merge.
template <int THREADS, int WAVES>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void gray2(const uint8_t *__restrict__ rgb, uint8_t *__restrict__ gray_out, uint32_t *__restrict__ hist, int N) {
    __shared__ uint32_t sh[256 * COLS];
    const int b = blockIdx.y;
    const uint8_t *R = rgb + (long long)b * 3 * N, *G = R + N, *B = G + N;
    uint8_t *dst = gray_out + (long long)b * N;
    const int col = threadIdx.x & (COLS - 1);
    const int nchunks = N >> 4;
    int c = blockIdx.x * THREADS + threadIdx.x;
    const int step = gridDim.x * THREADS;
    u32x4 r, g, bl;
    {
        const long long o = (long long)(c < nchunks ? c : 0) << 4;
        r = __builtin_nontemporal_load((const u32x4 *)(R + o));
        g = __builtin_nontemporal_load((const u32x4 *)(G + o));
        bl = __builtin_nontemporal_load((const u32x4 *)(B + o));
    }
    for (int i = threadIdx.x; i < 256 * COLS / 4; i += THREADS) ((u32x4 *)sh)[i] = u32x4{0, 0, 0, 0};
    __syncthreads();
    while (c < nchunks) {
        const int cn = c + step;
        u32x4 rn = r, gn = g, bn = bl;
        if (cn < nchunks) {
            const long long on = (long long)cn << 4;
            rn = __builtin_nontemporal_load((const u32x4 *)(R + on));
            gn = __builtin_nontemporal_load((const u32x4 *)(G + on));
            bn = __builtin_nontemporal_load((const u32x4 *)(B + on));
        }
        u32x4 y;
#pragma unroll
        for (int w = 0; w < 4; w++) y[w] = gray4<1>(r[w], g[w], bl[w]);
        *(u32x4 *)(dst + ((long long)c << 4)) = y;
#pragma unroll
        for (int w = 0; w < 4; w++) hist_add4(sh, y[w], col);
        r = rn; g = gn; bl = bn;
        c = cn;
    }
    __syncthreads();
    constexpr int PER = THREADS / 256;       // lanes per bin
    constexpr int CPL = COLS / PER;          // columns per lane
    const int v = threadIdx.x / PER, part = threadIdx.x % PER;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < CPL; i++) s += sh[v * COLS + ((part * CPL + i + v) & (COLS - 1))];
#pragma unroll
    for (int m = 1; m < PER; m <<= 1) s += __shfl_xor(s, m);
    if (part == 0 && s) atomicAdd(&hist[b * 256 + v], s);
}
int main() {
    const int W = 768, H = 568, B = 500, N = W * H;
    const long long bytes = (long long)B * N;
    uint8_t *in, *out; uint32_t *hist;
    CK(hipMalloc(&in, 3 * bytes)); CK(hipMalloc(&out, bytes)); CK(hipMalloc(&hist, 256 * 4 * B));
    {   // cell-like: bright background with noise, dark blobs
        std::vector<uint8_t> h(3 * (size_t)N * 8);
        uint32_t s = 12345;
        for (size_t i = 0; i < h.size(); i++) { s = s * 1664525u + 1013904223u; h[i] = (uint8_t)(((i / N) % 3 == 0 ? 190 : 200) + ((s >> 24) & 15)); }
        for (int b = 0; b < B; b += 8) CK(hipMemcpy(in + (size_t)b * 3 * N, h.data(), std::min<size_t>(h.size(), (size_t)(B - b) * 3 * N), hipMemcpyHostToDevice));
    }
    CK(hipMemset(hist, 0, 256 * 4 * B));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int NC = 13, ROUNDS = 15;
    std::vector<std::vector<float>> t(NC);
    const char *names[NC] = {"g1 thr=1024 u/waves=2 gx=5","g1 thr=1024 u/waves=2 gx=10","g1 thr=512 u/waves=1 gx=14","g2 thr=1024 u/waves=8 gx=10","g2 thr=1024 u/waves=8 gx=14","g2 thr=1024 u/waves=8 gx=7","g2 thr=512 u/waves=8 gx=14","g2 thr=512 u/waves=8 gx=18","g2 thr=512 u/waves=8 gx=27","g2 thr=512 u/waves=8 gx=10","g2 thr=256 u/waves=5 gx=27","g2 thr=256 u/waves=5 gx=54","g2 thr=512 u/waves=6 gx=14"};
    for (int round = 0; round < ROUNDS; round++) {
        for (int k = 0; k < NC; k++) {
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0, 0);
                switch (k) {
                case 0: hipLaunchKernelGGL((gray<1, true, 1024, 2, true, false, false>), dim3(5, B), dim3(1024), 0, 0, in, out, hist, N); break;
                case 1: hipLaunchKernelGGL((gray<1, true, 1024, 2, true, false, false>), dim3(10, B), dim3(1024), 0, 0, in, out, hist, N); break;
                case 2: hipLaunchKernelGGL((gray<1, true, 512, 1, true, false, false>), dim3(14, B), dim3(512), 0, 0, in, out, hist, N); break;
                case 3: hipLaunchKernelGGL((gray2<1024, 8>), dim3(10, B), dim3(1024), 0, 0, in, out, hist, N); break;
                case 4: hipLaunchKernelGGL((gray2<1024, 8>), dim3(14, B), dim3(1024), 0, 0, in, out, hist, N); break;
                case 5: hipLaunchKernelGGL((gray2<1024, 8>), dim3(7, B), dim3(1024), 0, 0, in, out, hist, N); break;
                case 6: hipLaunchKernelGGL((gray2<512, 8>), dim3(14, B), dim3(512), 0, 0, in, out, hist, N); break;
                case 7: hipLaunchKernelGGL((gray2<512, 8>), dim3(18, B), dim3(512), 0, 0, in, out, hist, N); break;
                case 8: hipLaunchKernelGGL((gray2<512, 8>), dim3(27, B), dim3(512), 0, 0, in, out, hist, N); break;
                case 9: hipLaunchKernelGGL((gray2<512, 8>), dim3(10, B), dim3(512), 0, 0, in, out, hist, N); break;
                case 10: hipLaunchKernelGGL((gray2<256, 5>), dim3(27, B), dim3(256), 0, 0, in, out, hist, N); break;
                case 11: hipLaunchKernelGGL((gray2<256, 5>), dim3(54, B), dim3(256), 0, 0, in, out, hist, N); break;
                case 12: hipLaunchKernelGGL((gray2<512, 6>), dim3(14, B), dim3(512), 0, 0, in, out, hist, N); break;
                }
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && round > 0) t[k].push_back(ms);
            }
        }
    }
    for (int k = 0; k < NC; k++) {
        std::sort(t[k].begin(), t[k].end());
        printf("%-44s med %7.1f us  min %7.1f us  q1 %7.1f us\n", names[k], t[k][t[k].size() / 2] * 1e3, t[k][0] * 1e3, t[k][t[k].size() / 4] * 1e3);
    }
    return 0;
}
