// Streaming-pattern microbenchmark: which load/store form and launch shape moves bytes fastest.
// build: hipcc --offload-arch=gfx950 -O3 -o membench membench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NT_LD, int NT_ST>
__global__ __launch_bounds__(256) void copy_gs(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        u32x4 v = NT_LD ? __builtin_nontemporal_load(in + i) : in[i];
        if (NT_ST) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
// each block owns a contiguous range, 4 loads in flight per lane
template <int NT_LD, int NT_ST>
__global__ __launch_bounds__(256) void copy_blk(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, long long n) {
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
    for (long long i = lo + threadIdx.x; i < hi; i += 4 * 256) {
        u32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            long long j = i + k * 256;
            if (j < hi) v[k] = NT_LD ? __builtin_nontemporal_load(in + j) : in[j];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            long long j = i + k * 256;
            if (j < hi) { if (NT_ST) __builtin_nontemporal_store(v[k], out + j); else out[j] = v[k]; }
        }
    }
}
template <int NT_LD>
__global__ __launch_bounds__(256) void read_only(const u32x4 *__restrict__ in, uint32_t *__restrict__ sink, long long n) {
    uint32_t acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        u32x4 v = NT_LD ? __builtin_nontemporal_load(in + i) : in[i];
        acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <int NT_ST>
__global__ __launch_bounds__(256) void write_only(u32x4 *__restrict__ out, long long n) {
    u32x4 v = {1, 2, 3, 4};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        if (NT_ST) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
// 3 reads + 1 write (planar RGB -> gray shape)
template <int NT_LD, int NT_ST>
__global__ __launch_bounds__(256) void three_one(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        u32x4 a = NT_LD ? __builtin_nontemporal_load(in + i) : in[i];
        u32x4 b = NT_LD ? __builtin_nontemporal_load(in + n + i) : in[n + i];
        u32x4 c = NT_LD ? __builtin_nontemporal_load(in + 2 * n + i) : in[2 * n + i];
        u32x4 v = a + b + c;
        if (NT_ST) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}

int main() {
    const long long bytes = 218112000LL;  // 500 x 768 x 568
    const long long n = bytes / 16;
    void *in, *out; uint32_t *sink;
    CK(hipMalloc(&in, 3 * bytes)); CK(hipMalloc(&out, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(in, 1, 3 * bytes)); CK(hipMemset(out, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char *name, double gbytes, auto launch) {
        std::vector<float> t;
        for (int r = 0; r < 12; r++) {
            hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (r >= 2) t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        printf("%-34s med %7.1f us  min %7.1f us  %7.1f GB/s\n", name, t[t.size() / 2] * 1e3, t[0] * 1e3, gbytes / (t[t.size() / 2] * 1e-3));
    };
    const double GB = bytes / 1e9;
    const u32x4 *I = (const u32x4 *)in; u32x4 *O = (u32x4 *)out;
    for (int grid : {1024, 2048, 4096, 8192, 16384, 65536}) {
        char nm[64];
        snprintf(nm, 64, "copy gs grid=%d", grid); time(nm, 2 * GB, [&] { hipLaunchKernelGGL((copy_gs<0, 0>), dim3(grid), dim3(256), 0, 0, I, O, n); });
    }
    time("copy gs nt-store g=4096", 2 * GB, [&] { hipLaunchKernelGGL((copy_gs<0, 1>), dim3(4096), dim3(256), 0, 0, I, O, n); });
    time("copy gs nt-load g=4096", 2 * GB, [&] { hipLaunchKernelGGL((copy_gs<1, 0>), dim3(4096), dim3(256), 0, 0, I, O, n); });
    time("copy gs nt-both g=4096", 2 * GB, [&] { hipLaunchKernelGGL((copy_gs<1, 1>), dim3(4096), dim3(256), 0, 0, I, O, n); });
    for (int grid : {2048, 8192, 32768}) {
        char nm[64];
        snprintf(nm, 64, "copy blk grid=%d", grid); time(nm, 2 * GB, [&] { hipLaunchKernelGGL((copy_blk<0, 0>), dim3(grid), dim3(256), 0, 0, I, O, n); });
        snprintf(nm, 64, "copy blk nt-both grid=%d", grid); time(nm, 2 * GB, [&] { hipLaunchKernelGGL((copy_blk<1, 1>), dim3(grid), dim3(256), 0, 0, I, O, n); });
    }
    time("read only g=4096", GB, [&] { hipLaunchKernelGGL((read_only<0>), dim3(4096), dim3(256), 0, 0, I, sink, n); });
    time("read only nt g=4096", GB, [&] { hipLaunchKernelGGL((read_only<1>), dim3(4096), dim3(256), 0, 0, I, sink, n); });
    time("read only 3x g=4096", 3 * GB, [&] { hipLaunchKernelGGL((read_only<0>), dim3(4096), dim3(256), 0, 0, I, sink, 3 * n); });
    time("write only g=4096", GB, [&] { hipLaunchKernelGGL((write_only<0>), dim3(4096), dim3(256), 0, 0, O, n); });
    time("write only nt g=4096", GB, [&] { hipLaunchKernelGGL((write_only<1>), dim3(4096), dim3(256), 0, 0, O, n); });
    time("3r1w g=4096", 4 * GB, [&] { hipLaunchKernelGGL((three_one<0, 0>), dim3(4096), dim3(256), 0, 0, I, O, n); });
    time("3r1w nt-both g=4096", 4 * GB, [&] { hipLaunchKernelGGL((three_one<1, 1>), dim3(4096), dim3(256), 0, 0, I, O, n); });
    time("3r1w nt-store g=4096", 4 * GB, [&] { hipLaunchKernelGGL((three_one<0, 1>), dim3(4096), dim3(256), 0, 0, I, O, n); });
    time("3r1w g=16384", 4 * GB, [&] { hipLaunchKernelGGL((three_one<0, 0>), dim3(16384), dim3(256), 0, 0, I, O, n); });
    time("hipMemcpyDtoD", 2 * GB, [&] { hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
