// Writes the first two XORWOW draws of subsequences 0..n-1 (seed given, offset 0) as produced by
// rocRAND's device API, AMD's implementation of cuRAND's generator: rocrand_init(seed, tid, 0)
// then rocrand() twice -- the call sequence of the reference's noise kernels
// (imageAnalysis.cu:713,724-726 with curand_init / curand).  Used by the GPU tests as an
// independent source for the random planes.
//   rocrand_check <n> <seed> <out.bin>     out: n pairs of uint32
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void draws(unsigned int *out, long long n, unsigned long long seed) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n) return;
    rocrand_state_xorwow st;
    rocrand_init(seed, (unsigned long long)tid, 0ULL, &st);
    out[2 * tid] = rocrand(&st);
    out[2 * tid + 1] = rocrand(&st);
}

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const long long n = atoll(argv[1]);
    const unsigned long long seed = strtoull(argv[2], nullptr, 10);
    unsigned int *d = nullptr;
    if (hipMalloc((void **)&d, sizeof(unsigned int) * 2 * n) != hipSuccess) return 1;
    hipLaunchKernelGGL(draws, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d, n, seed);
    std::vector<unsigned int> h(2 * n);
    if (hipMemcpy(h.data(), d, sizeof(unsigned int) * 2 * n, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    FILE *f = fopen(argv[3], "wb");
    if (!f) return 1;
    fwrite(h.data(), sizeof(unsigned int), h.size(), f);
    fclose(f);
    return 0;
}
