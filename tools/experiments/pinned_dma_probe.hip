// Round 3: the page-locked trace of a run sometimes uploads at 28 - 37 GB/s instead of 56 (bench.py, config #4, after an earlier
// pinned buffer had been freed), with hipMemcpyAsync blocking the host for milliseconds.  This probe replays allocation patterns and
// compares hipMemcpyAsync on a second stream with a copy KERNEL that reads the pinned memory over PCIe itself.
// build: hipcc -O2 --offload-arch=gfx950 pinned_dma_probe.hip -o pinned_dma_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void __launch_bounds__(256) pull_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride],
              d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
static void measure(const char* what, void* h, size_t bytes, void* dev, hipStream_t st, size_t piece) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double best_dma = 1e9, best_host = 0, best_pull[3] = {1e9, 1e9, 1e9};
    for (int r = 0; r < 3; ++r) {
        double t0 = now();
        hipEventRecord(e0, st);
        for (size_t off = 0; off < bytes; off += piece) hipMemcpyAsync((char*)dev + off, (char*)h + off, std::min(piece, bytes - off), hipMemcpyHostToDevice, st);
        hipEventRecord(e1, st);
        double t1 = now();
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best_dma) { best_dma = ms; best_host = t1 - t0; }
    }
    const int grids[3] = {256, 1024, 4096};
    for (int g = 0; g < 3; ++g)
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(e0, st);
            for (size_t off = 0; off < bytes; off += piece)
                hipLaunchKernelGGL(pull_kernel, dim3(grids[g]), dim3(256), 0, st, (const uint4*)((char*)h + off), (uint4*)((char*)dev + off), std::min(piece, bytes - off) / 16);
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best_pull[g] = std::min<double>(best_pull[g], ms);
        }
    printf("%-44s hipMemcpyAsync %5.1f GB/s (host blocked %.2f ms)   pull kernel %5.1f / %5.1f / %5.1f GB/s (256 / 1024 / 4096 blocks)\n", what,
           bytes / best_dma * 1e-6, best_host, bytes / best_pull[0] * 1e-6, bytes / best_pull[1] * 1e-6, bytes / best_pull[2] * 1e-6);
}
int main() {
    const size_t A = (size_t)1088 << 20, B = (size_t)544 << 20;
    void* dev = nullptr; hipMalloc(&dev, A);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    void *a = nullptr, *b = nullptr, *c = nullptr;
    hipHostMalloc(&a, A, 0); memset(a, 1, A);
    measure("A (1.1 GB), 32 MB pieces", a, A, dev, st, (size_t)32 << 20);
    hipHostMalloc(&b, B, 0); memset(b, 2, B);
    measure("B (544 MB) while A is alive, 16 MB pieces", b, B, dev, st, (size_t)16 << 20);
    hipHostFree(a); hipHostFree(b);
    hipHostMalloc(&c, B, 0); memset(c, 3, B);
    measure("C (544 MB) after A and B were freed, 16 MB", c, B, dev, st, (size_t)16 << 20);
    measure("C again, 32 MB pieces", c, B, dev, st, (size_t)32 << 20);
    hipHostFree(c);
    hipHostMalloc(&a, A, 0); memset(a, 1, A);
    measure("A' (1.1 GB) after the frees, 32 MB pieces", a, A, dev, st, (size_t)32 << 20);
    return 0;
}
