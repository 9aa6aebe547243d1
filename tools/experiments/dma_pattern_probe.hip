// Which enqueue pattern makes hipMemcpyAsync from page-locked memory slow (27 GB/s instead of 56, round 3)?  A compute stream runs a
// long ALU kernel chain, a copy stream uploads a pinned buffer in pieces; pieces of several sizes, back to back or each behind an
// event of the compute stream, with and without kernels of its own between the copies.
// build: hipcc -O2 --offload-arch=gfx950 dma_pattern_probe.hip -o dma_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void spin_kernel(uint32_t* out, int iters) {
    uint32_t x = threadIdx.x + blockIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1664525u + 1013904223u;
    if (x == 12345u) out[0] = x;
}
__global__ void touch_kernel(uint4* p, size_t n16) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) { uint4 v = p[i]; v.x ^= 1; p[i] = v; }
}
int main() {
    const size_t total = (size_t)544 << 20;
    void *h = nullptr, *dev = nullptr; uint32_t* dummy = nullptr;
    hipHostMalloc(&h, total, 0); memset(h, 1, total);
    hipMalloc(&dev, total); hipMalloc(&dummy, 4);
    hipStream_t comp, copy;
    hipStreamCreate(&comp); hipStreamCreateWithFlags(&copy, hipStreamNonBlocking);
    hipEvent_t e0, e1, ec;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreateWithFlags(&ec, hipEventDisableTiming);
    hipEvent_t et[64], en[64];
    for (auto& e : et) hipEventCreate(&e);
    for (auto& e : en) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    const size_t sizes[] = {(size_t)16 << 20, (size_t)32 << 20, (size_t)64 << 20, (size_t)136 << 20, total};
    for (int busy = 0; busy < 2; ++busy)
        for (int pattern = 0; pattern < 6; ++pattern)
            for (size_t piece : sizes) {
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    hipDeviceSynchronize();
                    if (busy) for (int k = 0; k < 40; ++k) hipLaunchKernelGGL(spin_kernel, dim3(4096), dim3(256), 0, comp, dummy, 20000);
                    hipEventRecord(e0, copy);
                    for (size_t off = 0; off < total; off += piece) {
                        const size_t b = std::min(piece, total - off);
                        if (pattern == 1) { hipEventRecord(ec, comp); hipStreamWaitEvent(copy, ec, 0); }
                        hipMemcpyAsync((char*)dev + off, (char*)h + off, b, hipMemcpyHostToDevice, copy);
                        if (pattern == 3) { hipEventRecord(et[(off / piece) % 64], copy); }
                        if (pattern == 4) { hipEventRecord(en[(off / piece) % 64], copy); }
                        if (pattern == 5) { hipEventRecord(en[(off / piece) % 64], copy); hipStreamWaitEvent(comp, en[(off / piece) % 64], 0); }
                        if (pattern == 2) hipLaunchKernelGGL(touch_kernel, dim3((unsigned)((b / 16 + 255) / 256)), dim3(256), 0, copy, (uint4*)((char*)dev + off), b / 16);
                    }
                    hipEventRecord(e1, copy);
                    hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    best = std::min(best, ms);
                }
                printf("compute stream %s, %-34s pieces of %4zu MB: %5.1f GB/s\n", busy ? "busy" : "idle",
                       pattern == 0 ? "copies back to back" : pattern == 1 ? "each copy behind a compute event" : pattern == 2 ? "a kernel after every copy" : pattern == 3 ? "a TIMING event after every copy" : pattern == 4 ? "a no-timing event after every copy" : "no-timing event + compute waits", piece >> 20, total / best * 1e-6);
            }
    return 0;
}
