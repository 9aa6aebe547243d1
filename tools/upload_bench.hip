// How fast can one column group of a row-major host trace (n rows x 34 columns x 32 B, pageable memory) reach the GPU?
// Times the candidates for commit_trace_pipelined on the GPU box: usage upload_bench [log_n = 20]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void gather(const uint8_t* src, uint64_t n, size_t row_bytes, size_t off, size_t width, uint8_t* dst, unsigned T) {
    auto work = [&](uint64_t r0, uint64_t r1) {
        const uint8_t* s = src + r0 * row_bytes + off;
        uint8_t* d = dst + r0 * width;
        for (uint64_t i = r0; i < r1; ++i, s += row_bytes, d += width) std::memcpy(d, s, width);
    };
    std::vector<std::thread> th;
    const uint64_t per = (n + T - 1) / T;
    for (unsigned t = 1; t < T; ++t) th.emplace_back(work, std::min<uint64_t>(n, t * per), std::min<uint64_t>(n, (t + 1) * per));
    work(0, std::min<uint64_t>(n, per));
    for (auto& x : th) x.join();
}
// zero-copy: the kernel reads the (registered) host rows itself and writes the row-major column group to device memory
__global__ void __launch_bounds__(256) pull_kernel(const uint4* __restrict__ host_rows, uint64_t n, uint32_t row_u4, uint32_t off_u4, uint32_t w_u4, uint4* __restrict__ out) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;   // one 16-byte piece each
    if (idx >= n * w_u4) return;
    const uint64_t row = idx / w_u4, k = idx - row * w_u4;
    out[idx] = host_rows[row * row_u4 + off_u4 + k];
}
int main(int argc, char** argv) {
    const int logn = argc > 1 ? atoi(argv[1]) : 20;
    const uint64_t n = 1ull << logn;
    const uint32_t cols = 34, gc = 7;
    const size_t row_bytes = cols * 32, total = n * row_bytes, chunk = n * gc * 32;
    uint8_t* host = (uint8_t*)malloc(total);
    for (size_t i = 0; i < total; i += 4096) host[i] = (uint8_t)i;   // touch every page
    memset(host, 1, total);
    uint8_t *dev, *land, *pinned;
    CHECK(hipMalloc(&dev, total)); CHECK(hipMalloc(&land, chunk)); CHECK(hipHostMalloc(&pinned, chunk, hipHostMallocDefault));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        CHECK(hipMemcpyAsync(dev, host, total, hipMemcpyHostToDevice, st)); CHECK(hipStreamSynchronize(st));
        printf("whole trace, one pageable hipMemcpyAsync            %7.2f ms  (%.1f GB/s)\n", now() - t0, total / (now() - t0) / 1e6);
    }
    for (unsigned T : {4u, 8u, 16u, 32u, 64u}) {
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) { double t0 = now(); gather(host, n, row_bytes, 7 * 32, gc * 32, pinned, T); best = std::min(best, now() - t0); }
        printf("CPU gather of one 7-column group, %2u threads         %7.2f ms  (%.1f GB/s written)\n", T, best, chunk / best / 1e6);
    }
    {
        double t0 = now();
        CHECK(hipMemcpyAsync(land, pinned, chunk, hipMemcpyHostToDevice, st)); CHECK(hipStreamSynchronize(st));
        printf("pinned chunk -> device (DMA)                         %7.2f ms  (%.1f GB/s)\n", now() - t0, chunk / (now() - t0) / 1e6);
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        CHECK(hipMemcpy2DAsync(land, gc * 32, host + 7 * 32, row_bytes, gc * 32, n, hipMemcpyHostToDevice, st)); CHECK(hipStreamSynchronize(st));
        printf("hipMemcpy2DAsync from pageable rows                  %7.2f ms  (%.1f GB/s)\n", now() - t0, chunk / (now() - t0) / 1e6);
    }
    {
        double t0 = now();
        CHECK(hipHostRegister(host, total, hipHostRegisterDefault));
        printf("hipHostRegister of the whole trace                   %7.2f ms\n", now() - t0);
        uint8_t* hdev = nullptr;
        CHECK(hipHostGetDevicePointer((void**)&hdev, host, 0));
        for (int rep = 0; rep < 2; ++rep) {
            t0 = now();
            CHECK(hipMemcpy2DAsync(land, gc * 32, host + 7 * 32, row_bytes, gc * 32, n, hipMemcpyHostToDevice, st)); CHECK(hipStreamSynchronize(st));
            printf("hipMemcpy2DAsync from registered rows                %7.2f ms  (%.1f GB/s)\n", now() - t0, chunk / (now() - t0) / 1e6);
        }
        for (int rep = 0; rep < 3; ++rep) {
            t0 = now();
            const uint64_t pieces = n * gc * 2;
            hipLaunchKernelGGL(pull_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, (const uint4*)hdev, n, cols * 2, 7 * 2, gc * 2, (uint4*)land);
            CHECK(hipStreamSynchronize(st));
            printf("zero-copy pull kernel from registered rows           %7.2f ms  (%.1f GB/s)\n", now() - t0, chunk / (now() - t0) / 1e6);
        }
        t0 = now();
        CHECK(hipHostUnregister(host));
        printf("hipHostUnregister                                    %7.2f ms\n", now() - t0);
    }
    return 0;
}
