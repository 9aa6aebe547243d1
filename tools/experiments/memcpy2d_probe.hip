// hipMemcpy2DAsync straight from a pageable row-major table (no host gather): how fast, and how long does the call hold the host?
// A fresh malloc'd table each round (a prover sees a new trace every proof).
// build: hipcc -O2 --offload-arch=gfx950 memcpy2d_probe.hip -o memcpy2d_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = 1 << 20, cols = 34, row = cols * 32;
    void* dev = nullptr; hipMalloc(&dev, n * 8 * 32);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int round = 0; round < 3; ++round) {
        uint8_t* table = (uint8_t*)malloc(n * row);
        memset(table, round + 1, n * row);
        for (int w : {2, 4, 8}) {
            for (int rep = 0; rep < 2; ++rep) {
                const size_t c0 = (size_t)(rep * 8 + w) % 24;
                hipEventRecord(e0, st);
                double t0 = now();
                hipMemcpy2DAsync(dev, (size_t)w * 32, table + c0 * 32, row, (size_t)w * 32, n, hipMemcpyHostToDevice, st);
                double t1 = now();
                hipEventRecord(e1, st);
                hipEventSynchronize(e1);
                double t2 = now();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("table %d, %d columns (%3zu MB), call %d: host held %6.2f ms, done after %6.2f ms, stream time %6.2f ms -> %5.1f GB/s\n", round, w,
                       (n * w * 32) >> 20, rep, t1 - t0, t2 - t0, ms, (double)n * w * 32 / (t2 - t0) * 1e-6);
            }
        }
        free(table);
    }
    return 0;
}
