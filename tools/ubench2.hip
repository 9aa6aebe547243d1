// Throughput micro-benchmark, take 2: 8 waves per SIMD, 8 independent chains per lane, compiler-visible C so that no
// artificial dependency or VCC hazard serialises the stream. Prints cycles per wave-instruction per SIMD at the measured clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 2048

template <int OP>
__global__ void __launch_bounds__(256) bench(uint64_t* out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + threadIdx.x;
    uint64_t acc[8];
    uint32_t x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { acc[k] = a + k; x[k] = b + k; }
    long long t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (OP == 0) acc[k] = (uint64_t)(uint32_t)acc[k] * b + acc[k];        // v_mad_u64_u32
                else if (OP == 1) x[k] = (x[k] ^ a) + b;                               // 2 full-rate ops (xor + add)
                else if (OP == 6) { unsigned co; x[k] = __builtin_addc(x[k], b, (unsigned)(x[(k+1)&7] & 1), &co); x[(k+1)&7] ^= co; }
                else if (OP == 7) acc[k] = acc[k] + (((uint64_t)b << 32) | a);         // 64-bit add
                else if (OP == 2) x[k] = x[k] * b + a;                                 // v_mad_u32 (mul_lo + add) 
                else if (OP == 3) x[k] = __builtin_amdgcn_alignbit(x[k], x[(k + 1) & 7], 7);
                else if (OP == 4) x[k] = __umulhi(x[k], b) + a;                        // v_mul_hi_u32 (+add)
                else if (OP == 5) x[k] = (x[k] & 0xffffff) * (b & 0xffffff) + a;       // v_mad_u32_u24
            }
        }
    }
    long long t1 = clock64();
    uint64_t r = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) r += acc[k] + x[k];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * 256] = (uint64_t)(t1 - t0);
}

template <int OP>
void run(const char* name, uint64_t* d_out, double ops_per_item) {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    dim3 grid(cus * 8), block(256);  // 8 blocks x 4 waves = 32 waves per CU = 8 per SIMD
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<OP>, grid, block, 0, 0, d_out, 12345u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(bench<OP>, grid, block, 0, 0, d_out, 12345u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double items_per_simd = (double)ITERS * 4 * 8 * 8;  // wave-items issued on one SIMD (8 waves)
    uint64_t ticks = 0;
    (void)hipMemcpy(&ticks, d_out + (size_t)grid.x * 256, 8, hipMemcpyDeviceToHost);
    printf("%-28s %8.3f ms  %7.2f ns/wave-item/SIMD  | clock64 ticks of wave 0: %llu -> %.2f ticks per wave-item per SIMD, %.0f MHz tick rate (%.0f ops/item)\n",
           name, ms, ms * 1e6 / items_per_simd, (unsigned long long)ticks, (double)ticks / items_per_simd, ticks / (ms * 1e3), ops_per_item);
}

int main() {
    uint64_t* d; (void)hipMalloc(&d, 256 * 8 * 256 * 8 * 2 + 64);
    run<1>("shift+xor (2 full-rate ops)", d, 2); run<0>("v_mad_u64_u32", d, 1); run<2>("mul_lo+add (v_mad_u32?)", d, 1);
    run<3>("v_alignbit_b32", d, 1); run<4>("v_mul_hi_u32 + add", d, 2); run<5>("v_mad_u32_u24 (+2 and)", d, 3); run<6>("addc + xor", d, 2); run<7>("64-bit add", d, 1);
    return 0;
}
