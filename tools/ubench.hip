// Instruction-throughput micro-benchmark for the integer ops the Stark252 field multiply is built from (gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -o /tmp/ubench ; prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 4096
#define UNROLL 16

template <int OP>
__global__ void __launch_bounds__(256) bench(uint32_t* out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + threadIdx.x, c = 7, d = 9;
    uint64_t acc0 = a, acc1 = b, acc2 = c, acc3 = d;
    double f0 = a, f1 = b, f2 = 1.0000001, f3 = 0.5;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (OP == 0) {  // v_mad_u64_u32, 4 independent chains
                asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3"
                             : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(a), "v"(b) : "vcc");
            } else if (OP == 1) {  // v_mul_lo_u32
                asm volatile("v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));
            } else if (OP == 2) {  // v_mul_hi_u32
                asm volatile("v_mul_hi_u32 %0, %0, %4\n\tv_mul_hi_u32 %1, %1, %4\n\tv_mul_hi_u32 %2, %2, %4\n\tv_mul_hi_u32 %3, %3, %4"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));
            } else if (OP == 3) {  // v_fma_f64
                asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f2 * 0 + 1.0000001), "v"(0.25));
            } else if (OP == 4) {  // v_lshl_add_u64
                asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n\tv_lshl_add_u64 %1, %1, 0, %4\n\tv_lshl_add_u64 %2, %2, 0, %4\n\tv_lshl_add_u64 %3, %3, 0, %4"
                             : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(acc0 | 1));
            } else if (OP == 5) {  // v_add_co_u32 + v_addc_co_u32 pairs
                asm volatile("v_add_co_u32 %0, vcc, %0, %4\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc\n\tv_add_co_u32 %2, vcc, %2, %4\n\tv_addc_co_u32 %3, vcc, %3, %4, vcc"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed) : "vcc");
            } else if (OP == 6) {  // v_mad_u32_u24
                asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n\tv_mad_u32_u24 %1, %1, %4, %2\n\tv_mad_u32_u24 %2, %2, %4, %3\n\tv_mad_u32_u24 %3, %3, %4, %0"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));
            } else if (OP == 7) {  // v_mul_hi_u32_u24
                asm volatile("v_mul_hi_u32_u24 %0, %0, %4\n\tv_mul_hi_u32_u24 %1, %1, %4\n\tv_mul_hi_u32_u24 %2, %2, %4\n\tv_mul_hi_u32_u24 %3, %3, %4"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));
            } else if (OP == 8) {  // v_xor_b32 (full-rate reference)
                asm volatile("v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_xor_b32 %2, %2, %4\n\tv_xor_b32 %3, %3, %4"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));
            } else if (OP == 9) {  // v_alignbit_b32 (64-bit rotate half)
                asm volatile("v_alignbit_b32 %0, %0, %1, 7\n\tv_alignbit_b32 %1, %1, %2, 7\n\tv_alignbit_b32 %2, %2, %3, 7\n\tv_alignbit_b32 %3, %3, %0, 7"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (OP == 10) {  // v_mad_u64_u32 single dependent chain (latency)
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0"
                             : "+v"(acc0) : "v"(a), "v"(b) : "vcc");
            } else if (OP == 11) {  // v_bfi / v_and_or (3-op logic)
                asm volatile("v_bfi_b32 %0, %4, %0, %1\n\tv_bfi_b32 %1, %4, %1, %2\n\tv_bfi_b32 %2, %4, %2, %3\n\tv_bfi_b32 %3, %4, %3, %0"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ (uint32_t)acc0 ^ (uint32_t)acc1 ^ (uint32_t)acc2 ^ (uint32_t)acc3 ^ (uint32_t)(f0 + f1 + f2 + f3);
}

template <int OP>
void run(const char* name, uint32_t* d_out) {
    int dev_cus = 256;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); dev_cus = prop.multiProcessorCount;
    int waves_per_simd = 4;
    dim3 grid(dev_cus * waves_per_simd), block(256);  // 256 threads = 4 waves = 1 per SIMD per block
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<OP>, grid, block, 0, 0, d_out, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(bench<OP>, grid, block, 0, 0, d_out, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts_per_simd = (double)ITERS * UNROLL * 4 * waves_per_simd;  // wave-instructions issued on one SIMD
    double clk = prop.clockRate * 1e3;  // Hz
    printf("%-22s %8.3f ms  %6.2f cycles/wave-inst/SIMD (at %.0f MHz nominal)\n", name, ms, ms * 1e-3 * clk / insts_per_simd, clk / 1e6);
}

int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
    run<8>("v_xor_b32", d); run<0>("v_mad_u64_u32 x4", d); run<10>("v_mad_u64_u32 chain", d); run<1>("v_mul_lo_u32", d); run<2>("v_mul_hi_u32", d);
    run<3>("v_fma_f64", d); run<4>("v_lshl_add_u64", d); run<5>("v_add_co/addc_co", d); run<6>("v_mad_u32_u24", d);
    run<7>("v_mul_hi_u32_u24", d); run<9>("v_alignbit_b32", d); run<11>("v_bfi_b32", d);
    return 0;
}
