// Does the Montgomery product pay for every extra VALU instruction?  fe_mul_lazy with N extra independent v_mov_b32 per product
// (asm volatile, so they stay): if the time grows by ~2.8 cycles per move the product is issue-bound instruction for instruction
// and the 48 moves the compiler emits are worth removing; if it does not, they sit in hazard slots.  Registers only, 8 waves/SIMD.
#include "../../lambdaworks_cairo_prover_amd/csrc/fp.h"
#include <cstdio>
#ifndef ITERS
#define ITERS 8192
#endif
template <int EXTRA>
__global__ void __launch_bounds__(256) k(fe* out, const fe* in) {
    fe x = in[threadIdx.x & 63], y = in[(threadIdx.x + 7) & 63];
    uint32_t d0 = threadIdx.x, d1 = 0, d2 = 0, d3 = 0;
    for (int it = 0; it < ITERS; ++it) {
        x = fe_mul_lazy(x, y);
        x.v[7] &= 0x0fffffffu;
#pragma unroll
        for (int e = 0; e < EXTRA; e += 4) {
            asm volatile("v_mov_b32 %0, %1" : "=v"(d1) : "v"(d0));
            asm volatile("v_mov_b32 %0, %1" : "=v"(d2) : "v"(d1));
            asm volatile("v_mov_b32 %0, %1" : "=v"(d3) : "v"(d2));
            asm volatile("v_mov_b32 %0, %1" : "=v"(d0) : "v"(d3));
        }
    }
    x.v[0] ^= d0;
    out[blockIdx.x * 256 + threadIdx.x] = x;
}
template <int EXTRA>
void run(fe* d_out, fe* d_in) {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    dim3 grid(prop.multiProcessorCount * 8), block(256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<EXTRA>), grid, block, 0, 0, d_out, d_in); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<EXTRA>), grid, block, 0, 0, d_out, d_in);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)grid.x * 256 * ITERS;
    printf("fe_mul_lazy + %3d extra v_mov: %8.3f ms  %8.2f G products/s\n", EXTRA, ms, ops / ms / 1e6);
}
int main() {
    fe h[64];
    for (int i = 0; i < 64; ++i) for (int j = 0; j < 8; ++j) h[i].v[j] = 0x01234567u * (i + 3) + 0x9e3779b9u * j + (j == 7 ? 0 : 0x80000000u);
    for (int i = 0; i < 64; ++i) h[i].v[7] &= 0x07ffffff;
    fe *d_in, *d_out; (void)hipMalloc(&d_in, sizeof(h)); (void)hipMalloc(&d_out, sizeof(fe) * 256 * 8 * 256);
    (void)hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>(d_out, d_in); run<16>(d_out, d_in); run<32>(d_out, d_in); run<64>(d_out, d_in); run<128>(d_out, d_in); run<0>(d_out, d_in);
    return 0;
}
