// Two independent Montgomery products per thread, rows interleaved at source level: does the VALU-writes-VCC -> VALU-reads-VCC
// hazard (2 wait states on gfx950: the compiler pads the carry chains with s_nop) cost issue time that a second, independent
// carry chain can fill?  Registers only, 8 waves per SIMD.
#include "../../lambdaworks_cairo_prover_amd/csrc/fp.h"
#include <cstdio>
#ifndef ITERS
#define ITERS 8192
#endif
struct Row { uint32_t t[8]; };
__device__ __forceinline__ void mul_row(uint32_t t[8], uint32_t ai, const fe& b) {
    uint64_t D[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) D[j] = (uint64_t)ai * b.v[j] + t[j];
    const uint32_t u0 = (uint32_t)D[0];
    const uint32_t m = 0u - u0;
    unsigned c = (u0 != 0), c1, c2;
#pragma unroll
    for (int j = 1; j <= 5; ++j) { t[j - 1] = SP_ADDC(D[j], D[j - 1] >> 32, c, c1); c = c1; }
    const uint64_t m17 = (uint64_t)m * 17u;
    const uint32_t x6 = SP_ADDC(D[6], D[5] >> 32, c, c1);
    t[5] = SP_ADDC(x6, m17, 0u, c2);
    const uint32_t x7 = SP_ADDC(D[7], D[6] >> 32, c1, c1);
    const uint32_t k7 = (uint32_t)(m17 >> 32) + (m << 27);
    t[6] = SP_ADDC(x7, k7, c2, c2);
    const uint32_t x8 = SP_ADDC(D[7] >> 32, m >> 5, c1, c1);
    t[7] = SP_ADDC(x8, 0u, c2, c2);
}
// rows of the two products alternate
__device__ __forceinline__ void mul2(const fe& a1, const fe& b1, const fe& a2, const fe& b2, fe& r1, fe& r2) {
    uint32_t t1[8], t2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { t1[j] = 0; t2[j] = 0; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { mul_row(t1, a1.v[i], b1); mul_row(t2, a2.v[i], b2); }
#pragma unroll
    for (int j = 0; j < 8; ++j) { r1.v[j] = t1[j]; r2.v[j] = t2[j]; }
}
template <int OP>
__global__ void __launch_bounds__(256) k(fe* out, const fe* in) {
    fe x = in[threadIdx.x & 63], y = in[(threadIdx.x + 7) & 63], x2 = in[(threadIdx.x + 13) & 63], y2 = in[(threadIdx.x + 29) & 63];
    for (int it = 0; it < ITERS; ++it) {
        if (OP == 0) { x = fe_mul_lazy(x, y); x2 = fe_mul_lazy(x2, y2); x.v[7] &= 0x0fffffffu; x2.v[7] &= 0x0fffffffu; }
        else if (OP == 1) { fe r1, r2; mul2(x, y, x2, y2, r1, r2); x = r1; x2 = r2; x.v[7] &= 0x0fffffffu; x2.v[7] &= 0x0fffffffu; }
        else if (OP == 2) {   // two butterflies, sequential
            fe t = fe_mul_lazy(x, y); fe u = fe_add_raw(y, t); x = fe_sub_add_2p(y, t); y = u; x.v[7] &= 0x0fffffffu; y.v[7] &= 0x07ffffffu;
            fe t2 = fe_mul_lazy(x2, y2); fe u2 = fe_add_raw(y2, t2); x2 = fe_sub_add_2p(y2, t2); y2 = u2; x2.v[7] &= 0x0fffffffu; y2.v[7] &= 0x07ffffffu;
        } else if (OP == 3) {  // two butterflies, products interleaved
            fe t, t2; mul2(x, y, x2, y2, t, t2);
            fe u = fe_add_raw(y, t); x = fe_sub_add_2p(y, t); y = u; x.v[7] &= 0x0fffffffu; y.v[7] &= 0x07ffffffu;
            fe u2 = fe_add_raw(y2, t2); x2 = fe_sub_add_2p(y2, t2); y2 = u2; x2.v[7] &= 0x0fffffffu; y2.v[7] &= 0x07ffffffu;
        }
    }
    fe r;
    for (int j = 0; j < 8; ++j) r.v[j] = x.v[j] ^ x2.v[j] ^ y.v[j] ^ y2.v[j];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int OP>
void run(const char* name, fe* d_out, fe* d_in) {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    dim3 grid(prop.multiProcessorCount * 8), block(256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP>), grid, block, 0, 0, d_out, d_in); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP>), grid, block, 0, 0, d_out, d_in);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)grid.x * 256 * ITERS * 2;
    printf("%-52s %8.3f ms  %8.2f G ops/s\n", name, ms, ops / ms / 1e6);
}
int main() {
    fe h[64];
    for (int i = 0; i < 64; ++i) for (int j = 0; j < 8; ++j) h[i].v[j] = 0x01234567u * (i + 3) + 0x9e3779b9u * j + (j == 7 ? 0 : 0x80000000u);
    for (int i = 0; i < 64; ++i) h[i].v[7] &= 0x07ffffff;
    fe *d_in, *d_out; (void)hipMalloc(&d_in, sizeof(h)); (void)hipMalloc(&d_out, sizeof(fe) * 256 * 8 * 256);
    (void)hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>("2 x fe_mul_lazy, sequential", d_out, d_in);
    run<1>("2 x fe_mul_lazy, rows interleaved", d_out, d_in);
    run<2>("2 butterflies, sequential", d_out, d_in);
    run<3>("2 butterflies, products interleaved", d_out, d_in);
    return 0;
}
