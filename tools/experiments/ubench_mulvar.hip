// Variants of the Montgomery product to find where its ~910 cycles go (registers only, 8 waves per SIMD).
#include "../../lambdaworks_cairo_prover_amd/csrc/fp.h"
#include <cstdio>
#ifndef ITERS
#define ITERS 256
#endif
// (b) lazy: no final conditional subtraction (result in [0, 2p))
__device__ __forceinline__ fe mul_lazy(const fe& a, const fe& b) {
    uint32_t t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t D[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) D[j] = (uint64_t)a.v[i] * b.v[j] + t[j];
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
    fe r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r.v[j] = t[j];
    return r;
}
// (c) products only: 64 mads, results xor-folded (not a real product) -> cost of the multiplier part alone
__device__ __forceinline__ fe mul_mads_only(const fe& a, const fe& b) {
    fe r = a;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { uint64_t d = (uint64_t)a.v[i] * b.v[j] + r.v[j]; r.v[(i + j) & 7] ^= (uint32_t)d ^ (uint32_t)(d >> 32); }
    }
    return r;
}

// (d) persistent addend pairs, opaque zero high halves (non-volatile, one distinct asm per column: hoistable / shareable)
template <int VOL>
__device__ __forceinline__ fe mul_pairs(const fe& a, const fe& b) {
    uint64_t T[8];
    if (VOL) {
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_mov_b64 %0, 0" : "=v"(T[j]));
    } else {
        asm("v_mov_b64 %0, 0 ; col 0" : "=v"(T[0])); asm("v_mov_b64 %0, 0 ; col 1" : "=v"(T[1]));
        asm("v_mov_b64 %0, 0 ; col 2" : "=v"(T[2])); asm("v_mov_b64 %0, 0 ; col 3" : "=v"(T[3]));
        asm("v_mov_b64 %0, 0 ; col 4" : "=v"(T[4])); asm("v_mov_b64 %0, 0 ; col 5" : "=v"(T[5]));
        asm("v_mov_b64 %0, 0 ; col 6" : "=v"(T[6])); asm("v_mov_b64 %0, 0 ; col 7" : "=v"(T[7]));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t D[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) D[j] = (uint64_t)a.v[i] * b.v[j] + T[j];
        const uint32_t u0 = (uint32_t)D[0];
        const uint32_t m = 0u - u0;
        unsigned c = (u0 != 0), c1, c2;
        uint32_t t[8];
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
#pragma unroll
        for (int j = 0; j < 8; ++j) T[j] = (T[j] & 0xFFFFFFFF00000000ull) | t[j];
    }
    fe r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r.v[j] = (uint32_t)T[j];
    return r;
}

// (e) 17 m from shifts and adds instead of a ninth multiply-add per row
__device__ __forceinline__ fe mul_shift17(const fe& a, const fe& b) {
    uint32_t t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t D[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) D[j] = (uint64_t)a.v[i] * b.v[j] + t[j];
        const uint32_t u0 = (uint32_t)D[0];
        const uint32_t m = 0u - u0;
        unsigned c = (u0 != 0), c1, c2, c3;
#pragma unroll
        for (int j = 1; j <= 5; ++j) { t[j - 1] = SP_ADDC(D[j], D[j - 1] >> 32, c, c1); c = c1; }
        const uint32_t m17lo = SP_ADDC(m, m << 4, 0u, c3);
        const uint32_t m17hi = (m >> 28) + c3;
        const uint32_t x6 = SP_ADDC(D[6], D[5] >> 32, c, c1);
        t[5] = SP_ADDC(x6, m17lo, 0u, c2);
        const uint32_t x7 = SP_ADDC(D[7], D[6] >> 32, c1, c1);
        const uint32_t k7 = m17hi + (m << 27);
        t[6] = SP_ADDC(x7, k7, c2, c2);
        const uint32_t x8 = SP_ADDC(D[7] >> 32, m >> 5, c1, c1);
        t[7] = SP_ADDC(x8, 0u, c2, c2);
    }
    fe r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r.v[j] = t[j];
    return r;
}
template <int OP>
__global__ void __launch_bounds__(256) k(fe* out, const fe* in) {
    fe x = in[threadIdx.x & 63], y = in[(threadIdx.x + 7) & 63];
    for (int it = 0; it < ITERS; ++it) {
        if (OP == 0) x = fe_mul(x, y);
        else if (OP == 1) x = mul_lazy(x, y);
        else if (OP == 2) x = mul_mads_only(x, y);
        else if (OP == 3) x = fe_reduce_once(fe_add(x, y));
        else if (OP == 4) x = fe_mul_lazy(x, y);
        else if (OP == 5) x = mul_pairs<1>(x, y);
        else if (OP == 6) x = mul_pairs<0>(x, y);
        else if (OP == 7) x = mul_shift17(x, y);
        else if (OP == 8) { fe t = fe_mul_lazy(x, y); fe u = fe_add_raw(y, t); x = fe_reduce_lazy_2p(fe_sub_add_2p(y, t)); y = fe_reduce_lazy_2p(u); }   // deferred-reduction butterfly (+ a fold per output so the chain stays bounded)
        else if (OP == 9) { fe t = fe_mul_lazy(x, y); fe u = fe_add_raw(y, t); x = fe_sub_add_2p(y, t); y = u; x.v[7] &= 0x0fffffffu; y.v[7] &= 0x0fffffffu; }   // the same without the folds (top bits masked)
    }
    out[blockIdx.x * 256 + threadIdx.x] = x;
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
    double ops = (double)grid.x * 256 * ITERS;
    printf("%-40s %8.3f ms  %8.2f G ops/s  (%.0f cycles per wave-op per SIMD at 2.1 GHz)\n", name, ms, ops / ms / 1e6, 2.1e9 * 64 * 1024 / (ops / ms * 1e3));
}
int main() {
    fe h[64];
    for (int i = 0; i < 64; ++i) for (int j = 0; j < 8; ++j) h[i].v[j] = 0x01234567u * (i + 3) + 0x9e3779b9u * j + (j == 7 ? 0 : 0x80000000u);
    for (int i = 0; i < 64; ++i) h[i].v[7] &= 0x07ffffff;
    fe *d_in, *d_out; (void)hipMalloc(&d_in, sizeof(h)); (void)hipMalloc(&d_out, sizeof(fe) * 256 * 8 * 256);
    (void)hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>("fe_mul (canonical result)", d_out, d_in); run<1>("fe_mul lazy (no final subtraction)", d_out, d_in);
    run<2>("64 v_mad_u64_u32 + 128 xor", d_out, d_in); run<3>("fe_add + extra reduce", d_out, d_in);
    run<4>("fp.h fe_mul_lazy", d_out, d_in); run<5>("pairs, volatile zero init", d_out, d_in); run<6>("pairs, hoistable zero init", d_out, d_in);
    run<7>("17 m by shift and add (64 multiply-adds)", d_out, d_in);
    run<8>("lazy butterfly: mul_lazy + add_raw + sub_add_2p + 2 folds", d_out, d_in);
    run<9>("lazy butterfly: mul_lazy + add_raw + sub_add_2p", d_out, d_in);
    return 0;
}
