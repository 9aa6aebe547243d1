// Round 6: two independent Montgomery products with their carry chains interleaved INSTRUCTION BY INSTRUCTION (every statement of the
// CIOS row written for product 1 and product 2 in turn), with and without persistent addend pairs, under scheduler flags that keep the
// source order.  Question: can the second chain fill the two wait states gfx950 wants between a VALU write and a VALU read of a carry
// (s_nop / v_mov fillers today), and does that buy issue time?  Registers only, 8 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 [-mllvm -enable-misched=0] tools/experiments/ubench_mul3.hip -o tools/bin/ubench_mul3
#include "../../lambdaworks_cairo_prover_amd/csrc/fp.h"
#include <cstdio>
#include <cstdlib>
#ifndef ITERS
#define ITERS 8192
#endif
#ifndef SB
#define SB() ((void)0)
#endif

// one CIOS row of two products, statement by statement in lockstep
__device__ __forceinline__ void row2(uint32_t t[8], uint32_t s[8], uint32_t ai, uint32_t ci, const fe& b, const fe& d) {
    uint64_t D[8], E[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { D[j] = (uint64_t)ai * b.v[j] + t[j]; E[j] = (uint64_t)ci * d.v[j] + s[j]; SB(); }
    const uint32_t u0 = (uint32_t)D[0], w0 = (uint32_t)E[0];
    const uint32_t m = 0u - u0, n = 0u - w0;
    unsigned c = (u0 != 0), c1, c2, e = (w0 != 0), e1, e2;
#pragma unroll
    for (int j = 1; j <= 5; ++j) {
        t[j - 1] = SP_ADDC(D[j], D[j - 1] >> 32, c, c1); c = c1;
        s[j - 1] = SP_ADDC(E[j], E[j - 1] >> 32, e, e1); e = e1; SB();
    }
    const uint64_t m17 = (uint64_t)m * 17u, n17 = (uint64_t)n * 17u;
    const uint32_t x6 = SP_ADDC(D[6], D[5] >> 32, c, c1);
    const uint32_t y6 = SP_ADDC(E[6], E[5] >> 32, e, e1); SB();
    t[5] = SP_ADDC(x6, m17, 0u, c2);
    s[5] = SP_ADDC(y6, n17, 0u, e2); SB();
    const uint32_t x7 = SP_ADDC(D[7], D[6] >> 32, c1, c1);
    const uint32_t y7 = SP_ADDC(E[7], E[6] >> 32, e1, e1); SB();
    const uint32_t k7 = (uint32_t)(m17 >> 32) + (m << 27), l7 = (uint32_t)(n17 >> 32) + (n << 27);
    t[6] = SP_ADDC(x7, k7, c2, c2);
    s[6] = SP_ADDC(y7, l7, e2, e2); SB();
    const uint32_t x8 = SP_ADDC(D[7] >> 32, m >> 5, c1, c1);
    const uint32_t y8 = SP_ADDC(E[7] >> 32, n >> 5, e1, e1); SB();
    t[7] = SP_ADDC(x8, 0u, c2, c2);
    s[7] = SP_ADDC(y8, 0u, e2, e2); SB();
}
__device__ __forceinline__ void mul2_lockstep(const fe& a1, const fe& b1, const fe& a2, const fe& b2, fe& r1, fe& r2) {
    uint32_t t1[8], t2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { t1[j] = 0; t2[j] = 0; }
#pragma unroll
    for (int i = 0; i < 8; ++i) row2(t1, t2, a1.v[i], a2.v[i], b1, b2);
#pragma unroll
    for (int j = 0; j < 8; ++j) { r1.v[j] = t1[j]; r2.v[j] = t2[j]; }
}


// the same, software-pipelined: the multiply-adds of column j sit between the carry adds of columns j - 2 and j - 1, so that a chain's
// consecutive carry adds are three instructions apart (the other chain's add and two multiply-adds) - no s_nop, no filler
template <bool PAIRS>
__device__ __forceinline__ void row2p(uint64_t T[8], uint64_t S[8], uint32_t ai, uint32_t ci, const fe& b, const fe& d) {
    uint64_t D[8], E[8];
    uint32_t t[8], s[8];
    unsigned c, c1, c2, e, e1, e2;
    D[0] = (uint64_t)ai * b.v[0] + (PAIRS ? T[0] : (uint64_t)(uint32_t)T[0]); E[0] = (uint64_t)ci * d.v[0] + (PAIRS ? S[0] : (uint64_t)(uint32_t)S[0]); SB();
    D[1] = (uint64_t)ai * b.v[1] + (PAIRS ? T[1] : (uint64_t)(uint32_t)T[1]); E[1] = (uint64_t)ci * d.v[1] + (PAIRS ? S[1] : (uint64_t)(uint32_t)S[1]); SB();
    const uint32_t u0 = (uint32_t)D[0], w0 = (uint32_t)E[0];
    const uint32_t m = 0u - u0, n = 0u - w0;
    c = (u0 != 0); e = (w0 != 0); SB();
#pragma unroll
    for (int j = 1; j <= 5; ++j) {
        D[j + 1] = (uint64_t)ai * b.v[j + 1] + (PAIRS ? T[j + 1] : (uint64_t)(uint32_t)T[j + 1]);
        E[j + 1] = (uint64_t)ci * d.v[j + 1] + (PAIRS ? S[j + 1] : (uint64_t)(uint32_t)S[j + 1]); SB();
        t[j - 1] = SP_ADDC(D[j], D[j - 1] >> 32, c, c1); c = c1;
        s[j - 1] = SP_ADDC(E[j], E[j - 1] >> 32, e, e1); e = e1; SB();
    }
    D[7] = (uint64_t)ai * b.v[7] + (PAIRS ? T[7] : (uint64_t)(uint32_t)T[7]); E[7] = (uint64_t)ci * d.v[7] + (PAIRS ? S[7] : (uint64_t)(uint32_t)S[7]); SB();
    const uint32_t x6 = SP_ADDC(D[6], D[5] >> 32, c, c1);
    const uint32_t y6 = SP_ADDC(E[6], E[5] >> 32, e, e1); SB();
    const uint64_t m17 = (uint64_t)m * 17u, n17 = (uint64_t)n * 17u; SB();
    const uint32_t x7 = SP_ADDC(D[7], D[6] >> 32, c1, c1);
    const uint32_t y7 = SP_ADDC(E[7], E[6] >> 32, e1, e1); SB();
    const uint32_t k7 = (uint32_t)(m17 >> 32) + (m << 27), l7 = (uint32_t)(n17 >> 32) + (n << 27); SB();
    const uint32_t x8 = SP_ADDC(D[7] >> 32, m >> 5, c1, c1);
    const uint32_t y8 = SP_ADDC(E[7] >> 32, n >> 5, e1, e1); SB();
    t[5] = SP_ADDC(x6, m17, 0u, c2);
    s[5] = SP_ADDC(y6, n17, 0u, e2); SB();
    t[6] = SP_ADDC(x7, k7, c2, c2);
    s[6] = SP_ADDC(y7, l7, e2, e2); SB();
    t[7] = SP_ADDC(x8, 0u, c2, c2);
    s[7] = SP_ADDC(y8, 0u, e2, e2); SB();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (PAIRS) { T[j] = (T[j] & 0xFFFFFFFF00000000ull) | t[j]; S[j] = (S[j] & 0xFFFFFFFF00000000ull) | s[j]; }
        else { T[j] = t[j]; S[j] = s[j]; }
    }
}
template <bool PAIRS>
__device__ __forceinline__ void mul2_pipelined(const fe& a1, const fe& b1, const fe& a2, const fe& b2, fe& r1, fe& r2) {
    uint64_t T[8], S[8];
    if (PAIRS) {
        asm("v_mov_b64 %0, 0 ; T0" : "=v"(T[0])); asm("v_mov_b64 %0, 0 ; T1" : "=v"(T[1])); asm("v_mov_b64 %0, 0 ; T2" : "=v"(T[2])); asm("v_mov_b64 %0, 0 ; T3" : "=v"(T[3]));
        asm("v_mov_b64 %0, 0 ; T4" : "=v"(T[4])); asm("v_mov_b64 %0, 0 ; T5" : "=v"(T[5])); asm("v_mov_b64 %0, 0 ; T6" : "=v"(T[6])); asm("v_mov_b64 %0, 0 ; T7" : "=v"(T[7]));
        asm("v_mov_b64 %0, 0 ; S0" : "=v"(S[0])); asm("v_mov_b64 %0, 0 ; S1" : "=v"(S[1])); asm("v_mov_b64 %0, 0 ; S2" : "=v"(S[2])); asm("v_mov_b64 %0, 0 ; S3" : "=v"(S[3]));
        asm("v_mov_b64 %0, 0 ; S4" : "=v"(S[4])); asm("v_mov_b64 %0, 0 ; S5" : "=v"(S[5])); asm("v_mov_b64 %0, 0 ; S6" : "=v"(S[6])); asm("v_mov_b64 %0, 0 ; S7" : "=v"(S[7]));
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { T[j] = 0; S[j] = 0; }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) row2p<PAIRS>(T, S, a1.v[i], a2.v[i], b1, b2);
#pragma unroll
    for (int j = 0; j < 8; ++j) { r1.v[j] = (uint32_t)T[j]; r2.v[j] = (uint32_t)S[j]; }
}


// carry-free limb arithmetic: the carry out of x + y + cin is the top bit of (x & y) | ((x | y) & ~s) - one v_bitop3_b32 and a shift, no
// lane mask through the scalar operand path.  Four "cheap" VALU instructions per limb instead of one carry instruction.
__device__ __forceinline__ uint32_t cf_carry(uint32_t x, uint32_t y, uint32_t s) {   // (x & y) | ((x | y) & ~s) >> 31
    return __builtin_amdgcn_bitop3_b32(x, y, s, 0xd4) >> 31;
}
__device__ __forceinline__ uint32_t cf_borrow(uint32_t x, uint32_t y, uint32_t d) {  // (~x & y) | (~(x ^ y) & d) >> 31
    return __builtin_amdgcn_bitop3_b32(x, y, d, 0x8e) >> 31;
}
__device__ __forceinline__ fe cf_add_raw(const fe& a, const fe& b) {
    fe r; uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const uint32_t s = a.v[j] + b.v[j] + c; c = cf_carry(a.v[j], b.v[j], s); r.v[j] = s; }
    return r;
}
__device__ __forceinline__ fe cf_sub_add_2p(const fe& a, const fe& b) {   // a - b + 2p
    fe d; uint32_t br = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const uint32_t x = a.v[j] - b.v[j] - br; br = cf_borrow(a.v[j], b.v[j], x); d.v[j] = x; }
    const uint32_t P2[8] = {2u, 0u, 0u, 0u, 0u, 0u, 34u, 0x10000000u};
    fe r; uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const uint32_t s = d.v[j] + P2[j] + c; c = cf_carry(d.v[j], P2[j], s); r.v[j] = s; }
    return r;
}


// Reduction of a row through two multiply-adds (needs the carry-out of v_mad_u64_u32, which only inline assembly exposes):
//   E = 17 m + {x7, x6} (carry-out cE), F = 2^27 m + {x8, hi(E)}, t5 = lo(E), t6 = lo(F), t7 = hi(F) + cE
// 10 multiply-adds + 9 carries a row instead of 9 + 11.
__device__ __forceinline__ fe fe_mul_lazy_v2(const fe& a, const fe& b) {
    uint32_t t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = 0;
    const uint32_t two27 = 0x08000000u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t D[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) D[j] = (uint64_t)a.v[i] * b.v[j] + t[j];
        const uint32_t u0 = (uint32_t)D[0];
        const uint32_t m = 0u - u0;
        unsigned c = (u0 != 0), c1;
#pragma unroll
        for (int j = 1; j <= 5; ++j) { t[j - 1] = SP_ADDC(D[j], D[j - 1] >> 32, c, c1); c = c1; }
        const uint32_t x6 = SP_ADDC(D[6], D[5] >> 32, c, c1); c = c1;
        const uint32_t x7 = SP_ADDC(D[7], D[6] >> 32, c, c1); c = c1;
        const uint32_t x8 = SP_ADDC(D[7] >> 32, 0u, c, c1);
        const uint64_t X = ((uint64_t)x7 << 32) | x6;
        uint64_t E, cE, F, cF;
        asm("v_mad_u64_u32 %0, %1, %2, 17, %3" : "=v"(E), "=s"(cE) : "v"(m), "v"(X));
        const uint64_t Y = ((uint64_t)x8 << 32) | (uint32_t)(E >> 32);
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(F), "=s"(cF) : "v"(m), "s"(two27), "v"(Y));
        uint32_t t7; uint64_t dummy;
        asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(t7), "=s"(dummy) : "v"((uint32_t)(F >> 32)), "s"(cE));
        t[5] = (uint32_t)E; t[6] = (uint32_t)F; t[7] = t7;
    }
    fe r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r.v[j] = t[j];
    return r;
}

template <int OP>
__global__ void __launch_bounds__(256) k(fe* out, const fe* in) {
    fe x = in[threadIdx.x & 63], y = in[(threadIdx.x + 7) & 63], x2 = in[(threadIdx.x + 13) & 63], y2 = in[(threadIdx.x + 29) & 63];
    for (int it = 0; it < ITERS; ++it) {
        if (OP == 0) { x = fe_mul_lazy(x, y); x2 = fe_mul_lazy(x2, y2); x.v[7] &= 0x0fffffffu; x2.v[7] &= 0x0fffffffu; }
        else if (OP == 1) { fe r1, r2; mul2_lockstep(x, y, x2, y2, r1, r2); x = r1; x2 = r2; x.v[7] &= 0x0fffffffu; x2.v[7] &= 0x0fffffffu; }
        else if (OP == 4) { fe r1, r2; mul2_pipelined<false>(x, y, x2, y2, r1, r2); x = r1; x2 = r2; x.v[7] &= 0x0fffffffu; x2.v[7] &= 0x0fffffffu; }
        else if (OP == 5) { fe r1, r2; mul2_pipelined<true>(x, y, x2, y2, r1, r2); x = r1; x2 = r2; x.v[7] &= 0x0fffffffu; x2.v[7] &= 0x0fffffffu; }
        else if (OP == 6 || OP == 7) {  // two butterflies, products pipelined (7: persistent pairs)
            fe t, t2;
            if (OP == 6) mul2_pipelined<false>(x, y, x2, y2, t, t2); else mul2_pipelined<true>(x, y, x2, y2, t, t2);
            fe u = fe_add_raw(y, t); x = fe_sub_add_2p(y, t); y = u; x.v[7] &= 0x0fffffffu; y.v[7] &= 0x07ffffffu;
            fe u2 = fe_add_raw(y2, t2); x2 = fe_sub_add_2p(y2, t2); y2 = u2; x2.v[7] &= 0x0fffffffu; y2.v[7] &= 0x07ffffffu;
        }
        else if (OP == 8) {   // two butterflies, sequential, carry-free add / sub
            fe t = fe_mul_lazy(x, y); fe u = cf_add_raw(y, t); x = cf_sub_add_2p(y, t); y = u; x.v[7] &= 0x0fffffffu; y.v[7] &= 0x07ffffffu;
            fe t2 = fe_mul_lazy(x2, y2); fe u2 = cf_add_raw(y2, t2); x2 = cf_sub_add_2p(y2, t2); y2 = u2; x2.v[7] &= 0x0fffffffu; y2.v[7] &= 0x07ffffffu;
        }
        else if (OP == 9) { x = fe_mul_lazy_v2(x, y); x2 = fe_mul_lazy_v2(x2, y2); x.v[7] &= 0x0fffffffu; x2.v[7] &= 0x0fffffffu; }
        else if (OP == 10) {
            fe t = fe_mul_lazy_v2(x, y); fe u = fe_add_raw(y, t); x = fe_sub_add_2p(y, t); y = u; x.v[7] &= 0x0fffffffu; y.v[7] &= 0x07ffffffu;
            fe t2 = fe_mul_lazy_v2(x2, y2); fe u2 = fe_add_raw(y2, t2); x2 = fe_sub_add_2p(y2, t2); y2 = u2; x2.v[7] &= 0x0fffffffu; y2.v[7] &= 0x07ffffffu;
        }
        else if (OP == 2) {   // two butterflies, sequential
            fe t = fe_mul_lazy(x, y); fe u = fe_add_raw(y, t); x = fe_sub_add_2p(y, t); y = u; x.v[7] &= 0x0fffffffu; y.v[7] &= 0x07ffffffu;
            fe t2 = fe_mul_lazy(x2, y2); fe u2 = fe_add_raw(y2, t2); x2 = fe_sub_add_2p(y2, t2); y2 = u2; x2.v[7] &= 0x0fffffffu; y2.v[7] &= 0x07ffffffu;
        } else if (OP == 3) {  // two butterflies, products in lockstep
            fe t, t2; mul2_lockstep(x, y, x2, y2, t, t2);
            fe u = fe_add_raw(y, t); x = fe_sub_add_2p(y, t); y = u; x.v[7] &= 0x0fffffffu; y.v[7] &= 0x07ffffffu;
            fe u2 = fe_add_raw(y2, t2); x2 = fe_sub_add_2p(y2, t2); y2 = u2; x2.v[7] &= 0x0fffffffu; y2.v[7] &= 0x07ffffffu;
        }
    }
    fe r;
    for (int j = 0; j < 8; ++j) r.v[j] = x.v[j] ^ x2.v[j] ^ y.v[j] ^ y2.v[j];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
static int g_waves_per_simd = 8;     // argv[1]: resident waves per SIMD (256-thread blocks: one wave per SIMD each)
template <int OP>
void run(const char* name, fe* d_out, fe* d_in, uint32_t* sig) {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    dim3 grid(prop.multiProcessorCount * g_waves_per_simd), block(256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP>), grid, block, 0, 0, d_out, d_in); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP>), grid, block, 0, 0, d_out, d_in);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    fe h; (void)hipMemcpy(&h, d_out + 5, sizeof(fe), hipMemcpyDeviceToHost);
    *sig = h.v[0] ^ h.v[3];
    double ops = (double)grid.x * 256 * ITERS * 2;
    printf("%-52s %8.3f ms  %8.2f G ops/s   sig %08x\n", name, ms, ops / ms / 1e6, *sig);
}
int main(int argc, char** argv) {
    if (argc > 1) g_waves_per_simd = atoi(argv[1]);
    printf("== %d waves per SIMD\n", g_waves_per_simd);
    fe h[64];
    for (int i = 0; i < 64; ++i) for (int j = 0; j < 8; ++j) h[i].v[j] = 0x01234567u * (i + 3) + 0x9e3779b9u * j + (j == 7 ? 0 : 0x80000000u);
    for (int i = 0; i < 64; ++i) h[i].v[7] &= 0x07ffffff;
    fe *d_in, *d_out; (void)hipMalloc(&d_in, sizeof(h)); (void)hipMalloc(&d_out, sizeof(fe) * 256 * 8 * 256);
    (void)hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
    uint32_t s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10;
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("2 x fe_mul_lazy, sequential", d_out, d_in, &s0);
        run<1>("2 x fe_mul_lazy, lockstep", d_out, d_in, &s1);
        run<2>("2 butterflies, sequential", d_out, d_in, &s2);
        run<3>("2 butterflies, products in lockstep", d_out, d_in, &s3);
        run<4>("2 x product, pipelined lockstep", d_out, d_in, &s4);
        run<5>("2 x product, pipelined lockstep, persistent pairs", d_out, d_in, &s5);
        run<6>("2 butterflies, pipelined lockstep", d_out, d_in, &s6);
        run<7>("2 butterflies, pipelined lockstep, persistent pairs", d_out, d_in, &s7);
        run<8>("2 butterflies, sequential, carry-free add / sub", d_out, d_in, &s8);
        printf("carry-free %s\n", s8 == s2 ? "MATCH" : "DIFFER");
        run<9>("2 x product, reduction through two multiply-adds", d_out, d_in, &s9);
        run<10>("2 butterflies, reduction through two multiply-adds", d_out, d_in, &s10);
        printf("two-mad reduction %s\n", (s9 == s0 && s10 == s2) ? "MATCH" : "DIFFER");
        printf("results %s\n", (s0 == s1 && s0 == s4 && s0 == s5 && s2 == s3 && s2 == s6 && s2 == s7) ? "MATCH" : "DIFFER");
    }
    return 0;
}
