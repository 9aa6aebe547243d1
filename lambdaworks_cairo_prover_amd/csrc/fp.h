// Stark252 prime field for gfx950 (and the host side of the library).
//
// Replaces `FieldElement<Stark252PrimeField>` (reference src/lib.rs:12-13; arithmetic in lambdaworks-math
// @ a17b951) on the device: p = 2^251 + 17*2^192 + 1, Montgomery form with R = 2^256, eight 32-bit limbs,
// least-significant limb first (one element = one aligned 32-byte little-endian integer in HBM).
//
// gfx950 has no 64x64 multiplier: the 8x8-limb product is 64 v_mad_u64_u32 (quarter rate) and dominates;
// the Montgomery reduction needs NO multiplier because p = 1 (mod 2^64):  -p^-1 = -1 (mod 2^32), so the
// per-round quotient is m = -t_0 and m*p = m + 17*m*2^192 + m*2^251 is shifts and adds on the top limbs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SP_HD __host__ __device__ __forceinline__

struct alignas(16) fe {
    uint32_t v[8];
};

// p, little-endian 32-bit limbs
#define SP_P0 0x00000001u
#define SP_P6 0x00000011u
#define SP_P7 0x08000000u

SP_HD fe fe_zero() {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = 0;
    return r;
}
// R mod p  (Montgomery one)
SP_HD fe fe_one() {
    fe r;
    r.v[0] = 0xffffffe1u; r.v[1] = 0xffffffffu; r.v[2] = 0xffffffffu; r.v[3] = 0xffffffffu;
    r.v[4] = 0xffffffffu; r.v[5] = 0xffffffffu; r.v[6] = 0xfffffdf0u; r.v[7] = 0x07ffffffu;
    return r;
}
// R^2 mod p
SP_HD fe fe_r2() {
    fe r;
    r.v[0] = 0x7e000401u; r.v[1] = 0xfffffd73u; r.v[2] = 0x330fffffu; r.v[3] = 0x00000001u;
    r.v[4] = 0xff6f8000u; r.v[5] = 0xffffffffu; r.v[6] = 0x5e008810u; r.v[7] = 0x07ffd4abu;
    return r;
}

SP_HD bool fe_is_zero(const fe& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= a.v[i];
    return o == 0;
}
SP_HD bool fe_eq(const fe& a, const fe& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= a.v[i] ^ b.v[i];
    return o == 0;
}

// 32-bit add/sub with carry: clang lowers these to v_addc_co_u32 / v_subb_co_u32 chains on gfx950
#define SP_ADDC(x, y, cin, cout) __builtin_addc((uint32_t)(x), (uint32_t)(y), (unsigned)(cin), &(cout))
#define SP_SUBC(x, y, bin, bout) __builtin_subc((uint32_t)(x), (uint32_t)(y), (unsigned)(bin), &(bout))

// r = a - p if a >= p else a   (a < 2p)
SP_HD fe fe_reduce_once(const fe& a) {
    fe d;
    unsigned br = 0, bo;
    d.v[0] = SP_SUBC(a.v[0], SP_P0, br, bo); br = bo;
#pragma unroll
    for (int i = 1; i < 6; ++i) { d.v[i] = SP_SUBC(a.v[i], 0u, br, bo); br = bo; }
    d.v[6] = SP_SUBC(a.v[6], SP_P6, br, bo); br = bo;
    d.v[7] = SP_SUBC(a.v[7], SP_P7, br, bo); br = bo;
    fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = br ? a.v[i] : d.v[i];
    return r;
}

SP_HD fe fe_add(const fe& a, const fe& b) {
    fe s;
    unsigned c = 0, co;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s.v[i] = SP_ADDC(a.v[i], b.v[i], c, co); c = co; }
    return fe_reduce_once(s);  // a+b < 2p < 2^256: no carry out
}

SP_HD fe fe_sub(const fe& a, const fe& b) {
    fe d;
    unsigned br = 0, bo;
#pragma unroll
    for (int i = 0; i < 8; ++i) { d.v[i] = SP_SUBC(a.v[i], b.v[i], br, bo); br = bo; }
    // add p back when the subtraction borrowed
    const uint32_t m = 0u - (uint32_t)br;
    fe r;
    unsigned c = 0, co;
    r.v[0] = SP_ADDC(d.v[0], m & SP_P0, c, co); c = co;
#pragma unroll
    for (int i = 1; i < 6; ++i) { r.v[i] = SP_ADDC(d.v[i], 0u, c, co); c = co; }
    r.v[6] = SP_ADDC(d.v[6], m & SP_P6, c, co); c = co;
    r.v[7] = SP_ADDC(d.v[7], m & SP_P7, c, co);
    return r;
}

SP_HD fe fe_neg(const fe& a) { return fe_sub(fe_zero(), a); }

// The reduction tail of a CIOS row on the device, through two multiply-adds (round 6).  After the merge chain a row holds x6, x7, x8 in
// words 6..8 and owes them m p = m + 17 m 2^192 + m 2^251:  E = 17 m + {x7, x6},  F = 2^27 m + {x8, hi(E)},  t5 = lo(E), t6 = lo(F),
// t7 = hi(F) + carry(E).  E can overflow 64 bits (x7 is any word); v_mad_u64_u32 reports that in its scalar carry-out, which the compiler
// does not expose - hence the inline assembly; F cannot (x8 < 2^30 for every operand the prover multiplies, m 2^27 < 2^59).  Instead of one
// multiply-add and a chain of three carries: a carry instruction (a lane mask through the scalar operand path) costs 0.83 of a multiply-add
// on gfx950, so - 2 carries + 1 multiply-add pays: 197 -> 211 G products/s, 174 -> 182 G butterflies/s in registers
// (tools/experiments/ubench_mul3.hip, profiles/r06_mul3_v2_ubench.txt).  The s_nop covers the two wait states gfx950 wants between a VALU
// write and a VALU read of an SGPR: the hazard recogniser does not look inside inline assembly.  SP_FE_NO_MAD_REDUCTION: the carry-chain
// form (what the host compiles).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SP_FE_NO_MAD_REDUCTION)
#define SP_FE_MAD_REDUCTION 1
__device__ __forceinline__ void sp_row_reduce(uint32_t m, uint32_t x6, uint32_t x7, uint32_t x8, uint32_t& t5, uint32_t& t6, uint32_t& t7) {
    const uint64_t X = ((uint64_t)x7 << 32) | x6;
    uint64_t E, cE, F, cF, unused;
    uint32_t r7;
    asm("v_mad_u64_u32 %0, %1, %2, 17, %3" : "=v"(E), "=s"(cE) : "v"(m), "v"(X));
    const uint64_t Y = ((uint64_t)x8 << 32) | (uint32_t)(E >> 32);
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(F), "=s"(cF) : "v"(m), "s"(0x08000000u), "v"(Y));
    asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(r7), "=s"(unused) : "v"((uint32_t)(F >> 32)), "s"(cE));
    t5 = (uint32_t)E; t6 = (uint32_t)F; t7 = r7;
}
#endif

// Montgomery product, CIOS with the reduction round merged into the accumulate carry chain.
// Row i:  u = t + a_i * b  (eight v_mad_u64_u32, D_j = a_i b_j + t_j),  m = -u_0 (p = 1 mod 2^32, so -p^-1 = -1),
//         u += m * p  with  m * p = m + 17 m 2^192 + m 2^251  (one mad and two shifts),  t = u >> 32.
// t stays < 2p < 2^253, so eight limbs suffice between rows; one conditional subtraction at the end.
// fe_mul_lazy omits that subtraction: for ANY a < 2^256 and b < p the result is a*b/R mod p in [0, 2p)
// (t < a*b/R + p < p + p).
SP_HD fe fe_mul_lazy(const fe& a, const fe& b) {
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
        unsigned c = (u0 != 0), c1, c2;  // u0 + m = c * 2^32
#pragma unroll
        for (int j = 1; j <= 5; ++j) { t[j - 1] = SP_ADDC(D[j], D[j - 1] >> 32, c, c1); c = c1; }
#ifdef SP_FE_MAD_REDUCTION
        const uint32_t x6 = SP_ADDC(D[6], D[5] >> 32, c, c1);
        const uint32_t x7 = SP_ADDC(D[7], D[6] >> 32, c1, c1);
        const uint32_t x8 = SP_ADDC(D[7] >> 32, 0u, c1, c1);
        (void)c2;
        sp_row_reduce(m, x6, x7, x8, t[5], t[6], t[7]);
#else
        const uint64_t m17 = (uint64_t)m * 17u;
        const uint32_t x6 = SP_ADDC(D[6], D[5] >> 32, c, c1);
        t[5] = SP_ADDC(x6, m17, 0u, c2);
        const uint32_t x7 = SP_ADDC(D[7], D[6] >> 32, c1, c1);
        const uint32_t k7 = (uint32_t)(m17 >> 32) + (m << 27);  // <= 16 + multiple of 2^27: no overflow
        t[6] = SP_ADDC(x7, k7, c2, c2);
        const uint32_t x8 = SP_ADDC(D[7] >> 32, m >> 5, c1, c1);
        t[7] = SP_ADDC(x8, 0u, c2, c2);
#endif
    }
    fe r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r.v[j] = t[j];
    return r;
}
SP_HD fe fe_mul(const fe& a, const fe& b) { return fe_reduce_once(fe_mul_lazy(a, b)); }

// ---- lazily reduced arithmetic for the NTT butterflies: products come back in [0, 2p) from fe_mul_lazy, sums and
// differences are left unreduced (see "deferred reduction" below); the helpers with 2p keep a value below 4p < 2^254.
#define SP_2P0 0x00000002u
#define SP_2P6 0x00000022u
#define SP_2P7 0x10000000u
// a >= 2p ? a - 2p : a        (a < 4p  ->  result < 2p)
SP_HD fe fe_reduce_2p(const fe& a) {
    fe d;
    unsigned br = 0, bo;
    d.v[0] = SP_SUBC(a.v[0], SP_2P0, br, bo); br = bo;
#pragma unroll
    for (int i = 1; i < 6; ++i) { d.v[i] = SP_SUBC(a.v[i], 0u, br, bo); br = bo; }
    d.v[6] = SP_SUBC(a.v[6], SP_2P6, br, bo); br = bo;
    d.v[7] = SP_SUBC(a.v[7], SP_2P7, br, bo); br = bo;
    fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = br ? a.v[i] : d.v[i];
    return r;
}
// a + b as 256-bit integers (caller guarantees a + b < 2^256)
SP_HD fe fe_add_raw(const fe& a, const fe& b) {
    fe s;
    unsigned c = 0, co;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s.v[i] = SP_ADDC(a.v[i], b.v[i], c, co); c = co; }
    return s;
}
// a - b + 2p  (a, b < 2p  ->  result in (0, 4p); exact modulo 2^256 whatever the intermediate borrow)
SP_HD fe fe_sub_add_2p(const fe& a, const fe& b) {
    fe d;
    unsigned br = 0, bo;
#pragma unroll
    for (int i = 0; i < 8; ++i) { d.v[i] = SP_SUBC(a.v[i], b.v[i], br, bo); br = bo; }
    fe r;
    unsigned c = 0, co;
    r.v[0] = SP_ADDC(d.v[0], SP_2P0, c, co); c = co;
#pragma unroll
    for (int i = 1; i < 6; ++i) { r.v[i] = SP_ADDC(d.v[i], 0u, c, co); c = co; }
    r.v[6] = SP_ADDC(d.v[6], SP_2P6, c, co); c = co;
    r.v[7] = SP_ADDC(d.v[7], SP_2P7, c, co);
    return r;
}
// [0, 4p) -> canonical [0, p)
SP_HD fe fe_canonical_4p(const fe& a) { return fe_reduce_once(fe_reduce_2p(a)); }

// ---- deferred reduction for the DIT passes: p > 2^251, so ANY 256-bit value is < 32p and a DIT butterfly
// (u, t) -> (u + t, u - t + 2p) with t = v w in [0, 2p) needs no correction at all while the bound (2 + 2 stages) p stays
// below 32p, i.e. for up to 14 stages.  One quotient estimate from the top five bits brings a value back:
// x = q 2^251 + r0 (q < 32, r0 < 2^251) and p = 2^251 + d, d = 17 2^192 + 1 < 2^197, so q d < 2^202.
// x - k p with k = q - (q != 0):  in [2^251 - 30 d, 2^252) for q >= 1, x itself (< 2^251) for q = 0  ->  [0, 2p).
SP_HD fe fe_reduce_lazy_2p(const fe& a) {
    const uint32_t q = a.v[7] >> 27;
    const uint32_t k = q - (q != 0u);
    fe r;
    unsigned br = 0, bo;
    r.v[0] = SP_SUBC(a.v[0], k, br, bo); br = bo;
#pragma unroll
    for (int i = 1; i < 6; ++i) { r.v[i] = SP_SUBC(a.v[i], 0u, br, bo); br = bo; }
    r.v[6] = SP_SUBC(a.v[6], 17u * k, br, bo); br = bo;
    r.v[7] = SP_SUBC(a.v[7], k << 27, br, bo);
    return r;
}
// any 256-bit value -> canonical [0, p):  y = x - q p lies in (-q d, 2^251); a borrow means y + p in (p - 31 d, p).
SP_HD fe fe_canonical_lazy(const fe& a) {
    const uint32_t q = a.v[7] >> 27;
    fe d;
    unsigned br = 0, bo;
    d.v[0] = SP_SUBC(a.v[0], q, br, bo); br = bo;
#pragma unroll
    for (int i = 1; i < 6; ++i) { d.v[i] = SP_SUBC(a.v[i], 0u, br, bo); br = bo; }
    d.v[6] = SP_SUBC(a.v[6], 17u * q, br, bo); br = bo;
    d.v[7] = SP_SUBC(a.v[7], q << 27, br, bo); br = bo;
    const uint32_t m = 0u - (uint32_t)br;
    fe r;
    unsigned c = 0, co;
    r.v[0] = SP_ADDC(d.v[0], m & SP_P0, c, co); c = co;
#pragma unroll
    for (int i = 1; i < 6; ++i) { r.v[i] = SP_ADDC(d.v[i], 0u, c, co); c = co; }
    r.v[6] = SP_ADDC(d.v[6], m & SP_P6, c, co); c = co;
    r.v[7] = SP_ADDC(d.v[7], m & SP_P7, c, co);
    return r;
}

// a - b + k p  for k in {2, 4, 8}  (b < k p, a < k p  ->  result in (0, 2 k p); exact modulo 2^256)
SP_HD fe fe_sub_add_kp(const fe& a, const fe& b, uint32_t k) {
    fe d;
    unsigned br = 0, bo;
#pragma unroll
    for (int i = 0; i < 8; ++i) { d.v[i] = SP_SUBC(a.v[i], b.v[i], br, bo); br = bo; }
    fe r;
    unsigned c = 0, co;
    r.v[0] = SP_ADDC(d.v[0], k, c, co); c = co;
#pragma unroll
    for (int i = 1; i < 6; ++i) { r.v[i] = SP_ADDC(d.v[i], 0u, c, co); c = co; }
    r.v[6] = SP_ADDC(d.v[6], 17u * k, c, co); c = co;
    r.v[7] = SP_ADDC(d.v[7], k << 27, c, co);
    return r;
}
// -1 in Montgomery form: p - (R mod p)
SP_HD fe fe_neg_one() {
    fe r = fe_zero();
    r.v[0] = 0x20u; r.v[6] = 0x220u;
    return r;
}

// Montgomery square: the CIOS rows of fe_mul_lazy with the triangle of the product only.  Row i adds
// a_i * (a_i 2^(32 i) + 2 * sum_{j > i} a_j 2^(32 j)) - the diagonal term once, every off-diagonal product once and doubled - so
// the eight rows hold 8 + 7 + ... + 1 = 36 multiply-adds instead of 64 (plus the eight of the reduction); the doubled operand is
// one funnel shift per limb.  a < 2^253 (anything the prover squares is below 2p); result a^2 / R mod p in [0, 2p).
SP_HD fe fe_sqr_lazy(const fe& a) {
    uint32_t d[8];   // 2a as limbs: the tail of row i starts with (a_(i+1) << 1), whose lost top bit is the low bit of d[i + 2]
    d[0] = a.v[0] << 1;
#pragma unroll
    for (int j = 1; j < 8; ++j) d[j] = (a.v[j] << 1) | (a.v[j - 1] >> 31);
    uint32_t t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t D[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < i) D[j] = t[j];
            else if (j == i) D[j] = (uint64_t)a.v[i] * a.v[i] + t[j];
            else if (j == i + 1) D[j] = (uint64_t)a.v[i] * (a.v[j] << 1) + t[j];
            else D[j] = (uint64_t)a.v[i] * d[j] + t[j];
        }
        const uint32_t u0 = (uint32_t)D[0];
        const uint32_t m = 0u - u0;
        unsigned c = (u0 != 0), c1, c2;  // u0 + m = c * 2^32
#pragma unroll
        for (int j = 1; j <= 5; ++j) { t[j - 1] = SP_ADDC(D[j], D[j - 1] >> 32, c, c1); c = c1; }
#ifdef SP_FE_MAD_REDUCTION
        const uint32_t x6 = SP_ADDC(D[6], D[5] >> 32, c, c1);
        const uint32_t x7 = SP_ADDC(D[7], D[6] >> 32, c1, c1);
        const uint32_t x8 = SP_ADDC(D[7] >> 32, 0u, c1, c1);
        (void)c2;
        sp_row_reduce(m, x6, x7, x8, t[5], t[6], t[7]);
#else
        const uint64_t m17 = (uint64_t)m * 17u;
        const uint32_t x6 = SP_ADDC(D[6], D[5] >> 32, c, c1);
        t[5] = SP_ADDC(x6, m17, 0u, c2);
        const uint32_t x7 = SP_ADDC(D[7], D[6] >> 32, c1, c1);
        const uint32_t k7 = (uint32_t)(m17 >> 32) + (m << 27);
        t[6] = SP_ADDC(x7, k7, c2, c2);
        const uint32_t x8 = SP_ADDC(D[7] >> 32, m >> 5, c1, c1);
        t[7] = SP_ADDC(x8, 0u, c2, c2);
#endif
    }
    fe r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r.v[j] = t[j];
    return r;
}
SP_HD fe fe_sqr(const fe& a) { return fe_reduce_once(fe_sqr_lazy(a)); }

// canonical integer (little-endian limbs, < p) -> Montgomery
SP_HD fe fe_to_mont(const fe& raw) { return fe_mul(raw, fe_r2()); }
// Montgomery -> canonical integer: a R^-1 mod p by four 64-bit Montgomery rounds.  p = 1 (mod 2^64), so the quotient
// digit is m = -(t mod 2^64) and  t <- (t + m p) / 2^64 = (t >> 64) + [t mod 2^64 != 0] + 17 m 2^128 + m 2^187
// (112 instead of 134 instructions for the product with the constant 1; every leaf element of a commitment takes this path).
SP_HD fe fe_from_mont(const fe& a) {
    uint32_t t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = a.v[j];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        unsigned br, bo, c, co, d, d2;
        const uint32_t m0 = SP_SUBC(0u, t[0], 0u, br);
        const uint32_t m1 = SP_SUBC(0u, t[1], br, bo);
        const unsigned c0 = (t[0] | t[1]) != 0;
        const uint64_t q0 = (uint64_t)m0 * 17u;
        const uint64_t q1 = (uint64_t)m1 * 17u + (q0 >> 32);
        const uint32_t s7 = m0 << 27, s8 = (m0 >> 5) | (m1 << 27), s9 = m1 >> 5;
        uint32_t u[8];
        u[0] = SP_ADDC(t[2], 0u, c0, c);
        u[1] = SP_ADDC(t[3], 0u, c, co); c = co;
        u[2] = SP_ADDC(t[4], 0u, c, co); c = co;
        u[3] = SP_ADDC(t[5], 0u, c, co); c = co;
        u[4] = SP_ADDC(t[6], (uint32_t)q0, c, co); c = co;
        u[5] = SP_ADDC(t[7], (uint32_t)q1, c, co); c = co;
        u[6] = SP_ADDC((uint32_t)(q1 >> 32), 0u, c, co);   // <= 17: no carry out
        u[5] = SP_ADDC(u[5], s7, 0u, d);
        u[6] = SP_ADDC(u[6], s8, d, d2);
        u[7] = s9 + d2;                                    // the value stays below 2p < 2^253
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = u[j];
    }
    fe r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r.v[j] = t[j];
    return fe_reduce_once(r);
}

SP_HD fe fe_from_u64(uint64_t x) {
    fe r = fe_zero();
    r.v[0] = (uint32_t)x; r.v[1] = (uint32_t)(x >> 32);
    return fe_to_mont(r);
}

SP_HD fe fe_pow_u64(const fe& a, uint64_t e) {
    fe r = fe_one(), b = a;
    while (e) {
        if (e & 1) r = fe_mul(r, b);
        b = fe_sqr(b);
        e >>= 1;
    }
    return r;
}

// a^(p-2) - the inversion used until round 3, kept as the cross-check of fe_inv below (tests/test_fp_host.py); zero maps to zero.
// p - 2 = 2^251 + 2^196 + (2^192 - 1): a^(2^192 - 1) by the doubling chain 1, 2, 3, 6, 12, 24, 48, 96, 192 (191 squarings,
// 8 products), then a^(2^192) = that * a, squared 4 times -> a^(2^196), 55 more times -> a^(2^251): 250 squarings and 11
// products instead of the 251 + 193 of square-and-multiply.
// (the whole chain runs on lazily reduced values: squares and products of operands below 2p stay below 2p, so the 261 conditional
// subtractions of the canonical forms are not on the dependent chain - one at the end)
SP_HD fe fe_sqr_n(fe x, int n) {
    for (int i = 0; i < n; ++i) x = fe_sqr_lazy(x);
    return x;
}
SP_HD fe fe_inv_fermat(const fe& a) {
    const fe x2 = fe_mul_lazy(fe_sqr_lazy(a), a);           // a^(2^2 - 1)
    const fe x3 = fe_mul_lazy(fe_sqr_lazy(x2), a);          // a^(2^3 - 1)
    const fe x6 = fe_mul_lazy(fe_sqr_n(x3, 3), x3);
    const fe x12 = fe_mul_lazy(fe_sqr_n(x6, 6), x6);
    const fe x24 = fe_mul_lazy(fe_sqr_n(x12, 12), x12);
    const fe x48 = fe_mul_lazy(fe_sqr_n(x24, 24), x24);
    const fe x96 = fe_mul_lazy(fe_sqr_n(x48, 48), x48);
    const fe x192 = fe_mul_lazy(fe_sqr_n(x96, 96), x96);    // a^(2^192 - 1)
    const fe b = fe_sqr_n(fe_mul_lazy(x192, a), 4);         // a^(2^196)
    const fe c = fe_sqr_n(b, 55);                           // a^(2^251)
    return fe_reduce_once(fe_mul_lazy(fe_mul_lazy(c, b), x192));
}

// Modular inverse by Bernstein-Yang division steps ("safegcd", the constant-time form: no branch depends on the data, every lane of
// a wave runs the same 600 steps).  The Fermat chain above is 261 DEPENDENT Montgomery products - about 56 000 instructions on the one
// lane that inverts a batch's running product, 0.28 ms of pure latency four times per proof; here the state (f, g) = (p, x) is
// reduced 30 division steps at a time on its low 32 bits only (12 one-cycle operations per step) and the resulting 2 x 2 transition
// matrix is applied to the nine signed 30-bit limbs of (f, g) and of the Bezout pair (d, e) - 20 rounds, about 12 000 instructions.
// 590 steps suffice for a 256-bit modulus in the half-delta variant (Bernstein-Yang 2019; the limb and bound bookkeeping follows the
// 32-bit instance of that method as it is commonly written: d, e stay in (-2p, p), limbs in (-2^30, 2^30) except the signed top one).
// p = 1 (mod 2^30), so the inverse of the modulus modulo 2^30 that makes t (d, e) divisible by 2^30 is 1, and p has three non-zero
// limbs (1, 17 2^12 in limb 6, 2^11 in limb 8).
// Input and output in Montgomery form: the steps invert the integer a R, giving a^-1 R^-1; one product with R^3 makes that a^-1 R.
struct fe_s30 { int32_t v[9]; };
SP_HD fe fe_inv(const fe& a) {
    constexpr int32_t M30 = 0x3fffffff;
    constexpr int32_t P6 = 17 << 12, P8 = 1 << 11;      // limbs 6 and 8 of p (limb 0 is 1, the others 0)
    fe_s30 d, e, f, g;
#pragma unroll
    for (int i = 0; i < 9; ++i) { d.v[i] = 0; e.v[i] = 0; f.v[i] = 0; }
    e.v[0] = 1;
    f.v[0] = 1; f.v[6] = P6; f.v[8] = P8;
    {   // g = the integer a.v (eight 32-bit limbs) as nine 30-bit limbs
        const fe x = fe_reduce_once(a);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
            uint64_t two = x.v[w];
            if (w + 1 < 8) two |= (uint64_t)x.v[w + 1] << 32;
            g.v[i] = (int32_t)((uint32_t)(two >> sh) & (uint32_t)M30);
        }
    }
    int32_t zeta = -1;   // -(delta + 1/2), delta = 1/2 at the start
#pragma unroll 1
    for (int round = 0; round < 20; ++round) {
        // 30 division steps on the low limbs: the transition matrix [[u, v], [q, r]] (entries in [-2^30, 2^30])
        uint32_t u = 1, v = 0, q = 0, r = 1, fl = (uint32_t)f.v[0], gl = (uint32_t)g.v[0];
#pragma unroll 2
        for (int i = 0; i < 30; ++i) {
            uint32_t mask1 = (uint32_t)(zeta >> 31);                 // zeta < 0
            const uint32_t mask2 = 0u - (gl & 1u);                   // g odd
            const uint32_t x = (fl ^ mask1) - mask1, y = (u ^ mask1) - mask1, z = (v ^ mask1) - mask1;   // +-(f, u, v)
            gl += x & mask2; q += y & mask2; r += z & mask2;
            mask1 &= mask2;
            zeta = (int32_t)((uint32_t)zeta ^ mask1) - 1;            // -zeta - 2 when both hold, zeta - 1 otherwise
            fl += gl & mask1; u += q & mask1; v += r & mask1;
            gl >>= 1; u <<= 1; v <<= 1;
        }
        const int32_t tu = (int32_t)u, tv = (int32_t)v, tq = (int32_t)q, tr = (int32_t)r;
        {   // (d, e) <- t (d, e) / 2^30 modulo p: multiples md, me of p make the low 30 bits vanish
            const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
            int32_t md = (tu & sd) + (tv & se), me = (tq & sd) + (tr & se);
            int64_t cd = (int64_t)tu * d.v[0] + (int64_t)tv * e.v[0];
            int64_t ce = (int64_t)tq * d.v[0] + (int64_t)tr * e.v[0];
            md -= (int32_t)(((uint32_t)cd + (uint32_t)md) & (uint32_t)M30);      // (p^-1 mod 2^30 = 1)
            me -= (int32_t)(((uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
            cd += md; ce += me;                                                  // limb 0 of p is 1
            cd >>= 30; ce >>= 30;
#pragma unroll
            for (int i = 1; i < 9; ++i) {
                cd += (int64_t)tu * d.v[i] + (int64_t)tv * e.v[i];
                ce += (int64_t)tq * d.v[i] + (int64_t)tr * e.v[i];
                if (i == 6) { cd += (int64_t)P6 * md; ce += (int64_t)P6 * me; }
                if (i == 8) { cd += (int64_t)P8 * md; ce += (int64_t)P8 * me; }
                d.v[i - 1] = (int32_t)cd & M30; cd >>= 30;
                e.v[i - 1] = (int32_t)ce & M30; ce >>= 30;
            }
            d.v[8] = (int32_t)cd; e.v[8] = (int32_t)ce;
        }
        {   // (f, g) <- t (f, g) / 2^30 (exact)
            int64_t cf = (int64_t)tu * f.v[0] + (int64_t)tv * g.v[0];
            int64_t cg = (int64_t)tq * f.v[0] + (int64_t)tr * g.v[0];
            cf >>= 30; cg >>= 30;
#pragma unroll
            for (int i = 1; i < 9; ++i) {
                cf += (int64_t)tu * f.v[i] + (int64_t)tv * g.v[i];
                cg += (int64_t)tq * f.v[i] + (int64_t)tr * g.v[i];
                f.v[i - 1] = (int32_t)cf & M30; cf >>= 30;
                g.v[i - 1] = (int32_t)cg & M30; cg >>= 30;
            }
            f.v[8] = (int32_t)cf; g.v[8] = (int32_t)cg;
        }
    }
    // g = 0 and f = +-1 now: d = +-x^-1 in (-2p, p).  Add p when negative, negate when f is negative, propagate, add p once more.
    {
        int32_t cond = d.v[8] >> 31;
        d.v[0] += 1 & cond; d.v[6] += P6 & cond; d.v[8] += P8 & cond;
        const int32_t neg = f.v[8] >> 31;
#pragma unroll
        for (int i = 0; i < 9; ++i) d.v[i] = (d.v[i] ^ neg) - neg;
#pragma unroll
        for (int i = 0; i < 8; ++i) { d.v[i + 1] += d.v[i] >> 30; d.v[i] &= M30; }
        cond = d.v[8] >> 31;
        d.v[0] += 1 & cond; d.v[6] += P6 & cond; d.v[8] += P8 & cond;
#pragma unroll
        for (int i = 0; i < 8; ++i) { d.v[i + 1] += d.v[i] >> 30; d.v[i] &= M30; }
    }
    fe y;   // nine 30-bit limbs -> eight 32-bit limbs
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int bit = 32 * k, i = bit / 30, sh = bit - 30 * i;      // limb i holds bits [30 i, 30 i + 30)
        uint64_t acc = (uint64_t)(uint32_t)d.v[i] >> sh;
        acc |= (uint64_t)(uint32_t)d.v[i + 1] << (30 - sh);
        if (i + 2 < 9) acc |= (uint64_t)(uint32_t)d.v[i + 2] << (60 - sh);
        y.v[k] = (uint32_t)acc;
    }
    fe r3;   // R^3 mod p
    r3.v[0] = 0x406df18eu; r3.v[1] = 0xcc7177d1u; r3.v[2] = 0x77ffcc06u; r3.v[3] = 0x75457066u;
    r3.v[4] = 0x36300018u; r3.v[5] = 0xf47d84f8u; r3.v[6] = 0x873c0a6du; r3.v[7] = 0x038e5f79u;
    return fe_mul(y, r3);
}

// ---- byte codecs ---------------------------------------------------------------------------------------
SP_HD uint32_t sp_bswap32(uint32_t x) { return (x >> 24) | ((x >> 8) & 0xff00u) | ((x << 8) & 0xff0000u) | (x << 24); }

// canonical 32-byte big-endian (reference wire format, `to_bytes_be`) -> Montgomery fe. Input must be < p.
SP_HD fe fe_from_bytes_be(const uint8_t* b) {
    fe raw;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint8_t* q = b + 4 * (7 - i);
        raw.v[i] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
    }
    return fe_to_mont(raw);
}
SP_HD void fe_to_bytes_be(const fe& a, uint8_t* b) {
    fe raw = fe_from_mont(a);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint8_t* q = b + 4 * (7 - i);
        q[0] = (uint8_t)(raw.v[i] >> 24); q[1] = (uint8_t)(raw.v[i] >> 16); q[2] = (uint8_t)(raw.v[i] >> 8); q[3] = (uint8_t)raw.v[i];
    }
}
// lambdaworks in-memory layout: 4 x u64, limb 0 MOST significant, Montgomery form (R = 2^256) -> zero-cost reorder
SP_HD fe fe_from_lw_limbs(const uint64_t* l) {
    fe r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.v[2 * i] = (uint32_t)l[3 - i]; r.v[2 * i + 1] = (uint32_t)(l[3 - i] >> 32); }
    return r;
}
SP_HD void fe_to_lw_limbs(const fe& a, uint64_t* l) {
#pragma unroll
    for (int i = 0; i < 4; ++i) l[3 - i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
}
