// Stark252 arithmetic on 9 x 28-bit limbs with lazy reduction, for the butterfly loops of the NTT passes.
//
// Why a second representation: on gfx950 a carry-chain instruction (v_addc_co_u32) costs ~1.6x and v_mad_u64_u32 ~2x a
// plain VOP2 op (profiles/r01_instruction_ubench3.txt), and the 8 x 32-bit CIOS product of fp.h spends more issue
// slots on carry chains than on multiplies.  With 28-bit limbs the 81 partial products accumulate into 64-bit
// column registers with NO carries (9 * 2^32 * 2^28 < 2^64), and the Montgomery reduction by
//     p = 2^251 + 17*2^192 + 1 = [1, 0, 0, 0, 0, 0, 2^24, 1, 2^27]  (radix 2^28)
// is two multiply-adds per row because -p^-1 = -1 (mod 2^28) and p's limbs are 1, 2^24 + 2^28 (limbs 6, 7) and 2^27.
// Additions and subtractions are limb-wise with no carries at all; values and limbs are allowed to grow for a few
// butterfly stages and are folded back with 2^251 = -(17*2^192 + 1) (mod p).
//
// Scaling: fe9_mul(a, w) = a * w * 2^-252 (R' = 2^252), while fp.h uses R = 2^256.  The NTT keeps its DATA in the
// fp.h Montgomery form and stores its TWIDDLES as w * 2^252 mod p, so data * twiddle stays in the fp.h form and no
// conversion of the data is ever needed.
//
// Vocabulary used in the bounds below: "tight" = limbs 0..7 < 2^28; "loose" = limbs < 2^32 - 2^29.
// value(a) = sum l[i] * 2^(28 i).  All bounds are proved in DESIGN.md section 4.1b and exercised by tests/.
#pragma once
#include "../../lambdaworks_cairo_prover_amd/csrc/fp.h"

struct fe9 { uint32_t l[9]; };
#define SP_M28 0x0fffffffu

// Any 256-bit value -> limbs 0..7 tight, l[8] = top 32 bits.
SP_HD fe9 fe9_unpack(const fe& a) {
    fe9 r;
    r.l[0] = a.v[0] & SP_M28;
    r.l[1] = ((a.v[0] >> 28) | (a.v[1] << 4)) & SP_M28;
    r.l[2] = ((a.v[1] >> 24) | (a.v[2] << 8)) & SP_M28;
    r.l[3] = ((a.v[2] >> 20) | (a.v[3] << 12)) & SP_M28;
    r.l[4] = ((a.v[3] >> 16) | (a.v[4] << 16)) & SP_M28;
    r.l[5] = ((a.v[4] >> 12) | (a.v[5] << 20)) & SP_M28;
    r.l[6] = ((a.v[5] >> 8) | (a.v[6] << 24)) & SP_M28;
    r.l[7] = a.v[6] >> 4;
    r.l[8] = a.v[7];
    return r;
}
// Tight limbs, value < 2^256 -> 8 x 32.
SP_HD fe fe9_pack(const fe9& a) {
    fe r;
    r.v[0] = a.l[0] | (a.l[1] << 28);
    r.v[1] = (a.l[1] >> 4) | (a.l[2] << 24);
    r.v[2] = (a.l[2] >> 8) | (a.l[3] << 20);
    r.v[3] = (a.l[3] >> 12) | (a.l[4] << 16);
    r.v[4] = (a.l[4] >> 16) | (a.l[5] << 12);
    r.v[5] = (a.l[5] >> 20) | (a.l[6] << 8);
    r.v[6] = (a.l[6] >> 24) | (a.l[7] << 4);
    r.v[7] = a.l[8];
    return r;
}

SP_HD void fe9_mad(uint64_t& d, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    // keeps multiplications by a power of two on the multiply-add pipe (the compiler would split them into a 64-bit
    // shift and a 64-bit add, which costs more issue slots than one v_mad_u64_u32)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b) : "vcc");
#else
    d += (uint64_t)a * b;
#endif
}

// Montgomery product with R' = 2^252.
//   in : a loose (any limbs < 2^32), w tight with value(w) < p.
//   out: tight limbs, l[8] < 2^32, value = a * w * 2^-252 (mod p), value < value(a) * value(w) / 2^252 + p.
// Column bound: every 64-bit column receives at most 9 products < 2^32 * 2^28, two reduction terms < 2^57 + 2^55 and one
// carry < 2^36, i.e. < 9 * 2^60 + 2^58 < 2^64.
SP_HD fe9 fe9_mul(const fe9& a, const fe9& w) {
    uint64_t D[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) D[k] = 0;
    const uint32_t c27 = 0x08000000u;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
#pragma unroll
        for (int j = 0; j < 9; ++j) D[i + j] += (uint64_t)a.l[i] * w.l[j];
        // m = -D[i] mod 2^28 makes column i vanish mod 2^28:  D[i] + m * p_0 = multiple of 2^28
        const uint32_t m = (0u - (uint32_t)D[i]) & SP_M28;
        D[i + 1] += (D[i] + m) >> 28;                    // exact: the low 28 bits of D[i] + m are zero
        D[i + 6] += (uint64_t)m * 0x11000000u;           // m * (p_6 + p_7 * 2^28) = m * (2^24 + 2^28)
        fe9_mad(D[i + 8], m, c27);                       // m * p_8 = m * 2^27
    }
    fe9 r;
#pragma unroll
    for (int k = 9; k < 17; ++k) { D[k + 1] += D[k] >> 28; r.l[k - 9] = (uint32_t)D[k] & SP_M28; }
    r.l[8] = (uint32_t)D[17];
    return r;
}

// Limb-wise sum (no carries): limbs add up.
SP_HD fe9 fe9_add(const fe9& a, const fe9& b) {
    fe9 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}

// a - t + K*p, limb-wise, never borrowing: K*p is spread so that every limb of it exceeds any tight limb of t.
//   requires: t tight and value(t) < K * 2^251  (so that t.l[8] <= K * 2^27 - 1).
//   limbs grow by < 2^28 + K * 2^24, the value by exactly K*p - t.
template <int K>
SP_HD fe9 fe9_sub_biased(const fe9& a, const fe9& t) {
    fe9 r;
    r.l[0] = a.l[0] + ((K + (1u << 28)) - t.l[0]);
#pragma unroll
    for (int i = 1; i < 6; ++i) r.l[i] = a.l[i] + (SP_M28 - t.l[i]);
    r.l[6] = a.l[6] + ((SP_M28 + K * (1u << 24)) - t.l[6]);
    r.l[7] = a.l[7] + ((SP_M28 + K) - t.l[7]);
    r.l[8] = a.l[8] + ((K * (1u << 27) - 1u) - t.l[8]);
    return r;
}

// Fold the bits above 2^251 and propagate carries ("weak reduction").
//   in : limbs < 2^32 - 2^29, value < 2^256.
//   out: tight limbs, l[8] < 2^28 + 32, value = in (mod p), value < 2^252 * (1 + 2^-23).
// hi * 2^251 = hi * (p - d), d = 17*2^192 + 1, so the top bits are replaced by p - hi * d (added limb-wise with a
// pre-charged copy of p so that no limb goes negative: +2^28 at limb 0, +(2^28 - 1) at limbs 1..7, -1 at limb 8).
SP_HD fe9 fe9_fold(const fe9& a) {
    const uint32_t hi = a.l[8] >> 27;
    const uint32_t h17 = hi * 17u;  // < 2^10
    uint32_t r[9];
    r[0] = a.l[0] + ((SP_M28 + 2u) - hi);
#pragma unroll
    for (int i = 1; i < 6; ++i) r[i] = a.l[i] + SP_M28;
    r[6] = a.l[6] + ((SP_M28 + (1u << 24)) - ((h17 & 15u) << 24));
    r[7] = a.l[7] + ((SP_M28 + 1u) - (h17 >> 4));
    r[8] = (a.l[8] & 0x07ffffffu) + 0x07ffffffu;
    fe9 o;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const uint32_t t = r[i] + c; o.l[i] = t & SP_M28; c = t >> 28; }
    o.l[8] = r[8] + c;
    return o;
}

// Loose fe9 -> canonical fe in [0, p).  in: limbs < 2^32 - 2^4, value < 2^256.
SP_HD fe fe9_canonical(const fe9& a) {
    fe9 t;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const uint32_t s = a.l[i] + c; t.l[i] = s & SP_M28; c = s >> 28; }
    t.l[8] = a.l[8] + c;                      // exact: value < 2^256
    // now value = hi * 2^251 + lo with lo < 2^251 exactly, so fold gives lo + p - hi * d < 2^251 + p < 2p
    return fe_reduce_once(fe9_pack(fe9_fold(t)));
}
