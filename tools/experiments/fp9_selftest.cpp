// Host self-test of fp9.h (the 9 x 28-bit lazy arithmetic of the NTT butterflies) against fp.h, including the
// extreme limb patterns the bounds in fp9.h allow.  Built by `make fp9_selftest`, run by tests/test_fp9_host.py.
#include "fp9.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

static uint64_t g_s = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return (uint32_t)(g_s >> 16); }

// value(a) mod p as a Montgomery fe (fp.h form)
static fe value_mod_p(const fe9& a) {
    fe acc = fe_zero();
    fe sh = fe_one();                       // 2^(28 i)
    const fe two28 = fe_from_u64(1ull << 28);
    for (int i = 0; i < 9; ++i) {
        acc = fe_add(acc, fe_mul(fe_from_u64(a.l[i]), sh));
        sh = fe_mul(sh, two28);
    }
    return acc;
}
static fe pow2(int e) { fe r = fe_one(), two = fe_from_u64(2); for (int i = 0; i < e; ++i) r = fe_mul(r, two); return r; }

static fe9 random_tight_below_p() {
    fe x;
    for (int j = 0; j < 8; ++j) x.v[j] = rnd();
    x.v[7] &= 0x07ffffffu;
    x = fe_reduce_once(x);                  // < 2^251 + ... : one subtraction is enough
    return fe9_unpack(x);
}
static fe9 random_loose(uint32_t limb_max, uint32_t top_max) {
    fe9 a;
    for (int i = 0; i < 8; ++i) a.l[i] = (uint32_t)(((uint64_t)rnd() * limb_max) >> 32);
    a.l[8] = (uint32_t)(((uint64_t)rnd() * top_max) >> 32);
    return a;
}
static bool tight(const fe9& a) { for (int i = 0; i < 8; ++i) if (a.l[i] > SP_M28) return false; return true; }

static int fails = 0;
#define CHECK(cond, msg) do { if (!(cond)) { if (fails < 20) std::printf("FAIL: %s (line %d)\n", msg, __LINE__); ++fails; } } while (0)

int main() {
    const fe inv252 = fe_inv(pow2(252));
    fe pm1 = fe_zero(); pm1.v[0] = SP_P0 - 1; pm1.v[6] = SP_P6; pm1.v[7] = SP_P7;   // p - 1 (raw limbs)
    // ---- pack / unpack
    for (int it = 0; it < 2000; ++it) {
        fe x; for (int j = 0; j < 8; ++j) x.v[j] = rnd();
        if (it == 0) for (int j = 0; j < 8; ++j) x.v[j] = 0xffffffffu;
        if (it == 1) for (int j = 0; j < 8; ++j) x.v[j] = 0;
        fe9 u = fe9_unpack(x);
        CHECK(tight(u), "unpack tight");
        CHECK(fe_eq(fe9_pack(u), x), "pack(unpack(x)) == x");
    }
    // ---- product: random and extreme operands
    for (int it = 0; it < 20000; ++it) {
        fe9 w = random_tight_below_p();
        fe9 a;
        if (it % 4 == 0) a = random_loose(0xffffffffu, 0xffffffffu);
        else if (it % 4 == 1) { for (int i = 0; i < 9; ++i) a.l[i] = 0xffffffffu; }           // every limb maximal
        else if (it % 4 == 2) a = random_tight_below_p();
        else a = random_loose(0xe0000000u, 0xffffffffu);
        if (it % 8 == 1) w = fe9_unpack(pm1);                                                   // largest twiddle value
        if (it == 5) { for (int i = 0; i < 9; ++i) a.l[i] = 0; }
        fe9 r = fe9_mul(a, w);
        CHECK(tight(r), "mul result tight");
        fe expect = fe_mul(fe_mul(value_mod_p(a), value_mod_p(w)), inv252);
        CHECK(fe_eq(value_mod_p(r), expect), "mul value");
    }
    // ---- biased subtraction, fold, canonical
    for (int it = 0; it < 20000; ++it) {
        fe9 a = random_loose(0x74000000u, 0xffffffffu);       // limbs up to 7.25 * 2^28
        if (it % 2) a.l[8] = (uint32_t)(((uint64_t)rnd() * 0xb0000000u) >> 32);  // room for + K * 2^27 in the top limb
        else a.l[8] >>= 2;
        fe9 t = random_tight_below_p();
        if (it % 3 == 0) { for (int i = 0; i < 8; ++i) t.l[i] = SP_M28; t.l[8] = 3 * (1u << 27) - 1; }  // largest t allowed for K = 3
        fe9 y = fe9_sub_biased<3>(a, t);
        CHECK(fe_eq(value_mod_p(y), fe_sub(value_mod_p(a), value_mod_p(t))), "sub_biased<3> value");
        for (int i = 0; i < 9; ++i) CHECK(y.l[i] >= a.l[i], "sub_biased limb did not borrow");
        fe9 y9 = fe9_sub_biased<9>(a, t);
        CHECK(fe_eq(value_mod_p(y9), fe_sub(value_mod_p(a), value_mod_p(t))), "sub_biased<9> value");
        fe9 f = fe9_fold(a);
        CHECK(tight(f) && f.l[8] < (1u << 28) + 32u, "fold output bounds");
        CHECK(fe_eq(value_mod_p(f), value_mod_p(a)), "fold value");
        fe9 ac = a; if (ac.l[8] > 0xf0000000u) ac.l[8] = 0xf0000000u;   // canonical needs value < 2^256
        fe c = fe9_canonical(ac);
        fe9 cu = fe9_unpack(c);
        CHECK(fe_eq(value_mod_p(cu), value_mod_p(ac)), "canonical value");
        // canonical: c < p  <=>  reduce_once(c) == c
        CHECK(fe_eq(fe_reduce_once(c), c), "canonical < p");
    }
    // ---- canonical of extreme inputs
    {
        fe9 a; for (int i = 0; i < 8; ++i) a.l[i] = 0xfffffff0u; a.l[8] = 0xffffff00u;
        // value may exceed 2^256 here -> shrink the top so that it does not
        a.l[8] = 0xefffffffu;
        fe c = fe9_canonical(a);
        CHECK(fe_eq(value_mod_p(fe9_unpack(c)), value_mod_p(a)), "canonical extreme value");
        CHECK(fe_eq(fe_reduce_once(c), c), "canonical extreme < p");
        fe9 z; for (int i = 0; i < 9; ++i) z.l[i] = 0;
        CHECK(fe_is_zero(fe9_canonical(z)), "canonical(0) == 0");
        fe9 pp = fe9_unpack(pm1); pp.l[0] += 1;  // p itself
        CHECK(fe_is_zero(fe9_canonical(pp)), "canonical(p) == 0");
    }
    // ---- a chain of lazy butterflies with the NTT's K schedule stays within the documented bounds
    {
        const int KS[4] = {3, 4, 6, 9};
        for (int it = 0; it < 2000; ++it) {
            fe9 u = fe9_fold(random_loose(0x74000000u, 0xffffffffu)), v = fe9_fold(random_loose(0x74000000u, 0xffffffffu));
            fe eu = value_mod_p(u), ev = value_mod_p(v);
            for (int s = 0; s < 4; ++s) {
                fe9 w = (it % 2) ? fe9_unpack(pm1) : random_tight_below_p();
                fe ew = fe_mul(value_mod_p(w), inv252);  // the field element the twiddle stands for
                fe9 t = fe9_mul(v, w);
                fe9 x = fe9_add(u, t), y;
                switch (KS[s]) { case 3: y = fe9_sub_biased<3>(u, t); break; case 4: y = fe9_sub_biased<4>(u, t); break;
                                 case 6: y = fe9_sub_biased<6>(u, t); break; default: y = fe9_sub_biased<9>(u, t); }
                fe et = fe_mul(ev, ew);
                fe ex = fe_add(eu, et), ey = fe_sub(eu, et);
                CHECK(fe_eq(value_mod_p(x), ex) && fe_eq(value_mod_p(y), ey), "butterfly chain value");
                CHECK(t.l[8] <= (uint32_t)KS[s] * (1u << 27) - 1u, "t below K * 2^251");
                for (int i = 0; i < 8; ++i) CHECK(x.l[i] < 0x74000001u && y.l[i] < 0x74000001u, "limb growth bound");
                // feed the larger-growth output back as both operands of the next stage
                u = y; v = y; eu = ey; ev = ey;
            }
        }
    }
    if (fails) { std::printf("fp9 selftest: %d failures\n", fails); return 1; }
    std::printf("fp9 selftest: ok\n");
    return 0;
}
