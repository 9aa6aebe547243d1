// Host-side probe of csrc/fp.h for tests/test_fp_host.py: reads "op a b" lines (hex, 64 digits each), prints the result.
#include "fp.h"
#include <cstdio>
#include <cstring>
#include <cstdlib>
static fe parse(const char* h) {
    fe a;
    for (int i = 0; i < 8; ++i) { char t[9]; memcpy(t, h + 8 * (7 - i), 8); t[8] = 0; a.v[i] = (uint32_t)strtoul(t, 0, 16); }
    return a;
}
static void show(const fe& r) { for (int i = 7; i >= 0; --i) printf("%08x", r.v[i]); printf("\n"); }
int main() {
    char op[32], A[80], B[80];
    unsigned k;
    while (scanf("%31s %79s %79s %u", op, A, B, &k) == 4) {
        fe a = parse(A), b = parse(B);
        if (!strcmp(op, "lazy2p")) show(fe_reduce_lazy_2p(a));
        else if (!strcmp(op, "canon")) show(fe_canonical_lazy(a));
        else if (!strcmp(op, "subkp")) show(fe_sub_add_kp(a, b, k));
        else if (!strcmp(op, "mullazy")) show(fe_mul_lazy(a, b));
        else if (!strcmp(op, "mul")) show(fe_mul(a, b));
        else if (!strcmp(op, "sqrlazy")) show(fe_sqr_lazy(a));
        else if (!strcmp(op, "sqr")) show(fe_sqr(a));
        else if (!strcmp(op, "negone")) show(fe_neg_one());
        else if (!strcmp(op, "sub2p")) show(fe_sub_add_2p(a, b));
        else if (!strcmp(op, "inv")) show(fe_inv(a));
        else if (!strcmp(op, "invfermat")) show(fe_inv_fermat(a));
        else if (!strcmp(op, "frommont")) show(fe_from_mont(a));
        else return 2;
    }
    return 0;
}
