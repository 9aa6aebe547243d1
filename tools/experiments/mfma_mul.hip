// Experiment: Stark252 butterflies with the twiddle product on the matrix cores (V_MFMA_I32_32X32X32_I8).
//
// A product by a CONSTANT w is linear in the data: with the data x written in balanced radix-256 digits d_j in [-128, 127],
//   w x = sum_j d_j (w 2^{8j} mod p)   (mod p),
// i.e. a 32 x 32 matrix of signed bytes M[k][j] = digit k of (w 2^{8j} mod p) applied to the digit vector: one i8 MFMA
// multiplies 32 elements by the same w and leaves 32 column sums (|.| < 2^19) per element in the accumulators.  An element
// occupies two lanes (j, j + 32): lane half h feeds digits 16h .. 16h+15 and receives the column sums 16h .. 16h+15.
// Adding u is a second MFMA with the identity.  What is left for the VALU is the recombination of the column sums into
// 32 bytes again (carry chain of four limbs per lane, one fold of the bits above 2^252 with 2^252 = -(34 2^192 + 2) mod p).
//
// Digit form of a value x:  x' = x + CBAL (CBAL = 0x8080...80), stored as xb = x' ^ 0x8080...80: byte k of xb as a signed
// char is digit k.  The accumulators start from biases beta_k = 2^20 + delta_k (all sums positive) whose total
// sum_k beta_k 2^{8k} = CBAL - QMAX K (mod p), K = 34 2^192 + 2, QMAX = 2^17: after the fold the bytes are digits again.
#include "../../lambdaworks_cairo_prover_amd/csrc/fp.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr uint32_t QMAX = 1u << 17;

// ---------------------------------------------------------------- device ------------------------------------------------
// 16 column sums (positive, < 2^21) of this lane half -> its four limbs in digit form.  h = lane >> 5.
__device__ __forceinline__ v4i recombine(const v16i c, const uint32_t h) {
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int L = 0; L < 4; ++L) {
        lo[L] = (uint32_t)c[4 * L] + ((uint32_t)c[4 * L + 1] << 8);
        hi[L] = (uint32_t)c[4 * L + 2] + ((uint32_t)c[4 * L + 3] << 8);
    }
    // quotient estimate from the top of the upper half (bits >= 252 of the whole number): never above the true quotient, at most 5 below
    const uint32_t qe_mine = hi[3] >> 12;
    const auto sw = __builtin_amdgcn_permlane32_swap(qe_mine, qe_mine, false, false);
    const uint32_t qe = sw[1];              // the upper half's value, in both halves
    const uint32_t f = QMAX - qe;           // (QMAX - qe) K is added instead of subtracting qe K; the biases carry -QMAX K
    uint32_t carry = h ? 0u : 2u * f;
    const uint32_t f34 = h ? 34u * f : 0u;
    uint32_t limb[4];
#pragma unroll
    for (int L = 0; L < 4; ++L) {
        uint32_t t = lo[L] + carry;
        if (L == 2) t += f34;
        const uint64_t acc = (uint64_t)hi[L] * 65536u + t;
        limb[L] = (uint32_t)acc;
        carry = (uint32_t)(acc >> 32);
    }
    // the lower half's carry enters the upper half
    const auto sw2 = __builtin_amdgcn_permlane32_swap(carry, carry, false, false);
    const uint32_t cin = h ? sw2[0] : 0u;
    unsigned c0, c1;
    limb[0] = SP_ADDC(limb[0], cin, 0u, c0);
    limb[1] = SP_ADDC(limb[1], 0u, c0, c1);
    limb[2] = SP_ADDC(limb[2], 0u, c1, c0);
    limb[3] = SP_ADDC(limb[3], 0u, c0, c1);
    limb[3] -= h ? (qe << 28) : 0u;
    v4i r;
#pragma unroll
    for (int L = 0; L < 4; ++L) r[L] = (int)(limb[L] ^ 0x80808080u);
    return r;
}

// standard element (any value < 2^255 - 2^252) -> digit form, the half of lane half h
__device__ __forceinline__ v4i to_digits(const fe& x, const uint32_t h) {
    fe s;
    unsigned c = 0, co;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s.v[i] = SP_ADDC(x.v[i], 0x80808080u, c, co); c = co; }
    v4i r;
#pragma unroll
    for (int L = 0; L < 4; ++L) r[L] = (int)((h ? s.v[4 + L] : s.v[L]) ^ 0x80808080u);
    return r;
}

struct MatSet {           // per lane: 16 bytes of each A operand
    v4i W, Wn, I;
};

// out1 = u + w v, out2 = u - w v  for 32 columns (u, v: digit form halves of this lane)
__device__ __forceinline__ void butterfly(const v4i W, const v4i Wn, const v4i I, const v16i bias, v4i& u, v4i& v, const uint32_t h) {
    v16i a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(I, u, bias, 0, 0, 0);
    v16i a2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(I, u, bias, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(W, v, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(Wn, v, a2, 0, 0, 0);
    u = recombine(a1, h);
    v = recombine(a2, h);
}

// correctness: one butterfly per wave on 32 columns
__global__ void check_kernel(const fe* u_in, const fe* v_in, const v4i* mats /* [tw][3][64] */, const int* bias_tab /* [2][16] */,
                             uint32_t* out1, uint32_t* out2, int stages) {
    const uint32_t lane = threadIdx.x & 63u, j = lane & 31u, h = lane >> 5;
    const uint32_t tw = blockIdx.x;
    const v4i W = mats[(tw * 3 + 0) * 64 + lane], Wn = mats[(tw * 3 + 1) * 64 + lane], I = mats[(tw * 3 + 2) * 64 + lane];
    v16i bias;
#pragma unroll
    for (int r = 0; r < 16; ++r) bias[r] = bias_tab[h * 16 + r];
    v4i u = to_digits(u_in[tw * 32 + j], h), v = to_digits(v_in[tw * 32 + j], h);
    for (int s = 0; s < stages; ++s) butterfly(W, Wn, I, bias, u, v, h);
#pragma unroll
    for (int L = 0; L < 4; ++L) {
        out1[(tw * 32 + j) * 8 + 4 * h + L] = (uint32_t)u[L] ^ 0x80808080u;
        out2[(tw * 32 + j) * 8 + 4 * h + L] = (uint32_t)v[L] ^ 0x80808080u;
    }
}

#ifndef ITERS
#define ITERS 4096
#endif
// throughput: dependent butterfly chains in registers, like the VALU microbenchmarks of round 1
template <int CHAINS>
__global__ void __launch_bounds__(256) bench_kernel(const fe* u_in, const v4i* mats, const int* bias_tab, uint32_t* sink) {
    const uint32_t lane = threadIdx.x & 63u, j = lane & 31u, h = lane >> 5;
    const v4i W = mats[0 * 64 + lane], Wn = mats[1 * 64 + lane], I = mats[2 * 64 + lane];
    v16i bias;
#pragma unroll
    for (int r = 0; r < 16; ++r) bias[r] = bias_tab[h * 16 + r];
    v4i u[CHAINS], v[CHAINS];
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) {
        u[k] = to_digits(u_in[(threadIdx.x + k) & 31], h);
        v[k] = to_digits(u_in[32 + ((j + k) & 31)], h);
    }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int k = 0; k < CHAINS; ++k) butterfly(W, Wn, I, bias, u[k], v[k], h);
    }
    uint32_t x = 0;
#pragma unroll
    for (int k = 0; k < CHAINS; ++k)
#pragma unroll
        for (int L = 0; L < 4; ++L) x ^= (uint32_t)u[k][L] ^ (uint32_t)v[k][L];
    if (x == 0x12345678u) sink[0] = x;
}

// the shipping VALU butterfly for comparison (registers only)
__global__ void __launch_bounds__(256) bench_valu_kernel(const fe* u_in, uint32_t* sink) {
    fe u = u_in[threadIdx.x & 31], v = u_in[32 + (threadIdx.x & 31)];
    const fe w = u_in[5];
    for (int it = 0; it < ITERS; ++it) {
        const fe t = fe_mul_lazy(v, w);
        const fe a = fe_add_raw(u, t);
        const fe b = fe_sub_add_2p(u, t);
        u = fe_reduce_lazy_2p(a);
        v = fe_reduce_lazy_2p(b);
    }
    uint32_t x = 0;
#pragma unroll
    for (int L = 0; L < 8; ++L) x ^= u.v[L] ^ v.v[L];
    if (x == 0x12345678u) sink[0] = x;
}

// ---------------------------------------------------------------- host --------------------------------------------------
struct U320 { uint64_t l[5]; };
static U320 u320_zero() { U320 r; memset(&r, 0, sizeof r); return r; }
static U320 P() { U320 r = u320_zero(); r.l[0] = 1; r.l[3] = 0x0800000000000011ull; return r; }
static int ucmp(const U320& a, const U320& b) { for (int i = 4; i >= 0; --i) { if (a.l[i] != b.l[i]) return a.l[i] < b.l[i] ? -1 : 1; } return 0; }
static U320 uadd(const U320& a, const U320& b) { U320 r; unsigned __int128 c = 0; for (int i = 0; i < 5; ++i) { c += (unsigned __int128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; } return r; }
static U320 usub(const U320& a, const U320& b) { U320 r; __int128 c = 0; for (int i = 0; i < 5; ++i) { c += (__int128)a.l[i] - b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; } return r; }
static U320 dbl_mod(const U320& a) { U320 r = uadd(a, a); if (ucmp(r, P()) >= 0) r = usub(r, P()); return r; }
static U320 modp(U320 a) { while (ucmp(a, P()) >= 0) { // coarse: subtract (top) p
        uint64_t q = (a.l[3] >> 59) | (a.l[4] << 5); if (q > 1) { U320 t = u320_zero(); // (q-1) p
            unsigned __int128 m = (unsigned __int128)(q - 1); U320 pp = P(); unsigned __int128 c = 0; for (int i = 0; i < 5; ++i) { c += m * pp.l[i]; t.l[i] = (uint64_t)c; c >>= 64; } a = usub(a, t); } else a = usub(a, P()); } return a; }
static U320 from_fe(const fe& x) { U320 r = u320_zero(); for (int i = 0; i < 4; ++i) r.l[i] = (uint64_t)x.v[2 * i] | ((uint64_t)x.v[2 * i + 1] << 32); return r; }
static U320 cbal() { U320 r = u320_zero(); for (int i = 0; i < 4; ++i) r.l[i] = 0x8080808080808080ull; return r; }
static void balanced_digits(const U320& x /* < 2^255 */, int8_t d[32]) {
    U320 s = uadd(x, cbal());
    for (int k = 0; k < 32; ++k) d[k] = (int8_t)((int)((s.l[k / 8] >> (8 * (k % 8))) & 0xff) - 128);
}
// output byte position of row index i of the 32x32 result, and the lane half that receives it
static int row_pos(int i) { const int hp = (i >> 2) & 1, r = (i & 3) + 4 * (i >> 3); return 16 * hp + r; }

// A operand of the matrix "multiply by the integer w (canonical, < p)": lane l = (row i = l & 31, k half = l >> 5), 16 bytes k = 16 (l >> 5) + t
static void build_matrix(const U320& w, int8_t out[64][16]) {
    int8_t dig[32][32];   // dig[j][k] = digit k of (w 2^{8j} mod p)
    U320 x = w;
    for (int j = 0; j < 32; ++j) {
        balanced_digits(x, dig[j]);
        for (int b = 0; b < 8; ++b) x = dbl_mod(x);
    }
    for (int l = 0; l < 64; ++l) {
        const int i = l & 31, kh = l >> 5, pos = row_pos(i);
        for (int t = 0; t < 16; ++t) out[l][t] = dig[16 * kh + t][pos];
    }
}
static void build_identity(int8_t out[64][16]) {
    for (int l = 0; l < 64; ++l) { const int i = l & 31, kh = l >> 5, pos = row_pos(i); for (int t = 0; t < 16; ++t) out[l][t] = (16 * kh + t == pos) ? 1 : 0; }
}

static uint64_t rng_state = 0x5EED0000ull;
static uint64_t splitmix() { uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static fe rand_fe() { fe r; for (int i = 0; i < 4; ++i) { uint64_t z = splitmix(); r.v[2 * i] = (uint32_t)z; r.v[2 * i + 1] = (uint32_t)(z >> 32); } r.v[7] &= 0x07ffffffu; return fe_reduce_once(r); }

int main(int argc, char** argv) {
    const int NTW = 64, STAGES = argc > 1 ? atoi(argv[1]) : 1;
    // biases
    U320 K = u320_zero(); K.l[0] = 2; K.l[3] = 34;
    U320 qk = u320_zero(); { unsigned __int128 c = 0; for (int i = 0; i < 5; ++i) { c += (unsigned __int128)QMAX * K.l[i]; qk.l[i] = (uint64_t)c; c >>= 64; } }
    U320 base = u320_zero();   // 2^20 sum_k 2^{8k}
    for (int k = 0; k < 32; ++k) { U320 t = u320_zero(); const int bit = 8 * k + 20; t.l[bit / 64] = 1ull << (bit % 64); base = uadd(base, t); }
    // D = CBAL - QMAX K - base (mod p), in [0, p)
    U320 D = modp(cbal());
    D = usub(uadd(D, P()), modp(qk)); D = modp(D);
    D = usub(uadd(D, P()), modp(base)); D = modp(D);
    std::vector<int> bias(32);
    for (int pos = 0; pos < 32; ++pos) { const int delta = (int)((D.l[pos / 8] >> (8 * (pos % 8))) & 0xff); bias[pos] = (1 << 20) + delta; }   // [h][r] = pos 16h + r

    std::vector<fe> u(NTW * 32), v(NTW * 32), w(NTW);
    std::vector<int8_t> mats((size_t)NTW * 3 * 64 * 16);
    for (int t = 0; t < NTW; ++t) {
        w[t] = rand_fe();
        if (t == 0) w[t] = fe_one();
        const fe wplain = fe_from_mont(w[t]);
        U320 wi = from_fe(wplain);
        build_matrix(wi, reinterpret_cast<int8_t(*)[16]>(&mats[((size_t)t * 3 + 0) * 64 * 16]));
        U320 wneg = ucmp(wi, u320_zero()) == 0 ? wi : usub(P(), wi);
        build_matrix(wneg, reinterpret_cast<int8_t(*)[16]>(&mats[((size_t)t * 3 + 1) * 64 * 16]));
        build_identity(reinterpret_cast<int8_t(*)[16]>(&mats[((size_t)t * 3 + 2) * 64 * 16]));
        for (int j = 0; j < 32; ++j) { u[t * 32 + j] = rand_fe(); v[t * 32 + j] = rand_fe(); }
    }
    // extreme inputs in the second twiddle's columns
    for (int j = 0; j < 8; ++j) { u[32 + j] = fe_zero(); v[32 + j] = fe_zero(); }
    { fe pm1 = fe_zero(); pm1.v[6] = 0x11; pm1.v[7] = 0x08000000u; for (int j = 8; j < 16; ++j) { u[32 + j] = pm1; v[32 + j] = pm1; } }

    fe *d_u, *d_v; v4i* d_m; int* d_b; uint32_t *d_o1, *d_o2, *d_sink;
    HIPCHECK(hipMalloc(&d_u, u.size() * sizeof(fe))); HIPCHECK(hipMalloc(&d_v, v.size() * sizeof(fe)));
    HIPCHECK(hipMalloc(&d_m, mats.size())); HIPCHECK(hipMalloc(&d_b, 32 * sizeof(int)));
    HIPCHECK(hipMalloc(&d_o1, (size_t)NTW * 32 * 8 * 4)); HIPCHECK(hipMalloc(&d_o2, (size_t)NTW * 32 * 8 * 4)); HIPCHECK(hipMalloc(&d_sink, 64));
    HIPCHECK(hipMemcpy(d_u, u.data(), u.size() * sizeof(fe), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_v, v.data(), v.size() * sizeof(fe), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_m, mats.data(), mats.size(), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_b, bias.data(), 32 * sizeof(int), hipMemcpyHostToDevice));

    hipLaunchKernelGGL(check_kernel, dim3(NTW), dim3(64), 0, 0, d_u, d_v, d_m, d_b, d_o1, d_o2, STAGES);
    HIPCHECK(hipDeviceSynchronize());
    std::vector<uint32_t> o1((size_t)NTW * 32 * 8), o2(o1.size());
    HIPCHECK(hipMemcpy(o1.data(), d_o1, o1.size() * 4, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(o2.data(), d_o2, o2.size() * 4, hipMemcpyDeviceToHost));
    // expected
    U320 off = usub(uadd(u320_zero(), u320_zero()), u320_zero());
    { U320 p17 = u320_zero(); unsigned __int128 c = 0; U320 pp = P(); for (int i = 0; i < 5; ++i) { c += (unsigned __int128)17 * pp.l[i]; p17.l[i] = (uint64_t)c; c >>= 64; } off = usub(p17, cbal()); }
    int bad = 0; uint64_t maxtop = 0;
    for (int t = 0; t < NTW; ++t) for (int j = 0; j < 32; ++j) {
        fe a = u[t * 32 + j], b = v[t * 32 + j];
        for (int s = 0; s < STAGES; ++s) { const fe tt = fe_mul(b, w[t]); const fe na = fe_add(a, tt), nb = fe_sub(a, tt); a = na; b = nb; }
        for (int which = 0; which < 2; ++which) {
            const uint32_t* o = (which ? o2 : o1).data() + (size_t)(t * 32 + j) * 8;
            U320 x = u320_zero(); for (int i = 0; i < 4; ++i) x.l[i] = (uint64_t)o[2 * i] | ((uint64_t)o[2 * i + 1] << 32);
            if ((x.l[3] >> 60) > maxtop) maxtop = x.l[3] >> 60;
            U320 val = modp(uadd(x, off));
            const fe e = which ? b : a;
            U320 ev = from_fe(e);
            if (ucmp(val, ev) != 0) { if (bad < 8) printf("MISMATCH tw %d col %d out%d: got %016llx.. want %016llx.. (x' top %llx)\n", t, j, which + 1, (unsigned long long)val.l[0], (unsigned long long)ev.l[0], (unsigned long long)(x.l[3] >> 52)); ++bad; }
        }
    }
    printf("check (%d stages): %d mismatches of %d outputs; max x' >> 252 = %llu\n", STAGES, bad, NTW * 64, (unsigned long long)maxtop);

    // throughput
    hipEvent_t e0, e1; HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    const int blocks = 256 * 8;   // 8 waves per SIMD
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        HIPCHECK(hipEventRecord(e0)); hipLaunchKernelGGL(bench_kernel<1>, dim3(blocks), dim3(256), 0, 0, d_u, d_m, d_b, d_sink); HIPCHECK(hipEventRecord(e1)); HIPCHECK(hipEventSynchronize(e1));
        HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("mfma butterfly, 1 chain/wave, 8 waves/SIMD: %8.3f ms  %8.2f G butterflies/s\n", ms, (double)blocks * 4 * 32 * ITERS / ms / 1e6);
        HIPCHECK(hipEventRecord(e0)); hipLaunchKernelGGL(bench_kernel<2>, dim3(blocks / 2), dim3(256), 0, 0, d_u, d_m, d_b, d_sink); HIPCHECK(hipEventRecord(e1)); HIPCHECK(hipEventSynchronize(e1));
        HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("mfma butterfly, 2 chains/wave, 4 waves/SIMD: %8.3f ms  %8.2f G butterflies/s\n", ms, (double)(blocks / 2) * 4 * 32 * 2 * ITERS / ms / 1e6);
        HIPCHECK(hipEventRecord(e0)); hipLaunchKernelGGL(bench_valu_kernel, dim3(blocks), dim3(256), 0, 0, d_u, d_sink); HIPCHECK(hipEventRecord(e1)); HIPCHECK(hipEventSynchronize(e1));
        HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("VALU butterfly (shipping arithmetic), 8 waves/SIMD: %8.3f ms  %8.2f G butterflies/s\n", ms, (double)blocks * 256 * ITERS / ms / 1e6);
    }
    return bad != 0;
}
