// Prototype: Stark252 Montgomery product on 9 x 28-bit limbs with 64-bit column accumulators (no carry chains in the
// product; the reduction by p = 2^251 + 17*2^192 + 1 is shifts/adds because p = [1,0,0,0,0,0,2^24,1,2^27] in radix 2^28).
// Checks mul9 against fe_mul and compares throughput of mul and of a radix-2 butterfly.
//   hipcc -O3 --offload-arch=gfx950 -I lambdaworks_cairo_prover_amd/csrc tools/ubench_mul9.hip -o tools/bin/ubench_mul9
#include "../../lambdaworks_cairo_prover_amd/csrc/fp.h"
#include <cstdio>
#include <vector>
#ifndef ITERS
#define ITERS 256
#endif

struct fe9 { uint32_t l[9]; };
#define M28 0x0fffffffu

__host__ __device__ __forceinline__ fe9 unpack9(const fe& a) {
    fe9 r;
    r.l[0] = a.v[0] & M28;
    r.l[1] = ((a.v[0] >> 28) | (a.v[1] << 4)) & M28;
    r.l[2] = ((a.v[1] >> 24) | (a.v[2] << 8)) & M28;
    r.l[3] = ((a.v[2] >> 20) | (a.v[3] << 12)) & M28;
    r.l[4] = ((a.v[3] >> 16) | (a.v[4] << 16)) & M28;
    r.l[5] = ((a.v[4] >> 12) | (a.v[5] << 20)) & M28;
    r.l[6] = ((a.v[5] >> 8) | (a.v[6] << 24)) & M28;
    r.l[7] = a.v[6] >> 4;
    r.l[8] = a.v[7];
    return r;
}
// tight limbs (l[0..7] < 2^28) -> 8 x 32
__host__ __device__ __forceinline__ fe pack9(const fe9& a) {
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

// a: any limbs < 2^32 ("loose"), w: tight limbs (< 2^28, value < 2^252). Returns tight limbs, value = a*w*2^-252 mod p,
// value < a_value/2^252 * w + p.
__device__ __forceinline__ fe9 mul9(const fe9& a, const fe9& w) {
    uint64_t D[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) D[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
#pragma unroll
        for (int j = 0; j < 9; ++j) D[i + j] += (uint64_t)a.l[i] * w.l[j];
        const uint32_t m = (0u - (uint32_t)D[i]) & M28;
        const uint64_t s = D[i] + m;               // low 28 bits are zero
        D[i + 1] += s >> 28;
        D[i + 6] += (uint64_t)m * 0x11000000u;     // m * (2^24 + 2^28): limbs 6 and 7 of p
        D[i + 8] += (uint64_t)m * 0x08000000u;     // m * 2^27: limb 8 of p
    }
    fe9 r;
#pragma unroll
    for (int k = 9; k < 17; ++k) { D[k + 1] += D[k] >> 28; r.l[k - 9] = (uint32_t)D[k] & M28; }
    r.l[8] = (uint32_t)D[17];
    return r;
}


__device__ __forceinline__ uint64_t shr28(uint64_t x) {
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    return ((uint64_t)(hi >> 28) << 32) | __builtin_amdgcn_alignbit(hi, lo, 28);
}
__device__ __forceinline__ void mad_acc(uint64_t& d, uint32_t a, uint32_t b) {
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b) : "vcc");
}
// VAR bit 0: m*2^27 through a mad instead of a 64-bit shift; bit 1: >>28 through alignbit; bit 2: carry as (D>>28) + (m != 0)
template <int VAR>
__device__ __forceinline__ fe9 mul9v(const fe9& a, const fe9& w) {
    uint64_t D[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) D[k] = 0;
    const uint32_t c27 = 0x08000000u;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
#pragma unroll
        for (int j = 0; j < 9; ++j) D[i + j] += (uint64_t)a.l[i] * w.l[j];
        const uint32_t m = (0u - (uint32_t)D[i]) & M28;
        if (VAR & 4) {
            D[i + 1] += ((VAR & 2) ? shr28(D[i]) : (D[i] >> 28)) + (m != 0 ? 1u : 0u);
        } else {
            const uint64_t s = D[i] + m;
            D[i + 1] += (VAR & 2) ? shr28(s) : (s >> 28);
        }
        D[i + 6] += (uint64_t)m * 0x11000000u;
        if (VAR & 1) mad_acc(D[i + 8], m, c27); else D[i + 8] += (uint64_t)m * 0x08000000u;
    }
    fe9 r;
#pragma unroll
    for (int k = 9; k < 17; ++k) { D[k + 1] += (VAR & 2) ? shr28(D[k]) : (D[k] >> 28); r.l[k - 9] = (uint32_t)D[k] & M28; }
    r.l[8] = (uint32_t)D[17];
    return r;
}
// k*p with every limb pre-charged so that limbwise subtraction of tight limbs never borrows:
// c_i = (kp)_i + 2^29 - 2 for 0 < i < 8, c_0 = (kp)_0 + 2^29, c_8 = (kp)_8 - 2.
template <int K>
__device__ __forceinline__ fe9 sub_bias() {
    // p limbs: [1,0,0,0,0,0,2^24,1,2^27]
    fe9 c;
    c.l[0] = K * 1u + (1u << 29);
    for (int i = 1; i < 8; ++i) c.l[i] = (1u << 29) - 2u;
    c.l[6] += K * (1u << 24);
    c.l[7] += K * 1u;
    c.l[8] = K * (1u << 27) - 2u;
    return c;
}

template <int OP>
__global__ void __launch_bounds__(256) k(fe* out, const fe* in) {
    fe x = in[threadIdx.x & 63], y = in[(threadIdx.x + 7) & 63];
    if (OP == 0) {
        for (int it = 0; it < ITERS; ++it) x = fe_mul(x, y);
        out[blockIdx.x * 256 + threadIdx.x] = x;
    } else if (OP == 1) {
        fe9 a = unpack9(x), w = unpack9(y);
        for (int it = 0; it < ITERS; ++it) a = mul9(a, w);
        out[blockIdx.x * 256 + threadIdx.x] = pack9(a);
    } else if (OP >= 10 && OP < 18) {
        fe9 a = unpack9(x), w = unpack9(y);
        for (int it = 0; it < ITERS; ++it) a = mul9v<OP - 10>(a, w);
        out[blockIdx.x * 256 + threadIdx.x] = pack9(a);
    } else if (OP == 2) {  // classic butterfly
        fe u = x, v = y, w = in[(threadIdx.x + 13) & 63];
        for (int it = 0; it < ITERS; ++it) { fe t = fe_mul(v, w); v = fe_sub(u, t); u = fe_add(u, t); }
        out[blockIdx.x * 256 + threadIdx.x] = fe_add(u, v);
    } else if (OP == 3) {  // 28-bit butterfly, lazy: x = u + t, y = u - t + 2p(biased); renormalised by the next product
        fe9 u = unpack9(x), v = unpack9(y), w = unpack9(in[(threadIdx.x + 13) & 63]);
        const fe9 bias = sub_bias<4>();
        for (int it = 0; it < ITERS; ++it) {
            fe9 t = mul9(v, w);
            fe9 nu, nv;
#pragma unroll
            for (int i = 0; i < 9; ++i) { nu.l[i] = u.l[i] + t.l[i]; nv.l[i] = u.l[i] + (bias.l[i] - t.l[i]); }
            // keep the additive chain bounded for the benchmark: route both through a product every iteration
            u = mul9(nu, w);  // (only to keep values bounded in this timing loop; counted as 2 products per iteration)
            v = nv;
        }
        fe9 s;
#pragma unroll
        for (int i = 0; i < 9; ++i) s.l[i] = u.l[i] + v.l[i];
        fe9 one; for (int i = 0; i < 9; ++i) one.l[i] = 0; one.l[0] = 1;
        out[blockIdx.x * 256 + threadIdx.x] = pack9(mul9(s, one));
    }
}

__global__ void check_kernel(const fe* a, const fe* w, fe* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe r0 = pack9(mul9(unpack9(a[i]), unpack9(w[i])));
    fe r7 = pack9(mul9v<7>(unpack9(a[i]), unpack9(w[i])));
    fe r3 = pack9(mul9v<3>(unpack9(a[i]), unpack9(w[i])));
    if (!fe_eq(r0, r7) || !fe_eq(r0, r3)) r0.v[7] = 0xffffffffu;  // poison: reported as a mismatch
    out[i] = r0;
}

template <int OP>
void run(const char* name, fe* d_out, fe* d_in, double ops_per_iter) {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    dim3 grid(prop.multiProcessorCount * 8), block(256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP>), grid, block, 0, 0, d_out, d_in); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL((k<OP>), grid, block, 0, 0, d_out, d_in);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 4;
    double ops = (double)grid.x * 256 * ITERS * ops_per_iter;
    printf("%-48s %8.3f ms  %8.2f G ops/s\n", name, ms, ops / ms / 1e6);
}

static fe fe_modulus() { fe p = fe_zero(); p.v[0] = SP_P0; p.v[6] = SP_P6; p.v[7] = SP_P7; return p; }
static bool ge_p(const fe& a) { fe p = fe_modulus(); for (int i = 7; i >= 0; --i) { if (a.v[i] != p.v[i]) return a.v[i] > p.v[i]; } return true; }
static fe sub_p(const fe& a) { fe p = fe_modulus(); fe r; uint64_t b = 0; for (int i = 0; i < 8; ++i) { uint64_t d = (uint64_t)a.v[i] - p.v[i] - b; r.v[i] = (uint32_t)d; b = (d >> 63) & 1; } return r; }

int main() {
    const int N = 4096;
    std::vector<fe> ha(N), hw(N), ho(N);
    uint64_t s = 0x9e3779b97f4a7c15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    for (int i = 0; i < N; ++i) {
        for (int j = 0; j < 8; ++j) { ha[i].v[j] = rnd(); hw[i].v[j] = rnd(); }
        ha[i].v[7] &= 0x07ffffffu; hw[i].v[7] &= 0x07ffffffu;
        if (i == 0) { for (int j = 0; j < 8; ++j) ha[i].v[j] = 0; }
        if (i == 1) { ha[i] = fe_modulus(); ha[i].v[0] -= 1; hw[i] = ha[i]; }  // (p-1)^2
        if (i == 2) { for (int j = 0; j < 8; ++j) { ha[i].v[j] = 0xffffffffu; hw[i].v[j] = 0xffffffffu; } ha[i].v[7] = 0x07ffffffu; hw[i].v[7] = 0x07ffffffu; }
        if (i > 2 && ge_p(ha[i])) ha[i].v[7] &= 0x03ffffffu;
        if (i > 2 && ge_p(hw[i])) hw[i].v[7] &= 0x03ffffffu;
    }
    fe *da, *dw, *dout;
    (void)hipMalloc(&da, N * sizeof(fe)); (void)hipMalloc(&dw, N * sizeof(fe)); (void)hipMalloc(&dout, sizeof(fe) * 256 * 8 * 256);
    (void)hipMemcpy(da, ha.data(), N * sizeof(fe), hipMemcpyHostToDevice);
    (void)hipMemcpy(dw, hw.data(), N * sizeof(fe), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check_kernel, dim3(N / 256), dim3(256), 0, 0, da, dw, dout, N);
    (void)hipMemcpy(ho.data(), dout, N * sizeof(fe), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < N; ++i) {
        if (i == 2) continue;  // operands >= p: only checked for not crashing
        fe e = fe_mul(ha[i], hw[i]);              // A*W*2^-256
        for (int d = 0; d < 4; ++d) e = fe_add(e, e);  // *16 -> A*W*2^-252
        fe g = ho[i];
        int guard = 0;
        while (ge_p(g) && guard++ < 64) g = sub_p(g);
        if (!fe_eq(g, e)) { if (bad < 5) printf("MISMATCH at %d\n", i); ++bad; }
    }
    printf("mul9 check: %d mismatches of %d\n", bad, N);
    run<0>("fe_mul (8x32 CIOS)", dout, da, 1);
    run<1>("mul9 (9x28 columns)", dout, da, 1);
    run<10>("mul9v<0>", dout, da, 1); run<11>("mul9v<1> mad27", dout, da, 1); run<12>("mul9v<2> alignbit", dout, da, 1);
    run<13>("mul9v<3> mad27+alignbit", dout, da, 1); run<14>("mul9v<4> carry-cmp", dout, da, 1); run<15>("mul9v<5>", dout, da, 1);
    run<16>("mul9v<6>", dout, da, 1); run<17>("mul9v<7>", dout, da, 1);
    run<2>("butterfly 8x32 (mul + add + sub)", dout, da, 1);
    run<3>("2 x mul9 + lazy add/sub (per iteration)", dout, da, 1);
    return bad != 0;
}
