// Keccak-f[1600] round variants on gfx950: plain (xor / bfi / alignbit) vs v_bitop3_b32 (3-input logic) for the theta
// parities, the theta application and chi.  One permutation chain per lane, 8 waves per SIMD, long enough to settle clocks.
//   hipcc -O3 --offload-arch=gfx950 -I lambdaworks_cairo_prover_amd/csrc tools/experiments/ubench_keccak.hip -o tools/bin/ubench_keccak
#include "../../lambdaworks_cairo_prover_amd/csrc/keccak.h"
#include <cstdio>
#ifndef PERMS
#define PERMS 512
#endif

__device__ __forceinline__ uint64_t x3(uint64_t a, uint64_t b, uint64_t c) {
    uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, 0x96);
    uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), 0x96);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t chi3(uint64_t a, uint64_t b, uint64_t c) {  // a ^ (~b & c)
    uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, 0xd2);
    uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), 0xd2);
    return ((uint64_t)hi << 32) | lo;
}
#define R3(x, n) sp_rotl64((x), (n))
#define KECCAK_ROUND_B3(rc)                                                                                   \
    {                                                                                                         \
        uint64_t c0 = x3(x3(s[0], s[5], s[10]), s[15], s[20]);                                                \
        uint64_t c1 = x3(x3(s[1], s[6], s[11]), s[16], s[21]);                                                \
        uint64_t c2 = x3(x3(s[2], s[7], s[12]), s[17], s[22]);                                                \
        uint64_t c3 = x3(x3(s[3], s[8], s[13]), s[18], s[23]);                                                \
        uint64_t c4 = x3(x3(s[4], s[9], s[14]), s[19], s[24]);                                                \
        uint64_t r0 = R3(c0, 1), r1 = R3(c1, 1), r2 = R3(c2, 1), r3 = R3(c3, 1), r4 = R3(c4, 1);              \
        /* theta applied inside the rho input: s ^ c[x-1] ^ rot(c[x+1]) */                                    \
        uint64_t b0 = x3(s[0], c4, r1);                                                                       \
        uint64_t b1 = R3(x3(s[6], c0, r2), 44);                                                               \
        uint64_t b2 = R3(x3(s[12], c1, r3), 43);                                                              \
        uint64_t b3 = R3(x3(s[18], c2, r4), 21);                                                              \
        uint64_t b4 = R3(x3(s[24], c3, r0), 14);                                                              \
        uint64_t b5 = R3(x3(s[3], c2, r4), 28);                                                               \
        uint64_t b6 = R3(x3(s[9], c3, r0), 20);                                                               \
        uint64_t b7 = R3(x3(s[10], c4, r1), 3);                                                               \
        uint64_t b8 = R3(x3(s[16], c0, r2), 45);                                                              \
        uint64_t b9 = R3(x3(s[22], c1, r3), 61);                                                              \
        uint64_t b10 = R3(x3(s[1], c0, r2), 1);                                                               \
        uint64_t b11 = R3(x3(s[7], c1, r3), 6);                                                               \
        uint64_t b12 = R3(x3(s[13], c2, r4), 25);                                                             \
        uint64_t b13 = R3(x3(s[19], c3, r0), 8);                                                              \
        uint64_t b14 = R3(x3(s[20], c4, r1), 18);                                                             \
        uint64_t b15 = R3(x3(s[4], c3, r0), 27);                                                              \
        uint64_t b16 = R3(x3(s[5], c4, r1), 36);                                                              \
        uint64_t b17 = R3(x3(s[11], c0, r2), 10);                                                             \
        uint64_t b18 = R3(x3(s[17], c1, r3), 15);                                                             \
        uint64_t b19 = R3(x3(s[23], c2, r4), 56);                                                             \
        uint64_t b20 = R3(x3(s[2], c1, r3), 62);                                                              \
        uint64_t b21 = R3(x3(s[8], c2, r4), 55);                                                              \
        uint64_t b22 = R3(x3(s[14], c3, r0), 39);                                                             \
        uint64_t b23 = R3(x3(s[15], c4, r1), 41);                                                             \
        uint64_t b24 = R3(x3(s[21], c0, r2), 2);                                                              \
        s[0] = chi3(b0, b1, b2) ^ (rc);                                                                       \
        s[1] = chi3(b1, b2, b3); s[2] = chi3(b2, b3, b4); s[3] = chi3(b3, b4, b0); s[4] = chi3(b4, b0, b1);   \
        s[5] = chi3(b5, b6, b7); s[6] = chi3(b6, b7, b8); s[7] = chi3(b7, b8, b9); s[8] = chi3(b8, b9, b5);   \
        s[9] = chi3(b9, b5, b6);                                                                              \
        s[10] = chi3(b10, b11, b12); s[11] = chi3(b11, b12, b13); s[12] = chi3(b12, b13, b14);                \
        s[13] = chi3(b13, b14, b10); s[14] = chi3(b14, b10, b11);                                             \
        s[15] = chi3(b15, b16, b17); s[16] = chi3(b16, b17, b18); s[17] = chi3(b17, b18, b19);                \
        s[18] = chi3(b18, b19, b15); s[19] = chi3(b19, b15, b16);                                             \
        s[20] = chi3(b20, b21, b22); s[21] = chi3(b21, b22, b23); s[22] = chi3(b22, b23, b24);                \
        s[23] = chi3(b23, b24, b20); s[24] = chi3(b24, b20, b21);                                             \
    }

template <int V>
__global__ void __launch_bounds__(256) k(uint64_t* out, const uint64_t* in) {
    uint64_t s[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) s[i] = in[(threadIdx.x + i * 7) & 255] + i;
    for (int p = 0; p < PERMS; ++p) {
        if (V == 0) { sp_keccak_f1600_dev(s); }
        else {
#pragma unroll 1
            for (int r = 0; r < 24; ++r) KECCAK_ROUND_B3(SP_KECCAK_RC_DEV[r])
        }
    }
    uint64_t x = 0;
#pragma unroll
    for (int i = 0; i < 25; ++i) x ^= s[i];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

template <int V>
double run(const char* name, uint64_t* d_out, uint64_t* d_in, uint64_t* h_out) {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    dim3 grid(prop.multiProcessorCount * 8), block(256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<V>), grid, block, 0, 0, d_out, d_in); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<V>), grid, block, 0, 0, d_out, d_in);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double perms = (double)grid.x * 256 * PERMS;
    (void)hipMemcpy(h_out, d_out, 8 * 1024, hipMemcpyDeviceToHost);
    printf("%-40s %8.3f ms  %7.3f G perms/s   out[0..1] = %016llx %016llx\n", name, ms, perms / ms / 1e6, (unsigned long long)h_out[0], (unsigned long long)h_out[1]);
    return ms;
}

int main() {
    uint64_t h[256], ho[1024];
    for (int i = 0; i < 256; ++i) h[i] = 0x9e3779b97f4a7c15ull * (i + 1);
    uint64_t *d_in, *d_out; (void)hipMalloc(&d_in, sizeof(h)); (void)hipMalloc(&d_out, 8u * 256 * 8 * 256);
    (void)hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>("keccak-f plain (xor/bfi/alignbit)", d_out, d_in, ho);
    uint64_t a0 = ho[0], a1 = ho[1];
    run<1>("keccak-f with v_bitop3_b32", d_out, d_in, ho);
    printf("outputs %s\n", (a0 == ho[0] && a1 == ho[1]) ? "MATCH" : "DIFFER");
    return 0;
}
