// Keccak-f[1600] round variants on gfx950, round 6: does the theta application pay better as 60 two-input v_xor_b32 (VOP2, 1.0 issue
// units) than as 50 three-input v_bitop3_b32 (VOP3 with three VGPR sources: 1.6 units like v_bfi / v_alignbit)?
//   V0: shipping round (keccak.h): 120 v_bitop3 + 58 v_alignbit + iota
//   V1: d_x = c_(x-1) ^ rot(c_(x+1), 1) once per column (10 v_xor), lanes ^= d_x (50 v_xor): 70 v_bitop3 + 58 v_alignbit + 62 v_xor
//   V2: V1 with the column parities as four two-input xors too: 50 v_bitop3 + 58 v_alignbit + 102 v_xor
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/ubench_keccak2.hip -o tools/bin/ubench_keccak2
#include "../../lambdaworks_cairo_prover_amd/csrc/keccak.h"
#include <cstdio>
#ifndef PERMS
#define PERMS 2048
#endif

#define ROUND_V(PAR, rc)                                                                                      \
    {                                                                                                         \
        const uint64_t c0 = PAR(s[0], s[5], s[10], s[15], s[20]);                                             \
        const uint64_t c1 = PAR(s[1], s[6], s[11], s[16], s[21]);                                             \
        const uint64_t c2 = PAR(s[2], s[7], s[12], s[17], s[22]);                                             \
        const uint64_t c3 = PAR(s[3], s[8], s[13], s[18], s[23]);                                             \
        const uint64_t c4 = PAR(s[4], s[9], s[14], s[19], s[24]);                                             \
        const uint64_t d0 = c4 ^ sp_rotl64(c1, 1), d1 = c0 ^ sp_rotl64(c2, 1), d2 = c1 ^ sp_rotl64(c3, 1),    \
                       d3 = c2 ^ sp_rotl64(c4, 1), d4 = c3 ^ sp_rotl64(c0, 1);                                \
        const uint64_t b0 = s[0] ^ d0;                                                                        \
        const uint64_t b1 = sp_rotl64(s[6] ^ d1, 44);                                                         \
        const uint64_t b2 = sp_rotl64(s[12] ^ d2, 43);                                                        \
        const uint64_t b3 = sp_rotl64(s[18] ^ d3, 21);                                                        \
        const uint64_t b4 = sp_rotl64(s[24] ^ d4, 14);                                                        \
        const uint64_t b5 = sp_rotl64(s[3] ^ d3, 28);                                                         \
        const uint64_t b6 = sp_rotl64(s[9] ^ d4, 20);                                                         \
        const uint64_t b7 = sp_rotl64(s[10] ^ d0, 3);                                                         \
        const uint64_t b8 = sp_rotl64(s[16] ^ d1, 45);                                                        \
        const uint64_t b9 = sp_rotl64(s[22] ^ d2, 61);                                                        \
        const uint64_t b10 = sp_rotl64(s[1] ^ d1, 1);                                                         \
        const uint64_t b11 = sp_rotl64(s[7] ^ d2, 6);                                                         \
        const uint64_t b12 = sp_rotl64(s[13] ^ d3, 25);                                                       \
        const uint64_t b13 = sp_rotl64(s[19] ^ d4, 8);                                                        \
        const uint64_t b14 = sp_rotl64(s[20] ^ d0, 18);                                                       \
        const uint64_t b15 = sp_rotl64(s[4] ^ d4, 27);                                                        \
        const uint64_t b16 = sp_rotl64(s[5] ^ d0, 36);                                                        \
        const uint64_t b17 = sp_rotl64(s[11] ^ d1, 10);                                                       \
        const uint64_t b18 = sp_rotl64(s[17] ^ d2, 15);                                                       \
        const uint64_t b19 = sp_rotl64(s[23] ^ d3, 56);                                                       \
        const uint64_t b20 = sp_rotl64(s[2] ^ d2, 62);                                                        \
        const uint64_t b21 = sp_rotl64(s[8] ^ d3, 55);                                                        \
        const uint64_t b22 = sp_rotl64(s[14] ^ d4, 39);                                                       \
        const uint64_t b23 = sp_rotl64(s[15] ^ d0, 41);                                                       \
        const uint64_t b24 = sp_rotl64(s[21] ^ d1, 2);                                                        \
        s[0] = sp_chi(b0, b1, b2) ^ (rc);                                                                     \
        s[1] = sp_chi(b1, b2, b3); s[2] = sp_chi(b2, b3, b4); s[3] = sp_chi(b3, b4, b0);                      \
        s[4] = sp_chi(b4, b0, b1);                                                                            \
        s[5] = sp_chi(b5, b6, b7); s[6] = sp_chi(b6, b7, b8); s[7] = sp_chi(b7, b8, b9);                      \
        s[8] = sp_chi(b8, b9, b5); s[9] = sp_chi(b9, b5, b6);                                                 \
        s[10] = sp_chi(b10, b11, b12); s[11] = sp_chi(b11, b12, b13); s[12] = sp_chi(b12, b13, b14);          \
        s[13] = sp_chi(b13, b14, b10); s[14] = sp_chi(b14, b10, b11);                                         \
        s[15] = sp_chi(b15, b16, b17); s[16] = sp_chi(b16, b17, b18); s[17] = sp_chi(b17, b18, b19);          \
        s[18] = sp_chi(b18, b19, b15); s[19] = sp_chi(b19, b15, b16);                                         \
        s[20] = sp_chi(b20, b21, b22); s[21] = sp_chi(b21, b22, b23); s[22] = sp_chi(b22, b23, b24);          \
        s[23] = sp_chi(b23, b24, b20); s[24] = sp_chi(b24, b20, b21);                                         \
    }
#define PAR3(a, b, c, d, e) sp_xor3(sp_xor3(a, b, c), d, e)
#define PAR2(a, b, c, d, e) ((((a) ^ (b)) ^ ((c) ^ (d))) ^ (e))

template <int V>
__global__ void __launch_bounds__(256) k(uint64_t* out, const uint64_t* in) {
    uint64_t s[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) s[i] = in[(threadIdx.x + i * 7) & 255] + i;
    for (int p = 0; p < PERMS; ++p) {
        if (V == 0) { sp_keccak_f1600_dev(s); }
        else if (V == 1) {
#pragma unroll 1
            for (int r = 0; r < 24; ++r) ROUND_V(PAR3, SP_KECCAK_RC_DEV[r])
        } else {
#pragma unroll 1
            for (int r = 0; r < 24; ++r) ROUND_V(PAR2, SP_KECCAK_RC_DEV[r])
        }
    }
    uint64_t x = 0;
#pragma unroll
    for (int i = 0; i < 25; ++i) x ^= s[i];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

template <int V>
double run(const char* name, uint64_t* d_out, uint64_t* d_in, uint64_t* h_out, int waves) {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    dim3 grid(prop.multiProcessorCount * waves / 4), block(256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<V>), grid, block, 0, 0, d_out, d_in); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<V>), grid, block, 0, 0, d_out, d_in);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double perms = (double)grid.x * 256 * PERMS;
    (void)hipMemcpy(h_out, d_out, 8 * 1024, hipMemcpyDeviceToHost);
    printf("%-64s waves/CU %2d %8.3f ms  %7.3f G perms/s   out[0..1] = %016llx %016llx\n", name, waves, ms, perms / ms / 1e6, (unsigned long long)h_out[0], (unsigned long long)h_out[1]);
    return ms;
}

int main() {
    uint64_t h[256], ho[1024];
    for (int i = 0; i < 256; ++i) h[i] = 0x9e3779b97f4a7c15ull * (i + 1);
    uint64_t *d_in, *d_out; (void)hipMalloc(&d_in, sizeof(h)); (void)hipMalloc(&d_out, 8u * 256 * 8 * 256);
    (void)hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep)
        for (int waves : {32, 16, 8}) {
            run<0>("V0 shipping: 120 bitop3 + 58 alignbit", d_out, d_in, ho, waves);
            uint64_t a0 = ho[0], a1 = ho[1];
            run<1>("V1 theta via d_x, two-input xors: 70 bitop3 + 58 alignbit + 62 xor", d_out, d_in, ho, waves);
            bool ok = a0 == ho[0] && a1 == ho[1];
            run<2>("V2 V1 + two-input parities: 50 bitop3 + 58 alignbit + 102 xor", d_out, d_in, ho, waves);
            ok = ok && a0 == ho[0] && a1 == ho[1];
            printf("outputs %s\n", ok ? "MATCH" : "DIFFER");
        }
    return 0;
}
