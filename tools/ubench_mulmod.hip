// Isolated throughput of the Stark252 Montgomery product (csrc/fp.h fe_mul), fe_add and fe_sub: registers only, up to 8
// waves per SIMD.  Separates "how fast is the arithmetic" from "how well does a kernel feed it".
#include "../lambdaworks_cairo_prover_amd/csrc/fp.h"
#include <cstdio>
#define ITERS 256
template <int OP, int CHAINS>
__global__ void __launch_bounds__(256) k(fe* out, const fe* in) {
    fe x[CHAINS], y = in[threadIdx.x & 63];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = in[(threadIdx.x + c) & 63];
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (OP == 0) x[c] = fe_mul(x[c], y);
            else if (OP == 1) x[c] = fe_add(x[c], y);
            else if (OP == 2) x[c] = fe_sub(x[c], y);
            else { fe t = fe_mul(x[c], y); fe u = x[(c + 1) % CHAINS]; x[c] = fe_add(u, t); x[(c + 1) % CHAINS] = fe_sub(u, t); }  // butterfly
        }
    }
    fe acc = x[0];
#pragma unroll
    for (int c = 1; c < CHAINS; ++c) acc = fe_add(acc, x[c]);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int OP, int CHAINS>
void run(const char* name, fe* d_out, fe* d_in, int blocks_per_cu) {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    dim3 grid(prop.multiProcessorCount * blocks_per_cu), block(256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP, CHAINS>), grid, block, 0, 0, d_out, d_in); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP, CHAINS>), grid, block, 0, 0, d_out, d_in);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)grid.x * 256 * ITERS * CHAINS;
    printf("%-34s chains=%d blocks/CU=%d  %8.3f ms  %8.2f G ops/s\n", name, CHAINS, blocks_per_cu, ms, ops / ms / 1e6);
}
int main() {
    fe h[64];
    for (int i = 0; i < 64; ++i) for (int j = 0; j < 8; ++j) h[i].v[j] = 0x01234567u * (i + 3) + 0x9e3779b9u * j + (j == 7 ? 0 : 0x80000000u);
    for (int i = 0; i < 64; ++i) h[i].v[7] &= 0x07ffffff;
    fe *d_in, *d_out; (void)hipMalloc(&d_in, sizeof(h)); (void)hipMalloc(&d_out, sizeof(fe) * 256 * 8 * 256);
    (void)hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0, 1>("fe_mul", d_out, d_in, 8); run<0, 2>("fe_mul", d_out, d_in, 8); run<0, 4>("fe_mul", d_out, d_in, 4);
    run<0, 1>("fe_mul", d_out, d_in, 4); run<0, 1>("fe_mul", d_out, d_in, 2);
    run<1, 2>("fe_add", d_out, d_in, 8); run<2, 2>("fe_sub", d_out, d_in, 8);
    run<3, 2>("butterfly (mul+add+sub)", d_out, d_in, 8); run<3, 4>("butterfly (mul+add+sub)", d_out, d_in, 4);
    return 0;
}
