/* Plain-C caller of the boundary (what a cgo / FFI shim links against): builds with `gcc -std=c99`, needs no GPU to compile.
 *   gcc -std=c99 -Iinclude examples/c_abi_smoke.c -Llambdaworks_cairo_prover_amd -lstark252_hip -Wl,-rpath,$PWD/lambdaworks_cairo_prover_amd -o c_abi_smoke
 * Without a GPU it reports SP_E_NO_DEVICE and exits 0 (the library has no CPU fallback); with one it runs a 2^10 NTT round trip. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "stark252_hip.h"

int main(void) {
    sp_config cfg;
    sp_ctx* ctx = NULL;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = 0;
    cfg.fe_encoding = SP_FE_CANON_BE;
    int rc = sp_ctx_create(&ctx, &cfg);
    if (rc != 0) {
        printf("sp_ctx_create: %d (%s)\n", rc, sp_last_error());
        return rc == SP_E_NO_DEVICE ? 0 : 1;
    }
    const uint64_t n = 1024;
    uint8_t* v = (uint8_t*)calloc(n, 32);
    uint8_t* w = (uint8_t*)malloc(n * 32);
    for (uint64_t i = 0; i < n; ++i) { v[32 * i + 31] = (uint8_t)(i * 7 + 1); v[32 * i + 30] = (uint8_t)(i >> 3); }
    memcpy(w, v, n * 32);
    rc = sp_ntt(ctx, w, n, 0, NULL);                 /* evaluate_fft */
    if (rc == 0) rc = sp_ntt(ctx, w, n, 1, NULL);    /* interpolate_fft */
    printf("ntt round trip: rc %d, %s\n", rc, (rc == 0 && memcmp(v, w, n * 32) == 0) ? "identical" : "MISMATCH");
    int ok = rc == 0 && memcmp(v, w, n * 32) == 0;
    free(v); free(w);
    sp_ctx_destroy(ctx);
    return ok ? 0 : 1;
}
