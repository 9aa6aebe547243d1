/* Plain-C caller of the boundary (what a cgo / FFI shim links against): builds with `gcc -std=c99`, needs no GPU to compile.
 *   gcc -std=c99 -Iinclude examples/c_abi_smoke.c -Llambdaworks_cairo_prover_amd -lstark252_hip -Wl,-rpath,$PWD/lambdaworks_cairo_prover_amd -o c_abi_smoke
 * Without a GPU it reports SP_E_NO_DEVICE and exits 0 (the library has no CPU fallback); with one it runs a 2^10 NTT round trip and then the
 * whole path of the reference's CLI (src/main.rs:85-108: run a program, prove, verify): front-end run -> sp_prewarm -> sp_cairo_prove_run
 * (the main trace built on the device) -> sp_cairo_verify -> the CLI's proof-file framing. */
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

    /* generate_prover_args + generate_cairo_proof + verify_cairo_proof (reference src/cairo/runner/run.rs:242-263, src/cairo/air.rs:1165-1182) */
    sp_proof_options opt;
    memset(&opt, 0, sizeof opt);
    opt.blowup_factor = 4; opt.fri_number_of_queries = 3; opt.coset_offset = 3; opt.grinding_factor = 1;   /* default_test_options, options.rs:144-151 */
    sp_cairo_run* run = NULL;
    rc = sp_cairo_run_fibonacci(100, &run);          /* 709 steps -> 2^10 rows x 34 columns */
    uint64_t rows = 0, steps = 0; uint32_t cols = 0;
    if (rc == 0) rc = sp_cairo_run_shape(run, &rows, &cols, &steps);
    if (rc == 0) rc = sp_prewarm_cancel(ctx);                                 /* the trace already exists here: no clock ramp (a caller with a VM to run calls this when the VM is done) */
    if (rc == 0) rc = sp_prewarm(ctx, rows, cols, 18, 0, &opt, 0);            /* optional: a caller does this on a thread beside its VM */
    uint8_t* proof = NULL; uint64_t proof_len = 0;
    if (rc == 0) rc = sp_cairo_prove_run(ctx, run, &opt, &proof, &proof_len);
    sp_cairo_public_inputs pub;
    int accepted = 0;
    if (rc == 0) rc = sp_cairo_run_public_inputs(run, &pub);
    if (rc == 0) accepted = sp_cairo_verify(proof, proof_len, &pub, &opt);
    uint8_t* file = NULL; uint64_t file_len = 0;
    if (rc == 0) rc = sp_proof_file_encode(proof, proof_len, run, &file, &file_len);
    printf("cairo proof: rc %d (%s), %llu steps, %llu x %u trace, %llu proof bytes, verifier %s, proof file %llu bytes\n", rc, rc ? sp_last_error() : "ok",
           (unsigned long long)steps, (unsigned long long)rows, cols, (unsigned long long)proof_len, accepted == 1 ? "accepts" : "REJECTS",
           (unsigned long long)file_len);
    ok = ok && rc == 0 && accepted == 1 && file_len > proof_len + 8;
    sp_free(file); sp_free(proof);
    sp_cairo_run_free(run);
    sp_ctx_destroy(ctx);
    return ok ? 0 : 1;
}
