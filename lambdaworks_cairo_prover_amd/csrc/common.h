// Shared host-side helpers for the stark252 HIP library.
#pragma once
#include "../../include/stark252_hip.h"
#include "fp.h"
#include <hip/hip_runtime.h>
#include <string>

void sp_set_error(const std::string& s);

#define SP_HIP_CHECK(expr)                                                                              \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            sp_set_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
            return SP_E_HIP;                                                                            \
        }                                                                                               \
    } while (0)

#define SP_TRY(expr)                \
    do {                            \
        int _r = (expr);            \
        if (_r != SP_OK) return _r; \
    } while (0)

// Waits for a stream by polling it: the proof's Fiat-Shamir round trips (root down, challenge up) are latency, and a blocking
// hipStreamSynchronize costs ~20 us of wake-up each.  Only for waits that are short by construction (the proof path).
static inline hipError_t sp_stream_wait_polling(hipStream_t st) {
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
    }
}

static inline int sp_log2_exact(uint64_t n) {
    if (n == 0 || (n & (n - 1))) return -1;
    int k = 0;
    while ((1ULL << k) < n) ++k;
    return k;
}

// Order of the LDE evaluations inside a column.  Natural: element i = p(h w_N^i).  Coset-major
// (the prover's trace and composition columns): the 2^log_bloc cosets this rank holds one after the other, 2^logn rows
// each; the evaluation with local natural index e = m * b_loc + c_loc sits at c_loc * n + m.
struct LdeOrder {
    uint32_t coset_major, log_bloc, logn;
    SP_HD uint64_t at(uint64_t e) const {
        return coset_major ? (((e & ((1ULL << log_bloc) - 1ULL)) << logn) | (e >> log_bloc)) : e;
    }
};
