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

// Waits for a stream.  The proof's Fiat-Shamir round trips (root down, challenge up) are latency and a blocking
// hipStreamSynchronize costs ~20 us of wake-up each, so the first 100 us are spent polling; a wait that lasts longer (the LDE and
// hashing of a trace segment, a collective of another rank) is not latency-critical any more and goes to sleep in the runtime
// instead of burning a core - eight ranks polling through every collective took 2m21 of system time in a 37 s test.
// With more ranks on the host than CPUs for them (sp::host_oversubscribed(): SP_OPT_HOST_RANKS against the cgroup's CPUs) nothing
// polls: a spinning rank would only take the core another rank's launch thread needs.
#include <chrono>
namespace sp { bool host_oversubscribed(); }
static inline hipError_t sp_stream_wait_polling(hipStream_t st) {
    if (sp::host_oversubscribed()) return hipStreamSynchronize(st);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
        if ((spins & 15u) == 15u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(100)) return hipStreamSynchronize(st);
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
