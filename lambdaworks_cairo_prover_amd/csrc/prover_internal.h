// Helpers shared by the translation units of the prover (prover.cpp, prover_upload.cpp, prover_driver.cpp).
#pragma once
#include "prover.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace sp {

static inline double wall_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static inline bool timing_enabled() { static int v = -1; if (v < 0) v = std::getenv("SP_TIMING") ? 1 : 0; return v == 1; }
// SP_TIMING=1: wall time since the previous point on stderr (synchronises the context stream); needs `sp_ctx* ctx` and `double _tp`
#define SP_TIMEPOINT(label)                                                                 \
    do { if (timing_enabled()) { (void)hipStreamSynchronize(ctx->stream); double _t = wall_ms(); \
         std::fprintf(stderr, "[sp_timing] %-28s %9.2f ms\n", label, _t - _tp); _tp = _t; } } while (0)

}  // namespace sp
