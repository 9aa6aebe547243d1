// Context object behind the C ABI (one per proof / per caller thread).
#pragma once
#include <atomic>
#include <vector>
#include "common.h"
#include "ntt.h"
#include "merkle.h"
#include "field_kernels.h"

struct sp_deletable { virtual ~sp_deletable() {} };

struct sp_ctx {
    int device = 0;
    int enc = SP_FE_CANON_BE;
    hipStream_t stream = nullptr;
    sp::NttEngine* ntt = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;    // around the last sp_*_dev call
    hipEvent_t tev0 = nullptr, tev1 = nullptr;  // sp_timer_start / sp_timer_stop
    bool last_pending = false;
    int* d_flag = nullptr;
    // 64 KB set aside at creation for the agreement words of sp_comm_measure (32 B per rank, own word first): a rank whose LOCAL
    // preparation failed - out of memory for the payload, say - can still tell its peers so instead of leaving them in a collective
    void* d_agree = nullptr;
    static constexpr uint64_t kAgreeBytes = 64 << 10, kAgreeMaxWorld = kAgreeBytes / 32 - 1;
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    float last_ms = 0.f;
    float round_ms[5] = {0, 0, 0, 0, 0};
    uint32_t proof_info[4] = {0, 0, 0, 0};   // sp_last_proof_info
    uint64_t prover_device_bytes = 0;         // sp_prover_device_bytes
    double upload_stats[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // sp_last_upload_stats
    sp_deletable* prover_state_deleter_holder = nullptr;  // round-level prover state (prover.cpp)
    // coset sharding across GPUs: world size (power of two), rank and the blocking all-gather hook (see sp_set_collective)
    int world = 1, rank = 0;
    sp_allgather_fn allgather = nullptr;
    sp_alltoall_fn alltoall = nullptr;    // optional (sp_set_alltoall); same user pointer
    // stream-ordered form of the all-gather (sp_comm_init_rccl, sp_set_collective_async): enqueued on the given stream, returns at
    // once - lets the prover run an exchange beside its transforms instead of in front of them
    sp_allgather_async_fn allgather_async = nullptr;
    sp_alltoall_async_fn alltoall_async = nullptr;
    void* allgather_user = nullptr;
    bool comm_is_null = false;            // the timing-only transport (sp_comm_init_null): the other ranks do not exist and deliver zeros
    uint64_t stat_ag_calls = 0, stat_ag_bytes = 0, stat_a2a_calls = 0, stat_a2a_bytes = 0, stat_recv_bytes = 0;
    // sp_comm_time_ms: how long the collectives took.  Stream-ordered ones are bracketed by two events on the stream they are enqueued on
    // (what the exchange occupied that stream for, the wait for the slowest peer included), read back when the figure is asked for;
    // blocking ones by the wall clock around the hook.
    std::vector<hipEvent_t> comm_ev;          // pairs (begin, end); [0, comm_ev_used) are recorded and not yet read back
    size_t comm_ev_used = 0;
    double stat_comm_stream_ms = 0.0, stat_comm_blocking_ms = 0.0;
    hipEvent_t comm_event() {                  // nullptr beyond 8192 events between two read-backs: that exchange is not timed
        if (comm_ev_used == comm_ev.size()) {
            if (comm_ev.size() >= 8192) return nullptr;
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            comm_ev.push_back(e);
        }
        return comm_ev[comm_ev_used++];
    }
    uint32_t opt_fri_shard_min_log = 16;  // sp_set_option
    int opt_shard_interpolation = 2;          // 0 replicated, 1 by column + coefficient all-gather, 2 whichever the link model makes faster
    double opt_link_gbs = 46.0;               // what one xGMI link delivers per direction (76.8 GB/s x 0.6): the model behind mode 2 ...
    bool opt_link_gbs_explicit = false;       // ... when the caller stated it (SP_OPT_LINK_GBS); otherwise a measurement, when there is one, wins
    // sp_comm_measure: {all-gather ms, GB/s per link and direction of that all-gather, all-to-all ms, GB/s per link of that all-to-all,
    // bytes per rank, world}; the rates are the MINIMUM over the ranks, so every rank takes the same decision from them.  0 = not measured.
    double measured_link[6] = {0, 0, 0, 0, 0, 0};
    // A MEASURED rate decides with a margin (SP_LINK_MEASURED_MARGIN): the threshold of mode 2 sits where realistic xGMI all-gather rates
    // are (45 - 60 GB/s per link at G = 8, n = 2^20), and a figure within the run-to-run spread of it must not flip the mode from one
    // communicator to the next - by-column interpolation has to beat the threshold by the margin; a STATED rate is taken at its word.
    double link_gbs_for_model() const { return opt_link_gbs_explicit ? opt_link_gbs : (measured_link[1] > 0 ? measured_link[1] / SP_LINK_MEASURED_MARGIN : opt_link_gbs); }
    uint32_t opt_upload_threads = 24;
    int opt_merkle_backend = SP_MERKLE_KECCAK256;
    bool opt_device_trace = true;             // sp_cairo_prove_run builds the main trace on the device from the run's registers and memory
    std::atomic<int> prewarm_cancel{0};       // sp_prewarm_cancel: a running (or coming) sp_prewarm cuts its clock ramp short
    bool opt_merkle_one_column_rows = false;   // sp_merkle_build* with fe_per_leaf == 1 under Poseidon: row tree instead of the FRI-layer tree
    sp_deletable* comm_holder = nullptr;  // RCCL communicator when sp_comm_init_rccl is used
};
