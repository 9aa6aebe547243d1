// Round-by-round STARK prover on the device (see prover.h). Host code only: it sequences kernels on the context
// stream, keeps every polynomial / evaluation / tree resident in HBM and moves only roots, challenges and openings.
#include "prover_internal.h"
#include "cairo_host.h"
#include <array>
#include "keccak.h"
#include <algorithm>
#include <unordered_map>
#include <cstring>
#include <cmath>
#include <stdexcept>
#include <functional>
#include <memory>

namespace sp {

void host_pool_delete(HostPool* p);

StarkProver::~StarkProver() {
    free_all();
    if (arena_) (void)hipFree(arena_);
    arena_ = nullptr; arena_cap_ = 0;
    c_->prover_device_bytes = 0;
    for (auto& e : ev_dma_) if (e) (void)hipEventDestroy(e);
    for (auto& u : up_ev_) for (hipEvent_t e : {u.dma0, u.dma1, u.ready, u.done}) if (e) (void)hipEventDestroy(e);
    if (up_start_) (void)hipEventDestroy(up_start_);
    if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
    if (pool_) host_pool_delete(pool_);
    for (auto& p : h_stage_) { if (p) (void)hipHostFree(p); p = nullptr; }
    if (h_pin_) (void)hipHostFree(h_pin_);
    if (h_open_pin_) (void)hipHostFree(h_open_pin_);
    if (h_wide_) (void)hipHostFree(h_wide_);
    for (hipEvent_t e : {ev_side_fork_, ev_side_deep_, ev_side_bnd_, ev_side_aux_, ev_side_presort_}) if (e) (void)hipEventDestroy(e);
    if (side_stream_) (void)hipStreamDestroy(side_stream_);
    if (ev_comm_fork_) (void)hipEventDestroy(ev_comm_fork_);
    for (auto& e : ev_comm_done_) if (e) (void)hipEventDestroy(e);
    if (comm_stream_) (void)hipStreamDestroy(comm_stream_);
}

int StarkProver::ensure_side() {
    if (!side_stream_) SP_HIP_CHECK(hipStreamCreateWithFlags(&side_stream_, hipStreamNonBlocking));
    for (hipEvent_t* e : {&ev_side_fork_, &ev_side_deep_, &ev_side_bnd_, &ev_side_aux_, &ev_side_presort_})
        if (!*e) SP_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    return SP_OK;
}

int StarkProver::wait_stream() {
    SP_HIP_CHECK(sp_stream_wait_polling(c_->stream));
    return SP_OK;
}

int StarkProver::readback(void* dst_host, const void* src_dev, size_t bytes) {
    if (bytes > 4096) { sp_set_error("readback: more than the 4 KB pinned slot"); return SP_E_INVALID_ARG; }
    if (!h_pin_ && hipHostMalloc(&h_pin_, 4096, hipHostMallocDefault) != hipSuccess) { h_pin_ = nullptr; sp_set_error("pinned read-back slot: allocation failed"); return SP_E_ALLOC; }
    SP_HIP_CHECK(hipMemcpyAsync(h_pin_, src_dev, bytes, hipMemcpyDeviceToHost, c_->stream));
    SP_TRY(wait_stream());
    memcpy(dst_host, h_pin_, bytes);
    return SP_OK;
}

void StarkProver::free_all() {
    (void)hipSetDevice(c_->device);
    (void)hipStreamSynchronize(c_->stream);
    if (copy_stream_) (void)hipStreamSynchronize(copy_stream_);
    if (side_stream_) (void)hipStreamSynchronize(side_stream_);
    if (comm_stream_) (void)hipStreamSynchronize(comm_stream_);
    d_bpre_ = nullptr; bpre_cap_ = 0; bpre_valid_ = false; deep_pref_ = false; d_flag_side_ = nullptr;
    d_fri_chain_ = nullptr; fri_chain_layers_ = 0; d_comp_consts_chk_ = nullptr; check_pending_ = false; presorted_ = false; presort_pub_ = nullptr;
    d_flagbits_ = nullptr; flagbits_words_ = 0;     // carved from the arena / allocs_ like the rest: gone with the shape
    // (the page-locked upload ring does not depend on the shape: it stays until the prover goes)
    for (void* p : allocs_) (void)hipFree(p);
    allocs_.clear();
    alloc_bytes_ = 0;
    arena_off_ = 0;      // the arena itself stays: the next shape is carved out of it
    publish_device_bytes();
}

int StarkProver::alloc(void** p, size_t bytes) {
    const uint64_t aligned = ((uint64_t)(bytes ? bytes : 1) + 255) & ~(uint64_t)255;
    if (measuring_) { *p = reinterpret_cast<void*>(uintptr_t(256)); measured_ += aligned; return SP_OK; }   // (sizing pass of setup_impl)
    if (arena_ && arena_off_ + aligned <= arena_cap_) {
        *p = arena_ + arena_off_;
        arena_off_ += aligned;
        return SP_OK;
    }
    *p = nullptr;
    if (hipMalloc(p, bytes ? bytes : 1) != hipSuccess) {
        (void)hipGetLastError();
        sp_set_error("hipMalloc failed (" + std::to_string(bytes) + " bytes)");
        return SP_E_ALLOC;
    }
    allocs_.push_back(*p);
    alloc_bytes_ += bytes;
    publish_device_bytes();
    return SP_OK;
}

int StarkProver::setup(uint64_t n, uint32_t main_cols, uint32_t aux_cols, bool has_rc, const ProofOptionsHost& opt) {
    const int rc = setup_impl(n, main_cols, aux_cols, has_rc, opt);
    if (rc != SP_OK) {   // a failed (re)shaping leaves nothing behind: the next setup() of the same shape starts from scratch
        free_all();
        n_ = 0; ready_ = false; stage_ = 0;
    }
    return rc;
}

int StarkProver::setup_impl(uint64_t n, uint32_t main_cols, uint32_t aux_cols, bool has_rc, const ProofOptionsHost& opt) {
    offsets_ = {0, 1};
    int k = sp_log2_exact(n), lb = sp_log2_exact(opt.blowup_factor);
    if (k < 1 || lb < 1 || k + lb > 30 || (1u << lb) > CAIRO_MAX_BLOWUP) { sp_set_error("setup: trace length and blowup factor must be powers of two (blowup 2 .. 128, at most 2^30 LDE points)"); return SP_E_INVALID_ARG; }
    if (main_cols + aux_cols > 64) return SP_E_INVALID_ARG;
    SP_HIP_CHECK(hipSetDevice(c_->device));
    if (c_->world < 1 || (c_->world & (c_->world - 1)) || c_->rank < 0 || c_->rank >= c_->world) {
        sp_set_error("setup: world size must be a power of two");
        return SP_E_INVALID_ARG;
    }
    if (c_->world > 1 && !c_->allgather) { sp_set_error("setup: world > 1 needs sp_set_collective / sp_comm_init_rccl"); return SP_E_STATE; }
    if (ready_ && (arena_ || !allocs_.empty()) && n == n_ && main_cols == Cm_ && aux_cols == Ca_ && has_rc == has_rc_ && opt.blowup_factor == opt_.blowup_factor &&
        opt.coset_offset == opt_.coset_offset && (uint32_t)c_->world == world_ && (uint32_t)c_->rank == wrank_ && c_->opt_shard_interpolation == shard_mode_) {
        // same shape as the previous proof on this context: keep every device buffer and table
        opt_ = opt; stage_ = 1; fri_layer_ = 0;
        bpre_valid_ = false; deep_pref_ = false; check_pending_ = false; presorted_ = false; presort_pub_ = nullptr;
        return SP_OK;
    }
    free_all();
    ready_ = false; stage_ = 0;
    d_auxws_ = nullptr; auxws_bytes_ = 0; auxws_pm_cap_ = 0; d_hfull_ = nullptr; d_hnat_ = nullptr; h_full_ = false;
    d_air_prog_ = nullptr; d_ex_roots_ = nullptr; ex_roots_cap_ = 0;
    d_flags_all_ = nullptr;
    d_gather_ = nullptr; gather_cap_ = 0; d_fullN_ = nullptr; d_small_ = nullptr; d_deepx_ = nullptr; deepx_cap_ = 0; d_cstage_ = nullptr; d_local_ = nullptr; d_recv_ = nullptr; d_roots_ = nullptr;
    opt_ = opt; n_ = n; logn_ = (uint32_t)k; logb_ = (uint32_t)lb; logN_ = logn_ + logb_; N_ = n << lb;
    Cm_ = main_cols; Ca_ = aux_cols; C_ = main_cols + aux_cols; has_rc_ = has_rc;
    world_ = (uint32_t)c_->world; wrank_ = (uint32_t)c_->rank; shard_mode_ = c_->opt_shard_interpolation;
    // one or more LDE cosets per group; with more ranks than cosets the surplus ranks replicate a role (moving half a coset's
    // LDE over one xGMI link costs more than computing it)
    G_ = std::min<uint32_t>(world_, 1u << lb); logG_ = (uint32_t)sp_log2_exact(G_); rank_ = wrank_ & (G_ - 1);
    Nl_ = N_ >> logG_;
    // Interpolation by column with an all-gather of the coefficients, or on every rank?  A rank saves
    // (1 - 1/G) of the size-n inverse transforms (n log n / 2 butterflies per column at ~1.35e11 / s) and receives (1 - 1/G) of the
    // coefficients (32 n bytes per column over G - 1 links): sharding pays when  64 x 1.35e11 < (G - 1) x link bytes/s x log2 n.
    // (sp_model_shard_interpolation.)  On 46 GB/s per link that needs (G - 1) log2 n > 188 - no shape this prover sees - so mode 2
    // interpolates everywhere unless the fabric is faster: the rate sp_comm_measure found (sp_comm_init_rccl runs it once per
    // communicator) or the one the caller states (SP_OPT_LINK_GBS wins); an exchange that overlaps the transforms completely
    // (stream-ordered transport) is worth at most the inverse transforms it replaces, 3 - 5 ms at 2^20 rows.
    shard_interp_ = false;
    if (G_ > 1) {
        if (c_->opt_shard_interpolation == 1) shard_interp_ = true;
        else if (c_->opt_shard_interpolation == 2) shard_interp_ = sp_model_shard_interpolation(c_->link_gbs_for_model(), G_, (uint32_t)k) == 1;
        // SP_COMM_LOG: the mode and the rate it was chosen from, once per set-up shape and rank (what a first multi-GPU run is read by)
        static const bool comm_log = std::getenv("SP_COMM_LOG") != nullptr;
        if (comm_log)
            std::fprintf(stderr, "[stark252 rank %u/%u] 2^%d rows, %u groups: interpolation %s (SP_OPT_SHARD_INTERPOLATION = %d; link %.1f GB/s per direction - %s; by column pays above %.1f)\n",
                         wrank_, world_, k, G_, shard_interp_ ? "by column + coefficient all-gather" : "on every rank", c_->opt_shard_interpolation, c_->link_gbs_for_model(),
                         c_->opt_link_gbs_explicit ? "stated" : (c_->measured_link[1] > 0 ? "measured all-gather rate / 1.25" : "assumed"),
                         64.0 * 1.35e11 / ((double)(G_ - 1) * (double)k) / 1e9);
    }
    if (G_ > 1 && N_ < 2ull * G_ * G_) { sp_set_error("setup: the LDE domain is too small for this many ranks"); return SP_E_INVALID_ARG; }
    double _tp = wall_ms();
    sp_ctx* ctx = c_;
    h_ = fe_from_u64(opt.coset_offset);
    if (fe_is_zero(h_)) return SP_E_INVALID_ARG;
    hinv_ = fe_inv(h_);
    half_ = fe_inv(fe_from_u64(2)); binv_ = fe_inv(fe_from_u64(1ull << lb));
    g_ = host_primitive_root((int)logn_);
    // Every buffer whose size setup() knows, in one pass that runs twice: first to size the arena, then to carve it.
    auto allocate_all = [&]() -> int {
        SP_TRY(alloc((void**)&d_coeffs_, sizeof(fe) * n_ * C_));
        SP_TRY(alloc((void**)&d_trace_, sizeof(fe) * n_ * C_));
        SP_TRY(alloc((void**)&d_lde_, sizeof(fe) * std::max<uint64_t>(Nl_, n_) * C_));  // >= n per column: also stages the raw rows
        SP_TRY(alloc((void**)&d_t1_, sizeof(fe) * n_));
        SP_TRY(alloc((void**)&d_t2_, sizeof(fe) * n_));
        SP_TRY(alloc((void**)&d_h12s_, sizeof(fe) * n_ * 2));
        SP_TRY(alloc((void**)&d_h12_, sizeof(fe) * Nl_ * 2));
        SP_TRY(alloc((void**)&d_scratch_, sizeof(fe) * scratch_elems()));
        if (G_ > 1) {
            SP_TRY(alloc((void**)&d_local_, sizeof(fe) * Nl_));
            SP_TRY(alloc((void**)&d_recv_, sizeof(fe) * Nl_));
            SP_TRY(alloc((void**)&d_roots_, sizeof(digest32) * world_));
            if (shard_interp_) {
                cpr_max_ = (std::max(Cm_, Ca_) + G_ - 1) / G_;
                SP_TRY(alloc((void**)&d_cstage_, sizeof(fe) * (uint64_t)world_ * cpr_max_ * n_));
            }
        }
        SP_TRY(alloc_tree(tree_main_, N_, G_ > 1));
        SP_TRY(alloc_tree(tree_aux_, N_, G_ > 1));
        SP_TRY(alloc_tree(tree_comp_, N_, G_ > 1));
        SP_TRY(alloc((void**)&d_comp_consts_, sizeof(CompositionConsts)));
        SP_TRY(alloc((void**)&d_deep_consts_, sizeof(DeepConsts)));
        SP_TRY(alloc((void**)&d_nonce_, sizeof(unsigned long long)));
        // FRI: layers of at least 2^opt_fri_shard_min_log leaves (and at least 2 G^2, so that every rank owns whole blocks of the
        // digest exchange) stay sharded; from layer fri_rep_ on every rank holds the whole layer.  The last, uncommitted fold
        // output (layer log n) is always replicated.
        fri_rep_ = 0;
        if (G_ > 1)
            while (fri_rep_ < logn_ && (N_ >> fri_rep_) >= std::max<uint64_t>(1ull << c_->opt_fri_shard_min_log, 2ull * G_ * G_)) ++fri_rep_;
        d_fri_evals_.clear(); fri_trees_.clear();
        for (uint32_t l = 0; l <= logn_; ++l) {
            fe* e = nullptr;
            const uint64_t M = N_ >> l;
            SP_TRY(alloc((void**)&e, sizeof(fe) * (fri_sharded(l) ? M >> logG_ : M)));
            d_fri_evals_.push_back(e);
            if (l < logn_) { TreeBuf t; SP_TRY(alloc_tree(t, M, fri_sharded(l))); fri_trees_.push_back(t); }
        }
        SP_TRY(alloc((void**)&d_post_comp_, sizeof(fe) * 2 * n_));
        SP_TRY(alloc((void**)&d_post_deep_, sizeof(fe) * n_));
        d_post_comp0_ = nullptr;
        if (G_ > 1 && logG_ == logb_) SP_TRY(alloc((void**)&d_post_comp0_, sizeof(fe) * 2 * n_));
        return SP_OK;
    };
    measuring_ = true; measured_ = 0;
    const int rc_measure = allocate_all();
    measuring_ = false;
    SP_TRY(rc_measure);
    {
        // room for what a Cairo proof allocates on first use (auxiliary-trace workspace, side-stream inverses): those allocations find
        // their place in the arena too instead of costing a hipMalloc each
        size_t sort_tmp = 0;
        const uint64_t lazy = (Ca_ == 18 ? aux_workspace_bytes(n_, 4096, &sort_tmp) : 0) + sizeof(fe) * 19 * n_ + (4u << 20);
        const uint64_t need = measured_ + lazy;
        if (arena_cap_ < need) {
            if (arena_) (void)hipFree(arena_);
            arena_ = nullptr; arena_cap_ = 0;
            void* a = nullptr;
            if (hipMalloc(&a, need) == hipSuccess) { arena_ = static_cast<uint8_t*>(a); arena_cap_ = need; }
            else (void)hipGetLastError();     // no single block of that size: the buffers are allocated one by one
        }
        arena_off_ = 0;
        publish_device_bytes();
    }
    SP_TRY(allocate_all());
    d_memcols_ = d_trace_ + 19 * n_;  // pc .. off_op1 columns of the main trace (input of the Cairo auxiliary trace)
    SP_TIMEPOINT("  setup: device allocations");
    // T1[q] = n^-1 h^rev(q): turns the unscaled DIF output into h-scaled coefficients c_k h^k (bit-reversed order)
    fe ninv = fe_inv(fe_from_u64(n_));
    SP_TRY(gen_power_table(c_->stream, d_t1_, n_, logn_, h_, ninv));
    // T2[q] = N^-1 h^-rev(q): composition-polynomial split
    fe Ninv = fe_inv(fe_from_u64(N_));
    SP_TRY(gen_power_table(c_->stream, d_t2_, n_, logn_, hinv_, Ninv));
    const fe* roots = nullptr;
    SP_TRY(c_->ntt->roots((int)logN_, &roots));
    // post factors of the 2n-point composition split and of the one-coset DEEP interpolation: functions of the shape and of
    // this rank's first coset only, so they are generated once per setup instead of once per proof
    {
        const fe wN = host_primitive_root((int)logN_);
        const fe u = fe_inv(fe_pow_u64(wN, rank_));  // w_N^-c0
        const fe minv = fe_inv(fe_from_u64(2 * n_));
        const fe base = fe_mul(hinv_, fe_sqr(u));
        SP_TRY(gen_power_table(c_->stream, d_post_comp_, n_, logn_, base, minv));
        SP_TRY(gen_power_table(c_->stream, d_post_comp_ + n_, n_, logn_, base, fe_mul(minv, fe_mul(hinv_, u))));
        SP_TRY(gen_power_table(c_->stream, d_post_deep_, n_, logn_, u, fe_inv(fe_from_u64(n_))));
        if (d_post_comp0_) {   // one coset per rank: the composition pair (0, b/2) is interpolated with c0 = 0 everywhere
            SP_TRY(gen_power_table(c_->stream, d_post_comp0_, n_, logn_, hinv_, minv));
            SP_TRY(gen_power_table(c_->stream, d_post_comp0_ + n_, n_, logn_, hinv_, fe_mul(minv, hinv_)));
        }
    }
    SP_TIMEPOINT("  setup: tables");
    ready_ = true;
    stage_ = 1;
    return SP_OK;
}

// sp_prewarm, first half: everything a first proof would otherwise create on its critical path that is not device memory of the
// shape - the page-locked read-back slots (hipHostMalloc costs ~1 ms a piece), the side stream and its events, the copy stream
// and the upload timers, and for callers of the row-major entry points the page-locked ring and the parked gather threads.
int StarkProver::warm_plumbing(bool host_rows) {
    SP_HIP_CHECK(hipSetDevice(c_->device));
    if (!h_pin_ && hipHostMalloc(&h_pin_, 4096, hipHostMallocDefault) != hipSuccess) { h_pin_ = nullptr; return SP_E_ALLOC; }
    if (!h_wide_ && hipHostMalloc(reinterpret_cast<void**>(&h_wide_), 64, hipHostMallocDefault) != hipSuccess) { h_wide_ = nullptr; return SP_E_ALLOC; }
    SP_TRY(ensure_side());
    SP_TRY(ensure_upload((uint32_t)UPLOAD_MAX_GROUPS));
    if (host_rows) SP_TRY(ensure_ring_and_pool());
    return SP_OK;
}
// sp_prewarm, second half: round 1's kernel sequence at the REAL shape on whatever the arena holds (the transforms have no
// data-dependent control flow and accept any 256-bit operand; the hash kernels convert and absorb whatever they read) - the
// size-specific kernel variants take their first launch here, and the device reaches its clocks before the trace exists.
int StarkProver::warm_round1() {
    if (!ready_ || stage_ != 1) return SP_E_STATE;
    SP_HIP_CHECK(hipSetDevice(c_->device));
    SP_HIP_CHECK(hipMemsetAsync(d_trace_, 0, sizeof(fe) * n_ * C_, c_->stream));
    // Column slice by column slice, with a look at sp_prewarm_cancel's flag between slices: a caller whose trace is ready does not wait
    // for the rest of the ramp.  The whole of it costs 15 ms at config #4's shape and 50 ms at config #3's and makes the first proof
    // 1 - 2 ms faster than a fifth of it does (tools/experiments/ab_prewarm_r1.sh; SP_PREWARM_R1_FRAC bounds it for experiments).
    static const double frac = [] { const char* e = std::getenv("SP_PREWARM_R1_FRAC"); return e ? std::min(1.0, std::max(0.0, std::atof(e))) : 1.0; }();
    auto cancelled = [this] { return c_->prewarm_cancel.load(std::memory_order_acquire) != 0; };
    bool stop = false;
    for (int seg = 0; seg < 2 && !stop; ++seg) {
        const uint32_t col0 = seg ? Cm_ : 0, cols = seg ? Ca_ : Cm_;
        if (!cols) continue;
        const uint32_t tc = std::max<uint32_t>(1, (uint32_t)(cols * frac)), slice = std::max<uint32_t>(1, cols / 8);
        for (uint32_t c0 = 0; c0 < tc && !stop; c0 += slice) {
            const uint32_t w = std::min(slice, tc - c0);
            SP_TRY(c_->ntt->dif_natural_to_bitrev_inverse(d_coeffs_ + (uint64_t)(col0 + c0) * n_, (int)logn_, w, n_, d_t1_, d_trace_ + (uint64_t)(col0 + c0) * n_));
            SP_TRY(c_->ntt->lde_coset_major(d_coeffs_ + (uint64_t)(col0 + c0) * n_, d_lde_ + (uint64_t)(col0 + c0) * Nl_, (int)logn_, (int)logb_, w, n_, Nl_, (int)logG_, (int)rank_));
            SP_TRY(wait_stream());
            stop = cancelled();
        }
        if (stop || tc < cols) break;
        TreeBuf& t = seg ? tree_aux_ : tree_main_;
        const MerkleHash mh = merkle_hash(false);
        if (t.top == t.sub) {
            SP_TRY(merkle_hash_leaves(c_->stream, d_lde_ + (uint64_t)col0 * Nl_, Nl_, cols, Nl_, t.sub, lde_order(), mh));
            SP_TRY(merkle_reduce(c_->stream, t.sub, Nl_, nullptr, mh));
        } else {
            SP_TRY(merkle_hash_leaves_flat(c_->stream, d_lde_ + (uint64_t)col0 * Nl_, Nl_, cols, Nl_, reinterpret_cast<digest32*>(d_local_), lde_order(), mh));
            SP_TRY(merkle_reduce(c_->stream, t.sub, t.sub_leaves, nullptr, mh));
        }
        SP_TRY(wait_stream());
        stop = cancelled();
    }
    // the composition columns' shape too: two columns, the 2n-point inverse transform
    SP_TRY(c_->ntt->dif_natural_to_bitrev_inverse(d_h12s_, (int)logn_ + 1, 1, 2 * n_, d_post_comp_));
    SP_TRY(c_->ntt->lde_coset_major(d_h12s_, d_h12_, (int)logn_, (int)logb_, 2, n_, Nl_, (int)logG_, (int)rank_));
    SP_TRY(wait_stream());
    return SP_OK;
}

int StarkProver::alloc_tree(TreeBuf& t, uint64_t leaves_total, bool sharded) {
    t.sub_leaves = sharded ? leaves_total >> logG_ : leaves_total;
    SP_TRY(alloc((void**)&t.sub, sizeof(digest32) * (2 * t.sub_leaves - 1)));
    t.top = t.sub;
    if (sharded) SP_TRY(alloc((void**)&t.top, sizeof(digest32) * (2ull * G_ - 1)));
    return SP_OK;
}

// Frees a buffer this prover outgrew (everything that could still read it has finished first).
void StarkProver::release(void* p, size_t bytes) {
    if (!p) return;
    (void)hipStreamSynchronize(c_->stream);
    if (side_stream_) (void)hipStreamSynchronize(side_stream_);
    if (copy_stream_) (void)hipStreamSynchronize(copy_stream_);
    auto it = std::find(allocs_.begin(), allocs_.end(), p);
    if (it == allocs_.end()) return;     // carved out of the arena: the space comes back with the next setup()
    allocs_.erase(it);
    (void)hipFree(p);
    alloc_bytes_ -= std::min<uint64_t>(alloc_bytes_, bytes);
    publish_device_bytes();
}

int StarkProver::ensure_gather(uint64_t elems) {
    if (elems <= gather_cap_) return SP_OK;
    release(d_gather_, sizeof(fe) * gather_cap_);
    d_gather_ = nullptr; gather_cap_ = 0;
    SP_TRY(alloc((void**)&d_gather_, sizeof(fe) * elems));
    gather_cap_ = elems;
    return SP_OK;
}

// DEEP inverses beyond the shared scratch (many frame rows on a small blowup): one buffer, grown on demand, kept across proofs
// DEEP denominators of round 4 on the side stream while round 3 evaluates the polynomials at z (same arrays, same batch
// inversion as deep_fri_begin's inline path; valid-trace form only: one coset of n points).
int StarkProver::prefetch_deep_inverses() {
    deep_pref_ = false;
    if (h_full_) return SP_OK;
    const uint32_t R = (uint32_t)offsets_.size(), npts = R + 1;
    SP_TRY(ensure_side());
    SP_TRY(ensure_deep_scratch((2ull * npts + 1) * n_));
    if (!d_flag_side_) SP_TRY(alloc((void**)&d_flag_side_, 4 * sizeof(int)));
    fe pts[AIR_MAX_OFFSETS + 1];
    for (uint32_t k = 0; k < R; ++k) pts[k] = fe_mul(z_, fe_pow_u64(g_, offsets_[k]));
    pts[R] = fe_sqr(z_);
    const fe* roots_n = nullptr;
    SP_TRY(c_->ntt->roots((int)logn_, &roots_n));
    const fe hp = fe_mul(h_, fe_pow_u64(host_primitive_root((int)logN_), rank_));
    SP_HIP_CHECK(hipEventRecord(ev_side_fork_, c_->stream));          // (the buffer's previous readers are behind this point)
    SP_HIP_CHECK(hipStreamWaitEvent(side_stream_, ev_side_fork_, 0));
    SP_HIP_CHECK(hipMemsetAsync(d_flag_side_, 0, sizeof(int), side_stream_));
    SP_TRY(coset_minus_points(side_stream_, d_deepx_, n_, logn_, roots_n, hp, pts, npts, ShardMap{0, 0, 0}));
    SP_TRY(batch_inverse(side_stream_, d_deepx_, d_deepx_ + (uint64_t)npts * n_, (uint64_t)npts * n_, d_flag_side_));
    SP_HIP_CHECK(hipEventRecord(ev_side_deep_, side_stream_));
    deep_pref_ = true;
    return SP_OK;
}

// Boundary denominators of round 2 for a constraint-satisfying trace (the 2n points of this rank's coset pair), computed on
// the side stream while round 1 runs.  composition_core uses them when it takes that path with the same boundary points.
int StarkProver::prefetch_boundary_inverses(const std::vector<uint64_t>& steps_in) {
    bpre_valid_ = false;
    if (!ready_ || logb_ < logG_ + 1) return SP_OK;
    std::vector<uint64_t> steps;
    for (uint64_t s : steps_in) if (std::find(steps.begin(), steps.end(), s) == steps.end()) steps.push_back(s);
    if (steps.empty() || steps.size() > 3) return SP_OK;
    SP_HIP_CHECK(hipSetDevice(c_->device));
    SP_TRY(ensure_side());
    const uint64_t M = 2 * n_;
    if (bpre_cap_ < 6 * M) { SP_TRY(alloc((void**)&d_bpre_, sizeof(fe) * 6 * M)); bpre_cap_ = 6 * M; }
    if (!d_flag_side_) SP_TRY(alloc((void**)&d_flag_side_, 4 * sizeof(int)));
    bpre_points_.clear();
    for (uint64_t s : steps) bpre_points_.push_back(fe_pow_u64(g_, s));
    const fe* roots_m = nullptr;
    SP_TRY(c_->ntt->roots((int)logn_ + 1, &roots_m));
    const fe hp = fe_mul(h_, fe_pow_u64(host_primitive_root((int)logN_), rank_));
    const uint32_t nd = (uint32_t)bpre_points_.size();
    SP_HIP_CHECK(hipEventRecord(ev_side_fork_, c_->stream));
    SP_HIP_CHECK(hipStreamWaitEvent(side_stream_, ev_side_fork_, 0));
    SP_HIP_CHECK(hipMemsetAsync(d_flag_side_ + 1, 0, sizeof(int), side_stream_));
    SP_TRY(coset_minus_points(side_stream_, d_bpre_, M, logn_ + 1, roots_m, hp, bpre_points_.data(), nd, ShardMap{0, 0, 0}));
    SP_TRY(batch_inverse(side_stream_, d_bpre_, d_bpre_ + 3 * M, (uint64_t)nd * M, d_flag_side_ + 1));
    SP_HIP_CHECK(hipEventRecord(ev_side_bnd_, side_stream_));
    bpre_valid_ = true;
    return SP_OK;
}

int StarkProver::ensure_deep_scratch(uint64_t elems) {
    if (elems <= deepx_cap_) return SP_OK;
    release(d_deepx_, sizeof(fe) * deepx_cap_);
    d_deepx_ = nullptr; deepx_cap_ = 0; deep_pref_ = false;
    SP_TRY(alloc((void**)&d_deepx_, sizeof(fe) * elems));
    deepx_cap_ = elems;
    return SP_OK;
}

// [N] scratch for the paths that need the whole domain on every rank (constraint-violating traces, G = b)
int StarkProver::full_domain_buffer(fe** out) {
    if (!fri_sharded(0)) { *out = d_fri_evals_[0]; return SP_OK; }   // free until round 4
    if (!d_fullN_) SP_TRY(alloc((void**)&d_fullN_, sizeof(fe) * N_));
    *out = d_fullN_;
    return SP_OK;
}

// Blocking all-gather through the context hook: every rank contributes bytes_per_rank, recv = [world][bytes_per_rank]
// (the first G slots are the G distinct roles).
namespace {
// brackets one stream-ordered exchange with two events on the stream it is enqueued on (sp_comm_time_ms reads them back)
struct CommSpan {
    sp_ctx* c; hipStream_t st; bool on = false;
    CommSpan(sp_ctx* ctx, hipStream_t stream) : c(ctx), st(stream) {
        if (c->comm_ev_used + 2 > 8192) return;
        hipEvent_t e = c->comm_event();
        if (e && hipEventRecord(e, st) == hipSuccess) on = true; else if (e) --c->comm_ev_used;
    }
    ~CommSpan() {
        if (!on) return;
        hipEvent_t e = c->comm_event();
        if (!e || hipEventRecord(e, st) != hipSuccess) c->comm_ev_used -= e ? 2 : 1;      // (an unpaired begin is dropped)
    }
};
struct BlockingSpan {
    sp_ctx* c; double t0;
    explicit BlockingSpan(sp_ctx* ctx) : c(ctx), t0(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count()) {}
    ~BlockingSpan() { c->stat_comm_blocking_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0; }
};
}  // namespace

int StarkProver::all_gather(const void* send_dev, void* recv_dev, uint64_t bytes_per_rank, bool stream_ordered) {
    if (stream_ordered && comm_async()) {
        int rc;
        { CommSpan span(c_, c_->stream); rc = c_->allgather_async(c_->allgather_user, send_dev, recv_dev, bytes_per_rank, c_->stream); }
        if (rc != 0) { sp_set_error("stream-ordered all-gather failed (" + std::to_string(rc) + ")"); return SP_E_HIP; }
        c_->stat_ag_calls += 1; c_->stat_ag_bytes += bytes_per_rank; c_->stat_recv_bytes += bytes_per_rank * (world_ - 1);
        return SP_OK;
    }
    SP_HIP_CHECK(sp_stream_wait_polling(c_->stream));
    int rc;
    { BlockingSpan span(c_); rc = c_->allgather(c_->allgather_user, send_dev, recv_dev, bytes_per_rank); }
    if (rc != 0) { sp_set_error("all-gather hook failed (" + std::to_string(rc) + ")"); return SP_E_HIP; }
    c_->stat_ag_calls += 1; c_->stat_ag_bytes += bytes_per_rank; c_->stat_recv_bytes += bytes_per_rank * (world_ - 1);
    return SP_OK;
}

int StarkProver::all_gather_begin(const void* send_dev, void* recv_dev, uint64_t bytes_per_rank, int slot) {
    if (slot < 0 || slot >= COMM_BLOCKS) return SP_E_INVALID_ARG;
    if (!comm_async()) return all_gather(send_dev, recv_dev, bytes_per_rank);
    if (!comm_stream_) {
        SP_HIP_CHECK(hipStreamCreateWithFlags(&comm_stream_, hipStreamNonBlocking));
        SP_HIP_CHECK(hipEventCreateWithFlags(&ev_comm_fork_, hipEventDisableTiming));
        for (auto& e : ev_comm_done_) SP_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    SP_HIP_CHECK(hipEventRecord(ev_comm_fork_, c_->stream));            // the send block is complete behind this point
    SP_HIP_CHECK(hipStreamWaitEvent(comm_stream_, ev_comm_fork_, 0));
    int rc;
    { CommSpan span(c_, comm_stream_); rc = c_->allgather_async(c_->allgather_user, send_dev, recv_dev, bytes_per_rank, comm_stream_); }
    if (rc != 0) { sp_set_error("stream-ordered all-gather failed (" + std::to_string(rc) + ")"); return SP_E_HIP; }
    SP_HIP_CHECK(hipEventRecord(ev_comm_done_[slot], comm_stream_));
    c_->stat_ag_calls += 1; c_->stat_ag_bytes += bytes_per_rank; c_->stat_recv_bytes += bytes_per_rank * (world_ - 1);
    return SP_OK;
}
int StarkProver::all_gather_end(int slot) {
    if (!comm_async()) return SP_OK;
    SP_HIP_CHECK(hipStreamWaitEvent(c_->stream, ev_comm_done_[slot], 0));
    return SP_OK;
}

// Block d of `send` goes to the rank with role d; recv[s] = what role s addressed to this rank.  One all-to-all when the
// hook exists (every rank is its own role then); otherwise an all-gather of the whole send array and a local selection.
int StarkProver::exchange_blocks(const void* send_dev, void* recv_dev, uint64_t bytes, bool stream_ordered) {
    if (stream_ordered && comm_async() && c_->alltoall_async && world_ == G_) {
        int rc;
        { CommSpan span(c_, c_->stream); rc = c_->alltoall_async(c_->allgather_user, send_dev, recv_dev, bytes, c_->stream); }
        if (rc != 0) { sp_set_error("stream-ordered all-to-all failed (" + std::to_string(rc) + ")"); return SP_E_HIP; }
        c_->stat_a2a_calls += 1; c_->stat_a2a_bytes += bytes * (G_ - 1); c_->stat_recv_bytes += bytes * (G_ - 1);
        return SP_OK;
    }
    if (c_->alltoall && world_ == G_) {   // (also with a stream-ordered all-gather but no such all-to-all: one host round trip beats G times the bytes)
        SP_HIP_CHECK(sp_stream_wait_polling(c_->stream));
        int rc;
        { BlockingSpan span(c_); rc = c_->alltoall(c_->allgather_user, send_dev, recv_dev, bytes); }
        if (rc != 0) { sp_set_error("all-to-all hook failed (" + std::to_string(rc) + ")"); return SP_E_HIP; }
        c_->stat_a2a_calls += 1; c_->stat_a2a_bytes += bytes * (G_ - 1); c_->stat_recv_bytes += bytes * (G_ - 1);
        return SP_OK;
    }
    const uint64_t per_rank = bytes * G_;
    SP_TRY(ensure_gather((per_rank * world_ + sizeof(fe) - 1) / sizeof(fe)));
    SP_TRY(all_gather(send_dev, d_gather_, per_rank, stream_ordered));
    const uint8_t* g = reinterpret_cast<const uint8_t*>(d_gather_);
    for (uint32_t src = 0; src < G_; ++src)   // the first G slots are the G roles
        SP_HIP_CHECK(hipMemcpyAsync(static_cast<uint8_t*>(recv_dev) + (uint64_t)src * bytes, g + (uint64_t)src * per_rank + (uint64_t)rank_ * bytes, bytes,
                                    hipMemcpyDeviceToDevice, c_->stream));
    return SP_OK;
}

// batch_commit (reference prover.rs:96-104) / FriLayer::new's tree (fri_commitment.rs:39) over leaves this rank holds in local
// natural order.  Several ranks: the 32-byte leaf digests are exchanged so that rank d owns the contiguous leaves
// [d N/G, (d+1) N/G) - block d of the local digest array is exactly this rank's share of that range - then every rank reduces
// its subtree, the G subtree roots are all-gathered and the top log2 G levels finished everywhere.
int StarkProver::commit_local(const fe* cols_dev, uint64_t stride, uint32_t ncols, uint64_t L, LdeOrder order, TreeBuf& tree, uint8_t root_out[32],
                              bool single_element_tree, const FriChallenge* ch) {
    const MerkleHash mh = merkle_hash(single_element_tree);
    const bool head_done = leaf_head_done_;
    leaf_head_done_ = false;
    if (tree.top == tree.sub) {   // the whole tree on this rank
        if (head_done && mh == MerkleHash::KECCAK256 && merkle_split_supported(ncols))   // the first 17 columns were absorbed while the rest uploaded
            SP_TRY(merkle_hash_leaves_tail(c_->stream, cols_dev, stride, ncols, L, reinterpret_cast<const uint64_t*>(d_scratch_), tree.sub, order));
        else
        SP_TRY(merkle_hash_leaves(c_->stream, cols_dev, stride, ncols, L, tree.sub, order, mh));
        SP_TRY(merkle_reduce(c_->stream, tree.sub, L, ch, mh));
    } else {
        if (L != tree.sub_leaves || L > Nl_) return SP_E_STATE;
        // (both exchanges are consumed on the compute stream, and the read-back of the root below waits for it: stream-ordered
        // where the transport can - two host round trips less per commitment)
        SP_TRY(merkle_hash_leaves_flat(c_->stream, cols_dev, stride, ncols, L, reinterpret_cast<digest32*>(d_local_), order, mh));
        SP_TRY(exchange_blocks(d_local_, d_recv_, (L >> logG_) * sizeof(digest32), true));
        // recv[s][j] = leaf (first + j) G + s of the global order = leaf j G + s of this rank's range
        SP_TRY(interleave_shards(c_->stream, d_recv_, tree.sub + (L - 1), L >> logG_, ShardMap{logG_, logG_, 0}));
        SP_TRY(merkle_reduce(c_->stream, tree.sub, L, nullptr, mh));
        SP_TRY(all_gather(tree.sub, d_roots_, sizeof(digest32), true));
        SP_HIP_CHECK(hipMemcpyAsync(tree.top + (G_ - 1), d_roots_, G_ * sizeof(digest32), hipMemcpyDeviceToDevice, c_->stream));
        SP_TRY(merkle_reduce(c_->stream, tree.top, G_, ch, mh));
    }
    if (ch) return SP_OK;
    return readback(root_out, tree.top, 32);
}

int StarkProver::commit_trace(int segment, const uint8_t* rows_host, uint32_t cols, uint8_t root_out[32], TraceSource src, int col_enc, uint64_t col_stride) {
    const uint32_t binary_hint = binary_cols_hint_;   // (this call only, whatever becomes of it: a later table on this prover need not be a Cairo trace)
    binary_cols_hint_ = 0;
    if (!rows_host || !root_out) return SP_E_INVALID_ARG;
    if (!((segment == 0 && stage_ == 1 && cols == Cm_) || (segment == 1 && stage_ == 2 && cols == Ca_))) {
        sp_set_error("commit_trace: wrong segment order or column count");
        return SP_E_STATE;
    }
    SP_HIP_CHECK(hipSetDevice(c_->device));
    for (double& x : c_->upload_stats) x = 0.0;
    leaf_head_done_ = false;
    // An upload that fails half-way must not return while copies from the caller's buffer are still in flight (the caller is free
    // to release it): every non-OK exit of the host paths waits for the copy and the compute stream first.
    auto drained = [this](int rc) {
        if (rc != SP_OK) {
            if (copy_stream_) (void)hipStreamSynchronize(copy_stream_);
            (void)hipStreamSynchronize(c_->stream);
            (void)hipGetLastError();
            leaf_head_done_ = false;
        }
        return rc;
    };
    if (src == TRACE_DEVICE_BUILD) {
        if (segment != 0) return SP_E_INVALID_ARG;
        return drained(commit_trace_built(*reinterpret_cast<const TraceBuildInput*>(rows_host), root_out));
    }
    if (src == TRACE_HOST_COLUMNS) {
        if (col_enc >= 0 && col_enc != SP_FE_MONT_LIMBS && col_enc != SP_FE_CANON_BE) return SP_E_INVALID_ARG;
        if (col_stride && col_stride < n_) return SP_E_INVALID_ARG;
        return drained(commit_trace_columns(segment, rows_host, cols, col_enc, col_stride ? col_stride : n_, root_out));
    }
    const bool rows_on_device = src == TRACE_DEVICE_ROWS;
    static const uint64_t pipeline_min_bytes = [] { const char* e = std::getenv("SP_UPLOAD_MIN_MB"); return (uint64_t)(e ? std::max(0, std::atoi(e)) : 64) << 20; }();
    static const bool no_pack = std::getenv("SP_UPLOAD_NO_FLAG_PACK") != nullptr;      // (A/B switch)
    const uint32_t binary_cols = (segment == 0 && !no_pack && (n_ & 63) == 0) ? std::min(binary_hint, cols) : 0u;
    if (!rows_on_device && G_ == 1 && cols >= 8 && (uint64_t)n_ * cols * 32 >= pipeline_min_bytes) {
        int rc = drained(commit_trace_pipelined(segment, rows_host, cols, root_out, 0, 0, false, binary_cols));
        if (rc == SP_RETRY_RAW_UPLOAD) rc = drained(commit_trace_pipelined(segment, rows_host, cols, root_out));   // (not a trace with 0 / 1 flags)
        return rc;
    }
    if (!rows_on_device && G_ > 1 && cols >= 2 * G_ && (uint64_t)n_ * cols * 32 >= pipeline_min_bytes) {
        int rc = drained(commit_trace_rows_sharded(segment, rows_host, cols, root_out, binary_cols));
        if (rc == SP_RETRY_RAW_UPLOAD) rc = drained(commit_trace_rows_sharded(segment, rows_host, cols, root_out, 0));
        return rc;
    }
    const uint32_t col0 = segment == 0 ? 0 : Cm_;
    // staging: the raw rows sit in this segment's (not yet written) LDE area: cols*N*32 >= cols*n*32 bytes
    uint8_t* raw = reinterpret_cast<uint8_t*>(d_lde_ + (uint64_t)col0 * std::max<uint64_t>(Nl_, n_));
    fe* trace = d_trace_ + (uint64_t)col0 * n_;
    if (rows_on_device) {
        SP_TRY(rows_to_columns(c_->stream, c_->enc, rows_host, n_, cols, trace, n_));
    } else {
        SP_HIP_CHECK(hipMemcpyAsync(raw, rows_host, (size_t)n_ * cols * 32, hipMemcpyHostToDevice, c_->stream));
        SP_TRY(rows_to_columns(c_->stream, c_->enc, raw, n_, cols, trace, n_));
    }
    if (segment == 0) SP_TRY(launch_aux_presort());
    return commit_segment_resident(segment, cols, root_out);
}

// interpolate_and_commit (reference prover.rs:126-159) from the reference's row-major host table with SEVERAL ranks.  Every rank
// holds the same table (the shim's `&TraceTable`), so no rank needs to push all of it through its own PCIe link - and on one host
// the G gathers of the whole table would go through the same memory controllers: the rank with role r gathers and uploads the
// ceil(cols / G) columns from min(r cpr, cols - cpr) on only (the upload ring of the one-GPU path, restricted to that window), and
// the natural-order columns are all-gathered over the fabric: cols n 32 bytes in total (1.1 GB at 2^20 x 34: ~3 ms on the link
// model) instead of G times that through the host (20 ms of PCIe per rank, and the host's memory bandwidth shared by all).
// Every rank needs the whole trace anyway - the auxiliary trace and the exact constraint check read it - so the all-gather carries
// trace VALUES and the interpolation follows as configured (SP_OPT_SHARD_INTERPOLATION).
int StarkProver::commit_trace_rows_sharded(int segment, const uint8_t* rows_host, uint32_t cols, uint8_t root_out[32], uint32_t binary_cols) {
    const uint32_t col0 = segment == 0 ? 0 : Cm_;
    const uint32_t cpr = (cols + G_ - 1) / G_;
    auto first_col = [&](uint32_t role) { return std::min(role * cpr, cols - cpr); };
    fe* trace = d_trace_ + (uint64_t)col0 * n_;
    {
        // (a rank whose window holds a cell that breaks the 0 / 1 hint must not leave the others waiting in the all-gather: the verdict
        // of the window upload is agreed on first - one flag per rank through the same all-gather)
        // The verdict of the window upload is agreed on before the all-gather of the trace - one word per rank through the same
        // transport - so that a rank whose window breaks the 0 / 1 hint, or whose upload failed, does not leave the others waiting.
        const int rc = commit_trace_pipelined(segment, rows_host, cols, root_out, first_col(rank_), cpr, true, binary_cols);
        if (!d_flags_all_) SP_TRY(alloc((void**)&d_flags_all_, sizeof(int) * world_));
        const int mine = rc == SP_OK ? 0 : (rc == SP_RETRY_RAW_UPLOAD ? 1 : 2);
        SP_HIP_CHECK(hipMemcpyAsync(c_->d_flag, &mine, sizeof(int), hipMemcpyHostToDevice, c_->stream));
        SP_TRY(all_gather(c_->d_flag, d_flags_all_, sizeof(int)));
        std::vector<int> flags(world_, 0);
        SP_HIP_CHECK(hipMemcpy(flags.data(), d_flags_all_, sizeof(int) * world_, hipMemcpyDeviceToHost));
        if (rc != SP_OK && rc != SP_RETRY_RAW_UPLOAD) return rc;
        int worst = 0;
        for (int f : flags) worst = std::max(worst, f);
        if (worst == 2) { sp_set_error("commit_trace: the upload of another rank failed"); return SP_E_HIP; }
        if (worst == 1) return SP_RETRY_RAW_UPLOAD;
    }
    // (the LDE area of this segment is free until the transforms below: landing zone of the all-gather)
    const uint64_t block = (uint64_t)cpr * n_;
    fe* stage = nullptr;
    if ((uint64_t)world_ * block <= std::max<uint64_t>(Nl_, n_) * cols) stage = d_lde_ + (uint64_t)col0 * std::max<uint64_t>(Nl_, n_);
    else { SP_TRY(ensure_gather((uint64_t)world_ * block)); stage = d_gather_; }
    SP_TRY(all_gather(trace + (uint64_t)first_col(rank_) * n_, stage, block * sizeof(fe), true));
    for (uint32_t role = 0; role < G_; ++role) {      // (the first G slots are the G roles; the own block is in place already)
        if (role == rank_) continue;
        SP_HIP_CHECK(hipMemcpyAsync(trace + (uint64_t)first_col(role) * n_, stage + (uint64_t)role * block, block * sizeof(fe), hipMemcpyDeviceToDevice, c_->stream));
    }
    if (segment == 0) SP_TRY(launch_aux_presort());
    SP_TRY(commit_segment_resident(segment, cols, root_out));
    return finish_upload_stats(pending_up_groups_, pending_up_bytes_, pending_up_gather_ms_, pending_up_host_ms_, 1);
}

// interpolate_and_commit (reference prover.rs:126-159) of the Cairo main segment from the RUN instead of the table: the register
// states and the memory cross PCIe (24 B per step + 32 B per cell: 70 MB where the table has 1.1 GB at 2^20 rows) and the device
// writes the table itself - build_main_trace, reference src/cairo/execution_trace.rs:57-87 (trace_kernels.hip).  With several
// ranks every rank builds the whole trace from the same 70 MB: no rank waits for a gigabyte of host table.
int StarkProver::commit_trace_built(const TraceBuildInput& in, uint8_t root_out[32]) {
    if (!in.plan || !in.image) return SP_E_INVALID_ARG;
    const TracePlan& P = *in.plan;
    TraceImage& I = *in.image;
    if (!I.base || !P.dense || P.n != n_ || P.cols != Cm_ || P.steps == 0 || P.steps > n_) { sp_set_error("commit_trace: the run does not fit the prover's shape"); return SP_E_INVALID_ARG; }
    (void)I.try_pin();          // a run built before this process had a context: page-locked from here on
    bool image_pinned = false;
    const uint8_t* image = I.current(&image_pinned);
    constexpr uint32_t REG_CHUNKS = 4;
    SP_TRY(ensure_upload(REG_CHUNKS));
    if (!h_wide_ && hipHostMalloc(reinterpret_cast<void**>(&h_wide_), 64, hipHostMallocDefault) != hipSuccess) { h_wide_ = nullptr; sp_set_error("pinned flag slot: allocation failed"); return SP_E_ALLOC; }
    // staging: the image and the builder's scratch sit in this segment's (not yet written) LDE area when they fit
    const size_t need = (size_t)I.bytes + main_trace_scratch_bytes(P.steps);
    uint8_t* stage = reinterpret_cast<uint8_t*>(d_lde_);
    struct Tmp { void* p = nullptr; hipStream_t st; ~Tmp() { if (p) { (void)hipStreamSynchronize(st); (void)hipFree(p); } } } tmp;
    tmp.st = c_->stream;
    if (need > sizeof(fe) * std::max<uint64_t>(Nl_, n_) * Cm_) {
        if (hipMalloc(&tmp.p, need) != hipSuccess) { (void)hipGetLastError(); sp_set_error("commit_trace: staging allocation failed"); return SP_E_ALLOC; }
        stage = static_cast<uint8_t*>(tmp.p);
    }
    // The memory (and the two hole lists behind it) first, then the register states in four chunks on the copy stream: the rows of a
    // chunk's steps are written while the next chunk crosses PCIe - what stays exposed is the memory (29 MB: 0.5 ms at 2^20 rows) and
    // one chunk of registers (6 MB) instead of the whole image (54 MB: 0.96 ms).
    const double t0 = wall_ms();
    SP_HIP_CHECK(hipEventRecord(up_start_, c_->stream));             // the staging area's previous users are behind this point
    SP_HIP_CHECK(hipStreamWaitEvent(copy_stream_, up_start_, 0));
    SP_HIP_CHECK(hipMemsetAsync(c_->d_flag, 0, sizeof(int), c_->stream));
    MainTraceArgs a{};
    a.regs = reinterpret_cast<const uint64_t*>(stage + I.off_regs);
    a.mem = reinterpret_cast<const fe*>(stage + I.off_mem);
    a.missing = reinterpret_cast<const uint16_t*>(stage + I.off_missing);
    a.holes = reinterpret_cast<const uint64_t*>(stage + I.off_holes);
    a.steps = P.steps; a.cells = P.mem_cells; a.n = n_; a.r_rc = P.r_rc; a.r_holes = P.r_holes; a.r_dummy = P.r_dummy; a.n_holes = P.holes.size();
    a.rc_start = P.rc_start; a.rc_count = P.rc_count; a.cols = Cm_; a.trace = d_trace_;
    SP_HIP_CHECK(hipEventRecord(up_ev_[0].dma0, copy_stream_));
    SP_HIP_CHECK(hipMemcpyAsync(stage + I.off_mem, image + I.off_mem, I.bytes - I.off_mem, hipMemcpyHostToDevice, copy_stream_));
    const uint64_t per = ((P.steps + REG_CHUNKS - 1) / REG_CHUNKS + 255) & ~(uint64_t)255;
    uint32_t chunks_used = 0;
    for (uint32_t k = 0; k < REG_CHUNKS; ++k) {
        const uint64_t s0 = std::min<uint64_t>(P.steps, k * per), s1 = std::min<uint64_t>(P.steps, (k + 1) * per);
        if (s0 == s1) break;
        // (one copy in flight at a time: a copy enqueued while the engine is busy may be given a second SDMA engine, and a stream
        // hopping between two ran at 27 - 37 GB/s instead of 56 - see commit_trace_columns)
        SP_HIP_CHECK(hipStreamSynchronize(copy_stream_));
        if (k) SP_HIP_CHECK(hipEventRecord(up_ev_[k].dma0, copy_stream_));
        SP_HIP_CHECK(hipMemcpyAsync(stage + I.off_regs + 24 * s0, image + I.off_regs + 24 * s0, 24 * (s1 - s0), hipMemcpyHostToDevice, copy_stream_));
        SP_HIP_CHECK(hipEventRecord(up_ev_[k].dma1, copy_stream_));
        SP_HIP_CHECK(hipEventRecord(up_ev_[k].ready, copy_stream_));
        SP_HIP_CHECK(hipStreamWaitEvent(c_->stream, up_ev_[k].ready, 0));
        SP_TRY(cairo_main_trace_steps(c_->stream, a, stage + I.bytes, c_->d_flag, s0, s1));
        SP_HIP_CHECK(hipEventRecord(up_ev_[k].done, c_->stream));
        chunks_used = k + 1;
    }
    SP_TRY(cairo_main_trace_finish(c_->stream, a, stage + I.bytes, c_->d_flag));
    SP_HIP_CHECK(hipMemcpyAsync(h_wide_ + 4, c_->d_flag, sizeof(int), hipMemcpyDeviceToHost, c_->stream));
    const double host_ms = wall_ms() - t0;
    SP_TRY(launch_aux_presort());
    SP_TRY(commit_segment_resident(0, Cm_, root_out));        // (its read-back of the root waits for everything above)
    if (h_wide_[4]) { sp_set_error("commit_trace: a trace row reads beyond the run's memory image"); return SP_E_INVALID_ARG; }
    return finish_upload_stats(chunks_used, I.bytes, 0.0, host_ms, image_pinned ? 4 : 5);
}

// Second half of interpolate_and_commit: the segment's columns sit in natural order in d_trace_.
int StarkProver::commit_segment_resident(int segment, uint32_t cols, uint8_t root_out[32]) {
    const uint32_t col0 = segment == 0 ? 0 : Cm_;
    fe* coeffs = d_coeffs_ + (uint64_t)col0 * n_;
    // interpolate_fft (reference trace.rs:104-110): natural -> bit-reversed h-scaled coefficients (the trace stays intact)
    fe* lde = d_lde_ + (uint64_t)col0 * Nl_;
    if (G_ > 1 && shard_interp_ && d_cstage_ && cols >= G_) {
        // columns are independent (prover.rs:174-183): role s interpolates the cpr columns from min(s cpr, cols - cpr) on
        // (the last blocks overlap instead of being ragged), all-gathers bring every coefficient everywhere (§8(e) item 1).
        // With a stream-ordered transport the cpr columns go in up to four blocks: the exchange of block k runs on the
        // communication stream beside the inverse transforms of block k + 1 and the LDE of block k - 1; a blocking transport
        // keeps the one exchange (every call is a host round trip through the hook).
        const uint32_t cpr = (cols + G_ - 1) / G_;
        auto first_col = [&](uint32_t role) { return std::min(role * cpr, cols - cpr); };
        const uint32_t K = comm_async() ? std::min<uint32_t>(cpr, (uint32_t)COMM_BLOCKS) : 1u;
        const uint32_t bc = (cpr + K - 1) / K;
        std::vector<uint8_t> extended(cols, 0);
        struct Block { uint32_t j0, w; fe* stage; };
        std::vector<Block> blocks;
        uint64_t stage_off = 0;
        for (uint32_t j0 = 0; j0 < cpr; j0 += bc) {
            const uint32_t w = std::min(bc, cpr - j0);
            blocks.push_back(Block{j0, w, d_cstage_ + stage_off});
            stage_off += (uint64_t)world_ * w * n_;
        }
        auto finish = [&](size_t k) -> int {   // block k has arrived: into the coefficient array, then evaluate_offset_fft of its columns
            const Block& b = blocks[k];
            SP_TRY(all_gather_end((int)k));
            for (uint32_t role = 0; role < G_; ++role)
                SP_HIP_CHECK(hipMemcpyAsync(coeffs + (uint64_t)(first_col(role) + b.j0) * n_, b.stage + (uint64_t)role * b.w * n_, (uint64_t)b.w * n_ * sizeof(fe),
                                            hipMemcpyDeviceToDevice, c_->stream));
            if (K == 1) return SP_OK;          // one exchange: the whole segment is extended in one launch below
            for (uint32_t role = 0; role < G_; ++role)
                for (uint32_t j = 0; j < b.w;) {   // runs of columns not extended yet (the ranges of the last roles overlap)
                    const uint32_t c = first_col(role) + b.j0 + j;
                    if (extended[c]) { ++j; continue; }
                    uint32_t run = 0;
                    while (j + run < b.w && !extended[c + run]) { extended[c + run] = 1; ++run; }
                    SP_TRY(c_->ntt->lde_coset_major(coeffs + (uint64_t)c * n_, lde + (uint64_t)c * Nl_, (int)logn_, (int)logb_, run, n_, Nl_, (int)logG_, (int)rank_));
                    j += run;
                }
            return SP_OK;
        };
        for (size_t k = 0; k < blocks.size(); ++k) {
            const Block& b = blocks[k];
            fe* mine = b.stage + (uint64_t)wrank_ * b.w * n_;      // in place: slot `rank` of the receive block is the send buffer
            SP_TRY(c_->ntt->dif_natural_to_bitrev_inverse(mine, (int)logn_, b.w, n_, d_t1_, d_trace_ + (uint64_t)(col0 + first_col(rank_) + b.j0) * n_));
            SP_TRY(all_gather_begin(mine, b.stage, (uint64_t)b.w * n_ * sizeof(fe), (int)k));
            if (k > 0) SP_TRY(finish(k - 1));
        }
        SP_TRY(finish(blocks.size() - 1));
        if (K == 1) SP_TRY(c_->ntt->lde_coset_major(coeffs, lde, (int)logn_, (int)logb_, cols, n_, Nl_, (int)logG_, (int)rank_));
    } else {
        SP_TRY(c_->ntt->dif_natural_to_bitrev_inverse(coeffs, (int)logn_, cols, n_, d_t1_, d_trace_ + (uint64_t)col0 * n_));
        // evaluate_offset_fft on the LDE coset (reference prover.rs:161-185)
        SP_TRY(c_->ntt->lde_coset_major(coeffs, lde, (int)logn_, (int)logb_, cols, n_, Nl_, (int)logG_, (int)rank_));
    }
    // batch_commit (reference prover.rs:96-104) straight from the column-major LDE
    SP_TRY(commit_columns(lde, Nl_, cols, segment == 0 ? tree_main_ : tree_aux_, root_out));
    stage_ = segment == 0 ? 2 : 3;
    return SP_OK;
}

// get_pub_memory_addrs (reference cairo/air.rs:500-517) and the matching values -> pm_addr_h_, pm_val_h_
int StarkProver::public_memory_lists(const PublicInputs& pub) {
    const uint64_t pm = pub.public_memory.size();
    pm_addr_h_.clear(); pm_val_h_.clear();
    std::vector<uint64_t> addrs;
    if (const MemorySegment* out = pub.segment(1)) {
        if (out->end < out->start || out->end - out->start > pm) { sp_set_error("commit_aux_cairo: output segment larger than the public memory"); return SP_E_INVALID_ARG; }
        uint64_t output_section = out->end - out->start, program_section = pm - output_section;
        for (uint64_t i = 1; i <= program_section; ++i) addrs.push_back(i);
        for (uint64_t a = out->start; a < out->end; ++a) addrs.push_back(a);
    } else {
        for (uint64_t i = 1; i <= pm; ++i) addrs.push_back(i);
    }
    std::unordered_map<uint64_t, const fe*> by_addr;   // the reference keeps the public memory in a HashMap (cairo/air.rs:163-181)
    by_addr.reserve(pub.public_memory.size() * 2);
    for (auto& kv : pub.public_memory) by_addr.emplace(kv.first, &kv.second);   // first entry of an address wins, as the linear scan did
    for (uint64_t a : addrs) {
        auto it = by_addr.find(a);
        if (it == by_addr.end()) { sp_set_error("commit_aux_cairo: public memory address missing"); return SP_E_INVALID_ARG; }
        pm_addr_h_.push_back(fe_from_u64(a)); pm_val_h_.push_back(*it->second);
    }
    return SP_OK;
}

int StarkProver::ensure_aux_workspace(uint64_t pm) {
    if (d_auxws_ && pm <= auxws_pm_cap_) return SP_OK;
    size_t sort_tmp = 0;
    uint64_t cap = std::max<uint64_t>(pm, 1024);
    size_t bytes = aux_workspace_bytes(n_, cap, &sort_tmp);
    release(d_auxws_, auxws_bytes_);   // (a context reused with a growing public memory must not keep every workspace it outgrew)
    d_auxws_ = nullptr; auxws_bytes_ = 0; auxws_pm_cap_ = 0; presorted_ = false;
    void* base = nullptr;
    SP_TRY(alloc(&base, bytes));
    d_auxws_ = base; auxws_bytes_ = bytes; auxws_pm_cap_ = cap;
    aux_workspace_carve(auxws_, base, n_, cap, sort_tmp);
    return SP_OK;
}

// The sorts of the auxiliary trace need the main trace and the public memory but no challenge: side stream, from the moment
// the natural-order main columns are queued on the compute stream (request_aux_presort + commit_trace(0, ..)).
int StarkProver::launch_aux_presort() {
    const PublicInputs* pub = presort_pub_;
    presort_pub_ = nullptr; presorted_ = false;
    if (!pub || Ca_ != 18 || Cm_ < 34) return SP_OK;
    SP_TRY(public_memory_lists(*pub));
    const uint64_t pm = pm_addr_h_.size();
    SP_TRY(ensure_aux_workspace(pm));
    SP_TRY(ensure_side());
    if (!d_flag_side_) SP_TRY(alloc((void**)&d_flag_side_, 4 * sizeof(int)));
    SP_HIP_CHECK(hipEventRecord(ev_side_fork_, c_->stream));          // the main trace columns are behind this point
    SP_HIP_CHECK(hipStreamWaitEvent(side_stream_, ev_side_fork_, 0));
    SP_HIP_CHECK(hipMemsetAsync(d_flag_side_ + 2, 0, 2 * sizeof(int), side_stream_));
    SP_TRY(cairo_aux_presort(side_stream_, auxws_, d_memcols_, n_, pm_addr_h_.data(), pm_val_h_.data(), pm, d_flag_side_ + 2, d_flag_side_ + 3));
    // the "address beyond the key bits" flag travels to the host behind the sorts: commit_aux_cairo reads it without a round trip of its own
    if (!h_wide_ && hipHostMalloc(reinterpret_cast<void**>(&h_wide_), 64, hipHostMallocDefault) != hipSuccess) { h_wide_ = nullptr; sp_set_error("pinned flag slot: allocation failed"); return SP_E_ALLOC; }
    SP_HIP_CHECK(hipMemcpyAsync(h_wide_, d_flag_side_ + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, side_stream_));   // [0]: malformed-input flag, [1]: key beyond the presort's bits
    SP_HIP_CHECK(hipEventRecord(ev_side_presort_, side_stream_));
    presorted_ = true;
    return SP_OK;
}

int StarkProver::commit_aux_cairo(const PublicInputs& pub, const fe rap[3], uint8_t root_out[32]) {
    if (stage_ != 2 || Ca_ != 18 || Cm_ < 34) { sp_set_error("commit_aux_cairo: main segment not committed or not a Cairo layout"); return SP_E_STATE; }
    SP_HIP_CHECK(hipSetDevice(c_->device));
    bool pre = presorted_;
    const bool presort_ran = presorted_;
    presorted_ = false;
    if (pre) {   // an address beyond the key bits the presort looked at (a trace with a discontinuous memory): sort again, all 64 bits
        SP_HIP_CHECK(hipStreamWaitEvent(c_->stream, ev_side_presort_, 0));
        SP_HIP_CHECK(hipEventSynchronize(ev_side_presort_));   // (the sorts ended beside round 1's transforms: no wait in practice)
        if (h_wide_[1] || h_wide_[0] == 2) pre = false;
    }
    else SP_TRY(public_memory_lists(pub));     // (the presort built them from the same public inputs)
    const uint64_t pm = pm_addr_h_.size();
    if (pm != pub.public_memory.size()) { sp_set_error("commit_aux_cairo: public memory changed since the presort"); return SP_E_STATE; }
    SP_TRY(ensure_aux_workspace(pm));
    SP_HIP_CHECK(hipMemsetAsync(c_->d_flag, 0, sizeof(int), c_->stream));
    if (pre) SP_HIP_CHECK(hipStreamWaitEvent(c_->stream, ev_side_presort_, 0));
    SP_TRY(ensure_side());
    fe* aux_out = d_trace_ + (uint64_t)Cm_ * n_;
    int flag = 0, flag_pre = 0;
    // An address beyond 2^64 (no VM writes one, but the reference proves whatever table it is given): the presort has already said so,
    // or the 64-bit sort below does - then once more with the four-limb sort.
    bool all_limbs = presort_ran && h_wide_[0] == 2;
    for (int attempt = 0; attempt < 2; ++attempt) {
        SP_TRY(cairo_aux_trace_device(c_->stream, auxws_, d_memcols_, n_, pm_addr_h_.data(), pm_val_h_.data(), pm, rap, aux_out, c_->d_flag,
                                      side_stream_, ev_side_fork_, ev_side_aux_, pre, all_limbs));
        SP_HIP_CHECK(hipMemcpyAsync(&flag, c_->d_flag, sizeof(int), hipMemcpyDeviceToHost, c_->stream));
        if (pre) SP_HIP_CHECK(hipMemcpyAsync(&flag_pre, d_flag_side_ + 2, sizeof(int), hipMemcpyDeviceToHost, c_->stream));
        SP_HIP_CHECK(sp_stream_wait_polling(c_->stream));  // flags
        if (!flag) flag = flag_pre;
        if (flag != 2 || all_limbs) break;
        all_limbs = true; pre = false; flag = flag_pre = 0;
        SP_HIP_CHECK(hipMemsetAsync(c_->d_flag, 0, sizeof(int), c_->stream));
    }
    if (flag) { sp_set_error("commit_aux_cairo: a permutation denominator of the auxiliary trace is zero (the reference's batch inversion fails on this trace and these challenges too)"); return flag == 1 ? SP_E_ZERO_INVERSE : SP_E_INVALID_ARG; }
    return commit_segment_resident(1, Ca_, root_out);
}

int StarkProver::composition_precheck(const fe rap[3], const std::vector<BoundaryConstraint>& bcs, uint32_t n_transitions) {
    check_pending_ = false;
    if (stage_ != 3 && !(stage_ == 2 && Ca_ == 0)) { sp_set_error("composition_precheck: trace segments not committed"); return SP_E_STATE; }
    const uint32_t B = (uint32_t)bcs.size();
    if (n_transitions > CAIRO_MAX_TRANSITIONS || B > CAIRO_MAX_BOUNDARY) return SP_E_INVALID_ARG;
    // the check only runs where composition_core would run it (2n-point paths)
    const bool sub_coset = logb_ >= logG_ + 1, pair_path = !sub_coset && G_ > 1 && logb_ == logG_ && d_post_comp0_;
    if (!sub_coset && !pair_path) return SP_OK;
    SP_HIP_CHECK(hipSetDevice(c_->device));
    if (!d_comp_consts_chk_) SP_TRY(alloc((void**)&d_comp_consts_chk_, sizeof(CompositionConsts)));
    if (!h_comp_chk_) h_comp_chk_.reset(new CompositionConsts());
    CompositionConsts& K = *h_comp_chk_;
    std::memset(&K, 0, composition_consts_bytes(1u << logb_));   // (the per-coset tables only as far as this proof's blowup factor reaches)
    for (uint32_t j = 0; j < B; ++j) {
        if (bcs[j].col >= C_) return SP_E_INVALID_ARG;
        K.bcol[j] = bcs[j].col; K.bvalue[j] = bcs[j].value; K.bstep[j] = bcs[j].step;
    }
    K.h = h_;
    K.rap[0] = rap[0]; K.rap[1] = rap[1]; K.rap[2] = rap[2];
    K.two = fe_from_u64(2);
    K.b15 = fe_from_u64(1ULL << 15); K.b16 = fe_from_u64(1ULL << 16); K.b32 = fe_from_u64(1ULL << 32); K.b48 = fe_from_u64(1ULL << 48);
    K.n_boundary = B; K.n_transitions = n_transitions; K.main_cols = Cm_; K.has_rc_builtin = has_rc_ ? 1 : 0;
    SP_HIP_CHECK(hipMemcpyAsync(d_comp_consts_chk_, &K, composition_consts_bytes(1u << logb_), hipMemcpyHostToDevice, c_->stream));
    SP_HIP_CHECK(hipMemsetAsync(c_->d_flag, 0, sizeof(int), c_->stream));
    SP_TRY(cairo_trace_check(c_->stream, d_trace_, n_, d_comp_consts_chk_, c_->d_flag, check_row0(), check_rows()));
    check_pending_ = true;
    return SP_OK;
}

int StarkProver::composition(const fe rap[3], const std::vector<BoundaryConstraint>& bcs, const std::vector<fe>& b_alpha,
                             const std::vector<fe>& b_beta, const std::vector<fe>& t_alpha, const std::vector<fe>& t_beta,
                             const std::vector<uint32_t>& degrees, const std::vector<uint32_t>& exemptions, uint8_t root_out[32]) {
    if (stage_ != 3 && !(stage_ == 2 && Ca_ == 0)) { sp_set_error("composition: trace segments not committed"); return SP_E_STATE; }
    const uint32_t T = (uint32_t)t_alpha.size(), B = (uint32_t)bcs.size();
    if (T > CAIRO_MAX_TRANSITIONS || B > CAIRO_MAX_BOUNDARY || t_beta.size() != T || b_alpha.size() != B || b_beta.size() != B ||
        degrees.size() != T || exemptions.size() != T) return SP_E_INVALID_ARG;
    SP_HIP_CHECK(hipSetDevice(c_->device));
    const uint32_t b = 1u << logb_;
    const fe* roots = nullptr;
    SP_TRY(c_->ntt->roots((int)logN_, &roots));
    // --- boundary denominators: distinct steps -> points g^step
    std::vector<uint64_t> steps;
    CompositionConsts K;
    std::memset(&K, 0, composition_consts_bytes(1u << logb_));   // (the per-coset tables only as far as this proof's blowup factor reaches)
    for (uint32_t j = 0; j < B; ++j) {
        auto it = std::find(steps.begin(), steps.end(), bcs[j].step);
        if (it == steps.end()) { steps.push_back(bcs[j].step); it = steps.end() - 1; }
        K.bden[j] = (uint32_t)(it - steps.begin());
        K.bcol[j] = bcs[j].col;
        K.bvalue[j] = bcs[j].value;
        K.bstep[j] = bcs[j].step;
        if (bcs[j].col >= C_) return SP_E_INVALID_ARG;
    }
    if (steps.size() > 3) { sp_set_error("composition: more than 3 distinct boundary steps"); return SP_E_UNSUPPORTED; }
    std::vector<fe> points;
    for (uint64_t s : steps) points.push_back(fe_pow_u64(g_, s));
    // --- per-coset constants: x^n takes b values h^n w_b^c (reference evaluator.rs:156-171)
    K.h = h_;
    K.rap[0] = rap[0]; K.rap[1] = rap[1]; K.rap[2] = rap[2];
    K.g_last = fe_pow_u64(g_, n_ - 1);
    K.two = fe_from_u64(2);
    K.b15 = fe_from_u64(1ULL << 15); K.b16 = fe_from_u64(1ULL << 16); K.b32 = fe_from_u64(1ULL << 32); K.b48 = fe_from_u64(1ULL << 48);
    K.n_boundary = B; K.n_transitions = T; K.main_cols = Cm_; K.has_rc_builtin = has_rc_ ? 1 : 0;
    {
        fe hn = fe_pow_u64(h_, n_);
        fe wb = host_primitive_root((int)logb_);
        std::vector<fe> zf(b);
        fe xn = hn;
        for (uint32_t c = 0; c < b; ++c) {
            // degree adjustment x^(D - n(deg-1)) with D = 2n (reference cairo/air.rs:855-857): (x^n)^(3-deg)
            fe pw[4];
            pw[0] = fe_one(); pw[1] = xn; pw[2] = fe_sqr(xn); pw[3] = fe_mul(pw[2], xn);
            for (uint32_t k = 0; k < T; ++k) {
                if (degrees[k] < 1 || degrees[k] > 3) return SP_E_UNSUPPORTED;
                K.coef[c][k] = fe_add(fe_mul(t_alpha[k], pw[3 - degrees[k]]), t_beta[k]);
                if (exemptions[k] > 1) return SP_E_UNSUPPORTED;
            }
            for (uint32_t j = 0; j < B; ++j) K.coef[c][T + j] = fe_add(fe_mul(b_alpha[j], xn), b_beta[j]);
            zf[c] = fe_sub(xn, fe_one());
            xn = fe_mul(xn, wb);
        }
        host_batch_inverse(zf);
        for (uint32_t c = 0; c < b; ++c) K.zerofier[c] = zf[c];
    }
    // the kernel hard-codes which Cairo constraints are exempted / selector-gated; check the caller agrees
    {
        CairoAirInfo ref;
        PublicInputs dummy;
        if (has_rc_) dummy.memory_segments.push_back({0, 0, 0});
        ref = cairo_air_info(dummy);
        if (ref.transition_degrees != degrees || ref.transition_exemptions != exemptions) {
            sp_set_error("composition: only the Cairo AIR constraint set is implemented on the device");
            return SP_E_UNSUPPORTED;
        }
    }
    return composition_core(K, points, nullptr, nullptr, true, root_out);
}

int StarkProver::composition_air(const AirDescHost& air, const std::vector<fe>& rap, const std::vector<fe>& b_alpha, const std::vector<fe>& b_beta,
                                 const std::vector<fe>& t_alpha, const std::vector<fe>& t_beta, uint8_t root_out[32]) {
    if (stage_ != 3 && !(stage_ == 2 && Ca_ == 0)) { sp_set_error("composition: trace segments not committed"); return SP_E_STATE; }
    const uint32_t T = (uint32_t)air.degrees.size(), B = (uint32_t)air.boundary.size(), R = (uint32_t)air.offsets.size();
    if (T == 0 || T > AIR_MAX_TRANSITIONS || B > COMP_MAX_BOUNDARY || R == 0 || R > AIR_MAX_OFFSETS || air.exemptions.size() != T ||
        t_alpha.size() != T || t_beta.size() != T || b_alpha.size() != B || b_beta.size() != B || air.ops.size() > AIR_MAX_OPS ||
        air.consts.size() + rap.size() > AIR_MAX_CONSTS || rap.size() != air.n_rap || air.main_cols != Cm_ || air.aux_cols != Ca_ ||
        air.degree_bound_factor < 1) {
        sp_set_error("composition_air: descriptor out of range or inconsistent with the committed trace");
        return SP_E_INVALID_ARG;
    }
    SP_HIP_CHECK(hipSetDevice(c_->device));
    const uint32_t b = 1u << logb_, f = air.degree_bound_factor;
    // --- validate the program (every operand refers to an earlier value, cells exist) and build the device copy: every
    //     value gets a slot of the per-point value file, released after its last use (the program is straight-line)
    std::unique_ptr<AirProgram> prog_holder(new AirProgram());
    AirProgram& prog = *prog_holder;
    std::memset(&prog, 0, sizeof(prog));
    prog.n_ops = (uint32_t)air.ops.size();
    prog.n_offsets = R;
    for (uint32_t k = 0; k < R; ++k) prog.offsets[k] = air.offsets[k];
    std::vector<bool> produced(T, false);
    std::vector<uint32_t> last_use(prog.n_ops, 0);
    for (uint32_t t = 0; t < prog.n_ops; ++t) {
        const AirOpHost& o = air.ops[t];
        bool ok = true;
        switch (o.op) {
            case 0: ok = o.a < R && o.b < C_; break;
            case 1: ok = o.a < air.consts.size() + rap.size(); break;
            case 2: case 3: case 4: ok = o.a < t && o.b < t && air.ops[o.a].op != 5 && air.ops[o.b].op != 5; if (ok) { last_use[o.a] = t; last_use[o.b] = t; } break;
            case 5: ok = o.a < T && o.b < t && air.ops[o.b].op != 5; if (ok) { produced[o.a] = true; last_use[o.b] = t; } break;
            default: ok = false;
        }
        if (!ok) { sp_set_error("composition_air: malformed constraint program"); return SP_E_INVALID_ARG; }
    }
    {
        // Values nobody reads (directly or through other unread values) are not part of the program the device runs: giving such
        // a value "any" slot would overwrite a live one when all 64 are taken.  Liveness backwards from the OUT ops, then slots.
        const uint32_t n_src = prog.n_ops;
        std::vector<uint8_t> live(n_src, 0);
        for (uint32_t t = n_src; t-- > 0;) {
            const AirOpHost& o = air.ops[t];
            if (o.op == 5) { live[t] = 1; live[o.b] = 1; }
            else if (live[t] && o.op >= 2 && o.op <= 4) { live[o.a] = 1; live[o.b] = 1; }
        }
        std::fill(last_use.begin(), last_use.end(), 0u);
        for (uint32_t t = 0; t < n_src; ++t) {
            if (!live[t]) continue;
            const AirOpHost& o = air.ops[t];
            if (o.op >= 2 && o.op <= 4) { last_use[o.a] = t; last_use[o.b] = t; }
            else if (o.op == 5) last_use[o.b] = t;
        }
        std::vector<uint16_t> slot_of(n_src, 0), free_slots;
        for (int sl = AIR_MAX_LIVE - 1; sl >= 0; --sl) free_slots.push_back((uint16_t)sl);
        std::vector<std::vector<uint32_t>> dying(n_src);    // values whose last use is op t
        for (uint32_t t = 0; t < n_src; ++t) if (live[t] && air.ops[t].op != 5) dying[last_use[t]].push_back(t);
        uint32_t emitted = 0;
        for (uint32_t t = 0; t < n_src; ++t) {
            if (!live[t]) continue;
            const AirOpHost& o = air.ops[t];
            AirOpDev d{};
            d.op = o.op;
            if (o.op >= 2 && o.op <= 4) { d.a = slot_of[o.a]; d.b = slot_of[o.b]; }
            else if (o.op == 5) { d.a = o.a; d.b = slot_of[o.b]; }
            else { d.a = o.a; d.b = o.b; }
            for (uint32_t v : dying[t]) free_slots.push_back(slot_of[v]);   // operands read before the result is written
            if (o.op != 5) {
                if (free_slots.empty()) { sp_set_error("composition_air: more than 64 values alive at once in the constraint program"); return SP_E_UNSUPPORTED; }
                d.dst = free_slots.back(); free_slots.pop_back();
                slot_of[t] = d.dst;
            }
            prog.ops[emitted++] = d;
        }
        prog.n_ops = emitted;
    }
    for (size_t i = 0; i < air.consts.size(); ++i) prog.consts[i] = air.consts[i];
    for (size_t i = 0; i < rap.size(); ++i) prog.consts[air.consts.size() + i] = rap[i];
    // --- transition exemptions (traits.rs:49-79, evaluator.rs:299-323): distinct non-zero counts; with
    //     num_transition_exemptions == 1 every exempted constraint uses the first of them
    std::vector<uint32_t> uniq;
    for (uint32_t e : air.exemptions) if (e > 0 && std::find(uniq.begin(), uniq.end(), e) == uniq.end()) uniq.push_back(e);
    if (uniq.size() > AIR_MAX_EXEMPT_KINDS) { sp_set_error("composition_air: too many distinct exemption counts"); return SP_E_UNSUPPORTED; }
    uint32_t max_ex = 0;
    for (size_t q = 0; q < uniq.size(); ++q) { prog.ex_count[q] = uniq[q]; max_ex = std::max(max_ex, uniq[q]); }
    if (max_ex >= n_) { sp_set_error("composition_air: exemptions exceed the trace length"); return SP_E_INVALID_ARG; }
    uint64_t deg_bound = 0;   // of H for a constraint-satisfying trace
    for (uint32_t k = 0; k < T; ++k) {
        const uint32_t e = air.exemptions[k], d = air.degrees[k];
        if (d < 1 || d > f + 1) { sp_set_error("composition_air: transition degree above the composition degree bound"); return SP_E_INVALID_ARG; }
        if (e) {
            size_t idx = air.num_transition_exemptions == 1 ? 0 : (size_t)(std::find(uniq.begin(), uniq.end(), e) - uniq.begin());
            prog.ex_kind[k] = 1 + (uint32_t)idx;
        }
        // deg C_k <= d (n - 1); times x^(n (f - d + 1)); times the exemption product; over x^n - 1
        const uint32_t ex_used = prog.ex_kind[k] ? prog.ex_count[prog.ex_kind[k] - 1] : 0;
        prog.ex_rows[k] = ex_used;   // rows the composition really exempts for this constraint (what the trace check must mirror)
        deg_bound = std::max<uint64_t>(deg_bound, (uint64_t)d * (n_ - 1) + n_ * (f - d + 1) + ex_used - n_ + 1);
    }
    deg_bound = std::max<uint64_t>(deg_bound, (n_ - 1) + n_ * (f - 1));   // boundary terms
    const bool allow_sub = deg_bound <= 2 * n_;                            // deg H < 2n: 2n evaluations fix it
    if (max_ex > ex_roots_cap_) {
        SP_TRY(alloc((void**)&d_ex_roots_, sizeof(fe) * std::max<uint32_t>(max_ex, 64)));
        ex_roots_cap_ = std::max<uint32_t>(max_ex, 64);
    }
    if (!d_air_prog_) SP_TRY(alloc((void**)&d_air_prog_, sizeof(AirProgram)));
    if (max_ex) {
        std::vector<fe> er(max_ex);
        for (uint32_t j = 0; j < max_ex; ++j) er[j] = fe_pow_u64(g_, n_ - 1 - j);
        SP_HIP_CHECK(hipMemcpyAsync(d_ex_roots_, er.data(), sizeof(fe) * max_ex, hipMemcpyHostToDevice, c_->stream));
        SP_HIP_CHECK(sp_stream_wait_polling(c_->stream));
    }
    SP_HIP_CHECK(hipMemcpyAsync(d_air_prog_, &prog, sizeof(prog), hipMemcpyHostToDevice, c_->stream));
    SP_HIP_CHECK(sp_stream_wait_polling(c_->stream));  // prog is a stack object
    // --- boundary data and per-coset constants
    std::vector<uint64_t> steps;
    CompositionConsts K;
    std::memset(&K, 0, composition_consts_bytes(1u << logb_));   // (the per-coset tables only as far as this proof's blowup factor reaches)
    for (uint32_t j = 0; j < B; ++j) {
        const BoundaryConstraint& bc = air.boundary[j];
        if (bc.col >= C_ || bc.step >= n_) return SP_E_INVALID_ARG;
        auto it = std::find(steps.begin(), steps.end(), bc.step);
        if (it == steps.end()) { steps.push_back(bc.step); it = steps.end() - 1; }
        K.bden[j] = (uint32_t)(it - steps.begin());
        K.bcol[j] = bc.col; K.bvalue[j] = bc.value; K.bstep[j] = bc.step;
    }
    if (steps.size() > 3) { sp_set_error("composition: more than 3 distinct boundary steps"); return SP_E_UNSUPPORTED; }
    std::vector<fe> points;
    for (uint64_t st : steps) points.push_back(fe_pow_u64(g_, st));
    K.h = h_;
    K.n_boundary = B; K.n_transitions = T; K.main_cols = Cm_;
    {
        fe hn = fe_pow_u64(h_, n_);
        fe wb = host_primitive_root((int)logb_);
        std::vector<fe> zf(b);
        fe xn = hn;
        for (uint32_t c = 0; c < b; ++c) {
            // degree adjustments x^(D - n (deg - 1)) and x^(D - n) with D = f n are powers of x^n (evaluator.rs:142-154, :78-82)
            for (uint32_t k = 0; k < T; ++k) K.coef[c][k] = fe_add(fe_mul(t_alpha[k], fe_pow_u64(xn, f - air.degrees[k] + 1)), t_beta[k]);
            for (uint32_t j = 0; j < B; ++j) K.coef[c][T + j] = fe_add(fe_mul(b_alpha[j], fe_pow_u64(xn, f - 1)), b_beta[j]);
            zf[c] = fe_sub(xn, fe_one());
            xn = fe_mul(xn, wb);
        }
        host_batch_inverse(zf);
        for (uint32_t c = 0; c < b; ++c) K.zerofier[c] = zf[c];
    }
    offsets_ = air.offsets;
    return composition_core(K, points, d_air_prog_, d_ex_roots_, allow_sub, root_out);
}

// Shared second half of round 2: K (per-coset coefficients, zerofier, boundary data) is complete; `points` are the distinct
// boundary points g^step.  prog_dev == nullptr: the Cairo kernels; otherwise the constraint program of a generic AIR.
// allow_sub_coset: the caller knows deg H < 2n for a constraint-satisfying trace.
int StarkProver::composition_core(const CompositionConsts& K, const std::vector<fe>& points, const AirProgram* prog_dev,
                                  const fe* ex_roots_dev, bool allow_sub_coset, uint8_t root_out[32]) {
    const fe* roots = nullptr;
    SP_TRY(c_->ntt->roots((int)logN_, &roots));
    fe* comp = nullptr;                      // [N] whole-domain composition evaluations (exceptional paths only, fetched below)
    const uint32_t nd = (uint32_t)points.size();
    auto evaluate = [&](uint64_t count, uint32_t stride_log, const fe* binv, fe* out) -> int {
        if (prog_dev) return air_composition(c_->stream, d_lde_, count, Nl_, stride_log, logN_, logb_, roots, d_comp_consts_, prog_dev, ex_roots_dev, binv, out, logG_, rank_);
        return cairo_composition(c_->stream, d_lde_, count, Nl_, stride_log, logN_, logb_, roots, d_comp_consts_, binv, out, logG_, rank_);
    };
    SP_HIP_CHECK(hipSetDevice(c_->device));
    const bool prechecked = check_pending_ && !prog_dev;   // composition_precheck queued the constraint check (and cleared the flag) already
    check_pending_ = false;
    SP_HIP_CHECK(hipMemcpyAsync(d_comp_consts_, &K, composition_consts_bytes(1u << logb_), hipMemcpyHostToDevice, c_->stream));
    if (!prechecked) SP_HIP_CHECK(hipMemsetAsync(c_->d_flag, 0, sizeof(int), c_->stream));
    // A trace that satisfies its constraints gives deg H < 2n, and then 2n evaluations fix H.  Decide that EXACTLY by
    // checking the constraints on the trace itself (n rows, no divisions): clean -> evaluate the composition on the 2n
    // points of the cosets 0 and b/2 only; otherwise (the reference still proves such traces, with longer H1/H2) fall
    // back to the whole domain and the general split, so the bytes are identical for every input.
    int flag = 0;
    int flag_pref = 0;
    bool pair_flag_pending = false;
    bool sub_coset = allow_sub_coset && logb_ >= logG_ + 1;  // this rank holds both cosets c0 = rank and c0 + b/2 (always on one GPU)
    // one coset per rank (G = b): the 2n points of the cosets 0 and b/2 live on two ranks - every rank evaluates its own coset, the
    // evaluations are all-gathered and the pair (0, b/2) is interpolated everywhere
    bool pair_path = allow_sub_coset && !sub_coset && G_ > 1 && logb_ == logG_ && d_post_comp0_;
    if (sub_coset || pair_path) {
        if (prog_dev) SP_TRY(air_trace_check(c_->stream, d_trace_, n_, d_comp_consts_, prog_dev, c_->d_flag));
        else if (!prechecked) SP_TRY(cairo_trace_check(c_->stream, d_trace_, n_, d_comp_consts_, c_->d_flag, check_row0(), check_rows()));
        if (!prog_dev && world_ > 1 && n_ >= 256ull * world_) {
            // every rank checked its own n / world rows of the (replicated) trace: one flag per rank, combined everywhere
            if (!d_flags_all_) SP_TRY(alloc((void**)&d_flags_all_, sizeof(int) * world_));
            SP_TRY(all_gather(c_->d_flag, d_flags_all_, sizeof(int), true));
            std::vector<int> flags(world_, 0);
            SP_HIP_CHECK(hipMemcpyAsync(flags.data(), d_flags_all_, sizeof(int) * world_, hipMemcpyDeviceToHost, c_->stream));
            SP_HIP_CHECK(sp_stream_wait_polling(c_->stream));  // (also: K is a stack object)
            for (int f : flags) flag |= f;
        } else {
            SP_HIP_CHECK(hipMemcpyAsync(&flag, c_->d_flag, sizeof(int), hipMemcpyDeviceToHost, c_->stream));
            SP_HIP_CHECK(sp_stream_wait_polling(c_->stream));  // (also: K is a stack object)
        }
        sub_coset = sub_coset && flag == 0;
        pair_path = pair_path && flag == 0;
        SP_HIP_CHECK(hipMemsetAsync(c_->d_flag, 0, sizeof(int), c_->stream));
    } else {
        if (prechecked) SP_HIP_CHECK(hipMemsetAsync(c_->d_flag, 0, sizeof(int), c_->stream));   // (unreachable today: same conditions)
        SP_HIP_CHECK(sp_stream_wait_polling(c_->stream));  // K is a stack object
    }
    if (sub_coset) {
        const uint64_t M = 2 * n_;
        const fe* roots_m = nullptr;
        SP_TRY(c_->ntt->roots((int)logn_ + 1, &roots_m));
        // the 2n points are x_i = hp w_2n^i with hp = h w_N^c0 (c0 = rank: under coset sharding every rank works on its own
        // pair of cosets and obtains the same polynomial, so the composition evaluations need no all-gather)
        const fe wN = host_primitive_root((int)logN_);
        const fe hp = fe_mul(h_, fe_pow_u64(wN, rank_));
        fe* binv = d_scratch_;                    // [ndist][2n]
        fe* inv_scratch = d_scratch_ + 3 * M;     // [3 * 2n]
        bool pref = bpre_valid_ && nd == bpre_points_.size();
        for (uint32_t j = 0; pref && j < nd; ++j) pref = fe_eq(points[j], bpre_points_[j]);
        if (nd && pref) {                         // computed beside round 1 (prefetch_boundary_inverses)
            binv = d_bpre_;
            SP_HIP_CHECK(hipStreamWaitEvent(c_->stream, ev_side_bnd_, 0));
        } else if (nd) {
            SP_TRY(coset_minus_points(c_->stream, binv, M, logn_ + 1, roots_m, hp, points.data(), nd, ShardMap{0, 0, 0}));
            SP_TRY(batch_inverse(c_->stream, binv, inv_scratch, (uint64_t)nd * M, c_->d_flag));
        }
        if (nd && pref) SP_HIP_CHECK(hipMemcpyAsync(&flag_pref, d_flag_side_ + 1, sizeof(int), hipMemcpyDeviceToHost, c_->stream));
        fe* comp2 = d_h12s_;                      // [2n] evaluations H(h w_2n^i), then [H1s | H2s]
        SP_TRY(evaluate(M, logb_ - logG_ - 1, binv, comp2));
        SP_HIP_CHECK(hipMemcpyAsync(&flag, c_->d_flag, sizeof(int), hipMemcpyDeviceToHost, c_->stream));
        // interpolate_offset_fft + even/odd split in one inverse transform: position q < n of the bit-reversed output is
        // 2n c_j hp^j for j = 2k, position n + q for j = 2k + 1 (k = rev_n(q)); the post factors leave a_k h^k = c_2k h^k
        // and b_k h^k = c_(2k+1) h^k:  (2n)^-1 (h^-1 u^2)^k  and  (2n)^-1 (h^-1 u) (h^-1 u^2)^k,  u = w_N^-c0.
        SP_TRY(c_->ntt->dif_natural_to_bitrev_inverse(comp2, (int)logn_ + 1, 1, M, d_post_comp_));   // (tables: setup())
        pair_flag_pending = true;                 // the two flags are looked at behind the commitment's read-back: no wait of its own here
        h_full_ = false;
        SP_TRY(c_->ntt->lde_coset_major(d_h12s_, d_h12_, (int)logn_, (int)logb_, 2, n_, Nl_, (int)logG_, (int)rank_));
    } else if (pair_path) {
        fe* binv = d_scratch_;                  // [ndist][n]
        fe* inv_scratch = d_scratch_ + 3 * Nl_;  // [3 n]
        if (nd) {
            SP_TRY(coset_minus_points(c_->stream, binv, Nl_, logN_, roots, h_, points.data(), nd, shard_map()));
            SP_TRY(batch_inverse(c_->stream, binv, inv_scratch, (uint64_t)nd * Nl_, c_->d_flag));
        }
        SP_TRY(evaluate(Nl_, 0, binv, d_local_));                   // H on this rank's coset
        SP_HIP_CHECK(hipMemcpyAsync(&flag, c_->d_flag, sizeof(int), hipMemcpyDeviceToHost, c_->stream));
        SP_TRY(ensure_gather((uint64_t)world_ * Nl_));
        SP_TRY(all_gather(d_local_, d_gather_, Nl_ * sizeof(fe), true));
        pair_flag_pending = true;                                    // (checked behind the commitment's read-back, which waits for the stream)
        const uint32_t other = G_ >> 1;                              // the rank that holds coset b/2
        if (other != 1) SP_HIP_CHECK(hipMemcpyAsync(d_gather_ + n_, d_gather_ + (uint64_t)other * n_, n_ * sizeof(fe), hipMemcpyDeviceToDevice, c_->stream));
        fe* comp2 = d_h12s_;                                         // H(h w_2n^i): even i from coset 0, odd i from coset b/2
        SP_TRY(interleave_shards(c_->stream, d_gather_, comp2, n_, ShardMap{1, 1, 0}));
        SP_TRY(c_->ntt->dif_natural_to_bitrev_inverse(comp2, (int)logn_ + 1, 1, 2 * n_, d_post_comp0_));   // post factors of c0 = 0
        h_full_ = false;
        SP_TRY(c_->ntt->lde_coset_major(d_h12s_, d_h12_, (int)logn_, (int)logb_, 2, n_, Nl_, (int)logG_, (int)rank_));
    } else {
        SP_TRY(full_domain_buffer(&comp));
        fe* comp_local = G_ == 1 ? comp : d_local_;
        fe* binv = d_scratch_;                  // [ndist][Nl]
        fe* inv_scratch = d_scratch_ + 3 * Nl_;  // [3 Nl]
        if (nd) {
            SP_TRY(coset_minus_points(c_->stream, binv, Nl_, logN_, roots, h_, points.data(), nd, shard_map()));
            SP_TRY(batch_inverse(c_->stream, binv, inv_scratch, (uint64_t)nd * Nl_, c_->d_flag));
        }
        SP_TRY(evaluate(Nl_, 0, binv, comp_local));
        if (G_ > 1) {  // composition-polynomial reduction: all-gather the per-coset evaluations
            SP_TRY(ensure_gather((uint64_t)world_ * Nl_));
            SP_TRY(all_gather(comp_local, d_gather_, Nl_ * sizeof(fe), true));
            SP_TRY(interleave_shards(c_->stream, d_gather_, comp, n_, shard_map()));
        }
        // --- interpolate_offset_fft + even/odd split (reference evaluation_table.rs:27-33, prover.rs:250-252)
        SP_TRY(c_->ntt->dif_natural_to_bitrev_inverse(comp, (int)logN_, 1, N_, nullptr));
        SP_HIP_CHECK(hipMemcpyAsync(&flag, c_->d_flag, sizeof(int), hipMemcpyDeviceToHost, c_->stream));
        SP_HIP_CHECK(sp_stream_wait_polling(c_->stream));
        if (flag) { sp_set_error("composition: zero boundary denominator"); return SP_E_ZERO_INVERSE; }
        SP_TRY(high_coeff_check(c_->stream, comp, N_, logb_, c_->d_flag));
        SP_HIP_CHECK(hipMemcpyAsync(&flag, c_->d_flag, sizeof(int), hipMemcpyDeviceToHost, c_->stream));
        SP_HIP_CHECK(sp_stream_wait_polling(c_->stream));
        h_full_ = flag != 0;
        if (!h_full_) {
            SP_TRY(split_composition(c_->stream, comp, n_, logb_, d_t2_, hinv_, d_h12s_, d_h12s_ + n_));
            SP_TRY(c_->ntt->lde_coset_major(d_h12s_, d_h12_, (int)logn_, (int)logb_, 2, n_, Nl_, (int)logG_, (int)rank_));
        } else {
            // the trace violates its constraints: deg H >= 2n and the reference still proves it (longer H1, H2).  Every rank
            // holds all of H, evaluates H1, H2 on the whole domain and keeps the points of its own cosets.
            if (!d_hfull_) SP_TRY(alloc((void**)&d_hfull_, sizeof(fe) * N_));
            fe* t_half = d_scratch_;  // N/2 entries: N^-1 h^(-rev_{N/2}(q))
            if ((N_ >> 1) > scratch_elems()) { sp_set_error("composition: scratch too small"); return SP_E_ALLOC; }
            fe Ninv = fe_inv(fe_from_u64(N_));
            SP_TRY(gen_power_table(c_->stream, t_half, N_ >> 1, logN_ - 1, hinv_, Ninv));
            SP_TRY(split_composition_full(c_->stream, comp, N_, t_half, hinv_, d_hfull_, d_hfull_ + (N_ >> 1)));
            // H1, H2 of N/2 coefficients each: natural-order evaluations first, then into the coset-major order of every other column
            if (!d_hnat_) SP_TRY(alloc((void**)&d_hnat_, sizeof(fe) * N_ * 2));
            SP_TRY(c_->ntt->lde_from_bitrev(d_hfull_, d_hnat_, (int)logN_ - 1, 1, 2, N_ >> 1, N_));
            SP_TRY(natural_to_coset_major(c_->stream, d_hnat_, N_, d_h12_, Nl_, 2, lde_order(), logG_, rank_));
        }
    }
    bpre_valid_ = false;
    c_->proof_info[0] = (sub_coset || pair_path) ? 1u : (h_full_ ? 3u : 2u);
    c_->proof_info[1] = fri_rep_; c_->proof_info[2] = G_; c_->proof_info[3] = (G_ > 1 && shard_interp_) ? 1u : 0u;
    SP_TRY(commit_columns(d_h12_, Nl_, 2, tree_comp_, root_out));
    if (pair_flag_pending && (flag | flag_pref)) { sp_set_error("composition: zero boundary denominator"); return SP_E_ZERO_INVERSE; }
    stage_ = 4;
    return SP_OK;
}

// sum_q A[q] y^rev(q) for `vectors` arrays of 2^k elements and `points` points, by repeated folding
// (DESIGN.md "Out-of-domain evaluation"): one level maps M elements to M >> l.
// scratch: at least 3 * 2^k elements (two ping-pong buffers and the per-level power tables).
// after_first (nullable): host work to do once the first - long - level is queued (it runs beside that kernel).
static int eval_bitrev(sp_ctx* c, const fe* arrays, uint64_t vec_stride, uint32_t vectors, uint32_t k, const std::vector<fe>& ys,
                       fe* scratch, uint64_t scratch_elems, std::vector<fe>& out /*[vectors][points]*/,
                       const std::function<int()>* after_first = nullptr) {
    const uint32_t points = (uint32_t)ys.size();
    std::vector<fe> ycur = ys;
    const fe* in = arrays;
    uint64_t in_stride = vec_stride;
    uint32_t in_points = 1;
    uint64_t M = 1ULL << k;
    // level sizes: 2^8 terms per output while that leaves >= 2^16 outputs in flight, then 2^4 (a level with few outputs and
    // long serial sums runs one wave per SIMD for hundreds of microseconds)
    std::vector<uint32_t> ls;
    {
        uint32_t kk = k;
        const double lt0 = std::log2((double)vectors * points);
        while (kk > 0) {
            const int room = (int)(lt0 + kk) - 16;
            uint32_t l = room >= 8 ? 8u : (uint32_t)std::max(4, room);
            l = std::min(l, kk);
            // the two ping-pong buffers hold the first level's outputs: longer sums if the scratch area demands it
            while (ls.empty() && l < std::min<uint32_t>(8, kk) && 2ull * vectors * points * (M >> l) + 8ull * points * 256 > scratch_elems) ++l;
            ls.push_back(l);
            kk -= l;
        }
    }
    const uint64_t level1 = k == 0 ? 0 : (uint64_t)vectors * points * (M >> ls[0]);
    const uint64_t tab_elems = (uint64_t)ls.size() * points * 256;
    if (2 * level1 + tab_elems > scratch_elems) { sp_set_error("eval_bitrev: scratch too small"); return SP_E_ALLOC; }
    fe* bufs[2] = {scratch, scratch + level1};
    fe* yp_dev = scratch + 2 * level1;
    int which = 0;
    if (k == 0) {
        if (after_first) SP_TRY((*after_first)());
        out.resize((size_t)vectors * points);
        for (uint32_t v = 0; v < vectors; ++v) {
            fe t;
            SP_HIP_CHECK(hipMemcpy(&t, arrays + v * vec_stride, sizeof(fe), hipMemcpyDeviceToHost));
            for (uint32_t p = 0; p < points; ++p) out[v * points + p] = t;
        }
        return SP_OK;
    }
    // power tables yp[level][p][t] = y_level^rev_l(t), y_(level+1) = y_level^(2^l): the table of a level goes up right before its
    // kernel, so the host computes the next level's table (~25 us) while the device runs this one; no synchronisation between levels
    std::vector<fe> yp(tab_elems, fe_zero());
    std::vector<fe> pw(256);
    for (size_t lev = 0; lev < ls.size(); ++lev) {
        const uint32_t l = ls[lev], Tn = 1u << l;
        for (uint32_t p = 0; p < points; ++p) {
            pw[0] = fe_one();
            for (uint32_t e = 1; e < Tn; ++e) pw[e] = fe_mul(pw[e - 1], ycur[p]);
            fe* dstp = &yp[lev * points * 256 + (size_t)p * Tn];   // compacted: the kernel indexes yp[p * T + t]
            for (uint32_t t = 0; t < Tn; ++t) {  // yp[t] = y^rev_l(t)
                uint32_t r = 0;
                for (uint32_t bit = 0; bit < l; ++bit) if ((t >> bit) & 1) r |= 1u << (l - 1 - bit);
                dstp[t] = pw[r];
            }
            fe y2 = ycur[p];
            for (uint32_t s = 0; s < l; ++s) y2 = fe_sqr(y2);
            ycur[p] = y2;
        }
        SP_HIP_CHECK(hipMemcpyAsync(yp_dev + lev * points * 256, &yp[lev * points * 256], sizeof(fe) * points * Tn, hipMemcpyHostToDevice, c->stream));
        fe* outb = bufs[which];
        SP_TRY(fold_eval_level(c->stream, in, in_stride, in_points, M, l, yp_dev + lev * points * 256, points, vectors, outb));
        if (lev == 0 && after_first) SP_TRY((*after_first)());
        M >>= l;
        in = outb; in_stride = (uint64_t)points * M; in_points = points;
        which ^= 1;
    }
    SP_HIP_CHECK(sp_stream_wait_polling(c->stream));  // yp is a local vector
    out.resize((size_t)vectors * points);
    SP_HIP_CHECK(hipMemcpyAsync(out.data(), in, out.size() * sizeof(fe), hipMemcpyDeviceToHost, c->stream));
    SP_HIP_CHECK(sp_stream_wait_polling(c->stream));
    return SP_OK;
}

int StarkProver::ood(const fe& z, fe* h1_z2, fe* h2_z2, std::vector<fe>& trace_ood) {
    if (stage_ != 4) { sp_set_error("ood: composition polynomial not committed"); return SP_E_STATE; }
    SP_HIP_CHECK(hipSetDevice(c_->device));
    z_ = z;
    // round 4's denominators depend on z only: side stream, beside the evaluations below (queued once the first of them runs)
    const std::function<int()> prefetch = [this]() { return prefetch_deep_inverses(); };
    // stored coefficients are c_k h^k, so evaluate at y / h (reference prover.rs:301-304, frame.rs:67-83)
    const uint32_t R = (uint32_t)offsets_.size();   // frame rows: z g^ofs for every transition offset (frame.rs:67-83)
    std::vector<fe> ys;
    for (uint32_t k = 0; k < R; ++k) ys.push_back(fe_mul(fe_mul(z, fe_pow_u64(g_, offsets_[k])), hinv_));
    std::vector<fe> tr;
    if (G_ > 1 && C_ >= G_) {
        // the polynomials are independent: role s evaluates the cpr columns from min(s cpr, C - cpr) on, the R values per column
        // are all-gathered (a few KB) - 1/G of the Horner-equivalent work per rank instead of all of it on every rank
        const uint32_t cpr = (C_ + G_ - 1) / G_;
        auto first_col = [&](uint32_t role) { return std::min(role * cpr, C_ - cpr); };
        std::vector<fe> mine;
        SP_TRY(eval_bitrev(c_, d_coeffs_ + (uint64_t)first_col(rank_) * n_, n_, cpr, logn_, ys, d_scratch_, scratch_elems(), mine, &prefetch));
        const size_t blk = (size_t)cpr * R;
        if (!d_small_) SP_TRY(alloc((void**)&d_small_, sizeof(fe) * (1 + (size_t)world_) * 64 * AIR_MAX_OFFSETS));
        SP_HIP_CHECK(hipMemcpyAsync(d_small_, mine.data(), blk * sizeof(fe), hipMemcpyHostToDevice, c_->stream));
        SP_TRY(all_gather(d_small_, d_small_ + 64 * AIR_MAX_OFFSETS, blk * sizeof(fe)));
        std::vector<fe> all(blk * world_);
        SP_HIP_CHECK(hipMemcpy(all.data(), d_small_ + 64 * AIR_MAX_OFFSETS, all.size() * sizeof(fe), hipMemcpyDeviceToHost));
        tr.resize((size_t)C_ * R);
        for (uint32_t role = 0; role < G_; ++role)
            std::copy(all.begin() + (size_t)role * blk, all.begin() + (size_t)(role + 1) * blk, tr.begin() + (size_t)first_col(role) * R);
    } else {
        SP_TRY(eval_bitrev(c_, d_coeffs_, n_, C_, logn_, ys, d_scratch_, scratch_elems(), tr, &prefetch));
    }
    trace_ood.resize((size_t)R * C_);
    for (uint32_t j = 0; j < C_; ++j)
        for (uint32_t k = 0; k < R; ++k) trace_ood[(size_t)k * C_ + j] = tr[(size_t)j * R + k];
    std::vector<fe> yh = {fe_mul(fe_sqr(z), hinv_)};
    std::vector<fe> hv;
    if (!h_full_) SP_TRY(eval_bitrev(c_, d_h12s_, n_, 2, logn_, yh, d_scratch_, scratch_elems(), hv));
    else SP_TRY(eval_bitrev(c_, d_hfull_, N_ >> 1, 2, logN_ - 1, yh, d_scratch_, scratch_elems(), hv));
    h1_z2_ = hv[0]; h2_z2_ = hv[1];
    *h1_z2 = hv[0]; *h2_z2 = hv[1];
    trace_ood_ = trace_ood;
    stage_ = 5;
    return SP_OK;
}

int StarkProver::deep_fri_begin(const fe& gamma, const fe& gamma_p, const std::vector<fe>& tg, uint8_t root0_out[32]) {
    if (stage_ != 5) { sp_set_error("deep_fri_begin: out-of-domain evaluations missing"); return SP_E_STATE; }
    const uint32_t R = (uint32_t)offsets_.size();
    if (tg.size() != (size_t)R * C_) return SP_E_INVALID_ARG;
    SP_HIP_CHECK(hipSetDevice(c_->device));
    const fe* roots = nullptr;
    SP_TRY(c_->ntt->roots((int)logN_, &roots));
    DeepConsts K;
    std::memset(&K, 0, sizeof(K));
    K.gamma_h1 = gamma; K.gamma_h2 = gamma_p;
    K.c_h = fe_add(fe_mul(gamma, h1_z2_), fe_mul(gamma_p, h2_z2_));
    K.cols = C_; K.rows = R;
    for (uint32_t k = 0; k < AIR_MAX_OFFSETS; ++k) K.c_t[k] = fe_zero();
    for (uint32_t j = 0; j < C_; ++j)
        for (uint32_t k = 0; k < R; ++k) {
            K.gammas[k][j] = tg[(size_t)j * R + k];  // reference prover.rs:457-476: gamma index = j * frame_len + k
            K.c_t[k] = fe_add(K.c_t[k], fe_mul(tg[(size_t)j * R + k], trace_ood_[(size_t)k * C_ + j]));
        }
    SP_HIP_CHECK(hipMemcpyAsync(d_deep_consts_, &K, sizeof(K), hipMemcpyHostToDevice, c_->stream));
    SP_HIP_CHECK(sp_stream_wait_polling(c_->stream));
    fe pts[AIR_MAX_OFFSETS + 1];                       // z g^ofs_k for every frame row, then z^2
    for (uint32_t k = 0; k < R; ++k) pts[k] = fe_mul(z_, fe_pow_u64(g_, offsets_[k]));
    pts[R] = fe_sqr(z_);
    const uint32_t npts = R + 1;
    SP_HIP_CHECK(hipMemsetAsync(c_->d_flag, 0, sizeof(int), c_->stream));
    bool used_pref = false;
    if (!h_full_) {
        // deg p0 <= n - 2 (every term is a quotient of a polynomial of degree < n by a linear factor), so p0 is fixed by
        // its values on ONE coset of n points: evaluate the quotient form there only (coset c0 = first coset this rank
        // holds), interpolate, and extend with the same LDE as every other column - 1/b of the pointwise work and of
        // the inversions, the same field elements.  Under coset sharding every rank gets the identical polynomial from
        // its own coset, so FRI layer 0 needs no all-gather.
        const uint32_t shift = logb_ - logG_;                 // local elements per row of the LDE matrix
        const fe* roots_n = nullptr;
        SP_TRY(c_->ntt->roots((int)logn_, &roots_n));
        const fe wN = host_primitive_root((int)logN_);
        const fe hp = fe_mul(h_, fe_pow_u64(wN, rank_));      // offset of that coset: h w_N^c0
        fe* inv = d_scratch_;                                  // [npts][n]
        fe* inv_scratch = d_scratch_ + (uint64_t)npts * n_;    // [npts n]
        fe* p0n = d_scratch_ + 2ull * npts * n_;               // [n]
        if ((2ull * npts + 1) * n_ > scratch_elems()) {   // many frame rows on a small blowup: the inverses outgrow the shared scratch
            SP_TRY(ensure_deep_scratch((2ull * npts + 1) * n_));
            inv = d_deepx_; inv_scratch = inv + (uint64_t)npts * n_; p0n = inv + 2ull * npts * n_;
        }
        if (deep_pref_ && d_deepx_ && deepx_cap_ >= (2ull * npts + 1) * n_) {   // computed beside round 3 (prefetch_deep_inverses)
            inv = d_deepx_; p0n = inv + 2ull * npts * n_;
            SP_HIP_CHECK(hipStreamWaitEvent(c_->stream, ev_side_deep_, 0));
            used_pref = true;
        } else {
            SP_TRY(coset_minus_points(c_->stream, inv, n_, logn_, roots_n, hp, pts, npts, ShardMap{0, 0, 0}));
            SP_TRY(batch_inverse(c_->stream, inv, inv_scratch, (uint64_t)npts * n_, c_->d_flag));
        }
        if (fri_sharded(0) && logb_ == logG_) {
            // one coset per rank and a sharded layer 0: the n points of that coset in natural order ARE this rank's share of the layer
            // (local index = row of the coset), so the quotient form is written there and nothing is interpolated or extended
            SP_TRY(deep_composition(c_->stream, d_lde_, d_h12_, d_h12_ + Nl_, n_, Nl_, shift, d_deep_consts_, inv, d_fri_evals_[0], lde_order(), R));
        } else {
        SP_TRY(deep_composition(c_->stream, d_lde_, d_h12_, d_h12_ + Nl_, n_, Nl_, shift, d_deep_consts_, inv, p0n, lde_order(), R));
        // coefficients c_j h^j in bit-reversed order: inverse DFT over the coset, times n^-1 w_N^(-c0 j)
        SP_TRY(c_->ntt->dif_natural_to_bitrev_inverse(p0n, (int)logn_, 1, n_, d_post_deep_));   // n^-1 w_N^(-c0 j): setup()
        // FRI layer 0: the evaluations this rank holds (local natural order) when the layer is sharded, the whole domain otherwise
        if (fri_sharded(0)) SP_TRY(c_->ntt->lde_from_bitrev(p0n, d_fri_evals_[0], (int)logn_, (int)logb_, 1, n_, Nl_, (int)logG_, (int)rank_));
        else SP_TRY(c_->ntt->lde_from_bitrev(p0n, d_fri_evals_[0], (int)logn_, (int)logb_, 1, n_, N_));
        }
    } else {
        // deg H >= 2n (constraint-violating trace): the quotient form on every LDE point this rank holds
        fe* inv = d_scratch_;
        if (2ull * npts * Nl_ > scratch_elems()) {   // more than two frame rows: the inverses outgrow the shared scratch
            SP_TRY(ensure_deep_scratch(2ull * npts * Nl_));
            inv = d_deepx_;
        }
        fe* inv_scratch = inv + (uint64_t)npts * Nl_;
        SP_TRY(coset_minus_points(c_->stream, inv, Nl_, logN_, roots, h_, pts, npts, shard_map()));
        SP_TRY(batch_inverse(c_->stream, inv, inv_scratch, (uint64_t)npts * Nl_, c_->d_flag));
        fe* p0_local = (G_ == 1 || fri_sharded(0)) ? d_fri_evals_[0] : d_local_;   // local natural order
        SP_TRY(deep_composition(c_->stream, d_lde_, d_h12_, d_h12_ + Nl_, Nl_, Nl_, 0, d_deep_consts_, inv, p0_local, lde_order(), R));
        if (G_ > 1 && !fri_sharded(0)) {
            SP_TRY(ensure_gather((uint64_t)world_ * Nl_));
            SP_TRY(all_gather(p0_local, d_gather_, Nl_ * sizeof(fe), true));
            SP_TRY(interleave_shards(c_->stream, d_gather_, d_fri_evals_[0], n_, shard_map()));
        }
    }
    // FRI layer 0 (reference fri/mod.rs:27-33)
    fri_layer_ = 0;
    fri_offset_ = h_; fri_offset_inv_ = hinv_;
    int flag = 0, flag_pref = 0;
    SP_HIP_CHECK(hipMemcpyAsync(&flag, c_->d_flag, sizeof(int), hipMemcpyDeviceToHost, c_->stream));
    if (used_pref) SP_HIP_CHECK(hipMemcpyAsync(&flag_pref, d_flag_side_, sizeof(int), hipMemcpyDeviceToHost, c_->stream));
    deep_pref_ = false;
    SP_TRY(commit_local(d_fri_evals_[0], 0, 1, fri_trees_[0].sub_leaves, LdeOrder{0, 0, 0}, fri_trees_[0], root0_out, true));   // synchronises
    flag |= flag_pref;
    if (flag) { sp_set_error("deep composition: z lies on the LDE coset"); return SP_E_ZERO_INVERSE; }
    fri_layer_ = 1;
    stage_ = 6;
    return SP_OK;
}

int StarkProver::fri_fold_commit(const fe& zeta, uint8_t root_out[32], fe* last_value, int* is_last) {
    if (stage_ != 6) { sp_set_error("fri_fold_commit: FRI not started or already finished"); return SP_E_STATE; }
    SP_HIP_CHECK(hipSetDevice(c_->device));
    const fe* roots = nullptr;
    SP_TRY(c_->ntt->roots((int)logN_, &roots));
    const uint32_t k = fri_layer_ - 1;  // layer being folded
    const uint64_t M = N_ >> k;
    const fe half = half_;                                           // (field inversions cost ~15 us on the host: none per layer)
    fe cst = fe_mul(fe_mul(zeta, half), fri_offset_inv_);
    // In evaluation form the fold is local to a rank: the partner i + M/2 of index i has the same residue mod G (§8(e) item 4)
    if (fri_sharded(k)) {
        const uint64_t Ml = M >> logG_;
        fe* next_local = fri_sharded(k + 1) ? d_fri_evals_[k + 1] : d_local_;
        SP_TRY(fri_fold(c_->stream, d_fri_evals_[k], next_local, Ml, logN_, k, roots, half, cst, logG_, rank_));
        if (!fri_sharded(k + 1)) {   // from here on the layers are small: gather this one once and continue on every rank
            SP_TRY(ensure_gather((uint64_t)world_ * (Ml >> 1)));
            SP_TRY(all_gather(next_local, d_gather_, (Ml >> 1) * sizeof(fe), true));
            SP_TRY(interleave_shards(c_->stream, d_gather_, d_fri_evals_[k + 1], (Ml >> 1), ShardMap{logG_, logG_, 0}));
        }
    } else {
        SP_TRY(fri_fold(c_->stream, d_fri_evals_[k], d_fri_evals_[k + 1], M, logN_, k, roots, half, cst));
    }
    fri_offset_ = fe_sqr(fri_offset_); fri_offset_inv_ = fe_sqr(fri_offset_inv_);
    if (k + 1 < logn_) {
        SP_TRY(commit_local(d_fri_evals_[k + 1], 0, 1, fri_trees_[k + 1].sub_leaves, LdeOrder{0, 0, 0}, fri_trees_[k + 1], root_out, true));
        fri_layer_ += 1;
        *is_last = 0;
    } else {
        // fri_last_value is coefficient 0 of the last folded polynomial (fri/mod.rs:58-67). Its degree is below the b
        // remaining evaluation points, so c_0 = (1/b) * sum of the evaluations on the coset (for a valid trace the
        // polynomial is constant and every evaluation already equals it).
        const uint32_t bb = 1u << logb_;
        std::vector<fe> ev(bb);
        SP_TRY(readback(ev.data(), d_fri_evals_[k + 1], sizeof(fe) * bb));
        fe sum = fe_zero();
        for (auto& e : ev) sum = fe_add(sum, e);
        *last_value = fe_mul(sum, binv_);
        *is_last = 1;
        stage_ = 7;
    }
    return SP_OK;
}

int StarkProver::fri_commit_chain(const fe& zeta0, const uint8_t state32[32], std::vector<std::array<uint8_t, 32>>& roots_out, fe* last_value) {
    if (!fri_chain_available()) { sp_set_error("fri_commit_chain: no layer committed yet, or the next layer to fold is sharded"); return SP_E_STATE; }
    SP_HIP_CHECK(hipSetDevice(c_->device));
    const fe* roots = nullptr;
    SP_TRY(c_->ntt->roots((int)logN_, &roots));
    const uint32_t L = logn_;
    if (!d_fri_chain_ || fri_chain_layers_ < L) {
        SP_TRY(alloc((void**)&d_fri_chain_, 32 + (size_t)L * 96));
        fri_chain_layers_ = L;
    }
    uint64_t* d_state = reinterpret_cast<uint64_t*>(d_fri_chain_);
    fe* d_cmul = reinterpret_cast<fe*>(d_fri_chain_ + 32);
    fe* d_cst = d_cmul + L;
    uint64_t* d_roots = reinterpret_cast<uint64_t*>(d_cst + L);
    // constants half / offset_k of every layer (offset_k = h^(2^k)) and the transcript state, in one upload
    std::vector<uint8_t>& up = h_up_fri_;     // (a member: the asynchronous copy below may still read it when an error path returns)
    up.assign(32 + (size_t)L * 32, 0);
    std::memcpy(up.data(), state32, 32);
    const uint32_t k0 = fri_layer_ - 1;           // the layer to fold first (0 on one GPU; the first replicated layer otherwise)
    fe oi = fri_offset_inv_;
    for (uint32_t k = k0; k < L; ++k) { const fe c = fe_mul(half_, oi); std::memcpy(up.data() + 32 + (size_t)k * 32, &c, 32); oi = fe_sqr(oi); }
    SP_HIP_CHECK(hipMemcpyAsync(d_fri_chain_, up.data(), up.size(), hipMemcpyHostToDevice, c_->stream));
    const fe cst0 = fe_mul(fe_mul(zeta0, half_), fri_offset_inv_);
    for (uint32_t k = k0; k < L; ++k) {           // fold layer k into layer k + 1 and commit it
        const uint64_t M = N_ >> k;
        const fe* c_dev = k == k0 ? nullptr : d_cst + k;   // zeta_k half / offset_k: left in device memory by the launch that produced root k
        const bool sharded_k = fri_sharded(k), sharded_k1 = k + 1 < L && fri_sharded(k + 1);
        static const bool fused = std::getenv("SP_FRI_NO_FUSED_LEAVES") == nullptr;      // (A/B switch)
        const bool fuse = fused && !sharded_k && k + 1 < L && merkle_hash(true) == MerkleHash::KECCAK256;
        if (sharded_k) {
            // several ranks, stream-ordered transport: the fold is local to a rank (the partner i + M/2 has the same residue mod G)
            const uint64_t Ml = M >> logG_;
            fe* next_local = sharded_k1 ? d_fri_evals_[k + 1] : d_local_;
            SP_TRY(fri_fold(c_->stream, d_fri_evals_[k], next_local, Ml, logN_, k, roots, half_, cst0, logG_, rank_, c_dev));
            if (!sharded_k1) {   // from here on the layers are small: gathered once, continued on every rank
                SP_TRY(ensure_gather((uint64_t)world_ * (Ml >> 1)));
                SP_TRY(all_gather(next_local, d_gather_, (Ml >> 1) * sizeof(fe), true));
                SP_TRY(interleave_shards(c_->stream, d_gather_, d_fri_evals_[k + 1], (Ml >> 1), ShardMap{logG_, logG_, 0}));
            }
        } else if (fuse) {
            SP_TRY(fri_fold_hash(c_->stream, d_fri_evals_[k], d_fri_evals_[k + 1], M, logN_, k, roots, half_, cst0, c_dev,
                                 fri_trees_[k + 1].sub + (fri_trees_[k + 1].sub_leaves - 1)));
        } else {
            SP_TRY(fri_fold(c_->stream, d_fri_evals_[k], d_fri_evals_[k + 1], M, logN_, k, roots, half_, cst0, 0, 0, c_dev));
        }
        if (k + 1 < L) {
            TreeBuf& t = fri_trees_[k + 1];
            const FriChallenge ch{d_state, d_cmul + (k + 1), d_cst + (k + 1), d_roots + 4 * (size_t)(k + 1)};
            if (sharded_k1) {
                SP_TRY(commit_local(d_fri_evals_[k + 1], 0, 1, t.sub_leaves, LdeOrder{0, 0, 0}, t, nullptr, true, &ch));
            } else {
                if (!fuse) SP_TRY(merkle_hash_leaves(c_->stream, d_fri_evals_[k + 1], 0, 1, t.sub_leaves, t.sub, LdeOrder{0, 0, 0}, merkle_hash(true)));
                SP_TRY(merkle_reduce(c_->stream, t.sub, t.sub_leaves, &ch, merkle_hash(true)));
            }
        }
    }
    for (uint32_t k = k0; k < L; ++k) { fri_offset_ = fe_sqr(fri_offset_); fri_offset_inv_ = fe_sqr(fri_offset_inv_); }
    // roots of layers k0 + 1 .. L-1 and the b evaluations of the last fold (fri/mod.rs:58-67, see fri_fold_commit)
    roots_out.assign(L - 1 - k0, std::array<uint8_t, 32>{});
    if (L - 1 > k0) {
        if ((size_t)(L - 1 - k0) * 32 > 4096) return SP_E_UNSUPPORTED;
        SP_TRY(readback(roots_out.data(), d_roots + 4 * (size_t)(k0 + 1), (size_t)(L - 1 - k0) * 32));
    }
    const uint32_t bb = 1u << logb_;
    std::vector<fe> ev(bb);
    SP_TRY(readback(ev.data(), d_fri_evals_[L], sizeof(fe) * bb));
    fe sum = fe_zero();
    for (auto& e : ev) sum = fe_add(sum, e);
    *last_value = fe_mul(sum, binv_);
    fri_layer_ = L;
    stage_ = 7;
    return SP_OK;
}

int StarkProver::grind(const uint8_t challenge[32], uint8_t factor, uint64_t* nonce_out) {
    SP_HIP_CHECK(hipSetDevice(c_->device));
    // expected number of trials 2^factor: ranges of about that size, four queued per host round trip (a range that starts
    // beyond an already found nonce returns at once)
    const uint64_t sub = 1ULL << std::min<uint32_t>(std::max<uint32_t>(factor, 16), 22);
    const uint64_t batch = 4 * sub;
    unsigned long long init = ~0ULL;
    SP_HIP_CHECK(hipMemcpyAsync(d_nonce_, &init, sizeof(init), hipMemcpyHostToDevice, c_->stream));
    for (uint64_t start = 0;; start += batch) {
        for (uint32_t u = 0; u < 4; ++u) SP_TRY(grind_range(c_->stream, challenge, factor, start + u * sub, sub, d_nonce_));
        unsigned long long r = 0;
        SP_TRY(readback(&r, d_nonce_, sizeof(r)));
        if (r != ~0ULL) { *nonce_out = r; return SP_OK; }
        if (start > (1ULL << 40)) { sp_set_error("grind: nonce not found"); return SP_E_UNSUPPORTED; }
    }
}

// fri_query_phase + open_deep_composition_poly (reference fri/mod.rs:74-127, prover.rs:484-529).  Every queried value and
// authentication path is gathered on the device into one staging block with the same layout on every rank; with several
// ranks the blocks are all-gathered once and the copy of the rank that owns the item is kept: LDE rows and sharded FRI values
// live on the rank with role index mod G, the lower levels of a sharded tree on the rank whose contiguous leaf range holds
// the index, the top log2 G levels everywhere.
int StarkProver::open(const std::vector<uint64_t>& iotas, Openings& o, bool values_canonical_be) {
    if (stage_ != 7) { sp_set_error("open: FRI commit phase not finished"); return SP_E_STATE; }
    SP_HIP_CHECK(hipSetDevice(c_->device));
    const uint32_t q = (uint32_t)iotas.size();
    if (q == 0 || q > 1024) return SP_E_INVALID_ARG;
    const uint32_t L = logn_, d0 = logN_;
    o.n_queries = q; o.n_layers = L; o.n_cols = C_; o.depth0 = d0; o.values_canonical_be = values_canonical_be;
    hipStream_t st = c_->stream;
    const LdeOrder ord = lde_order();
    struct TreeJob { const TreeBuf* t; std::vector<uint64_t> idx; size_t lower_off = 0, upper_off = 0, ipos = 0, iown = 0; uint32_t dl = 0, du = 0; };
    struct ValJob { const fe* base; uint64_t stride; uint32_t ncols; bool sharded; std::vector<uint64_t> idx, local; size_t off = 0, ipos = 0; };
    std::vector<TreeJob> tj;
    std::vector<ValJob> vj;
    std::vector<uint64_t> pos(q);
    for (uint32_t s = 0; s < q; ++s) pos[s] = iotas[s] % N_;
    auto local_row = [&](uint64_t i) -> uint64_t {   // position of LDE row i inside the coset-major columns of its owner
        return ord.at(i >> logG_);
    };
    {   // trace and composition rows
        ValJob t{d_lde_, Nl_, C_, G_ > 1, pos, {}}; ValJob c{d_h12_, Nl_, 2, G_ > 1, pos, {}};
        for (uint32_t s = 0; s < q; ++s) { uint64_t l = ((pos[s] & (G_ - 1)) == rank_) ? local_row(pos[s]) : 0; t.local.push_back(l); c.local.push_back(l); }
        vj.push_back(std::move(t)); vj.push_back(std::move(c));
    }
    tj.push_back(TreeJob{&tree_main_, pos});
    if (Ca_) tj.push_back(TreeJob{&tree_aux_, pos});
    tj.push_back(TreeJob{&tree_comp_, pos});
    for (uint32_t k = 0; k < L; ++k) {   // FRI layers: index iota mod |D_k| and its symmetric index
        const uint64_t M = N_ >> k;
        std::vector<uint64_t> idx(2 * (size_t)q);
        for (uint32_t s = 0; s < q; ++s) { idx[s] = iotas[s] % M; idx[q + s] = (iotas[s] + M / 2) % M; }
        ValJob v{d_fri_evals_[k], 0, 1, fri_sharded(k), idx, {}};
        for (uint64_t i : idx) v.local.push_back(fri_sharded(k) ? (((i & (G_ - 1)) == rank_) ? (i >> logG_) : 0) : i);
        vj.push_back(std::move(v));
        tj.push_back(TreeJob{&fri_trees_[k], idx});
    }
    // staging layout (32-byte items) and the index array (uint64)
    size_t items = 0, nidx = 0;
    for (auto& v : vj) { v.off = items; items += v.idx.size() * v.ncols; v.ipos = nidx; nidx += v.idx.size(); }
    const size_t value_items = items;        // the field elements come first in the staging block, the digests behind them
    for (auto& t : tj) {
        t.dl = (uint32_t)sp_log2_exact(t.t->sub_leaves); t.du = t.t->top == t.t->sub ? 0u : logG_;
        t.lower_off = items; items += t.idx.size() * t.dl;
        t.upper_off = items; items += t.idx.size() * t.du;
        t.ipos = nidx; nidx += t.idx.size();
        t.iown = nidx; if (t.du) nidx += t.idx.size();
    }
    std::vector<uint64_t>& hidx = h_idx_open_;   // (members, like `up` below: read by asynchronous copies)
    hidx.assign(nidx, 0);
    for (auto& v : vj) std::copy(v.local.begin(), v.local.end(), hidx.begin() + v.ipos);
    for (auto& t : tj)
        for (size_t s = 0; s < t.idx.size(); ++s) {
            hidx[t.ipos + s] = t.idx[s] & (t.t->sub_leaves - 1);
            if (t.du) hidx[t.iown + s] = t.idx[s] / t.t->sub_leaves;
        }
    // one job table for every array (values, lower and upper tree parts), uploaded with the indices: one copy, one launch
    std::vector<GatherJob> jobs;
    uint32_t max_items = 0;
    auto add_job = [&](const void* base, uint64_t stride_or_leaves, size_t idx_off, size_t out_off, size_t count, uint32_t width, uint32_t kind) {
        if (count == 0 || width == 0) return;
        jobs.push_back(GatherJob{base, stride_or_leaves, (uint64_t)idx_off, (uint64_t)out_off, (uint32_t)count, width, kind, 0u});
        max_items = std::max<uint32_t>(max_items, (uint32_t)count * width);
    };
    for (auto& v : vj) add_job(v.base, v.stride, v.ipos, v.off, v.idx.size(), v.ncols, 0);
    for (auto& t : tj) {
        add_job(t.t->sub, t.t->sub_leaves, t.ipos, t.lower_off, t.idx.size(), t.dl, 1);
        if (t.du) add_job(t.t->top, G_, t.iown, t.upper_off, t.idx.size(), t.du, 1);
    }
    const size_t idx_bytes = (nidx * sizeof(uint64_t) + 255) & ~size_t(255);
    const size_t job_bytes = (jobs.size() * sizeof(GatherJob) + 255) & ~size_t(255);
    const size_t blk_bytes = items * 32;
    const size_t need = idx_bytes + job_bytes + blk_bytes * (G_ > 1 ? 1 + (size_t)world_ : 1);
    struct Tmp { void* p = nullptr; ~Tmp() { if (p) (void)hipFree(p); } } tmp;   // many queries on a tiny domain: own staging buffer
    uint8_t* base = reinterpret_cast<uint8_t*>(d_scratch_);
    if (need > scratch_elems() * sizeof(fe)) {
        if (hipMalloc(&tmp.p, need) != hipSuccess) { sp_set_error("open: staging allocation failed"); return SP_E_ALLOC; }
        base = static_cast<uint8_t*>(tmp.p);
    }
    uint64_t* d_idx = reinterpret_cast<uint64_t*>(base);
    GatherJob* d_jobs = reinterpret_cast<GatherJob*>(base + idx_bytes);
    fe* blk = reinterpret_cast<fe*>(base + idx_bytes + job_bytes);
    fe* all_dev = blk + items;
    std::vector<uint8_t>& up = h_up_open_;
    up.assign(idx_bytes + job_bytes, 0);
    std::memcpy(up.data(), hidx.data(), nidx * sizeof(uint64_t));
    std::memcpy(up.data() + idx_bytes, jobs.data(), jobs.size() * sizeof(GatherJob));
    SP_HIP_CHECK(hipMemcpyAsync(base, up.data(), up.size(), hipMemcpyHostToDevice, st));
    SP_TRY(gather_jobs(st, d_jobs, (uint32_t)jobs.size(), max_items, d_idx, blk));
    if (values_canonical_be) SP_TRY(encode_elements(st, SP_FE_CANON_BE, blk, value_items, reinterpret_cast<uint8_t*>(blk)));   // in place, element by element
    // the download lands in a page-locked buffer kept across proofs (a pageable destination is staged by the runtime: ~2 MB per proof)
    const size_t host_items = items * (G_ > 1 ? world_ : 1);
    if (h_open_cap_ < host_items * sizeof(fe)) {
        if (h_open_pin_) (void)hipHostFree(h_open_pin_);
        h_open_pin_ = nullptr; h_open_cap_ = 0;
        const size_t cap = host_items * sizeof(fe) + (host_items * sizeof(fe)) / 4;
        if (hipHostMalloc(&h_open_pin_, cap, hipHostMallocDefault) != hipSuccess) { h_open_pin_ = nullptr; (void)hipGetLastError(); sp_set_error("open: page-locked download buffer: allocation failed"); return SP_E_ALLOC; }
        h_open_cap_ = cap;
    }
    const fe* host = static_cast<const fe*>(h_open_pin_);
    if (G_ > 1) {
        SP_TRY(all_gather(blk, all_dev, blk_bytes, true));
        SP_HIP_CHECK(hipMemcpyAsync(h_open_pin_, all_dev, host_items * sizeof(fe), hipMemcpyDeviceToHost, st));
    } else {
        SP_HIP_CHECK(hipMemcpyAsync(h_open_pin_, blk, host_items * sizeof(fe), hipMemcpyDeviceToHost, st));
    }
    SP_HIP_CHECK(sp_stream_wait_polling(st));   // (hidx is a local)
    auto slot = [&](uint32_t owner) -> const fe* { return host + (G_ > 1 ? (size_t)owner * items : 0); };
    auto take_values = [&](const ValJob& v, size_t s, fe* dst) {
        const uint32_t owner = v.sharded ? (uint32_t)(v.idx[s] & (G_ - 1)) : rank_;
        const fe* src = slot(owner) + v.off + s * v.ncols;
        std::copy(src, src + v.ncols, dst);
    };
    auto take_path = [&](const TreeJob& t, size_t s, digest32* dst) {
        const uint32_t owner = t.du ? (uint32_t)(t.idx[s] / t.t->sub_leaves) : rank_;
        std::memcpy(dst, slot(owner) + t.lower_off + s * t.dl, (size_t)t.dl * 32);
        if (t.du) std::memcpy(dst + t.dl, slot(rank_) + t.upper_off + s * t.du, (size_t)t.du * 32);
    };
    // (every entry is written below: no zero fill of arrays an Openings object reused across proofs already has at this size;
    // the auxiliary paths of an AIR without an auxiliary segment are the exception)
    o.trace_evals.resize((size_t)q * C_); o.comp_evals.resize((size_t)q * 2);
    o.main_paths.resize((size_t)q * d0); o.comp_paths.resize((size_t)q * d0);
    if (Ca_) o.aux_paths.resize((size_t)q * d0); else o.aux_paths.assign((size_t)q * d0, digest32{});
    size_t path_total = 0;
    for (uint32_t k = 0; k < L; ++k) path_total += d0 - k;
    o.fri_evals.resize((size_t)q * L); o.fri_evals_sym.resize((size_t)q * L);
    o.fri_paths.resize((size_t)q * path_total); o.fri_paths_sym.resize((size_t)q * path_total);
    size_t ti = 0;
    const TreeJob& jm = tj[ti++];
    const TreeJob* ja = Ca_ ? &tj[ti++] : nullptr;
    const TreeJob& jc = tj[ti++];
    for (uint32_t s = 0; s < q; ++s) {
        take_values(vj[0], s, &o.trace_evals[(size_t)s * C_]);
        take_values(vj[1], s, &o.comp_evals[(size_t)s * 2]);
        take_path(jm, s, &o.main_paths[(size_t)s * d0]);
        if (ja) take_path(*ja, s, &o.aux_paths[(size_t)s * d0]);
        take_path(jc, s, &o.comp_paths[(size_t)s * d0]);
    }
    size_t path_off = 0;
    for (uint32_t k = 0; k < L; ++k) {
        const uint32_t depth = d0 - k;
        const ValJob& v = vj[2 + k];
        const TreeJob& t = tj[ti + k];
        for (uint32_t s = 0; s < q; ++s) {
            take_values(v, s, &o.fri_evals[(size_t)s * L + k]);
            take_values(v, q + s, &o.fri_evals_sym[(size_t)s * L + k]);
            take_path(t, s, &o.fri_paths[(size_t)s * path_total + path_off]);
            take_path(t, q + s, &o.fri_paths_sym[(size_t)s * path_total + path_off]);
        }
        path_off += depth;
    }
    return SP_OK;
}

}  // namespace sp
