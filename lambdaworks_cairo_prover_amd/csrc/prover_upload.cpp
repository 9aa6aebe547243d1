// Host-buffer uploads of a trace segment (see prover.h): the row-major pipeline (host threads gather column groups into a ring of
// page-locked slots, DMA, rows -> columns, transforms of the groups overlapped) and the column path (one DMA per column group).
#include "prover_internal.h"
#include "cairo_host.h"
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>

namespace sp {

int StarkProver::ensure_upload(uint32_t groups) {
    if (groups > (uint32_t)UPLOAD_MAX_GROUPS) { sp_set_error("commit_trace: too many column groups"); return SP_E_UNSUPPORTED; }
    if (!copy_stream_) {
        // highest priority: the little kernels of the upload (rows -> columns, decode, the pull copy) must not queue behind the
        // thousands of work-groups of the transforms they feed (a kernel after every copy on an ordinary stream: 24 GB/s
        // instead of 56, tools/experiments/dma_pattern_probe.hip)
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        SP_HIP_CHECK(hipStreamCreateWithPriority(&copy_stream_, hipStreamNonBlocking, prio_hi));
        SP_HIP_CHECK(hipStreamCreateWithPriority(&r2c_stream_, hipStreamNonBlocking, prio_hi));
        for (int i = 0; i < UPLOAD_SLOTS; ++i) { SP_HIP_CHECK(hipEventCreateWithFlags(&ev_dma_[i], hipEventDisableTiming)); SP_HIP_CHECK(hipEventCreateWithFlags(&ev_r2c_[i], hipEventDisableTiming)); }
    }
    if (!up_start_) SP_HIP_CHECK(hipEventCreate(&up_start_));
    for (uint32_t g = 0; g < groups; ++g)
        for (hipEvent_t* e : {&up_ev_[g].dma0, &up_ev_[g].dma1, &up_ev_[g].ready, &up_ev_[g].done})
            if (!*e) SP_HIP_CHECK(hipEventCreate(e));
    SP_HIP_CHECK(hipStreamSynchronize(copy_stream_));   // (nothing pending unless an earlier call failed half-way)
    if (r2c_stream_) SP_HIP_CHECK(hipStreamSynchronize(r2c_stream_));
    return SP_OK;
}

// After the commitment's read-back (every event has completed): what the upload cost and how long the compute stream waited
// for it.  kind 1: gathered from a row-major host buffer, 2: DMA of host columns.
int StarkProver::finish_upload_stats(uint32_t groups, uint64_t bytes, double gather_ms, double host_ms, int kind) {
    double dma_ms = 0, exposed = 0, worst = 0;
    for (uint32_t g = 0; g < groups; ++g) {
        float d = 0, w = 0;
        if (hipEventElapsedTime(&d, up_ev_[g].dma0, up_ev_[g].dma1) == hipSuccess) dma_ms += d;
        // the compute stream could have started group g when it was done with group g - 1 (or, for the first one, at the start)
        if (hipEventElapsedTime(&w, g ? up_ev_[g - 1].done : up_start_, up_ev_[g].ready) == hipSuccess && w > 0) { exposed += w; worst = std::max<double>(worst, w); }
    }
    (void)hipGetLastError();
    double* u = c_->upload_stats;
    u[0] = kind; u[1] = groups; u[2] = (double)bytes; u[3] = gather_ms; u[4] = gather_ms > 0 ? bytes / gather_ms * 1e-6 : 0;
    u[5] = dma_ms; u[6] = dma_ms > 0 ? bytes / dma_ms * 1e-6 : 0; u[7] = exposed; u[8] = worst; u[9] = host_ms;
    return SP_OK;
}

// interpolate_and_commit (reference prover.rs:126-159) from host COLUMNS (the layout trace.rs:23-31 `cols()` produces, and what
// sp_cairo_run keeps): a column group is one contiguous DMA straight into the trace area - no gather, no landing slot - and the
// groups stay small (1, 1, 2, 2, ...) so that only the first column's 0.6 ms stay in front of the transforms.
int StarkProver::commit_trace_columns(int segment, const uint8_t* cols_host, uint32_t cols, int col_enc, uint64_t col_stride, uint8_t root_out[32]) {
    const uint32_t col0 = segment == 0 ? 0 : Cm_;
    std::vector<uint32_t> gsize;
    for (uint32_t done = 0; done < cols;) {
        // one, one, then two columns at a time: a group is ready when its DMA is (0.6 ms per column of 2^20 rows) and its
        // transforms take 0.7 ms per column, so with small groups the compute stream waits for the first column only; doubling
        // groups (1, 1, 2, 4, 8, 8, ...) made it wait 4 ms per proof - every group twice the size of the one being transformed
        uint32_t w = done < 2 ? 1u : 2u;
        if (G_ > 1 && d_cstage_) w = cols;            // column-sharded interpolation works on the whole segment
        w = std::min(w, cols - done);
        gsize.push_back(w);
        done += w;
    }
    const uint32_t groups = (uint32_t)gsize.size();
    SP_TRY(ensure_upload(groups));
    // page-locked source (sp_host_alloc, a run built after the context): DMA at PCIe speed; pageable: the runtime's staging copy.
    // SP_UPLOAD_PULL=1 replaces the DMA of a page-locked source by a copy kernel that reads it over PCIe (experiment).
    hipPointerAttribute_t attr{};
    const bool pinned = hipPointerGetAttributes(&attr, cols_host) == hipSuccess && attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    static const bool want_pull = std::getenv("SP_UPLOAD_PULL") != nullptr;
    const bool pull = pinned && want_pull && (reinterpret_cast<uintptr_t>(cols_host) % 16 == 0);
    const double t0 = wall_ms();
    fe* coeffs = d_coeffs_ + (uint64_t)col0 * n_;
    fe* trace = d_trace_ + (uint64_t)col0 * n_;
    fe* lde = d_lde_ + (uint64_t)col0 * Nl_;
    SP_HIP_CHECK(hipEventRecord(up_start_, c_->stream));             // the trace area's previous readers are behind this point
    SP_HIP_CHECK(hipStreamWaitEvent(copy_stream_, up_start_, 0));
    const bool sharded_interp = G_ > 1 && d_cstage_ && cols >= G_;
    uint32_t c0 = 0;
    for (uint32_t g = 0; g < groups; c0 += gsize[g], ++g) {
        const uint32_t w = gsize[g];
        fe* dst = trace + (uint64_t)c0 * n_;
        // One copy in flight at a time: the runtime picks the SDMA engine of a copy when it is ENQUEUED, and with the first engine
        // still busy it takes another one - copies of one stream hopping between engines ran at 27 - 37 GB/s instead of 56 for whole
        // proofs (profiles/r03_pinned_upload.txt: config #4 38 ms instead of 27).  The host has nothing else to do here.
        static const bool free_running = std::getenv("SP_UPLOAD_FREE_RUNNING") != nullptr;
        if (g >= 1 && !free_running) SP_HIP_CHECK(hipEventSynchronize(up_ev_[g - 1].dma1));
        SP_HIP_CHECK(hipEventRecord(up_ev_[g].dma0, copy_stream_));
        auto h2d = [&](void* to, const uint8_t* from, size_t bytes) -> int {
            if (pull) return pull_copy(copy_stream_, from, to, bytes);
            SP_HIP_CHECK(hipMemcpyAsync(to, from, bytes, hipMemcpyHostToDevice, copy_stream_));
            return SP_OK;
        };
        if (col_stride == n_) {
            SP_TRY(h2d(dst, cols_host + (size_t)c0 * n_ * 32, (size_t)w * n_ * 32));
        } else {
            for (uint32_t j = 0; j < w; ++j) SP_TRY(h2d(dst + (uint64_t)j * n_, cols_host + (size_t)(c0 + j) * col_stride * 32, (size_t)n_ * 32));
        }
        SP_HIP_CHECK(hipEventRecord(up_ev_[g].dma1, copy_stream_));
        if (col_enc >= 0) SP_TRY(decode_elements(copy_stream_, col_enc, reinterpret_cast<const uint8_t*>(dst), (uint64_t)w * n_, dst));   // element-wise, in place
        SP_HIP_CHECK(hipEventRecord(up_ev_[g].ready, copy_stream_));
        SP_HIP_CHECK(hipStreamWaitEvent(c_->stream, up_ev_[g].ready, 0));
        if (!sharded_interp) {
            // interpolate_fft + evaluate_offset_fft of this group (reference trace.rs:104-110, prover.rs:161-185)
            SP_TRY(c_->ntt->dif_natural_to_bitrev_inverse(coeffs + (uint64_t)c0 * n_, (int)logn_, w, n_, d_t1_, dst));
            SP_TRY(c_->ntt->lde_coset_major(coeffs + (uint64_t)c0 * n_, lde + (uint64_t)c0 * Nl_, (int)logn_, (int)logb_, w, n_, Nl_, (int)logG_, (int)rank_));
        }
        SP_HIP_CHECK(hipEventRecord(up_ev_[g].done, c_->stream));
    }
    const double host_ms = wall_ms() - t0;
    if (segment == 0) SP_TRY(launch_aux_presort());   // every column is behind this point of the compute stream
    int rc;
    if (sharded_interp) rc = commit_segment_resident(segment, cols, root_out);
    else {
        rc = commit_columns(lde, Nl_, cols, segment == 0 ? tree_main_ : tree_aux_, root_out);
        if (rc == SP_OK) stage_ = segment == 0 ? 2 : 3;
    }
    if (rc == SP_OK) {
        SP_TRY(finish_upload_stats(groups, (uint64_t)cols * n_ * 32, 0.0, host_ms, pinned ? 2 : 3));
    }
    return rc;
}

// A few parked host threads for the column gathers of the upload pipeline (creating them per group would put ~5 ms of
// pthread_create on the critical path of a proof).
class HostPool {
  public:
    explicit HostPool(unsigned workers) {
        for (unsigned w = 0; w < workers; ++w) threads_.emplace_back([this, w] { loop(w); });
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; gen_.fetch_add(1); }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    unsigned size() const { return (unsigned)threads_.size() + 1; }
    // Between begin_burst() and end_burst() idle workers spin on the generation counter instead of sleeping on the condition
    // variable: the groups of one upload follow each other within a millisecond and a futex wake-up of 30-60 threads costs
    // 50-100 us each time.
    void begin_burst() { { std::lock_guard<std::mutex> lk(m_); burst_.store(true, std::memory_order_release); } cv_.notify_all(); }
    void end_burst() { burst_.store(false, std::memory_order_release); }
    // runs job(part, parts) for part = 0 .. parts-1 (parts = workers + 1; the caller takes part 0) and waits for all of them
    void run(const std::function<void(unsigned, unsigned)>& job) {
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &job; pending_.store((unsigned)threads_.size(), std::memory_order_relaxed);
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        job(0, size());
        // the parts are equal: the others finish within microseconds of the caller
        for (int spin = 0; pending_.load(std::memory_order_acquire) != 0; ++spin) {
            if (spin < 20000) { sp_cpu_relax(); continue; }
            std::unique_lock<std::mutex> lk(m_);
            done_.wait(lk, [this] { return pending_.load(std::memory_order_acquire) == 0; });
        }
        job_ = nullptr;
    }
  private:
    static void sp_cpu_relax() { __builtin_ia32_pause(); }
    void loop(unsigned w) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(unsigned, unsigned)>* job;
            // (bounded: a worker that finds nothing for ~0.3 ms goes back to sleep, so a stalled upload does not burn the cores)
            for (int spin = 0; spin < 100000 && burst_.load(std::memory_order_acquire) && gen_.load(std::memory_order_acquire) == seen; ++spin) sp_cpu_relax();
            {
                std::unique_lock<std::mutex> lk(m_);
                if (gen_.load(std::memory_order_acquire) == seen) {
                    const bool was_burst = burst_.load(std::memory_order_acquire);
                    cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen || (!was_burst && burst_.load(std::memory_order_acquire)); });
                    if (gen_.load(std::memory_order_acquire) == seen) continue;   // woken into a burst: go spinning
                }
                seen = gen_.load(std::memory_order_acquire);
                if (stop_) return;
                job = job_;
            }
            (*job)(w + 1, size());
            if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) { std::lock_guard<std::mutex> lk(m_); done_.notify_one(); }
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(unsigned, unsigned)>* job_ = nullptr;
    std::atomic<unsigned> pending_{0};
    std::atomic<uint64_t> gen_{0};
    std::atomic<bool> burst_{false};
    bool stop_ = false;
};
void host_pool_delete(HostPool* p) { delete p; }

// dst[i][0..width) = src[i][off..off+width) for n rows of row_bytes: the column group of a row-major trace.  The rows are handed
// out in blocks through a shared counter, so a thread that is slow (a busy core, a remote NUMA node, a throttled container)
// takes fewer blocks instead of holding the whole group back.
static void host_gather_columns(HostPool& pool, const uint8_t* src, uint64_t n, size_t row_bytes, size_t off, size_t width, uint8_t* dst) {
    const uint64_t block = std::max<uint64_t>(256, (256u << 10) / width);   // ~256 KB written per block
    std::atomic<uint64_t> next{0};
    pool.run([&](unsigned, unsigned) {
        for (;;) {
            const uint64_t r0 = next.fetch_add(block, std::memory_order_relaxed);
            if (r0 >= n) return;
            const uint64_t r1 = std::min<uint64_t>(n, r0 + block);
            const uint8_t* s = src + r0 * row_bytes + off;
            uint8_t* d = dst + r0 * width;
            // fixed-size 32-byte copies inline as vector moves (a libc memcpy call per row costs more than the bytes it moves on
            // narrow groups)
            const size_t units = width / 32;
            for (uint64_t i = r0; i < r1; ++i, s += row_bytes, d += width)
                for (size_t u = 0; u < units; ++u) __builtin_memcpy(d + 32 * u, s + 32 * u, 32);
        }
    });
}

// interpolate_and_commit (reference prover.rs:126-159) from a row-major HOST buffer (the reference's TraceTable, trace.rs:9-13),
// in column groups: while group g is interpolated and extended on the compute stream, the chunks of the groups behind it are
// gathered out of the table into a small ring of page-locked slots by a few host threads and cross PCIe on a second stream.
// A chunk is a block of rows of one group, 32 MB at most: the DMA of one chunk runs beside the gather of the next whatever the
// size of the group, and the ring (4 x 32 MB) takes a sixth of the time to pin that three group-sized slots did (56 ms of a
// first proof at 2^20 rows).
int StarkProver::commit_trace_pipelined(int segment, const uint8_t* rows_host, uint32_t cols, uint8_t root_out[32]) {
    const uint32_t col0 = segment == 0 ? 0 : Cm_;
    // Column groups.  Group g can be transformed once it has crossed PCIe (~0.6 ms per column of 2^20 rows) and everything
    // behind it still has to be transformed (~0.7 ms per column; 0.2 ms at blowup 4, where the upload is the bound): two single
    // columns start the pipeline, then groups of two or four columns, always from an even column on (two columns share a 64-byte
    // line of a row).
    // How wide may a group get?  A group is usable when all of it has landed, so with the transforms as the bound (blowup 8: 0.70
    // ms per column against 0.65 ms of upload) narrow groups keep the compute stream fed - two columns: 75.0 ms where eight-column
    // groups gave 80 - and with the upload as the bound (blowup 4: 0.18 ms of transforms per column) the gather's throughput
    // decides, which grows with the width (two columns ~40 GB/s, four ~55, eight ~60): four columns.
    const double transform_ms_per_col = (1.0 + (double)(1u << logb_)) * (double)n_ * logn_ / 2 / 1.35e11 * 1e3;
    const double upload_ms_per_col = (double)n_ * 32 / 50e9 * 1e3;
    static const uint32_t maxw_env = [] { const char* e = std::getenv("SP_UPLOAD_MAXW"); return e ? (uint32_t)std::min(8, std::max(2, std::atoi(e))) : 0u; }();
    const uint32_t maxw = maxw_env ? maxw_env : (transform_ms_per_col >= 0.9 * upload_ms_per_col ? 2u : 4u);
    std::vector<uint32_t> gsize;
    for (uint32_t done = 0; done < cols;) {
        uint32_t w = done < 2 ? 1u : std::min<uint32_t>(maxw, std::max<uint32_t>(2, 2 * ((done + 1) / 2)));
        if (cols - done <= w + 1) w = cols - done;     // no one-column tail
        gsize.push_back(w);
        done += w;
    }
    const uint32_t groups = (uint32_t)gsize.size();
    // chunk size: 32 MB, less when the scratch area (the landing ring on the device) is small
    size_t chunk = std::min<size_t>((size_t)32 << 20, (scratch_elems() * sizeof(fe) / UPLOAD_SLOTS) & ~(size_t)4095);
    if (chunk < (size_t)64 * 256) { sp_set_error("commit_trace: scratch too small for the upload pipeline"); return SP_E_ALLOC; }
    double _tp = wall_ms();
    sp_ctx* ctx = c_;
    // gather threads: the option, but at most twice the CPUs this process can really have.  A cgroup quota counts CPU time per
    // 100 ms period and the upload is a burst of a quarter of a proof, so twice the quota's CPUs for that long stays inside it
    // (24 threads move 55-80 GB/s where the quota's own 14 move 40); far beyond it the whole process gets throttled - the 130 ms
    // proofs of a 64-thread gather in a 16-CPU container.
    if (!pool_) pool_ = new HostPool(std::max(2u, std::min(c_->opt_upload_threads, 2 * host_effective_cpus())) - 1u);
    SP_TRY(ensure_upload(groups));
    if (stage_bytes_ < chunk) {
        for (auto& p : h_stage_) { if (p) (void)hipHostFree(p); p = nullptr; }
        stage_bytes_ = 0;
        for (auto& p : h_stage_) if (hipHostMalloc(&p, chunk, hipHostMallocDefault) != hipSuccess) { sp_set_error("commit_trace: pinned staging allocation failed"); return SP_E_ALLOC; }
        stage_bytes_ = chunk;
    }
    uint8_t* landing[UPLOAD_SLOTS];
    for (int i = 0; i < UPLOAD_SLOTS; ++i) landing[i] = reinterpret_cast<uint8_t*>(d_scratch_) + (size_t)i * chunk;
    fe* coeffs = d_coeffs_ + (uint64_t)col0 * n_;
    fe* trace = d_trace_ + (uint64_t)col0 * n_;
    fe* lde = d_lde_ + (uint64_t)col0 * Nl_;
    SP_TIMEPOINT("  upload: threads, streams, pinned slots");
    const double t0 = wall_ms();
    double gather_ms = 0;
    struct Burst { HostPool* p; ~Burst() { p->end_burst(); } } burst{pool_};   // (also on the error paths)
    pool_->begin_burst();
    SP_HIP_CHECK(hipEventRecord(up_start_, c_->stream));             // the scratch and trace areas' previous users are behind this point
    SP_HIP_CHECK(hipStreamWaitEvent(copy_stream_, up_start_, 0));
    SP_HIP_CHECK(hipStreamWaitEvent(r2c_stream_, up_start_, 0));
    uint64_t chunk_no = 0;
    uint32_t c0 = 0;
    for (uint32_t g = 0; g < groups; c0 += gsize[g], ++g) {
        const uint32_t w = gsize[g];
        const double tg = wall_ms();
        double waited = 0;
        SP_HIP_CHECK(hipEventRecord(up_ev_[g].dma0, copy_stream_));
        {   // the group in blocks of rows: whole rows of the group (w x 32 contiguous bytes each: the wider, the better the gather
            // streams - 2 columns move ~40 GB/s, 8 columns ~58) and at most one ring slot of them at a time
            const uint32_t cw = w, c = c0;
            const uint64_t rows_per_chunk = std::max<uint64_t>(256, (chunk / ((size_t)cw * 32)) & ~(uint64_t)255);
            for (uint64_t r0 = 0; r0 < n_; r0 += rows_per_chunk, ++chunk_no) {
                const uint64_t rows = std::min<uint64_t>(rows_per_chunk, n_ - r0);
                const uint32_t slot = (uint32_t)(chunk_no % UPLOAD_SLOTS);
                const double tw = wall_ms();
                if (chunk_no >= UPLOAD_SLOTS) SP_HIP_CHECK(hipEventSynchronize(ev_dma_[slot]));   // the pinned slot has crossed PCIe
                waited += wall_ms() - tw;
                host_gather_columns(*pool_, rows_host + r0 * (size_t)cols * 32, rows, (size_t)cols * 32, (size_t)c * 32, (size_t)cw * 32,
                                    static_cast<uint8_t*>(h_stage_[slot]));
                // copy and rows -> columns both on the copy stream: the landing slot is free again as soon as the chunk has been
                // turned into columns, whatever the compute stream is busy with (queued behind the previous group's LDE the
                // upload stalled for ~3 ms twice per proof: profiles/r02_host_path_timeline.txt)
                if (chunk_no >= UPLOAD_SLOTS) SP_HIP_CHECK(hipStreamWaitEvent(copy_stream_, ev_r2c_[slot], 0));   // the landing slot has been turned into columns
                // (no "one copy in flight" wait here, unlike commit_trace_columns: a chunk's DMA is enqueued after a gather that took
                // about as long as the previous DMA, so the engine is mostly idle by then, and blocking this thread delays the next
                // gather - 38.5 against 35 ms at config #4 on a slow host)
                SP_HIP_CHECK(hipMemcpyAsync(landing[slot], h_stage_[slot], (size_t)rows * cw * 32, hipMemcpyHostToDevice, copy_stream_));
                SP_HIP_CHECK(hipEventRecord(ev_dma_[slot], copy_stream_));
                // rows -> columns on a stream of its own: on the copy stream the DMA engine sat idle through every one of these
                // kernels (~50 us x 34 chunks per proof); on the compute stream they queued behind the previous group's LDE
                SP_HIP_CHECK(hipStreamWaitEvent(r2c_stream_, ev_dma_[slot], 0));
                SP_TRY(rows_to_columns(r2c_stream_, c_->enc, landing[slot], rows, cw, trace + (uint64_t)c * n_ + r0, n_));
                SP_HIP_CHECK(hipEventRecord(ev_r2c_[slot], r2c_stream_));
            }
        }
        SP_HIP_CHECK(hipEventRecord(up_ev_[g].dma1, copy_stream_));
        SP_HIP_CHECK(hipEventRecord(up_ev_[g].ready, r2c_stream_));
        SP_HIP_CHECK(hipStreamWaitEvent(c_->stream, up_ev_[g].ready, 0));
        const double tge = wall_ms();
        gather_ms += tge - tg - waited;
        if (timing_enabled()) std::fprintf(stderr, "[sp_timing]   group %2u: %u columns, waited %.3f ms for slots, gather + enqueue %.3f ms (%.1f GB/s)\n", g, w, waited,
                                           tge - tg - waited, (double)n_ * w * 32 / (tge - tg - waited) * 1e-6);
        // interpolate_fft + evaluate_offset_fft of this group (reference trace.rs:104-110, prover.rs:161-185)
        SP_TRY(c_->ntt->dif_natural_to_bitrev_inverse(coeffs + (uint64_t)c0 * n_, (int)logn_, w, n_, d_t1_, trace + (uint64_t)c0 * n_));
        SP_TRY(c_->ntt->lde_coset_major(coeffs + (uint64_t)c0 * n_, lde + (uint64_t)c0 * Nl_, (int)logn_, (int)logb_, w, n_, Nl_, 0, 0));
        SP_HIP_CHECK(hipEventRecord(up_ev_[g].done, c_->stream));
    }
    const double host_ms = wall_ms() - t0;
    SP_TIMEPOINT("  upload + transforms of the groups");
    if (segment == 0) SP_TRY(launch_aux_presort());   // every group has been turned into columns behind this point of the compute stream
    SP_TIMEPOINT("  aux presort queued (+ its workspace)");
    SP_TRY(commit_columns(lde, Nl_, cols, segment == 0 ? tree_main_ : tree_aux_, root_out));
    SP_TIMEPOINT("  leaf hashing + tree");
    stage_ = segment == 0 ? 2 : 3;
    return finish_upload_stats(groups, (uint64_t)cols * n_ * 32, gather_ms, host_ms, 1);
}

}  // namespace sp
