// Stark252 NTT passes for gfx950. See ntt.h for the structure and DESIGN.md section 4.1 for the twiddle bookkeeping.
#include "ntt.h"
#include <algorithm>

namespace sp {

// ---------------------------------------------------------------------------------------------- device helpers
__device__ __forceinline__ fe ld_fe(const fe* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 lo = q[0], hi = q[1];
    fe r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
__device__ __forceinline__ void st_fe(fe* p, const fe& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}
// LDS tile: two planes of 16-byte halves so that consecutive lanes touch consecutive 16-byte slots.
__device__ __forceinline__ fe lds_ld(const uint4* lo, const uint4* hi, uint32_t i) {
    uint4 a = lo[i], b = hi[i];
    fe r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ void lds_st(uint4* lo, uint4* hi, uint32_t i, const fe& a) {
    lo[i] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    hi[i] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}
__device__ __forceinline__ uint32_t bitrev(uint32_t x, uint32_t bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

// One pass = r consecutive radix-2 stages of the plain in-place transform of size 2^logM on a tile of R = 2^r rows at
// stride 2^s (G adjacent elements per row).  Template flags are compile-time so each instantiation keeps only its own
// address math.
//  * DIT (Cooley-Tukey, bit-reversed in -> natural out): stage J = a.s + j pairs positions P and P + 2^(J-1) and computes
//    (u, v) -> (u + v w, u - v w) with w = w_(2^J)^(P mod 2^(J-1)).
//  * DIF (Gentleman-Sande, natural in -> bit-reversed out, inverse roots, unscaled): the same pairs in the opposite stage
//    order, (x, y) -> (x + y, (x - y) w^-1).  w^-1 = -w_(2^J)^(2^(J-1) - e), so the butterfly computes (y - x) times a
//    forward-table entry (and times -1 for e = 0): one table serves both directions.
//  * Twiddles: the s = 0 pass uses the pass-local table w_R^(+-e) (R/2 entries in LDS).  A strided pass stages the twiddles
//    of its own butterflies: w_M^(((i << a.s) + column) << (logM - a.s - j)), i = row mod 2^(j-1); the global column of
//    local column c is (c << shard_log) | shard_rank.  Stages j < r go to LDS (entry (2^(j-1) - 1 + i) * G + gl), the
//    twiddles of stage r serve one butterfly each and go straight to registers.  There is no inter-pass twiddle product.
//  * Deferred reduction (fp.h): p > 2^251, so every 256-bit value is < 32p.  Pass input < 2p.  DIT: t = v w < 2p and the
//    outputs u + t, u - t + 2p grow by 2p per stage - no correction for the <= 10 stages of a pass.  DIF: x + y doubles
//    the bound, (y - x + kb p) w is < 2p again; the sum is brought back below 2p every third stage (kb = 2, 4, 8).
//    Stores: fe_reduce_lazy_2p on every pass but the last of a transform (a.weak_out), fe_canonical_lazy on the last.
// Twiddles are always canonical, so every product is fe_mul_lazy(data, twiddle).
template <bool DIF, int LOADM, int STOREM, bool CONTIG, bool GTW>
__global__ void __launch_bounds__(NTT_THREADS) ntt_pass_kernel(NttPassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
#ifndef SP_NTT_PRIO
#define SP_NTT_PRIO 1
#endif
    // A fresh work-group competes with up to three computing ones for issue slots and, as the youngest, loses: raise its
    // priority until its tile and twiddle loads are on their way (SP_NTT_PRIO & 1: 34 x 2^22 10.95 -> 10.7 ms); doing the same
    // for the final stores (& 2) changes nothing (tools/sweep_ntt_variants.sh, profiles/r02_ntt_prio_sweep.txt)
    if (SP_NTT_PRIO & 1) __builtin_amdgcn_s_setprio(3);
    const uint32_t r = a.r, g = a.g;
    const uint32_t s = a.s;                // local (address) stride; the twiddles use the global stride s + a.tw_shift
    const uint32_t R = 1u << r, G = 1u << g, TILE = R << g;
    constexpr bool GLOBAL_TW = GTW;
    static_assert(GTW || CONTIG, "strided passes stage the twiddles of their own butterflies");
    uint4* Llo = smem;
    uint4* Lhi = smem + TILE;
    uint4* Twl = smem + 2 * TILE;
    uint4* Twh = Twl + (GLOBAL_TW ? (CONTIG ? R : (a.radix4 ? (TILE >> 2) : (TILE >> 1))) : (R >> 1));
    const uint32_t tid = threadIdx.x;
    // vector index fastest: consecutive work-groups run the same tile of different vectors, so the twiddles they gather
    // (the same table entries for every vector) are L2 hits for all but the first of them.
    // Work-groups are dealt round-robin to the 8 XCDs (each with its own L2): pin tile t to XCD t mod 8 and let that XCD
    // run the tile for all vectors back to back  (id = slot * 8 + t % 8, slot = (t / 8) * batch + vec).
    uint32_t tile, vec;
    if (a.xcd_map) {
        const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
        const uint32_t tg = slot / a.batch;
        vec = slot - tg * a.batch;
        tile = (tg << 3) | xcd;
    } else {
        tile = blockIdx.x / a.batch;
        vec = blockIdx.x - tile * a.batch;
    }
    // coset-major LDE: "vector" = (column v, local coset cl); every coset is its own array of 2^logL evaluations
    uint32_t tw_low = a.tw_low;
    const fe* src;
    fe* dst;
    if (a.coset_count) {
        const uint32_t v = vec / a.coset_count, cl = vec - v * a.coset_count;
        src = a.src + (uint64_t)v * a.src_vec_stride + (uint64_t)cl * a.src_coset_stride;
        dst = a.dst + (uint64_t)v * a.dst_vec_stride + (uint64_t)cl * a.dst_coset_stride;
        tw_low |= cl << a.shard_log;
    } else {
        src = a.src + (uint64_t)vec * a.src_vec_stride;
        dst = a.dst + (uint64_t)vec * a.dst_vec_stride;
    }
    const uint32_t logM = a.logM;          // transform size (twiddles)
    const uint32_t logL = a.logL;          // size of the array this pass addresses
    const uint32_t sg = s + a.tw_shift;    // global stride of this pass

    // tile coordinates
    uint32_t lo0 = 0, hi = 0;
    if (!CONTIG) {
        uint32_t lo_tiles = 1u << (s - g);
        lo0 = (tile & (lo_tiles - 1)) << g;
        hi = tile >> (s - g);
    }
    auto global_tw = [&](uint32_t jm1, uint32_t i, uint32_t gl) -> fe {
        const uint32_t col = ((lo0 + gl) << a.tw_shift) | tw_low;
        const uint32_t e = ((i << sg) + col) << (logM - sg - jm1 - 1u);
        if (!DIF) return ld_fe(a.big_tw + e);
        if (e == 0) return fe_neg_one();
        return ld_fe(a.big_tw + ((1u << (logM - 1)) - e));
    };
    constexpr int BPT = (1 << (NTT_TILE_LOG - 1)) / NTT_THREADS;   // butterflies per thread and stage (at most)
    fe tw_last[(GLOBAL_TW && !CONTIG) ? BPT : 1];
    fe tw_pen;   // strided passes in pairs: the stage-(r-1) twiddle of this thread's last unit
    if (GLOBAL_TW && CONTIG) {
        // contiguous tile: the twiddle depends on the row only, entry 2^(j-1) - 1 + i for stage j (R - 1 entries)
        for (uint32_t x = tid; x + 1 < R; x += NTT_THREADS) {
            const uint32_t y = x + 1u, jm1 = 31u - __clz(y);
            lds_st(Twl, Twh, x, global_tw(jm1, y - (1u << jm1), 0u));
        }
    } else if (GLOBAL_TW) {
        if (r >= 1) {
#pragma unroll
            for (int q = 0; q < BPT; ++q) {
                // stage-r butterflies of this thread: tid + q * threads, or - two stages per round trip (a.radix4) - the two
                // of its radix-4 unit, tid and tid + TILE/4
                const uint32_t b = tid + q * (a.radix4 ? (TILE >> 2) : (uint32_t)NTT_THREADS);
                if (b < (TILE >> 1)) tw_last[q] = global_tw(r - 1, b >> g, b & (G - 1));
            }
            // in pairs the twiddle of stage r - 1 also serves one unit only (row i = unit index): a register as well, and the
            // LDS table ends with stage r - 2 (40 KB per work-group instead of 48: four work-groups per CU)
            if (a.radix4 && r >= 2 && tid < (TILE >> 2)) tw_pen = global_tw(r - 2, tid >> g, tid & (G - 1));
            const uint32_t cnt = (a.radix4 ? (TILE >> 2) : (TILE >> 1)) - G;
            fe tmp[BPT];
#pragma unroll
            for (int q = 0; q < BPT; ++q) {
                const uint32_t x = tid + q * NTT_THREADS;
                if (x < cnt) {
                    const uint32_t y = (x >> g) + 1u;
                    const uint32_t jm1 = 31u - __clz(y);                 // j - 1
                    tmp[q] = global_tw(jm1, y - (1u << jm1), x & (G - 1));
                }
            }
#pragma unroll
            for (int q = 0; q < BPT; ++q) {
                const uint32_t x = tid + q * NTT_THREADS;
                if (x < cnt) lds_st(Twl, Twh, x, tmp[q]);
            }
        }
    } else {
        for (uint32_t i = tid; i < (R >> 1); i += NTT_THREADS) lds_st(Twl, Twh, i, ld_fe(a.small_tw + i));
    }
    // position (in the 2^logM working array) of tile element (t, gl)
    auto position = [&](uint32_t t, uint32_t gl) -> uint32_t {
        if (CONTIG) {
            if (LOADM == NTT_LOAD_GATHER_BITREV || STOREM == NTT_STORE_SCATTER_BITREV)
                return (bitrev((tile << g) + gl, logL - r) << r) + t;   // row whose bit-reversed index is adjacent
            return (tile << (r + g)) + (gl << r) + t;
        }
        return (hi << (s + r)) + (t << s) + lo0 + gl;
    };
    auto lidx = [&](uint32_t t, uint32_t gl) -> uint32_t { return CONTIG ? (gl << r) + t : (t << g) + gl; };

    // ------------------------------------------------------------------ load
    constexpr int LU = 4;   // loads in flight per thread
    for (uint32_t e0 = tid; e0 < TILE; e0 += NTT_THREADS * LU) {
        fe xs[LU];
        uint32_t li[LU];
#pragma unroll
        for (int q = 0; q < LU; ++q) {
            const uint32_t e = e0 + q * NTT_THREADS;
            if (e >= TILE) continue;
            uint32_t t, gl;
            if (CONTIG && LOADM == NTT_LOAD_INPLACE) {
                t = e & (R - 1); gl = e >> r;
                xs[q] = ld_fe(src + position(t, gl));
            } else if (CONTIG) {  // gather from the natural-order source: x[rev(pos)]
                gl = e & (G - 1); t = e >> g;
                uint32_t sidx = (bitrev(t, r) << (logL - r)) + (tile << g) + gl;
                xs[q] = ld_fe(src + sidx);
            } else {
                gl = e & (G - 1); t = e >> g;
                uint32_t pos = position(t, gl);
                if (LOADM == NTT_LOAD_EXPAND) xs[q] = ld_fe(src + (pos >> s));  // s = local log2(cosets held)
                else xs[q] = ld_fe(src + pos);
            }
            li[q] = lidx(t, gl);
        }
#pragma unroll
        for (int q = 0; q < LU; ++q)
            if (e0 + q * NTT_THREADS < TILE) lds_st(Llo, Lhi, li[q], xs[q]);
    }
    if (SP_NTT_PRIO & 1) __builtin_amdgcn_s_setprio(0);
    __syncthreads();

    // ------------------------------------------------------------------ radix-2 stages
    const uint32_t NB = TILE >> 1;
    uint32_t kb = 2;   // DIF: every element is < kb * p
    // DIT passes: after an odd leading stage, two stages per LDS round trip - a radix-4 unit of the four rows
    // t0 + {0, 1, 2, 3} * 2^(j-1) stays in registers between stage j and stage j + 1 (one unit per thread, TILE/4 units)
    const bool r4 = a.radix4 != 0;
    // DIF runs the stages downwards: pairs (r, r-1), (r-2, r-3), ... first, an odd stage 1 last
    if (DIF && r4) {
        for (uint32_t jt = r; jt >= 2; jt -= 2) {
            const uint32_t j = jt - 1, half = 1u << (j - 1);       // stages j + 1 (first) and j
            const uint32_t u = tid;
            const bool foldA = kb == 8;
            const uint32_t kbB = foldA ? 2u : 2u * kb;             // bound of the stage-(j+1) outputs
            const bool foldB = kbB == 8;
            if (u < (TILE >> 2)) {
                uint32_t gl, bq;
                if (CONTIG) { bq = u & ((R >> 2) - 1); gl = u >> (r - 2); } else { gl = u & (G - 1); bq = u >> g; }
                const uint32_t i = bq & (half - 1);
                const uint32_t t0 = ((bq >> (j - 1)) << (j + 1)) | i;
                const uint32_t l0 = lidx(t0, gl), l1 = lidx(t0 + half, gl), l2 = lidx(t0 + 2 * half, gl), l3 = lidx(t0 + 3 * half, gl);
                fe wa, wb, wc;          // stage j + 1: wb (rows 0, 2), wc (rows 1, 3); stage j: wa
                bool has_wa = true;
                if (GLOBAL_TW) {
                    if (j + 1 == r) { wb = tw_last[0]; wc = tw_last[BPT - 1]; wa = tw_pen; }
                    else { wb = lds_ld(Twl, Twh, ((2 * half - 1u + i) << g) + gl); wc = lds_ld(Twl, Twh, ((3 * half - 1u + i) << g) + gl);
                           wa = lds_ld(Twl, Twh, ((half - 1u + i) << g) + gl); }
                } else {                // pass-local table w_R^-e
                    wb = lds_ld(Twl, Twh, i << (r - j - 1)); wc = lds_ld(Twl, Twh, (i + half) << (r - j - 1));
                    if (j > 1) wa = lds_ld(Twl, Twh, i << (r - j)); else has_wa = false;
                }
                fe x0 = lds_ld(Llo, Lhi, l0), x1 = lds_ld(Llo, Lhi, l1), x2 = lds_ld(Llo, Lhi, l2), x3 = lds_ld(Llo, Lhi, l3);
                // stage j + 1: (x0, x2), (x1, x3)
                fe a0 = fe_add_raw(x0, x2), a1 = fe_add_raw(x1, x3);
                if (foldA) { a0 = fe_reduce_lazy_2p(a0); a1 = fe_reduce_lazy_2p(a1); }
                fe a2 = GLOBAL_TW ? fe_sub_add_kp(x2, x0, kb) : fe_sub_add_kp(x0, x2, kb);
                fe a3 = GLOBAL_TW ? fe_sub_add_kp(x3, x1, kb) : fe_sub_add_kp(x1, x3, kb);
                a2 = fe_mul_lazy(a2, wb); a3 = fe_mul_lazy(a3, wc);
                // stage j: (a0, a1), (a2, a3); every input is < kbB p
                fe s0 = fe_add_raw(a0, a1), s2 = fe_add_raw(a2, a3);
                if (foldB) { s0 = fe_reduce_lazy_2p(s0); s2 = fe_reduce_lazy_2p(s2); }
                fe d1 = GLOBAL_TW ? fe_sub_add_kp(a1, a0, kbB) : fe_sub_add_kp(a0, a1, kbB);
                fe d3 = GLOBAL_TW ? fe_sub_add_kp(a3, a2, kbB) : fe_sub_add_kp(a2, a3, kbB);
                if (has_wa) { d1 = fe_mul_lazy(d1, wa); d3 = fe_mul_lazy(d3, wa); }
                lds_st(Llo, Lhi, l0, s0); lds_st(Llo, Lhi, l1, d1); lds_st(Llo, Lhi, l2, s2); lds_st(Llo, Lhi, l3, d3);
            }
            kb = foldB ? 2u : 2u * kbB;
            __syncthreads();
        }
    }
    // radix-2 stages: all of them without a.radix4; with it the odd leading stage of a DIT pass / the odd last stage (1) of DIF
    const uint32_t singles = r4 ? (r & 1u) : r;
    uint32_t jj = 1;
    for (; jj <= singles; ++jj) {
        const uint32_t j = (DIF && !r4) ? r + 1 - jj : jj;     // (DIF with pairs: only stage 1 is left)
        const uint32_t half = 1u << (j - 1);
        const bool fold = DIF && kb == 8;   // the sums of this stage would reach 16p: bring them back below 2p
#pragma unroll
        for (int q = 0; q < BPT; ++q) {
            const uint32_t b = tid + q * NTT_THREADS;
            if (b >= NB) continue;
            uint32_t bf, gl;
            if (CONTIG) { bf = b & ((R >> 1) - 1); gl = b >> (r - 1); } else { gl = b & (G - 1); bf = b >> g; }
            const uint32_t i = bf & (half - 1);
            const uint32_t t0 = 2u * bf - i;                    // ((bf >> (j-1)) << j) | i
            const uint32_t i0 = lidx(t0, gl), i1 = lidx(t0 + half, gl);
            fe w;
            bool has_w = true;
            if (GLOBAL_TW && CONTIG) w = lds_ld(Twl, Twh, half - 1u + i);
            else if (GLOBAL_TW) { if (j == r) w = tw_last[q]; else w = lds_ld(Twl, Twh, ((half - 1u + i) << g) + gl); }
            else if (j > 1) w = lds_ld(Twl, Twh, i << (r - j));
            else has_w = false;                                 // w_2^0 = 1
            fe x = lds_ld(Llo, Lhi, i0), y = lds_ld(Llo, Lhi, i1);
            if (!DIF) {
                if (has_w) y = fe_mul_lazy(y, w);               // stage 1 of the s = 0 pass: y < 2p already
                lds_st(Llo, Lhi, i0, fe_add_raw(x, y));
                lds_st(Llo, Lhi, i1, fe_sub_add_2p(x, y));
            } else {
                fe sum = fe_add_raw(x, y);
                if (fold) sum = fe_reduce_lazy_2p(sum);
                fe d = GLOBAL_TW ? fe_sub_add_kp(y, x, kb) : fe_sub_add_kp(x, y, kb);
                if (has_w) d = fe_mul_lazy(d, w);
                lds_st(Llo, Lhi, i0, sum);
                lds_st(Llo, Lhi, i1, d);
            }
        }
        kb = fold ? 2u : 2u * kb;
        __syncthreads();
    }
    if (!DIF) {
        for (; jj + 1 <= r; jj += 2) {
            const uint32_t j = jj, half = 1u << (j - 1);
            const uint32_t u = tid;
            if (u < (TILE >> 2)) {
                uint32_t gl, bq;
                if (CONTIG) { bq = u & ((R >> 2) - 1); gl = u >> (r - 2); } else { gl = u & (G - 1); bq = u >> g; }
                const uint32_t i = bq & (half - 1);
                const uint32_t t0 = ((bq >> (j - 1)) << (j + 1)) | i;
                const uint32_t l0 = lidx(t0, gl), l1 = lidx(t0 + half, gl), l2 = lidx(t0 + 2 * half, gl), l3 = lidx(t0 + 3 * half, gl);
                fe wa, wb, wc;
                bool has_wa = true;
                if (GLOBAL_TW && CONTIG) {
                    wa = lds_ld(Twl, Twh, half - 1u + i); wb = lds_ld(Twl, Twh, 2 * half - 1u + i); wc = lds_ld(Twl, Twh, 3 * half - 1u + i);
                } else if (GLOBAL_TW) {
                    if (j + 1 == r) { wa = tw_pen; wb = tw_last[0]; wc = tw_last[BPT - 1]; }
                    else { wa = lds_ld(Twl, Twh, ((half - 1u + i) << g) + gl);
                           wb = lds_ld(Twl, Twh, ((2 * half - 1u + i) << g) + gl); wc = lds_ld(Twl, Twh, ((3 * half - 1u + i) << g) + gl); }
                } else {   // pass-local table w_R^e
                    if (j > 1) wa = lds_ld(Twl, Twh, i << (r - j)); else has_wa = false;   // stage 1: w = 1, inputs < 2p
                    if (j > 1) wb = lds_ld(Twl, Twh, i << (r - j - 1));                    // stages 1, 2: i = 0, so w_4^0 = 1 as well
                    wc = lds_ld(Twl, Twh, (i + half) << (r - j - 1));
                }
                fe x0 = lds_ld(Llo, Lhi, l0), x1 = lds_ld(Llo, Lhi, l1), x2 = lds_ld(Llo, Lhi, l2), x3 = lds_ld(Llo, Lhi, l3);
                if (has_wa) { x1 = fe_mul_lazy(x1, wa); x3 = fe_mul_lazy(x3, wa); }
                fe a0 = fe_add_raw(x0, x1), a1 = fe_sub_add_2p(x0, x1), a2 = fe_add_raw(x2, x3), a3 = fe_sub_add_2p(x2, x3);
                a3 = fe_mul_lazy(a3, wc);
                fe y0, y2;
                if (has_wa) { a2 = fe_mul_lazy(a2, wb); y0 = fe_add_raw(a0, a2); y2 = fe_sub_add_2p(a0, a2); }
                else { y0 = fe_add_raw(a0, a2); y2 = fe_sub_add_kp(a0, a2, 4u); }   // a2 = x2 + x3 < 4p unmultiplied: bias 4p, outputs < 8p
                                                                                    // (2p above the usual bound: 20p after ten stages, still < 32p)
                const fe y1 = fe_add_raw(a1, a3), y3 = fe_sub_add_2p(a1, a3);
                if (j + 1 == r && a.radix4 == 1) {
                    if (SP_NTT_PRIO & 2) __builtin_amdgcn_s_setprio(3);
                    // last pair of the pass: straight to global memory (no LDS write, barrier and re-read)
                    const uint32_t q = half;   // = R/4
                    st_fe(dst + position(t0, gl), a.weak_out ? fe_reduce_lazy_2p(y0) : fe_canonical_lazy(y0));
                    st_fe(dst + position(t0 + q, gl), a.weak_out ? fe_reduce_lazy_2p(y1) : fe_canonical_lazy(y1));
                    st_fe(dst + position(t0 + 2 * q, gl), a.weak_out ? fe_reduce_lazy_2p(y2) : fe_canonical_lazy(y2));
                    st_fe(dst + position(t0 + 3 * q, gl), a.weak_out ? fe_reduce_lazy_2p(y3) : fe_canonical_lazy(y3));
                } else {
                    lds_st(Llo, Lhi, l0, y0); lds_st(Llo, Lhi, l2, y2); lds_st(Llo, Lhi, l1, y1); lds_st(Llo, Lhi, l3, y3);
                }
            }
            if (j + 1 != r || a.radix4 != 1) __syncthreads();
        }
        if (a.radix4 == 1) return;   // the last pair has written the tile
    }

    // ------------------------------------------------------------------ store
    if (SP_NTT_PRIO & 2) __builtin_amdgcn_s_setprio(3);
    fe scal;
    const bool has_scalar = a.scalar != nullptr;
    if (has_scalar) scal = ld_fe(a.scalar);
    for (uint32_t e = tid; e < TILE; e += NTT_THREADS) {
        uint32_t t, gl;
        if (CONTIG && STOREM == NTT_STORE_INPLACE) { t = e & (R - 1); gl = e >> r; }
        else { gl = e & (G - 1); t = e >> g; }
        fe x = lds_ld(Llo, Lhi, lidx(t, gl));
        uint32_t pos = position(t, gl);
        uint32_t didx = pos;
        if (CONTIG && STOREM == NTT_STORE_SCATTER_BITREV) didx = (bitrev(t, r) << (logL - r)) + (tile << g) + gl;
        if (DIF && a.post_table) x = fe_mul_lazy(x, ld_fe(a.post_table + didx));
        if (has_scalar) x = fe_mul_lazy(x, scal);
        x = a.weak_out ? fe_reduce_lazy_2p(x) : fe_canonical_lazy(x);
        st_fe(dst + didx, x);
    }
}

// table[e] = w^e for e < count, from w^(2^i) (i < bits)
struct RootGenArgs { fe pw[32]; };
__global__ void gen_roots_kernel(fe* table, uint32_t count, uint32_t bits, RootGenArgs args) {
    uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    fe acc = fe_one();
    for (uint32_t i = 0; i < bits; ++i)
        if ((e >> i) & 1) acc = fe_mul(acc, args.pw[i]);
    st_fe(table + e, acc);
}

// data[i] *= base^i * c
struct PowArgs { fe pw[40]; fe c; uint32_t has_c; };
__global__ void scale_powers_kernel(fe* data, uint64_t n, uint64_t stride, PowArgs args) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe acc = args.has_c ? args.c : fe_one();
    for (uint32_t b = 0; b < 40; ++b)
        if ((i >> b) & 1) acc = fe_mul(acc, args.pw[b]);
    fe* p = data + (uint64_t)blockIdx.y * stride + i;
    st_fe(p, fe_mul(ld_fe(p), acc));
}

// ---------------------------------------------------------------------------------------------- host side
static const uint8_t TWO_ADIC_ROOT_BE[32] = {0x00, 0x52, 0x82, 0xdb, 0x87, 0x52, 0x9c, 0xfa, 0x3f, 0x04, 0x64, 0x51, 0x9c, 0x8b, 0x0f, 0xa5,
                                             0xad, 0x18, 0x71, 0x48, 0xe1, 0x1a, 0x61, 0x61, 0x60, 0x70, 0x02, 0x4f, 0x42, 0xf8, 0xef, 0x94};

fe host_primitive_root(int k) {
    fe w = fe_from_bytes_be(TWO_ADIC_ROOT_BE);  // order 2^192
    for (int i = k; i < 192; ++i) w = fe_sqr(w);
    return w;
}

NttEngine::~NttEngine() {
    for (auto& kv : roots_) (void)hipFree(kv.second);
    for (auto& kv : inv_small_) (void)hipFree(kv.second);
    if (d_scalar_) (void)hipFree(d_scalar_);
}

static int gen_table(hipStream_t st, fe w, int bits, uint32_t count, fe** out) {
    fe* d = nullptr;
    SP_HIP_CHECK(hipMalloc(&d, sizeof(fe) * (size_t)std::max<uint32_t>(count, 1)));
    RootGenArgs args;
    fe cur = w;
    for (int i = 0; i < 32; ++i) { args.pw[i] = cur; cur = fe_sqr(cur); }
    uint32_t blocks = (count + 255) / 256;
    hipLaunchKernelGGL(gen_roots_kernel, dim3(blocks), dim3(256), 0, st, d, count, (uint32_t)bits, args);
    SP_HIP_CHECK(hipGetLastError());
    *out = d;
    return SP_OK;
}

int NttEngine::roots(int k, const fe** out) {
    auto it = roots_.find(k);
    if (it == roots_.end()) {
        fe* d = nullptr;
        uint32_t count = k == 0 ? 1u : (1u << (k - 1));
        SP_TRY(gen_table(stream_, host_primitive_root(k), k, count, &d));
        it = roots_.emplace(k, d).first;
    }
    *out = it->second;
    return SP_OK;
}
int NttEngine::inv_roots_small(int k, const fe** out) {
    auto it = inv_small_.find(k);
    if (it == inv_small_.end()) {
        fe* d = nullptr;
        uint32_t count = k == 0 ? 1u : (1u << (k - 1));
        SP_TRY(gen_table(stream_, fe_inv(host_primitive_root(k)), k, count, &d));
        it = inv_small_.emplace(k, d).first;
    }
    *out = it->second;
    return SP_OK;
}

template <bool DIF, int LM, int SM, bool CONTIG, bool GTW>
static int launch_t(hipStream_t st, const NttPassArgs& a, uint32_t batch) {
    uint32_t tile_log = a.r + a.g;
    uint32_t tiles = 1u << (a.logL - tile_log);
    size_t lds = ((size_t)2 << tile_log) * sizeof(uint4) + ((size_t)1 << a.r) * sizeof(uint4);
    if (!CONTIG) lds = ((size_t)3 << tile_log) * sizeof(uint4);   // tile + the global twiddles of its stages j < r (j < r - 1 in pairs, below)
    else if (GTW) lds = ((size_t)2 << tile_log) * sizeof(uint4) + ((size_t)2 << a.r) * sizeof(uint4);
    NttPassArgs b = a;
    b.batch = batch;
    b.xcd_map = (tiles % 8 == 0) ? 1u : 0u;
    // two stages per LDS round trip on the tiles that give every thread a unit (a half-empty work-group loses more than the
    // saved round trips)
    b.radix4 = (a.r >= 2 && (1u << tile_log) == 4u * NTT_THREADS) ? 1u : 0u;
    if (!CONTIG && b.radix4) lds = ((size_t)5 << (tile_log - 1)) * sizeof(uint4);
    if ((uint64_t)tiles * batch >= (1ull << 31)) { sp_set_error("ntt: batch too large for one launch"); return SP_E_INVALID_ARG; }
    hipLaunchKernelGGL((ntt_pass_kernel<DIF, LM, SM, CONTIG, GTW>), dim3(tiles * batch), dim3(NTT_THREADS), lds, st, b);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

int NttEngine::launch_pass(bool dif, int lm, int sm, const NttPassArgs& a, uint32_t batch) {
    bool contig = a.s == 0 && lm != NTT_LOAD_EXPAND;
    if (!dif) {
        if (contig && a.coset_count) return launch_t<false, NTT_LOAD_INPLACE, NTT_STORE_INPLACE, true, true>(stream_, a, batch);
        if (contig && lm == NTT_LOAD_INPLACE) return launch_t<false, NTT_LOAD_INPLACE, NTT_STORE_INPLACE, true, false>(stream_, a, batch);
        if (contig && lm == NTT_LOAD_GATHER_BITREV) return launch_t<false, NTT_LOAD_GATHER_BITREV, NTT_STORE_INPLACE, true, false>(stream_, a, batch);
        if (!contig && lm == NTT_LOAD_INPLACE) return launch_t<false, NTT_LOAD_INPLACE, NTT_STORE_INPLACE, false, true>(stream_, a, batch);
        if (!contig && lm == NTT_LOAD_EXPAND) return launch_t<false, NTT_LOAD_EXPAND, NTT_STORE_INPLACE, false, true>(stream_, a, batch);
    } else {
        if (contig && sm == NTT_STORE_INPLACE) return launch_t<true, NTT_LOAD_INPLACE, NTT_STORE_INPLACE, true, false>(stream_, a, batch);
        if (contig && sm == NTT_STORE_SCATTER_BITREV) return launch_t<true, NTT_LOAD_INPLACE, NTT_STORE_SCATTER_BITREV, true, false>(stream_, a, batch);
        if (!contig) return launch_t<true, NTT_LOAD_INPLACE, NTT_STORE_INPLACE, false, true>(stream_, a, batch);
    }
    sp_set_error("ntt: unsupported pass mode");
    return SP_E_UNSUPPORTED;
}

// Builds the pass list for a size-2^k transform. first_contig_max limits the s = 0 pass (gather/scatter passes
// want G >= 4 rows per tile so that their strided side is coalesced).
struct PassGeom { int s, r, g; };
// footprint = bytes the whole batched transform touches.  1024-element tiles (256-byte rows; the only ones that run their
// stages in pairs) for everything beyond 64 MB, 512-element tiles (more work-groups in flight) for small transforms.
// Measured with the pair kernels: single 2^22 transform 0.383 -> 0.371 ms with 1024, single 2^20 0.114 -> 0.117 ms.
static std::vector<PassGeom> geometry(int k, int first_stride_log, int first_contig_max, uint64_t footprint) {
    const int min_tile_log = footprint > (64ull << 20) ? 10 : 9;
    std::vector<PassGeom> out;
    int s = first_stride_log;
    int rem = k - first_stride_log;
    if (rem <= 0) return out;
    if (s == 0) {
        int r1 = std::min(rem, first_contig_max);
        int g = std::min(NTT_TILE_LOG - r1, k - r1);
        out.push_back({0, r1, g});
        s = r1; rem -= r1;
    }
    if (rem > 0) {
        int np = (rem + NTT_MAX_STRIDED_LOG - 1) / NTT_MAX_STRIDED_LOG;
        std::vector<int> takes;
        for (int i = 0, left = rem; i < np; ++i) { int take = (left + (np - i) - 1) / (np - i); takes.push_back(take); left -= take; }
#ifndef SP_NTT_EVEN_SPLIT
#define SP_NTT_EVEN_SPLIT 1
#endif
        // a pass with an odd number of stages runs one of them as a single radix-2 stage - a whole LDS round trip for one stage
        // instead of two: trade stages between two odd passes where the tile allows it (14 = 8 + 6 instead of 7 + 7)
        if (SP_NTT_EVEN_SPLIT)
            for (size_t i = 0; i < takes.size(); ++i)
                for (size_t j = i + 1; j < takes.size(); ++j)
                    if ((takes[i] & 1) && (takes[j] & 1) && takes[i] + 1 <= NTT_MAX_STRIDED_LOG && takes[j] >= 3) { takes[i] += 1; takes[j] -= 1; }
        for (int take : takes) {
            // adjacent elements per row: at least 4 (128 B), more for short passes so that a tile never has fewer
            // than 2^min_tile_log elements (a 2^5-row pass with 4 columns would leave half the work-group idle)
            int g = std::min(std::max(NTT_STRIDED_G_LOG, min_tile_log - take), s);
            out.push_back({s, take, g});
            s += take;
        }
    }
    return out;
}

int NttEngine::dit_bitrev_to_natural(fe* data, int k, uint32_t batch, uint64_t stride) {
    if (k == 0) return SP_OK;
    const fe* big = nullptr;
    SP_TRY(roots(k, &big));
    std::vector<PassGeom> geo = geometry(k, 0, NTT_MAX_CONTIG_LOG, ((uint64_t)batch << k) * sizeof(fe));
    for (size_t i = 0; i < geo.size(); ++i) {
        const PassGeom& p = geo[i];
        NttPassArgs a{};
        a.src = data; a.dst = data; a.src_vec_stride = a.dst_vec_stride = stride;
        SP_TRY(roots(p.r, &a.small_tw));
        a.big_tw = big; a.logM = k; a.logL = k; a.s = p.s; a.r = p.r; a.g = p.g;
        a.weak_out = i + 1 < geo.size();
        SP_TRY(launch_pass(false, NTT_LOAD_INPLACE, NTT_STORE_INPLACE, a, batch));
    }
    return SP_OK;
}

int NttEngine::dif_natural_to_bitrev_inverse(fe* data, int k, uint32_t batch, uint64_t stride, const fe* post_table, const fe* src) {
    const fe* big = nullptr;
    SP_TRY(roots(k, &big));
    // beyond one tile the contiguous pass is kept short (2^7 rows x 8 contiguous runs): a 2^10-row pass stages a 16 KB
    // twiddle table per 32 KB tile and fits only three work-groups per CU (measured 55 % of the strided passes' rate)
    constexpr int contig_cap = 7;
    std::vector<PassGeom> geo = geometry(k, 0, k <= NTT_TILE_LOG ? NTT_TILE_LOG : contig_cap, ((uint64_t)batch << k) * sizeof(fe));
    if (geo.empty()) geo.push_back({0, 0, 0});
    for (size_t i = geo.size(); i-- > 0;) {
        const PassGeom& p = geo[i];
        NttPassArgs a{};
        a.src = (src && i + 1 == geo.size()) ? src : data;  // the first pass executed may read another array (same stride)
        a.dst = data; a.src_vec_stride = a.dst_vec_stride = stride;
        SP_TRY(inv_roots_small(p.r, &a.small_tw));
        a.big_tw = big; a.logM = k; a.logL = k; a.s = p.s; a.r = p.r; a.g = p.g;
        a.weak_out = i != 0;
        if (i == 0) a.post_table = post_table;
        SP_TRY(launch_pass(true, NTT_LOAD_INPLACE, NTT_STORE_INPLACE, a, batch));
    }
    return SP_OK;
}

int NttEngine::forward_natural(const fe* src, fe* dst, int k, uint32_t batch, uint64_t ss, uint64_t ds, fe* final_dst) {
    const fe* big = nullptr;
    SP_TRY(roots(k, &big));
    std::vector<PassGeom> geo = geometry(k, 0, k <= NTT_TILE_LOG ? NTT_TILE_LOG : NTT_TILE_LOG - NTT_STRIDED_G_LOG, ((uint64_t)batch << k) * sizeof(fe));
    if (geo.empty()) geo.push_back({0, 0, 0});
    bool first = true;
    // ping-pong: pass 1 src -> dst (gather, must be out of place); if final_dst is given (same stride as src), pass 2
    // writes dst -> final_dst and later passes run in place there, so the result lands in final_dst without a copy.
    const fe* cur = src; uint64_t cur_stride = ss;
    size_t pi = 0;
    for (const PassGeom& p : geo) {
        NttPassArgs a{};
        fe* out = dst; uint64_t out_stride = ds;
        if (final_dst && pi >= 1) { out = final_dst; out_stride = ss; }
        a.src = cur; a.dst = out;
        a.src_vec_stride = cur_stride; a.dst_vec_stride = out_stride;
        cur = out; cur_stride = out_stride; ++pi;
        SP_TRY(roots(p.r, &a.small_tw));
        a.big_tw = big; a.logM = k; a.logL = k; a.s = p.s; a.r = p.r; a.g = p.g;
        a.weak_out = pi < geo.size();
        SP_TRY(launch_pass(false, first ? NTT_LOAD_GATHER_BITREV : NTT_LOAD_INPLACE, NTT_STORE_INPLACE, a, batch));
        first = false;
    }
    if (final_dst && geo.size() == 1)
        for (uint32_t v = 0; v < batch; ++v)
            SP_HIP_CHECK(hipMemcpyAsync(final_dst + v * ss, dst + v * ds, sizeof(fe) << k, hipMemcpyDeviceToDevice, stream_));
    return SP_OK;
}

int NttEngine::inverse_natural(fe* data, fe* tmp, int k, uint32_t batch, uint64_t stride) {
    // Un-passes run last-to-first: the first one goes data -> tmp, the middle ones stay in tmp, and the final
    // bit-reversal scatter goes tmp -> data (a scatter must never run in place: other tiles still read their rows).
    const fe* big = nullptr;
    SP_TRY(roots(k, &big));
    std::vector<PassGeom> geo = geometry(k, 0, k <= NTT_TILE_LOG ? NTT_TILE_LOG : NTT_TILE_LOG - NTT_STRIDED_G_LOG, ((uint64_t)batch << k) * sizeof(fe));
    if (geo.empty()) geo.push_back({0, 0, 0});
    if (!d_scalar_) SP_HIP_CHECK(hipMalloc(&d_scalar_, sizeof(fe)));
    fe ninv = fe_inv(fe_from_u64(1ULL << k));
    SP_HIP_CHECK(hipMemcpyAsync(d_scalar_, &ninv, sizeof(fe), hipMemcpyHostToDevice, stream_));
    if (geo.size() == 1) {
        for (uint32_t v = 0; v < batch; ++v)
            SP_HIP_CHECK(hipMemcpyAsync(tmp + v * stride, data + v * stride, sizeof(fe) << k, hipMemcpyDeviceToDevice, stream_));
    }
    for (size_t i = geo.size(); i-- > 0;) {
        const PassGeom& p = geo[i];
        bool firstpass = (i + 1 == geo.size()) && geo.size() > 1;
        NttPassArgs a{};
        a.src = firstpass ? data : tmp;
        a.dst = (i == 0) ? data : tmp;
        a.src_vec_stride = a.dst_vec_stride = stride;
        SP_TRY(inv_roots_small(p.r, &a.small_tw));
        a.big_tw = big; a.logM = k; a.logL = k; a.s = p.s; a.r = p.r; a.g = p.g;
        a.weak_out = i != 0;
        int sm = NTT_STORE_INPLACE;
        if (i == 0) { sm = NTT_STORE_SCATTER_BITREV; a.scalar = d_scalar_; }
        SP_TRY(launch_pass(true, NTT_LOAD_INPLACE, sm, a, batch));
    }
    return SP_OK;
}

int NttEngine::lde_from_bitrev(const fe* coeffs, fe* dst, int k, int logb, uint32_t batch, uint64_t ss, uint64_t ds,
                               int shard_log, int shard_rank) {
    int K = k + logb;
    if (shard_log < 0 || shard_log > logb) { sp_set_error("lde: more shards than cosets"); return SP_E_INVALID_ARG; }
    if (logb == 0) {  // plain evaluation: copy then DIT in place
        for (uint32_t v = 0; v < batch; ++v)
            SP_HIP_CHECK(hipMemcpyAsync(dst + v * ds, coeffs + v * ss, sizeof(fe) << k, hipMemcpyDeviceToDevice, stream_));
        return dit_bitrev_to_natural(dst, k, batch, ds);
    }
    const fe* big = nullptr;
    SP_TRY(roots(K, &big));
    std::vector<PassGeom> geo = geometry(K, logb, NTT_MAX_CONTIG_LOG, ((uint64_t)batch << (K - shard_log)) * sizeof(fe));
    if (geo.empty()) {  // k == 0: constant polynomial replicated
        geo.push_back({logb, 0, std::min(logb, NTT_STRIDED_G_LOG)});
    }
    bool first = true;
    size_t pi = 0;
    for (const PassGeom& p : geo) {
        NttPassArgs a{};
        a.src = first ? coeffs : dst; a.dst = dst;
        a.src_vec_stride = first ? ss : ds; a.dst_vec_stride = ds;
        a.weak_out = ++pi < geo.size();
        SP_TRY(roots(p.r, &a.small_tw));
        a.big_tw = big; a.logM = K; a.logL = (uint32_t)(K - shard_log); a.s = (uint32_t)(p.s - shard_log); a.r = p.r; a.log_expand = logb;
        a.tw_shift = (uint32_t)shard_log; a.tw_low = (uint32_t)shard_rank;   // local index -> global: (index << shard_log) | rank
        a.g = (uint32_t)std::min<int>(p.g, p.s - shard_log);  // adjacent elements per row cannot exceed the local stride
        SP_TRY(launch_pass(false, first ? NTT_LOAD_EXPAND : NTT_LOAD_INPLACE, NTT_STORE_INPLACE, a, batch));
        first = false;
    }
    return SP_OK;
}

// Coset-major LDE: dst column v = [coset c_loc][m], element = p_v(h w_N^(m b + c)) with c = c_loc * 2^shard_log + rank.
// Every coset is a size-n DIT of the same bit-reversed coefficients whose twiddles carry the coset index in their low
// bits (w_N^(((m mod 2^(j-1)) b + c) << (K - logb - j)) at stage j): the first pass is a contiguous one (coalesced
// reads of the coefficients, no replication), and what consumes one or two cosets - the constraint composition, the
// DEEP quotient - reads contiguous memory instead of every b-th or (b/2)-th element of the natural order.
int NttEngine::lde_coset_major(const fe* coeffs, fe* dst, int k, int logb, uint32_t batch, uint64_t ss, uint64_t ds, int shard_log, int shard_rank) {
    const int K = k + logb;
    if (shard_log < 0 || shard_log > logb) { sp_set_error("lde: more shards than cosets"); return SP_E_INVALID_ARG; }
    const uint32_t b_loc = 1u << (logb - shard_log);
    const uint64_t n = 1ull << k;
    const fe* big = nullptr;
    SP_TRY(roots(K, &big));
    constexpr int contig_cap = NTT_TILE_LOG - NTT_STRIDED_G_LOG;
    std::vector<PassGeom> geo = geometry(k, 0, k <= NTT_TILE_LOG ? NTT_TILE_LOG : contig_cap, ((uint64_t)batch * b_loc << k) * sizeof(fe));
    if (geo.empty()) geo.push_back({0, 0, 0});
    bool first = true;
    size_t pi = 0;
    for (const PassGeom& p : geo) {
        NttPassArgs a{};
        a.src = first ? coeffs : dst; a.dst = dst;
        a.src_vec_stride = first ? ss : ds; a.dst_vec_stride = ds;
        a.coset_count = b_loc; a.src_coset_stride = first ? 0 : n; a.dst_coset_stride = n;
        a.weak_out = ++pi < geo.size();
        a.big_tw = big; a.logM = K; a.logL = (uint32_t)k; a.s = p.s; a.r = p.r; a.g = p.g;
        a.tw_shift = (uint32_t)logb; a.tw_low = (uint32_t)shard_rank; a.shard_log = (uint32_t)shard_log;
        SP_TRY(launch_pass(false, NTT_LOAD_INPLACE, NTT_STORE_INPLACE, a, batch * b_loc));
        first = false;
    }
    return SP_OK;
}

int NttEngine::scale_by_powers(fe* data, uint64_t n, uint32_t batch, uint64_t stride, const fe& base, const fe* c) {
    PowArgs args;
    fe cur = base;
    for (int i = 0; i < 40; ++i) { args.pw[i] = cur; cur = fe_sqr(cur); }
    args.has_c = c ? 1 : 0;
    args.c = c ? *c : fe_one();
    dim3 grid((unsigned)((n + 255) / 256), batch);
    hipLaunchKernelGGL(scale_powers_kernel, grid, dim3(256), 0, stream_, data, n, stride, args);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

}  // namespace sp
