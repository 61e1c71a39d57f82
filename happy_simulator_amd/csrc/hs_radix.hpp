// hs_radix.hpp -- stable LSD radix sort of (uint64 key, uint64 value) pairs for gfx950, 8-bit digits.
//
// Used by the load-balancer engine (hs_lb.hip) for the two places where the reference's global event heap
// (core/event_heap.py:54-108) really orders events of DIFFERENT entities against each other:
//   * requests fanned out by the LoadBalancer must reach each backend in (time, creation) order
//     -> sort by (backend, arrival ns);
//   * one Sink shared by all backends records completions in global processing order
//     -> sort by completion ns.
//
// Everything is sized for the worst case at engine creation and driven by a DEVICE-side element count, so a whole
// run is enqueued without a single host synchronisation.  Per pass (HBM-bound; 40 B per element):
//   radix_hist     8 B read   per-tile digit histogram -> hist[digit][tile]
//   radix_scan_*   --         exclusive scan of hist in (digit, tile) order
//   radix_scatter  16 B read + 16 B write; ranks from wavefront ballots (64 lanes), tile reordered through LDS so the
//                  stores are contiguous runs
// Round 3 also built the single-pass variant -- ONE histogram read per sort and scatters with decoupled look-back (Merrill &
// Garland; "Onesweep", Adinets & Merrill): 32 B per element and pass plus 8 B per element and sort -- and measured it SLOWER on
// MI355X (a dense pass of 11.8 M elements: 162 us against 101 + 35 + 23): the ~1 000 tiles of a launch start together, so a
// tile walks hundreds of predecessors' aggregates before it meets an inclusive prefix, and every descriptor load is a
// device-scope access that leaves its XCD's L2 (eight XCDs, one L2 each), ~1 us per window of 16 tiles.  It stays behind the
// load-balancer engine's debug flag 16 (bit-identical: tests/test_gpu_lb.py):
//   radix_hist_all     8 B read, once per sort: the digit histograms of ALL passes -> ghist[pass][digit] (the multiset of keys does
//                      not change from pass to pass)
//   radix_digit_bases  exclusive scan of every pass's 256 counters
//   radix_scatter_lb   16 B read + 16 B write per pass; ranks from wavefront ballots (64 lanes), tile reordered through LDS so the
//                      stores are contiguous runs; where a tile's digit run starts = the pass's digit base + what the tiles before
//                      it hold of that digit, found by looking back along the tiles' published (aggregate | inclusive prefix)
//                      words, one thread per digit; tiles are numbered by an atomic ticket so that a tile only ever waits for
//                      tiles that started before it

// A tile is 256 threads x kItems rows of 64 consecutive elements per wavefront, so ranks are stable by construction:
// element order = (wave, row, lane).  Elements can be masked out of the FIRST pass (ragged inputs) by a validity
// functor; from then on the data is dense.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hs {

constexpr int kRadixBits = 8;
constexpr int kRadixBins = 1 << kRadixBits;
constexpr int kRadixThreads = 256;
constexpr int kRadixWaves = kRadixThreads / 64;
constexpr int kRadixItems = 16;                                   // rows of 64 elements per wavefront per tile
constexpr int kRadixTile = kRadixThreads * kRadixItems;           // 4096 elements

// validity of input slot i in the first pass
struct RadixAll {
    __device__ __forceinline__ bool operator()(int64_t) const { return true; }
};

__device__ __forceinline__ uint64_t lanemask_lt() {
    const unsigned lane = __lane_id();
    return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

// lanes of the wavefront (among `valid` ones) whose 8-bit digit equals this lane's
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid) {
    uint64_t m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < kRadixBits; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t v = __ballot(bit);
        m &= bit ? v : ~v;
    }
    return m;
}

// hist[digit * n_tiles + tile] = number of valid elements of the tile with that digit
// (Round 4 measured two alternatives at the configs[4] size, both bit-identical, both rejected: a TILE-major histogram -- this kernel
//  33.6 -> 26.8 us with coalesced stores, but the scan down the columns needs two dependent phases of ~100 serial device-scope
//  loads: 115 us against radix_scan_rows' 18; and four tiles per block with one 16-byte store per digit row: 47.6 us, a quarter of
//  the workgroups keeps fewer loads in flight than the stores save.)
template <typename Valid>
__global__ void __launch_bounds__(kRadixThreads) radix_hist(const uint64_t *__restrict__ keys, const int64_t *n_ptr,
                                                            int shift, uint32_t *__restrict__ hist, int n_tiles,
                                                            Valid valid) {
    __shared__ uint32_t h[kRadixBins];
    const int tid = threadIdx.x;
    h[tid] = 0;
    __syncthreads();
    const int64_t n = *n_ptr;
    const int64_t base = (int64_t)blockIdx.x * kRadixTile;
    if (base < n) {
#pragma unroll 4
        for (int r = 0; r < kRadixItems; ++r) {
            const int64_t i = base + (int64_t)r * kRadixThreads + tid;
            if (i < n && valid(i)) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & (kRadixBins - 1)], 1u);
        }
    }
    __syncthreads();
    hist[(size_t)tid * n_tiles + blockIdx.x] = h[tid];
}

// exclusive scan of every digit row (one block per digit), row totals out; the LAST block to finish (a ticket) scans the 256 row
// totals into the digit bases and writes the element count of the pass -- what radix_scan_digits did in a launch of its own
// (round 4: six launches fewer per run)
__global__ void __launch_bounds__(kRadixThreads) radix_scan_rows(uint32_t *__restrict__ hist, int n_tiles,
                                                                 uint32_t *__restrict__ row_total, uint32_t *__restrict__ digit_base,
                                                                 int64_t *n_out, uint32_t *__restrict__ ticket) {
    // (round 4: 16 consecutive entries per thread through LDS -- one block-wide scan per 4 096 entries instead of one per 256: the row of
    //  the configs[4] load balancer, 2 880 tiles, took twelve barrier-separated rounds = 22 us per launch, six launches per run)
    constexpr int kPer = 16, kChunk = kRadixThreads * kPer;
    __shared__ uint32_t buf[kChunk + kChunk / kPer];               // entry e at e + e / 16: a thread's 16 entries hit 16 different banks
    __shared__ uint32_t wsum[kRadixWaves];
    __shared__ uint32_t carry;
    __shared__ bool is_last;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    uint32_t *row = hist + (size_t)blockIdx.x * n_tiles;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n_tiles; c0 += kChunk) {
#pragma unroll
        for (int k = 0; k < kPer; ++k) {                             // coalesced in
            const int e = k * kRadixThreads + tid, i = c0 + e;
            buf[e + e / kPer] = i < n_tiles ? row[i] : 0u;
        }
        __syncthreads();
        uint32_t loc[kPer], v = 0;
#pragma unroll
        for (int j = 0; j < kPer; ++j) { loc[j] = v; v += buf[tid * (kPer + 1) + j]; }   // exclusive inside the thread, v = its total
        uint32_t s = v;                                              // inclusive scan of the thread totals inside the wavefront
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(s, o, 64);
            if (lane >= o) s += t;
        }
        if (lane == 63) wsum[w] = s;
        __syncthreads();
        uint32_t wbase = 0;
        for (int k = 0; k < w; ++k) wbase += wsum[k];
        const uint32_t c = carry, base = c + wbase + s - v;
#pragma unroll
        for (int j = 0; j < kPer; ++j) buf[tid * (kPer + 1) + j] = base + loc[j];
        __syncthreads();
        if (tid == kRadixThreads - 1) carry = c + wbase + s;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {                             // coalesced out
            const int e = k * kRadixThreads + tid, i = c0 + e;
            if (i < n_tiles) row[i] = buf[e + e / kPer];
        }
        __syncthreads();
    }
    if (tid == 0) {
        __hip_atomic_store(&row_total[blockIdx.x], carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        is_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    const uint32_t v = __hip_atomic_load(&row_total[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // kRadixBins == kRadixThreads
    uint32_t s = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(s, o, 64);
        if (lane >= o) s += t;
    }
    if (lane == 63) wsum[w] = s;
    __syncthreads();
    uint32_t wbase = 0;
    for (int k = 0; k < w; ++k) wbase += wsum[k];
    digit_base[tid] = wbase + s - v;
    if (tid == kRadixThreads - 1 && n_out) *n_out = (int64_t)(wbase + s);
    if (tid == 0) *ticket = 0;
}

// exclusive scan of the 256 row totals -> digit bases; also the element count of the pass (all digits)
__global__ void __launch_bounds__(kRadixThreads) radix_scan_digits(const uint32_t *__restrict__ row_total,
                                                                   uint32_t *__restrict__ digit_base, int64_t *n_out) {
    __shared__ uint32_t wsum[kRadixWaves];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t v = row_total[tid];
    uint32_t s = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(s, o, 64);
        if (lane >= o) s += t;
    }
    if (lane == 63) wsum[w] = s;
    __syncthreads();
    uint32_t wbase = 0;
    for (int k = 0; k < w; ++k) wbase += wsum[k];
    digit_base[tid] = wbase + s - v;
    if (tid == kRadixThreads - 1 && n_out) *n_out = (int64_t)(wbase + s);
}

// Stable scatter of one pass.  MakeVal builds the value of input slot i when the pass starts from keys only
// (vals_in == nullptr): e.g. the slot index itself.  The tile is first put in digit order in LDS (32 KB, keys then values), then written
// out by consecutive threads, so that a wavefront store covers a handful of contiguous runs (one per digit value
// present) instead of 64 unrelated cache lines.
template <typename Valid, typename MakeVal>
__global__ void __launch_bounds__(kRadixThreads) radix_scatter(const uint64_t *__restrict__ keys_in,
                                                               const uint64_t *__restrict__ vals_in,
                                                               uint64_t *__restrict__ keys_out,
                                                               uint64_t *__restrict__ vals_out, const int64_t *n_ptr,
                                                               int shift, const uint32_t *__restrict__ hist,
                                                               const uint32_t *__restrict__ digit_base, int n_tiles,
                                                               Valid valid, MakeVal make_val) {
    __shared__ uint64_t sbuf[kRadixTile];                 // the tile in digit order: keys first, then values (32 KB)
    __shared__ uint32_t cnt[kRadixWaves][kRadixBins];     // per wave: elements with the digit; then: local start of its sub-run
    __shared__ uint32_t dstart[kRadixBins];               // start of the digit's run inside the sorted tile
    __shared__ uint32_t gbase[kRadixBins];                // global position of the digit's run of this tile
    __shared__ uint32_t wsum[kRadixWaves];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t n = *n_ptr;
    const int64_t base = (int64_t)blockIdx.x * kRadixTile;
    if (base >= n) return;
#pragma unroll
    for (int k = 0; k < kRadixWaves; ++k) cnt[k][tid] = 0;
    __syncthreads();
    uint64_t key[kRadixItems], val[kRadixItems];
    uint32_t rank[kRadixItems];                                    // low 24 bits rank in the wave's chunk, high 8 digit
    const uint64_t lt = lanemask_lt();
    const int64_t wbase = base + (int64_t)w * (kRadixItems * 64);
#pragma unroll
    for (int r = 0; r < kRadixItems; ++r) {
        const int64_t i = wbase + r * 64 + lane;
        const bool ok = i < n && valid(i);
        key[r] = ok ? keys_in[i] : 0ull;
        val[r] = ok ? (vals_in ? vals_in[i] : make_val(i)) : 0ull;
        const uint32_t d = (uint32_t)(key[r] >> shift) & (kRadixBins - 1);
        const uint64_t m = match_digit(d, ok);
        const uint32_t before = cnt[w][d];                         // elements of earlier rows with this digit
        const uint32_t pos = (uint32_t)__popcll(m & lt);
        rank[r] = ok ? ((d << 24) | (before + pos)) : 0xffffffffu;
        __builtin_amdgcn_wave_barrier();
        if (ok && pos == 0) cnt[w][d] = before + (uint32_t)__popcll(m);   // one leader per digit value
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    uint32_t total;
    {   // thread `tid` owns digit `tid`: the tile's count, its start in the sorted tile (block-wide exclusive scan), the
        // waves' sub-runs, and the run's global position
        uint32_t c[kRadixWaves];
        total = 0;
#pragma unroll
        for (int k = 0; k < kRadixWaves; ++k) { c[k] = cnt[k][tid]; total += c[k]; }
        uint32_t sc = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(sc, o, 64);
            if (lane >= o) sc += t;
        }
        if (lane == 63) wsum[w] = sc;
        __syncthreads();
        uint32_t wb = 0;
        for (int k = 0; k < w; ++k) wb += wsum[k];
        const uint32_t start = wb + sc - total;
        dstart[tid] = start;
        gbase[tid] = hist[(size_t)tid * n_tiles + blockIdx.x] + digit_base[tid];
        uint32_t run = start;
#pragma unroll
        for (int k = 0; k < kRadixWaves; ++k) { cnt[k][tid] = run; run += c[k]; }
    }
    __syncthreads();
    uint32_t n_tile = 0;
#pragma unroll
    for (int k = 0; k < kRadixWaves; ++k) n_tile += wsum[k];       // valid elements of the tile
    uint32_t lpos[kRadixItems];
#pragma unroll
    for (int r = 0; r < kRadixItems; ++r) {
        lpos[r] = rank[r] == 0xffffffffu ? 0xffffffffu : cnt[w][rank[r] >> 24] + (rank[r] & 0xffffffu);
        if (lpos[r] != 0xffffffffu) sbuf[lpos[r]] = key[r];
    }
    __syncthreads();
    size_t g[kRadixItems];                                         // global position of sorted element tid + r * 256
#pragma unroll
    for (int r = 0; r < kRadixItems; ++r) {
        const uint32_t p = (uint32_t)tid + (uint32_t)r * kRadixThreads;
        g[r] = 0;
        if (p < n_tile) {
            const uint64_t k = sbuf[p];
            const uint32_t d = (uint32_t)(k >> shift) & (kRadixBins - 1);
            g[r] = (size_t)gbase[d] + (p - dstart[d]);
            keys_out[g[r]] = k;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRadixItems; ++r)
        if (lpos[r] != 0xffffffffu) sbuf[lpos[r]] = val[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRadixItems; ++r) {
        const uint32_t p = (uint32_t)tid + (uint32_t)r * kRadixThreads;
        if (p < n_tile) vals_out[g[r]] = sbuf[p];
    }
}

// ---- round 3: one histogram read per sort, single-pass scatters with decoupled look-back ---------------------------------------
constexpr int kRadixMaxPasses = 8;
constexpr uint32_t kDescAggregate = 1u << 30, kDescPrefix = 2u << 30, kDescMask = (1u << 30) - 1u;

// ghist[p * 256 + d] += number of valid input slots whose key has digit d in pass p (bits shift0 + 8 p ...); grid-stride over tiles
template <typename Valid>
__global__ void __launch_bounds__(kRadixThreads) radix_hist_all(const uint64_t *__restrict__ keys, const int64_t *n_ptr, int shift0,
                                                                int passes, uint32_t *__restrict__ ghist, Valid valid) {
    __shared__ uint32_t h[kRadixMaxPasses][kRadixBins];
    const int tid = threadIdx.x;
    for (int p = 0; p < passes; ++p) h[p][tid] = 0;
    __syncthreads();
    const int64_t n = *n_ptr;
    for (int64_t base = (int64_t)blockIdx.x * kRadixTile; base < n; base += (int64_t)gridDim.x * kRadixTile) {
#pragma unroll 4
        for (int r = 0; r < kRadixItems; ++r) {
            const int64_t i = base + (int64_t)r * kRadixThreads + tid;
            if (i < n && valid(i)) {
                const uint64_t k = keys[i] >> shift0;
                for (int p = 0; p < passes; ++p) atomicAdd(&h[p][(uint32_t)(k >> (p * kRadixBits)) & (kRadixBins - 1)], 1u);
            }
        }
    }
    __syncthreads();
    for (int p = 0; p < passes; ++p) { const uint32_t c = h[p][tid]; if (c) atomicAdd(&ghist[p * kRadixBins + tid], c); }
}

// per pass: exclusive scan of its 256 counters (in place); the element count of the sort; the tile tickets back to 0
__global__ void __launch_bounds__(kRadixThreads) radix_digit_bases(uint32_t *__restrict__ ghist, int passes, int64_t *n_out,
                                                                   uint32_t *__restrict__ tickets) {
    __shared__ uint32_t wsum[kRadixWaves];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int p = 0; p < passes; ++p) {
        const uint32_t v = ghist[p * kRadixBins + tid];
        uint32_t s = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(s, o, 64);
            if (lane >= o) s += t;
        }
        if (lane == 63) wsum[w] = s;
        __syncthreads();
        uint32_t wbase = 0;
        for (int k = 0; k < w; ++k) wbase += wsum[k];
        ghist[p * kRadixBins + tid] = wbase + s - v;
        if (p == 0 && tid == kRadixThreads - 1 && n_out) *n_out = (int64_t)(wbase + s);
        __syncthreads();
    }
    if (tid < kRadixMaxPasses) tickets[tid] = 0;
}

// One pass.  `desc` [tiles][256], zero when the pass starts: what tile t holds of digit d, first as an AGGREGATE (its own count),
// then as an inclusive PREFIX (all tiles 0 .. t).  `ticket`: the pass's tile counter.  `err`: set when a look-back gave up.
template <typename Valid, typename MakeVal>
__global__ void __launch_bounds__(kRadixThreads) radix_scatter_lb(const uint64_t *__restrict__ keys_in,
                                                                  const uint64_t *__restrict__ vals_in,
                                                                  uint64_t *__restrict__ keys_out,
                                                                  uint64_t *__restrict__ vals_out, const int64_t *n_ptr,
                                                                  int shift, const uint32_t *__restrict__ digit_base,
                                                                  uint32_t *__restrict__ desc, uint32_t *__restrict__ ticket,
                                                                  int *__restrict__ err, Valid valid, MakeVal make_val) {
    __shared__ uint64_t sbuf[kRadixTile];                 // the tile in digit order: keys first, then values (32 KB)
    __shared__ uint32_t cnt[kRadixWaves][kRadixBins];     // per wave: elements with the digit; then: local start of its sub-run
    __shared__ uint32_t dstart[kRadixBins];               // start of the digit's run inside the sorted tile
    __shared__ uint32_t gbase[kRadixBins];                // global position of the digit's run of this tile
    __shared__ uint32_t wsum[kRadixWaves];
    __shared__ uint32_t s_tile;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t n = *n_ptr;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);         // tiles in the order they START: a tile only waits for earlier ones
#pragma unroll
    for (int k = 0; k < kRadixWaves; ++k) cnt[k][tid] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    const int64_t base = (int64_t)tile * kRadixTile;
    if (base >= n) return;
    uint64_t key[kRadixItems], val[kRadixItems];
    uint32_t rank[kRadixItems];                                    // low 24 bits rank in the wave's chunk, high 8 digit
    const uint64_t lt = lanemask_lt();
    const int64_t wbase = base + (int64_t)w * (kRadixItems * 64);
#pragma unroll
    for (int r = 0; r < kRadixItems; ++r) {
        const int64_t i = wbase + r * 64 + lane;
        const bool ok = i < n && valid(i);
        key[r] = ok ? keys_in[i] : 0ull;
        val[r] = ok ? (vals_in ? vals_in[i] : make_val(i)) : 0ull;
        const uint32_t d = (uint32_t)(key[r] >> shift) & (kRadixBins - 1);
        const uint64_t m = match_digit(d, ok);
        const uint32_t before = cnt[w][d];                         // elements of earlier rows with this digit
        const uint32_t pos = (uint32_t)__popcll(m & lt);
        rank[r] = ok ? ((d << 24) | (before + pos)) : 0xffffffffu;
        __builtin_amdgcn_wave_barrier();
        if (ok && pos == 0) cnt[w][d] = before + (uint32_t)__popcll(m);   // one leader per digit value
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    uint32_t total;
    {   // thread `tid` owns digit `tid`: the tile's count, its start in the sorted tile (block-wide exclusive scan), the
        // waves' sub-runs, and -- by look-back -- the run's global position
        uint32_t c[kRadixWaves];
        total = 0;
#pragma unroll
        for (int k = 0; k < kRadixWaves; ++k) { c[k] = cnt[k][tid]; total += c[k]; }
        uint32_t *mine = desc + (size_t)tile * kRadixBins + tid;
        __hip_atomic_store(mine, total | (tile == 0 ? kDescPrefix : kDescAggregate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t sc = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(sc, o, 64);
            if (lane >= o) sc += t;
        }
        if (lane == 63) wsum[w] = sc;
        __syncthreads();
        uint32_t wb = 0;
        for (int k = 0; k < w; ++k) wb += wsum[k];
        const uint32_t start = wb + sc - total;
        dstart[tid] = start;
        uint32_t excl = 0;
        if (tile > 0) {
            // a WINDOW of earlier tiles per round trip: the tiles of a launch start together, so the nearest inclusive prefix can be
            // hundreds of tiles back, and one dependent load per tile would serialise hundreds of memory latencies
            constexpr int W = 16;
            int64_t t = (int64_t)tile - 1;
            unsigned spins = 0;
            bool done = false;
            while (!done) {
                uint32_t v[W];
#pragma unroll
                for (int k = 0; k < W; ++k)
                    v[k] = (t - k) >= 0 ? __hip_atomic_load(desc + (size_t)(t - k) * kRadixBins + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                        : kDescPrefix;                   // (before tile 0: an empty prefix)
                int used = 0;
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    if (done || used != k) continue;
                    if (v[k] == 0u) continue;                            // not published yet: take what came before it, ask again
                    excl += v[k] & kDescMask;
                    used = k + 1;
                    if (v[k] & kDescPrefix) done = true;
                }
                t -= used;
                if (!done && used < W) {
                    if (++spins > (1u << 22)) { *err = 1; break; }       // bounded: report instead of hanging the device
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __hip_atomic_store(mine, (excl + total) | kDescPrefix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        gbase[tid] = digit_base[tid] + excl;
        uint32_t run = start;
#pragma unroll
        for (int k = 0; k < kRadixWaves; ++k) { cnt[k][tid] = run; run += c[k]; }
    }
    __syncthreads();
    uint32_t n_tile = 0;
#pragma unroll
    for (int k = 0; k < kRadixWaves; ++k) n_tile += wsum[k];       // valid elements of the tile
    uint32_t lpos[kRadixItems];
#pragma unroll
    for (int r = 0; r < kRadixItems; ++r) {
        lpos[r] = rank[r] == 0xffffffffu ? 0xffffffffu : cnt[w][rank[r] >> 24] + (rank[r] & 0xffffffu);
        if (lpos[r] != 0xffffffffu) sbuf[lpos[r]] = key[r];
    }
    __syncthreads();
    size_t g[kRadixItems];                                         // global position of sorted element tid + r * 256
#pragma unroll
    for (int r = 0; r < kRadixItems; ++r) {
        const uint32_t p = (uint32_t)tid + (uint32_t)r * kRadixThreads;
        g[r] = 0;
        if (p < n_tile) {
            const uint64_t k = sbuf[p];
            const uint32_t d = (uint32_t)(k >> shift) & (kRadixBins - 1);
            g[r] = (size_t)gbase[d] + (p - dstart[d]);
            keys_out[g[r]] = k;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRadixItems; ++r)
        if (lpos[r] != 0xffffffffu) sbuf[lpos[r]] = val[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRadixItems; ++r) {
        const uint32_t p = (uint32_t)tid + (uint32_t)r * kRadixThreads;
        if (p < n_tile) vals_out[g[r]] = sbuf[p];
    }
}

}  // namespace hs
