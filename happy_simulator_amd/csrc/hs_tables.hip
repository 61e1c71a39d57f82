// hs_tables.hip -- the tick-table kernel (hs_tables.hpp) and its launcher.
//
// Replaces, for Sources with a time-varying profile and for Probes, the per-tick numerical inversion of
// `ArrivalTimeProvider.next_arrival_time` (happysimulator/load/arrival_time_provider.py:84-144, numerics/integration.py:11-90,
// numerics/root_finding.py:27-152) by one table per tick stream, produced before the run.
#include <hip/hip_runtime.h>

#include "hs_tables.hpp"
#include "hs_tables_api.hpp"

using namespace hs;

// One wavefront per row: tick 0 from start_ns, tick k + 1 from tick k, until two ticks lie beyond the horizon (the first one is
// the run's pending SourceEvent beyond end_time, the second is what processing THAT one creates), the stream ends (kInfNs: the
// reference raises, load/source.py:176-180) or travels back in time (the tick is popped and dropped, core/simulation.py:480-489).
// status[0]: 2 + owner of the first row whose arrival exceeded the evaluation budget (atomicMax, 0 = none);
// status[1]: 2 + owner of the first row that did not fit its `cap` entries.
__global__ void __launch_bounds__(64) hs_tick_tables_kernel(const TickRow *rows, int n_rows, int64_t start_ns, int64_t horizon_ns,
                                                            int64_t cap, int64_t *times, int64_t *count,
                                                            unsigned long long *status, long long lane_budget) {
    __shared__ CoopTree T;
    const int r = blockIdx.x;
    if (r >= n_rows) return;
    const TickRow R = rows[r];
    Profile pf;
    pf.kind = R.kind; pf.p0 = R.p0; pf.p1 = R.p1; pf.p2 = R.p2; pf.p3 = R.p3; pf.owner = R.owner;
    Stream s;
    s.init(R.seed, R.sid, 0);
    int64_t *out = times + (size_t)r * (size_t)cap;
    int64_t t = start_ns, k = 0;
    int beyond = 0;
    for (;;) {
        if (k >= cap) { if (threadIdx.x == 0) atomicMax(&status[1], (unsigned long long)(R.owner + 2)); break; }
        const double area = R.poisson ? exp1_from_uniform(s.next_uniform()) : 1.0;   // poisson_arrival.py:31 / constant_arrival.py:23
        long long visits = 0;
        CoopIntegrator ci{&T, &visits, lane_budget};
        bool over = false;
        const int64_t a = prof_next_arrival_with(pf, t, area, ci, over);
        if (over) {
            if (threadIdx.x == 0) { atomicMax(&status[0], (unsigned long long)(R.owner + 2)); out[k] = kInfNs; }
            ++k;
            break;
        }
        if (threadIdx.x == 0) out[k] = a;
        ++k;
        if (a == kInfNs || a < t) break;
        if (a > horizon_ns && ++beyond >= 2) break;
        t = a;
    }
    if (threadIdx.x == 0) count[r] = k;
}

// the same chain on ONE lane with the sequential integrator of hs_profile.hpp (the restatement that the goldens pinned in rounds
// 1 - 2): the reference the cooperative kernel is compared with on the device (hs_debug_tick_table, tests/test_gpu_tables.py)
__global__ void hs_tick_tables_lone_kernel(const TickRow *rows, int n_rows, int64_t start_ns, int64_t horizon_ns, int64_t cap,
                                           int64_t *times, int64_t *count, unsigned long long *status, long long budget) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const TickRow R = rows[r];
    Profile pf;
    pf.kind = R.kind; pf.p0 = R.p0; pf.p1 = R.p1; pf.p2 = R.p2; pf.p3 = R.p3; pf.owner = R.owner;
    Stream s;
    s.init(R.seed, R.sid, 0);
    int64_t *out = times + (size_t)r * (size_t)cap;
    int64_t t = start_ns, k = 0;
    int beyond = 0;
    for (;;) {
        if (k >= cap) { atomicMax(&status[1], (unsigned long long)(R.owner + 2)); break; }
        const double area = R.poisson ? exp1_from_uniform(s.next_uniform()) : 1.0;
        bool over = false;
        const int64_t a = prof_next_arrival(pf, t, area, budget, over);
        if (over) { atomicMax(&status[0], (unsigned long long)(R.owner + 2)); out[k++] = kInfNs; break; }
        out[k++] = a;
        if (a == kInfNs || a < t) break;
        if (a > horizon_ns && ++beyond >= 2) break;
        t = a;
    }
    count[r] = k;
}

__global__ void hs_fill_i64(int64_t *p, int64_t v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

namespace hs {

hipError_t tick_tables_launch(hipStream_t stream, const TickRow *rows_dev, int n_rows, int64_t start_ns, int64_t horizon_ns,
                              int64_t cap, int64_t *times_dev, int64_t *count_dev, unsigned long long *status_dev,
                              long long lane_budget, bool lone) {
    if (n_rows <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(status_dev, 0, 2 * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    const size_t total = (size_t)n_rows * (size_t)cap;
    unsigned fb = (unsigned)((total + 255) / 256);
    if (fb > 65535u) fb = 65535u;
    hipLaunchKernelGGL(hs_fill_i64, dim3(fb ? fb : 1), dim3(256), 0, stream, times_dev, kInfNs, total);
    if (lone)
        hipLaunchKernelGGL(hs_tick_tables_lone_kernel, dim3((unsigned)((n_rows + 63) / 64)), dim3(64), 0, stream, rows_dev, n_rows,
                           start_ns, horizon_ns, cap, times_dev, count_dev, status_dev, lane_budget);
    else
        hipLaunchKernelGGL(hs_tick_tables_kernel, dim3((unsigned)n_rows), dim3(64), 0, stream, rows_dev, n_rows, start_ns,
                           horizon_ns, cap, times_dev, count_dev, status_dev, lane_budget);
    return hipGetLastError();
}

}  // namespace hs
