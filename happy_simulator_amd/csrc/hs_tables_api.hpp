// hs_tables_api.hpp -- host-side entry of the tick-table kernel (hs_tables.hip), shared by the station / network engine
// (hs_engine.hip) and the load-balancer engine (hs_lb.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "hs_tables.hpp"

namespace hs {

// default evaluation budget of ONE arrival: Simpson intervals a single lane may visit (64 lanes work on an integral, so an
// arrival may cost up to 64 x this; ~50 ns per interval and lane).  hs_engine_set_profile_budget / hs_lb_set_profile_budget.
constexpr long long kDefaultLaneBudget = 1ll << 24;

// Enqueue on `stream`: fill times[n_rows][cap] with kInfNs, then one wavefront per row (lone: one LANE per row with the
// sequential integrator -- the device-side reference of tests/test_gpu_tables.py).  status_dev: 2 x unsigned long long.
hipError_t tick_tables_launch(hipStream_t stream, const TickRow *rows_dev, int n_rows, int64_t start_ns, int64_t horizon_ns,
                              int64_t cap, int64_t *times_dev, int64_t *count_dev, unsigned long long *status_dev,
                              long long lane_budget, bool lone);

}  // namespace hs
