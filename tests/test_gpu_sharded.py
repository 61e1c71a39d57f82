"""GPU: a station network partitioned over several engines (virtual shards on one device, LocalComm) must give
exactly the single-engine result -- which is itself pinned against the live-reference ring goldens and the oracle
(tests/test_gpu_ring.py).  Same protocol and kernels as the one-process-per-GPU RCCL path (DistComm)."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _single(spec):
    eng, p = H.ring_engine_for_spec(spec)
    with eng:
        eng.run_until(p["end_ns"])
        s = eng.summary()
        return dict(events=s.events_processed, by_kind=s.events_by_kind, final=s.final_time_ns, stats=eng.lp_stats(),
                    net=eng.net_stats(), sinks=eng.read_sinks(), completed=s.requests_completed,
                    sink_records=s.sink_records)


def _sharded(spec, world, sync_every=16, msg_capacity=64, rounds=True, bag_capacity=0):
    exchange = "device"
    if rounds == "collective":               # asynchronous rounds over the all-to-all + all-reduce instead of the device-side exchange
        rounds, exchange = True, "collective"
    from happy_simulator_amd.sharded import LocalComm, ShardedNetwork

    st, net, cap, p = H.ring_arrays(spec, bag_capacity=bag_capacity)
    sn = ShardedNetwork.on_gpu(st, net, LocalComm(world), horizon_ns=p["end_ns"], seed=spec["seed"], log_capacity=cap,
                               sync_every=sync_every, msg_capacity=msg_capacity, rounds=rounds, exchange=exchange)
    assert sn.device_exchange == (rounds is True and exchange == "device") and not sn.live
    with sn:
        summ = sn.run_until(p["end_ns"])
        stats = {}
        for s in sn.shards:
            for k, v in s.engine.lp_stats().items():
                stats.setdefault(k, []).append(v)
        stats = {k: np.concatenate(v) for k, v in stats.items()}
        nl = net.n_links
        netst = {"routed": np.concatenate([s.engine.net_stats()["routed"] for s in sn.shards]),
                 "link_entered": np.zeros(nl, np.int64), "link_packets_sent": np.zeros(nl, np.int64),
                 "link_packets_dropped": np.zeros(nl, np.int64)}
        for s in sn.shards:
            ns = s.engine.net_stats()
            netst["link_entered"][s.gids] += ns["link_entered"]
            netst["link_packets_sent"][s.gids] += ns["link_packets_sent"]
            netst["link_packets_dropped"][s.gids] += ns["link_packets_dropped"]
        parts = [s.engine.read_sinks() for s in sn.shards]
        sinks = tuple(np.concatenate([p_[i] for p_ in parts]) for i in range(3))
        if spec.get("probes"):
            summ.probes = {(i, j): sn.read_probe(i, j) for i, prs in enumerate(H.ring_params(spec)["probe_list"])
                           for j in range(len(prs))}
        return summ, stats, netst, sinks


SPECS = [
    dict(name="ring_64", topology="ring", n=64, ext_rate=4.0, mean=0.1, lat_min=0.001, jitter_mean=0.01, end_s=6.0, seed=21),
    dict(name="ring_7_dense", topology="ring", n=7, ext_rate=4.5, mean=0.1, lat_min=0.0002, jitter_mean=0.0008, end_s=3.0, seed=22),
    dict(name="ring_300_c2_cap", topology="ring", n=300, ext_rate=9.0, mean=0.1, concurrency=2, queue_cap=4, lat_min=0.002,
         jitter_mean=0.004, end_s=4.0, seed=23),
    dict(name="ring_33_fixed_latency", topology="ring", n=33, ext_rate=4.0, mean=0.1, lat_min=0.0025, jitter_mean=None,
         end_s=5.0, seed=24),
]


# (the LIVE exchange -- one launch per rank and run -- needs one shard per PROCESS: tests/test_gpu_dist.py)
PROTOCOLS = pytest.mark.parametrize("rounds", [True, "collective", False], ids=["async_rounds_device_exchange", "async_rounds_collectives", "windows"])


@PROTOCOLS
@pytest.mark.parametrize("world", [1, 2, 3, 4])
@pytest.mark.parametrize("spec", SPECS, ids=[s["name"] for s in SPECS])
def test_sharded_equals_single_engine(spec, world, rounds):
    one = _single(spec)
    summ, stats, netst, sinks = _sharded(spec, world, rounds=rounds)
    assert summ.events_processed == one["events"]
    np.testing.assert_array_equal(summ.events_by_kind, one["by_kind"])
    assert summ.final_time_ns == one["final"]
    assert summ.requests_completed == one["completed"] and summ.sink_records == one["sink_records"]
    for k in ("generated", "accepted", "dropped", "completed", "rejected", "total_service_s", "sink_received",
              "queue_depth", "active", "events", "final_time_ns"):
        np.testing.assert_array_equal(stats[k], one["stats"][k], err_msg=k)
    for k in ("routed", "link_entered", "link_packets_sent", "link_packets_dropped"):
        np.testing.assert_array_equal(netst[k], one["net"][k], err_msg=k)
    for a, b in zip(sinks, one["sinks"]):
        np.testing.assert_array_equal(a, b)
    assert summ.world == world and summ.windows >= 1


@PROTOCOLS
@pytest.mark.parametrize("name", H.golden_names("ring"))
def test_sharded_matches_reference_golden(name, rounds):
    gold = H.Golden(name)
    spec = gold.spec
    world = 2 if spec["n"] < 6 else 3
    summ, stats, netst, sinks = _sharded(spec, world, sync_every=8, rounds=rounds)
    if "probe_t_ns" in gold.arrays:          # probes sample on whichever shard owns their station
        for (i, j), (pt, pv) in summ.probes.items():
            gt, gv = gold.probe_samples(i, j)
            np.testing.assert_array_equal(pt, gt)
            np.testing.assert_array_equal(pv, gv)
    assert summ.events_processed == gold.meta["total_events"][0]
    assert summ.final_time_ns == gold.meta["final_ns"][0]
    for k, g in (("generated", "generated"), ("accepted", "accepted"), ("dropped", "dropped"), ("completed", "completed"),
                 ("sink_received", "received"), ("queue_depth", "depth"), ("total_service_s", "total_service_s")):
        np.testing.assert_array_equal(stats[k], gold.arrays[g], err_msg=k)
    np.testing.assert_array_equal(netst["routed"], gold.routed)
    np.testing.assert_array_equal(netst["link_packets_sent"], gold.packets_sent)
    if "packets_dropped" in gold.arrays:
        np.testing.assert_array_equal(netst["link_packets_dropped"], gold.packets_dropped)
    np.testing.assert_array_equal(sinks[1], gold.sink_t_ns)


@PROTOCOLS
def test_sharded_profiles_and_schedule_on_a_large_ring_equal_single_engine(rounds):
    """Time-varying profiles and scheduled Requests on a 130-station ring over 3 shards (asynchronous rounds and the window
    protocol) == the single engine, which tests/test_gpu_ring.py pins against the oracle."""
    from test_gpu_ring import _big_ring_with_profiles_and_schedule

    spec = _big_ring_with_profiles_and_schedule(130, seed=220)
    one = _single(spec)
    summ, stats, netst, sinks = _sharded(spec, 3, rounds=rounds)
    assert summ.events_processed == one["events"] and summ.final_time_ns == one["final"]
    np.testing.assert_array_equal(summ.events_by_kind, one["by_kind"])
    for k in ("generated", "accepted", "completed", "total_service_s", "sink_received", "queue_depth", "active"):
        np.testing.assert_array_equal(stats[k], one["stats"][k], err_msg=k)
    for k in ("routed", "link_packets_sent"):
        np.testing.assert_array_equal(netst[k], one["net"][k], err_msg=k)
    for a, b in zip(sinks, one["sinks"]):
        np.testing.assert_array_equal(a, b)


def test_gvt_windows_skip_idle_time():
    """Sparse traffic: GVT-driven window ends jump over idle stretches, so far fewer windows than end / W run."""
    spec = dict(name="sparse", topology="ring", n=16, ext_rate=0.05, mean=0.1, lat_min=0.001, jitter_mean=0.01, end_s=40.0,
                seed=31)
    one = _single(spec)
    summ, *_ = _sharded(spec, 2, sync_every=4, rounds=False)
    assert summ.events_processed == one["events"] and summ.final_time_ns == one["final"]
    assert summ.windows < 0.2 * (40.0 / 0.001)
    rsum, *_ = _sharded(spec, 2, sync_every=4, rounds=True)              # asynchronous rounds: fewer exchanges still
    assert rsum.events_processed == one["events"] and rsum.final_time_ns == one["final"]
    assert rsum.windows < summ.windows


def test_exchange_row_overflow_is_reported():
    from happy_simulator_amd import _native as N

    spec = dict(name="tiny_rows", topology="ring", n=2, ext_rate=400.0, mean=0.001, lat_min=0.05, jitter_mean=None,
                end_s=2.0, seed=5)
    with pytest.raises(N.EngineError, match="overflow"):
        _sharded(spec, 2, msg_capacity=2, rounds=False)
    # asynchronous rounds size their length to the rows: an iteration may append 2 x 4 groups x 1 worker = 8 messages to a
    # cross link (pre-sending stations: a departure + the next request's pre-send per group), so rounds of one iteration ...
    ref, *_ = _sharded(spec, 2, msg_capacity=256, rounds=False, bag_capacity=128)   # (20 messages in flight per link)
    summ, *_ = _sharded(spec, 2, msg_capacity=8, rounds=True, bag_capacity=128)
    assert summ.events_processed == ref.events_processed and summ.final_time_ns == ref.final_time_ns
    with pytest.raises(ValueError, match="msg_capacity 4 is too small"):       # ... but not even one iteration fits here
        _sharded(spec, 2, msg_capacity=4, rounds=True)


@PROTOCOLS
def test_full_station_count_sharded_equals_single_engine(rounds):
    """BASELINE configs[3] at full size: the 65 536-station ring cut into 4 contiguous segments (virtual shards, the same
    kernels and exchange protocol as one process per GPU over RCCL) for the full 60 simulated seconds = 60 002 GVT-driven
    windows, 2.7e8 events; every total, statistic, link counter and Sink record equals the single-engine run."""
    import time

    spec = dict(name="ring_full_sharded", topology="ring", n=65536, ext_rate=4.0, mean=0.1, lat_min=0.001, jitter_mean=0.01,
                end_s=60.0, seed=42)
    one = _single(spec)
    t0 = time.perf_counter()
    summ, stats, netst, sinks = _sharded(spec, 4, sync_every=64 if not rounds else 8, rounds=rounds)
    wall = time.perf_counter() - t0
    print(f"4 shards, {'asynchronous rounds' if rounds else 'windows'}: {summ.windows} exchanges, {wall:.2f} s wall")
    assert summ.events_processed == one["events"] and summ.final_time_ns == one["final"]
    np.testing.assert_array_equal(summ.events_by_kind, one["by_kind"])
    for k in ("generated", "accepted", "completed", "total_service_s", "sink_received", "queue_depth", "active", "events"):
        np.testing.assert_array_equal(stats[k], one["stats"][k], err_msg=k)
    for k in ("routed", "link_entered", "link_packets_sent"):
        np.testing.assert_array_equal(netst[k], one["net"][k], err_msg=k)
    for a, b in zip(sinks, one["sinks"]):
        np.testing.assert_array_equal(a, b)
    assert summ.world == 4 and (59000 < summ.windows < 60100 if not rounds else summ.windows < 200)
    assert 2.5e8 < summ.events_processed < 2.9e8
    assert wall < 240.0


@PROTOCOLS
@pytest.mark.parametrize("world", [2, 5])
def test_sharded_mesh_with_two_links_per_station(world, rounds):
    """An irregular network: every station's RandomRouter chooses between links to the next and to the station seven
    further on (lossy, different jitters), c = 2 with bounded queues on every third station -- most links cross a shard
    boundary somewhere, several per pair of shards.  Both shard protocols against the single windowed engine."""
    from happy_simulator_amd import _native as N
    from happy_simulator_amd.engine import NetworkArrays, StationArrays, StationEngine
    from happy_simulator_amd.sharded import LocalComm, ShardedNetwork

    n, end_ns, seed = 61, 7_000_000_000, 29
    rates = np.array([2.0 + (i % 4) if i % 5 else 0.0 for i in range(n)])
    st = StationArrays(
        n=n, src_kind=np.where(rates > 0, N.SRC_POISSON, N.SRC_NONE).astype(np.uint8), src_rate=np.where(rates > 0, rates, 1.0),
        src_stop_after_ns=np.full(n, -1, np.int64), concurrency=np.array([2 if i % 3 == 0 else 1 for i in range(n)], np.int32),
        svc_kind=np.full(n, N.LAT_EXPONENTIAL, np.uint8), svc_mean_s=np.full(n, 0.04),
        queue_cap=np.array([3 if i % 3 == 0 else -1 for i in range(n)], np.int64), egress=np.full(n, N.EGRESS_NONE, np.uint8))
    net = NetworkArrays(
        egress_kind=np.full(n, N.EGRESS_ROUTER, np.uint8), router_target0=np.arange(0, 2 * n, 2, dtype=np.int32),
        router_target1=np.arange(1, 2 * n, 2, dtype=np.int32), link_of=np.full(n, -1, np.int32),
        link_src=np.repeat(np.arange(n), 2).astype(np.int32),
        link_dst=np.array([(i + (1 if l == 0 else 7)) % n for i in range(n) for l in range(2)], np.int32),
        link_lat_min_s=np.array([0.001 + 0.0005 * (l % 3) for l in range(2 * n)]),
        link_jitter_kind=np.array([N.LAT_EXPONENTIAL if l % 2 else N.LAT_CONSTANT for l in range(2 * n)], np.uint8),
        link_jitter_mean_s=np.array([0.003 if l % 2 else 0.0 for l in range(2 * n)]),
        link_stream_base=np.arange(500, 500 + 2 * n, dtype=np.uint64),
        link_loss_rate=np.array([0.4 + 0.02 * (l % 6) for l in range(2 * n)]))
    with StationEngine(st, mode=N.MODE_SINGLE, horizon_ns=end_ns, seed=seed, log_capacity=1024, network=net) as eng:
        eng.set_debug_flags(16)                                   # the windowed engine, one device
        eng.run_until(end_ns)
        s = eng.summary()
        one = (s.events_processed, s.final_time_ns, tuple(s.events_by_kind), {k: v.tobytes() for k, v in eng.lp_stats().items()},
               {k: v.tobytes() for k, v in eng.net_stats().items()})
    sn = ShardedNetwork.on_gpu(st, net, LocalComm(world), horizon_ns=end_ns, seed=seed, log_capacity=1024, msg_capacity=2048,
                               rounds=rounds is not False, exchange="collective" if rounds == "collective" else "device")
    with sn:
        summ = sn.run_until(end_ns)
        stats, counts, t, cr, netst = sn.collect(n, 2 * n)
    assert (summ.events_processed, summ.final_time_ns, tuple(summ.events_by_kind)) == one[:3]
    assert {k: v.tobytes() for k, v in stats.items()} == one[3]
    assert {k: v.tobytes() for k, v in netst.items()} == one[4]
    assert summ.events_processed > 5000


def test_a_cross_shard_election_that_rests_on_a_stand_in_rank_is_refused():
    """The stand-in tie of tests/test_gpu_ring.py with its two stations in DIFFERENT shards: every shard reports its candidate, the
    host finds two that share the whole lineage key and rank with a stand-in, and refuses (sharded.py run_until)."""
    from happy_simulator_amd import _native as N
    from test_gpu_ring import _stand_in_tie_spec

    spec = _stand_in_tie_spec()
    for rounds in (True, "collective", False):
        with pytest.raises(N.EngineError, match="lock-step tie"):
            _sharded(spec, 2, rounds=rounds)


def test_the_live_exchange_is_for_one_shard_per_process():
    """Round 6: `exchange="live"` (one launch per rank and run, hs_engine_shard_live_*) needs every shard's launch running at the
    same time; virtual shards of ONE process keep the asynchronous rounds (their launches may share a hardware queue) -- asked for
    live, they say so and still equal the single engine.  The live path itself: tests/test_gpu_dist.py (2 and 3 processes)."""
    from happy_simulator_amd.sharded import LocalComm, ShardedNetwork

    spec = SPECS[0]
    one = _single(spec)
    st, net, cap, p = H.ring_arrays(spec)
    with ShardedNetwork.on_gpu(st, net, LocalComm(2), horizon_ns=p["end_ns"], seed=spec["seed"], log_capacity=cap, exchange="live") as sn:
        assert not sn.live and sn.device_exchange
        summ = sn.run_until(p["end_ns"])
    assert summ.events_processed == one["events"] and summ.final_time_ns == one["final"]


def test_fifty_engines_in_one_process_do_not_disturb_the_next_one():
    """Round 6: a FRESH process that creates and destroys ~57 engines (ring_64 in every protocol and world, then two worlds of
    ring_300_c2_cap) and then runs ring_300_c2_cap on three shards.  With the shards' uncached link queues handed back to the runtime
    (hipFree) at every destroy, the last engine's ordinary arrays (routed, link counters) read back as ZEROS although its kernels had
    written them; the process-wide pool of csrc/hs_engine.hip (uncached_alloc: never back to the runtime) keeps the sequence clean."""
    import os, subprocess, sys

    def block(name, worlds):
        return [f"{name}:{w}:{p}" for w in worlds for p in "dcw"]

    seq = block("ring_64", (1, 2, 3, 4)) + block("ring_300_c2_cap", (1, 2)) + ["ring_300_c2_cap:3:d"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "sharded_seq.py"), ",".join(seq)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"n = {len(seq)} bad = 0" in r.stdout, r.stdout[-2000:]


def test_nothing_depends_on_what_the_allocator_hands_out():
    """HS_POISON_ALLOC (csrc/hs_engine.hip dev_alloc): every device allocation filled with 0xAB before use -- single engines and shards
    in every protocol still equal each other (a fresh process, so that the poison is what the engines see)."""
    import os, subprocess, sys

    seq = ["ring_64:2:d", "ring_7_dense:3:c", "ring_300_c2_cap:2:w", "ring_33_fixed_latency:4:d", "ring_300_c2_cap:3:d"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "sharded_seq.py"), ",".join(seq)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HS_POISON_ALLOC="0xAB"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"n = {len(seq)} bad = 0" in r.stdout, r.stdout[-2000:]
