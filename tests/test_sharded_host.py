"""CPU: the host path of a partitioned network run -- ShardedNetwork's window protocol, GVT all-reduce(min),
outbox all-to-all, overshoot election -- with virtual shards (LocalComm) and across two processes
(DistComm over gloo, world_size 2).  The GPU engine is replaced by tests/fake_shard.py (a token ring with the same
shard protocol); the real kernels behind the same protocol are covered by tests/test_gpu_sharded.py."""
import os

import numpy as np
import pytest

import fake_shard as F
from happy_simulator_amd.sharded import DistComm, LocalComm, ShardedNetwork, shard_arrays, shard_bounds

N_ST, W, END = 23, 1_000, 400_000


def test_shard_bounds_cover_and_balance():
    for n in (1, 7, 64, 65536):
        for w in (1, 2, 3, 8):
            b = shard_bounds(n, w)
            assert b[0] == 0 and b[-1] == n and len(b) == w + 1
            sizes = np.diff(b)
            assert sizes.max() - sizes.min() <= 1 and (sizes >= 0).all()


def test_shard_arrays_keep_every_link_that_touches_the_shard():
    from happy_simulator_amd import _native as N
    from happy_simulator_amd.engine import NetworkArrays, StationArrays

    n = 10
    st = StationArrays.uniform(n, rate=4.0)
    net = NetworkArrays(
        egress_kind=np.full(n, N.EGRESS_ROUTER, np.uint8), router_target0=np.full(n, -1, np.int32),
        router_target1=np.arange(n, dtype=np.int32), link_of=np.full(n, -1, np.int32),
        link_src=np.arange(n, dtype=np.int32), link_dst=((np.arange(n) + 1) % n).astype(np.int32),
        link_lat_min_s=np.full(n, 1e-3), link_jitter_kind=np.full(n, N.LAT_EXPONENTIAL, np.uint8),
        link_jitter_mean_s=np.full(n, 0.01))
    b = shard_bounds(n, 3)
    seen_out = []
    for r in range(3):
        lo, hi = int(b[r]), int(b[r + 1])
        s, sub = shard_arrays(st, net, lo, hi)
        assert s.n == hi - lo and list(s.stream_base) == list(range(lo, hi))
        assert sub.n_global_lp == n and sub.n_global_links == n
        src, dst = sub.link_src, sub.link_dst
        assert all((lo <= a < hi) or (lo <= d < hi) for a, d in zip(src, dst))
        # each station's router still points at ITS outgoing link, by local index
        for i in range(lo, hi):
            l = sub.router_target1[i - lo]
            assert sub.link_gid[l] == i and src[l] == i
            assert sub.link_stream_base[l] == i
        seen_out += [g for g, a in zip(sub.link_gid, src) if lo <= a < hi]
    assert sorted(seen_out) == list(range(n))           # every link is owned (as outgoing) by exactly one shard


@pytest.mark.parametrize("world", [1, 2, 3, 5])
@pytest.mark.parametrize("sync_every", [1, 7])
def test_virtual_shards_match_single_heap(world, sync_every):
    ev, last, counts = F.reference_run(N_ST, W, END)
    b = shard_bounds(N_ST, world)
    shards = [F.FakeShard(N_ST, r, b, W) for r in range(world)]
    sn = ShardedNetwork(shards, LocalComm(world), window_ns=W, sync_every=sync_every)
    s = sn.run_until(END)
    assert s.events_processed == ev and s.final_time_ns == last
    np.testing.assert_array_equal(np.concatenate([sh.counts for sh in shards]), counts)
    assert s.windows % sync_every == 0 and s.world == world
    assert s.run_seconds >= s.exchange_seconds > 0.0     # host time in the exchange / GVT calls = the reference's barrier time


def _cross_links(bounds):
    rank_of = lambda x: int(np.searchsorted(bounds, x, side="right") - 1)        # noqa: E731
    return [i for i in range(N_ST) if rank_of(i) != rank_of((i + 1) % N_ST)]


@pytest.mark.parametrize("world", [1, 2, 3, 5])
@pytest.mark.parametrize("per_round", [1, 4, 1000])
def test_virtual_shards_with_asynchronous_rounds_match_single_heap(world, per_round):
    """The round protocol of ShardedNetwork (execute below the cross links' bounds; all-to-all messages; all-reduce MAX
    bounds + the "still working" flag; inject) -- the host loop the GPU shards run, on the token ring."""
    ev, last, counts = F.reference_run(N_ST, W, END)
    b = shard_bounds(N_ST, world)
    shards = [F.FakeShard(N_ST, r, b, W) for r in range(world)]
    for sh in shards:
        sh.async_setup(_cross_links(b), per_round)
    sn = ShardedNetwork(shards, LocalComm(world), window_ns=W, sync_every=3, rounds=True)
    s = sn.run_until(END)
    assert s.events_processed == ev and s.final_time_ns == last
    np.testing.assert_array_equal(np.concatenate([sh.counts for sh in shards]), counts)
    assert s.windows % 3 == 0 and s.world == world


def _rank_main(rank, world, port, q, rounds=False):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = DistComm()
    b = shard_bounds(N_ST, world)
    shard = F.FakeShard(N_ST, rank, b, W)
    if rounds:
        shard.async_setup(_cross_links(b), 6)
    sn = ShardedNetwork([shard], comm, window_ns=W, sync_every=5, rounds=rounds)
    s = sn.run_until(END)
    q.put((rank, s.events_processed, s.final_time_ns, s.windows, shard.counts.tolist(), shard.events))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("rounds", [False, True], ids=["windows", "async_rounds"])
def test_two_process_gloo_run_matches_single_heap(rounds):
    import torch.multiprocessing as mp

    ev, last, counts = F.reference_run(N_ST, W, END)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + ((os.getpid() * 7 + (991 if rounds else 0)) % 2000)
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q, rounds)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [g[1] for g in got] == [ev, ev]               # every rank sees the network-wide total
    assert [g[2] for g in got] == [last, last]
    assert got[0][3] == got[1][3]                        # same number of windows on every rank
    np.testing.assert_array_equal(np.array(got[0][4] + got[1][4]), counts)
    assert got[0][5] + got[1][5] == ev                   # and owns only its share of the events


def test_election_ranks_are_network_wide_construction_positions():
    """The last key of the cross-shard election (sharded.ElectionRanks): a candidate's rank is its entity's position in the WHOLE
    network's construction order -- `sources=[...]` as listed, sourceless stations behind every Source, Probes behind all of them
    in `probes=[...]` order -- the table hs_engine_set_stations builds for one engine; a shard's own rank (its filtered, re-based
    order) cannot be compared across shards."""
    from happy_simulator_amd import _native as N
    from happy_simulator_amd.engine import StationArrays
    from happy_simulator_amd.sharded import ElectionRanks

    n = 8
    st = StationArrays.uniform(n, rate=4.0)
    st.src_kind[3] = N.SRC_NONE                                     # a sourceless station
    r = ElectionRanks(st)                                           # no order arrays: the engine's closed form (LP order)
    assert [r.rank(i, 2) for i in range(n)] == [4 * i for i in range(n)]
    assert r.rank(5, 0) == 20 and r.rank(5, 3) == 21 and r.rank(2, 9) == 4 * n + 2 * 4 + 1
    # Sources listed in reverse station order, station 6 with a second Source (slot 1) listed first of all
    st.src_more_kind = np.full((3, n), N.SRC_NONE, np.uint8)
    st.src_more_rate = np.ones((3, n))
    st.src_more_kind[0, 6] = N.SRC_CONSTANT
    st.source_order = np.array([6, 7, 6, 5, 4, 2, 1, 0], np.int32)
    st.source_slot_order = np.array([1, 0, 0, 0, 0, 0, 0, 0], np.uint8)
    st.probe_metric = np.full(n, N.PROBE_NONE, np.uint8)
    st.probe_metric[[1, 6]] = 0
    st.probe_interval_s = np.full(n, 0.5)
    st.probe_order = np.array([6, 1], np.int32)
    r = ElectionRanks(st)
    assert r.rank(6, 3) == 0 and r.rank(7, 2) == 1 and r.rank(6, 2) == 2 and r.rank(0, 2) == 7
    assert r.rank(6, 0) == 0                                        # a departure of station 6: its first-listed Source
    assert r.rank(3, 0) == 8 + 3                                    # sourceless: behind the 8 Sources, by station
    assert r.rank(6, 8) == 8 + n and r.rank(1, 8) == 8 + n + 1      # Probes last, in probes=[...] order
    # the lock-step tie of ADVICE r3: station 1 (shard 0) against station 6 (shard 1) -- both shards call their Source "rank 0"
    assert r.rank(6, 2) < r.rank(1, 2)
