"""CPU stand-in for a GPU shard (tests only): same protocol surface as happy_simulator_amd.sharded.GpuShard --
outbox / inbox rows, GVT slots, window / inject / final / overshoot -- driving a token ring: every station starts
with one token at a station-specific time; a token that arrives at station i at time t is counted there and sent on
to station (i + 1) % n, arriving at t + hop_ns(i).  Lets the host path of a partitioned run (ShardedNetwork +
DistComm over gloo, world_size 2) be exercised without a GPU."""
import heapq

import numpy as np
import torch

INF = np.iinfo(np.int64).max


def hop_ns(i, W):
    return W + (i * 7919) % (3 * W)          # >= W: W is the lookahead


def first_token_ns(i):
    return 1_000 + (i * 104729) % 50_000


class FakeShard:
    def __init__(self, n, rank, bounds, W, msg_capacity=64):
        self.n, self.rank, self.world = n, rank, len(bounds) - 1
        self.bounds = np.asarray(bounds, np.int64)
        self.lo, self.hi = int(bounds[rank]), int(bounds[rank + 1])
        self.W = W
        self.cap = msg_capacity
        row = 1 + 5 * msg_capacity
        self.outbox = torch.zeros((self.world, row), dtype=torch.int64)
        self.inbox = torch.zeros((self.world, row), dtype=torch.int64)
        self.gvt = torch.zeros(2, dtype=torch.int64)
        self.cand = torch.zeros(8, dtype=torch.int64)

    def begin(self, end_ns):
        self.end = end_ns
        self.heap = [(first_token_ns(i), i) for i in range(self.lo, self.hi)]
        heapq.heapify(self.heap)
        self.events = 0
        self.counts = np.zeros(self.hi - self.lo, np.int64)
        self.last = 0
        self.wend = [-1, -1]
        self.gvt[0], self.gvt[1] = INF, 0
        self.outbox.zero_()
        self.launches = 1

    def _rank_of(self, station):
        return int(np.searchsorted(self.bounds, station, side="right") - 1)

    def _process(self, t, i):
        self.events += 1
        self.counts[i - self.lo] += 1
        self.last = max(self.last, t)

    def window(self, k):
        prev, gvt = self.wend[(k + 1) & 1], int(self.gvt[(k + 1) & 1])
        base = max(prev + 1, gvt)
        wend = self.end if base > self.end - (self.W - 1) else base + self.W - 1
        self.wend[k & 1] = wend
        sent_min = INF
        while self.heap and self.heap[0][0] <= wend:
            t, i = heapq.heappop(self.heap)
            self._process(t, i)
            j, ta = (i + 1) % self.n, t + hop_ns(i, self.W)
            assert ta > wend                                   # lookahead: never inside the current window
            sent_min = min(sent_min, ta)
            r = self._rank_of(j)
            if r == self.rank:
                heapq.heappush(self.heap, (ta, j))
            else:
                c = int(self.outbox[r, 0])
                assert c < self.cap
                self.outbox[r, 1 + 5 * c:6 + 5 * c] = torch.tensor([ta, t, 0, (j << 32) | i, 0])
                self.outbox[r, 0] = c + 1
        nxt = self.heap[0][0] if self.heap else INF
        self.gvt[k & 1] = min(int(self.gvt[k & 1]), nxt, sent_min)
        self.launches += 1

    def inject(self, k):
        for r in range(self.world):
            for c in range(int(self.inbox[r, 0])):
                ta, _, _, w3, _ = (int(x) for x in self.inbox[r, 1 + 5 * c:6 + 5 * c])
                j = w3 >> 32
                assert self.lo <= j < self.hi
                heapq.heappush(self.heap, (ta, j))
        self.outbox[:, 0] = 0
        self.gvt[(k + 1) & 1] = INF

    # ---- asynchronous exchange rounds (GpuShard.async_setup / round / inject_async / round_done) -----------------
    def async_setup(self, cross_gid, max_events):
        """cross_gid: the links (link i = station i -> i + 1) that cross a shard boundary, the same list on every rank."""
        self.cross = [int(g) for g in cross_gid]
        self.xbounds = torch.zeros(len(self.cross) + 1, dtype=torch.int64)
        self.in_bound = {g: 0 for g in self.cross if self._rank_of((g + 1) % self.n) == self.rank}
        self.max_events = max_events

    def round(self):
        # every message a neighbour has not sent yet arrives at or after its link's bound: process strictly below the minimum
        H = min(self.in_bound.values()) if self.in_bound else INF
        done = 0
        while self.heap and self.heap[0][0] < H and self.heap[0][0] <= self.end and done < self.max_events:
            t, i = heapq.heappop(self.heap)
            self._process(t, i)
            done += 1
            j, ta = (i + 1) % self.n, t + hop_ns(i, self.W)
            r = self._rank_of(j)
            if r == self.rank:
                heapq.heappush(self.heap, (ta, j))
            else:
                c = int(self.outbox[r, 0])
                assert c < self.cap
                self.outbox[r, 1 + 5 * c:6 + 5 * c] = torch.tensor([ta, t, 0, (j << 32) | i, 0])
                self.outbox[r, 0] = c + 1
        nxt = min(self.heap[0][0] if self.heap else INF, H)          # nothing happens here before `nxt`
        for k, g in enumerate(self.cross):
            mine = self._rank_of(g) == self.rank
            self.xbounds[k] = (INF if nxt == INF else nxt + hop_ns(g, self.W)) if mine else -INF - 1
        self.xbounds[len(self.cross)] = 1 if nxt <= self.end else 0
        self.launches += 1

    def inject_async(self):
        for r in range(self.world):
            for c in range(int(self.inbox[r, 0])):
                ta, _, _, w3, _ = (int(x) for x in self.inbox[r, 1 + 5 * c:6 + 5 * c])
                j = w3 >> 32
                assert self.lo <= j < self.hi
                heapq.heappush(self.heap, (ta, j))
        self.outbox[:, 0] = 0
        for k, g in enumerate(self.cross):
            if g in self.in_bound:
                self.in_bound[g] = int(self.xbounds[k])

    def round_done(self):
        return int(self.xbounds[len(self.cross)]) == 0

    def gvt_slot(self, k):
        return self.gvt[(k & 1):(k & 1) + 1]

    def progress(self, k_last):
        return self.wend[k_last & 1]

    def final(self, k):
        if self.heap:
            t, i = self.heap[0]
            self.cand[:] = torch.tensor([1, t, 0, i, 0, 0, i, 0])
        else:
            self.cand[:] = torch.tensor([0, INF, 0, 0, 0, 0, 0, 0])

    def overshoot(self, lp_local):
        t, i = heapq.heappop(self.heap)
        assert i - self.lo == lp_local
        self._process(t, i)

    def totals(self):
        by_kind = np.zeros(11, np.int64)
        by_kind[0] = self.events
        return {"events": self.events, "by_kind": by_kind, "completed": 0, "sink_records": 0,
                "max_final_ns": self.last, "launches": self.launches}

    def close(self):
        pass


def reference_run(n, W, end_ns):
    """Single-heap run of the same token ring, incl. the one event beyond end (core/simulation.py:472)."""
    heap = [(first_token_ns(i), i) for i in range(n)]
    heapq.heapify(heap)
    counts = np.zeros(n, np.int64)
    events, last = 0, 0
    while heap:
        t, i = heapq.heappop(heap)
        events += 1
        counts[i] += 1
        last = t
        if t > end_ns:
            break
        heapq.heappush(heap, (t + hop_ns(i, W), (i + 1) % n))
    return events, last, counts
