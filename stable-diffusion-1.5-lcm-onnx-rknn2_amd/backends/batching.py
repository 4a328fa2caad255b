"""Queue-level micro-batching inside the worker.

The reference runs one ``self.pipe(...)`` per job and one worker thread per ``worker_id``
(backends/worker_pool.py:73-120, backends/cuda_worker.py:198-252).  On an MI355X a batch-8 sampler pass costs ~3.5x a
batch-1 pass, so jobs that are queued at the same moment and agree on (size, steps, guidance, style) are coalesced
into ONE batched pass here -- below the ``run_job`` boundary, so the pool above is untouched: every caller still
blocks in its own ``run_job`` and gets its own (png, seed) back, and PNG encoding stays on the callers' threads
(zlib releases the GIL), i.e. off the GPU dispatcher.

Batch sizes are restricted to ``sizes`` (default 1, 2, 4, 8) because every batch size owns a captured hipGraph and
its buffers: 5 waiting jobs run as 4 + 1, never as a new 5-wide plan.
"""
from __future__ import annotations

import threading
import time
from collections import deque
from concurrent.futures import Future
from typing import Callable, Hashable, List, Sequence


class MicroBatcher:
    """``submit(key, item) -> Future``; a single dispatcher thread calls ``run_batch(key, [items]) -> [results]``.

    Ordering: the oldest waiting item always leads the next batch (no starvation); younger items with the same key
    join it, items with other keys keep their place.  ``window_ms`` is how long the dispatcher will wait for more
    arrivals when the queue is shorter than ``max_batch`` -- 0 (default) never delays a lone request: coalescing then
    comes only from jobs that queued up while the previous pass was running, which is the loaded case that matters.
    """

    def __init__(self, run_batch: Callable[[Hashable, list], list], max_batch: int = 8, window_ms: float = 0.0,
                 sizes: Sequence[int] = (1, 2, 4, 8), name: str = "lcm-microbatch", lanes: int = 1):
        """lanes > 1: that many dispatcher threads pull from the one queue and call ``run_batch(key, items, lane)`` --
        passes on different lanes overlap on the GPU (pipeline.LcmHipPipeline lanes); with lanes == 1 the callback keeps
        its two-argument form."""
        self.run_batch = run_batch
        self.lanes = max(1, int(lanes))
        # Lanes beyond the first serve the two ENDS of the load range (tools/lanes_sweep.py: images/s @ latency with the same
        # number of requests in flight -- 2: 72 @ 28 ms on two lanes = 75 @ 27 as one batch of 2; 4: 97 @ 41 either way;
        # 8: 114 @ 70 as 4 + 4 against 116 @ 69 as one batch of 8; 16: 127 @ 126 as 8 + 8 on two lanes against 116 one after
        # the other).  With up to `lane_max_waiting` jobs waiting a second pass in flight is as good as batching them and
        # starts at once; in between, lane 0's next (larger) batched pass is the better use of the GPU; with a FULL batch
        # waiting behind a running pass the second lane takes it
        self.lane_max_waiting = 2
        self._lane0_busy = False                    # lane 0 is inside run_batch (a pass, a first-use tune / capture, a style wait)
        self._lane0_since = 0.0
        self.lane0_stall_s = 0.25                   # ... for longer than this: the other lanes stop deferring to it
        self.sizes = sorted(s for s in set(int(x) for x in sizes) if 1 <= s <= max(1, int(max_batch))) or [1]
        self.max_batch = self.sizes[-1]
        self.window = max(0.0, float(window_ms)) / 1e3
        self._q: deque = deque()                    # (key, item, future, t_arrival, burst)
        self._cv = threading.Condition()
        self._closed = False
        self.batches: List[int] = []                # sizes of the batches run so far (telemetry / tests)
        self._threads = [threading.Thread(target=self._loop, args=(i,), name=f"{name}-{i}", daemon=True) for i in range(self.lanes)]
        for t in self._threads:
            t.start()

    def submit(self, key: Hashable, item, burst: bool = False) -> Future:
        """burst: the item belongs to a set that is complete (the single consumer thread of a pool drained it from the queue and
        now blocks until it is served: nothing more can arrive meanwhile) -- what lane 0's pass leaves over goes to the other
        lanes at once instead of waiting for lane 0's next, larger batch."""
        fut: Future = Future()
        with self._cv:
            if self._closed:
                raise RuntimeError("MicroBatcher is closed")
            self._q.append((key, item, fut, time.monotonic(), burst))
            self._cv.notify_all()                   # every lane re-evaluates (a single notify may wake a lane that defers)
        return fut

    def _take(self, lane=0):
        """Called with the lock held and a non-empty queue: pop the next batch (same key as the head)."""
        key = self._q[0][0]
        same = [e for e in self._q if e[0] == key]
        n = max(s for s in self.sizes if s <= len(same))
        if same[0][4] and n < len(same) and (lane > 0 or self.lanes == 1):
            # the tail of a complete set (burst): one pass of the next plan size with the last item repeated (its copies'
            # results are dropped) when that is cheaper than the passes the remainder would otherwise take one after the other
            # -- 7 as one batch of 8 (67 ms) instead of 4 + 2 + 1 (88 ms), 3 as 4 (41) instead of 2 + 1 (47), 6 as 8; 5 stays
            # 4 + 1.  Bit-neutral: a request's bytes do not depend on its batch or its position in it.  Not on lane 0 while
            # other lanes exist: what it leaves over runs NEXT to its pass there (4 || 2 finishes before a padded 8).
            up = min((s for s in self.sizes if s >= len(same)), default=None)
            if up is not None and self._cost(up) <= self._split_cost(len(same)):
                batch = same + [(key, same[-1][1], None, same[-1][3], True)] * (up - len(same))
                ids = {id(e) for e in same}
                self._q = deque(e for e in self._q if id(e) not in ids)
                return key, batch
        batch = same[:n]
        ids = {id(e) for e in batch}
        self._q = deque(e for e in self._q if id(e) not in ids)
        return key, batch

    # what a pass costs by batch size, relative to batch 1 (SD1.5 512x512 4 steps on an MI355X: 19.9 / 27 / 41 / 67 ms); sizes
    # in between are interpolated.  Only ratios matter, and only to the choice above.
    PASS_COST = {1: 1.0, 2: 1.35, 4: 2.05, 8: 3.35}

    def _cost(self, n):
        if n in self.PASS_COST:
            return self.PASS_COST[n]
        return 1.0 + 0.335 * (n - 1)

    def _split_cost(self, n):
        c = 0.0
        while n > 0:
            s = max(x for x in self.sizes if x <= n)
            c, n = c + self._cost(s), n - s
        return c

    def _loop(self, lane=0):
        """A dispatcher thread never dies silently: whatever escapes one round (a bug in the gating, not a failed pass -- those
        go to the waiters' futures) is reported and the thread goes on serving; a dead lane 0 would otherwise leave the other
        lanes deferring to it forever."""
        while True:
            try:
                if not self._round(lane):
                    return
            except BaseException as exc:            # noqa
                import sys
                import traceback
                print(f"[lcm-microbatch] lane {lane}: dispatcher round failed: {exc!r}", file=sys.stderr)
                traceback.print_exc()
                if lane == 0:
                    with self._cv:
                        self._lane0_busy = False
                        self._cv.notify_all()
                time.sleep(0.01)

    def _lane0_alive(self):
        t = self._threads[0] if self._threads else None
        return t is not None and t.is_alive()

    def _round(self, lane):
        """One pass of a dispatcher thread: wait for work, take a batch, run it.  -> False when the batcher is closed and empty."""
        with self._cv:
            while not self._q and not self._closed:
                self._cv.wait()
            if not self._q and self._closed:
                return False
            if lane > 0 and not self._lane0_busy and not self._closed and self._lane0_alive():
                # lane 0 is idle and has been notified as well: the job is its to take.  A lone caller never touches
                # the other lanes (each owns ~1.3 GB of workspace, its own buffers, tune and graph capture on first use)
                self._cv.wait(0.05)
                return True
            if lane > 0 and len(self._q) > self.lane_max_waiting and not self._closed and self._lane0_alive() and not self._q[0][4]:
                # high load: leave the queue to lane 0's next (larger) batch -- unless lane 0 is stuck inside one call
                # (multi-second first-use tune / graph capture, waiting for a style re-merge): then serve.  Woken by
                # lane 0's notify when it comes back for work, not by polling.
                stalled = self._lane0_busy and time.monotonic() - self._lane0_since > self.lane0_stall_s
                # ... or a FULL batch is already waiting behind lane 0's pass: two largest-size passes in flight finish
                # 9 % more images per second than one after the other (126.7 against 116.4 images/s at batch 8,
                # tools/lanes_sweep.py), and the waiting batch is done sooner than if it queued behind the running one
                head = self._q[0][0]
                full = sum(1 for e in self._q if e[0] == head) >= self.max_batch
                if not stalled and not full:
                    self._cv.wait(self.lane0_stall_s)
                    return True
            if self.window > 0:
                head_key, deadline = self._q[0][0], self._q[0][3] + self.window
                while (self._q and sum(1 for e in self._q if e[0] == head_key) < self.max_batch and not self._closed):
                    left = deadline - time.monotonic()
                    if left <= 0:
                        break
                    self._cv.wait(left)
                if not self._q:                     # the lock was released while waiting: another lane took the batch
                    return True
            key, batch = self._take(lane)
            if lane == 0:
                self._lane0_busy, self._lane0_since = True, time.monotonic()
                if self._q and self.lanes > 1:
                    self._cv.notify_all()             # what is left may now go to the other lanes
        items = [e[1] for e in batch]
        try:
            try:
                results = self.run_batch(key, items, lane) if self.lanes > 1 else self.run_batch(key, items)
            finally:
                if lane == 0:
                    with self._cv:
                        self._lane0_busy = False
                        self._cv.notify_all()         # the other lanes re-evaluate their gate
            if len(results) != len(items):
                raise RuntimeError(f"run_batch returned {len(results)} results for {len(items)} items")
            self.batches.append(sum(1 for e in batch if e[2] is not None))
            for e, r in zip(batch, results):
                if e[2] is not None:                # None: a repeated item that filled the plan size (_take)
                    e[2].set_result(r)
        except BaseException as exc:                # every waiter of the failed pass sees the error (reference: the
            for e in batch:                         # exception propagates out of run_job, backends/worker_pool.py:100-113)
                if e[2] is not None and not e[2].done():
                    e[2].set_exception(exc)
        return True

    def close(self, timeout: float = 30.0):
        with self._cv:
            self._closed = True
            self._cv.notify_all()
        for t in self._threads:
            if threading.current_thread() is not t:             # a finalizer may run on a dispatcher thread itself
                t.join(timeout)
        with self._cv:
            for e in self._q:
                if not e[2].done():
                    e[2].set_exception(RuntimeError("MicroBatcher closed before the job ran"))
            self._q.clear()
