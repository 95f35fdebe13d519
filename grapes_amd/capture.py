"""hipGraph capture of a step that contains collectives.

A step without collectives is one hipGraph.  RCCL calls stay OUTSIDE the graphs: the step is recorded
as a list  [graph segment, collective, graph segment, collective, ...]  — whenever the step body reaches
a collective (through `run_collective`), the running capture is closed, the collective's closure is
recorded, and a new capture starts in the same memory pool.  Replaying walks the list in order on the
current stream.  This needs every message to have a fixed size and fixed buffers (dist.PartitionedGraph
and dist.GradSync guarantee both) and the body to be free of host reads.
"""
from __future__ import annotations

import gc
from typing import Callable, List, Union

import torch


class SegmentedGraph:
    def __init__(self):
        self.items: List[Union[torch.cuda.CUDAGraph, Callable[[], None]]] = []
        self._cur = None
        self._pool = None
        self._stream = None
        self.capturing = False

    # ---------------------------------------------------------------- recording
    # capture_begin / capture_end are driven directly (not through the torch.cuda.graph context manager, whose
    # __enter__ synchronises and empties the allocator cache): between two segments nothing may be released,
    # earlier segments have the addresses of this step's buffers baked in.
    def _open(self):
        # keep_graph: the hipGraph_t stays readable after instantiation (raw_graphs: step chains re-add its nodes)
        g = torch.cuda.CUDAGraph(keep_graph=True)
        # thread_local: the process group's watchdog thread may touch the runtime while we capture
        g.capture_begin(pool=self._pool, capture_error_mode="thread_local")
        self._cur = g
        self.items.append(g)

    def _close(self):
        self._cur.capture_end()
        self._cur = None

    def run_collective(self, fn: Callable[[], None]):
        """Hook for dist.PartitionedGraph / dist.GradSync: outside a recording the collective just runs."""
        if not self.capturing:
            fn()
            return
        self._close()
        self.items.append(fn)
        self._open()

    def record(self, body: Callable[[], None]):
        """Runs `body` once under capture (nothing executes; collectives are recorded, not issued)."""
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()
        self.items = []
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        if self._stream is None:
            self._stream = torch.cuda.Stream()
        self._stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._stream):
            self.capturing = True
            self._open()
            try:
                body()
            finally:
                self.capturing = False
                self._close()
        torch.cuda.current_stream().wait_stream(self._stream)
        torch.cuda.synchronize()

    # ---------------------------------------------------------------- replay
    def replay(self):
        for it in self.items:
            if isinstance(it, torch.cuda.CUDAGraph):
                it.replay()
            else:
                it()

    def raw_graphs(self) -> List[int]:
        """hipGraph_t handles of the segments, in order — only for a step without collectives (nothing host-driven between
        its segments): what grapes_graph_chain_create takes."""
        if self.num_collectives or self.capturing:
            raise RuntimeError("a step with collectives between its graph segments cannot be chained")
        return [int(it.raw_cuda_graph()) for it in self.items]

    @property
    def num_segments(self) -> int:
        return sum(isinstance(it, torch.cuda.CUDAGraph) for it in self.items)

    @property
    def num_collectives(self) -> int:
        return len(self.items) - self.num_segments

