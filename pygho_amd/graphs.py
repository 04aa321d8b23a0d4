"""
HIP-graph capture of a whole training step.

Below ~2000 ZINC-shape graphs per GPU the NGNN step is bound by host issue time (~270 launches, ~4 ms of Python / ATen /
ctypes work against ~2 ms of GPU work at 1024 graphs).  Every kernel of the library is launched on torch's current stream
through the C ABI and no entry point synchronises or allocates, so a step on static inputs can be captured once and replayed:
4.6 -> 2.9 ms per step at 1024 graphs (MI355X).  Requirements (all met by `pygho_amd.ngnn.SpModel` on a fixed `datadict`):

* the same tensors (batch, indices, targets) are reused across replays.  New FEATURE / TARGET values may be copied into them;
  the INDEX PATTERN must stay what it was at capture: the graph bakes in the plans built from the index tensors during warm-up
  (CSR pointers, permutations, grid sizes, output row counts).  Pass the index tensors as ``static_indices`` and ``replay()``
  raises if one of them was written to since capture; a new batch pattern needs a new capture (or the eager
  ``collate.BatchPrefetcher`` path);
* the plans are built before capture (the warm-up steps below do that: plan construction reads sizes back to the host);
* the optimizer is created with ``capturable=True``; gradients are reset with ``set_to_none=True`` inside the step;
* no host read-back (``.item()``, ``print(loss)``) inside the step -- return tensors and read them after ``replay()``.

Several captured steps (one per fixed mini-batch, ``examples/minimal.py``): with four or more captured SpModel steps sharing one
capturable AdamW, an EAGER kernel launched between replays made later replays return NaN in round 2 unless
``torch.cuda.synchronize()`` (device-wide; a stream synchronisation was not enough) ran after it; ``examples/minimal.py`` therefore
forms its statistics on the host (``float(loss)``).  Round 2 attributed this to the PyTorch 2.10 / ROCm 7.2 build; the plain
``torch.nn`` reduction committed as ``tools/repro_graph_nan.py`` does NOT show it on this build (0 non-finite losses of 120 with
2, 4 and 6 captured steps, with and without the synchronisation), so the cause is not pinned down -- it may as well lie in how
this package's step interacts with capture.  Keep the device synchronisation after eager work between replays until it is.
"""
from typing import Any, Callable, Iterable

import torch


class GraphedStep:
    """``GraphedStep(fn)`` warms ``fn`` up on a side stream, captures one call into a HIP graph and replays it."""

    def __init__(self, fn: Callable[[], Any], warmup: int = 3, static_indices: Iterable[torch.Tensor] = ()):
        assert torch.cuda.is_available(), "HIP graph capture needs the ROCm device"
        self._static = [(t, t._version) for t in static_indices]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.output = fn()

    def replay(self) -> Any:
        """run the captured step once more; returns the (static) output tensors of the captured call."""
        for t, v in self._static:
            if t._version != v:
                raise RuntimeError("GraphedStep.replay: an index tensor was modified after capture; the captured launches still "
                                   "use the plans of the old pattern -- capture a new GraphedStep for the new batch pattern")
        self.graph.replay()
        from . import _ops
        _ops.invalidate_cast_arenas()      # a captured optimizer step moved the parameters without moving their version counters
        return self.output
