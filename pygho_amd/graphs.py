"""
HIP-graph capture of a whole training step.

Below ~2000 ZINC-shape graphs per GPU the NGNN step is bound by host issue time (~270 launches, ~4 ms of Python / ATen /
ctypes work against ~2 ms of GPU work at 1024 graphs).  Every kernel of the library is launched on torch's current stream
through the C ABI and no entry point synchronises or allocates, so a step on static inputs can be captured once and replayed:
4.6 -> 2.9 ms per step at 1024 graphs (MI355X).  Requirements (all met by `pygho_amd.ngnn.SpModel` on a fixed `datadict`):

* the same tensors (batch, indices, targets) are reused across replays.  New FEATURE / TARGET values may be copied into them;
  the INDEX PATTERN must stay what it was at capture: the graph bakes in the plans built from the index tensors during warm-up
  (CSR pointers, permutations, grid sizes, output row counts).  Pass the index tensors as ``static_indices`` and ``replay()``
  raises if one of them was written to since capture.  A NEW batch every step -- the reference's loop, example/minimal.py:141-149 --
  is served by ``SlotStep`` below: one capture over a fixed-capacity ``slots.BatchSlot`` whose sizes are read on the device;
* the plans are built before capture (the warm-up steps below do that: plan construction reads sizes back to the host);
* the optimizer is created with ``capturable=True`` (and ``fused=True``: the foreach implementation is ~20 small launches per replay,
  0.2 ms of a 1.2 ms step); gradients are reset with ``set_to_none=True`` inside the step;
* no host read-back (``.item()``, ``print(loss)``) inside the step -- return tensors and read them after ``replay()``.

Several captured steps (one per fixed mini-batch, ``examples/minimal.py``) may share one model and one capturable optimizer, with
eager work between their replays and no device synchronisation (``tests/test_gpu_layers.py::
test_four_captured_steps_with_eager_work_between_replays``: 0 non-finite losses of 200 replays).

The NaN that rounds 2-3 reported for that regime is traced (round 4; ``tools/bisect_graph_nan.py``, ``tools/repro_graph_nan2.py``,
outputs under ``profiles/r04_graph_nan.md``).  It is NOT a race between replays: it needs no second graph and no eager work, and it
survives a device synchronisation after every replay.  It is not in this package's kernels either: swapping single modules shows that
it appears exactly when a node-level ``Linear`` takes torch's stock ``nn.Linear`` under bf16 autocast inside the captured step
(``PYGHO_ARENA_LINEAR=0``, or ``module._pygho_stock = True`` on ``lin_tupleinit1`` / ``poolmlp.lins.0`` alone: 45-47 of 50 replays
non-finite; the same steps run eagerly: 0 of 212), and 40 lines of plain torch reproduce it without this package: ONE captured
training step of ``Embedding -> nn.Linear -> nn.BatchNorm1d -> SiLU -> nn.Linear`` under bf16 autocast goes non-finite within a few
replays (57 of 60) while eager execution of the same step does not (0 of 60).  In the reduction the NaN needs autocast (f32: 0 of 60),
the aten batch_norm behind the autocast Linear (Identity or a hand-written affine map: 0 of 60) and a non-zero learning rate (lr 0: 0
of 60 -- a gradient goes bad, not the forward); MIOpen on / off, BatchNorm in training or eval mode, the autocast cache on / off, AdamW
or SGD, and the row count (3000 ... 28509, multiples of 256 included) make no difference.  Which gradient inside torch's autocast
``nn.Linear`` / ``batch_norm`` backward goes bad under replay on this PyTorch 2.10 / ROCm 7.2 build is not identified.  In rounds 2-3
the node-level Linears of ``SpModel`` still were autocast ``nn.Linear`` calls; since the end of round 3 they read the cast arena
(``honn.utils.Linear`` -> ``_ArenaLinearFn``, explicit 16-bit GEMMs outside autocast), which is why the symptom vanished.
Consequence for users: do not capture a step that sends 16-bit autocast through torch's own ``nn.Linear`` (``PYGHO_ARENA_LINEAR=0``
provokes it, as would a user module outside this package's layers) -- run such a step eagerly.
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


class SlotStep:
    """ONE captured training step that serves every mini-batch of ``batch_graphs`` graphs drawn from a ``DeviceGraphStore``
    (the reference's loop: a fresh shuffled batch per step, example/minimal.py:119, :141-149).

    ``step_fn(datadict)`` is the whole step -- zero_grad(set_to_none=True), forward, loss, backward, optimizer step (capturable) --
    and returns the tensors to read afterwards (e.g. the loss).  It is warmed up on the slot, then captured together with the
    slot's collate kernel; ``run(graph_ids)`` uploads the batch's offsets (one small asynchronous copy) and replays the graph.  A
    batch that does not fit the slot's capacities (or has another number of graphs) runs ``step_fn`` eagerly on
    ``store.collate(graph_ids)`` instead (counted in ``eager_steps``): same model, same optimizer, same result semantics.

    The captured step and an eager step on the SAME slot contents compute bit-identical results; against an eager step on the
    exactly sized batch the results are bit-identical as well wherever a reduction's partition does not depend on the row count
    (all BatchNorm / weight-gradient / embedding-gradient folds are written that way; ``tests/test_gpu_slots.py`` pins it)."""

    def __init__(self, store, batch_graphs: int, step_fn: Callable[[dict], Any], warmup_ids=None, warmup: int = 3,
                 capacities=None, capacity_sigmas: float = 4.5, sync=None):
        """`sync`: the `parallel.FlatGradSync` that `step_fn` exchanges its gradient through (data-parallel training, BASELINE config
        4).  With RCCL the collectives -- issued on a side stream from backward hooks -- are captured INSIDE the step, so a rank's
        whole step incl. its share of the all-reduce is one graph launch; every rank must then run the same number of captured /
        eager steps in the same order (a batch that does not fit falls back to an eager step, which exchanges too: the ranks stay
        matched).  A host-staged backend (gloo) cannot be captured: the object then runs EVERY step eagerly (`captured` False,
        `why_eager` says so) -- same results."""
        from . import _ops
        from .slots import BatchSlot
        assert torch.cuda.is_available(), "HIP graph capture needs the ROCm device"
        self.store, self.step_fn, self.sync = store, step_fn, sync
        self.eager_steps = self.replays = 0
        self.captured, self.why_eager = True, None
        if sync is not None and not sync.capturable:
            import torch.distributed as dist
            self.captured = False
            self.why_eager = (f"process-group backend {dist.get_backend(sync.group)!r}: a host-staged collective cannot be captured into a "
                              "HIP graph; every step runs eagerly")
            self.slot, self.graph, self.output = None, None, None
            return
        self.slot = BatchSlot(store, batch_graphs, capacities, capacity_sigmas)
        if warmup_ids is None:
            warmup_ids = self._first_fitting()
        slot, dd = self.slot, self.slot.datadict
        if not slot.upload(warmup_ids):
            raise ValueError("SlotStep: the warm-up batch does not fit the slot")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), slot.rows():
            slot.launch()
            slot.reset_caches()
            for _ in range(warmup):
                step_fn(dd)
        torch.cuda.current_stream().wait_stream(side)
        # what the warm-up derived from the slot's arrays (narrowed copies, lookups, reciprocal counts ...) is dropped: the captured
        # step derives it again INSIDE the graph, so every replay derives it for the batch that is in the slot
        slot.reset_caches()
        calls0 = sync.allreduce_calls if sync is not None else 0
        self.graph = torch.cuda.CUDAGraph()
        with slot.rows(), torch.cuda.graph(self.graph):
            slot.launch()
            self.output = step_fn(dd)
        self._calls_per_step = (sync.allreduce_calls - calls0) if sync is not None else 0     # collectives inside ONE captured step
        _ops.invalidate_cast_arenas()

    def _first_fitting(self):
        import numpy as np
        rng = np.random.default_rng(0)
        for _ in range(64):
            ids = rng.permutation(self.store.num_graphs)[:self.slot.g]
            if ids.shape[0] == self.slot.g and self.slot.fits(ids):
                return ids
        raise ValueError("SlotStep: no random batch of the store fits the slot's capacities")

    def run(self, graph_ids) -> Any:
        """one training step on the batch `graph_ids`; returns step_fn's outputs (the captured call's static tensors, or the
        eager call's own)"""
        from . import _ops
        slot = self.slot
        n = len(graph_ids)
        if self.captured and n == slot.g and slot.upload(graph_ids):
            self.graph.replay()
            self.replays += 1
            if self.sync is not None:
                self.sync.allreduce_calls += self._calls_per_step      # the replay issued them; no Python ran to count
            _ops.invalidate_cast_arenas()          # the captured optimizer step moved the parameters behind their version counters
            return self.output
        self.eager_steps += 1
        return self.step_fn(self.store.collate(graph_ids))
