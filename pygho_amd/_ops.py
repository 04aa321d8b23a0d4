"""
Torch-level wrappers over the C ABI, one namespace: the implementation lives in four modules by layer --

    plans.py     int32 / CSR index plans, deferred range checks, integer planner primitives
    segment.py   segment-kernel launches, message plans, autograd Functions of the sparse operators
    masked.py    dense (MaskedTensor) path: masked fill / reduce / broadcast, pair kernels, masked batched contraction
    blocks.py    fused dense steps around the aggregation: BatchNorm + activation, cast arena, row-block Linear, tuple blocks

-- and this module re-exports all of it under the historical name (`from pygho_amd import _ops`).  The modules carry A/B
switches (`USE_*`, thresholds) that tests and tools flip through this namespace: an assignment `_ops.NAME = value` is forwarded
to the module that defines NAME, so the code that reads its own global sees it.  Everything here runs on ROCm device memory;
there is no CPU path.
"""
import sys
import types

from . import blocks, masked, plans, segment

_PARTS = (plans, segment, masked, blocks)
_OWNER = {}
for _m in _PARTS:
    for _k, _v in vars(_m).items():
        if _k.startswith("__"):
            continue
        _OWNER.setdefault(_k, _m)             # the first module in dependency order is the definer; later ones re-import
        globals()[_k] = _v


class _Namespace(types.ModuleType):
    def __setattr__(self, name, value):
        owner = _OWNER.get(name)
        if owner is not None:
            for m in _PARTS:                  # the definer and every module that imported the name by value
                if name in vars(m):
                    setattr(m, name, value)
        super().__setattr__(name, value)


sys.modules[__name__].__class__ = _Namespace
