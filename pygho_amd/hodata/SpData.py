"""
Sparse-layout data helpers with the reference's names and arguments (pygho/hodata/SpData.py:14-200): key parsing, the
per-graph precomputation of the message triples a layer will need (``sp_datapreprocess``) and the batch -> SparseTensor wrapper
(``batch2sparse``).  The index work runs on the device planner (``backend.Spspmm.spspmm_ind`` / ``filterind`` ->
``csrc/plan.hip``) when the graph's tensors live there; batches themselves are collated on the device by
``pygho_amd.collate.DeviceGraphStore`` (reference SpData.py:56-77 ``__inc__``).

Graph / batch containers are plain attribute objects or dicts: torch_geometric's ``Data`` / ``Batch`` classes are not required.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Any, Callable, List, Tuple

import torch

from ..backend.SpTensor import SparseTensor, coalesce
from ..backend.Spspmm import filterind, spspmm_ind
from ..honn.SpOperator import KEYSEP


def parseop(op: str):
    """attribute that counts the rows an operand contributes when graphs are concatenated (reference SpData.py:14-31).
    An unknown operand name RAISES ``NotImplementedError``; the reference returns the tuple ``(NotImplementedError, msg)``
    there (SpData.py:31, evidently a missing ``raise``), which its caller then uses as an attribute name."""
    if op and op[0] == "X":
        return f"num_tuples{op[1:]}"
    if op == "A":
        return "num_edges"
    raise NotImplementedError(f"operator name {op} not implemented now")


def parsekey(key: str) -> Tuple[str, str, int, str, int]:
    """"X___A___1___X___0" -> ("X", "A", 1, "X", 0) (reference SpData.py:34-53)"""
    parts = key.split(KEYSEP)
    assert len(parts) == 5, "key format not match"
    op0, op1, dim1, op2, dim2 = parts
    for op in (op0, op1, op2):
        parseop(op)
    return op0, op1, int(dim1), op2, int(dim2)


def _get(obj: Any, key: str):
    return obj[key] if isinstance(obj, dict) else getattr(obj, key)


def _set(obj: Any, key: str, value) -> None:
    if isinstance(obj, dict):
        obj[key] = value
    else:
        setattr(obj, key, value)


def batch2sparse(batch: Any, keys: List[str] = [""]):
    """wrap a collated batch's ``edge_index / edge_attr`` and ``tupleid{key} / tuplefeat{key} / tupleshape{key}`` as
    SparseTensors ``A`` and ``X{key}`` (reference SpData.py:80-112)"""
    n = int(_get(batch, "num_nodes"))
    ea = _get(batch, "edge_attr")
    _set(batch, "A", SparseTensor(_get(batch, "edge_index"), ea, [n, n] if ea is None else [n, n] + list(ea.shape[1:]), is_coalesced=True))
    for key in keys:
        total = _get(batch, f"tupleshape{key}").sum(dim=0).tolist()
        tf = _get(batch, f"tuplefeat{key}")
        X = SparseTensor(_get(batch, f"tupleid{key}"), tf, shape=total if tf is None else total + list(tf.shape[1:]), is_coalesced=True)
        _set(batch, f"X{key}", X)
    return batch


def sp_datapreprocess(data: Any, tuplesamplers: List[Callable[[Any], SparseTensor]], annotate: List[str] = [""],
                      keys: List[str] = [""]) -> SimpleNamespace:
    """per-graph preprocessing (reference SpData.py:115-171): coalesce the edges, run the tuple samplers, and precompute for
    every key ``op0___op1___dim1___op2___dim2`` the message triples ``acd = filterind(ind(op0), *spspmm_ind(ind(op1), dim1,
    ind(op2), dim2))`` that the layers look up in the datadict."""
    assert len(tuplesamplers) == len(annotate), "number of tuple sampler should match the number of annotate"
    n = int(_get(data, "num_nodes"))
    ei, ea = _get(data, "edge_index"), _get(data, "edge_attr")
    ei, ea = coalesce(ei, ea, "sum")
    out = dict(data) if isinstance(data, dict) else dict(vars(data))
    out.update({"num_nodes": n, "num_edges": ei.shape[1], "x": _get(data, "x"), "edge_index": ei, "edge_attr": ea})
    view = SimpleNamespace(**out)
    for i, sampler in enumerate(tuplesamplers):
        feat = sampler(view)
        out.update({f"tupleid{annotate[i]}": feat.indices, f"tuplefeat{annotate[i]}": feat.values,
                    f"tupleshape{annotate[i]}": torch.tensor(list(feat.sparseshape), dtype=torch.int64).reshape(1, -1),
                    f"num_tuples{annotate[i]}": feat.indices.shape[1]})
    ind_of = lambda op: out[f"tupleid{op[1:]}"] if op[0] == "X" else out["edge_index"]
    for key in keys:
        op0, op1, dim1, op2, dim2 = parsekey(key)
        out[key + f"{KEYSEP}acd"] = filterind(ind_of(op0), *spspmm_ind(ind_of(op1), dim1, ind_of(op2), dim2))
    return SimpleNamespace(**out)
