"""
Seeded synthetic batches with the shape of the reference's data pipeline output.

The reference's ``pygho.hodata`` (PyG ``Data``/``Batch`` subclasses, tuple
samplers, pre-transform cache) is OUT OF SCOPE for this build;
what the operator path consumes is its *output contract*, which this module
reproduces on the host with numpy:

  * per-graph tuple sampling mirroring ``KhopSampler`` (hodata/SpTupleSampler.py:91-126)
    and ``I2Sampler`` (:129-173): 2-tuples (i, j) with dist(i, j) <= hop, or
    3-tuples (i, j, k) for every directed edge (i, j) and every k within `hop`
    of i or j; tuple features = shortest-path distances;
  * per-graph precompute ``acd = filterind(tar, *spspmm_ind(ind1, dim1, ind2, dim2))``
    for every precompute key (hodata/SpData.py:163-171) -- computed here by
    ``host_plan_acd`` (a direct enumeration, not the oracle);
  * block-diagonal collation with the offsets of ``SpHoData.__inc__`` /
    ``__cat_dim__`` (hodata/SpData.py:60-77): node ids += #nodes so far, tuple
    ids += tupleshape so far, ``acd`` rows += (#tuples(op0), #tuples(op1),
    #edges(op2)) so far, everything concatenated along the nnz axis;
  * the padded dense form of ``hodata/MaData.py:25-255`` for the masked path.

ZINC itself is not available offline; graph statistics follow SURVEY.md 8(d):
ZINC-shape molecules (mean 23.2 nodes / 24.9 bonds) and I2-shape random graphs
(mean 18.8 nodes, average degree 3.33).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

KEYSEP = "___"
I64 = np.int64


# --------------------------------------------------------------------------
# graph generators
# --------------------------------------------------------------------------
def _zinc_like_graph(rng: np.random.Generator) -> Tuple[int, np.ndarray]:
    """molecule-like graph: random tree (parent among the previous 4 nodes, max
    degree 3) plus ring-closing chords up to ~n*24.9/23.2 undirected edges, max degree 4."""
    n = int(np.clip(np.rint(rng.normal(23.2, 4.5)), 9, 37))
    deg = np.zeros(n, dtype=np.int64)
    adj = np.zeros((n, n), dtype=bool)
    for v in range(1, n):
        lo = max(0, v - 4)
        cand = [u for u in range(lo, v) if deg[u] < 3]
        if not cand:
            cand = [u for u in range(0, v) if deg[u] < 4]
        u = cand[int(rng.integers(len(cand)))]
        adj[u, v] = adj[v, u] = True
        deg[u] += 1
        deg[v] += 1
    target = int(np.rint(n * 24.9 / 23.2))
    tries = 0
    while adj.sum() // 2 < target and tries < 50:
        tries += 1
        u = int(rng.integers(n))
        span = int(rng.integers(4, 7))          # 5- and 6-rings mostly
        v = u + span - 1
        if v >= n or adj[u, v] or deg[u] >= 4 or deg[v] >= 4:
            continue
        adj[u, v] = adj[v, u] = True
        deg[u] += 1
        deg[v] += 1
    return n, adj


def _gnm_graph(rng: np.random.Generator) -> Tuple[int, np.ndarray]:
    """I2-GNN counting-set shape: G(n, m) with average degree 3.33, made connected."""
    n = int(np.clip(np.rint(rng.normal(18.8, 3.0)), 10, 30))
    m = int(np.rint(n * 3.33 / 2))
    adj = np.zeros((n, n), dtype=bool)
    perm = rng.permutation(n)
    for t in range(1, n):                        # random spanning tree first
        u, v = perm[t], perm[int(rng.integers(t))]
        adj[u, v] = adj[v, u] = True
    while adj.sum() // 2 < m:
        u, v = int(rng.integers(n)), int(rng.integers(n))
        if u != v:
            adj[u, v] = adj[v, u] = True
    return n, adj


def _hop_distance(adj: np.ndarray, hop: int) -> np.ndarray:
    """dist[i, j] = shortest path length if <= hop else -1."""
    n = adj.shape[0]
    dist = np.full((n, n), -1, dtype=np.int64)
    reach = np.eye(n, dtype=bool)
    dist[reach] = 0
    frontier = reach
    for h in range(1, hop + 1):
        nxt = (frontier.astype(np.int32) @ adj.astype(np.int32)) > 0
        new = nxt & ~reach
        dist[new] = h
        reach |= new
        frontier = new
    return dist


# --------------------------------------------------------------------------
# host planner (direct enumeration; independent of oracle/)
# --------------------------------------------------------------------------
def _pack(ind: np.ndarray, base: int) -> np.ndarray:
    out = np.zeros(ind.shape[1], dtype=I64)
    for r in range(ind.shape[0]):
        out = out * base + ind[r]
    return out


def host_plan_acd(tar_ind: np.ndarray, ind1: np.ndarray, dim1: int,
                  ind2: np.ndarray, dim2: int) -> np.ndarray:
    """
    All triples (a, c, d) with ind1[dim1, c] == ind2[dim2, d] and
    tar_ind[:, a] == concat(ind1[:, c] minus dim1, ind2[:, d] minus dim2);
    columns in lexicographic (a, c, d) order.  Equivalent to the reference's
    ``filterind(tar, *spspmm_ind(ind1, dim1, ind2, dim2))`` (Spspmm.py:57-222)
    up to the (non-canonical) order within equal `a`.
    """
    base = int(max(tar_ind.max(initial=0), ind1.max(initial=0), ind2.max(initial=0))) + 1
    k2 = ind2[dim2]
    order2 = np.argsort(k2, kind="stable")
    k2s = k2[order2]
    k1 = ind1[dim1]
    lo = np.searchsorted(k2s, k1, "left")
    cnt = np.searchsorted(k2s, k1, "right") - lo
    c = np.repeat(np.arange(ind1.shape[1], dtype=I64), cnt)
    first = np.cumsum(cnt) - cnt
    d = order2[lo[c] + (np.arange(c.shape[0], dtype=I64) - first[c])]
    key = _pack(np.concatenate((np.delete(ind1, dim1, 0)[:, c], np.delete(ind2, dim2, 0)[:, d])), base)
    tkey = _pack(tar_ind, base)
    pos = np.searchsorted(tkey, key)
    pos_c = np.minimum(pos, max(tkey.shape[0] - 1, 0))
    ok = (tkey[pos_c] == key) if tkey.shape[0] else np.zeros(key.shape, bool)
    a, c, d = pos_c[ok], c[ok], d[ok]
    order = np.lexsort((d, c, a))
    return np.stack((a[order], c[order], d[order])).astype(I64)


def parse_key(key: str) -> Tuple[str, str, int, str, int]:
    """precompute key "{op0}___{op1}___{dim1}___{op2}___{dim2}" (honn/SpOperator.py:135,
    hodata/SpData.py:30-46)."""
    parts = key.split(KEYSEP)
    assert len(parts) == 5, "key format not match"
    return parts[0], parts[1], int(parts[2]), parts[3], int(parts[4])


# --------------------------------------------------------------------------
# per-graph records and collation
# --------------------------------------------------------------------------
@dataclass
class GraphRecord:
    num_nodes: int
    x: np.ndarray                 # (n,) atom type
    edge_index: np.ndarray        # (2, e) coalesced, both directions
    edge_attr: np.ndarray         # (e,) bond type
    tupleid: np.ndarray           # (sd, t) coalesced
    tuplefeat: np.ndarray         # (t,) or (t, f) integer distance features
    acd: Dict[str, np.ndarray] = field(default_factory=dict)
    y: float = 0.0


def make_graph(rng: np.random.Generator, kind: str = "zinc", hop: int = 3,
               keys: Tuple[str, ...] = ("X___X___1___A___0",)) -> GraphRecord:
    if kind == "zinc":
        n, adj = _zinc_like_graph(rng)
    elif kind == "i2":
        n, adj = _gnm_graph(rng)
    else:
        raise ValueError(kind)
    ei = np.stack(np.nonzero(adj)).astype(I64)           # row-major => sorted, coalesced
    bond = rng.integers(1, 4, size=(n, n))
    bond = np.triu(bond, 1)
    bond = bond + bond.T
    ea = bond[ei[0], ei[1]].astype(I64)
    x = rng.integers(0, 28, size=n).astype(I64)
    dist = _hop_distance(adj, hop)
    if kind == "zinc":
        tid = np.stack(np.nonzero(dist >= 0)).astype(I64)          # (i, j), sorted
        tfeat = dist[tid[0], tid[1]].astype(I64)
    else:
        within = dist >= 0
        rows = []
        for e in range(ei.shape[1]):
            i, j = int(ei[0, e]), int(ei[1, e])
            ks = np.nonzero(within[i] | within[j])[0]
            rows.append(np.stack((np.full_like(ks, i), np.full_like(ks, j), ks)))
        tid = np.concatenate(rows, axis=1).astype(I64)               # sorted by (i, j, k)
        full = _hop_distance(adj, n)
        tfeat = np.stack((full[tid[0], tid[2]], full[tid[1], tid[2]]), axis=-1).astype(I64)
    rec = GraphRecord(n, x, ei, ea, tid, tfeat, y=float(rng.normal()))
    for key in keys:
        op0, op1, dim1, op2, dim2 = parse_key(key)
        pick = lambda op: tid if op[0] == "X" else ei
        rec.acd[key] = host_plan_acd(pick(op0), pick(op1), dim1, pick(op2), dim2)
    return rec


@dataclass
class HostBatch:
    """block-diagonal batch on the host (numpy); `to_datadict` moves it to a device."""
    num_graphs: int
    num_nodes: int
    x: np.ndarray
    batch: np.ndarray
    edge_index: np.ndarray
    edge_attr: np.ndarray
    tupleid: np.ndarray
    tuplefeat: np.ndarray
    acd: Dict[str, np.ndarray]
    y: np.ndarray
    nodes_per_graph: np.ndarray

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.shape[1])

    @property
    def num_tuples(self) -> int:
        return int(self.tupleid.shape[1])

    def num_messages(self, key: str) -> int:
        return int(self.acd[key].shape[1])


def collate(records: List[GraphRecord]) -> HostBatch:
    """block-diagonal concatenation with the increments of hodata/SpData.py:60-77."""
    keys = list(records[0].acd.keys())
    node_off = tup_off = edge_off = 0
    xs, bs, eis, eas, tids, tfs, ys, ns = [], [], [], [], [], [], [], []
    acds: Dict[str, List[np.ndarray]] = {k: [] for k in keys}
    for g, r in enumerate(records):
        xs.append(r.x)
        bs.append(np.full(r.num_nodes, g, dtype=I64))
        eis.append(r.edge_index + node_off)
        eas.append(r.edge_attr)
        tids.append(r.tupleid + node_off)
        tfs.append(r.tuplefeat)
        ys.append(r.y)
        ns.append(r.num_nodes)
        for k in keys:
            op0, op1, _, op2, _ = parse_key(k)
            inc = np.array([[tup_off if op[0] == "X" else edge_off] for op in (op0, op1, op2)], dtype=I64)
            acds[k].append(r.acd[k] + inc)
        node_off += r.num_nodes
        tup_off += r.tupleid.shape[1]
        edge_off += r.edge_index.shape[1]
    return HostBatch(
        num_graphs=len(records), num_nodes=node_off,
        x=np.concatenate(xs), batch=np.concatenate(bs),
        edge_index=np.concatenate(eis, axis=1), edge_attr=np.concatenate(eas),
        tupleid=np.concatenate(tids, axis=1), tuplefeat=np.concatenate(tfs, axis=0),
        acd={k: np.concatenate(v, axis=1) for k, v in acds.items()},
        y=np.asarray(ys, dtype=np.float32), nodes_per_graph=np.asarray(ns, dtype=I64))


def replicate(hb: HostBatch, times: int) -> HostBatch:
    """tile a host batch `times` times block-diagonally (cheap way to reach
    HBM-sized working sets from a few thousand distinct graphs)."""
    if times == 1:
        return hb
    n, t, e = hb.num_nodes, hb.num_tuples, hb.num_edges
    r = np.arange(times, dtype=I64)
    def rep_idx(a, step):
        return (a[None, ...] + (r * step).reshape((-1,) + (1,) * a.ndim))
    acd = {}
    for k, v in hb.acd.items():
        op0, op1, _, op2, _ = parse_key(k)
        inc = np.array([t if op[0] == "X" else e for op in (op0, op1, op2)], dtype=I64)
        acd[k] = np.concatenate([v + (inc * i)[:, None] for i in range(times)], axis=1)
    return HostBatch(
        num_graphs=hb.num_graphs * times, num_nodes=n * times,
        x=np.tile(hb.x, times), batch=rep_idx(hb.batch, hb.num_graphs).reshape(-1),
        edge_index=np.concatenate([hb.edge_index + n * i for i in range(times)], axis=1),
        edge_attr=np.tile(hb.edge_attr, times),
        tupleid=np.concatenate([hb.tupleid + n * i for i in range(times)], axis=1),
        tuplefeat=np.concatenate([hb.tuplefeat] * times, axis=0),
        acd=acd, y=np.tile(hb.y, times), nodes_per_graph=np.tile(hb.nodes_per_graph, times))


def make_batch(num_graphs: int, kind: str = "zinc", seed: int = 0, hop: int = 3,
               keys: Optional[Tuple[str, ...]] = None) -> HostBatch:
    if keys is None:
        keys = ("X___X___1___A___0",) if kind == "zinc" else ("X___X___2___A___0",)
    rng = np.random.default_rng(seed)
    return collate([make_graph(rng, kind, hop, keys) for _ in range(num_graphs)])


def shard_ranges(weights: np.ndarray, world_size: int) -> List[Tuple[int, int]]:
    """contiguous graph ranges per rank, balanced by cumulative weight (e.g. the
    per-graph message count), SURVEY.md 8(e)."""
    w = np.asarray(weights, dtype=np.float64)
    cum = np.concatenate(([0.0], np.cumsum(w)))
    total = cum[-1]
    cuts = [0]
    for r in range(1, world_size):
        cuts.append(int(np.searchsorted(cum, total * r / world_size, side="left")))
    cuts.append(len(w))
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return [(cuts[i], cuts[i + 1]) for i in range(world_size)]


# --------------------------------------------------------------------------
# device-side dictionaries
# --------------------------------------------------------------------------
def to_datadict(hb: HostBatch, device, kind: str = "zinc") -> dict:
    """mirror of ``batch2sparse`` + ``batch.to_dict()`` (hodata/SpData.py:80-112,
    example/minimal.py:144-145) for a synthetic batch."""
    import torch
    from .backend.SpTensor import SparseTensor
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    n = hb.num_nodes
    ei, ea = t(hb.edge_index), t(hb.edge_attr)
    tid, tf = t(hb.tupleid), t(hb.tuplefeat)
    sd = tid.shape[0]
    dd = {
        "x": t(hb.x), "batch": t(hb.batch), "num_graphs": hb.num_graphs,
        "y": t(hb.y), "num_nodes": n,
        "A": SparseTensor(ei, ea, [n, n] + list(ea.shape[1:]), is_coalesced=True),
        "X": SparseTensor(tid, tf, [n] * sd + list(tf.shape[1:]), is_coalesced=True),
    }
    for k, v in hb.acd.items():
        dd[k + KEYSEP + "acd"] = t(v)
    return dd


def to_dense_datadict(hb: HostBatch, device) -> dict:
    """mirror of ``batch2dense`` + ``batch.to_dict()`` (hodata/MaData.py:146-255) for a synthetic 2-tuple batch: padded
    integer MaskedTensors x (b, n), A (b, n, n) and X (b, n, n), built by the device builders of ``hodata.MaData``."""
    import torch
    from .hodata import to_dense_adj, to_dense_x
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    counts = np.bincount(hb.batch, minlength=hb.num_graphs)
    ptr = np.concatenate(([0], np.cumsum(counts)))
    n = int(counts.max())
    local = lambda idx: idx - ptr[hb.batch[idx]]
    eb, tb = hb.batch[hb.edge_index[0]], hb.batch[hb.tupleid[0]]
    return {
        "x": to_dense_x(t(hb.x), t(ptr)), "num_graphs": hb.num_graphs, "y": t(hb.y),
        "A": to_dense_adj(t(np.stack((local(hb.edge_index[0]), local(hb.edge_index[1])))), t(eb), t(hb.edge_attr), n, hb.num_graphs),
        "X": to_dense_adj(t(np.stack((local(hb.tupleid[0]), local(hb.tupleid[1])))), t(tb), t(hb.tuplefeat), n, hb.num_graphs),
    }


def make_dense_batch(num_graphs: int, seed: int = 0, hidden: int = 128, nmax: Optional[int] = None,
                     dtype=np.float32, clip_nodes: Optional[int] = None):
    """padded dense form for the masked path (hodata/MaData.py:25-255): node mask
    (b, n), A (b, n, n, d) zero-filled with the adjacency as mask, X (b, n, n, d)
    with the node-mask outer product as mask (every real pair valid, as
    ``spdsampler`` produces, hodata/MaTupleSampler.py:11-32)."""
    rng = np.random.default_rng(seed)
    graphs = [_zinc_like_graph(rng) for _ in range(num_graphs)]
    if clip_nodes is not None:               # tiny fixtures: keep the first few atoms of each molecule
        graphs = [(min(n, clip_nodes - g % 3), adj) for g, (n, adj) in enumerate(graphs)]
        graphs = [(n, adj[:n, :n]) for n, adj in graphs]
    nm = nmax or max(n for n, _ in graphs)
    nodemask = np.zeros((num_graphs, nm), dtype=bool)
    amask = np.zeros((num_graphs, nm, nm), dtype=bool)
    for g, (n, adj) in enumerate(graphs):
        nodemask[g, :n] = True
        amask[g, :n, :n] = adj
    xmask = nodemask[:, :, None] & nodemask[:, None, :]
    A = rng.standard_normal((num_graphs, nm, nm, hidden)).astype(dtype) * amask[..., None]
    X = rng.standard_normal((num_graphs, nm, nm, hidden)).astype(dtype) * xmask[..., None]
    x = rng.standard_normal((num_graphs, nm, hidden)).astype(dtype) * nodemask[..., None]
    return dict(nodemask=nodemask, Amask=amask, Xmask=xmask, A=A, X=X, x=x)
