"""Host reads (planner fetches, .item(), .tolist() of device tensors) that a training step on a FRESH collated batch still makes, per
sparse model family of pygho_amd.models.SpModel, with the call sites."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import traceback
import numpy as np, torch
from pygho_amd import _ops, synth
from pygho_amd.collate import DeviceGraphStore
from pygho_amd.models import SpModel
from pygho_amd.honn.SpOperator import parse_precomputekey
dev = torch.device("cuda:0")
for conv in ("NGNN", "SSWL", "SUN", "GNNAK", "DSSGNN", "I2GNN"):
    try:
        torch.manual_seed(0)
        model = SpModel(conv, num_layer=2, hiddim=64, act_dtype=torch.bfloat16).to(dev)
        keys = tuple(parse_precomputekey(model))
        kind = "i2" if conv == "I2GNN" else "zinc"
        rng = np.random.default_rng(1)
        recs = [synth.make_graph(rng, kind, 3, keys) for _ in range(24)]
        store = DeviceGraphStore(recs, dev)
        def step(dd):
            model.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pred = model(dd)
            torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float()).backward()
        step(store.collate(list(range(8))))
        torch.cuda.synchronize()
        items = []
        oi, ol = torch.Tensor.item, torch.Tensor.tolist
        where = lambda: " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_stack(limit=7)[-5:-1][::-1])
        torch.Tensor.item = lambda self: (items.append("item " + where()) if self.is_cuda else None, oi(self))[1]
        torch.Tensor.tolist = lambda self: (items.append("tolist " + where()) if self.is_cuda else None, ol(self))[1]
        f0 = _ops.FETCHES[0]
        try:
            step(store.collate([9, 3, 17, 20, 5, 5, 11, 23]))
        finally:
            torch.Tensor.item, torch.Tensor.tolist = oi, ol
        print(conv, "keys", keys, "planner fetches", _ops.FETCHES[0] - f0, "item/tolist", len(items), flush=True)
        for it in items:
            print("   ", it, flush=True)
    except Exception as e:
        print(conv, "FAILED", type(e).__name__, str(e)[:200], flush=True)
