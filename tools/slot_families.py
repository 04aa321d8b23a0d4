"""does ONE captured slot step serve the other 2-tuple model families?  For each family of pygho_amd.models.SpModel: an eager step on the
exactly sized batch, an eager step on the slot and captured replays, from the same parameters -- reports exceptions and the largest
difference of loss / gradients.  python tools/slot_families.py [FAMILY ...]"""
import copy
import json
import os
import sys
import traceback

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import synth  # noqa: E402
from pygho_amd.collate import DeviceGraphStore  # noqa: E402
from pygho_amd.graphs import SlotStep  # noqa: E402
from pygho_amd.honn.SpOperator import parse_precomputekey  # noqa: E402
from pygho_amd.models import SpModel  # noqa: E402
from pygho_amd.slots import BatchSlot  # noqa: E402

dev = torch.device("cuda:0")
fams = sys.argv[1:] or ["NGNN", "SSWL", "DSSGNN", "GNNAK", "SUN"]
G = 48
for fam in fams:
    out = {"family": fam}
    try:
        torch.manual_seed(0)
        model = SpModel(fam, num_layer=2, hiddim=128, act_dtype=torch.bfloat16).to(dev)
        keys = tuple(parse_precomputekey(model))
        rng = np.random.default_rng(3)
        recs = [synth.make_graph(rng, "zinc", 3, keys) for _ in range(256)]
        store = DeviceGraphStore(recs, dev)
        state0 = copy.deepcopy(model.state_dict())
        ids = [np.random.default_rng(7 + k).permutation(256)[:G] for k in range(4)]

        def grads(dd):
            model.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pred = model(dd)
            loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
            loss.backward()
            return loss.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        model.load_state_dict(state0)
        le, ge = grads(store.collate(ids[0]))
        slot = BatchSlot(store, G)
        model.load_state_dict(state0)
        dd = slot.collate(ids[0])
        with slot.rows():
            ls, gs = grads(dd)
        out["slot_vs_exact_loss"] = float((ls - le).abs())
        bad = {k: float((gs[k].float() - ge[k].float()).abs().max() / (ge[k].float().abs().max() + 1e-20)) for k in ge if not torch.equal(gs[k], ge[k])}
        out["slot_vs_exact_grads_differ"] = bad
        # captured
        model.load_state_dict(state0)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True)

        def step(d):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pred = model(d)
            loss = torch.nn.functional.l1_loss(d["y"].unsqueeze(-1), pred.float())
            loss.backward()
            opt.step()
            return loss.detach()
        ss = SlotStep(store, G, step, warmup_ids=ids[0])
        losses = [float(ss.run(i)) for i in ids[1:]]
        out["captured_losses"] = losses
        out["finite"] = bool(np.all(np.isfinite(losses)))
    except Exception as e:
        out["error"] = f"{type(e).__name__}: {e}"
        out["trace"] = traceback.format_exc().splitlines()[-6:]
    print(json.dumps(out), flush=True)
