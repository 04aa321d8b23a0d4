"""one captured step over a fixed-capacity slot, a fresh shuffled batch every step (pygho_amd.graphs.SlotStep): ms per step at the
reference's batch size (128) and at 1024 graphs, next to the eager fresh-batch loop.  python tools/slot_bench.py [--graphs 128,1024]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import synth  # noqa: E402
from pygho_amd.collate import BatchPrefetcher, DeviceGraphStore  # noqa: E402
from pygho_amd.graphs import SlotStep  # noqa: E402
from pygho_amd.ngnn import SpModel  # noqa: E402

KEY = "X___X___1___A___0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", default="128,1024")
    ap.add_argument("--store", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--no-eager", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    recs = [synth.make_graph(rng, "zinc", 3, (KEY,)) for _ in range(min(args.store, 2048))]
    store = DeviceGraphStore(recs * max(1, args.store // len(recs)), dev)
    out = {}

    def make_step(model, opt):
        def step(dd):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pred = model(dd)
            loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
            loss.backward()
            opt.step()
            return loss.detach()
        return step

    for graphs in [int(g) for g in args.graphs.split(",")]:
        gen = np.random.default_rng(1)
        ids = [gen.permutation(store.num_graphs)[:graphs] for _ in range(args.steps + 10)]
        torch.manual_seed(0)
        model = SpModel(1, args.layers, args.hidden, act_dtype=torch.bfloat16).to(dev)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True, fused=True)
        t0 = time.perf_counter()
        ss = SlotStep(store, graphs, make_step(model, opt))
        torch.cuda.synchronize()
        build_s = time.perf_counter() - t0
        for k, b in enumerate(ids):
            if k == 10:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            loss = ss.run(b)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        # host time of the upload alone
        t0 = time.perf_counter()
        for b in ids[:100]:
            ss.slot.layout(np.asarray(b))
        lay_us = (time.perf_counter() - t0) / 100 * 1e6
        res = {"captured_fresh_batch_ms_per_step": ms, "replays": ss.replays, "eager_fallbacks": ss.eager_steps, "loss": float(loss),
               "capture_s": build_s, "host_layout_us": lay_us, "capacities": {str(k): v for k, v in ss.slot.caps.items()},
               "mean_sizes": {str(f): float(np.mean(store.h_len[f])) * graphs for f in ss.slot.fams}}
        if not args.no_eager:
            torch.manual_seed(0)
            model = SpModel(1, args.layers, args.hidden, act_dtype=torch.bfloat16).to(dev)
            step = make_step(model, torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True))
            n = 0
            for k, dd in enumerate(BatchPrefetcher(store, [torch.from_numpy(b) for b in ids[:70]])):
                if k == 10:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                step(dd)
                n += 1
            torch.cuda.synchronize()
            res["eager_fresh_batch_ms_per_step"] = (time.perf_counter() - t0) / (n - 10) * 1e3
        out[f"bs{graphs}"] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
