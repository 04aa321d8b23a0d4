#!/usr/bin/env python3
"""
Headline benchmark: BASELINE.json metric "graphs/sec + 2-tuple msg-edges/sec, ZINC NGNN; HBM GB/s fraction of
roofline" on config[1] "NGNNConv 2-tuple sparse spspmm on ZINC, hidden=128 bf16, 1xMI355X".

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one full training step (forward, backward, gradient all-reduce when N > 1, AdamW) of the 6-layer NGNN
of example/minimal.py (hidden 128, bf16 activations / f32 master weights) over a NEW shuffled synthetic ZINC-shape batch
EVERY step -- the reference's loop, example/minimal.py:141-149.  The dataset (--store-graphs, default 16384 distinct graphs
per rank) is resident in HBM as a graph-local int32 store; a step's batch is collated on the device together with every
index plan (int32 / CSR views, transposed groupings, scatter plans), one batch ahead on a side stream, so the step's inputs
are resident when it starts and nothing crosses PCIe.  Every rank draws its own stream of --graphs graphs per step (weak
scaling, graphs shard with no data-path collective); value = graphs of all ranks over all timed steps / max-over-ranks time.
`--resident-batch` times ONE resident batch instead (the headline of rounds 1-4; also `regimes.resident_ms_per_step`).

`regimes` (N == 1): the resident-batch step, and the launch-bound sizes 128 / 1024 graphs -- eager on a resident batch, eager on
a fresh shuffled batch every step (`eager_fresh_batch_ms_per_step`), one resident batch captured into a HIP graph, and ONE
captured step over a fixed-capacity batch slot that serves a different shuffled batch every step
(`captured_fresh_batch_ms_per_step`, pygho_amd.graphs.SlotStep).

`configs` (N == 1): the other measured configurations of BASELINE.json in the same line -- config 5 (I2-shape 3-tuple spspmm
launch, d = 256 bf16, and an I2Conv layer step), config 3 (mamamm X A / X Y at (1024, 37, 37, 128) bf16 and a SUNConv DD layer step)
and one forward + backward of the other shipped sparse layers -- each kernel with HIP-event time, algorithmic bytes, roofline
fraction and committed PMC traffic (tools/bench_configs.py; rocprofv3 summaries under profiles/).

Alongside: msg-edges/s and the HBM roofline fraction of the spspmm kernel that carries most of the step's time, measured live
with HIP events around every launch in the timed region -- with 16-bit activations the FORWARD launch of a layer
(seg_fused_fwd_kernel: the aggregation with the layer's Linear -> BatchNorm -> act formed inside it, csrc/seg_fused.hip; bytes =
what has to move: x, out, the stored H, indices), with the by-tuple backward's seg_gmr_fast_kernel under `roofline.other` and every
spspmm launch of the step (both of them + the by-edge scatter) under `roofline.spspmm_all_launches`; and the CPU baseline = the reference's ATen op sequence (oracle/aten_port.py) on the host cores
for a bounded sample of the same workload (rank 0, N == 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

KEY = "X___X___1___A___0"
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--graphs", type=int, default=8192, help="graphs per GPU per step")
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--store-graphs", type=int, default=0,
                    help="graphs in the rank's device-resident store the batches are drawn from (0: twice the graphs drawn per step, i.e. "
                         "16384 at the default batch size)")
    ap.add_argument("--distinct", type=int, default=0,
                    help="distinct graphs generated for the store (0: every graph of the store is distinct; fewer are tiled)")
    ap.add_argument("--resident-batch", action="store_true",
                    help="time ONE resident batch every step (rounds 1-4's headline; profiling and A/B runs) instead of a new batch per step")
    ap.add_argument("--optimizer", default="fused", choices=["fused", "foreach"], help="AdamW implementation (same update rule)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-regimes", action="store_true", help="skip the fresh-batch / small-batch side measurements")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-kernel figures of BASELINE configs 3 and 5 and the layer steps")
    ap.add_argument("--global-stream", action="store_true",
                    help="N > 1: ONE global batch of N x --graphs graphs (same seed on every rank), sharded into contiguous graph "
                         "ranges balanced by message count (parallel.shard_ranges) instead of one independent batch per rank")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend of the N > 1 run: nccl (= RCCL over xGMI, the product path) or gloo (host-staged; with "
                         "--ranks-share-gpu it lets the N > 1 branch execute on a one-GPU box)")
    ap.add_argument("--ranks-share-gpu", action="store_true",
                    help="every rank computes on cuda:0 (RCCL refuses two ranks on one device: use --dist-backend gloo).  A functional run of "
                         "the N > 1 code path, not a scaling measurement")
    ap.add_argument("--collate-gate", default="none", choices=["readout", "none"],
                    help="where the next batch's collate kernel may start: `none` = the moment it is queued (shipped), `readout` = from the "
                         "first graph-level module of the step in flight (A/B, tools/gate_ab.sh: the window of small launches is shorter than "
                         "the kernel, which then lands on the first backward launch -- 0.1-0.2 ms WORSE per step)")
    ap.add_argument("--captured", action="store_true",
                    help="the whole step of every rank -- collation of the new batch, forward, backward, the gradient all-reduce, AdamW -- as "
                         "ONE captured HIP graph over a fixed-capacity batch slot (pygho_amd.graphs.SlotStep with the FlatGradSync inside; "
                         "RCCL only).  BASELINE config 4's operating point: python bench.py --gpus 8 --graphs 1024 --global-stream --captured")
    ap.add_argument("--event-steps", type=int, default=4,
                    help="HIP events around every aggregation launch (the `roofline` / `kernels` figures) in every N-th step of the timed "
                         "region: an event pair costs the timeline ~12 us per launch (the kernel behind a record starts ~6 us late: "
                         "profiles/r06_event_overhead_ab.txt), 0.2 ms per step with all ~18 timed launches of every step (1 = rounds 1-5)")
    ap.add_argument("--cpu-graphs", type=int, default=128)
    ap.add_argument("--cpu-seconds", type=float, default=40.0, help="bound of the CPU baseline's four legs together (each stops early at a quarter of it)")
    return ap.parse_args()


def host_cpu():
    """CPU model and PHYSICAL core count of this host from `lscpu`, plus the CPUs this process may run on."""
    import subprocess
    info = {"model": None, "physical_cores": None, "logical_cpus": os.cpu_count(),
            "usable_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()}
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=20).stdout
        kv = {l.split(":", 1)[0].strip(): l.split(":", 1)[1].strip() for l in txt.splitlines() if ":" in l}
        info["model"] = kv.get("Model name")
        info["physical_cores"] = int(kv["Core(s) per socket"]) * int(kv["Socket(s)"])
        info["threads_per_core"] = int(kv.get("Thread(s) per core", "1"))
    except Exception as e:                        # lscpu missing: the counts above still stand
        info["lscpu_error"] = f"{type(e).__name__}: {e}"
    return info


def cpu_baseline(args, seed):
    """The reference's op sequence on the host cores, by BASELINE.md section 3's protocol: the same 6-layer NGNN train step (f32,
    oracle/aten_port.py = the ATen sequence of pygho/backend/Spspmm.py:307-321) on the same seeded generator, with
    torch.set_num_threads(1) and with ALL physical cores, median of 10 steps after 2 warm-ups, on the reference's batch size (128
    graphs, example/minimal.py:119) and -- bounded -- on a 1024-graph batch.  A leg that would exceed its share of --cpu-seconds
    stops early and says how many steps its median is over."""
    from oracle import aten_port as P
    from pygho_amd import synth
    cpu = host_cpu()
    phys = cpu["physical_cores"] or cpu["logical_cpus"] or 1
    all_cores = max(1, min(phys, cpu["usable_cpus"] or phys))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    legs = []

    def leg(graphs, threads, warm, runs, budget_s):
        hb = synth.make_batch(graphs, "zinc", seed=seed)
        x, ea, tid, tf = t(hb.x), t(hb.edge_attr), t(hb.tupleid), t(hb.tuplefeat)
        acd, batch, y = t(hb.acd[KEY]), t(hb.batch), t(hb.y)
        torch.manual_seed(0)
        model = P.NGNNPort(args.hidden, args.layers)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
        torch.set_num_threads(threads)

        def step():
            opt.zero_grad()
            pred = model(x, ea, tid, tf, acd, batch, hb.num_graphs)
            loss = torch.nn.functional.l1_loss(y.unsqueeze(-1), pred)
            loss.backward()
            opt.step()

        t_leg = time.perf_counter()
        for _ in range(warm):
            step()
        times = []
        while len(times) < runs and (len(times) < 1 or time.perf_counter() - t_leg < budget_s):
            t0 = time.perf_counter()
            step()
            times.append(time.perf_counter() - t0)
        med = float(np.median(times))
        legs.append({"graphs": hb.num_graphs, "msg_edges": hb.num_messages(KEY), "threads": threads, "warmups": warm,
                     "steps_in_median": len(times), "median_s_per_step": med, "graphs_per_s": hb.num_graphs / med})
        return hb, acd

    share = args.cpu_seconds / 4.0
    hb128, acd128 = leg(args.cpu_graphs, 1, 2, 10, share)
    leg(args.cpu_graphs, all_cores, 2, 10, share)
    leg(1024, all_cores, 1, 3, share)
    leg(1024, 1, 0, 1, share)                 # bounded: ONE un-warmed step (tens of seconds on one core); labelled by its counts
    best = max(legs, key=lambda r: r["graphs_per_s"])
    torch.set_num_threads(best["threads"])
    # forward-only spspmm rate for the msg-edges figure (the reference's batch size)
    Xv, Av = torch.randn(hb128.num_tuples, args.hidden), torch.randn(hb128.num_edges, args.hidden)
    for _ in range(2):
        P.spspmm_values(Xv, Av, acd128, hb128.num_tuples)
    ts = []
    for _ in range(10):
        t1 = time.perf_counter()
        P.spspmm_values(Xv, Av, acd128, hb128.num_tuples)
        ts.append(time.perf_counter() - t1)
    return {"value": best["graphs_per_s"], "unit": "graphs/s", "cores": best["threads"], "kind": "port",
            "cpu_model": cpu["model"], "physical_cores": cpu["physical_cores"], "logical_cpus": cpu["logical_cpus"],
            "usable_cpus": cpu["usable_cpus"], "protocol": "BASELINE.md section 3: 1 thread and all physical cores, median after warm-ups, f32",
            "legs": legs,
            "sample": f"median of {best['steps_in_median']} train steps (after {best['warmups']} warm-ups) of the same {args.layers}-layer NGNN "
                      f"(h={args.hidden}, f32) on a {best['graphs']}-graph ZINC-shape batch ({best['msg_edges']} msg-edges), torch-CPU ATen op "
                      f"sequence of pygho/backend/Spspmm.py:307-321 (oracle/aten_port.py), {best['threads']} threads on {cpu['model']} "
                      f"({cpu['physical_cores']} physical cores); every (batch, threads) leg is in `legs`",
            "spspmm_fwd_msg_edges_per_sec": hb128.num_messages(KEY) / float(np.median(ts))}


KERNEL_SOURCES = {"seg_gmr_fast_kernel": ("common.h", "seg_reduce.hip"), "seg_fused_fwd_kernel": ("common.h", "seg_fused.hip"),
                  "seg_dual_kernel": ("common.h", "seg_dual.hip")}


def kernel_source_hash(kernel: str = "seg_gmr_fast_kernel") -> str:
    """sha256 over the sources of a kernel: a committed PMC traffic figure is only reported for the code it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES[kernel.split("<")[0]]:
        h.update(open(os.path.join(REPO, "pygho_amd", "csrc", name), "rb").read())
    return h.hexdigest()


def committed_traffic(config, kernel: str = "seg_gmr_fast_kernel<bf16,SUM,BOTH>"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (tools/collect_traffic.py), or (None, reason) when
    no file matches this configuration, this kernel AND the current kernel sources."""
    import glob
    want = kernel_source_hash(kernel)
    reason = "no profiles/*_traffic*.json for this configuration and kernel"
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "*_traffic*.json")), reverse=True):
        try:
            tj = json.load(open(f))
        except Exception:
            continue
        if tj.get("config") != config or tj.get("kernel", "").split("<")[0] != kernel.split("<")[0]:
            continue
        if tj.get("kernel_source_sha256") != want:
            reason = f"{os.path.basename(f)} was measured on other kernel sources (stale)"
            continue
        return tj["traffic_bytes_per_launch"], os.path.basename(f)
    return None, reason


def side_regimes(args, dev):
    """Outside `value`: the regimes the headline step does not show.  (a) a NEW batch every step -- the reference's loop
    (example/minimal.py:142-160) -- with device collation and plan preparation one batch ahead on a side stream;
    (b) the reference's own batch size (128, example/minimal.py:119) and 1024 graphs, eager and as a captured HIP graph."""
    from pygho_amd import synth
    from pygho_amd.collate import BatchPrefetcher, DeviceGraphStore
    from pygho_amd.graphs import GraphedStep
    from pygho_amd.ngnn import SpModel
    out = {}
    act = torch.bfloat16 if args.dtype == "bf16" else None

    def make_step(model, opt, dd_of):
        def step(dd=None):
            dd = dd_of() if dd is None else dd
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=act is not None):
                pred = model(dd)
            loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
            loss.backward()
            opt.step()
            return loss.detach()
        return step

    # (a) ONE resident batch every step (rounds 1-4's headline regime; the headline is now a new batch every step)
    rng = np.random.default_rng(0)
    recs = [synth.make_graph(rng, "zinc", 3, (KEY,)) for _ in range(min(2048, args.graphs))]
    store = DeviceGraphStore(recs * max(1, args.graphs // len(recs)), dev)
    torch.manual_seed(0)
    model = SpModel(1, args.layers, args.hidden, act_dtype=act).to(dev)
    step = make_step(model, torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True), None)
    dd = store.collate(np.random.default_rng(1).permutation(store.num_graphs)[:args.graphs])
    for _ in range(6):
        step(dd)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(20):
        step(dd)
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / 20 * 1e3
    out["resident_ms_per_step"] = ms
    out["resident_graphs_per_s"] = args.graphs / ms * 1e3
    out["resident_note"] = (f"20 timed steps on ONE {args.graphs}-graph batch that stays resident with its index plans (the headline of rounds 1-4); "
                            "the headline `value` is measured with a NEW batch every step")
    del store, model, step, dd
    gen = torch.Generator().manual_seed(0)
    # (b) small batches
    from pygho_amd.graphs import SlotStep
    small_store = DeviceGraphStore(recs * max(1, 4096 // len(recs)), dev)
    for graphs in (128, 1024):
        hb = synth.make_batch(graphs, "zinc", seed=7)
        dd = synth.to_datadict(hb, dev)
        res = {}
        for mode in ("eager", "hipgraph"):
            torch.manual_seed(0)
            model = SpModel(1, args.layers, args.hidden, act_dtype=act).to(dev)
            # the fused multi-tensor AdamW both times; captured: its capturable form (device-side step counters).  (Rounds 4-5 captured
            # the foreach implementation: ~0.2 ms of small launches per replay, 1.24 -> 1.05 ms at 128 graphs, profiles/r06_captured_adamw_ab.txt)
            optim = (torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True) if mode == "eager"
                     else torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True, fused=True))
            step = make_step(model, optim, lambda: dd)
            if mode == "eager":
                run = step
                for _ in range(5):
                    run()
            else:
                gs = GraphedStep(step, warmup=3)
                run = gs.replay
                run()
            torch.cuda.synchronize(dev)
            reps = 50
            t0 = time.perf_counter()
            for _ in range(reps):
                loss = run()
            torch.cuda.synchronize(dev)
            res[mode] = (time.perf_counter() - t0) / reps * 1e3
            assert bool(torch.isfinite(loss))
        # eager again, but every step a different shuffled batch collated from a resident store (what a training loop does)
        torch.manual_seed(0)
        model = SpModel(1, args.layers, args.hidden, act_dtype=act).to(dev)
        step = make_step(model, torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True), None)
        sids = [torch.randperm(small_store.num_graphs, generator=gen)[:graphs] for _ in range(60)]
        n = 0
        for k, fd in enumerate(BatchPrefetcher(small_store, sids)):
            if k == 10:
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
            step(fd)
            n += 1
        torch.cuda.synchronize(dev)
        res["fresh"] = (time.perf_counter() - t0) / (n - 10) * 1e3
        # ONE captured step over a fixed-capacity batch slot, a different shuffled batch EVERY step (graphs.SlotStep: the collate kernel
        # and the step in one HIP graph, row counts read on the device; per batch one small upload + one replay)
        torch.manual_seed(0)
        model = SpModel(1, args.layers, args.hidden, act_dtype=act).to(dev)
        ss = SlotStep(small_store, graphs, make_step(model, torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True, fused=True), None))
        cids = [torch.randperm(small_store.num_graphs, generator=gen)[:graphs].numpy() for _ in range(210)]
        for k, ids in enumerate(cids):
            if k == 10:
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
            loss = ss.run(ids)
        torch.cuda.synchronize(dev)
        res["captured_fresh"] = (time.perf_counter() - t0) / (len(cids) - 10) * 1e3
        assert bool(torch.isfinite(loss))
        # the same captured slot step with the data-parallel gradient exchange INSIDE the graph (parallel.FlatGradSync, two ranges issued on
        # a side stream from backward hooks) over a ONE-rank RCCL group: what every rank of `--gpus N --graphs 1024 --captured` replays
        # (BASELINE config 4's operating point; with one rank the collective's transport is trivial, its launch and stream joins are not)
        res["captured_sync"] = res["captured_sync_note"] = None
        try:
            res["captured_sync"], res["captured_sync_note"] = captured_under_sync(args, dev, small_store, graphs, cids, act)
        except Exception as e:
            res["captured_sync_note"] = f"{type(e).__name__}: {e}"
        out[f"bs{graphs}"] = {"graphs": graphs, "eager_ms_per_step": res["eager"], "hipgraph_ms_per_step": res["hipgraph"],
                              "captured_under_sync_ms_per_step": res["captured_sync"], "captured_under_sync_note": res["captured_sync_note"],
                              "eager_fresh_batch_ms_per_step": res["fresh"],
                              "captured_fresh_batch_ms_per_step": res["captured_fresh"],
                              "captured_fresh_batch_graphs_per_s": graphs / res["captured_fresh"] * 1e3,
                              "captured_fresh_batch_replays": ss.replays, "captured_fresh_batch_eager_fallbacks": ss.eager_steps,
                              "slot_capacities": {str(k): v for k, v in ss.slot.caps.items()},
                              "eager_graphs_per_s": graphs / res["eager"] * 1e3, "hipgraph_graphs_per_s": graphs / res["hipgraph"] * 1e3,
                              "eager_fresh_batch_graphs_per_s": graphs / res["fresh"] * 1e3}
    out["small_batch_note"] = ("same model and full train step; eager / hipgraph on one resident batch, eager_fresh_batch on 50 timed steps that "
                               "each take a different shuffled batch collated on the device from a resident graph store (BatchPrefetcher); 128 graphs is the reference's batch size "
                               "(example/minimal.py:119); hipgraph = the whole step captured once on ONE batch (pygho_amd.graphs.GraphedStep) and replayed; "
                               "captured_fresh_batch = ONE captured step over a fixed-capacity batch slot serving 200 different shuffled batches "
                               "(pygho_amd.graphs.SlotStep: a batch that would not fit the capacities runs eagerly and is counted)")
    return out


def captured_under_sync(args, dev, store, graphs, id_batches, act):
    """ms per step of ONE captured slot step that carries the FlatGradSync exchange (RCCL, this process as the only rank of a group
    created for the measurement when none exists) over `id_batches` (the first 10 untimed)"""
    import socket
    from pygho_amd.graphs import SlotStep
    from pygho_amd.ngnn import SpModel
    from pygho_amd.parallel import FlatGradSync
    own_group = not dist.is_initialized()
    if own_group:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        torch.manual_seed(0)
        model = SpModel(1, args.layers, args.hidden, act_dtype=act).to(dev)
        sync = FlatGradSync(model.parameters(), overlap=True, buckets=2)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True, fused=True)

        def step(dd):
            sync.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=act is not None):
                pred = model(dd)
            loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
            loss.backward()
            sync.sync()
            opt.step()
            return loss.detach()
        ss = SlotStep(store, graphs, step, sync=sync)
        for k, ids in enumerate(id_batches):
            if k == 10:
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
            loss = ss.run(ids)
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / (len(id_batches) - 10) * 1e3
        assert bool(torch.isfinite(loss))
        note = (f"{ss.replays} replays, {ss.eager_steps} eager fallbacks, {ss._calls_per_step} RCCL all-reduce ranges inside the graph per step, "
                f"{sync.allreduce_calls} collectives in all; one-rank group ({dist.get_backend()})")
        sync.close()
        return ms, note
    finally:
        if own_group:
            dist.destroy_process_group()


def visible_gpu_count():
    """GPUs this process may use, WITHOUT touching the HIP runtime (torch.cuda.device_count() falls back to hipGetDeviceCount when
    amdsmi is missing, which initialises the runtime in the parent): the visibility variables if set, else the KFD topology
    (nodes with SIMDs are GPUs).  None when neither source is readable -- the child ranks then fail on their own."""
    import glob
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    n, seen = 0, False
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(line.split()[:2] for line in open(f) if len(line.split()) >= 2)
        except OSError:
            continue
        seen = True
        n += int(props.get("simd_count", "0")) > 0
    if not seen and not os.path.isdir("/sys/class/kfd"):
        return 0                                             # no KFD driver node at all: no AMD GPU on this host
    return n if seen else None


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` outside a launcher: start N ranks as a CHILD `torch.distributed.run` (one process per
    GPU, RCCL rendezvous on 127.0.0.1) before this process has touched the GPU, pass its output through and return its
    exit code.  Nothing is re-exec'ed, and the parent makes no HIP runtime call (the device count comes from the environment / sysfs)."""
    import socket
    import subprocess
    have = visible_gpu_count()                           # from the environment / sysfs: no runtime call, the parent stays off the GPU
    need = 1 if args.ranks_share_gpu else args.gpus
    if args.ranks_share_gpu and args.dist_backend == "nccl":
        print("bench.py: --ranks-share-gpu needs --dist-backend gloo (RCCL refuses two ranks on one device)", file=sys.stderr)
        return 2
    if have is not None and have < need:
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs the MI355X (the HIP path has no CPU fallback)"
    if args.ranks_share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ          # under torch.distributed.run even a single rank goes through RCCL
    # stdout carries ONE JSON line: libraries that print to the C-level stdout (RCCL's version banner at communicator set-up)
    # are sent to stderr for the rest of the run, the line itself is written to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from pygho_amd import _native, _ops, synth
    from pygho_amd.ngnn import SpModel
    from pygho_amd.parallel import FlatGradSync
    _native.lib()

    # ---- the dataset of this rank: a device-resident int32 graph store.  EVERY step takes a NEW shuffled batch from it -- the
    # reference's loop (example/minimal.py:141-149) -- collated on the device together with its index plans, one batch ahead on a
    # side stream (collate.BatchPrefetcher).  The inputs of a step are resident in HBM when the step starts; nothing crosses PCIe.
    from pygho_amd.collate import BatchPrefetcher, DeviceGraphStore
    from pygho_amd.parallel import shard_ranges
    global_stream = args.global_stream and world > 1
    per_step = args.graphs * (world if global_stream else 1)                 # graphs drawn per step from THIS rank's store
    store_graphs = args.store_graphs if args.store_graphs > 0 else 2 * per_step
    distinct = min(args.distinct if args.distinct > 0 else store_graphs, store_graphs)
    rng = np.random.default_rng(1000 if global_stream else 1000 + rank)      # (global stream: the same store on every rank)
    recs = [synth.make_graph(rng, "zinc", 3, (KEY,)) for _ in range(distinct)]
    recs = (recs * ((store_graphs + distinct - 1) // distinct))[:store_graphs]
    store = DeviceGraphStore(recs, dev)
    msgs_of = np.asarray(store.h_len[("acd", KEY)])
    id_rng = np.random.default_rng(2000 if global_stream else 2000 + rank)
    n_batches = args.warmup + args.steps
    id_batches, scales = [], []
    resident_ids = id_rng.permutation(store.num_graphs)[:per_step]
    for _ in range(n_batches):
        ids = resident_ids if args.resident_batch else id_rng.permutation(store.num_graphs)[:per_step]
        if global_stream and args.captured:
            # a captured step serves batches of ONE graph count: equal contiguous ranges (the mean of the ranks' mean losses is then the
            # global mean loss, weight 1)
            scales.append(1.0)
            ids = ids[rank * args.graphs:(rank + 1) * args.graphs]
        elif global_stream:
            # BASELINE config 4's wording: a fixed global stream, sharded by graph.  Every rank draws the same global batch and keeps
            # its contiguous range (balanced by message count); the loss is weighted so that the averaged gradient equals the
            # gradient of the global mean loss whatever the shard sizes are.
            lo, hi = shard_ranges(msgs_of[ids], world)[rank]
            scales.append(world * (hi - lo) / len(ids))
            ids = ids[lo:hi]
        else:
            scales.append(1.0)
        id_batches.append(ids)
    del recs
    act_dtype = torch.bfloat16 if args.dtype == "bf16" else None
    torch.manual_seed(0)
    model = SpModel(1, args.layers, args.hidden, act_dtype=act_dtype).to(dev)
    # under a launcher: the gradient travels in two ranges issued on a side stream from backward hooks (the later layers' range
    # overlaps the rest of backward); a single process only packs
    sync = FlatGradSync(model.parameters(), overlap=use_dist, buckets=2)
    sync.broadcast_params(0)
    if args.captured and not sync.capturable:
        sys.exit("bench.py: --captured needs the RCCL backend (a host-staged collective cannot be captured into a HIP graph)")
    opt = (torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True, fused=True) if args.captured        # device-side step counters
           else torch.optim.AdamW(model.parameters(), lr=1e-3, fused=args.optimizer == "fused"))

    def step(datadict, loss_scale=1.0):
        sync.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=act_dtype is not None):
            pred = model(datadict)
        loss = torch.nn.functional.l1_loss(datadict["y"].unsqueeze(-1), pred.float())
        (loss if loss_scale == 1.0 else loss * loss_scale).backward()
        sync.mark_backward_end()
        sync.sync()
        opt.step()
        return loss.detach()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def batches():
        if args.resident_batch:                       # A/B and profiling: one batch, resident with its plans, every step
            dd = store.collate(id_batches[0])
            for _ in range(n_batches):
                yield dd
        else:
            pf = BatchPrefetcher(store, id_batches, gated=args.collate_gate == "readout")
            if pf.gated:
                # the next batch's upload + collate kernel run under the readout / loss / first backward launches of the step in flight
                model.lpool.register_forward_pre_hook(lambda _m, _a: pf.gate())
            yield from pf

    timer = _ops.LaunchTimer()
    t0 = None
    slot_step = None
    if args.captured:
        # ONE captured step per rank for every batch: the slot's collate kernel, the step and the collectives its backward hooks
        # issue on the side stream are one HIP graph; per step the host uploads the batch's offsets and replays
        from pygho_amd.graphs import SlotStep
        slot_step = SlotStep(store, args.graphs, step, warmup=3, sync=sync)
        for k, ids in enumerate(id_batches):
            if k == args.warmup:
                barrier()
                t0 = time.perf_counter()
            loss = slot_step.run(ids)
        barrier()
        elapsed = elapsed_own = time.perf_counter() - t0
        with timer:                                   # per-kernel HIP-event times: three EAGER steps after the timed region
            for ids in id_batches[:3]:               # (a replayed graph records no events)
                step(store.collate(ids))
        torch.cuda.synchronize(dev)
    else:
        every_n = max(1, args.event_steps)
        for k, dd in enumerate(batches()):
            if k == args.warmup:
                barrier()
                t0 = time.perf_counter()
            on = k >= args.warmup and (k - args.warmup) % every_n == 0
            if on:
                timer.__enter__()
            loss = step(dd, scales[k])
            if on:
                timer.__exit__(None, None, None)
        barrier()
        elapsed = elapsed_own = time.perf_counter() - t0
    timed = id_batches[args.warmup:]
    fam_total = lambda fam: float(sum(np.asarray(store.h_len[fam])[ids].sum() for ids in timed))
    own_graphs, own_msgs = float(sum(len(ids) for ids in timed)), fam_total(("acd", KEY))
    mean = {"graphs": own_graphs / args.steps, "nodes": fam_total("node") / args.steps, "edges": fam_total("edge") / args.steps,
            "tuples": fam_total("tup") / args.steps, "msg_edges": own_msgs / args.steps}
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        gg = torch.tensor([own_graphs, own_msgs], dtype=torch.float64, device=dev)
        dist.all_reduce(gg, op=dist.ReduceOp.SUM)
        total_graphs, total_msgs = float(gg[0].item()), float(gg[1].item())          # over all ranks and all timed steps
        per_rank = [torch.zeros(2, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(per_rank, torch.tensor([own_graphs / args.steps, elapsed_own], dtype=torch.float64, device=dev))
        per_rank = [[float(t[0].item()), float(t[1].item())] for t in per_rank]
        # after the exchange every rank must hold the SAME averaged gradient, bit for bit: element-wise max and min over the ranks agree
        hi_, lo_ = sync.flat.clone(), sync.flat.clone()
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        grads_equal = bool(torch.equal(hi_, lo_)) and bool(torch.isfinite(hi_).all()) and float(hi_.abs().max()) > 0
        overlap_rep = sync.overlap_report()
    else:
        total_graphs, total_msgs = own_graphs, own_msgs

    if rank == 0:
        assert bool(torch.isfinite(loss)), "loss is not finite"
        summ = timer.summary()
        # the dominant kernel is ONE template instance (seg_gmr_fast_kernel<T, SUM, BOTH>): the forward launches carry
        # the residual row in their epilogue (",res"), the two backward launches per layer do not
        # (the by-edge backward plan runs on its own kernels -- seg_scatter_kernel ",scatter" since round 4, seg_gmr_window_kernel
        # ",window" where the scatter form does not apply; they are reported beside it, in `spspmm_all_launches` and `kernels`)
        # (round 5: with 16-bit activations the FORWARD launch of a layer is the fused block kernel seg_fused_fwd_kernel -- Linear ->
        # BatchNorm -> act formed per chunk on the matrix cores inside the aggregation, csrc/seg_fused.hip; its bytes are what HAS to move:
        # x once, the output once, the stored H once, the indices.  Whichever of the two carries more of the step's time is `roofline`;
        # the other one is reported beside it under `roofline.other`)
        dom = f"seg_gmr[{'bfloat16' if act_dtype is not None else 'float32'},sum,both"
        fused = [v for k, v in summ.items() if k.startswith("seg_fused[")]
        # (round 6: with 16-bit activations BOTH gradients of a layer's aggregation come from one launch of seg_dual_kernel,
        # csrc/seg_dual.hip -- g and H rows fetched once for the by-tuple and the by-edge sum; bytes = what has to move)
        dual = [v for k, v in summ.items() if k.startswith("seg_dual[")]
        every = [v for k, v in summ.items() if k.startswith(dom)] + fused + dual
        parts = [v for k, v in summ.items() if k.startswith(dom) and ",window" not in k and ",scatter" not in k]

        def agg(vs):
            n_ = sum(v[0] for v in vs)
            return n_, sum(v[0] * v[1] for v in vs) / n_, sum(v[0] * v[2] for v in vs) / n_
        tname = "bf16" if act_dtype is not None else "float"
        cands = {f"seg_gmr_fast_kernel<{tname},SUM,BOTH>": parts}
        if fused:
            cands[f"seg_fused_fwd_kernel<{tname},SILU,SUM>"] = fused
        if dual:
            cands[f"seg_dual_kernel<{tname},SUM>"] = dual
        cands = {k: v for k, v in cands.items() if v}
        dom_name = max(cands, key=lambda k: sum(v[0] * v[1] for v in cands[k]))
        launches, ms, nbytes = agg(cands[dom_name])
        achieved = nbytes / (ms * 1e-3) / 1e9
        others = {}
        for k, vs in cands.items():
            if k != dom_name:
                n_, ms_, nb_ = agg(vs)
                t_, src_ = committed_traffic({"graphs_per_gpu": args.graphs, "hidden": args.hidden, "dtype": args.dtype}, k)
                others[k] = {"launches": n_, "avg_ms": ms_, "algorithmic_bytes_per_launch": nb_, "achieved": nb_ / (ms_ * 1e-3) / 1e9,
                             "frac": nb_ / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": t_, "traffic_source": src_}
        op_launches = sum(v[0] for v in every)
        op_ms = sum(v[0] * v[1] for v in every) / op_launches
        op_bytes = sum(v[0] * v[2] for v in every) / op_launches
        es = 2 if act_dtype is not None else 4
        fwd_bytes = es * args.hidden * (2 * mean["tuples"] + mean["edges"]) + 8 * mean["msg_edges"] + 4 * (mean["tuples"] + 1)
        replaced_bytes = fwd_bytes + es * args.hidden * mean["tuples"] + 2 * es * args.hidden * mean["tuples"]
        # HBM traffic of the dominant kernel: collected OUTSIDE this process in separate rocprofv3 --pmc passes of
        # this very command (FETCH_SIZE corrected by the calibrated gfx950 factor, WRITE_SIZE as is) and committed
        # under profiles/ with the hash of the kernel sources it was measured on; null when configuration or sources differ.
        traffic, traffic_src = committed_traffic({"graphs_per_gpu": args.graphs, "hidden": args.hidden, "dtype": args.dtype}, dom_name)
        line = {
            "metric": "graphs/sec, ZINC-shape NGNN train step (+ 2-tuple msg-edges/sec and HBM roofline fraction of the spspmm kernel)",
            "value": total_graphs / elapsed, "unit": "graphs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype if args.dtype == "f32" else "bf16", "data": "synthetic",
            "config": {"workload": "NGNNConv 2-tuple sparse spspmm on ZINC-shape synthetic batches, hidden=128 bf16 "
                                   "(6-layer NGNN of example/minimal.py, full train step)"
                                   + (", ONE resident batch every step (--resident-batch)" if args.resident_batch else
                                      ", a NEW shuffled batch every step, collated on the device from a resident graph store "
                                      "(the reference's loop, example/minimal.py:141-149)"),
                       "batch": "resident" if args.resident_batch else "fresh every step",
                       **({"captured": {"what": "every rank's whole step (collation of the new batch, forward, backward, gradient all-reduce, "
                                                "AdamW) is ONE captured HIP graph over a fixed-capacity batch slot (graphs.SlotStep)",
                                        "replays": slot_step.replays, "eager_fallbacks": slot_step.eager_steps,
                                        "collectives_inside_the_graph_per_step": slot_step._calls_per_step,
                                        "kernel_times": "from three eager steps after the timed region"}} if slot_step is not None else {}),
                       "graphs_per_gpu": mean["graphs"], "store_graphs": store.num_graphs, "store_distinct_graphs": distinct,
                       "nodes": mean["nodes"], "edges": mean["edges"], "tuples": mean["tuples"], "msg_edges": mean["msg_edges"],
                       "sizes": "means over the timed steps of rank 0", "hidden": args.hidden,
                       "layers": args.layers, "parallelism": f"graph-sharded data parallel x{world}"},
            "msg_edges_per_sec_train": total_msgs * args.layers / elapsed,
            "msg_edges_per_sec_kernel": mean["msg_edges"] / (ms * 1e-3),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "kernel": dom_name, "other": others,
                         "launches": launches, "avg_ms": ms, "algorithmic_bytes_per_launch": nbytes,
                         "timed_with": ("HIP events on the launch stream around every launch of "
                                        + ("every" if max(1, args.event_steps) == 1 else f"every {max(1, args.event_steps)}th") + " step of the timed region"
                                        if slot_step is None else "HIP events around every launch of three eager steps after the timed region"),
                         **({"replaced_launches": {
                             "what": "the two launches the fused forward kernel replaces (rounds 1-4): Linear+BatchNorm+act over the tuples "
                                     "(x in, H out) and the spspmm forward with the residual row (H, x in, out; edge rows; indices)",
                             "algorithmic_bytes": replaced_bytes,
                             "frac_at_this_kernels_time": replaced_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}
                            if dom_name.startswith("seg_fused") else {}),
                         "forward_bytes_per_msg_edge": fwd_bytes / mean["msg_edges"],
                         # every spspmm launch of the step (forward + both backward plans, BOTH kernels), same definition
                         "spspmm_all_launches": {"launches": op_launches, "avg_ms": op_ms, "algorithmic_bytes_per_launch": op_bytes,
                                                 "achieved": op_bytes / (op_ms * 1e-3) / 1e9,
                                                 "frac": op_bytes / (op_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}},
            "kernels": {k: {"launches": v[0], "avg_ms": v[1], "GBps": v[2] / (v[1] * 1e-3) / 1e9} for k, v in summ.items()},
        }
        # the segment kernels' XCD-aware sweep assumes workgroup b runs on XCD b % 8: report whether this box dispatches that way
        try:
            from pygho_amd._native import check as _check, ptr as _ptr, stream_ptr as _sp
            ids = torch.empty(2048, dtype=torch.int32, device=dev)
            _check(_native.lib().pygho_xcc_ids(_ptr(ids), 2048, _sp(dev)), "xcc_ids")
            # workgroup b on XCD (b + shift) % 8: the dispatcher continues its round-robin from wherever the previous launch stopped, so
            # the kernels' locality assumption (the 8 residue classes of b are the 8 XCDs) holds for any constant shift
            shift = int(ids[0])
            want = (torch.arange(2048, device=dev, dtype=torch.int32) + shift) % 8
            line["xcd_dispatch"] = {"workgroups": 2048, "xcds_seen": int(ids.unique().numel()), "shift": shift,
                                    "fraction_on_xcd_b_plus_shift_mod_8": float((ids == want).float().mean())}
        except Exception as e:
            line["xcd_dispatch"] = {"error": f"{type(e).__name__}: {e}"}
        if use_dist:
            line["collectives"] = {"backend": dist.get_backend(), "allreduce_calls": sync.allreduce_calls,
                                   "allreduce_bytes": sync.flat.numel() * sync.flat.element_size(),
                                   "allreduce_ms": sync.allreduce_ms(), "allreduce_ranges_per_step": len(getattr(sync, "_ranges", [0])),
                                   "graphs_all_ranks": int(round(total_graphs / args.steps)), "graphs_all_ranks_all_steps": int(total_graphs),
                                   "per_rank_graphs_and_seconds": per_rank,
                                   "flat_grad_equal_across_ranks": grads_equal, "overlap_last_step": overlap_rep,
                                   "ranks_share_gpu": bool(args.ranks_share_gpu),
                                   "overlap": "ranges are all-reduced on a side stream as backward completes them (pygho_amd/parallel.py)",
                                   "batch": "global stream sharded by message count" if global_stream
                                   else "one independent batch stream per rank"}
        if world == 1 and not args.no_regimes:
            del model, opt, sync, store, dd
            torch.cuda.empty_cache()
            try:
                line["regimes"] = side_regimes(args, dev)
            except Exception as e:        # a side measurement must never cost the headline line
                line["regimes"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_configs:
            # BASELINE configs 3 (dense MaskedTensor path) and 5 (3-tuple stress) and one forward + backward of every shipped layer:
            # per-kernel HIP-event times, algorithmic bytes, roofline fraction and committed PMC traffic (tools/bench_configs.py)
            sys.path.insert(0, os.path.join(REPO, "tools"))
            import bench_configs
            torch.cuda.empty_cache()
            try:
                line["configs"] = bench_configs.run(dev)
            except Exception as e:        # a side measurement must never cost the headline line
                line["configs"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, 1000)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if use_dist:
        slot_step = None                              # a captured step holds recorded collectives: gone before their communicator
        torch.cuda.synchronize(dev)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
