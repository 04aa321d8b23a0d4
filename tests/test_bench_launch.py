"""
bench.py's launch contract.  CPU part: a --gpus N that does not match the launcher's WORLD_SIZE is a hard error, and
`python bench.py --gpus N` without a launcher refuses when fewer than N GPUs are visible (it never silently runs one rank).
GPU part: bench.py under `torch.distributed.run --nproc-per-node 1` so that RCCL initialisation, the parameter broadcast and
the flat gradient all-reduce execute on the hardware; `python bench.py --gpus 2` (self-launched ranks) where two GPUs exist.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _gpus() -> int:
    """visible GPUs by bench.py's own runtime-free count (importing bench.py runs nothing: its work is under __main__)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_bench_for_tests", BENCH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n = mod.visible_gpu_count()
    return 0 if n is None else n


SMALL = ["--steps", "3", "--warmup", "2", "--graphs", "256", "--distinct", "128", "--no-cpu-baseline", "--no-regimes", "--no-configs"]


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    return env


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_world_size_mismatch_is_a_hard_error():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()


@pytest.mark.skipif(_gpus() >= 2, reason="only meaningful where fewer than 2 GPUs are visible")
def test_self_launch_refuses_without_enough_gpus():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) are visible" in r.stderr and not r.stdout.strip()


def _line(stdout):
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), stdout      # stdout carries the JSON line and nothing else (no RCCL banner)
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_under_torchrun_one_rank_uses_rccl():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), BENCH, "--gpus", "1"] + SMALL
    r = subprocess.run(cmd, env=_env(PYGHO_BENCH_TRACE_COLLECTIVES="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    line = _line(r.stdout)
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["roofline"]["frac"] > 0
    assert line["collectives"]["backend"] == "nccl" and line["collectives"]["allreduce_calls"] >= 5


@pytest.mark.gpu
def test_bench_captured_under_torchrun_one_rank_has_the_collective_inside_the_graph():
    """BASELINE config 4's operating point, on what one GPU allows: `bench.py --gpus 1 --captured` under torch.distributed.run -- the
    rank's whole step (collation, forward, backward, the two RCCL all-reduce ranges issued from backward hooks on a side stream,
    AdamW) is ONE captured HIP graph replayed per batch.  Checked: every timed and warm-up step was a replay, two collectives sit inside
    the graph, the counted collectives are the replays' (+ warm-up, capture and the three eager steps that time the kernels)."""
    steps, warmup = 4, 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), BENCH, "--gpus", "1", "--captured", "--steps", str(steps), "--warmup", str(warmup),
           "--graphs", "256", "--distinct", "256", "--no-cpu-baseline", "--no-regimes", "--no-configs"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    line = _line(r.stdout)
    cap = line["config"]["captured"]
    assert cap["replays"] + cap["eager_fallbacks"] == steps + warmup and cap["replays"] >= steps
    assert cap["collectives_inside_the_graph_per_step"] == 2
    c = line["collectives"]
    # 3 warm-up runs + 1 capture inside SlotStep, then one pair per replay / fallback, then the 3 eager steps that time the kernels
    assert c["backend"] == "nccl" and c["allreduce_calls"] == 2 * (4 + steps + warmup + 3)
    assert line["value"] > 0 and line["roofline"]["frac"] > 0


@pytest.mark.gpu
def test_captured_refuses_a_host_staged_backend():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dist-backend", "gloo", "--ranks-share-gpu", "--captured"] + SMALL,
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0


@pytest.mark.gpu
@pytest.mark.skipif(_gpus() < 2, reason="needs two GPUs")
def test_bench_self_launches_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + SMALL, env=_env(PYGHO_BENCH_TRACE_COLLECTIVES="1"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    line = _line(r.stdout)
    assert line["n_gpus"] == 2 and line["config"]["graphs_per_gpu"] == 256


@pytest.mark.gpu
@pytest.mark.parametrize("global_stream", [False, True])
def test_bench_two_ranks_share_one_gpu(global_stream):
    """bench.py's world > 1 branch on a one-GPU box: `python bench.py --gpus 2 --dist-backend gloo --ranks-share-gpu` goes through
    launch_ranks (child torch.distributed.run, two ranks, both on cuda:0, the HIP path in each), once with one independent batch per
    rank and once with --global-stream (one global batch sharded by message count).  Checked: one JSON line; n_gpus; every step
    exchanged its two ranges; a collective time was measured; value = graphs of BOTH ranks / max-over-ranks time; both ranks end with
    the same flat gradient; the first range's all-reduce was issued -- host clock AND device timeline -- before backward ended."""
    steps, warmup = 3, 2
    extra = ["--global-stream", "--distinct", "256"] if global_stream else []      # 512 distinct graphs: the message-balanced cut is not the middle
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dist-backend", "gloo", "--ranks-share-gpu"] + SMALL + extra,
                       env=_env(), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-4000:]
    line = _line(r.stdout)
    c = line["collectives"]
    assert line["n_gpus"] == 2 and line["steps"] == steps and c["backend"] == "gloo" and c["ranks_share_gpu"] is True
    assert c["allreduce_ranges_per_step"] == 2 and c["allreduce_calls"] == 2 * (steps + warmup)
    assert c["allreduce_ms"] is not None and c["allreduce_ms"] > 0
    (g0, t0), (g1, t1) = c["per_rank_graphs_and_seconds"]
    assert g0 + g1 == c["graphs_all_ranks"] == 512
    if global_stream:
        assert c["batch"] == "global stream sharded by message count" and g0 != 256 and 200 < g0 < 312   # balanced by messages, not graphs
    else:
        assert c["batch"] == "one independent batch stream per rank" and g0 == g1 == 256
    elapsed = line["ms_per_step"] * steps / 1e3
    assert abs(elapsed - max(t0, t1)) < 1e-6 * elapsed                   # the slowest rank's clock
    assert abs(line["value"] - (g0 + g1) * steps / elapsed) < 1e-6 * line["value"]
    assert c["flat_grad_equal_across_ranks"] is True
    ov = c["overlap_last_step"]
    assert ov["ranges"] == 2 and ov["ranges_issued_inside_backward"] >= 1 and ov["first_range"] == 1   # backward completes the LAST range first
    assert ov["host_ms_first_issue_before_backward_end"] > 0
    assert ov["device_ms_first_collective_start_before_backward_end"] > 0
    assert line["roofline"]["frac"] > 0 and "regimes" not in line
    # the wording the driver's scaling table is built from, and the regime: a new batch every step from a resident store
    assert line["scaling"] == "weak" and line["config"]["parallelism"] == "graph-sharded data parallel x2"
    assert line["config"]["batch"] == "fresh every step" and line["config"]["store_graphs"] == (1024 if global_stream else 512)
    assert len([l for l in r.stdout.splitlines() if l.strip()]) == 1            # rank 1 (and every library banner) stays off stdout


def test_ranks_share_gpu_refuses_rccl():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--ranks-share-gpu"], env=_env(HIP_VISIBLE_DEVICES="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "needs --dist-backend gloo" in r.stderr and not r.stdout.strip()


def test_visible_gpu_count_reads_the_environment_not_the_runtime(monkeypatch):
    """bench.py counts GPUs from the visibility variables (or the KFD topology) so that the parent of a self-launched run never
    initialises HIP: the variables win, an empty list means no GPU"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_bench_for_count", BENCH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0,1,2")
    assert mod.visible_gpu_count() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "3")                   # the first variable in HIP's own precedence order
    assert mod.visible_gpu_count() == 1
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert mod.visible_gpu_count() == 0
