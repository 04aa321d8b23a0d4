# what would perfect lhs locality buy?  same plan, lhs index replaced by the output slot (streaming) or by a constant row
import sys, os, json, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_ops as B
from pygho_amd import _ops, synth
dev = torch.device("cuda:0")
for kind, graphs, d in (("zinc", 8192, 128), ("i2", 2048, 256)):
    base = 1024 if kind == "zinc" else 128
    key = "X___X___1___A___0" if kind == "zinc" else "X___X___2___A___0"
    hb = synth.replicate(synth.make_batch(min(graphs, base), kind, seed=1), max(1, graphs // base))
    acd = torch.from_numpy(hb.acd[key]).to(dev)
    nt, ne, m = hb.num_tuples, hb.num_edges, acd.shape[1]
    X = torch.randn(nt, d, device=dev).to(torch.bfloat16)
    A = torch.randn(ne, d, device=dev).to(torch.bfloat16)
    plan = _ops.message_plan(acd, nt, nt, ne)
    a32 = _ops.narrow_i32(acd[0].contiguous())
    zero = torch.zeros_like(a32)
    nbytes = 2 * d * (2 * nt + ne) + 8 * m + 4 * (nt + 1)
    for name, c, dd in (("real", plan.c_fwd, plan.d_fwd), ("lhs=slot", a32, plan.d_fwd), ("lhs=slot,rhs=0", a32, zero), ("lhs=0,rhs=0", zero, zero)):
        ms = B.timed(lambda: _ops.seg_gmr(nt, X, A, plan.fwd.seg_ptr, c, dd, "sum"))
        print(kind, name, round(ms * 1e3, 1), "us", round(nbytes / ms / 1e6 / 8000, 3))
