# FETCH_SIZE calibration for 512-B row gathers: the I2 plan with lhs index = output slot, rhs index = 0: every lhs row is read
# exactly once (plus L1 hits), known read volume = nt * 512 B + indices
import sys, os, json, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_ops as B
from pygho_amd import _ops, synth
dev = torch.device("cuda:0")
hb = synth.replicate(synth.make_batch(128, "i2", seed=1), 16)
acd = torch.from_numpy(hb.acd["X___X___2___A___0"]).to(dev)
nt, ne, m = hb.num_tuples, hb.num_edges, acd.shape[1]
X = torch.randn(nt, 256, device=dev).to(torch.bfloat16)
A = torch.randn(ne, 256, device=dev).to(torch.bfloat16)
plan = _ops.message_plan(acd, nt, nt, ne)
a32 = _ops.narrow_i32(acd[0].contiguous())
zero = torch.zeros_like(a32)
for _ in range(6):
    _ops.seg_gmr(nt, X, A, plan.fwd.seg_ptr, a32, zero, "sum")
torch.cuda.synchronize()
print(json.dumps({"known_read_bytes": nt * 512 + 12 * m + 4 * nt, "write_bytes": nt * 512, "nt": nt, "m": m}))
