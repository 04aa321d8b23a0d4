import faulthandler, os, sys, socket
faulthandler.enable()
import torch, torch.distributed as dist
mode = sys.argv[1]
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.ones(1000, device=dev)
side = torch.cuda.Stream()
n_coll = 2 if "two" in mode else 1

def body():
    y = x * 2
    for k in range(n_coll):
        piece = y[k * 500:(k + 1) * 500]
        if "cur" in mode:
            if "sync" in mode:
                dist.all_reduce(piece)
            else:
                w = dist.all_reduce(piece, async_op=True); w.wait()
        else:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                w = dist.all_reduce(piece, async_op=True); w.wait()
    if "cur" not in mode:
        torch.cuda.current_stream().wait_stream(side)
    return y + 1

if "s1warm" in mode:
    s1 = torch.cuda.Stream(); s1.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1):
        for _ in range(3): body()
    torch.cuda.current_stream().wait_stream(s1)
else:
    for _ in range(3): body()
    torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    z = body()
g.replay(); torch.cuda.synchronize(); print(mode, "ok", float(z[0]), flush=True)
dist.destroy_process_group()
