"""Probe: can RCCL collectives (torch.distributed, backend nccl) be captured into a hipGraph on this stack?
Single rank (a one-GPU box cannot host two RCCL ranks); the capture path in ProcessGroupNCCL is the same."""
import os, sys
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
x = torch.ones(16, 4096, device="cuda", dtype=torch.float16)
g_out = torch.empty(16, 4096, device="cuda", dtype=torch.float16)
dist.all_reduce(x); dist.all_gather_into_tensor(g_out, x); torch.cuda.synchronize()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    dist.all_reduce(x); dist.all_gather_into_tensor(g_out, x)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        y = x * 2
        dist.all_reduce(y)
        dist.all_gather_into_tensor(g_out, y)
        z = g_out + 1
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("RCCL_GRAPH_CAPTURE_OK", float(z[0, 0]), flush=True)
except Exception as e:
    print("RCCL_GRAPH_CAPTURE_FAILED", repr(e)[:300], flush=True)
dist.destroy_process_group()
