import os, sys, socket
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
import bnv_fusion_amd
bnv_fusion_amd.configure_runtime()
from bnv_fusion_amd import _lib, streams
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
x = torch.zeros(4, device="cuda:0"); dist.all_reduce(x)
lib = _lib.require_device(0)
main = torch.cuda.current_stream()
print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"))
cands = [torch.cuda.Stream() for _ in range(8)] + [torch.cuda.Stream(priority=-1) for _ in range(3)]
for k, c in enumerate(cands):
    ok, one, two = streams._overlaps(lib, main, c)
    print(k, "prio", c.priority, hex(c.cuda_stream), "overlaps main:", ok, f"{one*1e3:.0f} {two*1e3:.0f} us")
