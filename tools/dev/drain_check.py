import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/speech-tranformer-pytorch_amd")
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("TORCH_NCCL_TRACE_BUFFER_SIZE", "512")      # (the flight recorder: trainer.drain_collective_watchdog)
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
from st_amd.trainer import drain_collective_watchdog
x = torch.ones(1 << 20, device="cuda")
for _ in range(5):
    dist.all_reduce(x)
t0 = time.time(); how = drain_collective_watchdog(); print("drain:", how, "%.1f ms" % ((time.time() - t0) * 1e3))
for _ in range(3):
    dist.all_reduce(x)
t0 = time.time(); how = drain_collective_watchdog(); print("drain:", how, "%.1f ms" % ((time.time() - t0) * 1e3))
dist.destroy_process_group()
