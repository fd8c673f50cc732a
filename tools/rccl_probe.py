"""dev helper: single-rank run of the RCCL code paths the multi-GPU bench relies on (process group on the nccl backend,
barrier, 8-byte SUM all-reduce of the captured-metric partial, MAX all-reduce of the step time)"""
import os, sys
sys.path.insert(0, ".")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
import torch as pt, torch.distributed as dist
pt.cuda.set_device(0)
dist.init_process_group("nccl", device_id=pt.device("cuda", 0))
print("backend", dist.get_backend(), "world", dist.get_world_size(), flush=True)
dist.barrier()
t = pt.tensor([1.25], dtype=pt.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.SUM); print("sum", t.item(), flush=True)
dist.all_reduce(t, op=dist.ReduceOp.MAX); print("max", t.item(), flush=True)
from sparsespatialsampling_amd import parallel
print("world()", parallel.world(), parallel.shard_range(10))
dist.barrier(); dist.destroy_process_group(); print("ok")
