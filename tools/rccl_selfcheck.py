"""One-rank RCCL self-check on the GPU box: the same calls bench.py makes at N > 1 (init with device_id,
barrier, MAX all-reduce through sharding.max_over_ranks, the mfcc key all-reduce), world_size 1 over RCCL.
usage: python3 tools/rccl_selfcheck.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist
from mlx_audio_primitives_amd import sharding
import mlx_audio_primitives_amd as ap

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
dist.barrier()
torch.cuda.synchronize()
print("max_over_ranks", sharding.max_over_ranks(1.25, device=dev))
y = torch.randn(4, 22050, device=dev)
a = ap.mfcc(y, n_mfcc=13)
b = sharding.mfcc_sharded(y, n_mfcc=13)
print("mfcc group == local:", bool(torch.equal(a, b)))
dist.barrier()
dist.destroy_process_group()
print("ok")
