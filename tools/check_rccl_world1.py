"""GPU-box check of the N>1 plumbing that a 1-GPU box can exercise: the RCCL process group of bench.py / the data-parallel
learner with WORLD_SIZE=1 (init with device_id, barrier, MAX / SUM reductions on the device, teardown)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RANK", "0"); os.environ.setdefault("LOCAL_RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch  # noqa: E402

from synthesis_amd import dist_util  # noqa: E402

rank, local_rank, world = dist_util.rank_info()
dist = dist_util.init_process_group("nccl", local_rank)
torch.cuda.synchronize(local_rank)
dist.barrier()
elapsed, (plies,) = dist_util.reduce_scalars(dist, f"cuda:{local_rank}", 1.25, [12345])
assert elapsed == 1.25 and plies == 12345
g = torch.arange(30492, dtype=torch.float32, device=f"cuda:{local_rank}")
dist.all_reduce(g)
assert float(g[-1].item()) == 30491.0
dist.barrier()
dist.destroy_process_group()
print("rccl world-1 ok", torch.cuda.get_device_name(local_rank))
