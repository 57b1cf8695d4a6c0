"""Worker of tests/test_gpu_training.py::test_data_parallel_learner_two_ranks — launched by torch.distributed.run with two
ranks that share GPU 0 and reduce over gloo (DP_BACKEND=nccl: one rank per GPU over RCCL, world size 1 on a one-GPU box). Each rank trains on its half of every batch; rank r writes its final state."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir = sys.argv[1]
    import torch
    import torch.distributed as dist

    import synthesis_amd as sa
    from synthesis_amd.learner import DataParallelLearner

    backend = os.environ.get("DP_BACKEND", "gloo")
    if backend == "nccl":   # RCCL: one GPU per rank (a one-GPU box runs it with one rank)
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    gold = np.load(os.path.join(ROOT, "tests", "golden", "train_torch_goldens.npz"))
    blob = np.load(os.path.join(ROOT, "tests", "golden", "c4net_blob_f32.npy"))
    eng = sa.Engine(concurrent_games=64, max_explores=64, device=0)
    eng.load_weights(blob)
    learner = DataParallelLearner(eng, blob, dist=dist, device=0, collective_at_world_1=True)
    assert learner.dist is not None and learner._staged == (backend != "nccl")
    steps, B = 4, gold["my_bb"].shape[1]
    half = B // world
    losses = []
    for s in range(steps):
        sl = slice(rank * half, (rank + 1) * half)
        losses.append(learner.step(gold["my_bb"][s, sl], gold["op_bb"][s, sl], gold["target_pi"][s, sl],
                                   gold["target_v"][s, sl], float(gold["lrs"][s])))
    st = learner.state()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), weights=st["weights"], m=st["m"], v=st["v"], step=st["step"],
             losses=np.stack(losses))
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
