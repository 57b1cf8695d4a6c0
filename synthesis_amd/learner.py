"""Data-parallel learner step (BASELINE configs[4]): the optimiser half of alpha_zero.rs:72-94 with the
gradient all-reduce over RCCL/xGMI that the single-process reference does not need.

One process per GPU. Every rank holds identical weights and Adam moments on its own engine; per step each rank computes
the gradients of ITS shard of the batch with the HIP training kernels (`syn_train_gradients_device`, loss scaled by
1 / local batch like alpha_zero.rs:44), the 30,492-float gradient buffer (122 KB — latency-bound, one message) is
all-reduced (sum) and every rank applies the same Adam update with `grad_scale = 1 / world`
(`syn_train_apply_device`), i.e. the step on the mean loss of the global batch. No parameter broadcast is ever needed.

torch is plumbing here: it owns the gradient buffer (so `torch.distributed` can reduce it in place — backend "nccl" is RCCL
on ROCm) and the device copies of the batch. With the "gloo" backend (CPU tests, or several ranks sharing one GPU) the
buffer is staged through host memory.
"""
import numpy as np

from .engine import NUM_PARAMS


class DataParallelLearner:
    def __init__(self, engine, blob, dist=None, device=0, **hyper):
        import torch

        self._torch = torch
        self.engine = engine
        self.dist = dist if (dist is not None and dist.is_initialized() and dist.get_world_size() > 1) else None
        self.world = self.dist.get_world_size() if self.dist else 1
        self.device = torch.device(f"cuda:{device}")
        engine.trainer_init(blob, **hyper)
        self.grads = torch.zeros(NUM_PARAMS, dtype=torch.float32, device=self.device)
        self._staged = self.dist is not None and self.dist.get_backend() != "nccl"

    def step(self, my_bb, op_bb, target_pi, target_v, lr):
        """One optimiser step; the arguments are THIS rank's shard of the batch. Returns the (pi_loss, v_loss) of the
        global batch (mean over ranks of the per-shard means)."""
        torch = self._torch
        my = torch.from_numpy(np.ascontiguousarray(my_bb, dtype=np.uint64).view(np.int64)).to(self.device)
        op = torch.from_numpy(np.ascontiguousarray(op_bb, dtype=np.uint64).view(np.int64)).to(self.device)
        tpi = torch.from_numpy(np.ascontiguousarray(target_pi, dtype=np.float32)).to(self.device)
        tv = torch.from_numpy(np.ascontiguousarray(target_v, dtype=np.float32)).to(self.device)
        torch.cuda.synchronize(self.device)  # the engine runs on its own stream
        losses = self.engine.train_gradients_device(my.data_ptr(), op.data_ptr(), tpi.data_ptr(), tv.data_ptr(),
                                                    int(my.numel()), self.grads.data_ptr())  # returns after the kernel
        if self.dist is not None:
            if self._staged:
                g = self.grads.cpu()
                self.dist.all_reduce(g)
                self.grads.copy_(g)
                l = torch.from_numpy(losses.copy())
                self.dist.all_reduce(l)
            else:
                self.dist.all_reduce(self.grads)  # RCCL over xGMI: one 122 KB message
                l = torch.from_numpy(losses.copy()).to(self.device)
                self.dist.all_reduce(l)
                l = l.cpu()
            losses = (l / self.world).numpy()
            torch.cuda.synchronize(self.device)
        self.engine.train_apply_device(self.grads.data_ptr(), lr, grad_scale=1.0 / self.world)
        return losses

    def publish(self):
        """The trained network becomes the engine's self-play network (model_{i+1}.ot of alpha_zero.rs:97)."""
        self.engine.trainer_publish_weights()

    def state(self):
        return self.engine.trainer_state()
