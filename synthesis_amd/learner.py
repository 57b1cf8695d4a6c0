"""The learner side of the reference's loop on N GPUs (BASELINE configs[4]):

    alpha_zero()            synthesis/src/alpha_zero.rs:16-118   iterate { gather_experience -> dedup -> Adam epochs -> save model }
    gather_experience()     alpha_zero.rs:120-179                worker threads play games_per_train games, buffers concatenated
    train step              alpha_zero.rs:72-94                  forward, kl_div losses, Adam

`LearningLoop` is that loop in the shape that scales on one node: EVERY rank plays its share of an iteration's games (sharded by
game index, no collective on the data path — games share nothing), rank 0 gathers the new positions, owns the replay buffer and
trains with the persistent epoch kernel (`syn_train_epoch`: 13 us per optimiser step at the reference's batch of 32 — a
data-parallel step at that batch size moves 122 KB over xGMI for 13 us of compute and can only be slower than one GPU), and the
new parameters (122 KB for Connect4Net, 50 KB for Connect4ConvNet) are broadcast once per iteration. One process per GPU;
torch.distributed is the transport (backend "nccl" is RCCL over xGMI; "gloo" in the CPU tests and when ranks share a GPU).
With one rank it is exactly the single-GPU loop. The games an iteration plays, the replay buffer and therefore the trained
weights do not depend on the number of ranks.

`DataParallelLearner` is the other reading of configs[4] — a gradient all-reduce per optimiser step — kept for callers who want
it (large batches): every rank computes the gradients of ITS shard of the batch (`syn_train_gradients_device`), the gradient
buffer and the two loss sums travel in ONE all-reduce, every rank applies the same Adam update with grad_scale = 1 / world
(`syn_train_apply_device`), so weights stay identical without a broadcast. Both networks. The data set and the epoch's
permutation live on the device; a step gathers its shard there and the losses stay on the device until asked for.

torch is plumbing in both: it owns the buffers torch.distributed moves. Nothing here touches oracle/.
"""
import time

import numpy as np

from .engine import CONV_NUM_PARAMS, NUM_PARAMS, shard_games


def _lr_at(schedule, i_iter):
    """alpha_zero.rs:62-69: the last (iteration, lr) entry whose iteration is <= i_iter + 1"""
    lr = schedule[0][1]
    for it, v in schedule:
        if it <= i_iter + 1:
            lr = v
    return lr


class LearningLoop:
    """alpha_zero.rs:16-118 with self-play on every rank and the learner on rank 0 (see the module docstring).

    engine       this rank's sa.Engine (self-play; on rank 0 also the learner and the de-duplication)
    net          "mlp" (Connect4Net, study-connect4/src/policies.rs:14-59) | "conv" (Connect4ConvNet)
    blob         initial parameters — identical on every rank (P::new(&vs), alpha_zero.rs:31)
    precision    "f32" | "bf16" (Connect4ConvNet's learner only: syn_trainer_set_precision)
    dist         torch.distributed (initialised) or None for one rank
    sampler      "numpy" (default: numpy's PCG64 permutation seeded by (seed, iteration, epoch)) | "torch": BatchRandSampler's own
                 `Tensor::randperm(n, INT64_CPU)` (data.rs:29) — libtorch's CPU randperm from ONE generator seeded with `seed` and
                 advanced epoch after epoch: the reference's algorithm and generator TYPE, not its stream — in a reference run
                 `tch::manual_seed(seed)` (alpha_zero.rs:28) is followed by `P::new(&vs)` (:31), whose parameter initialisation draws
                 from that same global generator before the first randperm, so the permutations of a Rust run differ
    logs_dir     None, or where the learner's rank writes what the reference writes per iteration (alpha_zero.rs:37,97-100):
                 models/model_{i}.ot (Connect4Net: a VarStore archive `vs.load` reads; Connect4ConvNet: the flat blob as .npy) and
                 latest_states.npy [n, 1, 7, 9] / latest_pis.npy [n, 9] / latest_vs.npy [n, 3] of the de-duplicated buffer
    """

    def __init__(self, engine, net, blob, dist=None, device=0, lr_schedule=((1, 1e-3),), seed=0, precision="f32", logs_dir=None, sampler="numpy", **hyper):
        import torch

        self._torch = torch
        self.engine = engine
        self.net = net
        self.dist = dist if (dist is not None and dist.is_initialized() and dist.get_world_size() > 1) else None
        self.world = self.dist.get_world_size() if self.dist else 1
        self.rank = self.dist.get_rank() if self.dist else 0
        self.lr_schedule = list(lr_schedule)
        self.seed = int(seed)
        self.n_params = CONV_NUM_PARAMS if net == "conv" else NUM_PARAMS
        blob = np.ascontiguousarray(blob, dtype=np.float32).ravel()
        assert blob.size == self.n_params
        self._load = engine.load_weights_conv if net == "conv" else engine.load_weights
        self._load(blob)
        self.weights = blob.copy()
        # the buffer the broadcast moves: on the GPU for RCCL, on the host for gloo
        on_gpu = self.dist is not None and self.dist.get_backend() == "nccl"
        self._wbuf = torch.zeros(self.n_params, dtype=torch.float32, device=torch.device(f"cuda:{device}") if on_gpu else "cpu")
        if self.rank == 0:
            (engine.trainer_init_conv if net == "conv" else engine.trainer_init)(blob, **hyper)
            if precision != "f32":
                engine.trainer_set_precision(precision)   # "bf16": the conv learner's bf16 matrix-core variant (BASELINE configs[4])
        # replay buffer (rank 0): positions as bitboards + targets + the game each step came from (data.rs:107-158)
        self.R = dict(my=np.zeros(0, np.uint64), op=np.zeros(0, np.uint64), pi=np.zeros((0, 9), np.float32),
                      v=np.zeros((0, 3), np.float32), gid=np.zeros(0, np.int64))
        self.games_played = 0
        self.iterations_done = 0
        if sampler not in ("numpy", "torch"):
            raise ValueError(f"sampler must be 'numpy' or 'torch', got {sampler!r}")
        self.sampler = sampler
        self._torch_gen = torch.Generator().manual_seed(self.seed) if sampler == "torch" else None
        self.logs_dir = logs_dir if self.rank == 0 else None
        if self.logs_dir:
            self._save_model(0)

    def _save_model(self, i):
        import os

        from .weights import save_ot

        d = os.path.join(self.logs_dir, "models")
        os.makedirs(d, exist_ok=True)
        if self.net == "mlp":
            save_ot(self.weights, os.path.join(d, f"model_{i}.ot"))
        else:
            np.save(os.path.join(d, f"model_{i}.npy"), self.weights)

    # one replay position on the wire: my_bb, op_bb (u64), gid (i64), pi[9], v[3] (f32) = 72 bytes, no pickling. A rank's buffer is
    # five contiguous sections [my | op | gid | pi | v] of `cap` positions each (five block copies to pack, views to unpack).
    _FIELDS = (("my", np.uint64, 1), ("op", np.uint64, 1), ("gid", np.int64, 1), ("pi", np.float32, 9), ("v", np.float32, 3))
    _POS_BYTES = 72

    def _gather_positions(self, new):
        """The ranks' new positions to the learner's rank (in rank order = game order) as ONE fixed-layout tensor gather: the
        counts travel first (one all_gather of an int64 per rank), then every rank contributes a buffer sized for the largest count —
        device buffers under RCCL, host buffers under gloo. (The round-3 form pickled a dict of arrays per rank: rank 0 unpickled
        8 x 18 MB per iteration.)"""
        t = self._torch
        n = int(new["my"].size)
        on_gpu = self._wbuf.is_cuda
        dev = self._wbuf.device
        cnt = t.tensor([n], dtype=t.int64, device=dev)
        counts = [t.zeros(1, dtype=t.int64, device=dev) for _ in range(self.world)]
        t_w = time.perf_counter()
        self.dist.all_gather(counts, cnt)     # (also where a rank waits for the slowest rank's self-play to end)
        counts = [int(c.item()) for c in counts]
        self._last_gather_wait = time.perf_counter() - t_w
        # every rank sends the same, slowly changing size: the largest count rounded up to 32,768 positions, so that the pack buffer and
        # rank 0's receive buffers live across iterations (first-touch page faults of fresh 10 MB buffers were most of this phase)
        cap = (max(max(counts), 1) + 32767) // 32768 * 32768
        if getattr(self, "_gcap", 0) != cap:
            self._gcap = cap
            self._gbuf = np.zeros(cap * self._POS_BYTES, np.uint8)
            self._gdev = t.empty(cap * self._POS_BYTES, dtype=t.uint8, device=dev) if on_gpu else None
            like = self._gdev if on_gpu else t.from_numpy(self._gbuf)
            self._gparts = [t.zeros_like(like) for _ in range(self.world)] if self.rank == 0 else None
        buf = self._gbuf
        off = 0
        for name, dt, width in self._FIELDS:
            sec = buf[off: off + cap * np.dtype(dt).itemsize * width].view(dt)
            sec[: n * width] = np.ascontiguousarray(new[name], dtype=dt).reshape(-1)
            off += cap * np.dtype(dt).itemsize * width
        mine = t.from_numpy(buf)
        if on_gpu:
            self._gdev.copy_(mine)
            mine = self._gdev
        parts = self._gparts
        self.dist.gather(mine, parts, dst=0)
        if self.rank != 0:
            return new
        out = {name: [] for name, _, _ in self._FIELDS}
        for p_, c in zip(parts, counts):
            raw = p_.cpu().numpy()
            off = 0
            for name, dt, width in self._FIELDS:
                sec = raw[off: off + cap * np.dtype(dt).itemsize * width].view(dt)[: c * width]
                out[name].append(sec.reshape(c, width) if width > 1 else sec)
                off += cap * np.dtype(dt).itemsize * width
        return {k: np.concatenate(v) for k, v in out.items()}

    def iteration(self, cfg, games_per_train, games_to_keep, epochs, batch_size):
        """One pass of the loop body (alpha_zero.rs:42-100). Returns this rank's record of it (rank 0's has the learner's numbers)."""
        it = self.iterations_done
        t0 = time.perf_counter()
        # ---- gather_experience: this rank's share of the new games; global game index = seed offset, never reused
        off, count = shard_games(games_per_train, self.rank, self.world)
        first = it * games_per_train + off
        sp = self.engine.selfplay(cfg, base_seed=self.seed, n_games=count, first_game=first)
        t_play = time.perf_counter() - t0
        n = sp["plies"]
        mask = np.arange(63)[None, :] < n[:, None]
        new = dict(my=sp["states_bb"][..., 0][mask], op=sp["states_bb"][..., 1][mask], pi=sp["pis"][mask], v=sp["vs"][mask],
                   gid=(first + np.arange(count))[:, None].repeat(63, 1)[mask])
        t1 = time.perf_counter()
        self._last_gather_wait = 0.0
        if self.dist is not None:
            new = self._gather_positions(new)
        t_gather = time.perf_counter() - t1 - self._last_gather_wait   # pack + gather + unpack, without the wait for the slowest rank
        self.games_played += games_per_train
        lr = _lr_at(self.lr_schedule, it)
        rec = dict(iteration=it + 1, lr=lr, games=int(games_per_train), games_this_rank=int(count),
                   plies_per_game=float(n.mean()) if count else 0.0)
        t_dedup = t_train = 0.0
        if self.rank == 0:
            R = {k: np.concatenate([self.R[k], new[k]]) for k in self.R}
            keep = R["gid"] >= self.games_played - games_to_keep   # keep_last_n_games (data.rs:160-194)
            self.R = {k: a[keep] for k, a in R.items()}
            # ---- deduplicate on the GPU (data.rs:196-235)
            t2 = time.perf_counter()
            D = self.engine.replay_deduplicate(self.R["my"], self.R["op"], self.R["pi"], self.R["v"])
            t_dedup = time.perf_counter() - t2
            n_unique = int(D["num"].size)
            # ---- epochs of optimiser steps (alpha_zero.rs:72-94): one upload, one persistent kernel launch per epoch
            t3 = time.perf_counter()
            self.engine.train_set_data(D["my_bb"], D["op_bb"], D["pis"], D["vs"])
            steps, epoch_losses = 0, []
            n_steps = n_unique // batch_size   # drop_last = true (data.rs:41-62)
            for ep in range(epochs):
                if self.sampler == "torch":   # BatchRandSampler::new (data.rs:29)
                    perm = self._torch.randperm(n_unique, generator=self._torch_gen, dtype=self._torch.int64).numpy()
                else:
                    perm = np.random.default_rng([self.seed, it, ep]).permutation(n_unique)
                if n_steps:
                    sl = self.engine.train_epoch(perm[: n_steps * batch_size], batch_size, lr)
                    epoch_losses.append((sl.astype(np.float64).sum(axis=0) * batch_size / n_unique).tolist())
                steps += n_steps
            t_train = time.perf_counter() - t3
            self.weights = self.engine.trainer_state()["weights"]
            rec.update(steps_in_buffer=int(self.R["my"].size), unique=n_unique, optimiser_steps=steps, epoch_losses=epoch_losses)
            if self.logs_dir:   # alpha_zero.rs:97-100
                import os

                self._save_model(it + 1)
                np.save(os.path.join(self.logs_dir, "latest_states.npy"),
                        np.asarray(self.engine.features(D["my_bb"], D["op_bb"]), np.float32).reshape(-1, 1, 7, 9))
                np.save(os.path.join(self.logs_dir, "latest_pis.npy"), np.asarray(D["pis"], np.float32).reshape(-1, 9))
                np.save(os.path.join(self.logs_dir, "latest_vs.npy"), np.asarray(D["vs"], np.float32).reshape(-1, 3))
        # ---- model_{i+1} (alpha_zero.rs:97,194): the trained parameters become every rank's self-play network
        t4 = time.perf_counter()
        if self.dist is not None:
            if self.rank == 0:
                self._wbuf.copy_(self._torch.from_numpy(self.weights))
            self.dist.broadcast(self._wbuf, src=0)
            self.weights = self._wbuf.cpu().numpy().copy()
            self._load(self.weights)
        elif self.rank == 0:
            self.engine.trainer_publish_weights()   # one rank: the learner's image is copied on the device
        t_bcast = time.perf_counter() - t4
        self.iterations_done += 1
        rec["seconds"] = dict(selfplay=round(t_play, 4), wait_for_ranks=round(self._last_gather_wait, 4), gather=round(t_gather, 4),
                              dedup=round(t_dedup, 4), train=round(t_train, 4),
                              broadcast=round(t_bcast, 4), total=round(time.perf_counter() - t0, 4))
        return rec


class DataParallelLearner:
    """Gradient all-reduce per optimiser step (see the module docstring). Every rank holds identical weights and Adam moments."""

    def __init__(self, engine, blob, dist=None, device=0, net="mlp", collective_at_world_1=False, **hyper):
        """collective_at_world_1: keep the all-reduce in the step when the group has one rank (bench.py's `data_parallel_world1`:
        the literal gradients -> RCCL all-reduce -> Adam path of BASELINE configs[4] on a one-GPU box)."""
        import torch

        self._torch = torch
        self.engine = engine
        self.net = net
        self.dist = dist if (dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or collective_at_world_1)) else None
        self.world = self.dist.get_world_size() if self.dist else 1
        self.device = torch.device(f"cuda:{device}")
        self.n_params = CONV_NUM_PARAMS if net == "conv" else NUM_PARAMS
        (engine.trainer_init_conv if net == "conv" else engine.trainer_init)(blob, **hyper)
        # [gradients | pi-loss sum | v-loss sum]: one buffer, one all-reduce per step
        self.buf = torch.zeros(self.n_params + 2, dtype=torch.float32, device=self.device)
        self._staged = self.dist is not None and self.dist.get_backend() != "nccl"
        self._host = torch.zeros(self.n_params + 2, dtype=torch.float32).pin_memory() if self._staged else None
        self._loss_sum = torch.zeros(2, dtype=torch.float64, device=self.device)
        self._steps = 0
        self._data = None

    # ---- the data set of an iteration and an epoch's order: uploaded once, gathered on the device per step
    def set_data(self, my_bb, op_bb, target_pi, target_v):
        t = self._torch
        self._data = (t.from_numpy(np.ascontiguousarray(my_bb, dtype=np.uint64).view(np.int64)).to(self.device),
                      t.from_numpy(np.ascontiguousarray(op_bb, dtype=np.uint64).view(np.int64)).to(self.device),
                      t.from_numpy(np.ascontiguousarray(target_pi, dtype=np.float32).reshape(-1, 9)).to(self.device),
                      t.from_numpy(np.ascontiguousarray(target_v, dtype=np.float32).reshape(-1, 3)).to(self.device))

    def step_indices(self, idx, lr):
        """One optimiser step on the samples `idx` (THIS rank's shard of the global batch) of the uploaded data set."""
        t = self._torch
        ix = t.as_tensor(np.ascontiguousarray(idx, dtype=np.int64), device=self.device)
        my, op, tpi, tv = (a.index_select(0, ix).contiguous() for a in self._data)
        self._step_device(my, op, tpi, tv, lr)

    def epoch(self, order, batch, lr):
        """len(order) // batch optimiser steps over the uploaded data set in the given order (THIS rank's shard of every global batch):
        the order is uploaded once, every step gathers its minibatch on the device and runs gradients -> all-reduce -> Adam in stream
        order; the host only enqueues. Returns the number of steps."""
        t = self._torch
        od = t.as_tensor(np.ascontiguousarray(order, dtype=np.int64), device=self.device)
        n = int(od.numel()) // int(batch)
        for s in range(n):
            ix = od[s * batch:(s + 1) * batch]
            my, op, tpi, tv = (a.index_select(0, ix) for a in self._data)
            self._step_device(my, op, tpi, tv, lr)
        return n

    def step(self, my_bb, op_bb, target_pi, target_v, lr):
        """One optimiser step; the arguments are THIS rank's shard of the batch (host arrays). Returns the (pi_loss, v_loss) of
        the global batch (mean over ranks of the per-shard means) — which costs a device synchronisation; `step_indices` +
        `take_losses` keep that off the step's critical path."""
        t = self._torch
        my = t.from_numpy(np.ascontiguousarray(my_bb, dtype=np.uint64).view(np.int64)).to(self.device)
        op = t.from_numpy(np.ascontiguousarray(op_bb, dtype=np.uint64).view(np.int64)).to(self.device)
        tpi = t.from_numpy(np.ascontiguousarray(target_pi, dtype=np.float32)).to(self.device)
        tv = t.from_numpy(np.ascontiguousarray(target_v, dtype=np.float32)).to(self.device)
        self._step_device(my, op, tpi, tv, lr)
        return (self.buf[self.n_params:] / self.world).cpu().numpy()   # (the read-back waits for the step)

    def _step_device(self, my, op, tpi, tv, lr):
        """gradients -> all-reduce -> Adam in ONE stream order (torch's current stream, which is also where the batch was gathered and
        where RCCL orders its collective): the gradient kernel writes its two loss sums behind the gradients, so they ride in the
        same 122 KB message; no host wait, no host-to-device copy, no synchronisation per step (13 us of compute per step would
        otherwise sit behind four of them). Under a host-staged backend (gloo: CPU tests, dry runs) the message crosses the host."""
        t = self._torch
        st = t.cuda.current_stream(self.device).cuda_stream
        self.engine.train_gradients_enqueue(st, my.data_ptr(), op.data_ptr(), tpi.data_ptr(), tv.data_ptr(), int(my.numel()),
                                            self.buf.data_ptr(), self.buf.data_ptr() + 4 * self.n_params)
        if self.dist is not None:
            if self._staged:
                self._host.copy_(self.buf)      # (stream-ordered copy to pinned memory, then the host waits for it)
                t.cuda.current_stream(self.device).synchronize()
                self.dist.all_reduce(self._host)
                self.buf.copy_(self._host, non_blocking=True)
            else:
                self.dist.all_reduce(self.buf)  # RCCL over xGMI: one 122 KB message (latency-bound), ordered behind the kernel
        self._loss_sum += self.buf[self.n_params:].double() / self.world
        self._steps += 1
        self.engine.train_apply_enqueue(st, self.buf.data_ptr(), lr, grad_scale=1.0 / self.world)

    def take_losses(self):
        """(sum of the per-step global (pi_loss, v_loss), steps) since the last call — one read-back per epoch instead of per step."""
        out = self._loss_sum.cpu().numpy().copy(), self._steps
        self._loss_sum.zero_()
        self._steps = 0
        return out

    def publish(self):
        """The trained network becomes the engine's self-play network (model_{i+1}.ot of alpha_zero.rs:97)."""
        self._torch.cuda.current_stream(self.device).synchronize()   # the steps ran on torch's stream, the hand-off runs on the engine's
        self.engine.trainer_publish_weights()

    def state(self):
        self._torch.cuda.current_stream(self.device).synchronize()
        return self.engine.trainer_state()
