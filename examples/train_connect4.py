#!/usr/bin/env python3
"""The reference's learning loop (synthesis/src/alpha_zero.rs:16-118, configured like study-connect4/src/main.rs:11-52)
with every heavy step on the MI355X: self-play (fused MCTS + network kernel), replay de-duplication (sort + segmented
reduce) and the optimiser steps (HIP forward/backward/Adam), the trained weights going back to the self-play network on
the device. Host code only moves indices: the replay buffer lives in numpy arrays, batches are drawn by a seeded
permutation (BatchRandSampler, data.rs:6-64, drop_last = true).

    python examples/train_connect4.py --iterations 3 --games-per-train 4096 --explores 200
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_connect4.py ...
        one rank per GPU. Default (synthesis_amd.learner.LearningLoop): every rank plays its share of the games (remainders
        included), the new positions are gathered on rank 0, which owns the replay buffer, de-duplicates it and trains with the
        persistent epoch kernel; the new parameters are broadcast once per iteration. The trained weights do not depend on the
        number of ranks.
        --data-parallel (synthesis_amd.learner.DataParallelLearner): the new games are all-gathered so every rank holds the SAME
        replay buffer and de-duplicates it to the same unique states; every global batch of --batch-size samples is split evenly
        over the ranks and the step is data-parallel with ONE gradient all-reduce over RCCL per step. Slower at the reference's
        batch of 32 (a step is 13 us of compute); there for large batches and as the other reading of BASELINE configs[4].

Differences from the reference: one StdRng stream per game, torch's randperm replaced by numpy's seeded permutation, and by
default the deterministic Fpu::Const(1.0) of the parity configuration; --reference-fpu selects the reference's own
Fpu::Func(|| Normal(1.0, 0.1)) (main.rs:43-47), sampled on the device from per-tree StdRng streams instead of thread_rng.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def lr_at(schedule, i_iter):
    """alpha_zero.rs:62-69: the last (iteration, lr) entry whose iteration is <= i_iter + 1"""
    lr = schedule[0][1]
    for it, v in schedule:
        if it <= i_iter + 1:
            lr = v
    return lr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=3)            # num_iterations (main.rs:16: 200)
    ap.add_argument("--games-per-train", type=int, default=4096)    # main.rs:27: 1000
    ap.add_argument("--games-to-keep", type=int, default=20000)     # main.rs:26
    ap.add_argument("--explores", type=int, default=200)            # main.rs:30: 1600
    ap.add_argument("--epochs", type=int, default=2)                # main.rs:21: 20
    ap.add_argument("--batch-size", type=int, default=32)           # main.rs:22
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--concurrent", type=int, default=65536)
    ap.add_argument("--eval-games", type=int, default=0, help="after every iteration: this many games as each colour against "
                                                              "the VanillaMCTS baselines (evaluator.rs:52-77), rank 0 only")
    ap.add_argument("--eval-explores", type=int, default=0, help="explores of the network player in evaluation (0 = --explores)")
    ap.add_argument("--eval-opponents", default="200,800", help="explores of the VanillaMCTS opponents")
    ap.add_argument("--out", default="")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one GPU per rank) | gloo (every rank on GPU 0: tests)")
    ap.add_argument("--reference-fpu", action="store_true", help="self-play with Fpu::Func(|| Normal(1.0, 0.1)) (main.rs:43-47)")
    ap.add_argument("--dump-weights", default="", help="write the final weights of every rank to <prefix>.rank<r>.npy")
    ap.add_argument("--sampler", default="numpy", choices=["numpy", "torch"], help="order of an epoch's batches: numpy's PCG64 or "
                    "libtorch's randperm stream (what the reference's BatchRandSampler draws, data.rs:29); learning loop only")
    ap.add_argument("--logs", default="", help="directory for what the reference writes per iteration (alpha_zero.rs:37,97-100): "
                    "models/model_{i}.ot and latest_{states,pis,vs}.npy (the learning loop's rank 0)")
    ap.add_argument("--net", default="mlp", choices=["mlp", "conv"], help="mlp = the reference's Connect4Net; conv = Connect4ConvNet "
                    "(north_star's Conv2d over the bitplanes + Linear heads)")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16"], help="arithmetic of the conv learner's gradient step: bf16 = bf16 "
                    "matrix cores with f32 accumulation, master weights and Adam (BASELINE configs[4]: \"bf16 conv\"; --net conv only)")
    ap.add_argument("--data-parallel", action="store_true", help="gradient all-reduce per optimiser step on every rank instead of "
                    "the learner on rank 0 + one weight broadcast per iteration")
    args = ap.parse_args()

    import torch  # noqa: F401  (before the engine: one HIP runtime per process)

    import synthesis_amd as sa
    from bench import make_conv_weights, make_weights
    from synthesis_amd import dist_util
    from synthesis_amd.learner import DataParallelLearner, LearningLoop

    rank, local_rank, world = dist_util.rank_info()
    dist = dist_util.init_process_group(args.dist_backend, local_rank) if world > 1 else None
    if args.dist_backend != "nccl":
        local_rank = 0
    if args.data_parallel and args.batch_size % world != 0:
        raise SystemExit(f"--batch-size {args.batch_size} must be a multiple of the world size {world}: every global batch is "
                         f"split evenly over the ranks so that the effective batch stays the reference's")
    lr_schedule = [(1, 1e-3), (20, 5e-4), (40, 1e-4), (60, 5e-5), (80, 1e-5)]  # main.rs:18
    cfg = sa.parity_rollout_config(args.explores)
    if args.reference_fpu:
        cfg.mcts_cfg = sa.reference_selfplay_mcts_config()

    eng = sa.Engine(concurrent_games=min(args.concurrent, max(16, args.games_per_train // world)), max_explores=args.explores,
                    device=local_rank)
    conv = args.net == "conv"
    blob = make_conv_weights(args.seed + 20260101) if conv else make_weights(args.seed + 20211003)  # P::new(&vs): identical on every rank
    hyper = dict(weight_decay=1e-6, policy_weight=1.0, value_weight=1.0)
    loop = learner = None
    if args.data_parallel:
        (eng.load_weights_conv if conv else eng.load_weights)(blob)
        learner = DataParallelLearner(eng, blob, dist=dist, device=local_rank, net=args.net, **hyper)
        if args.precision != "f32":
            eng.trainer_set_precision(args.precision)
        # replay buffer: positions as bitboards + targets + the game each step came from
        R = dict(my=np.zeros(0, np.uint64), op=np.zeros(0, np.uint64), pi=np.zeros((0, 9), np.float32),
                 v=np.zeros((0, 3), np.float32), gid=np.zeros(0, np.int64))
        games_played = 0
    else:
        loop = LearningLoop(eng, args.net, blob, dist=dist, device=local_rank, lr_schedule=lr_schedule, seed=args.seed,
                            precision=args.precision, logs_dir=args.logs or None, sampler=args.sampler, **hyper)
    log = []
    eval_eng = None
    for it in range(args.iterations):
        if loop is not None:
            rec = loop.iteration(cfg, args.games_per_train, args.games_to_keep, args.epochs, args.batch_size)
            sec = rec["seconds"]
            rec["selfplay_games_per_s"] = args.games_per_train / max(sec["selfplay"], 1e-9)
            if rank == 0:
                rec["train_steps_per_s"] = rec["optimiser_steps"] / max(sec["train"], 1e-9)
            current_weights = loop.weights
        else:
            t0 = time.perf_counter()
            # ---- gather_experience (alpha_zero.rs:120-179): this rank's share of the new games, seeds never reused
            off, count = sa.shard_games(args.games_per_train, rank, world)   # contiguous shards, remainders spread over the ranks
            first = it * args.games_per_train + off                           # global game index = seed offset, never reused
            sp = eng.selfplay(cfg, base_seed=args.seed, n_games=count, first_game=first)
            t_play = time.perf_counter() - t0
            n = sp["plies"]
            mask = np.arange(63)[None, :] < n[:, None]
            new = dict(my=sp["states_bb"][..., 0][mask], op=sp["states_bb"][..., 1][mask], pi=sp["pis"][mask], v=sp["vs"][mask],
                       gid=(first + np.arange(count))[:, None].repeat(63, 1)[mask])
            if dist is not None:
                # every rank appends ALL ranks' new games in rank order: one global replay buffer, identical everywhere
                parts = [None] * world
                dist.all_gather_object(parts, new)
                new = {k: np.concatenate([p_[k] for p_ in parts]) for k in new}
            R = {k: np.concatenate([R[k], new[k]]) for k in R}
            games_played += args.games_per_train
            keep = R["gid"] >= games_played - args.games_to_keep  # keep_last_n_games
            R = {k: a[keep] for k, a in R.items()}
            # ---- deduplicate on the GPU (data.rs:196-235)
            t1 = time.perf_counter()
            D = eng.replay_deduplicate(R["my"], R["op"], R["pi"], R["v"])
            t_dedup = time.perf_counter() - t1
            n_unique = D["num"].size
            lr = lr_at(lr_schedule, it)
            # ---- epochs of optimiser steps (alpha_zero.rs:72-94): the data set lives on the device, a step gathers its shard there
            t2 = time.perf_counter()
            learner.set_data(D["my_bb"], D["op_bb"], D["pis"], D["vs"])
            epoch_losses = []
            steps = 0
            share = args.batch_size // world
            for ep in range(args.epochs):
                perm = np.random.default_rng([args.seed, it, ep]).permutation(n_unique)   # the same permutation on every rank
                n_steps = n_unique // args.batch_size  # drop_last = true; n_unique is global, so every rank runs the same steps
                for b in range(0, n_steps * args.batch_size, args.batch_size):
                    learner.step_indices(perm[b + rank * share:b + (rank + 1) * share], lr)   # this rank's slice of the global batch
                el, _ = learner.take_losses()                                                  # one read-back per epoch
                steps += n_steps
                epoch_losses.append((el * args.batch_size / n_unique).tolist())
            t_train = time.perf_counter() - t2
            learner.publish()  # model_{i+1}: the next iteration's self-play runs on the trained network
            current_weights = None
            rec = dict(iteration=it + 1, lr=lr, games=int(args.games_per_train), steps_in_buffer=int(R["my"].size), unique=int(n_unique),
                       plies_per_game=float(n.mean()), draws=float((sp["final_kind"] == 1).mean()), optimiser_steps=steps,
                       epoch_losses=epoch_losses, seconds=dict(selfplay=round(t_play, 3), dedup=round(t_dedup, 3), train=round(t_train, 3)),
                       selfplay_games_per_s=args.games_per_train / t_play, train_steps_per_s=steps / max(t_train, 1e-9))
        evaluation = {}
        if args.eval_games > 0 and rank == 0:
            # evaluator.rs:52-77: the new model against every rollout baseline, as first and as second player
            from synthesis_amd import match

            opponents = [int(x) for x in args.eval_opponents.split(",") if x]
            my_explores = args.eval_explores or args.explores
            if eval_eng is None:  # its own engine: the opponents search deeper than self-play does
                eval_eng = sa.Engine(concurrent_games=max(16, args.eval_games), max_explores=max(opponents + [my_explores]),
                                     device=local_rank)
            (eval_eng.load_weights_conv if conv else eval_eng.load_weights)(current_weights if current_weights is not None else learner.state()["weights"])
            me = match.Player(f"model_{it + 1}", my_explores, cfg.mcts_cfg, cfg.action)
            for ox in opponents:
                opp = match.vanilla_player(ox)  # the evaluator's "VanillaMCTS<n>" baseline (FrozenMCTS over RolloutPolicy)
                r1, _ = match.play_match(eval_eng, me, opp, args.eval_games, seed=1000 * it)
                r2, _ = match.play_match(eval_eng, opp, me, args.eval_games, seed=1000 * it + 500)
                w, d, l, s, elo = match.score(np.concatenate([r1, -r2]))
                evaluation[opp.name] = dict(wins=w, draws=d, losses=l, score=round(s, 4), elo_diff=round(elo, 1))
        rec["evaluation"] = evaluation
        log.append(rec)
        if rank == 0:
            print(json.dumps(rec), flush=True)
    if rank == 0 and args.out:
        json.dump(log, open(args.out, "w"), indent=1)
    if args.dump_weights:
        np.save(f"{args.dump_weights}.rank{rank}.npy", loop.weights if loop is not None else learner.state()["weights"])
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
