#!/usr/bin/env python3
"""bench.py — self-play throughput of the MI355X engine on BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W
    N > 1 runs one process per GPU either way:
      * under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...):
        RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment;
      * started bare (python bench.py --gpus N): this process starts the N ranks itself as child processes BEFORE it
        touches the GPU (it never initialises HIP and never re-execs), forwards rank 0's JSON line and exits with the
        worst child status. A rank whose GPU is not visible fails loudly (SYN_ERR_NO_DEVICE) and takes the job down.

Metric   : self-play games/s (and leaf-evals/s, explores/s as extra fields), 9x7 Connect4, 800 explores per move,
           deterministic parity MCTS config (study-connect4/src/main.rs:58-66). Workload = BASELINE configs[2] (GPU-resident
           SoA MCTS node pool) at 196,608 concurrent games per GPU (768 per CU: the 12-wave lane-per-tree kernel);
           configs[1]'s 4096 concurrent games is measured in the same run and reported under "at_4096_concurrent_games".
Step     : one pass of the hot path over one batch = GAMES_PER_STEP (2,097,152: 10.7 per tree slot) self-play games per GPU played to completion by
           ONE launch of the fused kernel (finished games hand their tree slot to the next game index, so the slots stay busy).
Scaling  : weak — every rank plays its own GAMES_PER_STEP games per step (games share nothing; no collective on the
           data path). Timed region = barrier + device sync on both sides, max over ranks.
Roofline : the fused kernel's achieved rate against both roofs — f32 MFMA (60,288 FLOP per leaf evaluation,
           SURVEY.md §8d) and HBM (event-counted algorithmic bytes, formula of SURVEY.md §8d) — the nearer roof is
           reported as "roofline", the other under "roofline_other". Kernel time = HIP events on the engine's stream.
CPU side : the oracle (restated reference CPU path, thread-per-worker with PolicyWithCache like gather_experience)
           timed on this box's host cores on a bounded sample of the same workload — rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_EVAL = 60288           # 2 * (63*128 + 128*96 + 96*64 + 64*48 + 48*12)  
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: Peak FP32 (matrix)
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E peak BW (spec)
PEAK_F16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: BF16/F16 dense matrix peak (~2.5 PF; the 5 PF headline figure includes 2:1 sparsity)
F16X2_FLOP_PER_EVAL = 184320    # the f16x2 tile as executed: 180 v_mfma_f32_16x16x32_f16 x 16,384 FLOP per 16 positions (3 products, inputs padded to 32s)


def make_weights(seed=20211003):
    """Connect4Net shapes (policies.rs:20-24), fixed-seed U(+-1/sqrt(fan_in)) init — the same blob as
    tests/golden/c4net_blob_f32.npy (tests/golden/make_golden.py)."""
    dims = [63, 128, 96, 64, 48, 12]
    rng = np.random.RandomState(seed)
    parts = []
    for i in range(5):
        bound = 1.0 / np.sqrt(dims[i])
        parts.append(rng.uniform(-bound, bound, size=(dims[i + 1], dims[i])).astype(np.float32).ravel())
        parts.append(rng.uniform(-bound, bound, size=(dims[i + 1],)).astype(np.float32).ravel())
    return np.concatenate(parts)


CONV_FLOP_PER_EVAL = 54592      # Connect4ConvNet as slimnn executes it: 2 * (16 * 2 * 475 in-board conv taps + 1008 * 12)


def make_conv_weights(seed=20260101):
    """Connect4ConvNet (include/synthesis_amd.h: Conv2d<2,16,3,pad 1> + ReLU + Linear<1008,12>), fixed-seed U(+-1/sqrt(fan_in))
    init; the head weights x4 so that the priors are not flat. Blob: conv.weight, conv.bias, head.weight, head.bias."""
    rng = np.random.default_rng(seed)
    conv_w = rng.uniform(-1, 1, 288) / np.sqrt(18.0)
    conv_b = rng.uniform(-1, 1, 16) / np.sqrt(18.0)
    head_w = rng.uniform(-1, 1, 12 * 1008) / np.sqrt(1008.0) * 4.0
    head_b = rng.uniform(-1, 1, 12) / np.sqrt(1008.0)
    return np.concatenate([conv_w, conv_b, head_w, head_b]).astype(np.float32)


def algorithmic_bytes(c):
    """SURVEY.md §8(d), per event: select level = 12 B parent + 20 B per child scanned; expand = 16 B state +
    48 B per new node + 8 B parent update; NN I/O = 64 B per leaf evaluation (0 when fused — counted as 0 here);
    backprop level = 40 B + 4 B per child solution read by the solver."""
    return (12 * c["select_levels"] + 20 * c["children_scanned"] + 16 * c["expansions"] + 48 * c["new_nodes"]
            + 8 * c["expansions"] + 40 * c["backprop_levels"] + 4 * c["solver_children"])


SELFPLAY_KERNEL_SOURCES = ("device_common.cuh", "mcts.cuh", "mlp.cuh", "engine_kernels.cuh", "lane_kernel.cuh",
                           "noise.cuh", "zig_tables.cuh", "fpu_normal_table.cuh", "convnet.cuh", "f16x2_tile.cuh", "free_kernel.cuh")


def kernel_source_hash():
    """sha256 (16 hex digits) over the HIP sources the fused self-play kernels are compiled from (their include closure): ties a
    committed PMC summary to the kernels it measured."""
    import hashlib

    h = hashlib.sha256()
    for f in SELFPLAY_KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "synthesis_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(args, key="traffic_bytes_per_launch", games=None):
    """HBM bytes per launch of the fused kernel from the committed rocprofv3 PMC passes (profiles/rNN_pmc.json, produced
    by tools/collect_profiles.sh on this exact bench command). The counters cannot be read inside this process, so the
    figure comes from that committed run; it is reported ONLY when the summary was taken on this configuration AND on
    these kernel sources (csrc hash) — otherwise None."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    if d.get("csrc_sha16") != kernel_source_hash():
        return None, None
    if d.get("bench_config") != [args.concurrent, args.games_per_step, args.explores]:
        return None, None
    if games is not None and (d.get("extra_leg_games") or {}).get(key) != games:
        return None, None
    return d.get(key), os.path.relpath(files[-1], ROOT)


def host_cpu_budget():
    """CPUs this process may actually use: the smaller of the affinity mask and the cgroup CPU quota (the GPU boxes give
    a container 16 CPUs of quota on a 256-thread host; timing 256 threads there measures the throttle, not the code)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def cpu_baseline(blob, explores, sample_games, threads):
    from tests import oracle_lib

    oracle = oracle_lib.load()
    cfg = oracle_lib.parity_rollout_config(explores)
    r = oracle.c4_selfplay(cfg, blob, base_seed=0, n_games=sample_games, threads=threads, use_cache=True,
                           nn_mode=oracle.ACC_SLIMNN, outputs=False)
    secs = r["seconds"]
    c = r["counters"]
    return {
        "value": sample_games / secs, "unit": "games/s", "cores": threads, "kind": "port",
        "sample": f"{sample_games} games of the same workload (800-explore self-play, parity config, PolicyWithCache on, "
                  f"one policy+cache per thread) in {secs:.2f} s wall on {threads} host threads",
        "leaf_evals_per_s": c["policy_evals"] / secs, "explores_per_s": c["explores"] / secs,
        "cache_hit_rate": c["cache_hits"] / max(1, c["cache_hits"] + c["cache_misses"]),
    }


def cpu_baseline_single_thread(blob, explores, games=16):
    """BASELINE configs[0]: the reference's CPU path on ONE thread (study-connect4/src/main.rs:85-86 pins libtorch to one thread; one
    worker of gather_experience) — a few games of the same workload without and with PolicyWithCache (alpha_zero.rs:196-198)."""
    from tests import oracle_lib

    oracle = oracle_lib.load()
    cfg = oracle_lib.parity_rollout_config(explores)
    out = {"cores": 1, "kind": "port", "games": games, "unit": "games/s"}
    for name, cache in (("policy_cache_off", False), ("policy_cache_on", True)):
        r = oracle.c4_selfplay(cfg, blob, base_seed=0, n_games=games, threads=1, use_cache=cache, nn_mode=oracle.ACC_SLIMNN, outputs=False)
        c = r["counters"]
        out[name] = {"value": games / r["seconds"], "seconds": r["seconds"], "leaf_evals_per_s": c["policy_evals"] / r["seconds"],
                     "explores_per_s": c["explores"] / r["seconds"]}
    out["sample"] = f"{games} games of the same workload ({explores}-explore self-play, parity config, slimnn accumulation order) on one host thread, twice"
    return out


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU, RANK = LOCAL_RANK = 0..N-1) as
    children of this process, which has not touched the GPU and will not. Returns the worst exit status."""
    import socket
    import subprocess

    with socket.socket() as s:  # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    worst = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0:
                worst = worst or rc
                print(f"bench.py: rank {r} exited with status {rc}; stopping the other ranks", file=sys.stderr, flush=True)
                for o in sorted(pending):  # exactly the PIDs started above
                    procs[o].terminate()
        time.sleep(0.05)
    return worst


def main():
    t_run0 = time.perf_counter()

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--concurrent", type=int, default=196608,
                    help="concurrent games (tree slots) per GPU: 768 per CU = 12 waves of 64 trees, what the headline launch plays on "
                         "(rounds 1-5 created 262,144 slots and played on 196,608 of them); BASELINE configs[1] names 4096, reported as extra")
    ap.add_argument("--games-per-step", type=int, default=2097152, help="self-play games per GPU per step (10.7 per tree slot at the default engine size)")
    ap.add_argument("--no-4096", action="store_true", help="skip the extra 4096-concurrent-games measurement")
    ap.add_argument("--explores", type=int, default=800)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-warmup", action="store_true", help="warm-up steps at full size (the kernel-trace pass: every launch of the "
                    "kernel then has the timed steps' duration, so the trace's per-kernel average is the step's)")
    ap.add_argument("--skip-counted", action="store_true", help="profiling passes: only the warm-up and timed launches, a reduced line")
    ap.add_argument("--no-learner-loop", action="store_true", help="skip the three iterations of the N-rank learning loop (learner_loop)")
    ap.add_argument("--no-policy-cache", action="store_true", help="skip the extra PolicyWithCache measurement")
    ap.add_argument("--only-policy-cache", "--only-extra-legs", dest="only_policy_cache", action="store_true",
                    help="run nothing but the extra self-play legs (PolicyWithCache, reference configuration, trained network, conv network) "
                         "at their bench size: the command tools/collect_profiles.sh profiles for their rooflines")
    ap.add_argument("--no-extras", action="store_true", help="skip the trained-weights / replay-output / reference-config / conv / tail / learner legs")
    ap.add_argument("--time-budget-s", type=float, default=1500.0, help="wall-clock budget of the whole run: the extra legs (never the timed "
                    "steps, the roofline or the CPU baseline) are skipped, least important first, once they would overrun it")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one GPU per rank) | gloo (dry run: every rank "
                                                           "on GPU 0, used to exercise the N>1 path on a 1-GPU box)")
    ap.add_argument("--cpu-sample-games", type=int, default=0, help="0 = 64 games per worker thread")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = two per CPU the cgroup quota / affinity mask allows")
    ap.add_argument("--check-launch", action="store_true",
                    help="start the ranks, rendezvous, barrier and reduce exactly as a bench run does, print the rank "
                         "layout and stop before any GPU work (launch-plumbing test; prints no bench line)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))   # (the ranks inherit this process's stdout as it is)
    # ONE JSON line on stdout, whatever the libraries print: RCCL writes its version banner to stdout when a communicator is created
    # (every N > 1 run; the learner's world-size-1 leg at N = 1). From here on file descriptor 1 points at stderr and the line goes
    # to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    def emit(obj):
        real_stdout.write(json.dumps(obj) + "\n")
        real_stdout.flush()

    import torch  # device sync + (N > 1) the RCCL barrier / max-reduce; the engine itself does not use torch
                  # (import it BEFORE the engine library so the process holds one HIP runtime: torch's)

    import synthesis_amd as sa
    from synthesis_amd import dist_util

    rank, local_rank, world = dist_util.rank_info()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.check_launch:
        dist = dist_util.init_process_group("gloo") if world > 1 else None
        dist_util.barrier(dist)
        _, (ranks, lsum) = dist_util.reduce_scalars(dist, "cpu", 0.0, [1, local_rank])
        if rank == 0:
            emit({"check_launch": True, "n_gpus": world, "ranks_joined": ranks, "local_rank_sum": lsum})
        if dist is not None:
            dist.destroy_process_group()
        return
    if args.dist_backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} GPU(s) visible (one rank per GPU; "
                         f"--dist-backend gloo shares GPU 0 for a dry run)")
    dist = dist_util.init_process_group(args.dist_backend, local_rank) if world > 1 else None
    reduce_device = f"cuda:{local_rank}"
    if args.dist_backend != "nccl":
        local_rank, reduce_device = 0, "cpu"   # dry run: all ranks share GPU 0, scalars reduced over gloo

    blob = make_weights()
    cfg = sa.parity_rollout_config(args.explores)
    gps = args.games_per_step

    def hbm_roofline(alg_bytes, kernel_ms, traffic, src, mfma_flops):
        """HBM roofline object of a tree-bound leg. `achieved` = algorithmic bytes / kernel time (the contract's definition);
        `frac` is the counter-backed figure whenever the PMC passes of this configuration exist and saw FEWER bytes than the
        model charges (re-reads served by L2 / MALL are not HBM traffic: the model would overstate the HBM rate), else the model's."""
        secs = kernel_ms * 1e-3
        gbs = alg_bytes / secs / 1e9
        frac_model = gbs / PEAK_HBM_GBS
        frac_pmc = (traffic / secs / 1e9 / PEAK_HBM_GBS) if traffic else None
        use_pmc = frac_pmc is not None and frac_pmc < frac_model
        return {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": frac_pmc if use_pmc else frac_model, "frac_basis": "pmc" if use_pmc else "model",
                "frac_model": frac_model, "frac_pmc": frac_pmc, "traffic": traffic, "traffic_source": src,
                "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms_avg": kernel_ms,
                "mfma_frac_beside_it": mfma_flops / secs / 1e12 / PEAK_F32_MFMA_TFLOPS}

    def policy_cache_leg(n_games, trained=None, reference_mcts=False):
        """The reference's run_n_games wraps the policy in PolicyWithCache (alpha_zero.rs:197-198), and so does the CPU
        baseline; the headline evaluates every leaf with the network, this is the same workload with the device-side cache
        (2^28 entries, 17 GB): one launch of n_games games at the headline concurrency. Its bound is the tree phases (two thirds
        of the matrix work disappears), so it carries an HBM roofline object of its own.
        trained + reference_mcts = the configuration the reference actually runs self-play with (study-connect4/src/main.rs:37-49:
        a trained network behind PolicyWithCache, Fpu::Func(|| Normal(1.0, 0.1)))."""
        e3 = sa.Engine(concurrent_games=args.concurrent, max_explores=args.explores, device=local_rank, policy_cache_log2=28)
        out3 = {}
        for name, w3, cfg3 in (("with_policy_cache", blob, cfg),
                               ("reference_selfplay_config", trained,
                                sa.parity_rollout_config(args.explores, mcts_cfg=sa.reference_selfplay_mcts_config()) if reference_mcts else None)):
            if w3 is None or cfg3 is None:
                continue
            e3.load_weights(w3)   # (a new network empties the table)
            e3.selfplay(cfg3, base_seed=2, n_games=args.concurrent, outputs=False)
            t1 = time.perf_counter()
            r3 = e3.selfplay(cfg3, base_seed=2, n_games=n_games, first_game=args.concurrent, outputs=False)
            dt = time.perf_counter() - t1
            hits, misses = e3.last_cache_stats()
            shape3 = e3.last_launch_shape()
            # event counts of a 32,768-game sample of the same games (the cache never changes a tree), scaled to the launch
            ns = min(32768 if name == "with_policy_cache" else 8192, n_games)
            c3 = e3.selfplay(cfg3, base_seed=2, n_games=ns, first_game=args.concurrent, outputs=False, counters=True)["counters"]
            scale = n_games / ns
            # algorithmic bytes: the tree traffic of SURVEY §8d plus one 64-byte table entry read per Policy::eval call and
            # one written per miss
            alg = algorithmic_bytes(c3) * scale + 64 * (hits + misses) + 64 * misses
            key = "cache_traffic_bytes_per_launch" if name == "with_policy_cache" else "reference_traffic_bytes_per_launch"
            traffic, src = measured_traffic(args, key, n_games)
            o = {"games_per_s": n_games / dt, "kernel_ms": r3["kernel_ms"], "games": n_games, "concurrent_games": args.concurrent,
                 "plies_per_game": float(r3["plies"].mean()), "table_entries_log2": 28, "hit_rate": hits / max(1, hits + misses),
                 "network_evals_per_s": misses / dt, "policy_eval_calls_per_s": (hits + misses) / dt,
                 "select_levels_per_explore": c3["select_levels"] / max(1, c3["explores"]),
                 "launch_shape": list(shape3),
                 "roofline": hbm_roofline(alg, r3["kernel_ms"], traffic, src, misses * FLOP_PER_EVAL)}
            o["roofline_other"] = {"bound": "mfma", "achieved": misses * FLOP_PER_EVAL / (r3["kernel_ms"] * 1e-3) / 1e12,
                                   "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                   "frac": misses * FLOP_PER_EVAL / (r3["kernel_ms"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                   "traffic": traffic, "traffic_source": src, "kernel_ms_avg": r3["kernel_ms"], "flop_per_leaf_eval": FLOP_PER_EVAL}
            if name == "reference_selfplay_config":
                o["config"] = ("trained checkpoint tests/golden/c4net_trained_f32.npy + PolicyWithCache (2^28 entries) + "
                               "Fpu::Func(Normal(1.0, 0.1)) on the device (study-connect4/src/main.rs:37-49), %d explores" % args.explores)
                o["roofline_mfma"] = {"bound": "mfma", "achieved": misses * FLOP_PER_EVAL / (r3["kernel_ms"] * 1e-3) / 1e12,
                                      "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                      "frac": misses * FLOP_PER_EVAL / (r3["kernel_ms"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                      "kernel_ms_avg": r3["kernel_ms"], "flop_per_leaf_eval": FLOP_PER_EVAL}
            if name == "reference_selfplay_config":
                # the same configuration in the f16x2 network arithmetic (see with_f16x2_network): games/s only
                try:
                    e3.set_network_arithmetic("f16x2")
                    e3.selfplay(cfg3, base_seed=2, n_games=args.concurrent, outputs=False)
                    t1 = time.perf_counter()
                    r3h = e3.selfplay(cfg3, base_seed=2, n_games=n_games, first_game=args.concurrent, outputs=False)
                    dth = time.perf_counter() - t1
                    hh, hm = e3.last_cache_stats()
                    o["with_f16x2_network"] = {"games_per_s": n_games / dth, "kernel_ms": r3h["kernel_ms"], "games": n_games,
                                               "hit_rate": hh / max(1, hh + hm), "launch_shape": list(e3.last_launch_shape())}
                    e3.set_network_arithmetic("f32")
                except Exception as ex:  # noqa: BLE001
                    o["with_f16x2_network"] = {"error": str(ex)[:200]}
            out3[name] = o
        e3.close()
        return out3

    tpath = os.path.join(ROOT, "tests", "golden", "c4net_trained_f32.npy")
    trained_blob = np.load(tpath) if os.path.exists(tpath) else None
    gx = max(args.concurrent, gps // 4)   # games of the two side measurements (replay outputs to the host, launch tail): a quarter of a step
    gh = max(args.concurrent, gps // 2)   # games of every self-play leg with a roofline of its own: half a step (1,048,576 by default)

    def leg_rooflines(counters, sample_games, n_games, kernel_ms, traffic_key, flop_per_eval, extra_alg_bytes=0.0):
        """Both roofline objects of an extra leg from the event counts of a sample of its games (scaled to the launch), its kernel time
        and — when profiles/rNN_pmc.json holds a pass of this leg on these kernel sources and this size — the HBM traffic the counters saw."""
        scale = n_games / sample_games
        secs = kernel_ms * 1e-3
        flops = counters["policy_evals"] * scale * flop_per_eval
        alg = algorithmic_bytes(counters) * scale + extra_alg_bytes
        traffic, src = measured_traffic(args, traffic_key, n_games)
        mf = {"bound": "mfma", "achieved": flops / secs / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
              "frac": flops / secs / 1e12 / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": src, "kernel_ms_avg": kernel_ms,
              "flop_per_leaf_eval": flop_per_eval}
        hb = hbm_roofline(alg, kernel_ms, traffic, src, flops)
        return (mf, hb) if mf["frac"] >= hb["frac"] else (hb, mf)

    def trained_leg(eng, n_games, first):
        """a TRAINED network (tests/golden/c4net_trained_f32.npy, produced by examples/train_connect4.py — see tests/golden/README):
        priors are sharp, so trees are deep and narrow instead of the wide shallow trees of the random-init network (SURVEY §8d asks
        for both); same kernel"""
        eng.load_weights(trained_blob)
        eng.selfplay(cfg, base_seed=0, n_games=args.concurrent, first_game=first, outputs=False)
        t1 = time.perf_counter()
        rt = eng.selfplay(cfg, base_seed=0, n_games=n_games, first_game=first + args.concurrent, outputs=False)
        dt4 = time.perf_counter() - t1
        shape4 = list(eng.last_launch_shape())
        ct = eng.selfplay(cfg, base_seed=0, n_games=32768, first_game=first + args.concurrent, outputs=False, counters=True)["counters"]
        near, other = leg_rooflines(ct, 32768, n_games, rt["kernel_ms"], "trained_traffic_bytes_per_launch", FLOP_PER_EVAL)
        eng.load_weights(blob)
        return {"games_per_s": n_games / dt4, "games": n_games, "kernel_ms": rt["kernel_ms"], "plies_per_game": float(rt["plies"].mean()),
                "select_levels_per_explore": ct["select_levels"] / max(1, ct["explores"]),
                "backprop_levels_per_explore": ct["backprop_levels"] / max(1, ct["explores"]),
                "max_depth": ct["max_depth"], "leaf_evals_per_explore": ct["policy_evals"] / max(1, ct["explores"]),
                "solved_leaf_share": ct["solved_hits"] / max(1, ct["explores"]),
                "mfma_frac": (ct["policy_evals"] / 32768.0) * (n_games / dt4) * FLOP_PER_EVAL / 1e12 / PEAK_F32_MFMA_TFLOPS,
                "launch_shape": shape4, "roofline": near, "roofline_other": other}

    def f16x2_leg(eng, n_games, first):
        """Connect4Net in the f16x2 arithmetic (SYN_NET_ARITH_F16X2: every operand a pair of f16 numbers, products on
        v_mfma_f32_16x16x32_f16; include/synthesis_amd.h): the headline workload, one launch of n_games games, then the trained checkpoint.
        The arithmetic is not the headline's (23-bit operands instead of 24, results equal to the oracle's ACC_F16X2 and as close to f64 as
        f32 is: profiles/r05_f16_split.txt), so this is a leg of its own and never `value`. Two roofs — HBM against the tree traffic, the f16 matrix
        peak against the MFMAs it executes (3 products x zero-padded inputs: 184,320 FLOP per evaluation) — and, for comparison only, the
        f32-equivalent rate (the 60,288 FLOP the network needs, the figure the f32 kernel is priced by)."""
        eng.set_network_arithmetic("f16x2")
        res = {}
        try:
            return _f16x2_leg_body(eng, n_games, first, res)
        finally:   # whatever happened, the shared engine goes back to the headline's arithmetic and weights for the legs that follow
            eng.set_network_arithmetic("f32")
            eng.load_weights(blob)

    def _f16x2_leg_body(eng, n_games, first, res):
        for name, w in (("random_init", blob), ("trained_checkpoint", trained_blob)):
            if w is None:
                continue
            eng.load_weights(w)
            eng.selfplay(cfg, base_seed=0, n_games=args.concurrent, first_game=first, outputs=False)
            t1 = time.perf_counter()
            rt = eng.selfplay(cfg, base_seed=0, n_games=n_games, first_game=first + args.concurrent, outputs=False)
            dt = time.perf_counter() - t1
            shape = list(eng.last_launch_shape())
            ct = eng.selfplay(cfg, base_seed=0, n_games=32768, first_game=first + args.concurrent, outputs=False, counters=True)["counters"]
            key = "f16x2_traffic_bytes_per_launch" if name == "random_init" else "f16x2_trained_traffic_bytes_per_launch"
            a, b = leg_rooflines(ct, 32768, n_games, rt["kernel_ms"], key, FLOP_PER_EVAL)
            hb, f32eq = (a, b) if a["bound"] == "hbm" else (b, a)
            evals = ct["policy_evals"] * (n_games / 32768.0)
            secs = rt["kernel_ms"] * 1e-3
            # the kernel's matrix work runs on the f16 pipe: that roof is priced by the MFMAs executed; the tree traffic against HBM is the
            # nearer one (the kernel is bound by the tree phases' instruction issue: DESIGN §6.1c); the f32 figure is for comparison only
            f16 = {"bound": "mfma", "achieved": evals * F16X2_FLOP_PER_EVAL / secs / 1e12, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                   "frac": evals * F16X2_FLOP_PER_EVAL / secs / 1e12 / PEAK_F16_MFMA_TFLOPS, "flop_per_leaf_eval_executed": F16X2_FLOP_PER_EVAL,
                   "kernel_ms_avg": rt["kernel_ms"], "traffic": hb.get("traffic"), "traffic_source": hb.get("traffic_source")}
            hb.pop("mfma_frac_beside_it", None)
            near, other = (hb, f16) if hb["frac"] >= f16["frac"] else (f16, hb)
            res[name] = {"games_per_s": n_games / dt, "games": n_games, "kernel_ms": rt["kernel_ms"], "plies_per_game": float(rt["plies"].mean()),
                         "leaf_evals_per_s": evals / dt, "select_levels_per_explore": ct["select_levels"] / max(1, ct["explores"]),
                         "launch_shape": shape, "roofline": near, "roofline_other": other,
                         "f32_equivalent": {"tflops_at_60288_flop_per_eval": f32eq["achieved"], "frac_of_the_f32_matrix_peak": f32eq["frac"],
                                            "note": "what the f32 kernel is priced by; the f16x2 kernel does not run on that pipe"}}
            first += args.concurrent + n_games
        out16 = res.get("random_init", {})
        out16["dtype"] = "f16x2 (operands: pairs of f16, 22-23 significand bits; accumulation f32)"
        out16["parity"] = "searches and games bit-identical to the oracle run in ACC_F16X2 (tests/test_gpu_f16x2.py); network within 1e-5 / 3 of slimnn order"
        if "trained_checkpoint" in res:
            out16["trained_checkpoint"] = res["trained_checkpoint"]
        return out16

    def conv_leg(_eng, n_games, first):
        """the conv policy/value network of north_star (Connect4ConvNet, convnet.cuh) behind the same Policy::eval: same MCTS
        configuration, fixed-seed init; its matrix-core tile is 567 MFMAs per 16 positions (Connect4Net: 476). An engine of its own at
        the size this network's launch shape wants (16 waves x 1,024 trees per CU = 262,144 slots; 12 x 768 measures 48.7k against 62.7k)."""
        eng = sa.Engine(concurrent_games=max(args.concurrent, 262144), max_explores=args.explores, device=local_rank)
        try:
            return _conv_leg_body(eng, n_games, first)
        finally:
            eng.close()

    def _conv_leg_body(eng, n_games, first):
        eng.load_weights_conv(make_conv_weights())
        eng.selfplay(cfg, base_seed=0, n_games=args.concurrent, first_game=first, outputs=False)
        t1 = time.perf_counter()
        rt = eng.selfplay(cfg, base_seed=0, n_games=n_games, first_game=first + args.concurrent, outputs=False)
        dt5 = time.perf_counter() - t1
        conv_shape = list(eng.last_launch_shape())
        cc = eng.selfplay(cfg, base_seed=0, n_games=32768, first_game=first + args.concurrent, outputs=False, counters=True)["counters"]
        evals_per_s = (cc["policy_evals"] / 32768.0) * (n_games / dt5)
        near, other = leg_rooflines(cc, 32768, n_games, rt["kernel_ms"], "conv_traffic_bytes_per_launch", CONV_FLOP_PER_EVAL)
        return {"concurrent_games": conv_shape[1] * conv_shape[2], "network": "Connect4ConvNet: Conv2d<2,16,3,pad 1> + ReLU + Linear<1008,12> (12,412 parameters), fixed-seed init",
                "games_per_s": n_games / dt5, "games": n_games, "kernel_ms": rt["kernel_ms"], "plies_per_game": float(rt["plies"].mean()),
                "leaf_evals_per_s": evals_per_s, "select_levels_per_explore": cc["select_levels"] / max(1, cc["explores"]),
                "flop_per_eval": CONV_FLOP_PER_EVAL,
                "mfma_frac": evals_per_s * CONV_FLOP_PER_EVAL / 1e12 / PEAK_F32_MFMA_TFLOPS,
                "launch_shape": conv_shape, "roofline": near, "roofline_other": other}

    if args.only_policy_cache:
        out = policy_cache_leg(gh, trained_blob, reference_mcts=True)
        eng = sa.Engine(concurrent_games=args.concurrent, max_explores=args.explores, device=local_rank)
        if trained_blob is not None:
            out["with_trained_weights"] = trained_leg(eng, gh, 0)
        out["with_conv_policy"] = conv_leg(eng, gh, 4 * gh)
        out["with_f16x2_network"] = f16x2_leg(eng, gh, 8 * gh)
        eng.close()
        if rank == 0:
            emit(out)
        return

    eng = sa.Engine(concurrent_games=args.concurrent, max_explores=args.explores, device=local_rank)
    eng.load_weights(blob)

    def barrier():
        torch.cuda.synchronize(local_rank)
        if dist is not None:
            dist.barrier()

    def step(i, **kw):
        first, count = dist_util.step_game_range(i, rank, world, gps)
        return eng.selfplay(cfg, base_seed=0, n_games=count, first_game=first, outputs=False, **kw)

    # warm-up steps play a quarter of a step's games (the first of them pages the kernel in and touches the whole node pool: every
    # tree slot plays one game; a full-size warm-up would only repeat the timed steps — at --warmup 5 that was 150 s of the driver's run)
    for i in range(args.warmup):
        first, count = dist_util.step_game_range(i, rank, world, gps)
        eng.selfplay(cfg, base_seed=0, n_games=count if args.full_warmup else max(min(count, args.concurrent), count // 4),
                     first_game=first, outputs=False)
    barrier()
    t0 = time.perf_counter()
    kernel_ms = []
    plies = 0
    for i in range(args.steps):
        r = step(args.warmup + i)
        kernel_ms.append(r["kernel_ms"])
        plies += int(r["plies"].sum())
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed, (plies,) = dist_util.reduce_scalars(dist, reduce_device, elapsed, [plies])

    # Event counts of exactly the games of the last timed step (trajectories are deterministic, so an instrumented
    # re-run outside the timed region gives the counts of the timed run).
    if args.skip_counted:
        # tools/collect_profiles.sh's counter passes only need the launches themselves: no instrumented re-run, no roofline objects
        if rank == 0:
            emit({"metric": "self-play games/sec, 9x7 Connect4", "value": gps * world * args.steps / elapsed, "unit": "games/s",
                  "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                  "kernel_ms_avg": float(np.mean(kernel_ms)), "roofline": None, "profile_pass": True,
                  "config": {"games_per_step_per_gpu": gps, "concurrent_games_per_gpu": args.concurrent,
                             "explores_per_move": args.explores}})
        eng.close()
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    rc = step(args.warmup + args.steps - 1, counters=True)
    c = rc["counters"]
    shape, sgrid, sthreads = eng.last_launch_shape()
    kernel_name = {1: "selfplay_kernel<WPS=1>", 2: "selfplay_kernel<WPS=2>", 3: "selfplay_kernel_quads",
                   4: "selfplay_kernel_lanes", 5: "selfplay_kernel_pc", 6: "selfplay_kernel_lanes2"}.get(shape, "?") + f" <<<{sgrid}, {sthreads}>>>"
    last_ms = kernel_ms[-1]
    avg_ms = float(np.mean(kernel_ms))
    # concurrent games = trees the launch plays on: one per lane of the lane-per-tree kernels, 16 per workgroup of the row kernels
    played_slots = min(args.concurrent, sgrid * sthreads if shape in (4, 6, 8) else sgrid * 16 * (sthreads // 256) if shape in (1, 2, 3, 7) else args.concurrent)

    # ---- BASELINE configs[4] (self-play on every GPU + the training step): three iterations of the learning loop in the shape that
    #      scales (synthesis_amd.learner.LearningLoop: every rank plays 8,192 games at 200 explores, rank 0 gathers, de-duplicates and
    #      trains with the persistent epoch kernel, the 122 KB of weights are broadcast) — every rank takes part, rank 0 reports the
    #      THIRD iteration's seconds per phase (the first two pay the first-touch page faults of the host buffers: a loop runs
    #      hundreds of iterations). Not part of the timed steps.
    loop_rec = None
    if not args.no_learner_loop:
        try:
            from synthesis_amd.learner import LearningLoop

            e4 = sa.Engine(concurrent_games=8192, max_explores=200, device=local_rank)
            loop = LearningLoop(e4, "mlp", blob, dist=dist, device=local_rank, seed=7, weight_decay=1e-6)
            cfg4 = sa.parity_rollout_config(200)
            recs = [loop.iteration(cfg4, 8192 * world, 20000 * world, 1, 32) for _ in range(3)]
            e4.close()
            if rank == 0:
                r4 = recs[-1]
                loop_rec = {"games_per_iteration": r4["games"], "explores": 200, "ranks": world, "iteration_reported": "third of three", "optimiser_steps": r4["optimiser_steps"],
                            "unique_positions": r4["unique"], "seconds": r4["seconds"],
                            "games_per_s_of_the_whole_iteration": r4["games"] / max(1e-9, r4["seconds"]["total"]),
                            "collectives": "one fixed-layout tensor gather of the new positions to rank 0 + one %d-byte weight broadcast per iteration (%s)"
                                           % (blob.size * 4, args.dist_backend if world > 1 else "none: one rank"),
                            "note": "the 1 -> 8 GPU curve of this loop has not been measured on an 8-GPU node (one GPU per box here)"}
                # What the loop does at 8 ranks, from THIS run's phase times (alpha_zero.rs:45-100 has ONE learner: every added rank adds
                # its games AND its share of rank 0's serial work): play is per rank (weak scaling); the gather moves 72 B per position
                # from 7 peers to rank 0 (xGMI: one ~153 GB/s link per peer, in parallel) and rank 0 unpacks 8 x as many positions
                # (priced at 5 GB/s of host copies, this run's pack / unpack rate); de-duplication and the epochs grow with the
                # positions (x 8); the 122 KB broadcast does not grow.
                sec = r4["seconds"]
                pos = float(r4["games"]) * float(r4["plies_per_game"]) if world == 1 else None   # new positions of one iteration
                if world == 1 and pos:
                    per_rank_bytes = 72.0 * pos
                    g8 = per_rank_bytes / 153e9 + 8 * per_rank_bytes / 5e9
                    tot8 = sec["selfplay"] + g8 + 8 * sec["dedup"] + 8 * sec["train"] + sec["broadcast"]
                    ceiling = r4["games"] / max(1e-9, sec["dedup"] + sec["train"])
                    loop_rec["projected_8_ranks"] = {
                        "seconds": {"selfplay": sec["selfplay"], "gather": round(g8, 4), "dedup": round(8 * sec["dedup"], 4),
                                    "train": round(8 * sec["train"], 4), "broadcast": sec["broadcast"], "total": round(tot8, 4)},
                        "games_per_s_of_the_whole_iteration": 8 * r4["games"] / tot8,
                        "speedup_over_one_rank": (8 * r4["games"] / tot8) / (r4["games"] / max(1e-9, sec["total"])),
                        "amdahl_ceiling_games_per_s": ceiling,
                        "model": "play x1 (weak scaling), gather = 72 B x positions over one xGMI link per peer + 8 x the host unpack, dedup x8, "
                                 "train x8 (one learner, alpha_zero.rs:45-100), broadcast x1; NOT measured on 8 GPUs",
                        "reading": "the loop is serial in the learner: past ~%.1f ranks the seven other GPUs wait for rank 0; the data-parallel "
                                   "learner (DataParallelLearner: one fused all-reduce per step) is the shape that scales the training half"
                                   % max(1.0, (sec["selfplay"] + sec["dedup"] + sec["train"]) / max(1e-9, sec["dedup"] + sec["train"]))}
        except Exception as ex:   # never lose the bench line over the learner leg
            loop_rec = {"error": repr(ex)}

    if rank == 0:
        total_games = gps * world * args.steps
        games_per_s = total_games / elapsed
        evals_per_game = c["policy_evals"] / gps
        explores_per_game = c["explores"] / gps
        tflops = c["policy_evals"] * FLOP_PER_EVAL / (last_ms * 1e-3) / 1e12
        gbs = algorithmic_bytes(c) / (last_ms * 1e-3) / 1e9
        mfma = {"bound": "mfma", "achieved": tflops, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": tflops / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                "kernel": kernel_name + " (fused select/expand + Connect4Net f32 MFMA + backprop)",
                "kernel_ms_avg": avg_ms, "flop_per_leaf_eval": FLOP_PER_EVAL}
        hbm = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
               "traffic": None, "algorithmic_bytes_per_explore": algorithmic_bytes(c) / max(1, c["explores"]),
               "kernel_ms_avg": avg_ms}
        traffic, src = measured_traffic(args)
        for rf in (mfma, hbm):
            rf["traffic"] = traffic            # HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, guide §HBM), PMC passes
            rf["traffic_source"] = src         # the committed rocprofv3 run of this configuration and these kernel sources
        hbm["algorithmic_bytes_per_launch"] = algorithmic_bytes(c)
        near, other = (mfma, hbm) if mfma["frac"] >= hbm["frac"] else (hbm, mfma)
        out = {
            "metric": "self-play games/sec, 9x7 Connect4", "value": games_per_s, "unit": "games/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"9x7 Connect4 self-play, 1 MI355X per rank, GPU-resident SoA MCTS node pool "
                                   f"(BASELINE configs[2]) + fused f32-MFMA Connect4Net leaf inference, {played_slots} "
                                   f"concurrent games per GPU, {args.explores} explores/move, parity MCTS config; "
                                   f"fixed-seed random-init weights",
                       # trees the launch plays on (grid x threads of the lane-per-tree kernel: one tree per lane) and the slots the
                       # engine was created with — equal by default; a larger engine only holds idle slabs
                       "games_per_step_per_gpu": gps, "concurrent_games_per_gpu": played_slots, "tree_slots_allocated": args.concurrent,
                       "explores_per_move": args.explores, "parallelism": f"games sharded over {world} GPU(s), no collective"},
            "leaf_evals_per_s": games_per_s * evals_per_game, "explores_per_s": games_per_s * explores_per_game,
            "plies_per_game": plies / total_games,
            "roofline": near, "roofline_other": other,
        }
        if loop_rec is not None:
            out["learner_loop"] = loop_rec
        # ---- extra legs, most important first. Every leg is one launch of a QUARTER of a step (gx games; HALF a step, gh, for the
        #      trained-network and policy-cache legs) on the headline concurrency, so that the whole list fits beside the driver's
        #      K = 20 / W = 5 run (20 full + 5 quarter launches + the counted re-run = ~670 s of an ~870 s budget); a leg is skipped —
        #      and named — only if it would still overrun --time-budget-s. A shorter launch carries a larger share of tail (tree
        #      slots emptying at its end: ~5 % at a full step, ~9 % at half, ~15 % at a quarter). The CPU baseline always runs.
        t_step = elapsed / args.steps
        t_q = t_step * gx / gps          # a quarter launch of the headline configuration
        skipped = []

        def fits(name, est_s):
            ok = (time.perf_counter() - t_run0) + est_s <= args.time_budget_s
            if not ok:
                skipped.append(name)
            return ok

        if world == 1 and not args.no_cpu_baseline:
            budget, quota = host_cpu_budget()
            # two worker threads per usable CPU (measured best on the 16-CPU-quota boxes: 16 -> 103, 32 -> 130 games/s)
            threads = args.cpu_threads or min(os.cpu_count() or 1, 2 * budget)
            sample = args.cpu_sample_games or 64 * threads  # ~15-20 s of wall time
            out["cpu_baseline"] = cpu_baseline(blob, args.explores, sample, threads)
            out["cpu_baseline"]["host"] = {"logical_cpus": os.cpu_count(), "cgroup_cpu_quota": quota, "usable_cpus": budget}
            try:
                out["cpu_baseline"]["single_thread"] = cpu_baseline_single_thread(blob, args.explores)   # BASELINE configs[0]
            except Exception as ex:  # noqa: BLE001
                out["cpu_baseline"]["single_thread"] = {"error": str(ex)[:200]}
        next_first = (args.warmup + args.steps) * gps   # game indices never reused by a later leg
        extras = world == 1 and not args.no_extras
        if extras and fits("with_f16x2_network", 5.5 * t_q + 12):
            try:
                out["with_f16x2_network"] = f16x2_leg(eng, gh, next_first)
            except Exception as ex:  # noqa: BLE001 (never lose the bench line over an extra leg)
                out["with_f16x2_network"] = {"error": str(ex)[:200]}
            next_first += 2 * (args.concurrent + gh)
        if extras and trained_blob is not None and fits("with_trained_weights", 3.6 * t_q + 8):
            out["with_trained_weights"] = trained_leg(eng, gh, next_first)
            out["with_trained_weights"]["random_init_for_comparison"] = {"select_levels_per_explore": c["select_levels"] / max(1, c["explores"]),
                                                                         "max_depth": c["max_depth"]}
            next_first += args.concurrent + gh
        if extras and fits("with_replay_outputs_to_host", 1.2 * t_q + 6):
            # a launch with the replay outputs (positions, visit distributions, value targets, actions: 4.3 KB per game) copied to
            # host memory inside the timed region — the PCIe-inclusive rate (never `value`); kernel_ms is the same launch without
            # the copies
            try:
                import psutil
                room = psutil.virtual_memory().available > 3 * gx * 63 * 69
            except Exception:
                room = False
            if room:
                t1 = time.perf_counter()
                ro = eng.selfplay(cfg, base_seed=0, n_games=gx, first_game=next_first, outputs=True)
                dt3 = time.perf_counter() - t1
                next_first += gx
                nbytes = sum(v.nbytes for v in ro.values() if isinstance(v, np.ndarray))
                out["with_replay_outputs_to_host"] = {"games_per_s": gx / dt3, "games": gx, "bytes_copied": int(nbytes), "seconds": dt3,
                                                      "kernel_ms": ro["kernel_ms"],
                                                      "games_per_s_of_the_kernel_alone": gx / (ro["kernel_ms"] * 1e-3)}
                del ro
            else:
                out["with_replay_outputs_to_host"] = {"skipped": "not enough host memory for the 4.3 KB per game of outputs"}
        if world == 1 and not args.no_policy_cache and fits("with_policy_cache+reference_selfplay_config", 5.0 * t_q + 40):
            # (a second engine beside the first: 2 x 60 GB of node pools + the 17 GB table fit the 288 GB of HBM)
            out.update(policy_cache_leg(gh, trained_blob if extras else None, reference_mcts=extras))
        if extras and fits("with_conv_policy", 3.0 * t_q + 8):
            out["with_conv_policy"] = conv_leg(eng, gh, next_first)
            next_first += args.concurrent + gh
        if extras and fits("launch_tail", 1.1 * t_q + 2):
            # the launch tail: a launch ends when its LAST game ends, so its final stretch runs on emptying tree slots. A launch of
            # a quarter of the games has the same tail on a quarter of the work: the difference between a full and a quarter launch
            # is tail-free time. tail_share = the fraction of a default step that the tail costs against a tail-free (infinitely
            # long) launch.
            t1 = time.perf_counter()
            eng.selfplay(cfg, base_seed=0, n_games=gx, first_game=next_first, outputs=False)
            dt2 = time.perf_counter() - t1
            next_first += gx
            steady = (gps - gx) / max(1e-9, t_step - dt2)    # games/s of the tail-free three quarters of a step
            out["launch_tail"] = {"games_per_s_at_quarter_games_per_step": gx / dt2, "steady_state_games_per_s": steady,
                                  "tail_share_of_a_step": max(0.0, 1.0 - games_per_s / steady)}
        if extras and fits("policy_eval_call", 3):
            # the boundary as a CALL — Policy::eval for a batch from pageable host buffers, host link included (never `value`): what a
            # Rust `impl Policy` (n = 1) or a host-tree worker (hundreds of leaves) pays per call; syn_eval_ctx_eval is the same path on
            # a context's own stream
            try:
                rs = np.random.RandomState(3)
                a = rs.randint(0, 2 ** 62, 4096, dtype=np.uint64)
                b = rs.randint(0, 2 ** 62, 4096, dtype=np.uint64)
                pm, po = a & ~b, b & ~a
                ctx = eng.eval_context()
                call = {}
                for n in (1, 256, 4096):
                    for name, fn in (("syn_policy_eval_batch", eng.policy_eval), ("syn_eval_ctx_eval", ctx.eval)):
                        fn(pm[:n], po[:n])
                        t1 = time.perf_counter()
                        for _ in range(200):
                            fn(pm[:n], po[:n])
                        call.setdefault(name, {})[str(n)] = round((time.perf_counter() - t1) / 200 * 1e6, 1)
                ctx.close()
                out["policy_eval_call"] = {"us_per_call_by_positions": call, "includes": "ctypes + numpy output allocation (~3 us), staging, launch, "
                                           "kernel, synchronisation; positions and results cross the host link in place (pinned, mapped memory)"}
            except Exception as ex:  # noqa: BLE001
                out["policy_eval_call"] = {"error": str(ex)[:200]}
        if world == 1 and not args.no_4096 and fits("at_4096_concurrent_games", 15):
            # BASELINE configs[1] names 4096 concurrent games: same engine code at 16 trees per CU (latency-optimised
            # kernel, weights in registers), one 16,384-game step, reported beside the headline configuration
            e2 = sa.Engine(concurrent_games=4096, max_explores=args.explores, device=local_rank)
            e2.load_weights(blob)
            e2.selfplay(cfg, base_seed=1, n_games=4096, outputs=False)
            t1 = time.perf_counter()
            r2 = e2.selfplay(cfg, base_seed=1, n_games=16384, first_game=4096, outputs=False)
            dt = time.perf_counter() - t1
            out["at_4096_concurrent_games"] = {"games_per_s": 16384 / dt, "kernel_ms": r2["kernel_ms"],
                                               "games": 16384, "plies_per_game": float(r2["plies"].mean()),
                                               "launch_shape": list(e2.last_launch_shape())}
            # the same 16,384 games in the f16x2 network arithmetic: four free-running waves of four trees per CU, every wave evaluating
            # its own leaves (free_kernel.cuh) instead of 16 trees in lock step behind one f32 tile split over four waves
            try:
                e2.set_network_arithmetic("f16x2")
                e2.selfplay(cfg, base_seed=1, n_games=4096, outputs=False)
                t1 = time.perf_counter()
                r2h = e2.selfplay(cfg, base_seed=1, n_games=16384, first_game=4096, outputs=False)
                dth = time.perf_counter() - t1
                out["at_4096_concurrent_games"]["with_f16x2_network"] = {
                    "games_per_s": 16384 / dth, "kernel_ms": r2h["kernel_ms"], "games": 16384, "plies_per_game": float(r2h["plies"].mean()),
                    "launch_shape": list(e2.last_launch_shape()), "dtype": "f16x2 (see with_f16x2_network)"}
                e2.set_network_arithmetic("f32")
            except Exception as ex:  # noqa: BLE001
                out["at_4096_concurrent_games"]["with_f16x2_network"] = {"error": str(ex)[:200]}
            # configs[1] as worded — host trees, batched Policy::eval launches (include/synthesis_amd_lockstep.hpp behind
            # syn_mcts_search_lockstep: one worker thread per usable CPU, each with two halves of its trees taking turns, the workers'
            # leaves combined into one launch on one evaluation context) — beside the fused search on the same 4,096 roots (mid-game
            # positions of the self-play run above): the form a caller with another Game impl would use
            try:
                r2s = e2.selfplay(cfg, base_seed=3, n_games=4096)
                ply = np.minimum(r2s["plies"] - 1, 8)
                roots = r2s["states_bb"][np.arange(4096), ply]
                e2.mcts_search(cfg.mcts_cfg, roots[:, 0], roots[:, 1], 16)
                t1 = time.perf_counter()
                fs = e2.mcts_search(cfg.mcts_cfg, roots[:, 0], roots[:, 1], args.explores)
                dt_f = time.perf_counter() - t1
                e2.mcts_search_lockstep(cfg.mcts_cfg, roots[:64, 0], roots[:64, 1], 16)
                t1 = time.perf_counter()
                ls = e2.mcts_search_lockstep(cfg.mcts_cfg, roots[:, 0], roots[:, 1], args.explores)
                dt_l = time.perf_counter() - t1
                st = ls["stats"]
                out["at_4096_concurrent_games"]["lockstep_host_trees"] = {
                    "roots": 4096, "explores": args.explores, "searches_per_s": 4096 / dt_l, "leaf_evals_per_s": st["positions_evaluated"] / dt_l,
                    "policy_eval_launches": st["rounds"], "seconds": dt_l, "seconds_in_policy_eval": st["seconds_policy"],
                    "host_threads": "what the process may use (hardware concurrency cut to the cgroup CPU quota), at most 32",
                    "identical_to_fused_search": bool(all(np.array_equal(ls[k], fs[k]) for k in ("child_N", "child_W", "best_action", "num_nodes"))),
                    "fused_search_on_the_same_roots": {"searches_per_s": 4096 / dt_f, "seconds": dt_f}}
                if fits("lockstep_selfplay", 40):
                    # ... and whole games in that form (syn_selfplay_run_lockstep = run_n_games over the host trees): the fused leg's
                    # shape — 16,384 games over 4,096 slots, a finished game's slot takes the next game — held to the fused kernel's
                    # games on the same seeds
                    t1 = time.perf_counter()
                    lg = e2.selfplay_lockstep(cfg, base_seed=3, n_games=16384)
                    dt_g = time.perf_counter() - t1
                    sg = lg["stats"]
                    r2s = e2.selfplay(cfg, base_seed=3, n_games=16384)
                    played = np.arange(63)[None, :] < r2s["plies"][:, None]   # (entries past a game's last ply are unspecified)
                    out["at_4096_concurrent_games"]["lockstep_host_trees"].update({
                        "games": 16384, "concurrent_games": 4096, "games_per_s": 16384 / dt_g, "selfplay_seconds": dt_g,
                        "selfplay_seconds_in_policy_eval": sg["seconds_policy"], "selfplay_policy_eval_launches": sg["rounds"],
                        "selfplay_leaf_evals_per_s": sg["positions_evaluated"] / dt_g,
                        "games_identical_to_fused_selfplay": bool(
                            np.array_equal(lg["plies"], r2s["plies"]) and np.array_equal(lg["final_kind"], r2s["final_kind"]) and
                            all(np.array_equal(lg[k][played], r2s[k][played]) for k in ("actions", "root_nodes", "states_bb", "pis", "vs")))})
            except Exception as ex:  # noqa: BLE001
                out["at_4096_concurrent_games"]["lockstep_host_trees"] = {"error": str(ex)[:200]}
            e2.close()
        if extras and fits("learner", 10):
            # the step after the path: optimiser steps per second of the learner at the reference's batch of 32, on
            # de-duplicated positions of a small self-play run — both networks through their persistent epoch kernels
            try:
                e3 = sa.Engine(concurrent_games=4096, max_explores=64, device=local_rank)
                e3.load_weights(blob)
                r3 = e3.selfplay(sa.parity_rollout_config(64), base_seed=1, n_games=8192)
                sel = np.arange(63)[None, :] < r3["plies"][:, None]
                d3 = e3.replay_deduplicate(r3["states_bb"][..., 0][sel], r3["states_bb"][..., 1][sel], r3["pis"][sel], r3["vs"][sel])
                nu = int(d3["num"].size)
                perm = np.random.default_rng(1).permutation(nu).astype(np.int32)
                st3 = min(nu // 32, 3000)
                learner = {"batch": 32, "unique_positions": nu}
                for name, init, w0, n_steps in (("connect4net", e3.trainer_init, blob, st3),
                                                ("connect4convnet", e3.trainer_init_conv, make_conv_weights(), st3),
                                                ("connect4convnet_bf16", e3.trainer_init_conv, make_conv_weights(), st3)):
                    init(w0)
                    if name.endswith("_bf16"):
                        e3.trainer_set_precision("bf16")   # BASELINE configs[4] "bf16 conv": bf16 matrix cores, f32 accumulation / Adam
                    e3.train_set_data(d3["my_bb"], d3["op_bb"], d3["pis"], d3["vs"])
                    e3.train_epoch(perm[: 64 * 32], 32, 1e-3)
                    t1 = time.perf_counter()
                    e3.train_epoch(perm[: n_steps * 32], 32, 1e-3)
                    dt6 = time.perf_counter() - t1
                    learner[name] = {"steps": n_steps, "steps_per_s": n_steps / dt6, "us_per_step": dt6 / n_steps * 1e6}
                # BASELINE configs[4]'s literal step — gradients -> RCCL all-reduce -> Adam — through DataParallelLearner on the process
                # group a one-GPU box can build (backend nccl, world size 1): the 122 KB message really goes through RCCL; steps/s
                # of the whole stream-ordered step incl. the device-side minibatch gather (the epoch kernels above have no collective)
                try:
                    import socket

                    from synthesis_amd.learner import DataParallelLearner

                    own = None
                    if dist is None and args.dist_backend == "nccl":
                        with socket.socket() as sk:
                            sk.bind(("127.0.0.1", 0))
                            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
                        os.environ.update(MASTER_ADDR="127.0.0.1", RANK="0", WORLD_SIZE="1", LOCAL_RANK=str(local_rank))
                        own = dist_util.init_process_group("nccl", local_rank)
                    dp = DataParallelLearner(e3, blob, dist=own if own is not None else dist, device=local_rank, collective_at_world_1=True)
                    dp.set_data(d3["my_bb"], d3["op_bb"], d3["pis"], d3["vs"])
                    dp.epoch(perm[: 64 * 32], 32, 1e-3)
                    torch.cuda.synchronize(local_rank)
                    t1 = time.perf_counter()
                    nst = dp.epoch(perm[: min(st3, 2000) * 32], 32, 1e-3)
                    torch.cuda.synchronize(local_rank)
                    dt7 = time.perf_counter() - t1
                    learner["data_parallel_world1"] = {"steps": nst, "steps_per_s": nst / dt7, "us_per_step": dt7 / nst * 1e6, "network": "connect4net",
                                                       "collective": "all_reduce of 30,494 f32 (gradients + 2 loss sums) per step, backend %s, world size 1"
                                                                     % (args.dist_backend if (own is not None or dist is not None) else "none"),
                                                       "path": "device gather of the minibatch -> syn_train_gradients_enqueue -> all_reduce -> "
                                                               "syn_train_apply_enqueue on one stream, no host synchronisation per step"}
                    if own is not None:
                        own.destroy_process_group()
                except Exception as ex:  # noqa: BLE001
                    learner["data_parallel_world1"] = {"error": repr(ex)[:300]}
                out["learner"] = learner
                e3.close()
            except Exception as ex:  # the learner is not the benchmarked path: never lose the bench line over it
                out["learner"] = {"error": repr(ex)}
        if skipped:
            out["skipped_for_time_budget"] = {"legs": skipped, "budget_s": args.time_budget_s}
        emit(out)

    try:
        eng.close()
    except Exception:
        pass
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
