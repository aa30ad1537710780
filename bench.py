#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X MettaGrid step path (BASELINE.json).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: `mg_step` over 4096 arena environments x 24 agents per GPU
(BASELINE.json configs[1]; 98 304 agent-steps per GPU per step) with seeded uniform-random actions that are already
resident in HBM when the timed region starts (a pre-generated [64, N] int32 action table; each step reads one row in
place through `mg_step_actions`).  Auto-reset is on, so episode turnover is inside the timed region.
Environments shard across ranks as independent instances (no data-path collective: "scaling": "weak"); the only
collective is the RCCL all-reduce of the finished-episode summary vector every 128 steps.

Prints ONE JSON line on rank 0.  `roofline`: algorithmic bytes per launch (1035 B per agent-step, SURVEY §8d:
610 B boundary + ~425 B state; one launch = one step = the fused kernel mg_step_fused*_kernel) / mean kernel duration
measured with HIP events on the launch stream over the timed steps (every 8th step is bracketed: the event pair is not
free); `traffic` is the PMC-measured HBM bytes per launch from profiles/ (see DESIGN.md), null when not collected.
`cpu_baseline`: the reference's own compiled core (oracle/_ref, kind "reference") or the oracle port, stepped on
the host cores over a bounded sample of the same workload, rank 0 / N=1 only.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

ENVS_PER_GPU = 4096
AGENTS = 24
ALGO_BYTES_PER_AGENT_STEP = 1035.0  # 610 B boundary + ~425 B state (arena)
OBS_BYTES_PER_AGENT_STEP = 970.0  # the observation kernel alone, 600 B written + ~370 B read
PROFILE_STRIDE = 8  # steps between HIP-event brackets in the timed region
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak


def _load_pool():
    z = np.load(ROOT / "tests" / "golden" / "arena24_map_pool.npz", allow_pickle=False)
    meta = json.loads(bytes(z["meta"]).decode())
    names = np.array(meta["cell_names"])
    return [names[m] for m in z["maps"]]


# ----------------------------------------------------------------------------------------------------------
# CPU baseline (runs BEFORE anything touches the GPU: worker processes are forked)
# ----------------------------------------------------------------------------------------------------------
def _cpu_worker(args):
    kind, worker, n_envs, n_steps, budget_s = args
    sys.path.insert(0, str(ROOT))
    from metta_amd.builder import make_arena
    from metta_amd.compile import compile_game

    cfg = make_arena(AGENTS)
    pool = _load_pool()
    rs = np.random.RandomState(1000 + worker)
    acts = rs.randint(0, 5, size=(n_steps, n_envs, AGENTS)).astype(np.int32)
    if kind == "reference":
        from oracle.ref_bridge import RefEnv

        envs = [RefEnv(cfg.game, pool[(worker * n_envs + i) % len(pool)], worker * n_envs + i) for i in range(n_envs)]
        t0 = time.perf_counter()
        done = 0
        for t in range(n_steps):
            for i, e in enumerate(envs):
                e.step(acts[t, i])
            done += 1
            if time.perf_counter() - t0 > budget_s:
                break
        dt = time.perf_counter() - t0
    else:
        from oracle.mg_oracle import make_oracle_env

        comp = compile_game(cfg, init_obs_passes=2)
        envs = [
            make_oracle_env(comp, comp.encode_map(pool[(worker * n_envs + i) % len(pool)]), worker * n_envs + i)
            for i in range(n_envs)
        ]
        t0 = time.perf_counter()
        done = 0
        for t in range(n_steps):
            for i, e in enumerate(envs):
                e.step(acts[t, i])
            done += 1
            if time.perf_counter() - t0 > budget_s:
                break
        dt = time.perf_counter() - t0
    return done * n_envs * AGENTS, dt


def cpu_baseline(budget_s: float = 10.0) -> dict:
    import multiprocessing as mp

    try:
        from oracle.ref_bridge import have_reference

        kind = "reference" if have_reference() else "port"
    except Exception:
        kind = "port"
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", "16")), 16))
    n_envs, n_steps = 8, 100000
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(kind, w, n_envs, n_steps, budget_s) for w in range(cores)])
    rate = sum(n / dt for n, dt in res)
    return {
        "value": rate,
        "unit": "agent-steps/s",
        "cores": cores,
        "kind": kind,
        "sample": f"{cores} independent processes x {n_envs} arena-24 envs stepped serially for ~{budget_s:.0f} s each, "
        "no vecenv IPC (no shared-memory buffers, no semaphores: an upper bound for the reference's multiprocess CPU vecenv); "
        f"same config/maps/action distribution; {'reference C++ core compiled from /root/reference' if kind == 'reference' else 'oracle port'}",
    }


# ----------------------------------------------------------------------------------------------------------
def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--envs", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--force-dist", action="store_true", help="run the RCCL summary all-reduce even with one rank (rehearsal)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_seconds)

    import torch
    import torch.distributed as dist

    from metta_amd.batch import MettaGridBatch
    from metta_amd.builder import make_arena
    from metta_amd.dist import bench_shard

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=dev)

    E, A = args.envs, AGENTS
    cfg = make_arena(A)
    pool = _load_pool()
    seeds, map_ids = bench_shard(rank, E, len(pool))  # env g: seed g (pufferlib vector.py:90), map g mod pool
    batch = MettaGridBatch(cfg, E, pool, device=local_rank, init_obs_passes=2, seeds=seeds, map_ids=map_ids)
    batch.set_auto_reset(True, 1)
    # desynchronise episode phases like the reference's EarlyResetHandler (early_reset_handler.py:6-20)
    batch.set_early_reset(np.random.RandomState(rank).randint(1, cfg.game.max_steps + 1, size=E).astype(np.uint32))

    N = E * A
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    table = torch.randint(0, 5, (64, N), generator=g, device=dev, dtype=torch.int32)
    summary = torch.zeros(3 + len(batch.compiled.game_slot_names), dtype=torch.float64, device=dev)

    def sync_all():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    acc = batch.summary_tensor()  # f64 accumulators in HBM, added to by the step kernels when an episode ends

    def reduce_summary():
        # the only collective of the path: < 1 KB, on the stream, no host round trip
        s = acc.clone()
        acc.zero_()
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        summary.add_(s)

    def run(k: int, base: int):
        for t in range(k):
            batch.step(table[(base + t) % 64])
            if use_dist and (base + t) % 128 == 127:
                reduce_summary()

    run(args.warmup, 0)
    if use_dist:
        reduce_summary()  # communicator set-up and first-use costs stay out of the timed region
    sync_all()
    # HIP events around every 8th step of the timed region: an event pair per step costs ~7 % of the step it times
    batch.set_profiling(True, stride=PROFILE_STRIDE)
    t0 = time.perf_counter()
    run(args.steps, args.warmup)
    sync_all()
    dt = time.perf_counter() - t0
    kernel_ms = batch.profile_read(4096)
    batch.set_profiling(False)
    step_path = batch.step_path()
    fused = "fused" in step_path
    obs_ms = np.zeros(0, np.float32)
    if not fused:
        # after the timed region: a short pass with events around the observation kernel alone (SURVEY 8d metric 2)
        batch.set_profiling(2)
        n_obs = min(args.steps, 100)
        run(n_obs, args.warmup + args.steps)
        sync_all()
        obs_ms = batch.profile_read(4096)
        batch.set_profiling(False)
    batch.check_errors()  # an environment that left reference parity would void the measurement

    t_max = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    dt = float(t_max.item())

    if rank == 0:
        agent_steps = float(world) * E * A * args.steps
        value = agent_steps / dt
        mean_kernel_s = float(kernel_ms.mean()) * 1e-3 if len(kernel_ms) else float("nan")
        algo_bytes = ALGO_BYTES_PER_AGENT_STEP * E * A
        achieved = algo_bytes / mean_kernel_s / 1e9
        traffic = None
        tfile = ROOT / "profiles" / "traffic.json"
        if tfile.exists():
            try:
                tj = json.loads(tfile.read_text())
                if tj.get("envs") == E:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "agent-steps/sec (node) arena 24-agent @1/2/4/8 GPU; obs-encode HBM GB/s",
            "value": value,
            "unit": "agent-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": f"{E} arena envs x {A} agents per GPU (BASELINE configs[1]), uniform-random actions, "
                "full 200-token observations, auto-reset on, 64 reference-generated 62x62 maps",
                "envs_per_gpu": E,
                "agents_per_env": A,
                "parallelism": f"env-sharded x{world} (independent instances)",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel": step_path + (" (one launch per step: actions, assembler uses, observations, rewards)" if fused else ""),
                "kernel_ms_mean": float(kernel_ms.mean()) if len(kernel_ms) else None,
                "kernel_ms_min": float(kernel_ms.min()) if len(kernel_ms) else None,
                "kernel_ms_samples": int(len(kernel_ms)),
                "algorithmic_bytes_per_launch": algo_bytes,
                "obs_kernel": None
                if fused
                else {
                    "ms_mean": float(obs_ms.mean()) if len(obs_ms) else None,
                    "algorithmic_bytes_per_launch": OBS_BYTES_PER_AGENT_STEP * E * A,
                    "achieved": OBS_BYTES_PER_AGENT_STEP * E * A / (float(obs_ms.mean()) * 1e-3) / 1e9 if len(obs_ms) else None,
                    "frac": OBS_BYTES_PER_AGENT_STEP * E * A / (float(obs_ms.mean()) * 1e-3) / 1e9 / HBM_PEAK_GBS
                    if len(obs_ms)
                    else None,
                },
            },
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    batch.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
