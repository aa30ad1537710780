#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X MettaGrid step path (BASELINE.json).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: every one of the 4096 arena environments x 24 agents of a GPU
(BASELINE.json configs[1]; 98 304 agent-steps per GPU per step) advances one tick, with seeded uniform-random actions
that are already resident in HBM when the timed region starts (a pre-generated [64, N] int32 action table; each step
reads one row in place through `mg_step_actions`).  Auto-reset is on, so episode turnover is inside the timed region.

The environments are stepped as `--groups` G independent groups (default 4: environments [g E/G, (g+1) E/G) of the rank),
each group a handle of its own on its own HIP stream: a group's step k+1 is ordered after its own step k and nothing
else.  This is the reference's double buffering (`async_factor`, RL/training/training_environment.py:45-46,151-170;
pufferlib `Multiprocessing` with batch_size < num_envs, vector.py:415-503; here `metta_amd.vecenv.AsyncDeviceVecEnv`):
the environments of one group are stepped while the policy works on another's observations.  One kernel alone ends
with a tail in which most CUs wait for its slowest workgroups; with G launches in flight the other groups' workgroups
fill it.  Environment ids, seeds, maps and actions are those of the single-handle case; `single_stream` in the output
is the same workload through ONE handle on one stream, measured right after the timed region (`--groups 1` makes that
the headline).
Environments shard across ranks as independent instances (no data-path collective: "scaling": "weak"); the only
collective is the RCCL all-reduce of the finished-episode summary vector every 128 steps.

Prints ONE JSON line on rank 0.  `roofline.achieved` = algorithmic bytes of one step of all the rank's environments
(1035 B per agent-step, SURVEY §8d: 610 B boundary + ~425 B state; x E x A) / the WALL-CLOCK time per step of the timed
region, so `frac` == 101.7 MB / ms_per_step / 8 TB/s.  Beside it, `per_launch`: the same bytes for one launch (one
group's step = the fused kernel mg_step_fused*_kernel over E/G environments) / the mean duration of that launch
measured with HIP events on the launch's own stream (every 8th step of every group is bracketed: the event pair is not
free) -- never multiplied by the number of launches in flight.  `obs_kernel` (BASELINE's second metric, SURVEY §8d
metric 2) is measured after the timed region on a handle of the same environments that runs the step as three
launches, with events around the observation kernel `mg_obs_kernel` alone: 970 algorithmic bytes per agent-step (600
written + ~370 read) / its mean duration.  `traffic` is the PMC-measured HBM bytes per launch from profiles/ (see
DESIGN.md), null when not collected for this launch shape.
`cpu_baseline`: the reference's own compiled core (oracle/_ref, kind "reference") or the oracle port inside a
multiprocess vecenv of pufferlib's design (oracle/cpu_vecenv.py: one worker per core, environments stepped serially per
worker, shared-memory buffers, busy-polled semaphores), on a bounded sample of the same workload, rank 0 / N=1 only;
`upper_bound_no_ipc` beside it is the same core in independent processes with no vecenv at all.
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
BOUNDARY_BYTES_PER_AGENT_STEP = 610.0  # 600 B observations + actions, rewards, flags
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


def _cpu_vecenv_leg(kind: str, cores: int, n_envs: int, budget_s: float):
    """The named baseline: all environments behind ONE synchronous vecenv (send actions -> every worker steps its
    environments -> recv), shared memory + semaphores (oracle/cpu_vecenv.py)."""
    from metta_amd.builder import make_arena
    from oracle.cpu_vecenv import CpuVecEnv

    pool = _load_pool()
    T = make_arena(AGENTS).game.obs.num_tokens
    vec = CpuVecEnv(kind, lambda: make_arena(AGENTS), pool, cores, n_envs, AGENTS, T, seed=0)
    try:
        rs = np.random.RandomState(999)
        acts = rs.randint(0, 5, size=(256, cores * n_envs * AGENTS)).astype(np.int32)
        for t in range(20):  # warm-up (first touches of the shared pages)
            vec.send(acts[t % 256])
            vec.recv()
        t0 = time.perf_counter()
        done = 0
        while time.perf_counter() - t0 < budget_s:
            vec.send(acts[done % 256])
            vec.recv()
            done += 1
        dt = time.perf_counter() - t0
    finally:
        vec.close()
    return done * cores * n_envs * AGENTS / dt, done


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
    # the driver busy-polls the semaphores like the workers do (vector.py:415-455): it gets a core of its own, or a
    # descheduled worker holds every step back (8 workers on 8 cores: half the rate of 7 + driver)
    workers = max(1, cores - 1)
    vec_rate, vec_steps = _cpu_vecenv_leg(kind, workers, n_envs, budget_s)
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(kind, w, n_envs, n_steps, budget_s * 0.6) for w in range(cores)])
    rate = sum(n / dt for n, dt in res)
    core = "reference C++ core compiled from /root/reference" if kind == "reference" else "oracle port"
    return {
        "value": vec_rate,
        "unit": "agent-steps/s",
        "cores": cores,
        "kind": kind,
        "sample": f"multiprocess vecenv of pufferlib's design (vector.py:185-511 restated in oracle/cpu_vecenv.py): {workers} worker "
        f"processes (+ the busy-polling driver = {cores} cores) x {n_envs} arena-24 envs stepped serially per worker, zero-copy shared-memory buffers, busy-polled semaphores, "
        f"synchronous send/recv over all {workers * n_envs} envs for ~{budget_s:.0f} s ({vec_steps} vecenv steps); "
        f"same config/maps/action distribution; {core}",
        "upper_bound_no_ipc": {
            "value": rate,
            "sample": f"{cores} independent processes x {n_envs} envs stepped serially for ~{budget_s * 0.6:.0f} s each, no vecenv at all",
        },
    }


# ----------------------------------------------------------------------------------------------------------
def rollout_workload(args, wl, rank, local_rank, world, use_dist, saved_stdout, pool, E, G) -> None:
    """`--workload rollout` (BASELINE configs[4] on the same shards): recv -> LSTM policy forward (bf16) -> sampling ->
    send, device-resident (metta_amd/rollout.py; reference `CoreTrainingLoop.rollout_phase`, metta/rl/training/core.py:
    67-189), G asynchronous groups of environments per rank as the reference's async_factor; the finished-episode summary
    is all-reduced every 128 steps like the other workloads.  A step = every environment of the rank stepped once."""
    import torch
    import torch.distributed as dist

    from metta_amd.rollout import LSTMPolicy
    from metta_amd.vecenv import AsyncDeviceVecEnv, MettaGridDeviceVecEnv

    dev = torch.device("cuda", local_rank)
    A, cfg = wl["agents"], wl["cfg"]
    seed0 = rank * E  # global environment ids: rank r owns [r*E, (r+1)*E), environment g has seed g (bench_shard)
    if G > 1:
        env = AsyncDeviceVecEnv(cfg, E, E // G, device=local_rank, seed=seed0, maps=pool, early_reset=True)
        handles = [g.batch for g in env.groups]
    else:
        env = MettaGridDeviceVecEnv(cfg, E, maps=pool, device=local_rank, seed=seed0, early_reset=True)
        handles = [env.batch]
    env.async_reset(seed0)  # (stats_every = 0: finished episodes are collected without blocking a send)
    n_group = env.agents_per_batch if G > 1 else env.num_agents
    pol = LSTMPolicy(cfg.game.obs.num_tokens, env.single_action_space.n).to(dev).to(torch.bfloat16)
    states = [pol.initial_state(n_group, dev, torch.bfloat16) for _ in range(G)]
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    summary = torch.zeros(3 + len(handles[0].compiled.game_slot_names), dtype=torch.float64, device=dev)
    accs = [h.summary_tensor() for h in handles]

    def reduce_summary():
        s = torch.stack(accs).sum(0)
        for a in accs:
            a.zero_()
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        summary.add_(s)

    @torch.no_grad()
    def run(k: int, base: int):
        for t in range(k):
            for q in range(G):  # one step of the rank = every group once
                o, r, d, tr, ta, infos, ids, mask = env.recv()
                logits, value, st = pol(o, states[q])
                done = (d | tr).to(st[0].dtype).unsqueeze(1)
                states[q] = (st[0] * (1 - done), st[1] * (1 - done))
                acts = torch.multinomial(torch.softmax(logits.float(), dim=-1), 1, generator=gen).squeeze(1).to(torch.int32)
                env.send(acts)
            if use_dist and (base + t) % 128 == 127:
                reduce_summary()

    def sync_all():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    run(args.warmup, 0)
    env.flush_infos()
    if use_dist:
        reduce_summary()
    sync_all()
    if saved_stdout is not None:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    sync_all()
    t0 = time.perf_counter()
    run(args.steps, args.warmup)
    sync_all()
    dt = time.perf_counter() - t0
    for h in handles:
        h.check_errors()
    step_path = handles[0].step_path()
    env.close()
    t_max = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    dt = float(t_max.item())
    if rank == 0:
        step_s = dt / args.steps
        achieved = ALGO_BYTES_PER_AGENT_STEP * E * A / step_s / 1e9
        print(json.dumps({
            "metric": "agent-steps/sec (node), device-resident rollout with an LSTM policy @1/2/4/8 GPU (BASELINE configs[4]; not the headline metric)",
            "value": float(world) * E * A * args.steps / dt, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 environment step, bf16 policy", "data": "synthetic",
            "config": {"workload": f"{E} " + wl["describe"] + ", random-init policy, actions sampled on the device, auto-reset on",
                       "envs_per_gpu": E, "agents_per_env": A, "groups": G, "parallelism": f"env-sharded x{world} (independent instances)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "kernel": step_path,
                         "basis": "wall clock of the whole rollout step against the ENVIRONMENT path's algorithmic bytes: the policy's "
                         "GEMMs and the u8 -> bf16 observation cast bound this workload (profiles/r03_rollout_kernel_stats.csv), "
                         "so this is the env path's share, not a kernel roofline"},
            "cpu_baseline": None,
        }), flush=True)  # fmt: skip
    if use_dist:
        dist.destroy_process_group()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", choices=("arena", "combat", "rollout"), default="arena",
                    help="arena = BASELINE configs[1] (the headline); combat = configs[3], rollout = configs[4]: the same sharding "
                    "(bench_shard) and summary all-reduce, for the multi-GPU curves of those configurations")
    ap.add_argument("--envs", type=int, default=None, help="environments per GPU (default: the workload's)")
    ap.add_argument("--groups", type=int, default=None, help="independent environment groups, one HIP stream each (1 = one handle)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="skip the one-handle comparison run (profiling passes: one launch shape only)")
    ap.add_argument("--no-obs-pass", action="store_true", help="skip the observation-kernel pass (metric 2) after the timed region")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--force-dist", action="store_true", help="run the RCCL summary all-reduce even with one rank (rehearsal)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    arena = args.workload == "arena"
    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline and arena:
        cpu = cpu_baseline(args.cpu_seconds)

    import torch
    import torch.distributed as dist

    from metta_amd.batch import MettaGridBatch
    from metta_amd.dist import bench_shard, bench_workload

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    saved_stdout = None
    if use_dist:
        # RCCL prints a version banner on stdout when its communicator comes up: stdout carries the one JSON line only,
        # so file descriptor 1 points at stderr until the warm-up's first all-reduce is through
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=dev)

    wl = bench_workload(args.workload)
    E, A = args.envs or wl["envs"], wl["agents"]
    groups = args.groups or wl["groups"]
    G = groups if groups >= 1 and E % max(groups, 1) == 0 else 1
    n = E // G
    cfg = wl["cfg"]
    pool = wl["maps"] or _load_pool()
    seeds, map_ids = bench_shard(rank, E, len(pool))  # env g: seed g (pufferlib vector.py:90), map g mod pool
    # desynchronise episode phases like the reference's EarlyResetHandler (early_reset_handler.py:6-20)
    early = np.random.RandomState(rank).randint(1, max(int(cfg.game.max_steps), 1) + 1, size=E).astype(np.uint32)
    N = E * A
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    table = torch.randint(0, wl["num_actions"], (64, N), generator=g, device=dev, dtype=torch.int32)
    if args.workload == "rollout":
        return rollout_workload(args, wl, rank, local_rank, world, use_dist, saved_stdout, pool, E, G)

    def make_batches(groups: int):
        m = E // groups
        out = []
        for q in range(groups):
            b = MettaGridBatch(cfg, m, pool, device=local_rank, init_obs_passes=2, seeds=seeds[q * m : (q + 1) * m],
                               map_ids=map_ids[q * m : (q + 1) * m])  # fmt: skip
            b.set_auto_reset(True, 1)
            b.set_early_reset(early[q * m : (q + 1) * m])
            out.append(b)
        return out

    batches = make_batches(G)
    tables = [table[:, q * n * A : (q + 1) * n * A].contiguous() for q in range(G)] if G > 1 else [table]
    streams = [torch.cuda.Stream(device=dev) for _ in range(G)] if G > 1 else [torch.cuda.current_stream(dev)]
    summary = torch.zeros(3 + len(batches[0].compiled.game_slot_names), dtype=torch.float64, device=dev)

    def sync_all():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    accs = [b.summary_tensor() for b in batches]  # f64 accumulators in HBM, added to by the step kernels when an episode ends

    # The only collective of the path: < 1 KB, no host round trip, and no group waits for it: each group
    # hands its accumulator over on its own stream (copy + clear, in order with its steps), a side stream sums the
    # copies and runs the all-reduce.  (Joining the group streams on one stream instead drains the pipeline of launches
    # every time: 0.0509 against 0.0489 ms/step with a reduction every 128 steps.)
    side = torch.cuda.Stream(device=dev)
    stage = torch.zeros((len(accs), accs[0].numel()), dtype=accs[0].dtype, device=dev)
    stage_free = torch.cuda.Event()  # the previous reduction has read `stage`
    stage_free.record(side)

    def reduce_summary():
        handed = []
        for q, st in enumerate(streams):
            with torch.cuda.stream(st):
                st.wait_event(stage_free)
                stage[q].copy_(accs[q])
                accs[q].zero_()
                ev = torch.cuda.Event()
                ev.record(st)
                handed.append(ev)
        with torch.cuda.stream(side):
            for ev in handed:
                side.wait_event(ev)
            s = stage.sum(0)
            stage_free.record(side)
            dist.all_reduce(s, op=dist.ReduceOp.SUM)
            summary.add_(s)

    def make_stepper(bs, tbs, sts):
        """Everything the launch loop needs, built ONCE: the tensors are validated here, the loop hands the library raw
        pointers and stream handles (one library call per step)."""
        assert all(tb.dtype == torch.int32 and tb.is_contiguous() and tb.shape[1] == b.actions.numel() for b, tb in zip(bs, tbs))
        ptrs = [[tb[r].data_ptr() for r in range(64)] for tb in tbs]
        return MettaGridBatch.group_stepper(bs, ptrs, [st.cuda_stream for st in sts])

    def run(step, k: int, base: int, reduce: bool = True):
        for t in range(k):
            step((base + t) % 64)
            if reduce and use_dist and (base + t) % 128 == 127:
                reduce_summary()

    stepper = make_stepper(batches, tables, streams)
    run(stepper, args.warmup, 0)
    if use_dist:
        reduce_summary()  # communicator set-up and first-use costs stay out of the timed region
    sync_all()
    if saved_stdout is not None:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    # HIP events around every 8th step of the timed region: an event pair per step costs ~7 % of the step it times
    for b in batches:
        b.set_profiling(True, stride=PROFILE_STRIDE)
    sync_all()
    t0 = time.perf_counter()
    run(stepper, args.steps, args.warmup)
    sync_all()
    dt = time.perf_counter() - t0
    kernel_ms = np.concatenate([b.profile_read(4096) for b in batches])
    for b in batches:
        b.set_profiling(False)
    step_path = batches[0].step_path()
    fused = "fused" in step_path
    for b in batches:
        b.check_errors()  # an environment that left reference parity would void the measurement
    for b in batches:
        b.close()

    single = None
    if G > 1 and rank == 0 and not use_dist and not args.no_single and arena:
        # the same workload through one handle on one stream (what --groups 1 reports as the headline)
        one = make_batches(1)
        cur = [torch.cuda.current_stream(dev)]
        k1 = min(max(args.steps, 100), 200)
        step1 = make_stepper(one, [table], cur)
        run(step1, min(args.warmup, 30), 0, reduce=False)
        torch.cuda.synchronize(dev)
        one[0].set_profiling(True, stride=PROFILE_STRIDE)
        t1 = time.perf_counter()
        run(step1, k1, args.warmup, reduce=False)
        torch.cuda.synchronize(dev)
        d1 = time.perf_counter() - t1
        km = one[0].profile_read(4096)
        one[0].check_errors()
        p1 = one[0].step_path()
        one[0].close()
        single = {
            "steps": k1,
            "ms_per_step": d1 / k1 * 1e3,
            "value": E * A * k1 / d1,
            "kernel": p1,
            "kernel_ms_mean": float(km.mean()) if len(km) else None,
            "achieved": ALGO_BYTES_PER_AGENT_STEP * E * A / (d1 / k1) / 1e9,
            "frac": ALGO_BYTES_PER_AGENT_STEP * E * A / (d1 / k1) / 1e9 / HBM_PEAK_GBS,
        }

    obs_kernel = None
    if rank == 0 and not args.no_obs_pass and arena:
        # BASELINE's second metric (SURVEY 8d metric 2): the observation kernel alone.  The fused step has no kernel
        # boundary around its encode, so the same environments are stepped by a handle that runs the step as three
        # launches (MG_NO_FUSED is read when the pool is set), with HIP events around mg_obs_kernel only.
        os.environ["MG_NO_FUSED"] = "1"
        try:
            three = make_batches(1)
        finally:
            del os.environ["MG_NO_FUSED"]
        cur = [torch.cuda.current_stream(dev)]
        step3 = make_stepper(three, [table], cur)
        run(step3, 20, 0, reduce=False)
        torch.cuda.synchronize(dev)
        three[0].set_profiling(2)
        n_obs = 100
        run(step3, n_obs, 20, reduce=False)
        torch.cuda.synchronize(dev)
        obs_ms = three[0].profile_read(4096)
        three[0].set_profiling(False)
        three[0].check_errors()
        p3 = three[0].step_path()
        three[0].close()
        if len(obs_ms):
            obs_bytes = OBS_BYTES_PER_AGENT_STEP * E * A
            obs_s = float(obs_ms.mean()) * 1e-3
            obs_kernel = {
                "kernel": "mg_obs_kernel (the observation kernel of the three-launch step: " + p3 + ")",
                "ms_mean": float(obs_ms.mean()),
                "ms_min": float(obs_ms.min()),
                "samples": int(len(obs_ms)),
                "envs_per_launch": E,
                "algorithmic_bytes_per_launch": obs_bytes,
                "achieved": obs_bytes / obs_s / 1e9,
                "unit": "GB/s",
                "frac": obs_bytes / obs_s / 1e9 / HBM_PEAK_GBS,
            }

    t_max = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    dt = float(t_max.item())

    if rank == 0:
        agent_steps = float(world) * E * A * args.steps
        value = agent_steps / dt
        step_s = dt / args.steps
        # (combat: the 610 B boundary only -- its grid and agent records live in HBM and their traffic is not in the
        # figure, so `frac` is a lower bound there)
        algo_agent = ALGO_BYTES_PER_AGENT_STEP if arena else BOUNDARY_BYTES_PER_AGENT_STEP
        algo_step = algo_agent * E * A  # one step of all the rank's environments
        achieved = algo_step / step_s / 1e9  # from the wall clock of the timed region
        mean_kernel_s = float(kernel_ms.mean()) * 1e-3 if len(kernel_ms) else float("nan")
        algo_bytes = algo_agent * n * A  # one launch = one group's step
        per_launch = algo_bytes / mean_kernel_s / 1e9
        traffic = None
        tfile = ROOT / "profiles" / "traffic.json"
        if tfile.exists():
            try:
                tj = json.loads(tfile.read_text())
                one = tj.get("round3_single_stream_4096_envs_per_launch", {})
                if arena and tj.get("envs_per_launch", tj.get("envs")) == n:
                    traffic = tj.get("hbm_bytes_per_launch")
                elif arena and G == 1 and n == 4096 and one.get("kernel", "").split(" ")[0] == step_path:
                    traffic = one.get("hbm_bytes_per_launch")  # the one-handle form was counted in passes of its own
            except Exception:
                traffic = None
        out = {
            "metric": "agent-steps/sec (node) arena 24-agent @1/2/4/8 GPU; obs-encode HBM GB/s" if arena
            else "agent-steps/sec (node), combat stress 256x256x1024 @1/2/4/8 GPU (BASELINE configs[3]; not the headline metric)",
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
                "workload": f"{E} " + wl["describe"] + ", uniform-random actions, full 200-token observations, auto-reset on, "
                + ("64 reference-generated 62x62 maps" if arena else f"{len(pool)} maps"),
                "envs_per_gpu": E,
                "agents_per_env": A,
                "groups": G,
                "stepping": (f"{G} independent groups of {n} environments, one handle and one HIP stream each (the reference's "
                             "async_factor double buffering): a group's step k+1 follows its own step k only")
                if G > 1 else "one handle, one stream",
                "parallelism": f"env-sharded x{world} (independent instances)",
            },
            "roofline": {
                "bound": "hbm",
                "algorithmic_bytes_per_agent_step": algo_agent,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "basis": "wall clock: algorithmic bytes of one step of all environments / (timed region / steps)",
                "algorithmic_bytes_per_step": algo_step,
                "kernel": step_path + (" (one launch per group and step: actions, assembler uses, observations, rewards)" if fused else ""),
                "per_launch": {
                    "envs_per_launch": n,
                    "launches_in_flight": G,
                    "algorithmic_bytes_per_launch": algo_bytes,
                    "kernel_ms_mean": float(kernel_ms.mean()) if len(kernel_ms) else None,
                    "kernel_ms_min": float(kernel_ms.min()) if len(kernel_ms) else None,
                    "kernel_ms_samples": int(len(kernel_ms)),
                    "achieved": per_launch,
                    "frac": per_launch / HBM_PEAK_GBS,
                },
                "obs_kernel": obs_kernel,
            },
            "single_stream": single,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
