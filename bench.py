#!/usr/bin/env python
"""bench.py — physics steps/s of the MI355X rigid-body stepper on the BASELINE.json headline workload.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one pass of the hot path (physicsStepInternal: world colliders -> broad phase -> narrow phase
-> integrate forces -> schedule -> constraint init -> I PGS sweeps -> integrate velocities) over the
whole synthetic scene.  Workload at N=1: cfg3, the 262 144-body OBB pile (128 x 16 x 128 boxes, half-extents
U[0.3,0.6], friction 0.5, 20 solver iterations, walled pen) — the configuration BASELINE.json's metric is
quoted on; it fits one GPU.  N > 1 is weak scaling: every rank steps one 262 144-body tile (see DESIGN.md
"multi-GPU").  All scene state is resident in HBM before the timed region starts.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel = the
contact PGS solver: k_contact_solve_persist (k_contact_solve_flow with MI_SOLVER=flow), one launch for all sweeps of a step; algorithmic bytes from
SURVEY.md §8(d)) and `cpu_baseline` (CPU oracle, reference order, bounded sample) objects.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
BYTES_PER_CONTACT_ITER = 236    # per contact per PGS sweep (124 B row + 2 x 28 B body read, 8 + 2 x 24 B written)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=240)
    ap.add_argument("--grid", type=int, nargs=3, default=[128, 16, 128], help="boxes per axis of one GPU's tile (default = 262144 bodies)")
    ap.add_argument("--iterations", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-grid", type=int, nargs=3, default=[32, 16, 32])
    ap.add_argument("--cpu-warmup", type=int, default=200)
    ap.add_argument("--cpu-steps", type=int, default=20)
    ap.add_argument("--cpu-cores", type=int, default=0, help="oracle replicas for the CPU baseline (0 = all host cores)")
    return ap.parse_args()


def _cpu_replica(job):
    """One oracle replica: settle, then time `steps` steps of a (nx, ny, nz) tile of the workload."""
    nx, ny, nz, iterations, warmup, steps = job
    import oracle
    from d3d12renderer_amd import scenes
    sc = scenes.obb_pile(nx, ny, nz, solver_iterations=iterations)
    w = sc.populate(oracle.create_world(oracle.ORDER_REFERENCE))
    s = sc.settings()
    w.step_fixed(s, sc.dt, warmup)
    t0 = time.perf_counter()
    w.step_fixed(s, sc.dt, steps)
    return steps / (time.perf_counter() - t0), sc.num_bodies, w.counts()["num_contacts"]


def _usable_cores():
    """Host cores this process may actually use: affinity mask and cgroup CPU quota, not the machine's logical CPU count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(args, bodies_full):
    """CPU oracle (reference order = the reference's scalar path restated) on a bounded sample of the same workload: same
    generator and column height, smaller footprint.  The reference step is single-threaded, so "the host cores of the box"
    are used the only way that code can use them: one independent replica per core, each its own
    process; the value is the aggregate body-steps/s of all replicas expressed in steps/s of the full 262144-body scene."""
    import subprocess
    import oracle
    oracle.build()
    nx, ny, nz = args.cpu_grid
    cores = max(1, min(args.cpu_cores or _usable_cores(), 64))
    job = [nx, ny, nz, args.iterations, args.cpu_warmup, args.cpu_steps]
    cmd = [sys.executable, str(Path(__file__).resolve()), "--cpu-replica", json.dumps(job)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env) for _ in range(cores)]
    res = []
    for p in procs:
        out, _ = p.communicate(timeout=900)
        if p.returncode == 0 and out.strip():
            res.append(json.loads(out.strip().splitlines()[-1]))
    if not res:
        res = [list(_cpu_replica(job))]
    nb = res[0][1]
    agg = sum(r[0] for r in res)                      # replica-steps/s, all replicas together (they ran concurrently)
    return {
        "value": agg * nb / bodies_full, "unit": "steps/s", "cores": len(res), "kind": "port",
        "sample": (f"oracle (reference order, -O2 strict fp32) on obb_pile {nx}x{ny}x{nz} = {nb} bodies, I={args.iterations}, "
                   f"{args.cpu_warmup} settle + {args.cpu_steps} timed steps, one independent single-threaded replica per core, "
                   f"{len(res)} concurrent replicas: {agg:.1f} replica-steps/s in aggregate ({agg / len(res):.2f} per replica, "
                   f"{res[0][2]} contacts each); value = aggregate x {nb}/{bodies_full} (linear in bodies)"),
        "per_core_value": agg / len(res) * nb / bodies_full, "measured_replica_steps_per_s": agg, "sample_bodies": nb,
    }


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-replica":     # worker process of cpu_baseline()
        print(json.dumps(list(_cpu_replica(json.loads(sys.argv[2])))))
        return
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    torch.cuda.set_device(local_rank)   # torch's HIP runtime comes up before the library touches the device
    if world_size > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device("cuda", local_rank))
    if world_size != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_size}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)

    import d3d12renderer_amd as mi
    from d3d12renderer_amd import scenes
    from d3d12renderer_amd.distributed import ShardedWorld

    nx, ny, nz = args.grid
    sw = ShardedWorld(lambda: mi.create_world(local_rank), rank, world_size, dist, tile=(nx, ny, nz), iterations=args.iterations)
    settings = sw.settings()
    dt = sw.dt

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        sw.step(settings, dt)
    barrier()
    t0 = time.perf_counter()
    step_ms = []
    sw.world.accumulated_stage_times(reset=True)      # the library sums the per-stage device times of the timed steps itself
    for _ in range(args.steps):
        t_step = time.perf_counter()
        sw.step(settings, dt)
        step_ms.append((time.perf_counter() - t_step) * 1e3)
    barrier()
    elapsed = time.perf_counter() - t0
    stage_acc, n_acc, contact_iters = sw.world.accumulated_stage_times()
    assert n_acc == args.steps
    solve_ms = stage_acc["solve"]; total_dev_ms = stage_acc["total"]
    launches = sw.world.solve_launches() * args.steps
    # roofline of the dominant kernel: a few extra steps (outside the timed region) with a HIP event pair around every
    # k_contact_solve launch, on the stream the kernel is launched on
    prof_launches = 0; prof_ms = 0.0; prof_updates = 0
    sw.world.set_stage_timing(True)     # the per-stage breakdown comes from these extra steps too (an event pair per stage costs device time)
    stage_prof = {}
    for _ in range(3):
        n_l, ms, upd = sw.world.step_profiled(settings, dt)
        for k, v in sw.world.stage_times().items():
            stage_prof[k] = stage_prof.get(k, 0.0) + v / 3.0
        sw.exchange_ghosts()
        prof_launches += n_l; prof_ms += ms; prof_updates += upd
    sw.world.set_stage_timing(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    counts = sw.world.counts()
    bodies_per_gpu = sw.bodies_per_rank
    total_bodies = bodies_per_gpu * world_size

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        # whole-job throughput: every rank steps its 262144-body tile each step; the job advances one scene step per `ms_per_step`
        # and processes world_size tiles, so value = tiles-steps per second (at N=1: plain steps/s of the 262144-body scene).
        value = world_size * args.steps / elapsed
        # Dominant kernel = the contact PGS solver.  Default path: k_contact_solve_persist, ONE launch per step covering all
        # sweeps (MI_SOLVER=flow: k_contact_solve_flow, also one launch; MI_SOLVER=launch: one k_contact_solve launch per colour per sweep).  achieved = algorithmic bytes of all
        # contact updates (236 B per contact per sweep) / HIP-event time of the solve stage, recorded on
        # the world's own stream around exactly those launches, averaged over the timed steps.
        launches_per_step = launches / args.steps
        kernel = sw.world.solver_kernel()   # k_contact_solve_persist by default (k_contact_solve_flow / k_contact_solve with MI_SOLVER=flow / launch)
        achieved = (BYTES_PER_CONTACT_ITER * contact_iters) / (solve_ms * 1e-3) / 1e9 if solve_ms > 0 else 0.0
        event_pair = (BYTES_PER_CONTACT_ITER * prof_updates) / (prof_ms * 1e-3) / 1e9 if prof_ms > 0 else 0.0
        traffic = _measured_traffic(kernel)
        avg_launch_s = solve_ms * 1e-3 / max(launches, 1)
        roofline = {
            "bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
            # what really crossed the HBM interface per second (committed PMC pass / this run's launch time): the persistent kernel
            # serves a good part of the algorithmic bytes from LDS (accumulated impulses) and the XCDs' L2s (body hand-overs)
            "traffic_GBps": (traffic / avg_launch_s / 1e9) if traffic and avg_launch_s > 0 else None,
            "avg_launch_us": solve_ms * 1e3 / max(launches, 1), "launches_per_step": launches_per_step,
            "algorithmic_bytes_per_launch": BYTES_PER_CONTACT_ITER * contact_iters / max(launches, 1),
            "event_pair_per_launch": {"avg_launch_us": prof_ms * 1e3 / max(prof_launches, 1), "launches_per_step": prof_launches / 3,
                                      "achieved_GBps": event_pair,
                                      "note": "3 extra steps (outside the timed region) with a dedicated HIP event pair around each solver launch"},
            "note": ("rank 0, timed region: 236 B x contacts x sweeps / HIP-event time of the solve stage on the world's stream; "
                     "rocprofv3 --kernel-trace --stats of the same command: profiles/; traffic = HBM bytes per launch from separate "
                     "--pmc FETCH_SIZE / WRITE_SIZE passes (profiles/traffic.json)"),
        }
        out = {
            "metric": "physics steps/sec at 262144 rigid bodies per GPU (OBB pile, 20 solver iterations)",
            "value": value, "unit": "steps/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg3 obb_pile {nx}x{ny}x{nz} boxes per GPU ({bodies_per_gpu} bodies/GPU, {total_bodies} total), "
                                   f"friction 0.5, {args.iterations} solver iterations, dt=1/120",
                       "bodies_per_gpu": bodies_per_gpu, "contacts": counts["num_contacts"], "manifolds": counts["num_collisions"],
                       "broadphase_overlaps": counts["num_broadphase_overlaps"], "colors": counts["num_colors"],
                       "sharding": sw.sharding_note},
            "roofline": roofline,
            "stage_ms": stage_prof, "stage_ms_note": "per-stage device times of the 3 extra steps after the timed region (stage timing enabled only there)",
            "step_ms_median": float(np.median(step_ms)), "step_ms_p95": float(np.percentile(step_ms, 95)),
            "device_ms_per_step": total_dev_ms / args.steps,
        }
        if not args.no_cpu_baseline and world_size == 1:   # rank 0 at N = 1 only (the other ranks' hosts would wait in the next collective)
            out["cpu_baseline"] = cpu_baseline(args, bodies_per_gpu)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _measured_traffic(kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/traffic.json), or None."""
    p = ROOT / "profiles" / "traffic.json"
    if p.exists():
        try:
            return json.loads(p.read_text()).get(kernel + "_bytes_per_launch")
        except Exception:
            return None
    return None


if __name__ == "__main__":
    main()
