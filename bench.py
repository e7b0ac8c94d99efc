#!/usr/bin/env python
"""bench.py — physics steps/s of the MI355X rigid-body stepper on the BASELINE.json headline workload.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one pass of the hot path (physicsStepInternal: world colliders -> broad phase -> narrow phase
-> integrate forces -> schedule -> constraint init -> I PGS sweeps -> integrate velocities) over the
whole synthetic scene.  Workload at N=1: cfg3, the 262 144-body OBB pile (128 x 16 x 128 boxes, half-extents
U[0.3,0.6], friction 0.5, 20 solver iterations, walled pen) — the configuration BASELINE.json's metric is
quoted on; it fits one GPU.  N > 1 (one process per GPU, include/mi_shard.h): every rank holds the whole scene and simulates one
x tile of it (ownership by position, ghost strip, neighbour exchange over RCCL inside the library).  `--scaling weak` (default): the
pen grows with N (128 N x 16 x 128 boxes: 262 144 bodies per GPU); `--scaling strong`: the one 262 144-body pile is cut into N tiles.

THE TIMED STATE IS PART OF THE WORKLOAD, NOT OF THE FLAGS.  The lattice is first stepped SETTLE_STEPS = 240 times
(untimed, always; BASELINE.md §2 "settle") so that the boxes are piled up (~3.3 contacts per body); only then come the
caller's W warm-up steps and the K timed steps.  The CPU baseline settles its sample with the same 240 steps.  `--warmup`
therefore no longer changes what is being measured; the line carries `settle_steps`, contacts and contacts per body, and
the run refuses to time a scene with fewer than 2 contacts per body.  A second figure, `at_rest`, times the same pile after
1 500 steps in total (everything has come to rest: ~4.7 contacts per body).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel = the contact PGS solver,
k_contact_solve_persist, one launch for all sweeps of a step; algorithmic bytes from SURVEY.md §8(d)) and `cpu_baseline`
(the REFERENCE's own code, oracle/_ref/libref_fast.so: scalar path and AVX2 path, on the host cores) objects.
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
HBM_ACHIEVABLE_GBPS = 6300.0   # what a pure streaming kernel reaches on this part (same guide): the ceiling a real kernel is measured against
BYTES_PER_CONTACT_ITER = 236    # per contact per PGS sweep (124 B row + 2 x 28 B body read, 8 + 2 x 24 B written)
SETTLE_STEPS = 240              # part of the workload definition (see module docstring)
AT_REST_STEPS = 1500            # total steps before the `at_rest` measurement
MIN_CONTACTS_PER_BODY = 2.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240, help="timed steps (BASELINE.md §2: 240)")
    ap.add_argument("--warmup", type=int, default=10, help="untimed steps right before the timed region (the pile is settled separately, always)")
    ap.add_argument("--grid", type=int, nargs=3, default=[128, 16, 128], help="boxes per axis of one GPU's tile (default = 262144 bodies)")
    ap.add_argument("--iterations", type=int, default=20)
    ap.add_argument("--settle", type=int, default=SETTLE_STEPS, help="development only: anything but 240 is flagged in the output")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="N > 1: grow the pen with N (weak) or cut the one pile into N tiles (strong)")
    ap.add_argument("--tiles-z", type=int, default=1, help="N > 1: tiles along z (tiles along x = N / tiles-z)")
    ap.add_argument("--ghost-margin", type=float, default=2.5)
    ap.add_argument("--transport", choices=["rccl", "dist"], default=os.environ.get("MI_SHARD_TRANSPORT", "rccl"),
                    help="N > 1: neighbour exchange by the library's own RCCL send/recv (default) or by torch.distributed point-to-point through host buffers")
    ap.add_argument("--seam", choices=["jacobi", "exact"], default="jacobi",
                    help="N > 1: block-Jacobi seam (default: one exchange per step) or the exact seam (slabs only: seam colours first, one hand-over per sweep; "
                         "== one world told the tiling, bit for bit; about twice the cost) — include/mi_shard.h")
    ap.add_argument("--rebalance-every", type=int, default=0, help="N > 1: a load-balance round (tile borders follow the body counts) every K steps of the untimed settle phase; 0 = fixed uniform tiles")
    ap.add_argument("--pmc", action="store_true", help="N = 1: measure roofline.traffic for THIS run (two extra rocprofv3 --pmc passes of the same command: FETCH_SIZE, WRITE_SIZE) "
                                                      "instead of scaling the committed figure of profiles/traffic.json")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-at-rest", action="store_true", help="skip the second measurement after 1500 steps")
    ap.add_argument("--cpu-grid", type=int, nargs=3, default=[32, 16, 32])
    ap.add_argument("--cpu-steps", type=int, default=20)
    ap.add_argument("--cpu-cores", type=int, default=0, help="reference replicas for the CPU baseline (0 = all usable host cores)")
    return ap.parse_args()


# --------------------------------------------------------------------------------------------------------------- CPU baseline
def _cpu_replica(job):
    """One replica of the CPU baseline: the same generator at a smaller footprint, settled with the same SETTLE_STEPS, then timed.
    kind "reference": the reference's own physicsStep (oracle/_ref/libref_fast.so), AVX2 path and, optionally, scalar path from
    the same state; kind "port": the oracle restatement (only if the reference library is unavailable)."""
    nx, ny, nz, iterations, settle, steps, kind, with_scalar = job
    import ctypes as C
    import oracle
    from d3d12renderer_amd import scenes
    sc = scenes.obb_pile(nx, ny, nz, solver_iterations=iterations)
    s = sc.settings()
    out = {"bodies": sc.num_bodies}
    if kind == "reference":
        w = sc.populate(oracle.create_reference_world(simd=True, fast=True))
        w.step_fixed(s, sc.dt, settle)
        t0 = time.perf_counter(); w.step_fixed(s, sc.dt, steps); out["avx2"] = steps / (time.perf_counter() - t0)
        out["contacts"] = w.counts()["num_contacts"]
        if with_scalar:
            w.L.check(w.L.fn("world_set_simd")(w.h, C.c_uint32(0)), "world_set_simd")
            n = max(4, steps // 2)
            t0 = time.perf_counter(); w.step_fixed(s, sc.dt, n); out["scalar"] = n / (time.perf_counter() - t0)
    else:
        w = sc.populate(oracle.create_world(oracle.ORDER_REFERENCE))
        w.step_fixed(s, sc.dt, settle)
        t0 = time.perf_counter(); w.step_fixed(s, sc.dt, steps); out["scalar"] = steps / (time.perf_counter() - t0)
        out["contacts"] = w.counts()["num_contacts"]
    return out


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


def cpu_baseline(args, bodies_full, settle):
    """The reference timed beside the GPU on this box's host cores: (i) its scalar path on 1 core, (ii) its AVX2
    path (physics_settings::simd* = true, the default of the original) on 1 core, (iii) the AVX2 path on all usable cores as
    independent replicas (the reference step has no intra-step threading).  Sample: the same generator, 32 x 16 x 32 = 16 384
    boxes (the reference's 16-bit collider indices end at 65 535), settled with the same 240 steps; throughput is expressed in
    steps/s of the full 262 144-body scene, assuming cost linear in bodies (contacts per body are the same)."""
    import subprocess
    import oracle
    oracle.build()
    kind = "reference" if (oracle.REF_LIB.with_name("libref_fast.so").exists() or oracle.reference_available()) else "port"
    if kind == "reference":
        oracle.build_reference()
    nx, ny, nz = args.cpu_grid
    cores = max(1, min(args.cpu_cores or _usable_cores(), 64))
    env = dict(os.environ, OMP_NUM_THREADS="1")

    def launch(n, with_scalar):
        job = [nx, ny, nz, args.iterations, settle, args.cpu_steps, kind, with_scalar]
        cmd = [sys.executable, str(Path(__file__).resolve()), "--cpu-replica", json.dumps(job)]
        procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env) for _ in range(n)]
        res = []
        for p in procs:
            out, _ = p.communicate(timeout=1200)
            if p.returncode == 0 and out.strip():
                res.append(json.loads(out.strip().splitlines()[-1]))
        return res

    single = launch(1, True)           # one replica alone on the box: uncontended 1-core figures
    if not single:
        raise RuntimeError("cpu_baseline: the replica process failed")
    nb = single[0]["bodies"]; scale = nb / bodies_full
    fastest = "avx2" if kind == "reference" else "scalar"
    many = launch(cores, False) if cores > 1 else single
    agg = sum(r[fastest] for r in many)
    out = {
        "value": agg * scale, "unit": "steps/s", "cores": len(many), "kind": kind,
        "sample": (f"{'the reference itself (oracle/_ref/libref_fast.so: its sources, -O2 -ffast-math -mavx2 -mfma), AVX2 path' if kind == 'reference' else 'oracle restatement (scalar, -O2 strict fp32)'} "
                   f"on obb_pile {nx}x{ny}x{nz} = {nb} bodies, I={args.iterations}, {settle} settle + {args.cpu_steps} timed steps "
                   f"({single[0]['contacts']} contacts = {single[0]['contacts'] / nb:.2f} per body), {len(many)} independent single-threaded replicas "
                   f"running concurrently: {agg:.1f} replica-steps/s in aggregate; value = aggregate x {nb}/{bodies_full} (linear in bodies)"),
        "sample_bodies": nb, "sample_contacts": single[0]["contacts"], "settle_steps": settle,
        "scalar_1core": single[0].get("scalar", 0.0) * scale,
        "avx2_1core": single[0].get("avx2", 0.0) * scale if kind == "reference" else None,
        "avx2_all_cores": agg * scale if kind == "reference" else None,
        "measured_replica_steps_per_s": {"scalar_1core": single[0].get("scalar"), "avx2_1core": single[0].get("avx2"), "aggregate": agg},
    }
    return out


# --------------------------------------------------------------------------------------------------------------- GPU legs
def step_algorithmic_bytes(c, iterations):
    """B_step of SURVEY.md §8(d) / BASELINE.md §3 for a contacts-only scene, from this step's own counts."""
    nc, nb, p, m, k = c["num_colliders"], c["num_rigid_bodies"], c["num_broadphase_overlaps"], c["num_collisions"], c["num_contacts"]
    return (160 * nc + (24 * nc + 8 * p) + (112 * p + 40 * k + 9 * m) + 270 * nb + 316 * k + iterations * BYTES_PER_CONTACT_ITER * k + 140 * nb)


def timed_region(sw, settings, dt, steps, barrier):
    import gc
    gc.collect(); gc.disable()          # a generation-2 collection of the interpreter inside a 20 ms window is a 10 % outlier; nothing is allocated in the loop but floats
    try:
        return _timed_region(sw, settings, dt, steps, barrier)
    finally:
        gc.enable()


def _timed_region(sw, settings, dt, steps, barrier):
    barrier()
    t0 = time.perf_counter()
    step_ms = []
    sw.world.accumulated_stage_times(reset=True)      # the library sums the per-stage device times of the timed steps itself
    for _ in range(steps):
        t_step = time.perf_counter()
        sw.step(settings, dt)
        step_ms.append((time.perf_counter() - t_step) * 1e3)
    barrier()
    elapsed = time.perf_counter() - t0
    stage_acc, n_acc, contact_iters = sw.world.accumulated_stage_times()
    assert n_acc == steps
    return elapsed, step_ms, stage_acc, contact_iters


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-replica":     # worker process of cpu_baseline()
        print(json.dumps(_cpu_replica(json.loads(sys.argv[2]))))
        return
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as typed: become the launcher of N ranks (one process per GPU, torch.distributed.run on this node) and pass
        # their output through — rank 0 prints the ONE JSON line.  (Launched by torch.distributed.run already: WORLD_SIZE is set, this is a rank.)
        sys.exit(_launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    device = local_rank % max(1, torch.cuda.device_count())     # (several ranks on one GPU only make sense with --transport dist: a functional check)
    torch.cuda.set_device(device)   # torch's HIP runtime comes up before the library touches the device
    red_dev = "cuda"
    if world_size > 1:
        import torch.distributed as dist
        if args.transport == "rccl" and torch.cuda.device_count() < world_size:
            args.transport = "dist"      # ranks share a GPU (a functional check on a small box): RCCL wants one device per rank
        if args.transport == "rccl":
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device("cuda", device))
            except Exception as e:      # noqa: BLE001 — every rank sees the same failure (same image, same topology); the run goes on over gloo
                print(f"bench.py: RCCL process group failed ({e}); neighbour messages via host over gloo", file=sys.stderr)
                args.transport = "dist"
        if args.transport != "rccl":
            dist.init_process_group("gloo", rank=rank, world_size=world_size); red_dev = "cpu"
    if world_size != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_size}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)

    import d3d12renderer_amd as mi
    from d3d12renderer_amd import scenes, sharding

    nx, ny, nz = args.grid
    gx = nx * world_size if (world_size > 1 and args.scaling == "weak") else nx
    scene = scenes.obb_pile(gx, ny, nz, solver_iterations=args.iterations)       # the WHOLE scene, on every rank
    world = scene.populate(mi.create_world(device))

    class _Single:                      # N = 1: the plain world
        def __init__(self, w): self.world = w
        def step(self, s, dt): self.world.step_fixed(s, dt, 1)
    if world_size > 1:
        desc = sharding.tile_grid(scene, world_size, args.tiles_z, args.ghost_margin)
        sw = sharding.ShardedWorld(world, desc, rank, args.transport, dist)
        if args.seam == "exact":
            sw.enable_exact_seam()
        sharding_note = (f"{world_size} tiles ({desc.tiles_x} x {desc.tiles_z}) of one replicated scene, ghost margin {desc.ghost_margin:.2f} m, ownership by position each step, "
                         f"{desc.max_records} records (56 B) per neighbour message, "
                         f"neighbour exchange: {'RCCL send/recv inside the library' if sw.transport == 'rccl' else 'torch.distributed p2p via host'}; {args.scaling} scaling; seam: {'exact (one hand-over per sweep)' if args.seam == 'exact' else 'block Jacobi'}"
                         + (f"; {sw.note}" if sw.note else ""))
    else:
        sw = _Single(world); sharding_note = "single GPU, whole scene"
    settings = scene.settings()
    # HIP events over the timed region, as the bench contract wants them (roofline.achieved = bytes / the solver launch's duration from events on the world's
    # stream): level 3 = the solve stage alone — the dominant kernel's launch, which is what the roofline needs.  The library itself times nothing by default: the
    # start / stop events riding on the solver's dispatch cost ~11 us of idle device per step (~1.2 %), i.e. a caller who does not ask for times steps that much
    # faster than this bench reports.  (Rounds 2-4 also timed the whole step over the timed region — two more gaps, ~10 us per step; the whole step's device time
    # now comes from three extra steps after the timed region, like the per-stage breakdown.)
    TIMED_LEVEL = 3
    sw.world.set_stage_timing(TIMED_LEVEL)
    dt = scene.dt

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the workload's state: settle (always, untimed), then the caller's warm-up
    for i in range(args.settle):
        sw.step(settings, dt)
        if world_size > 1 and args.rebalance_every and i % args.rebalance_every == args.rebalance_every - 1 and i + 2 < args.settle:
            sw.rebalance()                              # (control plane: one small all-reduce; never inside the timed region)
    for _ in range(args.warmup):
        sw.step(settings, dt)
    barrier()
    c0 = sw.world.counts()
    total_bodies = scene.num_bodies
    bodies_per_gpu = sw.world.shard_counts()["owned_bodies"] if world_size > 1 else total_bodies
    local_bodies = max(1, bodies_per_gpu)            # N > 1: local counts cover the tile + its ghost strip; per owned body is close enough for the guard
    contacts_per_body = c0["num_contacts"] / (local_bodies if world_size > 1 else max(1, c0["num_rigid_bodies"]))
    if contacts_per_body < MIN_CONTACTS_PER_BODY and args.settle >= SETTLE_STEPS:
        raise SystemExit(f"bench.py: {contacts_per_body:.2f} contacts per body after {args.settle} settle steps — this is not the piled-up workload")

    if world_size > 1:
        sw.world.shard_exchange_stats(reset=True)       # (also reports a message overflow of the settle phase before anything is timed)
    mode0 = sw.world.step_mode_stats()
    elapsed, step_ms, stage_acc, contact_iters = timed_region(sw, settings, dt, args.steps, barrier)
    mode1 = sw.world.step_mode_stats()
    ex_timed = sw.world.shard_exchange_stats() if world_size > 1 else None     # (the exchanges of exactly the timed steps)
    solve_ms = stage_acc["solve"]
    launches = sw.world.solve_launches() * args.steps
    counts = sw.world.counts()

    # per-stage breakdown + a dedicated HIP event pair around every solver launch: 3 extra steps outside the timed region
    prof_launches = 0; prof_ms = 0.0; prof_updates = 0
    sw.world.set_stage_timing(1)       # every stage
    stage_prof = {}
    for _ in range(3):
        n_l, ms, upd = sw.world.step_profiled(settings, dt)
        if world_size > 1:
            sw.exchange()
        for k, v in sw.world.stage_times().items():
            stage_prof[k] = stage_prof.get(k, 0.0) + v / 3.0
        prof_launches += n_l; prof_ms += ms; prof_updates += upd
    sw.world.set_stage_timing(2)       # the whole step and the solve stage: three more steps for the whole step's device time
    total_dev_ms = 0.0
    for _ in range(3):
        sw.step(settings, dt)
        total_dev_ms += sw.world.stage_times()["total"] / 3.0
    total_dev_ms *= args.steps           # (kept as a sum over the timed steps' count: the fields below divide by args.steps)
    # the same number of steps once more with the events of rounds 2-4 (whole step + solve stage): what the protocol change of round 5 is worth on THIS box, in the line itself
    elapsed_l2 = None
    if world_size == 1:
        elapsed_l2, _, acc_l2, _ = timed_region(sw, settings, dt, args.steps, barrier)
        total_dev_ms_l2 = acc_l2.get("total", 0.0)
    sw.world.set_stage_timing(TIMED_LEVEL)

    # ---- second state: the same pile at rest (1500 steps in total)
    at_rest = None
    if not args.no_at_rest and world_size == 1 and args.settle >= SETTLE_STEPS:
        done = args.settle + args.warmup + args.steps + 6
        for _ in range(max(0, AT_REST_STEPS - done)):
            sw.step(settings, dt)
        n_rest = min(120, args.steps)
        e2, sm2, acc2, ci2 = timed_region(sw, settings, dt, n_rest, barrier)
        c2 = sw.world.counts()
        at_rest = {"steps_before": max(done, AT_REST_STEPS), "timed_steps": n_rest, "value": n_rest / e2, "unit": "steps/s", "ms_per_step": e2 / n_rest * 1e3,
                   "contacts": c2["num_contacts"], "manifolds": c2["num_collisions"], "broadphase_overlaps": c2["num_broadphase_overlaps"],
                   "contacts_per_body": c2["num_contacts"] / max(1, c2["num_rigid_bodies"]),
                   "solver_avg_launch_us": acc2["solve"] * 1e3 / max(1, sw.world.solve_launches() * n_rest),
                   "solver_frac_algorithmic": (BYTES_PER_CONTACT_ITER * ci2) / (acc2["solve"] * 1e-3) / 1e9 / HBM_PEAK_GBPS if acc2["solve"] > 0 else None,
                   "step_frac_algorithmic": step_algorithmic_bytes(c2, args.iterations) / (e2 / n_rest) / 1e9 / HBM_PEAK_GBPS}

    global_counts = None
    per_rank = None
    if dist is not None:
        # what explains a scaling curve: every rank's own clock, solver roofline, exchange cost and what it moved (gathered on rank 0)
        ex = ex_timed
        lc = sw.world.counts()
        l_launches = max(sw.world.solve_launches() * args.steps, 1)
        l_solve_s = stage_acc["solve"] * 1e-3 / l_launches
        mine = {
            "rank": rank, "device": device, "elapsed_s": elapsed, "ms_per_step": elapsed / args.steps * 1e3, "device_ms_per_step": total_dev_ms / args.steps,
            "owned_bodies": ex["owned_bodies"], "ghost_bodies": ex["ghost_bodies"], "local_contacts": lc["num_contacts"], "local_manifolds": lc["num_collisions"], "colors": lc["num_colors"],
            "roofline": {"kernel": sw.world.solver_kernel(), "avg_launch_us": l_solve_s * 1e6,
                         "achieved_GBps": (BYTES_PER_CONTACT_ITER * contact_iters / l_launches) / l_solve_s / 1e9 if l_solve_s > 0 else 0.0,
                         "frac": (BYTES_PER_CONTACT_ITER * contact_iters / l_launches) / l_solve_s / 1e9 / HBM_PEAK_GBPS if l_solve_s > 0 else 0.0},
            "exchange": {"transport": "library RCCL (ncclSend / ncclRecv + 72-byte ncclAllReduce on the world's stream)" if ex["library_transport"] else "caller's (torch.distributed via host)",
                         "exchanges_timed": ex["exchanges"], "device_ms_per_exchange": ex["device_ms_sum"] / max(1, ex["exchanges"]),
                         "message_bytes_per_neighbour": ex["message_bytes"], "neighbours": ex["neighbour_rank"],
                         "message_records_travelled_last": ex.get("message_records_last"), "bytes_sent_timed": ex.get("message_bytes_sum"),   # library transport: messages sized from the previous exchange (include/mi_shard.h)
                         "records_per_exchange": [v / max(1, ex["exchanges"]) for v in ex["records_sum"]],
                         "payload_bytes_per_exchange": [56 * v / max(1, ex["exchanges"]) for v in ex["records_sum"]],
                         "seam": args.seam, "seam_stats": sw.check_seam() if args.seam == "exact" else None,   # (raises when a manifold violated the seam classes: the ranks would no longer equal one world)
                         "sweep_exchanges_timed": ex["sweep_exchanges"], "sweep_message_bytes_per_neighbour": ex["sweep_message_bytes"],
                         "sweep_payload_bytes_per_exchange": [32 * v for v in ex["sweep_records_last"]]},
        }
        gathered = [None] * world_size
        dist.all_gather_object(gathered, mine)
        per_rank = gathered
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        oc = sw.world.shard_counts()
        g = torch.tensor([oc["owned_bodies"], oc["owned_manifolds"], oc["owned_contacts"]], dtype=torch.int64, device=red_dev)
        dist.all_reduce(g)
        global_counts = {"bodies": int(g[0]), "manifolds": int(g[1]), "contacts": int(g[2])}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        # whole-job throughput.  weak: every rank steps a 262144-body tile of an N times larger pen per step -> tiles-steps per second
        # (N = 1: plain steps/s of the 262144-body scene); strong: steps per second of the ONE 262144-body scene
        value = (world_size if args.scaling == "weak" else 1) * args.steps / elapsed
        launches_per_step = launches / args.steps
        kernel = sw.world.solver_kernel()
        avg_launch_s = solve_ms * 1e-3 / max(launches, 1)
        alg_per_launch = BYTES_PER_CONTACT_ITER * contact_iters / max(launches, 1)
        achieved = alg_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        traffic, traffic_note = _scaled_traffic(kernel, contact_iters / max(launches, 1))
        traffic_source = "committed constant (profiles/traffic.json) x this run's contact-sweeps" if traffic else None
        if args.pmc and world_size == 1:
            m, note = _measured_traffic(kernel, args)
            if m:
                traffic, traffic_note, traffic_source = m, note, "measured: rocprofv3 --pmc passes of this command, this run"
            else:
                traffic_note += f" (--pmc asked for but not measured: {note})"
        b_step = step_algorithmic_bytes(counts, args.iterations)
        frac = achieved / HBM_PEAK_GBPS
        traffic_frac = (traffic / avg_launch_s / 1e9 / HBM_PEAK_GBPS) if traffic and avg_launch_s > 0 else None
        # the serial chain of one launch: colours x sweeps hops (a body of maximal degree is a chain by itself: DESIGN.md §4); a launch covers all sweeps of a step
        sweeps_per_launch = args.iterations / max(launches_per_step, 1)
        hops = counts["num_colors"] * sweeps_per_launch
        roofline = {
            # the memory interface is the resource this kernel class is priced against; when the algorithmic fraction exceeds what the counters see crossing it
            # (bytes that stay in LDS / L2), bytes are not what bounds the launch: the chain below is
            "bound": "latency (dependency chain)" if (traffic_frac is not None and frac > traffic_frac) else "hbm", "bound_resource": "hbm",
            "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": frac, "frac_is_algorithmic": True, "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBPS, "achievable": HBM_ACHIEVABLE_GBPS, "traffic": traffic,
            # what crosses the HBM interface is LESS than the algorithmic bytes (impulses stay in LDS, ~95 % of the body hand-overs in L2):
            # `frac` says how fast the algorithmic work is done, `traffic_frac` how busy the memory interface really is
            "traffic_frac": traffic_frac,
            "chain": {"hops": hops, "colors": counts["num_colors"], "sweeps_per_launch": sweeps_per_launch,
                      "us_per_hop": (avg_launch_s * 1e6 / hops) if hops else None,
                      "round_trip_us": _committed_round_trip_us(),
                      "note": "hops = colours x sweeps of one launch; us_per_hop = this run's launch time / hops; round_trip_us = 'body loads issued -> first tag check' of the "
                              "committed per-visit stamps (profiles/solver_hop.json; a development build, not this run)"},
            "traffic_note": traffic_note, "traffic_source": traffic_source,
            "avg_launch_us": avg_launch_s * 1e6, "launches_per_step": launches_per_step,
            "algorithmic_bytes_per_launch": alg_per_launch,
            "bound_note": ("the kernel is bound by its dependency chain (colours x sweeps serial hops), not by bytes: DESIGN.md §4"),
            "event_pair_per_launch": {"avg_launch_us": prof_ms * 1e3 / max(prof_launches, 1), "launches_per_step": prof_launches / 3,
                                      "achieved_GBps": (BYTES_PER_CONTACT_ITER * prof_updates) / (prof_ms * 1e-3) / 1e9 if prof_ms > 0 else 0.0,
                                      "note": "3 extra steps (outside the timed region) with a dedicated HIP event pair around each solver launch"},
            "whole_step": {"algorithmic_bytes": b_step, "ms": total_dev_ms / args.steps, "achieved_GBps": b_step / (total_dev_ms / args.steps * 1e-3) / 1e9,
                           "frac": b_step / (total_dev_ms / args.steps * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                           "frac_of_achievable": b_step / (total_dev_ms / args.steps * 1e-3) / 1e9 / HBM_ACHIEVABLE_GBPS,
                           "note": "B_step of SURVEY.md §8(d) from this step's counts / device time of the whole step (HIP events on the world's stream; mean of three extra steps after the timed region — inside it only the solver launch carries events)"},
            "note": ("rank 0, timed region: 236 B x contacts x sweeps / HIP-event time of the solve stage on the world's stream; "
                     "rocprofv3 --kernel-trace --stats of the same command: profiles/"),
        }
        # N > 1: the whole scene's contacts per body (owner-rule counts summed over the ranks); rank 0's local counts cover its tile + ghost strip
        cpb_timed = (global_counts["contacts"] / max(1, global_counts["bodies"])) if global_counts else counts["num_contacts"] / max(1, counts["num_rigid_bodies"])
        out = {
            "metric": (f"physics steps/sec of ONE {total_bodies}-body OBB pile cut into {world_size} tiles ({args.iterations} solver iterations)" if world_size > 1 and args.scaling == "strong"
                       else f"physics steps/sec at {nx * ny * nz} rigid bodies per GPU (OBB pile, {args.iterations} solver iterations)"),
            "value": value, "unit": "steps/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling if world_size > 1 else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"cfg3 obb_pile {gx}x{ny}x{nz} boxes ({total_bodies} bodies in total, {bodies_per_gpu} owned by rank 0), friction 0.5, "
                                    f"{args.iterations} solver iterations, dt=1/120, settled {args.settle} steps before warm-up "
                                    f"({cpb_timed:.2f} contacts per body when timed)"),
                       "settle_steps": args.settle, "settle_is_standard": args.settle == SETTLE_STEPS,
                       "bodies_per_gpu": bodies_per_gpu, "contacts": counts["num_contacts"], "manifolds": counts["num_collisions"],
                       "contacts_per_body": cpb_timed,
                       "broadphase_overlaps": counts["num_broadphase_overlaps"], "colors": counts["num_colors"],
                       "global_counts": global_counts, "sharding": sharding_note},
            "roofline": roofline,
            "stage_ms": stage_prof, "stage_ms_note": "per-stage device times of the 3 extra steps after the timed region (stage timing enabled only there)",
            "step_ms_median": float(np.median(step_ms)),
            "step_ms_p95": float(np.percentile(step_ms, 95)) if args.steps >= 50 else None,
            "step_ms_p95_note": None if args.steps >= 50 else f"not reported: {args.steps} timed steps are too few for a 95th percentile (>= 50); step_ms_max is the slowest of them",
            "step_ms_max": float(np.max(step_ms)),
            "timed_ms_total": elapsed * 1e3,
            "step_modes_timed": {"internal_steps": mode1[0] - mode0[0], "speculative": mode1[1] - mode0[1], "synchronous_reruns": mode1[2] - mode0[2],
                                 "note": "a speculative step sizes its launches from the previous step and reads back once; one whose bounds did not hold is re-run synchronously (counted here, timed like any step)"},
            "device_ms_per_step": total_dev_ms / args.steps,
            "device_ms_per_step_note": "mean of a 3-step sample taken AFTER the timed region with whole-step events on (not the timed steps themselves)",
            "timed_region_events": "solver launch only (since r05; rounds 2-4 also bracketed the whole step: two more ~5 us gaps per step)",
            "with_whole_step_events": ({"value": args.steps / elapsed_l2, "ms_per_step": elapsed_l2 / args.steps * 1e3, "device_ms_per_step": total_dev_ms_l2 / args.steps,
                                        "note": "the same number of steps timed again right after the 3-step samples, with the rounds-2-4 event set (whole step + solver): compare with `value` for what measuring less is worth"}
                                       if elapsed_l2 else None),
            "solver_kind": sw.world.solver_kind(),
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
            out["per_rank_note"] = ("every rank's own wall clock over the timed steps, its solver launch against the HBM roofline, the device time of one exchange "
                                    "(pack -> send / receive -> unpack -> axis) and the records (56 B) it sent per neighbour; value uses the slowest rank")
        if world_size > 1:
            bx, bz = sw.world.shard_get_borders(desc.tiles_x, desc.tiles_z)
            out["config"]["tile_borders"] = {"x": [float(v) for v in bx], "z": [float(v) for v in bz],
                                             "rebalanced_every": args.rebalance_every if args.rebalance_every else "never (uniform tiles)"}
        if at_rest is not None:
            out["at_rest"] = at_rest
        if not args.no_cpu_baseline and world_size == 1:   # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args, bodies_per_gpu, args.settle)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _launch_ranks(n):
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0"); env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(Path(__file__).resolve()), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def _measured_traffic(kernel, args):
    """HBM bytes per launch of `kernel` for THIS workload: two extra passes of the same command under rocprofv3 (--pmc FETCH_SIZE, then
    --pmc WRITE_SIZE — separately, kernel-trace only, as MI355X_MICROARCH.md prescribes), mean over the kernel's last steps + 3 dispatches
    (the timed and the profiled steps); bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE reports half of a wide read stream)."""
    import csv
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if Path("/opt/rocm/bin/rocprofv3").exists() else None)
    if not prof:
        return None, "rocprofv3 not found"
    child = [sys.executable, str(Path(__file__).resolve()), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup), "--grid", *map(str, args.grid),
             "--iterations", str(args.iterations), "--settle", str(args.settle), "--no-cpu-baseline", "--no-at-rest"]
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
        try:
            r = subprocess.run([prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--", *child],
                               capture_output=True, text=True, timeout=1200, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp")
            files = sorted(Path(d).rglob("*_counter_collection.csv"))
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode})"
            v = []
            with open(files[0]) as f:
                for row in csv.DictReader(f):
                    if row.get("Counter_Name") == counter and kernel in row["Kernel_Name"]:
                        v.append(float(row["Counter_Value"]))
            if not v:
                return None, f"no {counter} samples for {kernel}"
            n = min(len(v), args.steps + 3)
            vals[counter] = sum(v[-n:]) / n
        finally:
            shutil.rmtree(d, ignore_errors=True)
    b = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    return b, (f"(2 x FETCH_SIZE {vals['FETCH_SIZE']:.0f} KiB + WRITE_SIZE {vals['WRITE_SIZE']:.0f} KiB) per launch, mean of the kernel's last {args.steps + 3} dispatches in two separate "
               f"rocprofv3 --pmc passes of this command (gfx950: FETCH_SIZE reports half of a wide coalesced read stream)")


def _committed_round_trip_us():
    """'body loads issued -> first tag check' of the persistent solver's per-visit wall-clock stamps (a -DMI_DBG_TIMELINE build, tools/gpu_timeline2.sh), as committed
    in profiles/solver_hop.json; None if the file is absent."""
    p = ROOT / "profiles" / "solver_hop.json"
    try:
        return json.loads(p.read_text())["issue_to_first_check_us"]
    except Exception:   # noqa: BLE001
        return None


def _scaled_traffic(kernel, contact_sweeps_per_launch):
    """HBM bytes per launch of the dominant kernel: the committed rocprofv3 --pmc passes (profiles/traffic.json: separate FETCH_SIZE /
    WRITE_SIZE passes, gfx950 corrections) give bytes per contact-sweep of that kernel at the profiled state; scaled by THIS run's
    contact-sweeps per launch.  None when there is no committed figure for the kernel (PMC counters cannot be read from inside this
    process)."""
    p = ROOT / "profiles" / "traffic.json"
    try:
        t = json.loads(p.read_text())
        per = t[kernel + "_bytes_per_contact_sweep"]
        return per * contact_sweeps_per_launch, (f"{per:.1f} B per contact-sweep from the committed PMC passes ({t.get('source', 'profiles/')}) x this run's "
                                                 f"{contact_sweeps_per_launch:.0f} contact-sweeps per launch")
    except Exception:   # noqa: BLE001
        return None, "no committed PMC figure for this kernel"


if __name__ == "__main__":
    main()
