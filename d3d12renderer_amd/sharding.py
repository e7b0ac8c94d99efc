"""Driver of a sharded world (include/mi_shard.h) from Python: tile grid from the scene, and the three ways the neighbour messages
can travel.  The sharding itself — ownership by the position of the centre of gravity, ghosts, migration, packing and unpacking the
records, the owner rule for counts — is in the library (csrc/kernels.hpp k_shard_*, csrc/world.hip) behind the C ABI; a C++ host uses
mi_world_shard_* directly and never sees this file.

  transport "rccl"   the library's own: RCCL send / receive on the world's stream (one process per GPU; torch.distributed is only used
                     to hand the 128-byte ncclUniqueId round)
  transport "dist"   the caller's, over torch.distributed point-to-point with host buffers (gloo on CPU: the tests)
  transport "local"  the caller's, between several worlds of ONE process (virtual ranks: what the multi-process result is compared with)
"""
import numpy as np

from . import capi


def tile_grid(scene, num_ranks, tiles_z=1, margin=2.5):
    """An x-z tile grid over the dynamic bodies of `scene`: num_ranks / tiles_z tiles along x, tiles_z along z."""
    assert num_ranks % tiles_z == 0
    tiles_x = num_ranks // tiles_z
    dyn = scene.entities["kind"] != capi.ENTITY_STATIC
    p = scene.entities["position"][dyn]
    lo = p.min(axis=0) - 0.5; hi = p.max(axis=0) + 0.5
    d = capi.ShardDesc()
    d.num_ranks = num_ranks; d.tiles_x = tiles_x; d.tiles_z = tiles_z
    d.origin_x = float(lo[0]); d.origin_z = float(lo[2])
    d.tile_size_x = float((hi[0] - lo[0]) / tiles_x); d.tile_size_z = float((hi[2] - lo[2]) / tiles_z)
    d.ghost_margin = float(min(margin, 0.45 * d.tile_size_x, 0.45 * d.tile_size_z))
    # A neighbour message always travels whole, so its capacity is sized for what a margin strip can hold (x 6: piles shift and compact), not for
    # the library's size-blind default: records = owned bodies whose centre lies within `margin` of the border to that neighbour.
    per_tile = int(dyn.sum()) / num_ranks
    strip = d.ghost_margin / d.tile_size_x if tiles_x > 1 else 0.0
    strip = max(strip, d.ghost_margin / d.tile_size_z if tiles_z > 1 else 0.0)
    d.max_records = max(4096, int(6.0 * strip * per_tile) + 1024)
    return d


def _desc_for(desc, rank):
    d = capi.ShardDesc()
    for name, _ in capi.ShardDesc._fields_:
        setattr(d, name, getattr(desc, name))
    d.rank = rank
    return d


class ShardedWorld:
    """One rank of a sharded scene.  `world` already holds the WHOLE scene (every rank populates the same one)."""

    def __init__(self, world, desc, rank, transport="local", dist=None):
        self.world, self.rank, self.transport, self.dist = world, rank, transport, dist
        self.desc = _desc_for(desc, rank)
        world.shard_enable(self.desc)
        self.neighbours = world.shard_neighbours()
        self.note = ""
        if transport == "rccl":
            # the library's own transport; if it cannot come up on ANY rank (no librccl, communicator refused) every rank falls back to
            # the caller's transport over the process group that is already there — slower (messages via host), same results.
            # ncclCommInitRank blocks until EVERY rank has entered it, so the ranks first agree — with a cheap, non-collective probe and one
            # all-reduce over the existing group — that all of them can; only then does anyone attach.
            import torch
            dev = "cuda" if dist.get_backend() == "nccl" else "cpu"

            def agree(failed):
                flag = torch.tensor([1 if failed else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                return bool(int(flag.item()))

            if agree(not world.L.shard_library_transport_available()):
                self.transport = "dist"; self.note = "library RCCL transport unavailable on some rank (librccl not found): neighbour messages go through torch.distributed"
                return
            failed = False
            try:
                ident = [world.L.shard_unique_id() if rank == 0 else None]
            except Exception as e:      # noqa: BLE001
                ident, failed, self.note = [None], True, str(e)
            dist.broadcast_object_list(ident, src=0)
            if ident[0] is None:        # rank 0 could not create the id: everybody learns it from the broadcast, nobody attaches
                self.transport = "dist"; self.note = "library RCCL transport unavailable (" + (self.note or "no unique id from rank 0") + "): neighbour messages go through torch.distributed"
                return
            try:
                world.shard_attach_rccl(ident[0])
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                failed, self.note = True, str(e)
            if agree(failed):
                if not failed:
                    world.shard_detach_rccl()
                self.transport = "dist"
                self.note = "library RCCL transport unavailable (" + (self.note or "on another rank") + "): neighbour messages go through torch.distributed"

    def enable_exact_seam(self, enable=True):
        """The exact seam (include/mi_shard.h): every sweep of a step ends with a hand-over of the shared bodies' velocities.  Library transport: inside
        the step, on the world's stream.  "dist": this process's sweep callback exchanges the messages with the neighbours over torch.distributed.
        "local" (virtual ranks): see step_local_exact — the ranks have to step side by side."""
        if self.transport == "rccl":
            self.world.shard_set_exact_seam(enable, None)
        elif self.transport == "dist":
            self.world.shard_set_exact_seam(enable, self._sweep_dist if enable else None)
        else:
            raise RuntimeError("virtual ranks step side by side: sharding.step_local_exact(ranks, ...)")
        self.exact = bool(enable)

    def check_seam(self):
        """Exact seam only: the step's seam statistics; raises when a manifold violated the seam classes (a body within the margin of two borders, a tile
        narrower than two margins): the ranks would no longer equal the single world told the tiling."""
        st = self.world.seam_stats()
        if getattr(self, "exact", False) and st["violations"]:
            raise RuntimeError(f"exact seam: {st['violations']} manifolds violate the seam classes this step (tiles must stay wider than two ghost margins)")
        return st

    def _sweep_dist(self, sweep):
        import torch
        dev = "cuda" if self.dist.get_backend() == "nccl" else "cpu"
        ops, inbox, keep = [], [], []
        full = self.world.shard_sweep_message_bytes() // 4          # point-to-point messages of a fixed size (the receiver does not know the count)
        for slot, peer in enumerate(self.neighbours):
            msg = np.zeros(full, np.float32); part = self.world.shard_export_sweep(slot); msg[: len(part)] = part
            out = torch.from_numpy(msg).to(dev); keep.append(out)
            buf = torch.zeros(out.numel(), dtype=torch.float32, device=dev); inbox.append(buf)
            ops.append(self.dist.P2POp(self.dist.isend, out, peer)); ops.append(self.dist.P2POp(self.dist.irecv, buf, peer))
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        for buf in inbox:
            self.world.shard_import_sweep(buf.cpu().numpy())

    def step(self, settings, dt):
        """One internal step of this rank's tile; with the library transport the exchange is part of it."""
        self.world.step_fixed(settings, dt, 1)
        self.exchange()

    def exchange(self):
        """The caller's-transport half of a step (nothing to do when the library exchanges itself)."""
        if self.transport == "dist":
            self._exchange_dist()

    def outgoing(self):
        return {peer: self.world.shard_export(slot) for slot, peer in enumerate(self.neighbours)}

    def _exchange_dist(self):
        import torch
        dev = "cuda" if self.dist.get_backend() == "nccl" else "cpu"      # RCCL moves device memory only
        ops, inbox, keep = [], [], []
        for slot, peer in enumerate(self.neighbours):
            out = torch.from_numpy(self.world.shard_export(slot)).to(dev); keep.append(out)
            buf = torch.zeros(out.numel(), dtype=torch.float32, device=dev); inbox.append(buf)
            ops.append(self.dist.P2POp(self.dist.isend, out, peer)); ops.append(self.dist.P2POp(self.dist.irecv, buf, peer))
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        for buf in inbox:
            self.world.shard_import(buf.cpu().numpy())
        # global sweep axis: the centre statistics of the colliders every rank owns, summed over the ranks (9 integers)
        sums = torch.from_numpy(self.world.shard_axis_sums().view(np.int64).copy()).to(dev)
        self.dist.all_reduce(sums)
        self.world.shard_set_axis_sums(sums.cpu().numpy().view(np.uint64))

    # --- load balance: move the tile borders to where the bodies are ("rebalanced every K steps by body count")
    BALANCE_BINS = 256

    def _extent(self, axis):
        d = self.desc
        return (d.origin_x, d.origin_x + d.tiles_x * d.tile_size_x) if axis == 0 else (d.origin_z, d.origin_z + d.tiles_z * d.tile_size_z)

    def histograms(self):
        """This rank's owned bodies along x and z (uint64, BALANCE_BINS each, over the extent of the tile grid): summed over the ranks
        they are the input of `apply_balance`."""
        return np.stack([self.world.shard_histogram(a, *self._extent(a), self.BALANCE_BINS).astype(np.uint64) for a in (0, 1)])

    def apply_balance(self, global_hist):
        """New borders from the global histograms (the same on every rank), set for the next step; returns (borders_x, borders_z)."""
        d = self.desc
        cur_x, cur_z = self.world.shard_get_borders(d.tiles_x, d.tiles_z)
        nx = self.world.L.shard_balance_borders(global_hist[0], *self._extent(0), d.tiles_x, cur_x, d.ghost_margin)
        nz = self.world.L.shard_balance_borders(global_hist[1], *self._extent(1), d.tiles_z, cur_z, d.ghost_margin)
        if getattr(self, "exact", False):     # exact seam: tiles stay wider than two margins; a proposal that would not is not taken (the same decision on every rank)
            if len(nx) > 1 and np.any(np.diff(nx) <= 2.0 * d.ghost_margin): nx = cur_x
            if len(nz) > 1 and np.any(np.diff(nz) <= 2.0 * d.ghost_margin): nz = cur_z
        self.world.shard_set_borders(nx, nz)
        return nx, nz

    def rebalance(self):
        """One rebalancing round of a multi-process run (between two steps, on every rank): all-reduce the histograms over the process
        group — control plane, a few KB every K steps — and move the borders."""
        if self.dist is None:
            raise RuntimeError("virtual ranks of one process rebalance together: sharding.rebalance_local(ranks)")
        if self.transport == "rccl":     # the library does the whole round: histograms, ONE ncclAllReduce on the world's stream, new borders
            self.world.shard_rebalance(self.BALANCE_BINS)
            return self.world.shard_get_borders(self.desc.tiles_x, self.desc.tiles_z)   # (still the old ones: the new ones are in force after the next step)
        import torch
        h = torch.from_numpy(self.histograms().astype(np.int64))
        if self.dist.get_backend() == "nccl":
            h = h.cuda()
        self.dist.all_reduce(h)
        return self.apply_balance(h.cpu().numpy().astype(np.uint64))

    def owned_states(self):
        ents = np.sort(self.world.shard_owned_entities())
        return ents, self.world.get_body_states(ents)


def step_local(ranks, settings, dt):
    """Virtual ranks in one process: every tile steps, then the messages are handed over — what R processes do, sequentially."""
    for r in ranks:
        r.world.step_fixed(settings, dt, 1)
    mail = [r.outgoing() for r in ranks]
    for r in ranks:
        for peer in r.neighbours:
            r.world.shard_import(mail[peer][r.rank])
    with np.errstate(over="ignore"):
        total = np.sum([r.world.shard_axis_sums() for r in ranks], axis=0, dtype=np.uint64)   # (wrap-around addition: S1 is two's complement)
    for r in ranks:
        r.world.shard_set_axis_sums(total)


def step_local_exact(ranks, settings, dt):
    """Virtual ranks with the EXACT seam (include/mi_shard.h): every sweep of the step ends with an exchange between the tiles, so the tiles have to
    step side by side — one thread per virtual rank (the library releases the GIL while it steps), meeting at a barrier inside the per-sweep
    callback: all export, barrier, all import, barrier.  What R processes do over their transport, in one process."""
    import threading
    n = len(ranks)
    barrier = threading.Barrier(n)
    mail = [None] * n
    errors = []

    def make_exchange(r):
        def exchange(sweep):
            try:
                mail[r.rank] = {peer: r.world.shard_export_sweep(slot) for slot, peer in enumerate(r.neighbours)}
                barrier.wait()
                for peer in r.neighbours:
                    r.world.shard_import_sweep(mail[peer][r.rank])
                barrier.wait()
            except threading.BrokenBarrierError:
                return -1
            except BaseException:
                barrier.abort()
                raise
        return exchange

    def run(r):
        try:
            r.world.shard_set_exact_seam(True, make_exchange(r))   # (the mode stays; only the callback — it closes over this step's barrier — is new)
            r.world.step_fixed(settings, dt, 1)
        except BaseException as e:
            errors.append((r.rank, e, getattr(r.world, "_sweep_error", None)))
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in ranks]
    for t in threads: t.start()
    for t in threads: t.join()
    if errors:
        raise RuntimeError(f"exact-seam step failed: {errors}")
    mail = [r.outgoing() for r in ranks]
    for r in ranks:
        for peer in r.neighbours:
            r.world.shard_import(mail[peer][r.rank])
    with np.errstate(over="ignore"):
        total = np.sum([r.world.shard_axis_sums() for r in ranks], axis=0, dtype=np.uint64)
    for r in ranks:
        r.world.shard_set_axis_sums(total)


def rebalance_local(ranks):
    """Virtual ranks: what `ShardedWorld.rebalance` does over a process group."""
    total = sum(r.histograms() for r in ranks)
    return [r.apply_balance(total) for r in ranks][0]


def gather_owned(ranks, num_bodies_entities):
    """(entity -> 13-float state) over all ranks; asserts that the owned sets partition the bodies."""
    seen = {}
    for r in ranks:
        ents, st = r.owned_states()
        for e, s in zip(ents, st):
            assert int(e) not in seen, f"entity {e} owned twice"
            seen[int(e)] = s
    assert len(seen) == num_bodies_entities, f"{len(seen)} of {num_bodies_entities} bodies owned"
    return np.stack([seen[k] for k in sorted(seen)])
