"""One process per GPU; each rank owns one spatial tile (x-slab) of the scene.

Sharding scheme (round 1):
  * the global cfg3 pen is split into `world_size` x-slabs of equal lattice width; rank r owns the boxes whose lattice
    column lies in its slab (static ownership: no migration yet — valid while boxes stay within the ghost margin of
    their slab, which holds for a settling pile);
  * each rank also simulates GHOST copies of the `ghost_cols` nearest columns of each neighbour, so contacts across a
    seam are generated and solved on both sides;
  * after every step the owners' new states of those boundary boxes are exchanged (one all-to-all-v over RCCL/xGMI with
    only neighbour slots non-empty; 13 floats = 52 B per boundary box) and overwrite the ghost copies.  This is the
    once-per-step seam exchange SURVEY §8(e) calls the Jacobi-across-the-seam variant: interiors are exact, the seam is
    an approximation that does NOT reproduce the 1-GPU trajectory bit-for-bit (tests bound the difference).  The
    per-iteration exchange that would is the next step.
  * global integer counts use one all-reduce of 5 int64.

The world factory is injected: the product passes the HIP library (device buffers go straight into the collective,
no host round trip); tests pass the CPU oracle with the gloo backend (host buffers).
"""
import numpy as np

from . import scenes

STATE_FLOATS = 13


class ShardedWorld:
    def __init__(self, world_factory, rank, world_size, dist, tile=(128, 16, 128), iterations=20, ghost_cols=2, seed=3, device_exchange=None):
        self.rank, self.world_size, self.dist = rank, world_size, dist
        nx, ny, nz = tile
        self.scene, self.info = scenes.obb_pile_tile(rank, world_size, nx, ny, nz, ghost_cols=ghost_cols, seed=seed, solver_iterations=iterations)
        self.world = self.scene.populate(world_factory())
        self.bodies_per_rank = self.info["owned"]
        self.dt = self.scene.dt
        self.device_exchange = (dist is not None and dist.get_backend() == "nccl") if device_exchange is None else device_exchange
        self.sharding_note = (f"{world_size} x-slab tiles, {ghost_cols} ghost columns per seam, per-step RCCL all-to-all-v of boundary states"
                              if world_size > 1 else "single GPU, whole scene")
        self._setup_exchange()

    def settings(self):
        return self.scene.settings()

    # ------------------------------------------------------------------ exchange plumbing
    def _setup_exchange(self):
        if self.world_size == 1:
            return
        import torch
        r, n = self.rank, self.world_size
        info = self.info
        self.send_counts = [0] * n
        self.recv_counts = [0] * n
        send_ids, recv_ids = [], []
        # all_to_all_single lays peers out in rank order: left neighbour (r-1) precedes right neighbour (r+1)
        if r > 0:
            self.send_counts[r - 1] = len(info["send_left"]); self.recv_counts[r - 1] = len(info["ghost_left"])
            send_ids.append(info["send_left"]); recv_ids.append(info["ghost_left"])
        if r < n - 1:
            self.send_counts[r + 1] = len(info["send_right"]); self.recv_counts[r + 1] = len(info["ghost_right"])
            send_ids.append(info["send_right"]); recv_ids.append(info["ghost_right"])
        self.send_entities = np.concatenate(send_ids) if send_ids else np.zeros(0, np.uint32)
        self.recv_entities = np.concatenate(recv_ids) if recv_ids else np.zeros(0, np.uint32)
        ns, nr = len(self.send_entities), len(self.recv_entities)
        if self.device_exchange:
            dev = torch.device("cuda", torch.cuda.current_device())
            self.send_body_ids = torch.from_numpy(self.world.entities_to_bodies(self.send_entities).astype(np.int32)).to(dev)
            self.recv_body_ids = torch.from_numpy(self.world.entities_to_bodies(self.recv_entities).astype(np.int32)).to(dev)
            self.send_buf = torch.zeros(max(ns, 1) * STATE_FLOATS, dtype=torch.float32, device=dev)
            self.recv_buf = torch.zeros(max(nr, 1) * STATE_FLOATS, dtype=torch.float32, device=dev)
            # the collective runs ON the world's stream: gather -> all-to-all -> scatter -> next step are ordered on the device
            self.stream = torch.cuda.ExternalStream(self.world.stream_ptr(), device=dev)
            torch.cuda.synchronize()   # the id tensors were filled on torch's default stream
        else:
            self.send_buf = torch.zeros(max(ns, 1) * STATE_FLOATS, dtype=torch.float32)
            self.recv_buf = torch.zeros(max(nr, 1) * STATE_FLOATS, dtype=torch.float32)

    def exchange_ghosts(self):
        """Owners' boundary states -> neighbours' ghost copies (all-to-all-v; only neighbour slots are non-empty)."""
        if self.world_size == 1:
            return
        import torch
        ns, nr = len(self.send_entities), len(self.recv_entities)
        in_splits = [c * STATE_FLOATS for c in self.send_counts]
        out_splits = [c * STATE_FLOATS for c in self.recv_counts]
        if self.device_exchange:
            # no host synchronisation: the copy kernels and the collective are enqueued on the world's own stream (RCCL's
            # internal stream is fenced against the current stream by torch on both sides of the call)
            with torch.cuda.stream(self.stream):
                if ns:
                    self.world.get_body_states_device_async(ns, self.send_body_ids.data_ptr(), self.send_buf.data_ptr())
                self.dist.all_to_all_single(self.recv_buf[: nr * STATE_FLOATS], self.send_buf[: ns * STATE_FLOATS], out_splits, in_splits)
                if nr:
                    self.world.set_body_states_device_async(nr, self.recv_body_ids.data_ptr(), self.recv_buf.data_ptr())
        else:
            if ns:
                self.send_buf[: ns * STATE_FLOATS] = torch.from_numpy(self.world.get_body_states(self.send_entities).reshape(-1))
            _all_to_all_v_host(self.dist, self.rank, self.world_size, self.recv_buf[: nr * STATE_FLOATS], self.send_buf[: ns * STATE_FLOATS],
                               out_splits, in_splits)
            if nr:
                self.world.set_body_states(self.recv_entities, self.recv_buf[: nr * STATE_FLOATS].numpy().reshape(nr, STATE_FLOATS))

    def step(self, settings, dt):
        self.world.step_fixed(settings, dt, 1)
        self.exchange_ghosts()

    def owned_states(self):
        return self.world.get_body_states(np.arange(self.info["owned"], dtype=np.uint32))

    def total_counts(self):
        c = self.world.counts()
        if self.dist is None or self.world_size == 1:
            return c
        import torch
        keys = ("num_rigid_bodies", "num_colliders", "num_broadphase_overlaps", "num_collisions", "num_contacts")
        dev = "cuda" if self.dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([c[k] for k in keys], dtype=torch.int64, device=dev)
        self.dist.all_reduce(t)
        out = dict(c)
        out.update({k: int(v) for k, v in zip(keys, t.tolist())})
        return out


def _all_to_all_v_host(dist, rank, world_size, out, inp, out_splits, in_splits):
    """all_to_all_single where the backend has it (NCCL); paired isend/irecv otherwise (gloo lacks all-to-all-v on CPU)."""
    try:
        dist.all_to_all_single(out, inp, out_splits, in_splits)
        return
    except (RuntimeError, NotImplementedError):
        pass
    ops, o_off, i_off = [], 0, 0
    for peer in range(world_size):
        if in_splits[peer]:
            ops.append(dist.P2POp(dist.isend, inp[i_off: i_off + in_splits[peer]], peer))
        if out_splits[peer]:
            ops.append(dist.P2POp(dist.irecv, out[o_off: o_off + out_splits[peer]], peer))
        i_off += in_splits[peer]; o_off += out_splits[peer]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
