"""One process per GPU; each rank owns one spatial tile of the scene.

Round-1 state: tiles are x-slabs of the global pen.  Each rank steps its tile with the single-GPU
pipeline; there is no data-path collective yet because tiles are separated by pen walls (every tile
is a closed pen), i.e. the job is N independent shards ("weak" scaling).  The ghost-region exchange
that lets bodies interact across a tile seam is the next row of the scope table.
The world factory is injected so the same logic runs over the HIP library (bench, product) and — in
tests only — over the CPU oracle with the gloo backend.
"""
from . import scenes


class ShardedWorld:
    def __init__(self, world_factory, rank, world_size, dist, tile=(128, 16, 128), iterations=20, scene_fn=None):
        self.rank, self.world_size, self.dist = rank, world_size, dist
        make = scene_fn or scenes.obb_pile
        self.scene = make(*tile, seed=3 + rank, solver_iterations=iterations)
        self.world = self.scene.populate(world_factory())
        self.bodies_per_rank = self.scene.num_bodies
        self.dt = self.scene.dt
        self.sharding_note = ("1 tile per GPU, tiles are independent closed pens (no seam exchange yet)" if world_size > 1
                              else "single GPU, whole scene")

    def settings(self):
        return self.scene.settings()

    def step(self, settings, dt):
        self.world.step_fixed(settings, dt, 1)

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
