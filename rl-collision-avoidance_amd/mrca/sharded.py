"""One world on several GPUs (a single circle of 50 000 robots).

Every rank holds the WHOLE world and owns a contiguous slice of its robots.  Per tick there is exactly one
exchange step: an all-gather of the commands ``act[N,2]`` (400 kB at 50 000 robots).  Each rank then advances all
robots -- the collision pass is sequential in robot order and needs every provisional pose, and identical arithmetic
on identical inputs keeps the replicas bit-identical, so no pose ever has to travel -- and casts the 512-beam lidar only
for its own slice (``mrca_step_slice``).

The replicated move phase is what bounds the speed-up (Amdahl; DESIGN.md 5.4 / 7 carry the current figures): at 50 000
robots it was 43.5 us of a 144 us tick on one MI355X in round 4 -- at most 2.6x on 8 GPUs, 2.2x projected from one rank's
measured share (``tools/bigworld_bench.py --shards 8``, profiles/r04_u_bigworld_shards8.jsonl).  Sharding the move phase too
would need the transitive closure of every slice's lower-indexed neighbours inside the ordered collision pass -- not built.
"""
import torch


def slice_bounds(n, size, rank):
    """Contiguous slices of ceil(n / size) robots; trailing ranks may be short or empty."""
    per = -(-n // size)
    lo = min(rank * per, n)
    return lo, min(lo + per, n), per


class ShardedWorld:
    def __init__(self, env, dist=None):
        self.env, self.dist = env, dist
        self.size = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.size > 1 else 0
        self.lo, self.hi, self.per = slice_bounds(env.N, self.size, self.rank)
        self._pad = torch.zeros(self.per, 2, dtype=torch.float32, device=env.device)
        self._all = torch.zeros(self.size * self.per, 2, dtype=torch.float32, device=env.device)

    @property
    def count(self):
        return self.hi - self.lo

    def local(self, field):
        """This rank's rows of a per-robot field of the replicated world."""
        return getattr(self.env, field)[self.lo:self.hi]

    def policy_obs(self):
        """-> (ring rows, head rows) of this rank's robots: what the policy's front end reads in place.  (The deque-ordered
        ``obs`` of the binding materialises the stacks of ALL N robots of the replicated world -- ~300 MB read + written
        per tick at 50 000 robots for rows only 1 / size of which were ray-cast -- so the sharded surface hands out the
        ring and never touches it.)"""
        ring, head = self.env.policy_obs()
        return ring[self.lo:self.hi], head[self.lo:self.hi]

    def obs(self):
        """f32[count,F,B] this rank's stacks in deque order, gathered from the ring rows of the slice only."""
        from .policy_ops import normalize_scans, unwrap_head
        ring, head = self.policy_obs()
        head, raw = unwrap_head(head)
        F = ring.shape[1]
        slots = (head.long().view(-1, 1) + 1 + torch.arange(F, device=ring.device).view(1, -1)) % F
        stacks = ring[torch.arange(ring.shape[0], device=ring.device).view(-1, 1), slots]
        return normalize_scans(stacks) if raw else stacks      # (a ring of raw scans: stage_world1.py:140 applied on read)

    def check(self):
        self.env.check()

    def reset(self):
        self.env.reset()        # replicated: every rank resets the whole world (same seeds -> same poses)
        return self.policy_obs(), self.local("local_goal"), self.local("speed")

    def gather_actions(self, local_actions):
        """The exchange step: act[count,2] of every rank -> act[N,2] on every rank."""
        if self.size == 1:
            return local_actions.contiguous()
        self._pad.zero_()
        self._pad[: self.count].copy_(local_actions)
        self.dist.all_gather_into_tensor(self._all, self._pad)
        return self._all[: self.env.N].contiguous()   # short slices sit at the end: the first N rows are the world

    def step(self, local_actions):
        full = self.gather_actions(local_actions)
        self.env.step(full, ray_slice=(self.lo, self.count))
        return (self.policy_obs(), self.local("local_goal"), self.local("speed"), self.local("reward"),
                self.local("done"), self.local("result"))
