"""Several batched worlds of DIFFERENT scenarios behind one VecStageWorld-like surface, for curricula that mix
scenarios under one policy (e.g. Stage-2 worlds + circle worlds).  The envs step one after another on the same
stream; the per-robot fields the learner reads are concatenated into preallocated tensors after every tick."""
import torch

_FIELDS = ("scan_ring", "ring_head", "local_goal", "speed", "reward", "done", "result", "live", "fresh", "first_result")


class ConcatEnv:
    def __init__(self, envs):
        self.envs = list(envs)
        self.device = self.envs[0].device
        self.N = sum(e.N for e in self.envs)
        self.bounds = []
        lo = 0
        for e in self.envs:
            self.bounds.append((lo, lo + e.N))
            lo += e.N
        for k in _FIELDS:
            ref = getattr(self.envs[0], k)
            setattr(self, k, torch.empty((self.N,) + tuple(ref.shape[1:]), dtype=ref.dtype, device=self.device))
        self._ar = torch.arange(self.N, device=self.device)
        self._order = torch.arange(self.scan_ring.shape[1], device=self.device).view(1, -1)
        # VecStageWorld keeps RAW ranges in its ring; a part may say otherwise (ring_is_raw = False: normalised frames)
        raw = {bool(getattr(e, "ring_is_raw", True)) for e in self.envs}
        if len(raw) != 1:
            raise ValueError("ConcatEnv: the parts disagree on what their rings hold")
        self.ring_is_raw = raw.pop()
        self._gather()

    # the scans are kept the way the parts keep them: as rings of raw ranges (see VecStageWorld.obs); observations are
    # formed by the library's own x / 6 - 0.5 (policy_ops.normalize_scans), never by torch's scalar division
    @property
    def obs(self):
        """f32[N,F,B] x / 6 - 0.5 in deque order (oldest frame first), gathered from the concatenated rings."""
        from .policy_ops import normalize_scans
        F = self.scan_ring.shape[1]
        slots = (self.ring_head.long().view(-1, 1) + 1 + self._order) % F
        stacks = self.scan_ring[self._ar.view(-1, 1), slots]
        return normalize_scans(stacks) if self.ring_is_raw else stacks

    def policy_obs(self):
        from .policy_ops import RingHead
        return self.scan_ring, RingHead(self.ring_head, raw=self.ring_is_raw)

    def newest_frame(self):
        from .policy_ops import normalize_scans
        rows = self.scan_ring[self._ar, self.ring_head.long()]
        return normalize_scans(rows) if self.ring_is_raw else rows

    def _gather(self):
        for k in _FIELDS:
            out = getattr(self, k)
            for e, (lo, hi) in zip(self.envs, self.bounds):
                out[lo:hi].copy_(getattr(e, k))

    def reset(self):
        for e in self.envs:
            e.reset()
        self._gather()
        return self

    def step(self, actions):
        for e, (lo, hi) in zip(self.envs, self.bounds):
            e.step(actions[lo:hi].contiguous())
        self._gather()
        return self

    def check(self):
        """mrca_check of every part (raises on a device-side failure flagged since the last check)."""
        for e in self.envs:
            e.check()

    def enable_timing(self, on=True):
        for e in self.envs:
            e.enable_timing(on)

    def close(self):
        for e in self.envs:
            e.close()
