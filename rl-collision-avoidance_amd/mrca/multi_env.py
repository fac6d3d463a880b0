"""Several batched worlds of DIFFERENT scenarios behind one VecStageWorld-like surface, for curricula that mix
scenarios under one policy (e.g. Stage-2 worlds + circle worlds).  The envs step one after another on the same
stream; the per-robot fields the learner reads are concatenated into preallocated tensors after every tick."""
import torch

_FIELDS = ("obs_ring", "obs_head", "local_goal", "speed", "reward", "done", "result", "live", "fresh", "first_result")


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
        self._order = torch.arange(self.obs_ring.shape[1], device=self.device).view(1, -1)
        self._gather()

    # the observation stacks are kept the way the parts keep them: as rings (see VecStageWorld.obs)
    @property
    def obs(self):
        """f32[N,F,B] in deque order (oldest frame first), gathered from the concatenated rings."""
        F = self.obs_ring.shape[1]
        slots = (self.obs_head.long().view(-1, 1) + 1 + self._order) % F
        return self.obs_ring[self._ar.view(-1, 1), slots]

    def policy_obs(self):
        return self.obs_ring, self.obs_head

    def newest_frame(self):
        return self.obs_ring[self._ar, self.obs_head.long()]

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
