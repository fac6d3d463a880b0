"""On-device PPO pieces: action sampling, rollout buffer, GAE, clipped-surrogate update.

Same names and argument meaning as the reference's ``model/ppo.py`` so the call sites read alike,
but every tensor lives on the GPU (no pickling, no host round trips) and the update optionally
all-reduces gradients over RCCL (one flat bucket) for data parallelism over robots.
"""
import math

import torch
import torch.nn.functional as F

from . import vec_env


_bound_cache = {}


def _bounds(action_bound, device, dtype):
    """The clip bounds as device tensors, uploaded once (not one host-to-device copy per tick)."""
    key = (tuple(map(tuple, action_bound)), str(device), dtype)
    if key not in _bound_cache:
        _bound_cache[key] = (torch.tensor(action_bound[0], device=device, dtype=dtype),
                             torch.tensor(action_bound[1], device=device, dtype=dtype))
    return _bound_cache[key]


# ---------------------------------------------------------------------------------------------
def generate_action(policy, obs, goal, speed, action_bound, generator=None, autocast_dtype=None, fused=False,
                    obs_head=None, noise=None):
    """model/ppo.py:57-82: sample a ~ N(mean, std); the UNclipped action and its logprob are what
    the buffer stores, the clipped one drives the robot.  ``fused=True`` evaluates the policy through its fp32
    rollout path (HIP conv front end + batched GEMMs, net.CNNPolicy.mean_value_fused; same numbers to 1e-5);
    ``autocast_dtype=torch.bfloat16`` runs the stock towers on the bf16 MFMA path (opt-in).  ``obs_head`` (fused
    only): ``obs`` is the env's frame ring, see ``policy_input``.  ``noise`` f32[N,2]: standard normal draws made by the
    caller (a rollout replayed as a hipGraph of several ticks draws all of them in one launch) instead of a ``randn``
    here."""
    from .net import gaussian_logprob
    if obs_head is not None and not fused:
        raise ValueError("a frame ring (obs_head) can only be read by the fused policy path")
    if fused and hasattr(policy, "act_fused"):
        lo, hi = _bounds(action_bound, goal.device, torch.float32)
        if noise is None:
            noise = torch.randn((goal.shape[0], 2), device=goal.device, dtype=torch.float32, generator=generator)
        v, a, logprob, scaled, _mean = policy.act_fused(obs, goal, speed, noise, lo, hi, head=obs_head)
        return v, a, logprob, scaled
    with torch.no_grad():
        if fused:
            mean, v = policy.mean_value_fused(obs, goal, speed, head=obs_head)
        elif autocast_dtype is not None:
            with torch.autocast(obs.device.type, dtype=autocast_dtype):
                mean, v = policy.mean_value(obs, goal, speed)
            mean, v = mean.float(), v.float()
        else:
            mean, v = policy.mean_value(obs, goal, speed)
        logstd = policy.logstd.expand_as(mean)
        if noise is None:
            noise = torch.randn(mean.shape, device=mean.device, dtype=mean.dtype, generator=generator)
        a = mean + torch.exp(logstd) * noise
        logprob = gaussian_logprob(a, mean, logstd)
        lo, hi = _bounds(action_bound, a.device, a.dtype)
        scaled = torch.minimum(torch.maximum(a, lo), hi)
    return v, a, logprob, scaled


def policy_input(env, fused):
    """-> (obs, obs_head) to hand to generate_action / generate_action_no_sampling: the env's frame ring when the
    fused policy path can read it in place (VecStageWorld.policy_obs), the stacks in deque order otherwise."""
    if fused and hasattr(env, "policy_obs"):
        return env.policy_obs()
    return env.obs, None


def generate_action_no_sampling(policy, obs, goal, speed, action_bound, fused=False, obs_head=None):
    """model/ppo.py:84-107: deterministic mean action (circle_test.py:58-59)."""
    if fused and hasattr(policy, "act_fused"):
        lo, hi = _bounds(action_bound, goal.device, torch.float32)
        _v, _a, _lp, scaled, mean = policy.act_fused(obs, goal, speed, None, lo, hi, head=obs_head)
        return mean, scaled
    with torch.no_grad():
        mean, _v = policy.mean_value_fused(obs, goal, speed, head=obs_head) if fused else policy.mean_value(obs, goal, speed)
        lo, hi = _bounds(action_bound, mean.device, mean.dtype)
        scaled = torch.minimum(torch.maximum(mean, lo), hi)
    return mean, scaled


# ---------------------------------------------------------------------------------------------
class FrameRows:
    """The observation stacks of a rollout as a lazily gathered [T*N, frames, beams] tensor: ``rows[index]`` builds
    the minibatch from the one-frame-per-tick store (RolloutBuffer, ``single_frame=True``).  Supports the two ways
    the update addresses its memory: integer index tensors and a boolean keep-mask (Stage-2 filtering)."""

    def __init__(self, frames, fidx, rowmap=None, lazy=False):
        self.frames, self.fidx, self.rowmap = frames, fidx, rowmap      # [T+2,N,B], [T,N,F] int64, optional [n'] -> t*N+i
        self.N = fidx.shape[1]
        # lazy: an integer index returns a policy_ops.FrameTable -- the frame store itself + the rows of the minibatch's stacks --
        # instead of a gathered [mb, F, B] copy (set by the update when the policy's front end reads through such a table)
        self.lazy = lazy

    @property
    def shape(self):
        n = self.fidx.shape[0] * self.N if self.rowmap is None else self.rowmap.numel()
        return (n, self.fidx.shape[2], self.frames.shape[2])

    def __getitem__(self, index):
        if index.dtype == torch.bool:
            base = torch.arange(self.fidx.shape[0] * self.N, device=index.device) if self.rowmap is None else self.rowmap
            return FrameRows(self.frames, self.fidx, base[index], self.lazy)
        flat = index if self.rowmap is None else self.rowmap[index]
        t, i = flat // self.N, flat % self.N
        fi = self.fidx[t, i]                                   # [mb, F]
        if self.lazy:
            from . import policy_ops
            rows = (fi * self.N + i.unsqueeze(1)).to(torch.int32)
            return policy_ops.FrameTable(self.frames.view(-1, self.frames.shape[2]), rows.contiguous())
        return self.frames[fi, i.unsqueeze(1)]                 # [mb, F, B]

    def materialise(self):
        T, N, Fr = self.fidx.shape
        i = torch.arange(N, device=self.fidx.device).view(1, N, 1).expand(T, N, Fr)
        return self.frames[self.fidx, i]                       # [T, N, F, B]


class RolloutBuffer:
    """Preallocated [T, N, ...] device tensors replacing the list-of-tuples ``buff`` +
    ``transform_buffer`` (ppo_stage1.py:102-103, model/ppo.py:22-54).

    ``single_frame=True`` stores ONE lidar frame per tick instead of the whole 3-frame stack (the stack at tick t
    shares two frames with the stack at t-1: SURVEY 8a a13): ``frames[T+2,N,B]`` (rows 0, 1 = the two older frames of
    the first tick's stack) plus, per tick and robot, which rows make up its stack (``fidx``; a robot that restarted
    has all three pointing at its fresh scan, ppo_stage1.py:59-60).  A third of the write traffic per tick and of the
    3.2 GB the stacks of a 128 x 4096 horizon take; minibatches are gathered through ``FrameRows``."""

    def __init__(self, horizon, num_env, frames, beams, device, act_size=2, single_frame=False):
        T, N = horizon, num_env
        f32 = dict(dtype=torch.float32, device=device)
        self.single_frame = single_frame
        if single_frame:
            self.frames = torch.empty(T + frames - 1, N, beams, **f32)
            self.fidx = torch.zeros(T, N, frames, dtype=torch.int64, device=device)
            self._cur = torch.zeros(N, frames, dtype=torch.int64, device=device)
            self._first = torch.arange(frames, dtype=torch.int64, device=device).view(1, frames)
            self._two = torch.full((1,), frames - 1, dtype=torch.int64, device=device)
            self._rows01 = torch.arange(frames - 1, dtype=torch.int64, device=device)
            self.obs = None
        else:
            self.obs = torch.empty(T, N, frames, beams, **f32)
        self.goal = torch.empty(T, N, 2, **f32)
        self.speed = torch.empty(T, N, 2, **f32)
        self.action = torch.empty(T, N, act_size, **f32)
        self.reward = torch.empty(T, N, **f32)
        self.done = torch.empty(T, N, dtype=torch.uint8, device=device)
        self.logprob = torch.empty(T, N, 1, **f32)
        self.value = torch.empty(T, N, **f32)
        self.horizon, self.num_env, self.nframes = T, N, frames
        self._env, self._rows, self._ticket = None, None, None

    # ---- the per-tick stores as two launches of the env's library (csrc/mrca_rollout_store.hip) instead of ~20 of PyTorch's
    def bind_env(self, env):
        """Let ``env`` (a VecStageWorld on this buffer's device, same N / frames / beams) write the rows itself:
        ``store_state_at`` then takes the newest frame, the goal and the speed from the env's arena, ``store_outcome_at``
        its reward and done flags AND moves the device-side row counter on.  One-frame buffers only."""
        if not self.single_frame:
            raise ValueError("bind_env: only the one-frame-per-tick buffer is written by the library")
        self._rows = env.rollout_rows(self)
        self._env = env
        self._ticket = torch.zeros(1, dtype=torch.int32, device=self.goal.device)

    @property
    def env_bound(self):
        return self._env is not None

    # ---- observation stacks
    def begin_horizon(self, obs):
        """single_frame: the older frames of the stack the first tick of a horizon will see (call before tick 0)."""
        if self.single_frame:
            self.frames[: self.nframes - 1].copy_(obs[:, : self.nframes - 1].transpose(0, 1))
            self._cur.copy_(self._first.expand_as(self._cur))

    def _store_obs_at(self, t_idx, obs, fresh, newest=None):
        if not self.single_frame:
            self.obs.index_copy_(0, t_idx, obs.unsqueeze(0))
            return
        row = t_idx + self._two                                         # the newest frame of tick t lives in row t+F-1
        self.frames.index_copy_(0, row, (obs[:, -1] if newest is None else newest).unsqueeze(0))
        shifted = torch.cat((self._cur[:, 1:], row.view(1, 1).expand(self.num_env, 1)), dim=1)
        # tick 0 of a horizon: rows 0..F-1 whatever the flags say (begin_horizon copied the real older frames)
        at_start = (t_idx == 0).view(1, 1)
        restarted = fresh.bool().view(-1, 1) & ~at_start
        self._cur.copy_(torch.where(at_start, self._first, torch.where(restarted, row.view(1, 1), shifted)))
        self.fidx.index_copy_(0, t_idx, self._cur.unsqueeze(0))

    def obs_rows(self):
        """[T*N, F, B] view of the stored stacks, however they are stored."""
        if self.single_frame:
            return FrameRows(self.frames, self.fidx)
        T, N = self.horizon, self.num_env
        return self.obs.reshape(T * N, self.nframes, -1)

    def store_state(self, t, obs, goal, speed, action, logprob, value, fresh=None, newest=None):
        """``newest`` f32[N,B] (single_frame only): the frame the last tick appended, for callers that keep the stacks
        as a ring and would rather not materialise ``obs`` (VecStageWorld.newest_frame()); ``obs`` may then be None."""
        if self.single_frame:
            self._store_obs_at(torch.tensor([t], dtype=torch.int64, device=goal.device), obs, fresh, newest)
        else:
            self.obs[t].copy_(obs)
        self.goal[t].copy_(goal)
        self.speed[t].copy_(speed)
        self.action[t].copy_(action)
        self.logprob[t].copy_(logprob.view(-1, 1))
        self.value[t].copy_(value.view(-1))

    def store_outcome(self, t, reward, done):
        self.reward[t].copy_(reward)
        self.done[t].copy_(done)

    # the same two stores with the row given as a DEVICE index tensor (int64[1]): no host value enters the launch, so a
    # whole tick can be captured once as a hipGraph and replayed for every row of the horizon
    def store_state_at(self, t_idx, obs, goal, speed, action, logprob, value, fresh=None, newest=None):
        if self._env is not None:       # (obs / goal / speed / fresh / newest: the bound env's own fields, read by the library)
            self._env.rollout_store_state(self._rows, t_idx, action.contiguous(), logprob.contiguous(), value.contiguous())
            return
        self._store_obs_at(t_idx, obs, fresh, newest)
        self.goal.index_copy_(0, t_idx, goal.unsqueeze(0))
        self.speed.index_copy_(0, t_idx, speed.unsqueeze(0))
        self.action.index_copy_(0, t_idx, action.unsqueeze(0))
        self.logprob.index_copy_(0, t_idx, logprob.view(1, -1, 1))
        self.value.index_copy_(0, t_idx, value.view(1, -1))

    def store_outcome_at(self, t_idx, reward, done):
        """-> True when the row counter ``t_idx`` was moved on by the store itself (a bound env), False when that is left
        to the caller."""
        if self._env is not None:
            self._env.rollout_store_outcome(self._rows, t_idx, self._ticket)
            return True
        self.reward.index_copy_(0, t_idx, reward.unsqueeze(0))
        self.done.index_copy_(0, t_idx, done.unsqueeze(0))
        return False


def generate_train_data(rewards, gamma, values, last_value, dones, lam):
    """model/ppo.py:122-139 on device through the HIP GAE kernel (mrca_gae)."""
    return vec_env.gae(rewards.contiguous(), values.contiguous(), last_value.reshape(-1).contiguous(),
                       dones.contiguous(), gamma, lam)


def get_filter_index(dones):
    """model/utils.py:65-78 on device: transitions whose done flag has been True for >= 2
    consecutive steps, INCLUDING the reference's quirk that the run counter is not reset between
    robots (the sequence is walked robot-major).  dones: [T,N] -> LongTensor of N*j+i."""
    T, N = dones.shape
    seq = dones.t().reshape(-1).bool()            # robot-major walk: i outer, j inner
    prev = torch.cat([seq.new_zeros(1), seq[:-1]])
    hit = (seq & prev).view(N, T)                 # [i, j]
    i, j = torch.nonzero(hit, as_tuple=True)
    return N * j + i


def get_group_terminal(terminal_list, index, refer=(0, 6, 10, 15, 19, 24, 34, 44)):
    """model/utils.py:81-87: True when every robot of `index`'s scenario group has terminated.  (The
    batched env applies the same rule on the device: MRCA_AUTO_GROUP, move_kernel's group ballots.)"""
    import bisect
    r = bisect.bisect(list(refer), index)
    return all(bool(t) for t in terminal_list[refer[r - 1]: refer[r]])


# ---------------------------------------------------------------------------------------------
class FlatGrads:
    """All parameter gradients as views of ONE contiguous bucket: a single RCCL all-reduce of
    2 172 101 floats (8.69 MB) per optimiser step."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=self.params[0].dtype, device=self.params[0].device)
        off = 0
        for p in self.params:
            p.grad = self.flat[off: off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def backward(self, loss):
        """d loss / d parameters straight into the bucket: ``torch.autograd.grad`` and ONE cat launch.  (``loss.backward()``
        into ``.grad`` tensors that are views of the bucket is a zero-fill plus one AccumulateGrad add launch per parameter
        tensor -- 22 adds of ~5 us per minibatch, 3 % of an update, profiles/r05_z_train_kernel_stats.csv.)  The ``.grad``
        views keep showing the bucket, i.e. the new gradients."""
        grads = torch.autograd.grad(loss, self.params, allow_unused=True)
        torch.cat([(torch.zeros_like(p) if g is None else g).reshape(-1) for g, p in zip(grads, self.params)], out=self.flat)

    def all_reduce_mean(self, dist):
        if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat)
            self.flat.div_(dist.get_world_size())


class FlatAdam:
    """``torch.optim.Adam(params, lr)`` (the reference's optimiser, ppo_stage1.py:176: betas 0.9 / 0.999, eps 1e-8, no weight
    decay, no amsgrad) for parameters whose gradients live in a ``FlatGrads`` bucket: the parameters are re-seated as views
    of ONE buffer laid out like the bucket, the two moment estimates are flat as well, and a step is one element-wise pass
    -- on the GPU one launch of csrc/mrca_adam.hip (PyTorch's fused multi-tensor step walks the tensor list on 34
    workgroups: 102 us per step against 61 MB of traffic), on the CPU the same expressions as five torch calls.

    The parts of the ``torch.optim.Optimizer`` surface the learner uses: ``param_groups`` (one group; the KL controller
    and ``--lr`` write its ``"lr"``), ``step()``, ``zero_grad()``, ``state_dict()`` / ``load_state_dict()`` -- both in
    ``torch.optim.Adam``'s own format, so optimiser states saved by either load into the other."""

    def __init__(self, flat_grads, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.grads = flat_grads
        self.params = flat_grads.params
        self.flat = torch.empty_like(flat_grads.flat)
        off = 0
        with torch.no_grad():
            for p in self.params:
                view = self.flat[off: off + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view                       # same values, new storage: the parameter IS this slice from now on
                off += p.numel()
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.steps = 0
        self.param_groups = [{"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": 0, "amsgrad": False,
                              "params": self.params}]

    def _check_seating(self):
        off = 0
        for p in self.params:
            if p.data_ptr() != self.flat.data_ptr() + 4 * off or p.device != self.flat.device:
                raise RuntimeError("FlatAdam: a parameter no longer lives in the optimiser's buffer (the module was moved "
                                   "or re-created after the optimiser was built): build the optimiser last")
            off += p.numel()

    def zero_grad(self, set_to_none=False):
        self.grads.zero()

    @torch.no_grad()
    def step(self):
        g = self.param_groups[0]
        if g.get("weight_decay", 0) or g.get("amsgrad", False):
            raise NotImplementedError("FlatAdam: weight decay / amsgrad are not part of the reference's optimiser")
        self._check_seating()
        self.steps += 1
        (b1, b2), lr, eps, t = g["betas"], g["lr"], g["eps"], self.steps
        p, grad, m, v = self.flat, self.grads.flat, self.exp_avg, self.exp_avg_sq
        if p.is_cuda:
            from . import policy_ops
            policy_ops.adam_step(p, grad, m, v, lr, b1, b2, eps, t)
            return
        # torch/optim/adam.py, _single_tensor_adam, expression for expression
        m.lerp_(grad, 1 - b1)
        v.mul_(b2).addcmul_(grad, grad, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
        p.addcdiv_(m, denom, value=-(lr / (1 - b1 ** t)))

    def state_dict(self):
        state, off = {}, 0
        for i, p in enumerate(self.params):
            sl = slice(off, off + p.numel())
            if self.steps:
                state[i] = {"step": torch.tensor(float(self.steps)), "exp_avg": self.exp_avg[sl].view_as(p).clone(),
                            "exp_avg_sq": self.exp_avg_sq[sl].view_as(p).clone()}
            off += p.numel()
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group.update({"maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                      "decoupled_weight_decay": False, "params": list(range(len(self.params)))})
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.params):
            raise ValueError("FlatAdam.load_state_dict: expected one parameter group of %d tensors" % len(self.params))
        for k in ("lr", "betas", "eps", "weight_decay", "amsgrad"):
            if k in groups[0]:
                self.param_groups[0][k] = tuple(groups[0][k]) if k == "betas" else groups[0][k]
        state, off, steps = sd["state"], 0, set()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for i, p in enumerate(self.params):
            st = state.get(i, state.get(str(i)))
            if st is not None:
                self.exp_avg[off: off + p.numel()].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off: off + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(float(st["step"])))
            off += p.numel()
        if len(steps) > 1:
            raise ValueError(f"FlatAdam.load_state_dict: the tensors are at different steps {sorted(steps)}")
        self.steps = steps.pop() if steps else 0


def _global_mean_std(x, dist):
    """Population mean / std of the advantages over ALL ranks (model/ppo.py:148 uses np.std)."""
    s = torch.stack([x.sum(dtype=torch.float64), (x.double() ** 2).sum(), torch.tensor(float(x.numel()),
                                                                                     dtype=torch.float64,
                                                                                     device=x.device)])
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(s)
    mean = s[0] / s[2]
    var = torch.clamp(s[1] / s[2] - mean * mean, min=0.0)
    return mean.to(x.dtype), torch.sqrt(var).to(x.dtype)


def _world_size(dist):
    return dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1


class KLAdaptiveLR:
    """Opt-in learning-rate controller for the large-batch regime (not in the reference, whose lr is fixed at
    5e-5 for 3072-sample updates): after every epoch the mean approximate KL(old || new) of that epoch's
    minibatches is compared with ``target``; above 2x the lr is divided by ``factor``, below 0.5x multiplied.
    The KL is averaged over ranks so that every rank's optimizer takes the same decision."""

    def __init__(self, target, lr_min=1e-6, lr_max=1e-3, factor=1.5, stop_factor=0.0):
        self.target, self.lr_min, self.lr_max, self.factor = float(target), lr_min, lr_max, factor
        # stop_factor > 0: the rest of an update is abandoned as soon as one minibatch reports KL > stop_factor x target
        # (the trust region is enforced inside the update, not only by next update's learning rate)
        self.stop_factor = float(stop_factor)
        self.last_kl = None
        self.stopped_early = 0

    def should_stop(self, kl_minibatch, dist):
        if self.stop_factor <= 0:
            return False
        k = kl_minibatch.detach().double().reshape(1)
        if _world_size(dist) > 1:
            dist.all_reduce(k)
            k = k / _world_size(dist)
        stop = float(k) > self.stop_factor * self.target
        self.stopped_early += int(stop)
        return stop

    def step(self, optimizer, kl_sum, count, dist):
        s = torch.stack([kl_sum.double(), torch.as_tensor(float(count), dtype=torch.float64, device=kl_sum.device)])
        if _world_size(dist) > 1:
            dist.all_reduce(s)
        kl = float(s[0] / s[1].clamp(min=1.0))
        self.last_kl = kl
        for g in optimizer.param_groups:
            if kl > 2.0 * self.target:
                g["lr"] = max(self.lr_min, g["lr"] / self.factor)
            elif kl < 0.5 * self.target:
                g["lr"] = min(self.lr_max, g["lr"] * self.factor)
        return kl


_side_streams = {}


class _ObsPrefetch:
    """The observation stacks of minibatch k + 1 gathered on a side stream while minibatch k computes.  The gather is the
    one memory-bound step of a minibatch (three 2 kB frame rows per sample: 100 MB read and written for 16 384 rows, 90 us,
    profiles/r05_z_train_kernel_stats.csv) between MFMA-bound ones, and depends on nothing the minibatch before it
    produces.  Same kernels, same values: only where they run.  On the CPU: a plain ``obss[index]``."""

    def __init__(self, obss, batches):
        self.obss, self.batches = obss, batches
        dev = batches[0].device if len(batches) else None
        self.on = dev is not None and dev.type == "cuda"
        self.ready = {}
        if self.on:
            self.main = torch.cuda.current_stream(dev)
            key = (dev.index, )
            if key not in _side_streams:
                _side_streams[key] = torch.cuda.Stream(device=dev)
            self.side = _side_streams[key]
            self._issue(0)

    def _issue(self, k):
        if k >= len(self.batches) or k in self.ready:
            return
        # behind everything the main stream holds so far (the permutation, the buffer, the minibatch before the one that
        # is about to start): the gather runs NEXT TO that minibatch, and the host cannot queue gathers without bound
        self.side.wait_stream(self.main)
        with torch.cuda.stream(self.side):
            self.ready[k] = self.obss[self.batches[k]]

    def take(self, k):
        if not self.on:
            return self.obss[self.batches[k]]
        self._issue(k)                               # (only if a caller skipped ahead)
        x = self.ready.pop(k)
        self.main.wait_stream(self.side)
        x.record_stream(self.main)                   # allocated on the side stream, consumed on the main one
        self._issue(k + 1)
        return x

    def close(self):
        """Leaving the minibatch loop (normally or on a KL early stop): a gather issued ahead and never taken still reads the
        permutation slices on the side stream -- the main stream waits for it before that memory can be handed out again,
        and the prefetched stacks are dropped."""
        if self.on and self.ready:
            self.main.wait_stream(self.side)
            for x in self.ready.values():
                x.record_stream(self.main)
            self.ready.clear()


def _ppo_epochs(policy, optimizer, batch_size, flat, epoch, coeff_entropy, clip_value, drop_last, value_coef,
                index_batches, dist, flat_grads, log, autocast_dtype=None, kl_ctl=None, max_grad_norm=0.0,
                logstd_min=None):
    obss, goals, speeds, actions, logprobs, targets, advs = flat
    n = advs.shape[0]
    multi = _world_size(dist) > 1
    if multi and flat_grads is None:
        raise ValueError("ppo update with a multi-rank process group needs flat_grads (ppo.FlatGrads): without it "
                         "the replicas would train unsynchronised")
    # the loss tail (ratio, clipped surrogate, value loss, entropy and their gradients) as ONE launch where the policy's
    # front end already runs through the HIP kernels (policy_ops.ppo_loss); the stock expression otherwise
    fused_loss = bool(getattr(policy, "fused_train", False)) and advs.is_cuda and autocast_dtype is None
    if isinstance(obss, FrameRows):
        # the fused front end reads the minibatch's stacks out of the frame store through a row table (policy_ops.FrameTable);
        # every other path gets the gathered tensor
        obss.lazy = (fused_loss and obss.frames.is_cuda and obss.frames.dtype == torch.float32 and obss.frames.is_contiguous() and
                     obss.frames.shape[2] == 512 and obss.fidx.shape[2] == 3 and obss.frames.numel() < 2 ** 31 * 512)
    small = (goals, speeds, actions, logprobs, targets, advs)
    for _ in range(epoch):
        if index_batches is not None:
            batches = index_batches(n)
            rows = None
        else:
            # ONE permutation per epoch: the six small per-sample tensors are permuted once, their minibatches are then
            # contiguous slices (views); only the observation stacks -- three frame rows per sample -- are gathered per
            # minibatch.  (Seven index kernels per minibatch before: 1 152 of them per 64-minibatch update.)
            perm = torch.randperm(n, device=advs.device)
            n_mb = n // batch_size if drop_last else -(-n // batch_size)
            batches = [perm[k * batch_size: min((k + 1) * batch_size, n)] for k in range(n_mb)]
            rows = tuple(x[perm] for x in small)
        if multi:
            # every rank must take the same number of optimiser steps (one gradient all-reduce each): Stage-2
            # filtering leaves a different row count on every rank, so agree on the minimum for this epoch
            s = torch.tensor([len(batches)], dtype=torch.int64, device=advs.device)
            dist.all_reduce(s, op=dist.ReduceOp.MIN)
            if int(s.item()) == 0:
                raise RuntimeError(f"a rank kept fewer than batch_size={batch_size} transitions ({n} on this one): "
                                   "lower --batch-size")
            batches = batches[:int(s.item())]
        kl_sum = torch.zeros((), device=advs.device)
        n_done, stop = 0, False
        fetch = _ObsPrefetch(obss, batches)
        for k, index in enumerate(batches):
            mb_obs = fetch.take(k)
            if rows is None:
                mb_goal, mb_speed, mb_action, mb_logprob, mb_target, adv = (x[index] for x in small)
            else:
                lo_, hi_ = k * batch_size, k * batch_size + index.numel()
                mb_goal, mb_speed, mb_action, mb_logprob, mb_target, adv = (x[lo_:hi_] for x in rows)
            if fused_loss:
                from . import policy_ops
                mean, new_value = policy.mean_value(mb_obs, mb_goal, mb_speed)
                loss, stats = policy_ops.ppo_loss(mean, new_value, policy.logstd, mb_action, mb_logprob, adv, mb_target,
                                                  clip_value, value_coef, coeff_entropy)
                policy_loss, value_loss, dist_entropy, kl_mb = stats[1], stats[2], stats[3], stats[4]
            else:
                with torch.autocast(advs.device.type, dtype=autocast_dtype, enabled=autocast_dtype is not None):
                    new_value, new_logprob, dist_entropy = policy.evaluate_actions(mb_obs, mb_goal, mb_speed, mb_action)
                new_value, new_logprob = new_value.float(), new_logprob.float()
                log_ratio = new_logprob - mb_logprob
                ratio = torch.exp(log_ratio)
                surrogate1 = ratio * adv
                surrogate2 = torch.clamp(ratio, 1 - clip_value, 1 + clip_value) * adv
                policy_loss = -torch.min(surrogate1, surrogate2).mean()
                value_loss = F.mse_loss(new_value, mb_target)
                loss = policy_loss + value_coef * value_loss - coeff_entropy * dist_entropy
                kl_mb = None
            if flat_grads is not None:
                flat_grads.backward(loss)
                flat_grads.all_reduce_mean(dist)
            else:
                optimizer.zero_grad()
                loss.backward()
            if max_grad_norm > 0:       # opt-in (not in the reference): global-norm clipping of the (averaged) gradient
                if flat_grads is not None:
                    flat_grads.flat.mul_(torch.clamp(max_grad_norm / (flat_grads.flat.norm() + 1e-6), max=1.0))
                else:
                    torch.nn.utils.clip_grad_norm_(policy.parameters(), max_grad_norm)
            optimizer.step()
            if logstd_min is not None:      # opt-in floor on the exploration noise (see trainer.HParams.logstd_min)
                with torch.no_grad():
                    policy.logstd.clamp_(min=logstd_min)
            stop = False
            if kl_ctl is not None:
                with torch.no_grad():
                    if kl_mb is None:
                        kl_mb = ((ratio - 1.0) - log_ratio).mean()        # k3 estimator of KL(old || new), >= 0
                    kl_sum += kl_mb
                n_done += 1
                stop = kl_ctl.should_stop(kl_mb, dist)
            if log is not None:
                log.append((policy_loss.detach(), value_loss.detach(), dist_entropy.detach()))
            if stop:
                break
        fetch.close()
        if kl_ctl is not None and n_done:
            kl_ctl.step(optimizer, kl_sum, n_done, dist)
        if stop:
            break


def ppo_update_stage1(policy, optimizer, batch_size, memory, epoch, coeff_entropy=0.02, clip_value=0.2,
                      num_step=2048, num_env=12, frames=1, obs_size=24, act_size=4, *, value_coef=20.0,
                      index_batches=None, dist=None, flat_grads=None, log=None, autocast_dtype=None, kl_ctl=None,
                      max_grad_norm=0.0, logstd_min=None):
    """model/ppo.py:143-194.  ``memory`` = (obss, goals, speeds, actions, logprobs, targets, values,
    rewards, advs) as device tensors shaped [T, N, ...] (obss may be a FrameRows: one stored frame per tick)."""
    obss, goals, speeds, actions, logprobs, targets, _values, _rewards, advs = memory
    mean, std = _global_mean_std(advs, dist)
    advs = (advs - mean) / std
    n = num_step * num_env
    obs_rows = obss if isinstance(obss, FrameRows) else obss.reshape(n, frames, obs_size)
    flat = (obs_rows, goals.reshape(n, 2), speeds.reshape(n, 2),
            actions.reshape(n, act_size), logprobs.reshape(n, 1), targets.reshape(n, 1), advs.reshape(n, 1))
    _ppo_epochs(policy, optimizer, batch_size, flat, epoch, coeff_entropy, clip_value, False, value_coef,
                index_batches, dist, flat_grads, log, autocast_dtype, kl_ctl, max_grad_norm, logstd_min)


def ppo_update_stage2(policy, optimizer, batch_size, memory, filter_index, epoch, coeff_entropy=0.02,
                      clip_value=0.2, num_step=2048, num_env=12, frames=1, obs_size=24, act_size=4, *,
                      value_coef=20.0, index_batches=None, dist=None, flat_grads=None, log=None, autocast_dtype=None,
                      kl_ctl=None, max_grad_norm=0.0, logstd_min=None):
    """model/ppo.py:197-259: the advantage statistics use ALL transitions, then the filtered rows
    are deleted and minibatches use drop_last=True."""
    obss, goals, speeds, actions, logprobs, targets, _values, _rewards, advs = memory
    mean, std = _global_mean_std(advs, dist)
    advs = (advs - mean) / std
    n = num_step * num_env
    keep = torch.ones(n, dtype=torch.bool, device=advs.device)
    if filter_index is not None and len(filter_index):
        keep[torch.as_tensor(filter_index, device=advs.device, dtype=torch.long)] = False
    obs_rows = obss if isinstance(obss, FrameRows) else obss.reshape(n, frames, obs_size)
    flat = tuple(x[keep] for x in (obs_rows, goals.reshape(n, 2), speeds.reshape(n, 2),
                                   actions.reshape(n, act_size), logprobs.reshape(n, 1), targets.reshape(n, 1),
                                   advs.reshape(n, 1)))
    _ppo_epochs(policy, optimizer, batch_size, flat, epoch, coeff_entropy, clip_value, True, value_coef,
                index_batches, dist, flat_grads, log, autocast_dtype, kl_ctl, max_grad_norm, logstd_min)
