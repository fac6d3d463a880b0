"""Vectorised twins of the reference's training loops.

``Stage1Trainer.run`` is ppo_stage1.py:39-131 with the 24 MPI ranks replaced by one batched device
env: state never leaves HBM, ``comm.gather`` / ``comm.scatter`` disappear, episodes restart inside
the env (auto reset), and the learner runs every HORIZON ticks exactly like rank 0 does.
Multi-GPU: one process per GPU, worlds sharded, no data-path collective; the only exchanges are the
flat gradient all-reduce and the 3-scalar advantage statistics, both RCCL.
"""
from dataclasses import dataclass

import torch

from . import ppo
from .net import CNNPolicy


@dataclass
class HParams:
    """ppo_stage1.py:22-35 (Stage-2 overrides: batch 512, epoch 4 -- ppo_stage2.py:28-29)."""
    horizon: int = 128
    gamma: float = 0.99
    lam: float = 0.95
    batch_size: int = 1024
    epoch: int = 2
    coeff_entropy: float = 5e-4
    clip_value: float = 0.1
    learning_rate: float = 5e-5
    laser_hist: int = 3
    obs_size: int = 512
    act_size: int = 2
    value_coef: float = 20.0          # model/ppo.py:185
    action_bound: tuple = ((0.0, -1.0), (1.0, 1.0))   # ppo_stage1.py:170
    inference_dtype: object = None    # None = fp32 like the reference; torch.bfloat16 = opt-in fast rollouts
    update_dtype: object = None       # autocast dtype of the PPO update's forward/backward (opt-in)
    single_frame_buffer: bool = True  # rollout buffer keeps one lidar frame per tick, not the 3-frame stack (ppo.RolloutBuffer)
    graph_tick: bool = False          # capture the rollout tick (policy + sampling + env tick + buffer stores) in a hipGraph
    rollout_fused: bool = False       # rollout inference through the HIP conv front end (net.mean_value_fused, fp32)
    # the PPO update differentiates the conv front end through the HIP forward / backward kernels (net.CNNPolicy.fused_train)
    update_fused: bool = False
    kl_target: float = 0.0            # > 0: KL-adaptive learning rate (ppo.KLAdaptiveLR; opt-in, large-batch regime)
    lr_max: float = 1e-3
    kl_stop: float = 0.0              # > 0: abandon the rest of an update when a minibatch reports KL > kl_stop x kl_target
    # > -inf: floor of the policy's log standard deviation.  The reference lets logstd run free (model/net.py:33) and
    # trains ~30 k optimiser steps at lr 5e-5; at 10x the steps the noise of the speed channel collapses (sigma 0.05,
    # then 0.003: profiles/r02/r02_b_*), the importance ratios blow up and the policy degrades -- opt-in floor.
    logstd_min: float = float("-inf")
    max_grad_norm: float = 0.0        # > 0: global-norm gradient clipping (opt-in; the reference clips nothing)


def broadcast_parameters(module, dist):
    """Rank 0's parameters to every rank.  The broadcast writes INTO the parameter under no_grad (not through
    ``p.data``, whose writes do not bump the parameter's version counter), and the tower-major copies the fused rollout
    path reads are rebuilt afterwards: a broadcast cannot leave a rank acting on its own initial weights."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        with torch.no_grad():
            for p in module.parameters():
                dist.broadcast(p, src=0)
        if hasattr(module, "refresh_rollout_cache") and getattr(module, "_rc", None) is not None:
            module.refresh_rollout_cache()


class Stage1Trainer:
    GRAPH_RUN = 8       # ticks per replay of the long graph of ``run`` (hp.graph_tick)

    def __init__(self, env, policy=None, hp=None, dist=None, seed=0, stage2=False):
        self.env, self.dist, self.stage2 = env, dist, stage2
        self.hp = hp or HParams()
        dev = env.device
        torch.manual_seed(seed)
        self.policy = policy or CNNPolicy(frames=self.hp.laser_hist, action_space=self.hp.act_size,
                                          beams=self.hp.obs_size)
        self.policy.to(dev)
        if self.hp.update_fused:
            self.policy.fused_train = True
        broadcast_parameters(self.policy, dist)
        # the reference's optimiser (ppo_stage1.py:176: Adam, lr 5e-5).  On the GPU the parameters, their gradients and the
        # moment estimates are flat buffers and a step is ONE element-wise launch (ppo.FlatAdam, csrc/mrca_adam.hip;
        # PyTorch's fused multi-tensor step is 102 us on 34 workgroups, its default "foreach" form a dozen launches); on the
        # CPU torch.optim.Adam itself.  Built LAST: it re-seats the parameters (after .to(dev) and the broadcast).
        self.flat_grads = ppo.FlatGrads(self.policy.parameters())
        if dev.type == "cuda":
            self.optimizer = ppo.FlatAdam(self.flat_grads, lr=self.hp.learning_rate)
        else:
            self.optimizer = torch.optim.Adam(self.policy.parameters(), lr=self.hp.learning_rate)
        self.kl_ctl = ppo.KLAdaptiveLR(self.hp.kl_target, lr_max=self.hp.lr_max, stop_factor=self.hp.kl_stop) \
            if self.hp.kl_target > 0 else None
        self.buffer = ppo.RolloutBuffer(self.hp.horizon, env.N, self.hp.laser_hist, self.hp.obs_size, dev,
                                        self.hp.act_size, single_frame=self.hp.single_frame_buffer)
        if self.hp.single_frame_buffer and hasattr(env, "rollout_rows") and self.hp.act_size == 2:
            self.buffer.bind_env(env)       # the per-tick stores as two launches of the env's library
        self.gen = torch.Generator(device=dev)
        rank = dist.get_rank() if (dist is not None and dist.is_initialized()) else 0
        self.gen.manual_seed(seed * 1000 + rank)
        self.t = 0
        self.global_update = 0
        self.loss_log = []
        self.started = False
        self._graph = None
        self._graph_run = None
        self._t_idx = torch.zeros(1, dtype=torch.int64, device=dev)     # buffer row of the next tick, on the device

    def start(self):
        self.env.reset()
        self.started = True

    def _tick_body(self, noise=None):
        """The device work of one tick with the buffer row taken from ``self._t_idx`` (a device tensor): nothing in
        here depends on a host value, so it can be captured as a hipGraph.  ``noise``: this tick's sampling draws when
        the caller made them (a replay of several ticks draws all of them in one launch)."""
        env, hp, buf = self.env, self.hp, self.buffer
        obs, head = ppo.policy_input(env, hp.rollout_fused)
        v, a, logprob, scaled = ppo.generate_action(self.policy, obs, env.local_goal, env.speed, hp.action_bound, self.gen,
                                                    hp.inference_dtype, hp.rollout_fused, head, noise=noise)
        so, sn = (None, None) if buf.env_bound else self._stored_obs()
        buf.store_state_at(self._t_idx, so, env.local_goal, env.speed, a, logprob, v, env.fresh, newest=sn)
        env.step(scaled.contiguous())
        if not buf.store_outcome_at(self._t_idx, env.reward, env.done):
            self._t_idx.add_(1)

    def _stored_obs(self):
        """-> (obs, newest) for the rollout buffer: with the one-frame store and an env that keeps its stacks as a ring
        only the newest frame is read (no materialised copy of the stacks per tick)."""
        if self.hp.single_frame_buffer and hasattr(self.env, "newest_frame"):
            return None, self.env.newest_frame()
        return self.env.obs, None

    def _capture(self):
        """One tick as a hipGraph: ~50 launches (policy layers, sampling, the two env kernels, eight buffer stores)
        become one graph launch.  The policy's parameters, the env's arena and the rollout buffer are all at fixed
        addresses, the action-noise generator is registered with the graph, the buffer row is a device counter."""
        if hasattr(self.env, "enable_timing"):
            self.env.enable_timing(False)               # event records cannot be part of the captured tick
        if self.hp.rollout_fused:
            self.policy.refresh_rollout_cache()
        side = torch.cuda.Stream(device=self.env.device)
        side.wait_stream(torch.cuda.current_stream(self.env.device))
        with torch.cuda.stream(side):
            # warm-up on the capture stream (lazy inits, workspace allocs).  The three ticks really advance the env (their
            # transitions are not logged: the first update's episode statistics miss them) but all write the buffer row
            # the horizon is at, which the first real tick overwrites -- never a row past a short horizon
            for _ in range(3):
                self._t_idx.fill_(self.t)
                self._tick_body()
        torch.cuda.current_stream(self.env.device).wait_stream(side)
        self._t_idx.fill_(self.t)
        g = torch.cuda.CUDAGraph()
        g.register_generator_state(self.gen)
        with torch.cuda.graph(g, stream=side):
            self._tick_body()
        self._graph = g
        # ... and EIGHT ticks as one graph for ``run``: a replay has a start-up of its own (a few us the eager path hides
        # behind the previous tick's kernels) and the eight ticks' sampling noise is one launch instead of eight
        if self.hp.horizon >= self.GRAPH_RUN:
            g8 = torch.cuda.CUDAGraph()
            g8.register_generator_state(self.gen)
            with torch.cuda.graph(g8, stream=side):
                noise = torch.randn((self.GRAPH_RUN, self.env.N, 2), device=self.env.device, dtype=torch.float32,
                                    generator=self.gen)
                for i in range(self.GRAPH_RUN):
                    self._tick_body(noise[i])
            self._graph_run = g8

    def tick(self):
        """One pass of the while-loop body of ppo_stage1.py:64-118 for all robots."""
        env, hp, buf = self.env, self.hp, self.buffer
        if hp.graph_tick and self._graph is None:
            self._capture()                 # (its warm-up ticks advance the env: before the horizon's first row is set up)
        if self.t == 0:
            buf.begin_horizon(env.obs)      # one-frame store: the older frames of the stack the first tick sees
        if hp.graph_tick:
            self._graph.replay()
            if hasattr(env, "invalidate_views"):
                env.invalidate_views()          # the replayed tick advanced the ring behind the binding's back
        elif buf.env_bound:
            # launched kernel by kernel, the same body the graph captures: the row's stores are the library's two launches
            # and the row comes from the device-side counter (which the second of them moves on)
            self._tick_body()
        else:
            obs, head = ppo.policy_input(env, hp.rollout_fused)
            v, a, logprob, scaled = ppo.generate_action(self.policy, obs, env.local_goal, env.speed,
                                                        hp.action_bound, self.gen, hp.inference_dtype, hp.rollout_fused, head)
            so, sn = self._stored_obs()
            buf.store_state(self.t, so, env.local_goal, env.speed, a, logprob, v, env.fresh, newest=sn)
            env.step(scaled.contiguous())
            buf.store_outcome(self.t, env.reward, env.done)
        self.t += 1
        if self.t == hp.horizon:
            self.update()
            self.t = 0
            self._t_idx.zero_()

    def update(self):
        env, hp, buf = self.env, self.hp, self.buffer
        with torch.no_grad():                                                           # ppo_stage1.py:94-97
            obs, head = ppo.policy_input(env, hp.rollout_fused)
            if hp.rollout_fused:
                _mean, last_v = self.policy.mean_value_fused(obs, env.local_goal, env.speed, head=head)
            else:
                _mean, last_v = self.policy.mean_value(obs, env.local_goal, env.speed)
        targets, advs = ppo.generate_train_data(buf.reward, hp.gamma, buf.value, last_v, buf.done, hp.lam)
        memory = (buf.obs_rows(), buf.goal, buf.speed, buf.action, buf.logprob, targets, buf.value, buf.reward, advs)
        kw = dict(policy=self.policy, optimizer=self.optimizer, batch_size=hp.batch_size, memory=memory,
                  epoch=hp.epoch, coeff_entropy=hp.coeff_entropy, clip_value=hp.clip_value, num_step=hp.horizon,
                  num_env=env.N, frames=hp.laser_hist, obs_size=hp.obs_size, act_size=hp.act_size,
                  value_coef=hp.value_coef, dist=self.dist, flat_grads=self.flat_grads, log=self.loss_log,
                  autocast_dtype=hp.update_dtype, kl_ctl=self.kl_ctl, max_grad_norm=hp.max_grad_norm,
                  logstd_min=hp.logstd_min if hp.logstd_min > float("-inf") else None)
        if self.stage2:
            ppo.ppo_update_stage2(filter_index=ppo.get_filter_index(buf.done), **kw)
        else:
            ppo.ppo_update_stage1(**kw)
        if hp.rollout_fused:
            self.policy.refresh_rollout_cache()      # the rollout path reads tower-major copies of the parameters
        # the env's sticky device status (include/mrca_env.h: mrca_check -- "once per PPO update"): the update has just
        # synchronised with the host anyway, so the round trip is free; raises instead of training on an untrusted state
        if hasattr(env, "check"):
            env.check()
        self.global_update += 1

    @property
    def last_kl(self):
        return None if self.kl_ctl is None else self.kl_ctl.last_kl

    def run(self, ticks):
        """``ticks`` ticks, eight per graph replay where they fit inside the horizon, ``tick()`` for the rest.

        Reproducibility: with ``hp.graph_tick`` an eight-tick replay draws its sampling noise with ONE randn(8, N, 2), a
        single tick with randn(N, 2); the generator maps Philox offsets to elements by launch shape, so the two are
        different noise streams of the same seed.  A rollout is reproduced by the same seed, generator state AND the same
        grouping of ticks (same ``run`` arguments from the same ``t``); ``run()`` and ``tick()`` must not be mixed where
        value-for-value replay matters -- mrca/train.py drives ``tick()`` only, bench.py ``run()`` only."""
        if not self.started:
            self.start()
        env, hp = self.env, self.hp
        while ticks > 0:
            if hp.graph_tick and self._graph is None:
                self._capture()
            n = self.GRAPH_RUN
            if hp.graph_tick and self._graph_run is not None and ticks >= n and hp.horizon - self.t >= n:
                # eight ticks in one replay (never across the end of a horizon: the update runs between two replays)
                if self.t == 0:
                    self.buffer.begin_horizon(env.obs)
                self._graph_run.replay()
                if hasattr(env, "invalidate_views"):
                    env.invalidate_views()
                self.t += n
                ticks -= n
                if self.t == hp.horizon:
                    self.update()
                    self.t = 0
                    self._t_idx.zero_()
            else:
                self.tick()
                ticks -= 1


def make_bench_step(env, mode, dist, batch_size=16384, inference_dtype=None, update_dtype=None, fused=False,
                    graph=False, update_fused=False):
    """bench.py --mode rollout|train: returns step_fn(k) doing one tick for all robots."""
    hp = HParams(batch_size=batch_size, inference_dtype=inference_dtype, update_dtype=update_dtype, rollout_fused=fused,
                 graph_tick=graph, update_fused=update_fused)
    tr = Stage1Trainer(env, hp=hp, dist=dist, seed=0)
    tr.started = True  # bench.py resets the env itself
    if mode == "rollout" and graph:
        # the rollout of bench.py: policy + sampling + env tick (no buffer stores), captured once, replayed per tick
        if fused:
            tr.policy.refresh_rollout_cache()

        def body(noise=None):
            obs, head = ppo.policy_input(env, hp.rollout_fused)
            _v, _a, _lp, scaled = ppo.generate_action(tr.policy, obs, env.local_goal, env.speed, hp.action_bound, tr.gen,
                                                      hp.inference_dtype, hp.rollout_fused, head, noise=noise)
            env.step(scaled.contiguous())
        env.enable_timing(False)
        side = torch.cuda.Stream(device=env.device)
        side.wait_stream(torch.cuda.current_stream(env.device))
        with torch.cuda.stream(side):
            for _ in range(3):
                body()
        torch.cuda.current_stream(env.device).wait_stream(side)
        # one graph of ONE tick and one of EIGHT: a replay has a start-up of its own (a few us the eager path hides behind
        # the previous tick's kernels: 265 vs 256 us per tick with one tick per replay, profiles/r04_v_bench_rollout*.json),
        # so a run of n ticks replays the long graph n // 8 times and the short one for the remainder
        # the sampling noise of all the ticks of a replay is drawn by ONE launch at its head (a randn of 8192 numbers is a
        # 5 us launch of its own in a 250 us tick).  NOT the same numbers as eight one-tick draws: the generator's Philox
        # offset -> element mapping depends on the launch's shape, so g8 and g1 are two different noise streams of one seed)
        def capture(ticks):
            g = torch.cuda.CUDAGraph()
            g.register_generator_state(tr.gen)
            with torch.cuda.graph(g, stream=side):
                noise = torch.randn((ticks, env.N, 2), device=env.device, dtype=torch.float32, generator=tr.gen)
                for i in range(ticks):
                    body(noise[i])
            return g
        g1, g8 = capture(1), capture(8)

        def step_fn(_k):
            g1.replay()

        def run_ticks(n):
            for _ in range(n // 8):
                g8.replay()
            for _ in range(n % 8):
                g1.replay()
        step_fn.run_ticks = run_ticks
        return step_fn
    if mode == "rollout":
        def step_fn(_k):
            obs, head = ppo.policy_input(env, hp.rollout_fused)
            _v, _a, _lp, scaled = ppo.generate_action(tr.policy, obs, env.local_goal, env.speed,
                                                      hp.action_bound, tr.gen, hp.inference_dtype, hp.rollout_fused, head)
            env.step(scaled.contiguous())
        return step_fn

    def train_fn(_k):
        tr.tick()
    train_fn.run_ticks = tr.run          # (with the rollout's ticks as hipGraphs: eight ticks per replay where they fit)
    return train_fn
