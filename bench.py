#!/usr/bin/env python3
"""bench.py -- agent-steps/s of the MI355X-native multi-robot environment (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode env|rollout|train] [--no-graph] [--fidelity]

A "step" is one pass of the hot path over one batch: every robot of every world on this rank
advances one Stage tick (latch action -> kinematics -> collision -> 512-beam ray cast -> reward /
terminal / auto-reset -> observation stack).  Workload at N=1 = BASELINE.json configs[1]: 4096
robots (128 independent Stage-1 rinks x 32 robots), 512 beams, state resident in HBM.  For N>1 the
worlds are sharded across ranks with no data-path collective (weak scaling: 4096 robots per GPU).

--mode env      (default, the figure `value`, the >=10 M target and `roofline` refer to)
--mode rollout  env + policy inference per tick                                  
--mode train    rollout + GAE + PPO update with RCCL gradient all-reduce         

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (raycast_kernel): its launches' own begin / end stamps
(hipExtLaunchKernel) in an eager pass over all worlds right after the timed region; `cpu_baseline` times the plain-C / OpenMP
port of the oracle on all host threads over the same workload (a port, not the reference binary, which cannot run here --
BASELINE.md 3) and reports the NumPy oracle's one-core figure beside it.  The line also carries side figures of the other
configurations the documents quote (Stage-2 map, fidelity mode, reference-shaped observations, rollout), each measured
after the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "rl-collision-avoidance_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

SIDE_FIGURE_TIMEOUT_S = 240
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_NS_PER_WAVE_INST = 1.03   # ns a SIMD takes per wave64 VALU instruction at 8 waves per SIMD (tools/valu_rate_probe.hip, measured)
# Algorithmic HBM bytes per agent-step.  Since round 4 (ABI 4) the frame history is a ring of RAW scans: a tick
# writes ONE 2 kB row per robot and ~0.1 kB of state -- SURVEY's strict B_env.  (Round 3 stored every beam twice, scan +
# normalised frame: 4188 B; rounds 1-2 shifted the stack as well: B_env_stack = 10 332 B.)
B_ENV_STRICT = 4 * 512 + 92                    # SURVEY 8(d) B_env: ONE 2 kB row + 92 B of state = 2140
RAY_BYTES_PER_AGENT_STEP = 4 * 512 + 48        # per LAUNCH: the scan row written; pose / head / goal / flags read, local goal
MOVE_BYTES_PER_AGENT_STEP = 140                # per LAUNCH: pose, speeds, goal, counters, flags, head record read + written
BYTES_PER_AGENT_STEP = RAY_BYTES_PER_AGENT_STEP + MOVE_BYTES_PER_AGENT_STEP    # both launches of the tick = 2236
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak (the reference's precision: model/net.py is fp32)


def policy_flops(beams=512, frames=3, act=2):
    """FLOPs (2 per multiply-add) of ONE forward pass of CNNPolicy for one sample, both towers, counted from the layer shapes of
    model/net.py:19-33: conv1 (frames -> 32, k 5, s 2, p 1), conv2 (32 -> 32, k 3, s 2, p 1), fc1 (32 x L2 -> 256), fc2
    (256 + 2 + 2 -> 128), heads (128 -> 1 twice for the actor, 128 -> 1 for the critic).  512 beams: 6.39 MFLOP (6.4)."""
    l1 = (beams + 2 - 5) // 2 + 1           # 255
    l2 = (l1 + 2 - 3) // 2 + 1              # 128
    tower = 2 * (frames * 5 * 32 * l1 + 32 * 3 * 32 * l2 + 32 * l2 * 256 + (256 + 2 + act) * 128)
    return 2 * tower + 2 * 128 * (act + 1)


def flop_roofline(value, flop_per_agent_step, what, kernels=None):
    """The `roofline` block of a figure whose work is the policy's matrix arithmetic: fp32 on the MFMA pipes (the reference
    trains and acts in fp32, so the roof is the fp32 matrix peak, not bf16's).  `achieved` = model FLOPs of the WHOLE figure
    (env kernels, sampling, gathers, optimiser and launch gaps all inside the measured time) -- a lower bound on what the
    matrix kernels themselves reach; `kernels`: the heaviest kernels' own rates from the committed rocprofv3 summaries."""
    tf = value * flop_per_agent_step / 1e12
    out = {"bound": "fp32_mfma", "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_MFMA_PEAK_TFLOPS,
           "flop_per_agent_step": flop_per_agent_step, "what": what}
    if kernels:
        out["kernels"] = kernels
    return out


# the heaviest kernels of the rollout tick / of one PPO minibatch, from the committed rocprofv3 --kernel-trace --stats
# summaries (average duration per call at 4096 robots / 16 384-row minibatches; FLOPs from policy_flops()'s layer terms)
def _kernel_rates():
    l1, l2 = 255, 128
    conv = 2 * 2 * (3 * 5 * 32 * l1 + 32 * 3 * 32 * l2)       # both towers, per sample
    fc1 = 2 * 2 * 32 * l2 * 256

    def row(name, us, flop, n, src):
        tf = flop * n / (us * 1e-6) / 1e12
        return {"kernel": name, "avg_us": us, "tflops": tf, "frac": tf / FP32_MFMA_PEAK_TFLOPS, "source": src}
    roll = [row("fc1 GEMM (hipBLASLt, both towers batched)", 124.5, fc1, 4096, "profiles/r06_am_rollout_kernel_stats.csv"),
            row("lidar_features_kernel<true> (conv1 + conv2, fp32 MFMA)", 91.8, conv, 4096, "profiles/r06_am_rollout_kernel_stats.csv")]
    train = [row("lidar_features_bwd_kernel (conv1 + conv2 backward)", 721.4, 2 * conv, 16384, "profiles/r06_am_train_kernel_stats.csv"),
             row("fc1 GEMMs of the update (forward / dgrad / wgrad, per tower)", (244.7 + 236.7 + 235.7) / 3, fc1 / 2, 16384, "profiles/r06_am_train_kernel_stats.csv"),
             row("lidar_features_kernel<false> (conv forward of the update)", 310.8, conv, 16384, "profiles/r06_am_train_kernel_stats.csv")]
    return roll, train


def code_only(src):
    """C++ source without comments and blank lines (what the kernel-source stamp hashes)."""
    import re
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    return "\n".join(ln.rstrip() for ln in src.splitlines() if ln.strip())


def kernel_source_hash():
    """sha256 (16 hex digits) of the CODE of the device sources the env kernels are compiled from (comments and blank
    lines stripped: counters measured on one version of the kernels say nothing about another, but a reworded comment
    changes nothing the counters saw)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("mrca_kernels.hip", "mrca_device.h", "mrca_kernels.h"):
        h.update(code_only(open(os.path.join(ROOT, "rl-collision-avoidance_amd", "csrc", f)).read()).encode())
    return h.hexdigest()[:16]


def pmc_traffic(robots, scenario):
    """HBM bytes per raycast_kernel launch from the committed rocprofv3 PMC passes (profiles/
    pmc_traffic.json: FETCH_SIZE and WRITE_SIZE in KiB per launch at the profiled robot count,
    FETCH doubled per MI355X_MICROARCH.md's gfx950 correction), scaled linearly to `robots`.
    Refused (None + a note) when the counters were collected on another version of the kernel sources or on
    another scenario: a stale figure must not ride along silently."""
    f = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(f):
        return None, "no profiles/pmc_traffic.json"
    d = json.load(open(f))
    if d.get("kernel_src_sha16") != kernel_source_hash():
        return None, (f"profiles/pmc_traffic.json was measured on kernel sources {d.get('kernel_src_sha16')}, this "
                      f"build is {kernel_source_hash()}: re-run tools/pmc_profile.sh")
    if d.get("scenario", "stage1") != scenario:
        return None, f"profiles/pmc_traffic.json is for scenario {d.get('scenario', 'stage1')}"
    per_robot = (2.0 * d["fetch_kib"] + d["write_kib"]) * 1024.0 / d["robots"]
    note = ("bytes/launch from profiles/pmc_traffic.json (separate rocprofv3 --pmc passes on these kernel sources; "
            "2*FETCH_SIZE + WRITE_SIZE)")
    if d.get("move_fetch_kib") is not None:
        mv = (2.0 * d["move_fetch_kib"] + d["move_write_kib"]) * 1024.0 / d["robots"] * robots
        note += f"; the move kernel's launch: {mv / 1e6:.2f} MB"
    global _SQ_VALU_PER_WAVE
    if d.get("raycast_sq_insts_valu_per_launch") and d.get("raycast_sq_waves_per_launch"):
        _SQ_VALU_PER_WAVE = d["raycast_sq_insts_valu_per_launch"] / d["raycast_sq_waves_per_launch"]
    return per_robot * robots, note


_SQ_VALU_PER_WAVE = None       # SQ_INSTS_VALU / SQ_WAVES of the ray cast from the committed counter pass (same kernel sources)


def cpu_baseline(sc_name, worlds, robots_per_world, seconds_target=12.0, fidelity=False):
    """The CPU port of the path timed on this box's host cores: the plain-C restatement of the oracle
    (oracle/mrca_oracle_c.c, OpenMP over worlds and robots, all cores) on the SAME workload, for a
    bounded number of ticks; the NumPy oracle's single-core figure is reported alongside."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    # a container may expose more CPUs than its cgroup lets it use: size the OpenMP team to the quota
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().split()
        cpu_limit = None if quota[0] == "max" else int(quota[0]) / int(quota[1])
    except Exception:
        cpu_limit = None
    usable = len(os.sched_getaffinity(0))
    if cpu_limit is not None:
        usable = max(1, min(usable, int(cpu_limit + 0.999)))
    import util as U
    from mrca import scenario as S
    sc = S.stage1(num_worlds=worlds, robots_per_world=robots_per_world, seed=0, stage_resolution=fidelity) \
        if sc_name == "stage1" else S.stage2(num_worlds=worlds, seed=0, stage_resolution=fidelity)
    env = U.COracleEnv(sc)
    env.lib.oc_set_threads(int(os.environ.get("OMP_NUM_THREADS", usable)))   # the OpenMP runtime is already up (torch)
    threads = int(env.lib.oc_max_threads())
    env.reset()
    rng = np.random.default_rng(1)
    env.step(U.random_actions(rng, sc.num_robots))
    t0 = time.perf_counter()
    ticks = 0
    while time.perf_counter() - t0 < seconds_target and ticks < 500:
        env.step(U.random_actions(rng, sc.num_robots))
        ticks += 1
    dt = time.perf_counter() - t0
    out = {"value": sc.num_robots * ticks / dt, "unit": "agent-steps/s", "cores": threads, "kind": "port",
           "cgroup_cpu_limit": "unlimited" if cpu_limit is None else f"{cpu_limit:.1f} CPUs",
           "visible_cpus": os.cpu_count(),
           "sample": f"C/OpenMP port of the oracle, {sc_name}: {worlds} worlds x {sc.robots_per_world} robots x 512 "
                     f"beams (the GPU workload), {ticks} ticks in {dt:.1f} s on {threads} host threads"}
    # NumPy oracle, one core, small sample
    sc_s = S.stage1(num_worlds=4, robots_per_world=32, seed=0)
    ora = U.oracle_env(sc_s, np.float32)
    ora.reset()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 4.0:
        ora.step(U.random_actions(rng, sc_s.num_robots))
        n += 1
    out["numpy_oracle_1core"] = sc_s.num_robots * n / (time.perf_counter() - t0)
    return out


class TickSchedule:
    """How env mode runs ticks [first, first + count) of every world, with tick k taking its actions from pool[k % len(pool)].

    graph=True (the default of bench.py): the ticks are replayed as hipGraphs of up to 64 ticks (+ remainder graphs): EXACTLY
    the ticks asked for run, with the eager run's action sequence.
    graph=False: every kernel is launched from the host.

    chains = P > 1: the worlds are dealt to P contiguous ranges, each a chain of its own -- move launch, ray cast, move launch,
    ... on its own stream (its own branch of the graph) -- held half a tick apart by one event per tick: range c's move launch
    waits for range c - 1's move launch of the same tick, so it runs NEXT TO that range's ray cast.  Worlds never interact
    (mrca_step_worlds), so the result is the one-chain result bit for bit (tests/test_gpu_parity.py); what changes is that
    the latency-bound move launch (10 us at 4 % of the chip's issue slots) no longer has the chip to itself."""

    def __init__(self, env, pool, chains=1, graph=True, native=False, chained=False):
        # native=True: one mrca_step_many call per run() -- the library enqueues every launch of the ticks itself (two plain
        # streams for two chains, one event to set them half a tick apart): no graph to instantiate, upload or launch, and no
        # Python between the launches
        self.env, self.pool, self.graph, self.native = env, pool, graph and not native, native
        # native, chained=True: round 5's schedule inside mrca_step_many (chains = -P: one `move, ray, move, ray ...` chain per
        # world range) instead of round 6's run-ahead schedule (move launches on a stream of their own, ahead of the ray casts)
        self.chained = chained
        self.chains = max(1, min(int(chains), env.W))
        W, P = env.W, self.chains
        self.ranges = [(c * W // P, (c + 1) * W // P - c * W // P) for c in range(P)]
        self._side, self._extra = None, None       # streams of the graph / eager schedules, made when first needed
        self.graphs = {}
        self.sync_every_tick = os.environ.get("MRCA_CHAIN_SYNC", "first") == "every"
        # ticks per hipGraph: a graph forks into the chains at its head and joins them at its end, and the chains are set half
        # a tick apart once per graph -- the longer the graph, the less of a tick that is
        # (native: ticks per mrca_step_many call -- the library's run-ahead pass covers up to 256 ticks, and every pass ends with
        # a join of its streams that costs ~90 us: 64-tick calls were 8 % of a long run)
        self.ticks_per_graph = int(os.environ.get("MRCA_TICKS_PER_GRAPH", "256" if (native and not chained) else "64"))

    @property
    def side(self):
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.env.device)
        return self._side

    @property
    def extra_streams(self):
        if self._extra is None:
            self._extra = [torch.cuda.Stream(device=self.env.device) for _ in range(self.chains - 1)]
        return self._extra

    def chunks(self, first, count):
        """(start, length) pieces of ticks [first, first + count): at most `ticks_per_graph` ticks each (a graph may run round the
        action pool several times: tick k takes pool[k % len(pool)] wherever it sits), cut at multiples of it so that the same
        few graphs serve a long run."""
        G = self.ticks_per_graph
        k = first
        while k < first + count:
            m = min(G - k % G, first + count - k)
            yield k, m
            k += m

    def issue(self, start, count):
        """the launches of ticks [start, start + count) on the current stream (and, with chains, on the extra streams forked
        from it and joined back into it)"""
        env, pool, L = self.env, self.pool, len(self.pool)
        if self.chains == 1:
            for j in range(count):
                env.step(pool[(start + j) % L])
            return
        cur = torch.cuda.current_stream(env.device)
        streams = [cur] + self.extra_streams
        for s in streams[1:]:
            s.wait_stream(cur)
        for j in range(count):
            a = pool[(start + j) % L]
            moved = None
            for s, r in zip(streams, self.ranges):
                with torch.cuda.stream(s):
                    if moved is not None and (j == 0 or self.sync_every_tick):
                        s.wait_event(moved)         # half a tick behind the previous range
                    env.move(a, r)
                    moved = torch.cuda.Event()
                    moved.record(s)
                    env.observe(r)
        for s in streams[1:]:
            cur.wait_stream(s)

    def graph_of(self, start, count):
        key = (start % len(self.pool), count)       # (a graph is its first pool entry and its length)
        if key not in self.graphs:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.side):
                self.issue(start, count)
            self.graphs[key] = g
        return self.graphs[key]

    def capture(self, first, count):
        """every capture the ticks [first, first + count) need (a capture replays nothing: the env stands still)"""
        if self.graph:
            for k, m in self.chunks(first, count):
                self.graph_of(k, m)

    def prime(self, first, count):
        """a graph's FIRST launch uploads it to the device: launch every graph of [first, first + count) once -> the number of
        (extra, really executed) ticks that took"""
        primed = 0
        if self.graph:
            for key in sorted({(k % len(self.pool), m) for k, m in self.chunks(first, count)}):
                self.graphs[key].replay()
                primed += key[1]
        return primed

    def run(self, first, count):
        if self.native:
            # (at most `ticks_per_graph` ticks per call: thousands of launches enqueued at once run the HIP runtime out of its
            # pools, and the region after such a call was measured 15 % slow, profiles/r05_n_*)
            for k, m in self.chunks(first, count):
                self.env.step_many(self.pool, k, m, -self.chains if self.chained else self.chains)
        elif not self.graph:
            self.issue(first, count)
        else:
            for k, m in self.chunks(first, count):
                self.graph_of(k, m).replay()
        if hasattr(self.env, "invalidate_views"):
            self.env.invalidate_views()


def action_pool(N, dev, seed, depth=16):
    """`depth` action batches v~U(0,1), w~U(-1,1) (the clipped range, ppo_stage1.py:170); tick k of a schedule takes entry
    k % depth.  bench.py makes the pool as deep as the region it times (<= 1024), so the actions are i.i.d. per tick as
    SURVEY 8(d) says; 32 kB per entry at 4096 robots."""
    gen = torch.Generator(device=dev).manual_seed(seed)
    return [torch.stack([torch.rand(N, generator=gen, device=dev),
                         torch.rand(N, generator=gen, device=dev) * 2 - 1], 1).contiguous() for _ in range(depth)]


def env_side_figure(sc, ticks, chains, lazy_obs=True, seed=1, note="", schedule="native"):
    """agent-steps/s of another configuration, measured outside the timed region with the same schedule as `value` (ticks
    replayed as hipGraphs off a 16-deep action pool): a new env, reset, the graphs captured and launched once, 32 warm-up
    ticks, then `ticks` timed ticks between two device synchronisations."""
    from mrca.vec_env import VecStageWorld
    env = VecStageWorld(sc, lazy_obs=lazy_obs)
    try:
        pool = action_pool(sc.num_robots, env.device, seed, depth=max(16, min(1024, 32 + ticks)) if schedule != "graph" else 16)
        sched = TickSchedule(env, pool, chains=chains, graph=schedule == "graph", native=schedule in ("native", "chained"),
                             chained=schedule == "chained")
        env.reset()
        for k in range(3):
            env.step(pool[k])
        torch.cuda.synchronize()
        sched.capture(0, 32)
        sched.capture(32, ticks)
        sched.prime(32, ticks)
        sched.run(0, 32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sched.run(32, ticks)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        env.check()
        return {"value": sc.num_robots * ticks / dt, "unit": "agent-steps/s", "ms_per_step": dt / ticks * 1e3,
                "robots": sc.num_robots, "ticks": ticks, "chains": sched.chains, "note": note}
    finally:
        env.close()


def side_scenario(name, scenario, worlds, robots_per_world, seed):
    """the scenario of the env side figure `name` (the three configurations DESIGN.md / README quote beside `value`)"""
    from mrca import scenario as S
    if name == "stage2_side_figure":
        return S.stage2(num_worlds=187, seed=seed), True
    if name == "fidelity_side_figure":
        return (S.stage1(num_worlds=worlds, robots_per_world=robots_per_world, seed=seed, stage_resolution=True)
                if scenario == "stage1" else S.stage2(num_worlds=worlds, seed=seed, stage_resolution=True)), True
    if name == "reference_shaped_obs_side_figure":
        return (S.stage1(num_worlds=worlds, robots_per_world=robots_per_world, seed=seed) if scenario == "stage1" else
                S.stage2(num_worlds=worlds, seed=seed)), False
    raise SystemExit(f"bench.py: no side figure called {name}")


def side_figure_main(argv):
    """`python bench.py --side-figure NAME ...`: ONE env side figure in a process of its own, one JSON dict on stdout.  The
    parent run starts these (see main): what a multi-stream schedule reaches depends on which hardware queues the runtime
    gives its streams, and that on everything the process has created before (torch's stream pools, captured graphs, other
    envs: tools/stream_pressure_probe.py measured 145 - 310 M for the SAME call on the Stage-2 map, profiles/r06_j_*) -- a
    figure quoted beside `value` is measured the way `value` is: by a process that has done nothing else yet."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--side-figure", required=True)
    ap.add_argument("--scenario", default="stage1")
    ap.add_argument("--worlds", type=int, default=128)
    ap.add_argument("--robots-per-world", type=int, default=32)
    ap.add_argument("--chains", type=int, default=2)
    ap.add_argument("--schedule", default="native")
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--ticks", type=int, default=300)
    ap.add_argument("--device-index", type=int, default=0)
    a = ap.parse_args(argv)
    torch.cuda.set_device(a.device_index)
    import __graft_entry__ as G
    G.build()
    sc, lazy = side_scenario(a.side_figure, a.scenario, a.worlds, a.robots_per_world, a.seed)
    print(json.dumps(env_side_figure(sc, ticks=a.ticks, chains=a.chains, lazy_obs=lazy, schedule=a.schedule)))


def assemble_line(*, args, sc, N, world_size, value, elapsed, ray_ms, mv_ms, launches, kernel_timing_note, sched, extra,
                  per_rank=None, ranks_seen=None, devices=None, backend=None, cpu_baseline_fn=None):
    """The ONE JSON line of a run as a dict, from what the run measured (rank 0; a pure function of its arguments, so that
    tests/test_host_bench_line.py can put an 8-rank run through it without eight GPUs).  `sched`: None or the schedule's
    {graph, native, chains}."""
    # the launches' own begin / end stamps (hipExtLaunchKernel start / stop events: what rocprofv3 reports); rounds 1-3
    # recorded events AROUND each launch, which read ~2.5 us longer per kernel -- their sum exceeded ms_per_step
    ray_avg_s = (ray_ms / launches) * 1e-3 if launches else float("nan")
    mv_avg_s = (mv_ms / launches) * 1e-3 if launches else float("nan")
    traffic, traffic_note = pmc_traffic(N, args.scenario + ("-fidelity" if args.fidelity else ""))
    achieved = RAY_BYTES_PER_AGENT_STEP * N / ray_avg_s / 1e9 if launches else None
    tick_achieved = BYTES_PER_AGENT_STEP * N / (ray_avg_s + mv_avg_s) / 1e9 if launches else None
    move_achieved = MOVE_BYTES_PER_AGENT_STEP * N / mv_avg_s / 1e9 if launches else None
    out = {
        "metric": "agent-steps/s (N robots x 512-beam lidar)" if args.mode == "env" else
                  f"agent-steps/s ({args.mode}: env + policy" + (" + GAE + PPO update)" if args.mode == "train" else ")"),
        "value": value, "unit": "agent-steps/s", "n_gpus": world_size, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.scenario}: {args.worlds} worlds x {sc.robots_per_world} robots = {N} "
                               f"robots/GPU, 512 beams, 3 frames, cell {sc.grid.cell} m, auto-reset, "
                               f"random actions v~U(0,1) w~U(-1,1); mode={args.mode}" +
                               ("; FIDELITY mode: Stage's resolutions, raster collisions and raster lidar returns of "
                                f"robots (collision_raster {sc.collision_raster} m)" if args.fidelity else ""),
                   "robots_per_gpu": N, "beams": sc.beams, "mode": args.mode,
                   "policy_inference_dtype": args.policy_dtype if args.mode != "env" else None,
                   "policy_inference_path": (args.policy_path if args.policy_dtype == "f32" else "stock")
                   if args.mode != "env" else None,
                   "tick_as_hipgraph": (sched['graph'] if sched is not None else not args.no_graph),
                   "launches_from": (None if sched is None else "one mrca_step_many call per timed region (the library "
                                     "enqueues every launch)" if sched['native'] else "hipGraph replays" if sched['graph'] else
                                     "Python, tick by tick"),
                   "chains": (sched['chains'] if sched is not None else None),
                   "schedule": (None if sched is None else
                                f"run-ahead (round 6): the move launches of the region's ticks on a stream of their own, each tick "
                                f"into a slot of its own, ahead of the ray casts of {sched['chains']} world range(s) on "
                                f"{sched['chains']} stream(s), every ray cast behind the event of its tick's move launch "
                                "(include/mrca_env.h mrca_step_many, DESIGN.md 5.10)"
                                if sched['native'] and not sched.get('chained') else
                                "one chain: move launch, ray cast over all worlds" if sched['chains'] == 1 else
                                f"{sched['chains']} world ranges half a tick apart (a range's move launch runs next to the "
                                "previous range's ray cast: mrca_move_worlds / mrca_observe_worlds on two streams / graph "
                                "branches)"),
                   "ppo_update_dtype": args.update_dtype if args.mode == "train" else None,
                   "ppo_update_path": (args.update_path if args.update_dtype == "f32" else "stock")
                   if args.mode == "train" else None},
        "roofline": {"bound": "hbm", "kernel": "raycast_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                     "traffic_note": traffic_note,
                     "bytes_per_agent_step": RAY_BYTES_PER_AGENT_STEP, "kernel_avg_us": ray_avg_s * 1e6,
                     "tick": {"achieved": tick_achieved, "frac": tick_achieved / HBM_PEAK_GBS if tick_achieved else None,
                              "bytes_per_agent_step": BYTES_PER_AGENT_STEP,
                              "frac_at_survey_B_env_2140": (tick_achieved * B_ENV_STRICT / BYTES_PER_AGENT_STEP / HBM_PEAK_GBS)
                              if tick_achieved else None,
                              "note": "both launches of the tick (move, ray cast) against ONE scan row + state = 2236 B per "
                                      "agent-step (SURVEY's strict B_env = 2140): since ABI 4 the frame history is a ring of "
                                      "raw scans, the observation x/6 - 0.5 is formed by its readers (round 3: 4188 B, every "
                                      "beam stored twice; rounds 1-2: 10 332 B with the shift)"},
                     "move_launch": {"achieved": move_achieved,
                                     "frac": move_achieved / HBM_PEAK_GBS if move_achieved else None,
                                     "bytes_per_agent_step": MOVE_BYTES_PER_AGENT_STEP},
                     "move_kernel_avg_us": mv_avg_s * 1e6 if launches else None,
                     # the OTHER roof (HBM is only the nominal one): vector-ALU issue.  Instructions per wave from the committed
                     # SQ counter pass x the launch's waves x the time a SIMD needs per wave64 instruction -- MEASURED: 1.03 ns
                     # with eight waves per SIMD (tools/valu_rate_probe.hip, profiles/r05_h_valu_rate_probe.txt: 2.2 cycles at
                     # the clock the chip holds; rounds 1-4 priced 4 cycles and called the launch issue-bound at 63 %)
                     "valu_issue": ({"insts_per_wave": _SQ_VALU_PER_WAVE,
                                     "frac_of_issue_slots": _SQ_VALU_PER_WAVE * (N * sc.beams / 2 / 64) * VALU_NS_PER_WAVE_INST * 1e-9 /
                                     (1024 * ray_avg_s),
                                     "note": "SQ_INSTS_VALU per wave (profiles/pmc_traffic.json, same kernel sources) x waves x "
                                             "1.03 ns per wave64 instruction and SIMD (measured, tools/valu_rate_probe.hip) / (1024 "
                                             "SIMDs x launch time)"}
                                    if (_SQ_VALU_PER_WAVE and launches and args.scenario == "stage1" and not args.fidelity)
                                    else None),
                     # the two launches of a tick over ALL worlds, by their own stamps: the tick a single chain cannot beat,
                     # free of the host and of the timed region's length (a 20-step region is 0.5 ms of wall clock)
                     "kernel_sum_us": (ray_avg_s + mv_avg_s) * 1e6 if launches else None,
                     "value_at_kernel_sum": N * world_size / (ray_avg_s + mv_avg_s) if launches else None,
                     "launches_timed": launches, "kernel_timing": kernel_timing_note,
                     # the same bytes at the rate the TIMED REGION sustained: with the run-ahead schedule the ray casts of two world
                     # ranges overlap and the move launches run beside them, so a tick costs less than one launch's own duration
                     "sustained": ({"achieved": value / world_size * B_ENV_STRICT / 1e9,
                                    "frac": value / world_size * B_ENV_STRICT / 1e9 / HBM_PEAK_GBS,
                                    "bytes_per_agent_step": B_ENV_STRICT,
                                    "note": "SURVEY 8d's B_env (2140 B per agent-step) x `value` per GPU: the whole tick at the "
                                            "rate the timed region ran at"} if args.mode == "env" else None),
                     "note": "HBM is the nominal roof (SURVEY 8d: 2.1 kB per agent-step).  What the launch is made of: two "
                             "residency rounds of 2048 workgroups at eight waves per SIMD, each wave a chain of dependent memory "
                             "round trips (robot record, ~2.6 field lookups per beam, LDS hand-offs) around ~546 VALU instructions "
                             "-- about half of the SIMDs' issue slots while it runs, neither issue- nor bandwidth-bound: "
                             "launch time = one workgroup's lifetime + robots / throughput (8.2 us + 3.05 us per 1000 robots, "
                             "profiles/r05_c_*), which is why overlapping launches pay (DESIGN.md 5.2, 5.10); a second pass over the "
                             "same robots inside one launch, everything cache-hot, costs 90 - 92 % of the first "
                             "(profiles/r06_o_hot_pass_probe.txt): the launch does not wait for its fetches"},
    }
    if args.mode == "rollout":
        # the rollout's own roofline: the policy forward is 6.4 MFLOP per agent-step (conv1 0.49 + conv2
        # 1.57 + fc1 4.19 + fc2/heads 0.13, both towers), fp32 on the MFMA / vector pipes (157.3 TFLOP/s dense)
        flops = 6.4e6 * N
        tick_s = elapsed / args.steps
        out["roofline_rollout"] = {
            "bound": "mfma", "achieved": flops / tick_s / 1e12, "peak": 157.3, "unit": "TFLOP/s",
            "frac": flops / tick_s / 1e12 / 157.3,
            "note": "policy FLOPs of one tick / the WHOLE tick time (env kernels, sampling and launch gaps "
                    "included): a lower bound on the policy kernels' own rate; fp32 in, fp32 accumulate"}
    if cpu_baseline_fn is not None and not args.no_cpu_baseline and world_size == 1:
        out["cpu_baseline"] = cpu_baseline_fn(args.scenario, args.worlds, args.robots_per_world, fidelity=args.fidelity)
        out["cpu_baseline"]["reference_structural_cap"] = "240 agent-steps/s (24 robots x 10 Hz, stageros.cpp:819-828)"
    if per_rank is not None:
        out["per_rank_agent_steps_per_s"] = per_rank      # each rank's own rate on the same per-GPU workload
        out["collective_backend"] = backend
        out["rccl_ranks_seen"] = ranks_seen               # sum over ranks of 1.0 through the same process group
        out["devices"] = devices
    out.update(extra)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--mode", default="env", choices=["env", "rollout", "train"])
    ap.add_argument("--worlds", type=int, default=None,
                    help="worlds per GPU (default: 128 Stage-1 rinks = configs[1] at every N; 187 with --scenario stage2)")
    ap.add_argument("--robots-per-world", type=int, default=32)
    ap.add_argument("--scenario", default=None, choices=["stage1", "stage2"],
                    help="default: stage1 (BASELINE configs[1]) at every N -- weak scaling of ONE per-GPU workload, so that "
                         "value(N) / (N x value(1)) is a scaling efficiency; stage2 = configs[2] / configs[3]'s per-GPU "
                         "workload (8192+ robots per GPU on the Stage-2 map)")
    ap.add_argument("--policy-dtype", default="f32", choices=["f32", "bf16"],
                    help="rollout/train: dtype of the policy INFERENCE pass (update stays fp32); f32 = the reference's")
    ap.add_argument("--update-dtype", default="f32", choices=["f32", "bf16"],
                    help="train: autocast dtype of the PPO update (master weights stay fp32); f32 = the reference's")
    ap.add_argument("--policy-path", default="fused", choices=["fused", "stock"],
                    help="rollout/train: policy INFERENCE through the fp32 HIP conv front end + batched GEMMs (fused, "
                         "default) or through the stock PyTorch layers (stock)")
    ap.add_argument("--update-path", default="fused", choices=["fused", "stock"],
                    help="train: the PPO update differentiates the conv front end through the HIP forward / backward "
                         "kernels (fused, default) or through the stock PyTorch layers / MIOpen (stock)")
    ap.add_argument("--no-graph", action="store_true", help="launch the tick kernel by kernel from the host instead of "
                                                             "replaying it as a hipGraph (env: 16 ticks per graph)")
    ap.add_argument("--no-gemm-choices", action="store_true",
                    help="rollout/train: let hipBLASLt's default heuristic pick the learner's GEMM kernels instead of the "
                         "recorded TunableOp choices (mrca/gemm_tuning.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the rollout/train side figures")
    ap.add_argument("--fidelity", action="store_true",
                    help="Stage's own resolutions (worlds/stage1.world:3): 0.2 m map cells, robots collide when their outlines "
                         "share a 0.2 m raster cell and see each other's bodies through that raster (a side line: `value` of "
                         "the default run is the exact-rectangle mode on 0.05 m cells)")
    ap.add_argument("--chains", type=int, default=None,
                    help="env mode: the worlds as this many world ranges, each a chain of (move launch, ray cast) on its own stream "
                         "/ graph branch, half a tick apart, so that one range's move launch runs next to another's ray cast "
                         "(TickSchedule; mrca_move_worlds / mrca_observe_worlds).  1 = every tick as two launches over all worlds "
                         "(rounds 1-4).  Default: 2")
    ap.add_argument("--schedule", default=None, choices=["native", "chained", "graph", "eager"],
                    help="env mode: how the timed ticks reach the GPU.  native (default): ONE mrca_step_many call, the library "
                         "enqueues every launch itself, the move launches running ahead of the ray casts on a stream of their own "
                         "(round 6); chained: the same call with round 5's schedule (one move-ray-move-ray chain per world range); "
                         "graph: replayed as hipGraphs captured from Python; eager (= --no-graph): launched tick by tick from Python")
    ap.add_argument("--graph", action="store_true",
                    help="(the default since round 4; kept for old command lines) env mode: the timed ticks are replayed as "
                         "hipGraphs -- 16 ticks per graph, one per entry of the action pool -- instead of being launched "
                         "kernel by kernel from the host; --no-graph launches eagerly")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU over RCCL, exactly what the
        # driver's `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` does) -- never a silent 1-GPU run
        # labelled as one.  A box with fewer GPUs than ranks is an error, not a smaller run.
        ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if ndev < args.gpus and os.environ.get("MRCA_BENCH_SAME_DEVICE") != "1":
            print(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) visible on this node: refusing to report a "
                  f"{args.gpus}-GPU figure from fewer devices", file=sys.stderr)
            raise SystemExit(2)
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # BASELINE.json: configs[1] (4096 robots, Stage-1 rink) is the per-GPU workload at EVERY N (weak scaling: rounds 1-2
    # switched to configs[3]'s Stage-2 workload at N > 1, which made value(N) / value(1) compare two different worlds).
    # configs[3] (65 536 robots over 8 GPUs = 8192+ per GPU on the Stage-2 map, ppo_stage2.py:32 / worlds/stage2.world)
    # is `--scenario stage2` (187 worlds x 44 robots per GPU).
    if args.scenario is None:
        args.scenario = "stage1"
    if args.worlds is None:
        args.worlds = 128 if args.scenario == "stage1" else 187
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda is unavailable and there is no CPU path")
    # test hooks (tests/test_gpu_bench_multirank.py): several ranks on ONE GPU over gloo, to exercise the
    # multi-rank code path on a 1-GPU box; production is one rank per GPU over RCCL
    same_device = os.environ.get("MRCA_BENCH_SAME_DEVICE") == "1"
    backend = os.environ.get("MRCA_BENCH_BACKEND", "nccl")
    dev_index = 0 if same_device else local_rank
    if dev_index >= torch.cuda.device_count():
        print(f"bench.py: rank {rank} wants cuda:{dev_index} but only {torch.cuda.device_count()} GPU(s) are visible",
              file=sys.stderr)
        raise SystemExit(2)
    torch.cuda.set_device(dev_index)
    # The library and the env come FIRST -- before the process group, its communicator and their streams exist: which hardware
    # queues the env's streams get depends on what the process has created before them (side_figure_main), and the state `value`
    # was measured in on one GPU (nothing but the env) is the state it should be measured in on eight.  Every rank may build
    # (the driver's run finds the library built; a cold tree is built once, under a file lock).
    import fcntl
    import __graft_entry__ as G
    with open(os.path.join(ROOT, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            G.build()
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    from mrca import scenario as S
    from mrca.vec_env import VecStageWorld

    if args.scenario == "stage1":
        sc = S.stage1(num_worlds=args.worlds, robots_per_world=args.robots_per_world, seed=1000 + rank,
                      stage_resolution=args.fidelity)
    else:
        sc = S.stage2(num_worlds=args.worlds, seed=1000 + rank, stage_resolution=args.fidelity)
    env = VecStageWorld(sc)
    N = sc.num_robots
    dev = env.device
    dist = None
    if world_size > 1:
        if args.mode == "env" and not args.no_graph and not args.graph and args.schedule in (None, "native", "chained"):
            # ... and its streams are USED once (a stream gets its hardware queue at its first launch)
            env.reset()
            env.step_many(action_pool(N, dev, 0, depth=2), 0, 2, args.chains or 2)
            torch.cuda.synchronize()
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    if args.gpus != world_size and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world_size}; using {world_size}", file=sys.stderr)
    # one entry per tick of warm-up + timed region (<= 1024): the actions are i.i.d. per tick, not a 16-deep loop
    pool = action_pool(N, dev, 1 + rank, depth=max(16, min(1024, args.warmup + args.steps)) if args.mode == "env" else 16)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    extra = {}
    sched = None
    if args.chains is None:
        # two ranges: one move launch apart IS half a chain's period, whatever the box (158 M over 2000 ticks, 140 M over 20).
        # Three ranges are one move launch apart as well -- 0.45 of a period where a third is wanted -- and end up between 151
        # and 168 M depending on how the chains settle (profiles/r05_g_*, r05_k_*, r05_l_region_sweep.txt): `--chains 3` to try
        args.chains = 2
    if args.mode == "env":
        # The tick replayed as hipGraphs (default) or launched from the host (--no-graph), as one chain over all worlds or as
        # --chains world ranges half a tick apart: TickSchedule above.
        if args.schedule is None:
            args.schedule = "eager" if args.no_graph else ("graph" if args.graph else "native")
        sched = TickSchedule(env, pool, chains=args.chains, graph=args.schedule == "graph",
                             native=args.schedule in ("native", "chained"), chained=args.schedule == "chained")
        step_fn = None
    else:
        from mrca import gemm_tuning
        from mrca.trainer import make_bench_step
        extra["gemm_choices"] = ("recorded TunableOp choices (mrca/data/gemm_choices_gfx950_rocm72.csv)"
                                 if (not args.no_gemm_choices and gemm_tuning.use_recorded_choices()) else
                                 "library default heuristic")
        step_fn = make_bench_step(env, args.mode, dist,
                                  inference_dtype=torch.bfloat16 if args.policy_dtype == "bf16" else None,
                                  update_dtype=torch.bfloat16 if args.update_dtype == "bf16" else None,
                                  fused=(args.policy_path == "fused" and args.policy_dtype == "f32"),
                                  graph=not args.no_graph,
                                  update_fused=(args.update_path == "fused" and args.update_dtype == "f32"))

    if args.mode == "train":
        # warm-up must cover whole horizons so that MIOpen tuning / allocator growth of the FIRST update
        # happen outside the timed region, and the timed region holds whole updates only
        hz = 128
        args.warmup = max(hz, (args.warmup + hz - 1) // hz * hz)
        args.steps = max(hz, (args.steps + hz - 1) // hz * hz)
    env.reset()
    if sched is not None:
        for k in range(3):                      # lazy initialisation outside any capture
            env.step(pool[k])
        torch.cuda.synchronize()
        sched.capture(0, args.warmup)           # every capture happens here (a capture replays nothing: the env stands still)
        sched.capture(args.warmup, args.steps)
        torch.cuda.synchronize()
        # a graph's FIRST launch uploads it to the device (tens of us): every graph the timed region replays is launched once
        # here, untimed and before the warm-up -- in a 20-step region (0.65 ms) the uploads are several per cent of the figure,
        # in a 1000-step one nothing.  These are extra untimed ticks (reported as `graph_prime_ticks`); the timed region stays
        # EXACTLY --steps ticks.
        extra["graph_prime_ticks"] = sched.prime(args.warmup, args.steps)
        if sched.native:
            # the same for the native schedule: the region's launches are enqueued once, untimed, before the warm-up -- the HIP
            # runtime grows its kernel-argument and signal pools the first time a stream sees that many launches in flight (a
            # 20-tick region of two ranges is 80), and a region that pays for that is 8 % slower than the next one of the
            # same length (tools/region_sweep.py vs a cold run: 140 vs 130 M).  Extra untimed ticks, reported here.
            reps = max(1, min(8, 256 // max(1, args.steps)))       # a short region a few times over: ~a quarter of a thousand ticks
            for _ in range(reps):
                sched.run(args.warmup, args.steps)
            extra["graph_prime_ticks"] = reps * args.steps
        torch.cuda.synchronize()
        sched.run(0, args.warmup)
        barrier()
        t0 = time.perf_counter()
        sched.run(args.warmup, args.steps)
        barrier()
        elapsed = time.perf_counter() - t0
        mv_ms, ray_ms, launches = 0.0, 0.0, 0
    else:
        replay_ticks = getattr(step_fn, "run_ticks", None)    # rollout as hipGraphs: eight ticks per replay (mrca/trainer.py)
        if replay_ticks is not None:
            replay_ticks(args.warmup)
        else:
            for k in range(args.warmup):
                step_fn(k)
        barrier()
        # begin / end stamps on the two launches of every 8th step of the timed region (stamped launches go through
        # hipExtLaunchKernel with four events: a few us of host time each -- taken on every step they cost a 20-step run a
        # fifth of `value`, round 3 measured)
        every = 8
        env.enable_timing(every)
        t0 = time.perf_counter()
        if replay_ticks is not None:
            replay_ticks(args.steps)
        else:
            for k in range(args.steps):
                step_fn(k)
        barrier()
        elapsed = time.perf_counter() - t0
        mv_ms, ray_ms, launches = env.read_timing()
        env.enable_timing(False)
    if step_fn is None:
        step_fn = lambda k: env.step(pool[k % len(pool)])  # noqa: E731  (the eager passes below)
    kernel_timing_note = "begin / end stamps of the launches (hipExtLaunchKernel) of every 8th step of the timed region"
    if 0 < launches < 16 and args.mode == "env":
        # a short region (the driver's 20 steps) leaves a handful of samples: add 64 more ticks with the events on EVERY
        # tick, after the timed region -- `value` does not see them, the kernel averages do
        env.enable_timing(1)
        for k in range(64):
            step_fn(k)
        torch.cuda.synchronize()
        mv2, ray2, l2 = env.read_timing()
        env.enable_timing(False)
        mv_ms, ray_ms, launches = mv_ms + mv2, ray_ms + ray2, launches + l2
        kernel_timing_note = ("begin / end stamps of the launches (hipExtLaunchKernel) of every 8th step of the timed region + "
                              "of each of 64 further ticks right after it (the region alone is too short for an average)")
    if launches == 0:
        # the tick was replayed as a hipGraph (the library's event records are not part of a captured tick): time the
        # two env kernels in a short eager pass AFTER the timed region instead
        env.enable_timing(1)
        for k in range(64):
            env.step(pool[k % len(pool)])
        torch.cuda.synchronize()
        mv_ms, ray_ms, launches = env.read_timing()
        env.enable_timing(False)
        kernel_timing_note = ("begin / end stamps of the env kernels' launches (hipExtLaunchKernel) in a separate eager pass of "
                              "64 ticks after the timed region")

    # side figure (single GPU, env mode only, outside the timed region above): the same world driven by the
    # fp32 policy instead of the action pool -- SURVEY 8d (ii).  `--mode rollout|train` time these properly.
    if args.mode == "env" and world_size == 1 and not args.no_extra:
        from mrca import gemm_tuning
        from mrca.trainer import make_bench_step
        # (as `--mode rollout` does: the recorded kernel choice for the fc1 GEMM -- switched on here, behind the timed region)
        recorded = (not args.no_gemm_choices) and gemm_tuning.use_recorded_choices()
        roll = make_bench_step(env, "rollout", None, fused=True, graph=True)
        roll.run_ticks(40)
        torch.cuda.synchronize()
        tr0 = time.perf_counter()
        n_roll = 400                        # (the region `--mode rollout` times by default: 40 warm-up ticks, 400 timed)
        roll.run_ticks(n_roll)
        torch.cuda.synchronize()
        v_roll = N * n_roll / (time.perf_counter() - tr0)
        extra["rollout_side_figure"] = {"value": v_roll, "unit": "agent-steps/s",
                                        "roofline": flop_roofline(v_roll, policy_flops(), "one fp32 CNNPolicy forward per agent-step "
                                                                  "(SURVEY 8d (ii))", _kernel_rates()[0]),
                                        "note": "env + fp32 CNNPolicy inference per tick (HIP conv front end, fc1 as a batched "
                                                "GEMM -- " + ("recorded TunableOp choice" if recorded else "library default heuristic") +
                                                " --, tail kernel; ticks replayed as hipGraphs of eight), 400 ticks after 40 "
                                                "warm-up ticks; not part of `value`"}

    # Side figures of the OTHER configurations DESIGN.md / README quote (each a fresh env, a few hundred ticks after the timed
    # region, same schedule as `value`; a failure is reported in its place and never costs the run its line):
    #   stage2_side_figure                 BASELINE configs[2] / the per-GPU share of configs[3]: 187 Stage-2 worlds x 44 robots
    #   fidelity_side_figure               this workload at Stage's own resolutions with raster collisions + raster lidar returns
    #   reference_shaped_obs_side_figure   this workload with lazy_obs = 0: every tick also forms MRCA_F_SCAN and the
    #                                      deque-ordered, normalised MRCA_F_OBS (SURVEY 8d's "obs normalise + frame-stack update")
    if args.mode == "env" and not args.no_extra and not args.fidelity:
        def side(name, note, ticks=300):
            # (a process of its own: side_figure_main says why)
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--side-figure", name, "--scenario", args.scenario, "--worlds",
                   str(args.worlds), "--robots-per-world", str(args.robots_per_world), "--chains", str(args.chains), "--schedule",
                   args.schedule, "--seed", str(1000 + rank), "--ticks", str(ticks), "--device-index", str(dev.index)]
            try:
                out = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
                lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
                if out.returncode != 0 or not lines:
                    raise RuntimeError(f"rc {out.returncode}: {out.stderr.strip().splitlines()[-1] if out.stderr.strip() else 'no output'}")
                extra[name] = json.loads(lines[-1])
                extra[name]["note"] = note + "; measured by a process of its own started from this run"
            except Exception as exc:
                extra[name] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        if args.scenario == "stage1":
            side("stage2_side_figure",
                 "BASELINE configs[2] (and configs[3]'s share of one GPU): 187 Stage-2 worlds x 44 robots = 8228 robots on the "
                 "800 x 800 obstacle map, group-synchronous episodes; not part of `value`", ticks=200)
        if world_size == 1:
            side("fidelity_side_figure",
                 "the same worlds in FIDELITY mode: Stage's own resolution (0.2 m, worlds/stage1.world:3), robots collide when "
                 "their outlines share a raster cell and see each other through that raster; not part of `value`")
            side("reference_shaped_obs_side_figure",
                 "the same workload with lazy_obs = 0: every tick also materialises MRCA_F_SCAN and the normalised, deque-ordered "
                 "MRCA_F_OBS [N,3,512] a reference-shaped caller reads (one more kernel per range and tick); not part of `value`")
        if dist is not None and "stage2_side_figure" in extra:
            # configs[3] = 65 536 robots over 8 GPUs: every rank ran its 8228 Stage-2 robots; the job's figure is all of them
            # over the slowest rank's time.  (Every rank takes part in the reduction whatever happened to its own figure: a
            # rank without one contributes +inf and the job's figure is reported as missing.)
            f = extra["stage2_side_figure"]
            tt = torch.tensor([f.get("ms_per_step", float("inf"))], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            if "value" in f and float(tt.item()) < float("inf"):
                f["per_rank_value"] = f["value"]
                f["value"] = f["robots"] * world_size / (float(tt.item()) * 1e-3)
                f["robots_all_ranks"] = f["robots"] * world_size
                f["note"] += f"; value = {world_size} ranks x {f['robots']} robots over the slowest rank's time"
            elif "value" in f:
                f["error"] = "another rank has no Stage-2 figure: the job's figure is missing (this rank's own: per_rank_value)"
                f["per_rank_value"] = f.pop("value")

    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    per_rank = None
    ranks_seen, devices = None, None
    if dist is not None:
        # evidence that the collective really spans the ranks: an all-reduce of ones, and every rank's device
        ones = torch.ones(1, device=dev, dtype=torch.float32)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        props = torch.cuda.get_device_properties(dev)
        mine_dev = f"rank {rank}: cuda:{dev.index} {props.name} ({getattr(props, 'gcnArchName', '?')}, " \
                   f"{props.multi_processor_count} CUs)"
        devices = [None] * world_size
        dist.all_gather_object(devices, mine_dev)
        mine = torch.tensor([N * args.steps / elapsed], device=dev, dtype=torch.float64)
        gathered = [torch.zeros_like(mine) for _ in range(world_size)]
        dist.all_gather(gathered, mine)
        per_rank = [float(x.item()) for x in gathered]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    total_robots = N * world_size
    value = total_robots * args.steps / elapsed


    import threading
    emit_lock = threading.RLock()
    emitted = [False]

    def emit():
        """rank 0: the ONE JSON line (everything it needs is final before the side figure below starts).  Main thread and
        the side figure's watchdog may both get here: whoever takes the lock first prints, the other returns."""
        with emit_lock:
            if emitted[0]:
                return
            emitted[0] = True
            _emit()

    def _emit():
        line = assemble_line(args=args, sc=sc, N=N, world_size=world_size, value=value, elapsed=elapsed, ray_ms=ray_ms, mv_ms=mv_ms,
                             launches=launches, kernel_timing_note=kernel_timing_note,
                             sched=None if sched is None else {"graph": sched.graph, "native": sched.native, "chains": sched.chains,
                                                               "chained": sched.chained},
                             extra=extra, per_rank=per_rank, ranks_seen=ranks_seen, devices=devices,
                             backend=dist.get_backend() if dist is not None else None, cpu_baseline_fn=cpu_baseline)
        print(json.dumps(line))

    # SURVEY 8d (iii), the full training loop, as a side figure at EVERY N (at N > 1 it is also what puts RCCL on the measured
    # path: the env tick itself needs no collective, so `value` alone would never touch it): `--mode train`'s configuration --
    # horizon 128 (ppo_stage1.py:24), two epochs (:30), minibatches of 16 384 rows per rank, fp32, the rollout tick as a
    # hipGraph, the update through the HIP front end / loss / heads / Adam kernels -- one warm-up update, then two timed.
    if args.mode == "env" and not args.no_extra and not args.fidelity:
        # (the first time RCCL runs on this code is the driver's SCALE run: a collective that never answers must not cost
        # the run its line -- after SIDE_FIGURE_TIMEOUT_S rank 0 prints the line without the side figure and every rank
        # leaves; `value` and the per-rank rates above are final before this starts)
        side_done = threading.Event()
        side_state = ["running"]          # guarded by emit_lock: "running" -> "done" (main thread) | "timed_out" (watchdog)

        def give_up():
            if side_done.wait(SIDE_FIGURE_TIMEOUT_S):
                return
            with emit_lock:
                if side_state[0] != "running":       # the side figure finished at the last moment: the main thread prints
                    return
                side_state[0] = "timed_out"
                extra["train_side_figure"] = {"error": f"no answer within {SIDE_FIGURE_TIMEOUT_S} s (a collective that hangs?)"}
                if rank == 0:
                    emit()
                sys.stdout.flush()
            os._exit(0)      # every rank's own watchdog does the same at the same time: nobody is left in a collective
        if world_size > 1:
            threading.Thread(target=give_up, daemon=True).start()
        try:
            from mrca import gemm_tuning
            from mrca.trainer import HParams, Stage1Trainer
            recorded = (not args.no_gemm_choices) and gemm_tuning.use_recorded_choices()
            hp = HParams(batch_size=16384, rollout_fused=True, graph_tick=True, update_fused=True)
            tr = Stage1Trainer(env, hp=hp, dist=dist, seed=0, stage2=False)
            tr.started = True
            tr.run(hp.horizon)                       # one warm-up update (MIOpen / allocator / RCCL set-up, graph captures)
            barrier()
            tt0 = time.perf_counter()
            n_upd = 2
            tr.run(n_upd * hp.horizon)
            barrier()
            dt_tr = time.perf_counter() - tt0
            v_tr = N * world_size * n_upd * hp.horizon / dt_tr
            # model FLOPs per agent-step: the rollout's forward + per epoch a forward and a backward (2 x forward) of the update
            f_step = policy_flops() * (1 + 3 * hp.epoch)
            side = {
                "value": v_tr, "unit": "agent-steps/s",
                "roofline": flop_roofline(v_tr / world_size, f_step, f"per GPU: rollout forward + {hp.epoch} epochs x (forward + backward) "
                                          "of the PPO update (SURVEY 8d (iii)); the last-value forward per horizon is not counted",
                                          _kernel_rates()[1]),
                "hparams": {"horizon": hp.horizon, "epoch": hp.epoch, "minibatch_per_rank": hp.batch_size, "dtype": "f32",
                            "gemm_choices": "recorded TunableOp choices" if recorded else "library default heuristic"},
                "collective": (None if dist is None else
                               {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                "gradient_bucket_bytes": int(tr.flat_grads.flat.numel() * 4),
                                "optimizer_steps": len(tr.loss_log) - len(tr.loss_log) // (n_upd + 1)}),
                "note": "env + fp32 policy + GAE + PPO update (ppo_stage1.py's horizon and epochs, minibatch 16384 per rank); at "
                        "N > 1 every optimiser step all-reduces the flat gradient bucket; not part of `value`"}
        except Exception as exc:      # a failure every rank shares (set-up, memory) must not cost the run its line
            side = {"error": f"{type(exc).__name__}: {exc}"[:400]}
        with emit_lock:               # timeout versus done is decided under the lock the line is printed under
            if side_state[0] == "running":
                side_state[0] = "done"
                extra["train_side_figure"] = side
        side_done.set()

    if rank == 0:
        emit()
        sys.stdout.flush()
    if dist is not None:
        # tidy shutdown, bounded: a peer that left through its watchdog must not hold this rank in the last barrier
        t_exit = threading.Timer(60.0, lambda: os._exit(0))
        t_exit.daemon = True
        t_exit.start()
        dist.barrier()
        dist.destroy_process_group()
        t_exit.cancel()


if __name__ == "__main__":
    if "--side-figure" in sys.argv[1:]:
        side_figure_main(sys.argv[1:])
    else:
        main()
