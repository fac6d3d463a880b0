"""VecStageWorld -- the batched, device-resident replacement of N ``StageWorld`` objects.

One instance owns one libmrca_env handle on one GPU.  Every reference getter
(stage_world1.py:116-160) is a zero-copy torch view on a field of the env's device arena;
``step`` is one trip round the reference's loop body (ppo_stage1.py:75-91) for all robots at
once.  torch is used for device memory and streams only; all arithmetic runs in the HIP library.
"""
import ctypes as C
import operator

import numpy as np
import torch

from . import _lib
from .scenario import Scenario

_DTYPES = {"f32": torch.float32, "u8": torch.uint8, "i32": torch.int32, "i64": torch.int64}

RESULT_NAMES = {0: 0, 1: "Reach Goal", 2: "Crashed", 3: "Time out"}  # stage_world1.py:190-208


def sparse_beam_index(raw_beams, beam_num):
    """The beams get_laser_observation keeps when the world was constructed with ``beam_num`` != the lidar's sample count
    (stage_world1.py:126-139): a float64 index advanced by raw / beam_num per pick and truncated, beam_num / 2 picks
    ascending from beam 0 and beam_num / 2 descending from the last beam (the right half then reversed)."""
    step = float(raw_beams) / beam_num
    left, right = [], []
    index = 0.0
    for _ in range(int(beam_num / 2)):
        left.append(int(index))
        index += step
    index = raw_beams - 1.0
    for _ in range(int(beam_num / 2)):
        right.append(int(index))
        index -= step
    return np.asarray(left + right[::-1], np.int32)


class VecStageWorld:
    def __init__(self, scenario: Scenario, device=None, lib_path=None, lazy_obs=True):
        """``lazy_obs=False``: the library forms the two reference-shaped views (MRCA_F_SCAN, MRCA_F_OBS) inside every
        ``step`` / ``reset`` -- what a plain C caller written against the reference's getters expects (mrca_config.lazy_obs
        = 0: one more kernel per call); the default leaves them to the ``scan`` / ``obs`` properties."""
        if not torch.cuda.is_available():
            raise RuntimeError("VecStageWorld needs an MI355X (torch.cuda is unavailable); there is no CPU path")
        self.lib = _lib.load(lib_path)      # lib_path: another build of the same library (the profiling build)
        self.scenario = sc = scenario
        if device is None:
            index = torch.cuda.current_device()
        else:
            d = torch.device(device)
            index = d.index if d.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", index)
        self.N, self.R, self.W = sc.num_robots, sc.robots_per_world, sc.num_worlds
        self.B, self.F = sc.beams, sc.frames

        # host-side tables must outlive mrca_create only
        bits = np.ascontiguousarray(sc.grid.bits, dtype=np.uint32)
        reset_mode = np.ascontiguousarray(sc.reset_mode, np.int32)
        goal_mode = np.ascontiguousarray(sc.goal_mode, np.int32)
        group_id = np.ascontiguousarray(sc.group_id, np.int32)
        init_table = np.ascontiguousarray(sc.init_table, np.float32)
        goal_table = np.ascontiguousarray(sc.goal_table, np.float32)
        cfg = _lib.MrcaConfig(
            abi_version=_lib.ABI_VERSION, device=self.device.index, num_worlds=sc.num_worlds,
            robots_per_world=sc.robots_per_world, beams=sc.beams, frames=sc.frames,
            map_width=sc.grid.width, map_height=sc.grid.height, map_words_per_row=sc.grid.words_per_row,
            map_cell=sc.grid.cell, map_x0=sc.grid.x0, map_y0=sc.grid.y0, map_bits=bits.ctypes.data,
            timeout=sc.timeout, w_thresh=sc.w_thresh, pre_dist_zero=int(sc.pre_dist_zero),
            auto_reset=sc.auto_reset, seed=sc.seed, reset_mode=reset_mode.ctypes.data,
            goal_mode=goal_mode.ctypes.data, init_table=init_table.ctypes.data,
            goal_table=goal_table.ctypes.data, group_id=group_id.ctypes.data,
            collision_raster=float(getattr(sc, "collision_raster", 0.0)),
            hold_velocity=int(bool(getattr(sc, "hold_velocity", False))),
            lazy_obs=1 if lazy_obs else 0)   # 1: MRCA_F_OBS is materialised by the ``obs`` property when somebody asks for it
        nbytes = C.c_size_t()
        _lib.check(self.lib.mrca_arena_bytes(C.byref(cfg), C.byref(nbytes)), "mrca_arena_bytes")
        # the arena is a torch allocation so that every field is a plain torch view (zero copy)
        self.arena = torch.empty(nbytes.value, dtype=torch.uint8, device=self.device)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mrca_create(C.byref(cfg), self.arena.data_ptr(), nbytes.value, C.byref(handle)),
                       "mrca_create")
        self._h = handle
        for k, (name, dt, shape) in enumerate(_lib.FIELDS):
            ptr, off, nb = C.c_void_p(), C.c_size_t(), C.c_size_t()
            _lib.check(self.lib.mrca_get_field(self._h, k, C.byref(ptr), C.byref(off), C.byref(nb)), "mrca_get_field")
            assert ptr.value == self.arena.data_ptr() + off.value
            flat = self.arena[off.value: off.value + nb.value].view(_DTYPES[dt])
            if shape == "B":
                t = flat.view(self.N, self.B)
            elif shape == "FB":
                t = flat.view(self.N, self.F, self.B)
            elif shape == "FW":
                t = flat.view(self.N, self.F, self.B // 64)
            elif shape == 1:
                t = flat.view(self.N)
            else:
                t = flat.view(self.N, shape)
            setattr(self, "_" + name if name in ("obs", "scan") else name, t)
        self._eager_views = 0 if lazy_obs else (_lib.VIEW_SCAN | _lib.VIEW_OBS)   # views the library itself keeps current
        self._views_current = 0          # bits of _lib.VIEW_*: which of MRCA_F_SCAN / MRCA_F_OBS follow the ring right now

    # ------------------------------------------------------------------ the scans and the observation stack
    # The env keeps the last F scans of every robot as a RING of raw ranges (``scan_ring`` + ``ring_head``: a tick writes
    # ONE row per robot -- no shift, no second normalised copy) and makes the two reference-shaped views when somebody
    # reads them after a tick (mrca_materialize: one kernel on the current stream).  The fused policy path reads the ring.
    def _view(self, bit, what):
        if not self._views_current & bit:
            _lib.check(self.lib.mrca_materialize(self._h, bit, self._stream()), "mrca_materialize")
            self._views_current |= bit
        return what

    @property
    def scan(self):
        """f32[N,B] the newest scan of every robot (base_scan ranges, stageros.cpp:479-516)."""
        return self._view(_lib.VIEW_SCAN, self._scan)

    @property
    def obs(self):
        """f32[N,F,B] the observation stacks x / 6 - 0.5 in deque order (oldest frame first; what ``CNNPolicy.forward``
        eats: stage_world1.py:140, ppo_stage1.py:59-60,87-89)."""
        return self._view(_lib.VIEW_OBS, self._obs)

    @property
    def hit_robot(self):
        """bool[N,B]: beam b of robot n returned from ANOTHER ROBOT (the newest scan's words of MRCA_F_HIT_BITS, one bit per
        beam; clear = the floorplan or no return).  stageros publishes it as LaserScan.intensities: 1 floorplan, 0 robot or
        miss (stageros.cpp:501-506, ranger_return 0.5 cast to uint8)."""
        ar = torch.arange(self.N, device=self.device)
        words = self.hit_bits[ar, self.ring_head.long()]                       # i64 [N, B/64]
        bit = torch.arange(64, device=self.device, dtype=torch.int64)
        return ((words.unsqueeze(-1) >> bit) & 1).bool().reshape(self.N, self.B)

    def invalidate_views(self):
        """Tell the binding that the env moved on without it: ticks replayed as a hipGraph (or stepped by another binding of
        the same handle) advance the scan ring behind its back, so ``scan`` / ``obs`` must be formed again at their next
        read.  (With ``lazy_obs=False`` the replayed ticks re-formed them themselves.)"""
        self._views_current = self._eager_views

    @property
    def _obs_current(self):
        return bool(self._views_current)

    @_obs_current.setter
    def _obs_current(self, value):      # (older spelling of invalidate_views(), kept for callers written against it)
        if not value:
            self.invalidate_views()

    def policy_obs(self):
        """-> (ring, heads) for consumers that understand the ring: (scan_ring f32[N,F,B] of RAW ranges, RingHead)."""
        from .policy_ops import RingHead
        return self.scan_ring, RingHead(self.ring_head, raw=True)

    def newest_frame(self, out=None):
        """f32[N,B]: every robot's newest observation row x / 6 - 0.5 (the frame a tick appended), from the ring."""
        if out is None:
            out = torch.empty(self.N, self.B, dtype=torch.float32, device=self.device)
        elif not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.numel() == self.N * self.B):
            raise ValueError("newest_frame: out must be a contiguous cuda float32 tensor of N x B elements")
        _lib.check(self.lib.mrca_newest_obs(self._h, out.data_ptr(), self._stream()), "mrca_newest_obs")
        return out

    # ---- the learner's rollout buffer written by the library (include/mrca_env.h: mrca_rollout_rows)
    def rollout_rows(self, buf):
        """The C view of a one-frame-per-tick ``ppo.RolloutBuffer`` of this env's shape (checked here, once): hand it to
        ``rollout_store_state`` / ``rollout_store_outcome``.  Keeps the buffer's tensors alive."""
        T, N, F, B = buf.horizon, self.N, self.F, self.B
        want = {"frames": (torch.float32, (T + F - 1, N, B)), "fidx": (torch.int64, (T, N, F)), "_cur": (torch.int64, (N, F)),
                "goal": (torch.float32, (T, N, 2)), "speed": (torch.float32, (T, N, 2)), "action": (torch.float32, (T, N, 2)),
                "logprob": (torch.float32, (T, N, 1)), "value": (torch.float32, (T, N)), "reward": (torch.float32, (T, N)),
                "done": (torch.uint8, (T, N))}
        rows = _lib.RolloutRows()
        for name, (dtype, shape) in want.items():
            t = getattr(buf, name, None)
            if not (torch.is_tensor(t) and t.device == self.device and t.dtype == dtype and tuple(t.shape) == shape and t.is_contiguous()):
                raise ValueError(f"rollout_rows: buffer.{name} must be a contiguous {dtype} tensor of shape {shape} on {self.device}"
                                 + (f", got {tuple(t.shape)} {t.dtype} {t.device}" if torch.is_tensor(t) else ""))
            setattr(rows, name.lstrip("_"), t.data_ptr())
        rows.horizon = T
        rows._keep = buf
        return rows

    def rollout_store_state(self, rows, tick, action, logprob, value):
        """Row ``tick`` (int64[1] on the device) of the buffer before the env steps: newest frame, stack rows, goal, speed and
        the policy's ``action`` f32[N,2] / ``logprob`` / ``value`` (N elements each) -- one launch."""
        if not (tick.is_cuda and tick.dtype == torch.int64 and tick.numel() == 1):
            raise ValueError("rollout_store_state: tick must be an int64[1] cuda tensor")
        _lib.check(self.lib.mrca_rollout_store_state(self._h, C.byref(rows), tick.data_ptr(), self._ptr(action, torch.float32, 2 * self.N),
                                                     self._ptr(logprob, torch.float32, self.N), self._ptr(value, torch.float32, self.N),
                                                     self._stream()), "mrca_rollout_store_state")

    def rollout_store_outcome(self, rows, tick, ticket):
        """... and after it stepped: reward / done into row ``tick``, then ``tick += 1`` on the device -- one launch.
        ``ticket``: a zeroed int32[1] cuda tensor of the caller's, the same one for every call."""
        if not (tick.is_cuda and tick.dtype == torch.int64 and tick.numel() == 1 and ticket.is_cuda and ticket.dtype == torch.int32
                and ticket.numel() == 1):
            raise ValueError("rollout_store_outcome: tick int64[1] and ticket int32[1] cuda tensors")
        _lib.check(self.lib.mrca_rollout_store_outcome(self._h, C.byref(rows), tick.data_ptr(), ticket.data_ptr(), self._stream()),
                   "mrca_rollout_store_outcome")

    def sparse_obs(self, beam_num):
        """f32[N,F,beam_num]: the observation stacks a ``StageWorld(beam_num, ...)`` with beam_num != 512 would build
        (stage_world1.py:126-140), formed on the device from the ring."""
        cache = self.__dict__.setdefault("_sparse_index", {})
        if beam_num not in cache:
            cache[beam_num] = torch.from_numpy(sparse_beam_index(self.B, beam_num)).to(self.device)
        idx = cache[beam_num]
        out = torch.empty(self.N, self.F, idx.numel(), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.mrca_sparse_obs(self._h, idx.data_ptr(), idx.numel(), out.data_ptr(), self._stream()),
                   "mrca_sparse_obs")
        return out

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            torch.cuda.synchronize(self.device)
            self.lib.mrca_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _ptr(t, dtype, numel):
        if t is None:
            return None
        if not (t.is_cuda and t.dtype == dtype and t.is_contiguous() and t.numel() == numel):
            raise ValueError(f"expected a contiguous cuda {dtype} tensor with {numel} elements")
        return C.c_void_p(t.data_ptr())

    # ------------------------------------------------------------------ the StageWorld surface, batched
    def reset(self, mask=None, poses=None, goals=None):
        """reset_pose + generate_goal_point (+ first observation) for the masked robots
        (stage_world1.py:171-177,213-223; ppo_stage1.py:52-62)."""
        _lib.check(self.lib.mrca_reset(self._h, self._ptr(mask, torch.uint8, self.N),
                                       self._ptr(poses, torch.float32, self.N * 3),
                                       self._ptr(goals, torch.float32, self.N * 2), self._stream()), "mrca_reset")
        self._views_current = self._eager_views
        return self

    def step(self, actions, ray_slice=None, worlds=None):
        """control_vel + one Stage tick + get_reward_and_terminate + next observation for every
        robot (ppo_stage1.py:75-91).  ``actions`` f32[N,2] = clipped (v, omega).
        ``ray_slice=(first, count)``: one world sharded over ranks (mrca_step_slice) -- all robots advance, the
        lidar is cast for robots [first, first + count) only.
        ``worlds=(first, count)``: only these worlds tick (mrca_step_worlds), on the current stream; disjoint ranges may be
        stepped on different streams at the same time (worlds never interact)."""
        a = self._ptr(actions, torch.float32, self.N * 2)
        if worlds is not None:
            if ray_slice is not None:
                raise ValueError("step: ray_slice and worlds exclude each other")
            _lib.check(self.lib.mrca_step_worlds(self._h, a, int(worlds[0]), int(worlds[1]), self._stream()),
                       "mrca_step_worlds")
            self._views_current = 0       # (lazy_obs=False re-formed the views of this range only)
        elif ray_slice is None:
            _lib.check(self.lib.mrca_step(self._h, a, self._stream()), "mrca_step")
            self._views_current = self._eager_views
        else:
            _lib.check(self.lib.mrca_step_slice(self._h, a, int(ray_slice[0]), int(ray_slice[1]), self._stream()),
                       "mrca_step_slice")
            self._views_current = 0
        return self

    def move(self, actions, worlds):
        """First launch of ``step(actions, worlds=...)``: control_vel, the Stage tick, reward / terminal and episode
        bookkeeping of the worlds (first, count) (mrca_move_worlds).  ``observe`` must follow before the next ``move``."""
        _lib.check(self.lib.mrca_move_worlds(self._h, self._ptr(actions, torch.float32, self.N * 2), int(worlds[0]),
                                             int(worlds[1]), self._stream()), "mrca_move_worlds")
        self._views_current = 0
        return self

    def observe(self, worlds):
        """Second launch: the lidar scans and local goals of the worlds (first, count) at their current poses
        (mrca_observe_worlds)."""
        _lib.check(self.lib.mrca_observe_worlds(self._h, int(worlds[0]), int(worlds[1]), self._stream()),
                   "mrca_observe_worlds")
        self._views_current = 0
        return self

    def step_many(self, actions, first_tick, num_ticks, chains=1):
        """``num_ticks`` ticks from ONE call: tick k takes ``actions[(first_tick + k) % len(actions)]`` (a list of f32[N,2]
        device tensors: a scripted scenario, the benchmark's action pool).  Equal to ``num_ticks`` calls of ``step``.
        ``chains`` = P > 0: the library's run-ahead schedule -- the move launches on a stream of the env's own, ahead of the ray
        casts of P world ranges on P streams (mrca_step_many); ``chains`` = -P: round 5's P chains half a tick
        apart.  The first call on a stream spends ~1 ms warming and checking the env's streams (and synchronises once)."""
        # (the pointer table of a list is built once per list object: a 20-tick region is 0.6 ms, sixteen data_ptr() calls and
        # their checks are 2 % of it.  The entry keeps the tensors themselves: an element replaced in place -- pool[i] = t --
        # fails the identity compare below and rebuilds the table, and a table never outlives the memory it points at)
        cache = self.__dict__.setdefault("_many_ptrs", {})
        entry = cache.get(id(actions))
        if entry is None or entry[0] is not actions or len(actions) != entry[2] or \
                not all(map(operator.is_, actions, entry[3])):
            ptrs = [self._ptr(t, torch.float32, self.N * 2).value for t in actions]
            cache.clear()
            entry = cache[id(actions)] = (actions, (C.c_void_p * len(actions))(*ptrs), len(actions), tuple(actions))
        rc = self.lib.mrca_step_many(self._h, entry[1], entry[2], int(first_tick), int(num_ticks), int(chains), self._stream())
        if rc:
            _lib.check(rc, "mrca_step_many")
        self._views_current = self._eager_views if chains <= 1 else 0
        return self

    def ring_ranges(self):
        """f32[N,F,B]: a copy of the scan ring.  (ABI 4-5 kept what a beam hit in the sign bit of its ring entry and this
        accessor stripped it; since ABI 6 the flag is ``hit_bits`` / ``hit_robot`` and the ring holds plain ranges -- kept
        for the callers written against it.)"""
        return self.scan_ring.abs()

    def check(self):
        """Host round trip: raises if a kernel flagged a device-side failure since the last check (mrca_check)."""
        _lib.check(self.lib.mrca_check(self._h, self._stream()), "mrca_check")

    # ------------------------------------------------------------------ timing (bench / profiles)
    def enable_timing(self, on=True):
        _lib.check(self.lib.mrca_enable_timing(self._h, int(on)), "mrca_enable_timing")

    def set_debug_flags(self, flags):
        """Profiling build only (MRCA_ENV_LIB=.../libmrca_env_prof.so, include/mrca_env.h): ablation switches and
        launch-shape knobs of the kernels.  The product library does not have them."""
        if not hasattr(self.lib, "mrca_set_debug_flags"):
            raise RuntimeError("this is the product build of libmrca_env.so: ablation switches exist only in the "
                               "profiling build (csrc/build.sh --profiling, MRCA_ENV_LIB=...)")
        _lib.check(self.lib.mrca_set_debug_flags(self._h, int(flags)), "mrca_set_debug_flags")

    def event_pair_overhead(self, samples=200):
        """us an empty HIP-event pair reads on the current stream: contained once per kernel in ``read_timing``'s figures."""
        us = C.c_float()
        _lib.check(self.lib.mrca_event_pair_overhead(self._stream(), int(samples), C.byref(us)), "mrca_event_pair_overhead")
        return us.value

    def read_timing(self):
        mv, ry, n = C.c_float(), C.c_float(), C.c_int32()
        _lib.check(self.lib.mrca_read_timing(self._h, C.byref(mv), C.byref(ry), C.byref(n)), "mrca_read_timing")
        return mv.value, ry.value, n.value


def gae(rewards, values, last_value, dones, gamma, lam):
    """generate_train_data (model/ppo.py:122-139) on device: rewards/values f32[T,N], last_value
    f32[N], dones u8[T,N] -> (targets, advs) f32[T,N]."""
    lib = _lib.load()
    T, N = rewards.shape
    for t, dt in ((rewards, torch.float32), (values, torch.float32), (last_value, torch.float32),
                  (dones, torch.uint8)):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == dt):
            raise ValueError("gae: expected contiguous cuda tensors (f32, f32, f32, u8)")
    targets = torch.empty_like(rewards)
    advs = torch.empty_like(rewards)
    stream = C.c_void_p(torch.cuda.current_stream(rewards.device).cuda_stream)
    _lib.check(lib.mrca_gae(rewards.data_ptr(), values.data_ptr(), last_value.data_ptr(), dones.data_ptr(),
                            float(gamma), float(lam), T, N, targets.data_ptr(), advs.data_ptr(), stream), "mrca_gae")
    return targets, advs
