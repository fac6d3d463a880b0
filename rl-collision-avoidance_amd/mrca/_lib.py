"""ctypes binding of libmrca_env.so (include/mrca_env.h).  There is no fallback: if the HIP
library is missing this module raises, loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PROFILING_LIB_PATH = os.path.join(_HERE, "libmrca_env_prof.so")
# MRCA_ENV_LIB selects another build of the SAME library (tools/ablate.py points it at the profiling build
# libmrca_env_prof.so); whichever it is, it must exist -- there is no fallback
LIB_PATH = os.environ.get("MRCA_ENV_LIB") or os.path.join(_HERE, "libmrca_env.so")

ABI_VERSION = 6
VIEW_SCAN, VIEW_OBS = 1, 2        # enum mrca_view

FIELDS = [  # order = enum mrca_field
    ("pose", "f32", 3), ("speed", "f32", 2), ("speed_gt", "f32", 2), ("goal", "f32", 2), ("init_pose", "f32", 3),
    ("scan", "f32", "B"), ("obs", "f32", "FB"), ("local_goal", "f32", 2), ("reward", "f32", 1), ("done", "u8", 1),
    ("result", "u8", 1), ("first_result", "u8", 1), ("crashed", "u8", 1), ("live", "u8", 1), ("fresh", "u8", 1),
    ("t", "i32", 1), ("episode", "i32", 1), ("prev_dist", "f32", 1),
    ("scan_ring", "f32", "FB"), ("ring_head", "u8", 1),
    ("hit_bits", "i64", "FW"),      # u64 [N,F,B/64] (torch has no arithmetic on uint64: the words are viewed as int64)
]

EXPORTS = ["mrca_abi_version", "mrca_last_error", "mrca_arena_bytes", "mrca_create", "mrca_destroy", "mrca_reset",
           "mrca_step", "mrca_step_slice", "mrca_step_worlds", "mrca_move_worlds", "mrca_observe_worlds", "mrca_step_many", "mrca_materialize", "mrca_newest_obs", "mrca_sparse_obs", "mrca_normalize_scans", "mrca_check", "mrca_get_field", "mrca_gae", "mrca_enable_timing", "mrca_read_timing",
           "mrca_event_pair_overhead",
           "mrca_lidar_features", "mrca_lidar_features_backward_scratch", "mrca_lidar_features_backward",
           "mrca_lidar_features_rows", "mrca_lidar_features_backward_rows",
           "mrca_policy_tail", "mrca_ppo_loss", "mrca_ppo_loss_scratch", "mrca_adam_step",
           "mrca_rollout_store_state", "mrca_rollout_store_outcome",
           "mrca_policy_heads", "mrca_policy_heads_backward_scratch", "mrca_policy_heads_backward", "mrca_relu_cat",
           "mrca_relu_cat_backward", "mrca_policy_heads_backward_bias", "mrca_relu_cat_backward_bias_scratch",
           "mrca_relu_cat_backward_bias"]


class RolloutRows(C.Structure):
    """include/mrca_env.h: mrca_rollout_rows"""
    _fields_ = [(k, C.c_void_p) for k in ("frames", "fidx", "cur", "goal", "speed", "action", "logprob", "value", "reward",
                                          "done")] + [("horizon", C.c_int32)]


class MrcaConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32), ("num_worlds", C.c_int32),
        ("robots_per_world", C.c_int32), ("beams", C.c_int32), ("frames", C.c_int32),
        ("map_width", C.c_int32), ("map_height", C.c_int32), ("map_words_per_row", C.c_int32),
        ("map_cell", C.c_float), ("map_x0", C.c_float), ("map_y0", C.c_float),
        ("map_bits", C.c_void_p),
        ("timeout", C.c_int32), ("w_thresh", C.c_float), ("pre_dist_zero", C.c_int32), ("auto_reset", C.c_int32),
        ("seed", C.c_uint64),
        ("reset_mode", C.c_void_p), ("goal_mode", C.c_void_p), ("init_table", C.c_void_p),
        ("goal_table", C.c_void_p), ("group_id", C.c_void_p),
        ("collision_raster", C.c_float), ("lazy_obs", C.c_int32), ("hold_velocity", C.c_int32),
    ]


_libs = {}


def load(path=None):
    """Load the in-tree HIP library (built by ``__graft_entry__.build()`` / csrc/build.sh).  ``path`` selects
    another build of the same ABI (the profiling build); each is loaded once per process."""
    path = path or LIB_PATH
    if path in _libs:
        return _libs[path]
    # torch FIRST.  The library needs libamdhip64.so.7; the torch wheel bundles its own HIP runtime under another file name
    # (torch/lib/libamdhip64.so).  Loaded after torch, the library binds to the runtime torch has already loaded (same SONAME):
    # one runtime in the process.  Loaded BEFORE torch it pulls in /opt/rocm's copy, torch then loads its own, and the second HSA
    # runtime to initialise finds "no ROCm-capable device" -- seen as a failing mrca_create on a healthy MI355X box when
    # __graft_entry__.build() (which ends in this call) ran before the first `import torch` of the process.
    import torch  # noqa: F401
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: the MI355X HIP library has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or csrc/build.sh). "
            "There is no CPU fallback.")
    lib = C.CDLL(path)
    lib.mrca_abi_version.restype = C.c_int
    lib.mrca_last_error.restype = C.c_char_p
    lib.mrca_arena_bytes.argtypes = [C.POINTER(MrcaConfig), C.POINTER(C.c_size_t)]
    lib.mrca_create.argtypes = [C.POINTER(MrcaConfig), C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    lib.mrca_destroy.argtypes = [C.c_void_p]
    lib.mrca_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mrca_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mrca_step_slice.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.mrca_step_worlds.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.mrca_move_worlds.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.mrca_observe_worlds.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.mrca_step_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.mrca_check.argtypes = [C.c_void_p, C.c_void_p]
    lib.mrca_materialize.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.mrca_newest_obs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mrca_normalize_scans.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.mrca_sparse_obs.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.mrca_get_field.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                   C.POINTER(C.c_size_t)]
    lib.mrca_gae.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int32,
                             C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mrca_lidar_features.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mrca_lidar_features_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 6
    lib.mrca_lidar_features_backward_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 11 + \
        [C.c_size_t, C.c_void_p]
    lib.mrca_lidar_features_backward_scratch.argtypes = [C.POINTER(C.c_size_t)]
    lib.mrca_lidar_features_backward.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 11 + \
        [C.c_size_t, C.c_void_p]
    lib.mrca_policy_tail.argtypes = [C.c_void_p] * 4 + [C.c_int32] + [C.c_void_p] * 16
    lib.mrca_ppo_loss_scratch.argtypes = [C.POINTER(C.c_size_t)]
    lib.mrca_ppo_loss.argtypes = [C.c_void_p] * 7 + [C.c_int32, C.c_float, C.c_float, C.c_float] + [C.c_void_p] * 4 + \
        [C.c_size_t, C.c_void_p]
    lib.mrca_rollout_store_state.argtypes = [C.c_void_p, C.POINTER(RolloutRows)] + [C.c_void_p] * 5
    lib.mrca_rollout_store_outcome.argtypes = [C.c_void_p, C.POINTER(RolloutRows)] + [C.c_void_p] * 3
    lib.mrca_policy_heads.argtypes = [C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 6 + [C.c_int32] + [C.c_void_p] * 3
    lib.mrca_policy_heads_backward_scratch.argtypes = [C.POINTER(C.c_size_t)]
    lib.mrca_policy_heads_backward.argtypes = [C.c_void_p] * 5 + [C.c_int32] + [C.c_void_p] * 3 + [C.c_int32] + \
        [C.c_void_p] * 4 + [C.c_size_t, C.c_void_p]
    lib.mrca_policy_heads_backward_bias.argtypes = [C.c_void_p] * 5 + [C.c_int32] + [C.c_void_p] * 3 + [C.c_int32] + \
        [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p]
    lib.mrca_relu_cat_backward_bias_scratch.argtypes = [C.POINTER(C.c_size_t)]
    lib.mrca_relu_cat_backward_bias.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                                C.c_void_p]
    lib.mrca_relu_cat.argtypes = [C.c_void_p] * 3 + [C.c_int32, C.c_void_p, C.c_void_p]
    lib.mrca_relu_cat_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.mrca_adam_step.argtypes = [C.c_void_p] * 4 + [C.c_int64] + [C.c_double] * 4 + [C.c_int32, C.c_void_p]
    lib.mrca_enable_timing.argtypes = [C.c_void_p, C.c_int32]
    if hasattr(lib, "mrca_set_debug_flags"):      # profiling build only
        lib.mrca_set_debug_flags.argtypes = [C.c_void_p, C.c_int32]
        lib.mrca_debug_move_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        lib.mrca_debug_ray_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        lib.mrca_debug_fwd_stamps.argtypes = [C.POINTER(C.c_double)]
    lib.mrca_event_pair_overhead.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_float)]
    lib.mrca_read_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    if lib.mrca_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path} ABI {lib.mrca_abi_version()} != binding ABI {ABI_VERSION}")
    _libs[path] = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = "; ".join(lib.mrca_last_error().decode("utf-8", "replace") for lib in _libs.values()
                        if lib.mrca_last_error())
        raise RuntimeError(f"{what} failed ({rc}): {msg}")
