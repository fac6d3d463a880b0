"""CPU-side checks of the C ABI: the library builds for gfx950 without a GPU, loads, exports
every symbol include/mrca_env.h declares, and validates configs without touching a device."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

import util as U
from util import S


def _declared_symbols():
    hdr = open(os.path.join(U.ROOT, "include", "mrca_env.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    hdr = re.sub(r"#ifdef MRCA_PROFILING.*?#endif", "", hdr, flags=re.S)     # not part of the product ABI
    return sorted(set(re.findall(r"\b(mrca_[a-z_]+)\s*\(", hdr)))


def test_header_symbols_exported(built_lib):
    names = _declared_symbols()
    assert "mrca_step" in names and "mrca_create" in names and len(names) >= 10
    for n in names:
        assert hasattr(built_lib, n), f"libmrca_env.so does not export {n}"
    from mrca import _lib
    assert sorted(_lib.EXPORTS) == names
    # the ablation switches live in the profiling build only (csrc/build.sh --profiling)
    assert not hasattr(built_lib, "mrca_set_debug_flags")


def test_abi_version(built_lib):
    from mrca import _lib
    assert built_lib.mrca_abi_version() == 6 == _lib.ABI_VERSION


def _cfg(sc):
    from mrca import _lib
    bits = np.ascontiguousarray(sc.grid.bits, np.uint32)
    cfg = _lib.MrcaConfig(abi_version=_lib.ABI_VERSION, device=0, num_worlds=sc.num_worlds,
                          robots_per_world=sc.robots_per_world, beams=sc.beams, frames=sc.frames,
                          map_width=sc.grid.width, map_height=sc.grid.height,
                          map_words_per_row=sc.grid.words_per_row, map_cell=sc.grid.cell, map_x0=sc.grid.x0,
                          map_y0=sc.grid.y0, map_bits=bits.ctypes.data, timeout=sc.timeout, w_thresh=sc.w_thresh,
                          pre_dist_zero=int(sc.pre_dist_zero), auto_reset=sc.auto_reset, seed=sc.seed)
    return cfg, bits


def test_arena_bytes_and_layout(built_lib):
    sc = S.stage1(num_worlds=128, robots_per_world=32)
    cfg, _keep = _cfg(sc)
    n = C.c_size_t()
    assert built_lib.mrca_arena_bytes(C.byref(cfg), C.byref(n)) == 0
    N = sc.num_robots
    need = N * 512 * 4 * (1 + 3 + 3)  # the ring of raw scans + the two materialised views (newest scan, normalised stack)
    assert need < n.value < need + 3 * 1024 * 1024      # + the quadrant free-rectangle field: 8 B per cell of the map
    assert n.value % 256 == 0


@pytest.mark.parametrize("field,value,code", [
    ("robots_per_world", 0, -1), ("beams", 500, -1), ("frames", 0, -1),
    ("abi_version", 99, -1), ("map_cell", 0.0, -1), ("auto_reset", 7, -1), ("num_worlds", 0, -1),
    ("map_width", 0, -1), ("map_height", 20000, -4),
])
def test_config_validation_errors(built_lib, field, value, code):
    cfg, _keep = _cfg(S.stage1(num_worlds=2, robots_per_world=8))
    setattr(cfg, field, value)
    n = C.c_size_t()
    assert built_lib.mrca_arena_bytes(C.byref(cfg), C.byref(n)) == code
    assert len(built_lib.mrca_last_error()) > 0


def test_big_worlds_validate(built_lib):
    """More than 64 robots per world: accepted with per-robot or no restarts, refused with group-synchronous ones."""
    cfg, _keep = _cfg(S.stage1(num_worlds=2, robots_per_world=300))
    n = C.c_size_t()
    assert built_lib.mrca_arena_bytes(C.byref(cfg), C.byref(n)) == 0 and n.value > 600 * 512 * 16
    cfg.auto_reset = 2
    assert built_lib.mrca_arena_bytes(C.byref(cfg), C.byref(n)) == -4


def test_product_has_no_cpu_fallback():
    """Constructing the env without a GPU must fail loudly, and the package never imports oracle/."""
    import torch
    from mrca.vec_env import VecStageWorld
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU path"):
            VecStageWorld(S.stage1(num_worlds=1, robots_per_world=2))
    pkg = os.path.join(U.ROOT, "rl-collision-avoidance_amd")
    for dp, _dn, fn in os.walk(pkg):
        for f in fn:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                txt = open(os.path.join(dp, f)).read()
                assert "mrca_oracle" not in txt and "import oracle" not in txt, os.path.join(dp, f)


def test_committed_pmc_traffic_matches_the_kernel_sources():
    """profiles/pmc_traffic.json feeds roofline.traffic of bench.py and is refused when it was measured on other kernel
    code: the committed file must belong to the committed sources (re-run the PMC passes after touching a kernel), and
    the stamp must ignore comments but not code."""
    import importlib
    import sys
    sys.path.insert(0, U.ROOT)
    bench = importlib.import_module("bench")
    traffic, note = bench.pmc_traffic(4096, "stage1")
    assert traffic is not None and 10e6 < traffic < 20e6, note      # ABI 4: ONE 2 kB row per robot (round 3: 21 MB)
    src = open(os.path.join(U.ROOT, "rl-collision-avoidance_amd", "csrc", "mrca_kernels.h")).read()
    assert bench.code_only(src + "\n// a comment\n/* another\n one */\n") == bench.code_only(src)
    assert bench.code_only(src + "\nstatic const int kNotThere = 1;\n") != bench.code_only(src)
    assert "//" not in bench.code_only(src) and len(bench.code_only(src)) > 1000


def test_bench_refuses_a_multi_gpu_run_it_cannot_start():
    """`python bench.py --gpus N` without torchrun starts its own N ranks -- and where fewer than N GPUs are visible (here:
    none) it must say so and exit non-zero instead of printing a line labelled with fewer GPUs."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MRCA_BENCH_SAME_DEVICE")}
    out = subprocess.run([sys.executable, os.path.join(U.ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True,
                         text=True, timeout=300)
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("this box really has 8 GPUs")
    assert out.returncode == 2 and "refusing" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_library_loaded_before_torch_still_shares_torchs_hip_runtime():
    """mrca._lib.load() imports torch before it dlopens libmrca_env.so: whatever the order of the caller's imports, ONE HIP runtime is
    mapped into the process (two -- /opt/rocm's next to the torch wheel's -- and the second to initialise sees no device:
    mrca_create failed on the GPU box when build() ran before `import torch`)."""
    import subprocess
    import sys
    pkg = os.path.join(U.ROOT, "rl-collision-avoidance_amd")
    code = "\n".join([
        "import sys",
        f"sys.path.insert(0, {pkg!r})",
        "from mrca import _lib",
        "_lib.load()",
        "import torch",
        "maps = {l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}",
        "print(len(maps), sorted(maps))"])
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-500:]
    assert out.stdout.split()[0] == "1", out.stdout
