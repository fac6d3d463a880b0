"""Properties of the compiled gfx950 code that the design relies on (no GPU needed: hipcc cross-compiles).

* fp32 determinism: the tick is specified as separately rounded IEEE operations, so the device code may
  contain fused multiply-adds ONLY inside the compiler's division / square-root expansions (fp32, and the
  float-assisted expansion of integer division) and in the one place the source asks for them: norm_obs'
  x/6 - 0.5 (mrca_device.h; two explicit fmaf with the literals -6 and RN(1/6), proven equal to the IEEE
  division over the whole range)
  (-ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt).
* occupancy: the ray-cast kernel must fit 8 waves per SIMD (<= 64 VGPRs) and no kernel may spill to scratch.
"""
import bisect
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rl-collision-avoidance_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "mrca_kernels.s"
    # the flags of csrc/build.sh that shape the device code
    flags = re.findall(r"^\s+(-f[\w=-]+(?:\s+-f[\w=-]+)*)", open(os.path.join(CSRC, "build.sh")).read(), re.M)
    flags = " ".join(flags).split()
    assert "-ffp-contract=off" in flags and "-fhip-fp32-correctly-rounded-divide-sqrt" in flags, flags
    # + the per-file flags of mrca_kernels.hip (the kernel-argument preload)
    per = re.search(r'"\$\{src\}" == "mrca_kernels" \]\] && per=\(([^)]*)\)', open(os.path.join(CSRC, "build.sh")).read())
    assert per, "build.sh no longer gives mrca_kernels.hip its per-file flags"
    flags += per.group(1).split()
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", *flags, "-S", "--cuda-device-only",
                    os.path.join(CSRC, "mrca_kernels.hip"), "-o", str(out)], check=True, capture_output=True)
    return open(out).read().split("\n")


def test_fused_multiply_adds_only_inside_division_and_sqrt_expansions(device_asm):
    fused = [i for i, l in enumerate(device_asm)
             if re.match(r"\s+v_(pk_fma|fma|fmac|mad|mac|madmk|madak|fmamk|fmaak)_(legacy_)?f(16|32)", l)]
    anchors = [i for i, l in enumerate(device_asm)
               if re.match(r"\s+v_(div_scale|div_fmas|div_fixup|rsq|sqrt|rcp|rcp_iflag)_f32", l)]
    assert fused and anchors
    # norm_obs (x / 6 - 0.5, two explicit __builtin_fmaf): since ABI 4 only the READERS of the scan ring form it --
    # materialize_kernel, newest_obs_kernel, normalize_kernel and sparse_obs_kernel, mostly four floats at a time, which the
    # compiler packs (v_pk_mul / v_pk_fma with
    # RN(1/6) and 6.0 in scalar registers).  Every fused operation inside those two kernels is one of these; no ray-cast
    # variant holds any (it stores raw ranges).
    starts = [(i, m.group(1)) for i, l in enumerate(device_asm) for m in [re.match(r"(_Z\w+):\s*(;.*)?$", l)] if m]
    ends = [i for i, l in enumerate(device_asm) if re.match(r"\.Lfunc_end\d+:", l)]

    def owner(i):
        k = bisect.bisect_right([a for a, _ in starts], i) - 1
        return starts[k][1] if k >= 0 and any(starts[k][0] < e and i < e for e in ends) else ""
    # (since round 6 also the VIEWS instantiations of the ray cast -- raycast_kernel<..., true>: with lazy_obs = 0 its epilogue
    # writes MRCA_F_OBS itself -- and ONLY those: the default kernel stores raw ranges)
    def is_reader(o):
        return any(r in o for r in ("materialize_kernel", "newest_obs_kernel", "normalize_kernel", "sparse_obs_kernel")) or \
            bool(re.search(r"raycast_kernelILi\dELb[01]ELb[01]ELi\dELb1E", o))
    norm = [i for i in fused if is_reader(owner(i))]
    assert norm and all(re.match(r"\s+v_(pk_fma|fma|fmac|fmamk|fmaak)_f32", device_asm[i]) for i in norm), len(norm)
    assert {o for o in map(owner, norm)} == {o for _, o in starts if is_reader(o)}
    assert not any("raycast_kernel" in owner(i) and not is_reader(owner(i)) and
                   ("0xc0c00000" in device_asm[i] or "0x3e2aaaab" in device_asm[i]) for i in fused)
    fused = [i for i in fused if i not in set(norm)]
    for i in fused:
        k = bisect.bisect_left(anchors, i)
        dist = min(abs(anchors[j] - i) for j in (k - 1, k) if 0 <= j < len(anchors))
        assert dist <= 30, f"fused multiply-add outside a div/sqrt expansion: line {i}: {device_asm[i].strip()}"


def test_register_budget_and_no_scratch(device_asm):
    text = "\n".join(device_asm)
    kernels = re.findall(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S)
    # move, materialize, newest_obs, normalize, sparse_obs, head_init, reset, gae, 7 x bw_* (integrate, collide, finish, lidar count / scan x 2 /
    # fill), raycast<1 | 2 lock-step | 2 sequential | 4 lock-step | 4 sequential> + big-world <1 | 2 | 4 lock-step, 2 | 4 sequential> + fidelity mode
    # <1 | 2 seq> x outline windows of <4 | 8> cells per side; every ray cast twice: without / with the lazy_obs = 0 views in its epilogue
    assert len(kernels) == 43, [k for k, _ in kernels]
    for name, body in kernels:
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        assert scratch == 0, f"{name} spills {scratch} B/lane to scratch"
        ray = re.search(r"raycast_kernelILi(\d)ELb([01])ELb([01])ELi(\d)E", name)      # <K, BIG, SEQ, RKW>
        assert ("raycast_kernel" in name) == bool(ray), name
        if ray and ((ray.group(1) == "4" and ray.group(3) == "0") or ray.group(2) == "1" or
                    (ray.group(1) == "1" and ray.group(4) == "8")):
            # 4 rays in lock step / big worlds / fewer than 256 beams at a raster below 0.195 m: not the hot shapes
            assert vgpr <= 128, f"{name} needs {vgpr} VGPRs"
        elif ray:
            assert vgpr <= 64, f"{name} needs {vgpr} VGPRs: fewer than 8 waves per SIMD"
        else:
            assert vgpr <= 128, f"{name} needs {vgpr} VGPRs"


def test_env_kernels_get_their_leading_arguments_preloaded(device_asm):
    """move_kernel and every raycast_kernel lead with the scalars / pointers their first loads need, and the build asks for
    gfx950's kernel-argument preload: the descriptor of each must say so (14 dwords for the ray cast; 14 for the move
    kernel since round 6: R, the first world of the launch + SIX pointers -- what the tick reads of the tick before comes
    through arguments of its own, mrca_step_many's run-ahead slots), or the loads wait for the s_load of the EnvView again
    (DESIGN.md 5.3)."""
    text = "\n".join(device_asm)
    lengths = {m.group(1): int(m.group(2)) for m in
               re.finditer(r"\.amdhsa_kernel (\S+).*?\.amdhsa_user_sgpr_kernarg_preload_length (\d+)", text, re.S)}
    ray = [v for k, v in lengths.items() if "raycast_kernel" in k]
    move = [v for k, v in lengths.items() if "move_kernel" in k]
    assert len(ray) == 28 and all(v == 14 for v in ray), lengths
    assert move == [14], lengths
