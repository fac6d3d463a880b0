"""Recorded GEMM kernel choices for the learner's plain library GEMMs (fc1 / fc2 / heads of ``CNNPolicy``, forward and
backward).

hipBLASLt's default heuristic picks poor kernels for the skinny shapes of the PPO update -- fc2's weight gradient
``[260 x 16384] x [16384 x 128]`` runs 124 us on 8 workgroups, its data gradient 117 us, the heads' 47 us: 0.67 ms of a
3.4 ms minibatch (profiles/r04_e_train_kernel_stats.csv).  PyTorch's TunableOp times every rocBLAS / hipBLASLt solution
for a shape once and records the winner; ``data/gemm_choices_gfx950_rocm72.csv`` is that record for the shapes of
``bench.py --mode train | rollout`` (4096 robots, 16 384-row minibatches), made on an MI355X by

    PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=out.csv python bench.py --mode train --no-graph

(fc2 wgrad 124 -> 39 us, dgrad 117 -> 16 us; train 1.86 -> 2.00 M agent-steps/s, profiles/r04_f_*).  The file carries the
library versions it was made with; PyTorch ignores it when they differ, and shapes it does not list take the default
kernel.  Still plain library GEMMs, selected -- nothing here computes anything.
"""
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_FILE = os.path.join(_HERE, "data", "gemm_choices_gfx950_rocm72.csv")


def use_recorded_choices(path=None, tune_missing=False):
    """Turn TunableOp on with the recorded choices (and, with ``tune_missing``, time the shapes the record does not have:
    seconds per new shape, written back to ``path`` + device ordinal at exit).  -> True when the record was accepted."""
    import torch
    if not torch.cuda.is_available() or os.environ.get("MRCA_NO_GEMM_CHOICES") == "1":
        return False
    t = torch.cuda.tunable
    t.enable(True)
    t.tuning_enable(bool(tune_missing))
    if tune_missing:
        # newly timed shapes are written at exit -- into `path` (+ the device ordinal PyTorch appends), not into a
        # tunableop_results<N>.csv in whatever the current directory happens to be
        try:
            t.set_filename(path or DEFAULT_FILE, insert_device_ordinal=True)
        except Exception:
            pass
    ok = False
    try:
        ok = bool(t.read_file(path or DEFAULT_FILE))
    except Exception:          # an unreadable record must never cost a run: default kernels
        ok = False
    if not ok and not tune_missing:
        t.enable(False)
    return ok
