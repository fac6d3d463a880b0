"""HIP ops of the rollout path of the policy (csrc/mrca_policy.hip through the C ABI).  Inference only: the
training path keeps the stock PyTorch layers (BASELINE.json north_star: "the 1-D-conv policy kept in PyTorch").
There is no fallback: without the HIP library these raise."""
import ctypes as C

import torch

from . import _lib


def lidar_features(obs, w1, b1, w2, b2, out=None):
    """relu(conv2(relu(conv1(obs)))) of the actor and the critic tower in one launch.
    obs f32[N,3,512]; w1 f32[2,32,3,5], b1 f32[2,32], w2 f32[2,32,32,3], b2 f32[2,32] (tower-major: actor, critic)
    -> f32[2,N,4096], rows in the flatten order of [32,128]."""
    lib = _lib.load()
    N, F, B = obs.shape
    for t, shape in ((obs, (N, 3, 512)), (w1, (2, 32, 3, 5)), (b1, (2, 32)), (w2, (2, 32, 32, 3)), (b2, (2, 32))):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == shape):
            raise ValueError(f"lidar_features: expected a contiguous cuda float32 tensor of shape {shape}, got "
                             f"{tuple(t.shape)} {t.dtype} {t.device}")
    if out is None:
        out = torch.empty(2, N, 4096, dtype=torch.float32, device=obs.device)
    with torch.cuda.device(obs.device):
        stream = C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)
        _lib.check(lib.mrca_lidar_features(obs.data_ptr(), N, F, B, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                           b2.data_ptr(), out.data_ptr(), stream), "mrca_lidar_features")
    return out
