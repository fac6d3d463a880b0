"""HIP ops of the policy's lidar front end (csrc/mrca_policy.hip, csrc/mrca_policy_bwd.hip through the C ABI): the
fused forward of both towers' Conv1d + ReLU pairs (rollout AND update) and its backward pass, tied together as a
``torch.autograd.Function`` (``lidar_features_fn``) so that the PPO update (model/ppo.py:158-192) differentiates through
hand-written fp32 MFMA kernels instead of MIOpen.  The rest of the policy stays stock PyTorch.
There is no fallback: without the HIP library these raise."""
import ctypes as C

import numpy as np
import torch

from . import _lib


class RingHead:
    """The head slots of a frame ring plus what the ring holds: ``raw`` = raw lidar ranges (the env's MRCA_F_SCAN_RING;
    the front end applies x / 6 - 0.5 while it stages) or normalised observations.  Travels opaquely through
    ppo.generate_action / CNNPolicy.act_fused as their ``head`` argument; a plain u8 tensor means "normalised"."""
    __slots__ = ("slots", "raw")

    def __init__(self, slots, raw):
        self.slots, self.raw = slots, bool(raw)

    def __getitem__(self, idx):
        return RingHead(self.slots[idx], self.raw)


def unwrap_head(head):
    """-> (u8 tensor or None, raw)"""
    if isinstance(head, RingHead):
        return head.slots, head.raw
    return head, False


def normalize_scans(scans):
    """x / 6 - 0.5 (stage_world1.py:140) of raw lidar ranges, rounded exactly as the env's own MRCA_F_OBS view is
    (mrca_normalize_scans).  torch's ``x / 6.0`` is NOT that on the GPU: a scalar divisor becomes a multiplication by
    RN(1/6).  CPU tensors (the tests' stand-ins) take numpy's correctly rounded quotient, which equals the kernel's
    result."""
    if not scans.is_cuda:
        a = np.abs(scans.numpy())           # (|x| as the kernel does: rows stored under ABI 4-5 carried a flag in the sign bit)
        return torch.from_numpy(a / a.dtype.type(6.0) - a.dtype.type(0.5))
    x = scans.contiguous()
    if x.dtype != torch.float32 or x.numel() % 4:
        raise ValueError("normalize_scans: expected a float32 tensor with a multiple of 4 elements")
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(_lib.load().mrca_normalize_scans(x.data_ptr(), out.data_ptr(), x.numel(), stream), "mrca_normalize_scans")
    return out


class FrameTable:
    """Observation stacks addressed through a row table instead of gathered: ``frames`` f32[*, 512] (a matrix of normalised
    frames: the rollout buffer's one-frame-per-tick store, flattened) and ``rows`` i32[n, 3], the row of each sample's three
    frames, oldest first.  What ``lidar_features_fn`` takes in place of a [n, 3, 512] tensor (mrca_lidar_features_rows):
    the minibatch's copy of its stacks -- 100 MB written and read per 16 384 samples -- is never made.  ``gather()`` makes it
    for whoever needs a tensor (the stock policy path)."""

    def __init__(self, frames, rows):
        if not (frames.dtype == torch.float32 and frames.is_contiguous() and frames.dim() == 2 and frames.shape[1] == 512):
            raise ValueError("FrameTable: frames must be a contiguous float32 matrix [*, 512]")
        if not (rows.dtype == torch.int32 and rows.is_contiguous() and rows.dim() == 2 and rows.shape[1] == 3 and
                rows.device == frames.device):
            raise ValueError("FrameTable: rows must be a contiguous int32 tensor [n, 3] on the frames' device")
        self.frames, self.rows = frames, rows

    @property
    def shape(self):
        return (self.rows.shape[0], 3, 512)

    @property
    def device(self):
        return self.frames.device

    @property
    def is_cuda(self):
        return self.frames.is_cuda

    def gather(self):
        return self.frames[self.rows.long()]

    def record_stream(self, stream):
        if self.rows.is_cuda:
            self.rows.record_stream(stream)


def lidar_features(obs, w1, b1, w2, b2, out=None, head=None):
    """relu(conv2(relu(conv1(obs)))) of the actor and the critic tower in one launch.
    obs f32[N,3,512]; w1 f32[2,32,3,5], b1 f32[2,32], w2 f32[2,32,32,3], b2 f32[2,32] (tower-major: actor, critic)
    -> f32[2,N,4096], rows in the flatten order of [32,128].
    ``head`` u8[N] or a ``RingHead``: ``obs`` is a frame RING (VecStageWorld.policy_obs()) and head[n] the slot of
    robot n's newest frame; the kernel reads the frames in deque order while staging -- and, for a ring of RAW scans,
    forms the observation x / 6 - 0.5 on the way.  None: ``obs`` is in deque order."""
    lib = _lib.load()
    if isinstance(obs, FrameTable):
        if head is not None:
            raise ValueError("lidar_features: a FrameTable is in deque order already (no head)")
        N = obs.rows.shape[0]
        if not obs.is_cuda:
            raise ValueError("lidar_features: the FrameTable must live on the GPU")
        for t, shape in ((w1, (2, 32, 3, 5)), (b1, (2, 32)), (w2, (2, 32, 32, 3)), (b2, (2, 32))):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == shape):
                raise ValueError(f"lidar_features: expected a contiguous cuda float32 tensor of shape {shape}, got "
                                 f"{tuple(t.shape)} {t.dtype} {t.device}")
        if out is None:
            out = torch.empty(2, N, 4096, dtype=torch.float32, device=obs.device)
        with torch.cuda.device(obs.device):
            stream = C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)
            _lib.check(lib.mrca_lidar_features_rows(obs.frames.data_ptr(), obs.rows.data_ptr(), N, 3, 512, w1.data_ptr(),
                                                    b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), out.data_ptr(), stream),
                       "mrca_lidar_features_rows")
        return out
    head, raw = unwrap_head(head)
    N, F, B = obs.shape
    for t, shape in ((obs, (N, 3, 512)), (w1, (2, 32, 3, 5)), (b1, (2, 32)), (w2, (2, 32, 32, 3)), (b2, (2, 32))):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == shape):
            raise ValueError(f"lidar_features: expected a contiguous cuda float32 tensor of shape {shape}, got "
                             f"{tuple(t.shape)} {t.dtype} {t.device}")
    if head is not None and not (head.is_cuda and head.dtype == torch.uint8 and head.is_contiguous() and head.numel() == N):
        raise ValueError("lidar_features: head must be a contiguous cuda uint8 tensor with one entry per robot")
    if out is None:
        out = torch.empty(2, N, 4096, dtype=torch.float32, device=obs.device)
    with torch.cuda.device(obs.device):
        stream = C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)
        _lib.check(lib.mrca_lidar_features(obs.data_ptr(), None if head is None else head.data_ptr(), int(raw), N, F, B,
                                           w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), out.data_ptr(),
                                           stream), "mrca_lidar_features")
    return out


def policy_tail(h1, goal, speed, fc2_w, fc2_b, head_w, head_b, critic_w, critic_b, logstd, noise, lo, hi, fc1_b=None):
    """Everything of the rollout inference behind fc1 in one launch (include/mrca_env.h: mrca_policy_tail).
    h1 f32[2,N,256] = fc1 outputs before the ReLU -- with their bias, or without it and ``fc1_b`` f32[2,256] (any shape
    of 512 elements) given: the kernel adds it while it stages h1.  noise f32[N,2] or None (deterministic mean action).
    -> value [N,1], action [N,2], logprob [N,1], scaled [N,2], mean [N,2]"""
    lib = _lib.load()
    N = h1.shape[1]
    dev = h1.device
    tensors = [h1, goal, speed, fc2_w, fc2_b, head_w, head_b, critic_w, critic_b, logstd, lo, hi] + ([] if noise is None else [noise])
    shapes = [(2, N, 256), (N, 2), (N, 2), (2, 260, 128), None, (128, 2), (2,), None, None, (2,), (2,), (2,)] + ([] if noise is None else [(N, 2)])
    for t, shp in zip(tensors, shapes):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and (shp is None or tuple(t.shape) == shp)):
            raise ValueError(f"policy_tail: expected a contiguous cuda float32 tensor of shape {shp}, got {tuple(t.shape)} {t.dtype}")
    if fc2_b.numel() != 256 or critic_w.numel() != 128 or critic_b.numel() != 1:
        raise ValueError("policy_tail: fc2_b / critic_w / critic_b sizes")
    if fc1_b is not None and not (fc1_b.is_cuda and fc1_b.dtype == torch.float32 and fc1_b.is_contiguous() and fc1_b.numel() == 512
                                  and fc1_b.device == dev):
        raise ValueError("policy_tail: fc1_b must be a contiguous cuda float32 tensor of 2 x 256 elements on h1's device")
    value = torch.empty(N, 1, dtype=torch.float32, device=dev)
    action = torch.empty(N, 2, dtype=torch.float32, device=dev)
    logprob = torch.empty(N, 1, dtype=torch.float32, device=dev)
    scaled = torch.empty(N, 2, dtype=torch.float32, device=dev)
    mean = torch.empty(N, 2, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.mrca_policy_tail(h1.data_ptr(), None if fc1_b is None else fc1_b.data_ptr(), goal.data_ptr(), speed.data_ptr(), N, fc2_w.data_ptr(), fc2_b.data_ptr(),
                                        head_w.data_ptr(), head_b.data_ptr(), critic_w.data_ptr(), critic_b.data_ptr(),
                                        logstd.data_ptr(), None if noise is None else noise.data_ptr(), lo.data_ptr(),
                                        hi.data_ptr(), value.data_ptr(), action.data_ptr(), logprob.data_ptr(),
                                        scaled.data_ptr(), mean.data_ptr(), stream), "mrca_policy_tail")
    return value, action, logprob, scaled, mean


_scratch = {}


def _backward_scratch(device, stream=0):
    """Scratch of the backward kernel's per-wave partial sums, one buffer per (device, STREAM) like the loss kernel's and the
    heads': the partials live between the kernel and its finalize launch on one stream, and two trainers, rank threads or side
    streams of one process must not overwrite each other's."""
    key = (device.type, device.index, int(stream))
    if key not in _scratch:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("lidar_features_backward: first use on this stream inside a graph capture -- call it once before capturing")
        lib = _lib.load()
        n = C.c_size_t()
        with torch.cuda.device(device):
            _lib.check(lib.mrca_lidar_features_backward_scratch(C.byref(n)), "mrca_lidar_features_backward_scratch")
        _scratch[key] = torch.empty(n.value, dtype=torch.uint8, device=device)
    return _scratch[key]


def lidar_features_backward(obs, w1, b1, w2, feat, gfeat_act, gfeat_crt):
    """Gradients of lidar_features with respect to (w1, b1, w2, b2) given dLoss/dfeat of the two towers (two [N,4096]
    buffers: they come out of two independent fc1 backward GEMMs); see include/mrca_env.h.
    -> dw1 f32[2,32,3,5], db1 f32[2,32], dw2 f32[2,32,32,3], db2 f32[2,32]"""
    lib = _lib.load()
    N = obs.shape[0]
    table = obs if isinstance(obs, FrameTable) else None
    for t, shape in (((obs, (N, 3, 512)),) if table is None else ()) + (
                     (w1, (2, 32, 3, 5)), (b1, (2, 32)), (w2, (2, 32, 32, 3)), (feat, (2, N, 4096)),
                     (gfeat_act, (N, 4096)), (gfeat_crt, (N, 4096))):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == shape):
            raise ValueError(f"lidar_features_backward: expected a contiguous cuda float32 tensor of shape {shape}, got "
                             f"{tuple(t.shape)} {t.dtype} {t.device}")
    dev = obs.device
    dw1, db1 = torch.empty_like(w1), torch.empty_like(b1)
    dw2, db2 = torch.empty_like(w2), torch.empty(2, 32, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        scratch = _backward_scratch(dev, stream.value or 0)
        tail = (N, 3, 512, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), feat.data_ptr(), gfeat_act.data_ptr(), gfeat_crt.data_ptr(),
                dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(), scratch.data_ptr(), scratch.numel(), stream)
        if table is None:
            _lib.check(lib.mrca_lidar_features_backward(obs.data_ptr(), *tail), "mrca_lidar_features_backward")
        else:
            _lib.check(lib.mrca_lidar_features_backward_rows(table.frames.data_ptr(), table.rows.data_ptr(), *tail),
                       "mrca_lidar_features_backward_rows")
    return dw1, db1, dw2, db2


class _LidarFeatures(torch.autograd.Function):
    """-> (actor features [N,4096], critic features [N,4096]): two outputs, so that autograd hands the backward kernel
    the two fc1 gradients as they are (one stacked output cost a zero-fill + two copies + an add of [2,N,4096] per
    minibatch: profiles/r03a_update_bench.txt)."""

    @staticmethod
    def forward(ctx, obs, w1, b1, w2, b2):
        w1, b1, w2, b2 = (t.detach().contiguous() for t in (w1, b1, w2, b2))
        table = obs if isinstance(obs, FrameTable) else None
        if table is None:
            obs = obs.detach().contiguous()
        feat = lidar_features(obs, w1, b1, w2, b2)
        if table is None:
            ctx.save_for_backward(obs, w1, b1, w2, feat)
        else:       # (the frame store and the row table are data: kept by reference, like a saved tensor)
            ctx.save_for_backward(table.frames, table.rows, w1, b1, w2, feat)
        ctx.table = table is not None
        return feat[0], feat[1]

    @staticmethod
    def backward(ctx, g_act, g_crt):
        if ctx.table:
            frames, rows, w1, b1, w2, feat = ctx.saved_tensors
            obs = FrameTable(frames, rows)
        else:
            obs, w1, b1, w2, feat = ctx.saved_tensors
        g_act = torch.zeros_like(feat[0]) if g_act is None else g_act.contiguous()
        g_crt = torch.zeros_like(feat[1]) if g_crt is None else g_crt.contiguous()
        dw1, db1, dw2, db2 = lidar_features_backward(obs, w1, b1, w2, feat, g_act, g_crt)
        return None, dw1, db1, dw2, db2


def lidar_features_fn(obs, w1, b1, w2, b2):
    """Differentiable lidar_features: same arguments, returns (actor features, critic features), gradients flow to
    w1 / b1 / w2 / b2 (the scan is data)."""
    return _LidarFeatures.apply(obs, w1, b1, w2, b2)


# ------------------------------------------------------------------------------------------------ the PPO loss tail
_loss_scratch = {}


def _ppo_loss_scratch(device, stream):
    """The loss kernel's partial sums + ticket: one buffer per (device, STREAM) -- two trainers / rank threads / side streams in
    one process must not share the ticket -- allocated outside any graph capture (a buffer first created inside a capture
    would live in that graph's private pool)."""
    key = (device.type, device.index, int(stream))
    if key not in _loss_scratch:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("ppo_loss: first use on this stream inside a graph capture -- call it once before capturing")
        lib = _lib.load()
        n = C.c_size_t()
        _lib.check(lib.mrca_ppo_loss_scratch(C.byref(n)), "mrca_ppo_loss_scratch")
        _loss_scratch[key] = torch.zeros(n.value, dtype=torch.uint8, device=device)
    return _loss_scratch[key]


class _PPOLoss(torch.autograd.Function):
    """loss = L_pi + value_coef * L_v - coeff_entropy * H of model/ppo.py:172-185 with its gradients with respect to the
    network's outputs (mean, value, logstd) from ONE launch (csrc/mrca_ppo_loss.hip); backward scales them by the incoming
    gradient (1 for ``loss.backward()``)."""

    @staticmethod
    def forward(ctx, mean, value, logstd, action, old_logprob, adv, target, clip_value, value_coef, coeff_entropy):
        lib = _lib.load()
        n = mean.shape[0]
        args = [t.detach().contiguous().view(-1) for t in (mean, value, logstd, action, old_logprob, adv, target)]
        for t, numel in zip(args, (2 * n, n, 2, 2 * n, n, n, n)):
            if not (t.is_cuda and t.dtype == torch.float32 and t.numel() == numel):
                raise ValueError("ppo_loss: expected cuda float32 tensors mean [n,2], value [n,1], logstd [2], action [n,2], "
                                 "old_logprob / adv / target [n,1]")
        dev = mean.device
        out = torch.empty(8, dtype=torch.float32, device=dev)
        gmean = torch.empty(n, 2, dtype=torch.float32, device=dev)
        gvalue = torch.empty(n, 1, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            scratch = _ppo_loss_scratch(dev, stream.value or 0)
            _lib.check(lib.mrca_ppo_loss(*[t.data_ptr() for t in args], n, float(clip_value), float(value_coef),
                                         float(coeff_entropy), out.data_ptr(), gmean.data_ptr(), gvalue.data_ptr(),
                                         scratch.data_ptr(), scratch.numel(), stream), "mrca_ppo_loss")
        ctx.save_for_backward(gmean, gvalue, out)
        ctx.value_shape, ctx.logstd_shape = value.shape, logstd.shape
        ctx.mark_non_differentiable(out)
        return out[0], out

    @staticmethod
    def backward(ctx, g_loss, _g_out):
        gmean, gvalue, out = ctx.saved_tensors
        return (gmean * g_loss, (gvalue * g_loss).view(ctx.value_shape), (out[5:7] * g_loss).view(ctx.logstd_shape),
                None, None, None, None, None, None, None)


def ppo_loss(mean, value, logstd, action, old_logprob, adv, target, clip_value, value_coef, coeff_entropy):
    """-> (loss, stats): ``loss`` differentiable with respect to mean [n,2], value [n,1] and logstd [2]; ``stats`` f32[8] =
    loss, policy loss, value loss, entropy, k3 KL(old || new), dloss/dlogstd[0], dloss/dlogstd[1], 0."""
    return _PPOLoss.apply(mean, value, logstd, action, old_logprob, adv, target, clip_value, value_coef, coeff_entropy)


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step):
    """One Adam step on flat CUDA buffers in ONE launch (include/mrca_env.h: mrca_adam_step; torch.optim.Adam's rule, no
    weight decay, no amsgrad).  ``param``, ``exp_avg``, ``exp_avg_sq`` are updated in place; ``step`` counts from 1."""
    lib = _lib.load()
    n = param.numel()
    for t in (param, grad, exp_avg, exp_avg_sq):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 1 and t.numel() == n
                and t.device == param.device):
            raise ValueError(f"adam_step: expected four flat contiguous cuda float32 tensors of {n} elements on one device, "
                             f"got {tuple(t.shape)} {t.dtype} {t.device}")
    with torch.cuda.device(param.device):
        stream = C.c_void_p(torch.cuda.current_stream(param.device).cuda_stream)
        _lib.check(lib.mrca_adam_step(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), n,
                                      float(lr), float(beta1), float(beta2), float(eps), int(step), stream), "mrca_adam_step")


_heads_scratch = {}


def _heads_backward_scratch(device, stream):
    """The partial records between the heads' backward and finalize kernels: one buffer per (device, STREAM), like the loss
    kernel's -- two trainers, rank threads or side streams of one process must not overwrite each other's partials."""
    key = (device.type, device.index, int(stream))
    if key not in _heads_scratch:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("policy_heads backward: first use on this stream inside a graph capture -- call it once before capturing")
        lib = _lib.load()
        n = C.c_size_t()
        _lib.check(lib.mrca_policy_heads_backward_scratch(C.byref(n)), "mrca_policy_heads_backward_scratch")
        _heads_scratch[key] = torch.empty(n.value, dtype=torch.uint8, device=device)
    return _heads_scratch[key]


class _PolicyHeads(torch.autograd.Function):
    """mean = [sigmoid(actor1(a)), tanh(actor2(a))], value = critic(c) (model/net.py:47-55,61-63) and their gradients as
    row kernels (csrc/mrca_policy_heads.hip) instead of three skinny GEMMs forward and six backward."""

    @staticmethod
    def forward(ctx, a, c, w1, b1, w2, b2, wc, bc, relu_inputs=False, zb_a=None, zb_c=None):
        # zb_a / zb_c: the biases of the layers that produced a / c (act_fc2.bias / crt_fc2.bias), or None.  Not used forward --
        # the caller added them with the bias DETACHED (F.linear(x, W, b.detach())) -- they are here to receive their gradient:
        # the column sums of da / dc, which the backward kernel forms on the way (mrca_policy_heads_backward_bias)
        lib = _lib.load()
        n = a.shape[0]
        a, c = a.detach().contiguous(), c.detach().contiguous()
        ws = [t.detach().contiguous().view(-1) for t in (w1, b1, w2, b2, wc, bc)]
        for t, numel in zip([a, c] + ws, (n * 128, n * 128, 128, 1, 128, 1, 128, 1)):
            if not (t.is_cuda and t.dtype == torch.float32 and t.numel() == numel and t.device == a.device):
                raise ValueError("policy_heads: expected cuda float32 tensors a, c [n,128], three weight rows [1,128] and biases [1] "
                                 "on one device")
        mean = torch.empty(n, 2, dtype=torch.float32, device=a.device)
        value = torch.empty(n, 1, dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            stream = C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
            _lib.check(lib.mrca_policy_heads(a.data_ptr(), c.data_ptr(), n, *[t.data_ptr() for t in ws], int(bool(relu_inputs)),
                                             mean.data_ptr(), value.data_ptr(), stream), "mrca_policy_heads")
        ctx.save_for_backward(a, c, mean, ws[0], ws[2], ws[4])
        ctx.shapes = (w1.shape, b1.shape, w2.shape, b2.shape, wc.shape, bc.shape)
        ctx.relu_inputs = bool(relu_inputs)
        ctx.z_bias = (zb_a is not None, zb_c is not None)
        return mean, value

    @staticmethod
    def backward(ctx, gmean, gvalue):
        lib = _lib.load()
        a, c, mean, w1, w2, wc = ctx.saved_tensors
        n, dev = a.shape[0], a.device
        gmean = None if gmean is None else gmean.contiguous()
        gvalue = None if gvalue is None else gvalue.contiguous()
        da, dc = torch.empty_like(a), torch.empty_like(c)
        dw = torch.empty(3 * 128 + 3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            scratch = _heads_backward_scratch(dev, stream.value or 0)
            head = (a.data_ptr(), c.data_ptr(), mean.data_ptr(), None if gmean is None else gmean.data_ptr(),
                    None if gvalue is None else gvalue.data_ptr(), n, w1.data_ptr(), w2.data_ptr(), wc.data_ptr(),
                    int(ctx.relu_inputs), da.data_ptr(), dc.data_ptr(), dw.data_ptr())
            dzb = None
            if any(ctx.z_bias):
                dzb = torch.empty(256, dtype=torch.float32, device=dev)
                _lib.check(lib.mrca_policy_heads_backward_bias(*head, dzb.data_ptr(), scratch.data_ptr(), scratch.numel(), stream),
                           "mrca_policy_heads_backward_bias")
            else:
                _lib.check(lib.mrca_policy_heads_backward(*head, scratch.data_ptr(), scratch.numel(), stream),
                           "mrca_policy_heads_backward")
        s = ctx.shapes
        return (da, dc, dw[0:128].view(s[0]), dw[384:385].view(s[1]), dw[128:256].view(s[2]), dw[385:386].view(s[3]),
                dw[256:384].view(s[4]), dw[386:387].view(s[5]), None,
                dzb[0:128] if ctx.z_bias[0] else None, dzb[128:256] if ctx.z_bias[1] else None)


def policy_heads(a, c, w_actor1, b_actor1, w_actor2, b_actor2, w_critic, b_critic, relu_inputs=False, z_bias=None):
    """-> (mean [n,2], value [n,1]) of the three output heads, differentiable with respect to the first eight arguments
    (include/mrca_env.h: mrca_policy_heads / _backward).  ``relu_inputs``: ``a`` / ``c`` are fc2's outputs BEFORE their ReLU --
    the kernels apply it (and its mask on the way back).  ``z_bias`` = (bias of the layer that produced a, ... that produced c):
    they receive the column sums of da / dc as their gradient -- for a caller that formed a / c with those biases detached
    (``F.linear(x, W, b.detach())``), so that autograd does not sum the same columns again."""
    zb_a, zb_c = z_bias if z_bias is not None else (None, None)
    return _PolicyHeads.apply(a, c, w_actor1, b_actor1, w_actor2, b_actor2, w_critic, b_critic, relu_inputs, zb_a, zb_c)


_relu_cat_scratches = {}


def _relu_cat_scratch(device, stream):
    """per (device, stream), like the heads' scratch: the per-workgroup records of relu_cat's backward with bias sums"""
    key = (device.type, device.index, int(stream))
    if key not in _relu_cat_scratches:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("relu_cat backward: first use on this stream inside a graph capture -- call it once before capturing")
        lib = _lib.load()
        n = C.c_size_t()
        _lib.check(lib.mrca_relu_cat_backward_bias_scratch(C.byref(n)), "mrca_relu_cat_backward_bias_scratch")
        _relu_cat_scratches[key] = torch.empty(n.value, dtype=torch.uint8, device=device)
    return _relu_cat_scratches[key]


class _ReluCat(torch.autograd.Function):
    """[relu(h1), goal, speed] (model/net.py:43-45) and dh1 = gout[:, :256] where h1 > 0, one launch each."""

    @staticmethod
    def forward(ctx, h1, goal, speed, h1_bias=None):
        # h1_bias: the bias of the layer that produced h1 (act_fc1.bias / crt_fc1.bias) or None -- not used forward (the caller
        # added it DETACHED), here to receive its gradient: the column sums of dh1 (mrca_relu_cat_backward_bias)
        lib = _lib.load()
        n = h1.shape[0]
        h1, goal, speed = h1.detach().contiguous(), goal.detach().contiguous(), speed.detach().contiguous()
        for t, shape in ((h1, (n, 256)), (goal, (n, 2)), (speed, (n, 2))):
            if not (t.is_cuda and t.dtype == torch.float32 and tuple(t.shape) == shape and t.device == h1.device):
                raise ValueError("relu_cat: expected cuda float32 tensors h1 [n,256], goal [n,2], speed [n,2] on one device")
        out = torch.empty(n, 260, dtype=torch.float32, device=h1.device)
        with torch.cuda.device(h1.device):
            stream = C.c_void_p(torch.cuda.current_stream(h1.device).cuda_stream)
            _lib.check(lib.mrca_relu_cat(h1.data_ptr(), goal.data_ptr(), speed.data_ptr(), n, out.data_ptr(), stream), "mrca_relu_cat")
        ctx.save_for_backward(h1)
        ctx.h1_bias = h1_bias is not None
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        (h1,) = ctx.saved_tensors
        gout = gout.contiguous()
        dh1 = torch.empty_like(h1)
        db = None
        with torch.cuda.device(h1.device):
            stream = C.c_void_p(torch.cuda.current_stream(h1.device).cuda_stream)
            if ctx.h1_bias:
                db = torch.empty(256, dtype=torch.float32, device=h1.device)
                scratch = _relu_cat_scratch(h1.device, stream.value or 0)
                _lib.check(lib.mrca_relu_cat_backward_bias(h1.data_ptr(), gout.data_ptr(), h1.shape[0], dh1.data_ptr(), db.data_ptr(),
                                                           scratch.data_ptr(), scratch.numel(), stream), "mrca_relu_cat_backward_bias")
            else:
                _lib.check(lib.mrca_relu_cat_backward(h1.data_ptr(), gout.data_ptr(), h1.shape[0], dh1.data_ptr(), stream),
                           "mrca_relu_cat_backward")
        return dh1, None, None, db


def relu_cat(h1, goal, speed, h1_bias=None):
    """-> f32[n,260] = cat(relu(h1), goal, speed); differentiable with respect to ``h1`` (goal and speed are data).
    ``h1_bias``: the bias of the layer that produced ``h1``, for a caller that added it detached: it receives the column sums of
    dh1 as its gradient (see ``policy_heads``' ``z_bias``)."""
    return _ReluCat.apply(h1, goal, speed, h1_bias)
