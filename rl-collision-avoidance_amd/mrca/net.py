"""The lidar actor-critic, kept in PyTorch (BASELINE.json north_star: "the 1-D-conv policy in
model/net.py kept in PyTorch"): an ``nn.Module`` with the reference's parameters whose two Conv1d + ReLU pairs may run
-- forward and backward -- through the hand-written HIP front end (``fused_train``, mrca/policy_ops.py) instead of MIOpen.

Architecture and parameter names follow the reference's ``CNNPolicy`` (model/net.py:16-80) so that
``policy/stage2.pth`` style checkpoints load unchanged (state_dict keys: ``logstd``,
``act_fea_cv1``, ``act_fea_cv2``, ``act_fc1``, ``act_fc2``, ``actor1``, ``actor2``,
``crt_fea_cv1``, ``crt_fea_cv2``, ``crt_fc1``, ``crt_fc2``, ``critic``):

    tower(x[N,F,512]) = relu(Conv1d(F,32,k5,s2,p1)) -> relu(Conv1d(32,32,k3,s2,p1)) -> flatten 32*128
                        -> relu(Linear(4096,256)) -> cat(goal[N,2], speed[N,2]) -> relu(Linear(260,128))
    mean  = [sigmoid(Linear(128,1)), tanh(Linear(128,1))]      (actor tower)
    value = Linear(128,1)                                       (critic tower)
    action ~ N(mean, exp(logstd)), logstd a free parameter of size 2.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

_HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


def gaussian_logprob(x, mean, logstd):
    """log N(x; mean, exp(logstd)) summed over the action dimension, keepdim
    (model/utils.py:90-97 log_normal_density)."""
    var = torch.exp(logstd) ** 2
    lp = -((x - mean) ** 2) / (2.0 * var) - _HALF_LOG_2PI - logstd
    return lp.sum(dim=-1, keepdim=True)


class CNNPolicy(nn.Module):
    TOWERS = ("act", "crt")

    def __init__(self, frames=3, action_space=2, beams=512):
        super().__init__()
        self.logstd = nn.Parameter(torch.zeros(action_space))
        len1 = (beams + 2 - 5) // 2 + 1
        len2 = (len1 + 2 - 3) // 2 + 1
        self.flat = 32 * len2
        for tw in self.TOWERS:
            setattr(self, f"{tw}_fea_cv1", nn.Conv1d(frames, 32, kernel_size=5, stride=2, padding=1))
            setattr(self, f"{tw}_fea_cv2", nn.Conv1d(32, 32, kernel_size=3, stride=2, padding=1))
            setattr(self, f"{tw}_fc1", nn.Linear(self.flat, 256))
            setattr(self, f"{tw}_fc2", nn.Linear(256 + 2 + 2, 128))
            if tw == "act":
                self.actor1 = nn.Linear(128, 1)
                self.actor2 = nn.Linear(128, 1)
        self.critic = nn.Linear(128, 1)

    # True: mean_value (and so forward / evaluate_actions, i.e. the PPO update) evaluates the conv front end of BOTH
    # towers through policy_ops.lidar_features_fn -- fp32 MFMA forward and backward kernels, gradients to the same
    # nn.Conv1d parameters.  Needs a GPU and frames = 3, beams = 512; results differ from the stock layers by
    # summation order only (tests/test_gpu_policy_bwd.py).
    fused_train = False

    def _tail(self, tw, h, goal, speed):
        h = torch.relu(getattr(self, f"{tw}_fc1")(h))
        h = torch.cat((h, goal, speed), dim=-1)
        return torch.relu(getattr(self, f"{tw}_fc2")(h))

    def _tower(self, tw, x, goal, speed):
        # the two Conv1d layers evaluated as H=1 conv2d on a channels-last tensor: same parameters, bitwise
        # the same result on gfx950, ~15 % faster because MIOpen skips its NCHW<->NHWC transposes
        # (profiles/r01/r01_l_policy_formulations.txt)
        c1, c2 = getattr(self, f"{tw}_fea_cv1"), getattr(self, f"{tw}_fea_cv2")
        h = x.unsqueeze(2).contiguous(memory_format=torch.channels_last)
        h = torch.relu(F.conv2d(h, c1.weight.unsqueeze(2), c1.bias, stride=(1, 2), padding=(0, 1)))
        h = torch.relu(F.conv2d(h, c2.weight.unsqueeze(2), c2.bias, stride=(1, 2), padding=(0, 1)))
        return self._tail(tw, h.contiguous().flatten(1), goal, speed)

    def mean_value(self, x, goal, speed):
        from . import policy_ops
        table = isinstance(x, policy_ops.FrameTable)
        if table and not self.fused_train:
            x = x.gather()
        if self.fused_train and x.is_cuda:
            st = lambda a, c: torch.stack((a, c))      # noqa: E731  (its backward hands each tower its slice)
            fa, fc = policy_ops.lidar_features_fn(
                x if table else x.float(), st(self.act_fea_cv1.weight, self.crt_fea_cv1.weight), st(self.act_fea_cv1.bias, self.crt_fea_cv1.bias),
                st(self.act_fea_cv2.weight, self.crt_fea_cv2.weight), st(self.act_fea_cv2.bias, self.crt_fea_cv2.bias))
            # behind the front end the layers are the module's own Linear layers (library GEMMs); what sits BETWEEN them runs
            # as row kernels (csrc/mrca_policy_heads.hip): relu + cat with goal and speed in one launch, fc2's ReLU applied
            # by the head kernels as they load, the three Linear(128, 1) heads with sigmoid / tanh forward and backward (as
            # library GEMMs the heads alone are ~25 launches per minibatch)
            # fc1 / fc2 add their biases DETACHED: the bias gradients are column sums of dh1 / dz, which the row kernels behind
            # them form on the way (relu_cat's and the heads' backward) -- as autograd nodes they were four `sum` launches over
            # the matrices those kernels had just written
            z, zb = [], []
            for tw, f in (("act", fa), ("crt", fc)):
                fc1, fc2 = getattr(self, f"{tw}_fc1"), getattr(self, f"{tw}_fc2")
                x2 = policy_ops.relu_cat(F.linear(f, fc1.weight, fc1.bias.detach()), goal, speed, h1_bias=fc1.bias)
                z.append(F.linear(x2, fc2.weight, fc2.bias.detach()))
                zb.append(fc2.bias)
            return policy_ops.policy_heads(z[0], z[1], self.actor1.weight, self.actor1.bias, self.actor2.weight,
                                           self.actor2.bias, self.critic.weight, self.critic.bias, relu_inputs=True,
                                           z_bias=tuple(zb))
        else:
            a = self._tower("act", x, goal, speed)
            c = self._tower("crt", x, goal, speed)
        mean = torch.cat((torch.sigmoid(self.actor1(a)), torch.tanh(self.actor2(a))), dim=-1)
        return mean, self.critic(c)

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        if getattr(self, "_rc", None) is not None:      # the fused rollout path reads derived copies: keep them in step
            self.refresh_rollout_cache()
        return out

    # ------------------------------------------------------------------ rollout fast path (inference only)
    def refresh_rollout_cache(self):
        """Tower-major copies of the parameters the fused rollout path reads (call after every optimiser step
        sequence, i.e. once per PPO update): the two conv layers for the HIP front end, fc1 / fc2 / heads stacked for
        batched GEMMs.  Same parameters, same state_dict keys -- these are derived buffers, never saved."""
        with torch.no_grad():
            st = lambda a, c: torch.stack([a.detach(), c.detach()]).contiguous()   # noqa: E731
            new = {
                "w1": st(self.act_fea_cv1.weight, self.crt_fea_cv1.weight), "b1": st(self.act_fea_cv1.bias, self.crt_fea_cv1.bias),
                "w2": st(self.act_fea_cv2.weight, self.crt_fea_cv2.weight), "b2": st(self.act_fea_cv2.bias, self.crt_fea_cv2.bias),
                "fc1_w": st(self.act_fc1.weight.t(), self.crt_fc1.weight.t()),        # [2, 4096, 256]
                "fc1_b": st(self.act_fc1.bias, self.crt_fc1.bias).unsqueeze(1),      # [2, 1, 256]
                "fc2_w": st(self.act_fc2.weight.t(), self.crt_fc2.weight.t()),        # [2, 260, 128]
                "fc2_b": st(self.act_fc2.bias, self.crt_fc2.bias).unsqueeze(1),      # [2, 1, 128]
                "head_w": torch.cat([self.actor1.weight, self.actor2.weight]).detach().t().contiguous(),   # [128, 2]
                "head_b": torch.cat([self.actor1.bias, self.actor2.bias]).detach(),
                # .clone(): reshape / contiguous of an already contiguous parameter is a VIEW of it -- the in-place
                # refresh below would copy the parameter onto itself and bump its version counter
                "critic_w": self.critic.weight.detach().reshape(-1).clone(),          # [128]
                "critic_b": self.critic.bias.detach().reshape(1).clone(),
                "logstd": self.logstd.detach().reshape(-1).clone(),
            }
            old = getattr(self, "_rc", None)
            if old is not None and all(old[k].shape == v.shape and old[k].device == v.device for k, v in new.items()):
                for k, v in new.items():          # IN PLACE: a captured hipGraph keeps reading the same addresses
                    old[k].copy_(v)
            else:
                self._rc = new
            self._rc_stamp = self._param_stamp()
        return self._rc

    def _param_stamp(self):
        """What the derived copies were made from: every parameter's in-place version counter (optimiser steps and
        ``copy_`` / broadcasts INTO the parameter under no_grad bump it; writes through ``p.data`` do NOT -- ``p.data`` is
        a detached alias with a counter of its own -- so whoever writes through ``.data`` calls ``refresh_rollout_cache``
        itself, as trainer.broadcast_parameters does), storage address and device (``.to()`` / ``.cuda()`` swap the
        storage)."""
        return tuple((p._version, p.data_ptr(), p.device) for p in self.parameters())

    def _rollout_cache(self):
        """The cache, rebuilt first if any parameter changed since it was made -- whoever changed it (the trainer
        refreshes it itself after an update, so this is one tuple comparison per eager call; a tick replayed as a
        hipGraph does not come through here, the refresh is in place for its sake)."""
        if getattr(self, "_rc", None) is None or getattr(self, "_rc_stamp", None) != self._param_stamp():
            self.refresh_rollout_cache()
        return self._rc

    def mean_value_fused(self, x, goal, speed, head=None):
        """mean_value for the rollout: the conv front end of BOTH towers in one HIP kernel (csrc/mrca_policy.hip,
        fp32 MFMA), fc1 / fc2 of both towers as batched fp32 GEMMs.  fp32 throughout; differs from mean_value by
        summation order only (tests/test_gpu_policy_ops.py: 1e-5).  No autograd.  ``head``: ``x`` is the env's frame
        ring and head[n] the slot of robot n's newest frame (VecStageWorld.policy_obs())."""
        from . import policy_ops
        rc = self._rollout_cache()
        with torch.no_grad():
            feat = policy_ops.lidar_features(x, rc["w1"], rc["b1"], rc["w2"], rc["b2"], head=head)   # [2, N, 4096]
            h = torch.relu(torch.baddbmm(rc["fc1_b"], feat, rc["fc1_w"]))                      # [2, N, 256]
            gs = torch.cat((goal, speed), dim=-1).unsqueeze(0).expand(2, -1, -1)
            h = torch.relu(torch.baddbmm(rc["fc2_b"], torch.cat((h, gs), dim=-1), rc["fc2_w"]))   # [2, N, 128]
            m = torch.addmm(rc["head_b"], h[0], rc["head_w"])
            mean = torch.stack((torch.sigmoid(m[:, 0]), torch.tanh(m[:, 1])), dim=-1)
            v = self.critic(h[1])
        return mean, v

    def act_fused(self, x, goal, speed, noise, lo, hi, head=None):
        """generate_action for the rollout in three launches: the conv front end of both towers (csrc/mrca_policy.hip),
        fc1 of both towers as one batched fp32 GEMM (a plain bmm: a baddbmm first copies the broadcast bias into its
        output, 8 MB and 7 us per tick at 4096 robots -- the tail adds the bias while it stages h1), and everything
        behind it -- bias, ReLU, cat, fc2, heads, sample, logprob, clip -- in csrc/mrca_policy_tail.hip.  ``noise`` f32[N,2] standard normal draws or None (mean action).
        -> value [N,1], action [N,2], logprob [N,1], scaled [N,2], mean [N,2].  fp32; differs from the stock path by
        summation order only (tests/test_gpu_policy_ops.py)."""
        from . import policy_ops
        rc = self._rollout_cache()
        with torch.no_grad():
            feat = policy_ops.lidar_features(x, rc["w1"], rc["b1"], rc["w2"], rc["b2"], head=head)   # [2, N, 4096]
            h1 = torch.bmm(feat, rc["fc1_w"])                                        # [2, N, 256], before bias and ReLU
            return policy_ops.policy_tail(h1, goal.contiguous(), speed.contiguous(), rc["fc2_w"], rc["fc2_b"], rc["head_w"],
                                          rc["head_b"], rc["critic_w"], rc["critic_b"], rc["logstd"], noise, lo, hi,
                                          fc1_b=rc["fc1_b"])

    def forward(self, x, goal, speed, generator=None):
        """-> (value, sampled action, logprob, mean)   (model/net.py:37-70)"""
        mean, v = self.mean_value(x, goal, speed)
        logstd = self.logstd.expand_as(mean)
        noise = torch.randn(mean.shape, device=mean.device, dtype=mean.dtype, generator=generator)
        action = mean + torch.exp(logstd) * noise
        return v, action, gaussian_logprob(action, mean, logstd), mean

    def evaluate_actions(self, x, goal, speed, action):
        """-> (value, logprob of `action`, mean entropy)   (model/net.py:72-80)"""
        mean, v = self.mean_value(x, goal, speed)
        logstd = self.logstd.expand_as(mean)
        entropy = (0.5 + _HALF_LOG_2PI + logstd).sum(-1).mean()
        return v, gaussian_logprob(action, mean, logstd), entropy
