"""The lidar actor-critic, kept in stock PyTorch (BASELINE.json north_star: "the 1-D-conv policy in
model/net.py kept in PyTorch").

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

    def _tower(self, tw, x, goal, speed):
        # the two Conv1d layers evaluated as H=1 conv2d on a channels-last tensor: same parameters, bitwise
        # the same result on gfx950, ~15 % faster because MIOpen skips its NCHW<->NHWC transposes
        # (profiles/r01_l_policy_formulations.txt)
        c1, c2 = getattr(self, f"{tw}_fea_cv1"), getattr(self, f"{tw}_fea_cv2")
        h = x.unsqueeze(2).contiguous(memory_format=torch.channels_last)
        h = torch.relu(F.conv2d(h, c1.weight.unsqueeze(2), c1.bias, stride=(1, 2), padding=(0, 1)))
        h = torch.relu(F.conv2d(h, c2.weight.unsqueeze(2), c2.bias, stride=(1, 2), padding=(0, 1)))
        h = torch.relu(getattr(self, f"{tw}_fc1")(h.contiguous().flatten(1)))
        h = torch.cat((h, goal, speed), dim=-1)
        return torch.relu(getattr(self, f"{tw}_fc2")(h))

    def mean_value(self, x, goal, speed):
        a = self._tower("act", x, goal, speed)
        mean = torch.cat((torch.sigmoid(self.actor1(a)), torch.tanh(self.actor2(a))), dim=-1)
        v = self.critic(self._tower("crt", x, goal, speed))
        return mean, v

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
