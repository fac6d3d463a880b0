"""The self-trained checkpoint committed under mrca/data/ on the circle test (circle_test.py:36-83), CPU side: the C
oracle env, the policy in stock PyTorch on the CPU.  (``policy/stage2.pth`` is absent from the reference checkout, so
this checkpoint -- Stage-1, then Stage-2 worlds mixed with circles of 10-50 robots, profiles/r02/r02_d_* -- is the only
end-to-end evidence that the loop learns the task.  The GPU legs are in tests/test_gpu_circle.py.)"""
import hashlib
import os

import numpy as np
import torch

import util as U
from util import S

CHECKPOINT = os.path.join(U.ROOT, "rl-collision-avoidance_amd", "mrca", "data", "policy_r02_stage2_circles.pth")


class _OracleVec:
    def __init__(self, sc):
        self.o = U.COracleEnv(sc)
        self.N = sc.num_robots
        for k in ("obs", "local_goal", "speed", "speed_gt", "done", "first_result", "reward", "pose", "goal", "init_pose"):
            setattr(self, k, torch.from_numpy(getattr(self.o, k)))

    def reset(self):
        self.o.reset()

    def step(self, a):
        self.o.step(a.numpy())


def test_checkpoint_is_the_one_the_profiles_describe():
    h = hashlib.sha256(open(CHECKPOINT, "rb").read()).hexdigest()
    assert h.startswith("23b64ea95aeb5af9"), h
    sd = torch.load(CHECKPOINT, map_location="cpu")
    assert set(sd) >= {"logstd", "act_fea_cv1.weight", "crt_fc2.bias", "actor1.weight", "critic.bias"}   # reference keys


def test_round3_checkpoint_is_the_one_the_profiles_describe():
    p = os.path.join(os.path.dirname(CHECKPOINT), "policy_r03_fused_update_11min.pth")
    assert hashlib.sha256(open(p, "rb").read()).hexdigest().startswith("5729e04243095894")
    sd = torch.load(p, map_location="cpu")
    ref = torch.load(CHECKPOINT, map_location="cpu")
    assert set(sd) == set(ref) and all(sd[k].shape == ref[k].shape for k in ref)


def test_trained_checkpoint_solves_the_circle_test_on_the_oracle_env():
    from mrca import evaluate
    from mrca.net import CNNPolicy
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    pol = CNNPolicy(3, 2)
    pol.load_state_dict(torch.load(CHECKPOINT, map_location="cpu"))
    env = _OracleVec(S.circle(num_worlds=1, seed=0))
    m = evaluate.circle_test(env, evaluate.cnn_policy_fn(pol), max_ticks=1200)
    print(m)
    assert m["success_rate"] >= 0.9 and m["crash_rate"] <= 0.1      # measured: 50 of 50 robots, 600 ticks
    assert m["mean_path_ratio"] < 1.2 and m["average_speed_mps"] > 0.5


CHECKPOINT_ALL_SIZES = os.path.join(U.ROOT, "rl-collision-avoidance_amd", "mrca", "data", "policy_r02_all_circle_sizes.pth")


def test_second_checkpoint_solves_circles_of_every_size_on_the_oracle_env():
    """The continuation of the same run with circles of 10-50 robots in the mix and validation on 20 / 30 / 40 / 50
    robots at once (profiles/r02/r02_h_*; sha256 f9e51e72...): circles the paper evaluates (Long et al. 2018, Sec. V: 4-20
    robots) and beyond.  Measured on this env: 1.00 / 1.00 / 1.00 / 0.98 for 20 / 30 / 40 / 50 robots."""
    from mrca import evaluate
    from mrca.net import CNNPolicy
    assert hashlib.sha256(open(CHECKPOINT_ALL_SIZES, "rb").read()).hexdigest().startswith("f9e51e72baf0896e")
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    pol = CNNPolicy(3, 2)
    pol.load_state_dict(torch.load(CHECKPOINT_ALL_SIZES, map_location="cpu"))
    for robots, radius in ((20, 12.0), (30, 16.0), (40, 20.0), (50, 25.0)):
        sc = S.circle(num_worlds=1) if robots == 50 else S.circle_n(robots, radius)
        m = evaluate.circle_test(_OracleVec(sc), evaluate.cnn_policy_fn(pol), max_ticks=1200)
        print(robots, m)
        assert m["success_rate"] >= 0.9, (robots, m)
