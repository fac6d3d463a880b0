"""Which aten ops (and which source lines) the small launches of a PPO minibatch come from: torch.profiler over ONE update of
bench.py --mode train's trainer (4096 robots, 16 384-row minibatches, a short horizon).  GPU box only.
    python tools/update_profile.py [--horizon 16] > gpurun_out/update_profile.txt"""
import argparse
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "rl-collision-avoidance_amd"))
sys.path.insert(0, R)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--horizon", type=int, default=16)
    ap.add_argument("--stacks", type=int, default=5)
    a = ap.parse_args()
    import __graft_entry__ as g
    g.build()
    from mrca import gemm_tuning, scenario, vec_env
    from mrca.trainer import HParams, Stage1Trainer
    gemm_tuning.use_recorded_choices()
    env = vec_env.VecStageWorld(scenario.stage1(num_worlds=128, robots_per_world=32, seed=0))
    hp = HParams(horizon=a.horizon, batch_size=16384, rollout_fused=True, update_fused=True, graph_tick=False)
    tr = Stage1Trainer(env, hp=hp, seed=0)
    tr.start()
    tr.run(a.horizon)                   # one update outside the profile (allocator, tuning)
    tr.run(a.horizon - 1)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        tr.run(1)                       # the tick that triggers the update
        torch.cuda.synchronize()
    n_mb = hp.epoch * (a.horizon * env.N // hp.batch_size)
    print(f"# one update = {n_mb} minibatches; counts below are per update")
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=70, max_name_column_width=70))
    print(prof.key_averages(group_by_stack_n=a.stacks).table(sort_by="self_cuda_time_total", row_limit=90, max_name_column_width=60,
                                                               max_src_column_width=110))
    env.close()


if __name__ == "__main__":
    main()
