"""A tiny SPMD script written against the reference's StageWorld / rospy / mpi4py surface
(stand-in for ppo_stage1.py on the GPU box, where /root/reference does not exist)."""
import json
import os

import numpy as np
import rospy
from mpi4py import MPI
from stage_world1 import StageWorld

NUM_ENV = int(os.environ.get("MINI_NUM_ENV", "6"))
comm = MPI.COMM_WORLD
rank = comm.Get_rank()
env = StageWorld(512, index=rank, num_env=NUM_ENV)
rng = np.random.RandomState(100 + rank)
log = []
if rank == 0:
    env.reset_world()
try:
    episode, need_reset, obs, step = 0, True, None, 1
    for it in range(int(os.environ.get("MINI_ITERS", "60"))):     # every rank takes part in every collective
        if need_reset:
            env.reset_pose()
            env.generate_goal_point()
            obs = env.get_laser_observation()
            assert obs.shape == (512,) and obs.dtype == np.float64 and -0.5 <= obs.min() and obs.max() <= 0.5
            step, need_reset = 1, False
        if rospy.is_shutdown():
            break
        states = comm.gather([obs, env.get_local_goal(), env.get_self_speed()], root=0)
        acts = None
        if rank == 0:
            assert len(states) == NUM_ENV
            acts = [[0.8, 0.3 * np.sin(0.1 * it + r)] for r in range(NUM_ENV)]
        a = comm.scatter(acts, root=0)
        env.control_vel(a)
        rospy.sleep(0.001)
        r, terminal, result = env.get_reward_and_terminate(step)
        obs = env.get_laser_observation()
        log.append([episode, step, float(r), bool(terminal), str(result), env.get_self_stateGT(), env.get_self_speed()])
        step += 1
        if terminal or step > 25:
            episode += 1
            need_reset = True
except KeyboardInterrupt:
    pass
out = os.environ.get("MINI_OUT")
if out:
    with open(f"{out}.{rank}.json", "w") as f:
        json.dump(log, f)
