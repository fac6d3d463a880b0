"""Drop-in for the slice of ``rospy`` the reference scripts touch (ppo_stage1.py:6,65,78;
ppo_stage2.py:40; circle_test.py:53): is_shutdown / sleep / Rate.  Time is simulator time
(/use_sim_time is true, stageros.cpp:367): a sleep after ``control_vel`` returns when the tick
that consumed the command has run."""
from mrca import spmd


def is_shutdown():
    return spmd.runtime().shutdown


def sleep(_duration):
    rt = spmd.runtime()
    w = rt.world
    rt.sleep(bool(w is not None and w.has_cmd[spmd.rank()]))


class Rate:
    def __init__(self, hz):
        self.hz = hz

    def sleep(self):
        sleep(1.0 / self.hz)


def init_node(*_a, **_k):
    return None


def on_shutdown(_fn):
    return None
