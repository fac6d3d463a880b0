"""SPMD-in-one-process runtime: runs an UNCHANGED reference script (``ppo_stage1.py``,
``ppo_stage2.py``, ``circle_test.py``) as ``-np`` rank threads against one batched device world.

    python -m mrca.spmd -np 24 /path/to/ppo_stage1.py

What it replaces: ``mpiexec -np 24 python ppo_stage1.py`` + ``rosrun stage_ros_add_pose_and_crash
stageros`` (README.md:30-34).  The drop-in modules under ``mrca/dropin`` shadow ``rospy``,
``mpi4py``, ``tf`` and ``stage_world1/2``, ``circle_world``; pickled MPI messages become in-memory
hand-offs between threads, ROS topics become reads of the device arena.

Tick rule: the simulator advances when EVERY rank is blocked -- in ``rospy.sleep`` (its action is
latched) or inside an MPI collective -- i.e. once per loop iteration of the scripts.  The ranks run under a deterministic
cooperative schedule (one at a time, lowest ready rank next: ``Runtime``), so a run is reproducible.
"""
import argparse
import os
import runpy
import sys
import threading

_tls = threading.local()


READY, RUNNING, BLOCKED, DONE = range(4)


class Runtime:
    """Rank threads under a DETERMINISTIC cooperative schedule: exactly one rank runs at a time (it holds the baton); when
    it blocks -- in ``rospy.sleep`` with a latched command, or inside a collective -- or leaves its script, the baton
    goes to the lowest-ranked rank that is ready to run.  Every interleaving of the ranks' code between two blocking
    points is therefore the same in every run: a rank's ``reset_pose`` teleport (which re-casts ITS lidar against the
    poses the others have at that moment) always sees the same world.  (mpiexec + Stage are racy there; rank threads
    left to the OS scheduler were too, and two runs of ppo_stage1.py differed from their second tick on.  With the GIL
    nothing ran in parallel anyway.)"""

    def __init__(self, size, max_ticks=None):
        self.size = size
        self.max_ticks = max_ticks
        lock = threading.RLock()
        self.cv = threading.Condition(lock)                                     # the runtime's lock
        self.rank_cv = [threading.Condition(lock) for _ in range(size)]         # one wake-up channel per rank: a baton
        self.blocked = 0            # ranks parked in sleep or in a collective (released ranks are
        self.sleepers = 0           # un-counted by the RELEASER, not when they wake up: a woken-late
        self.coll_waiters = 0       # rank must never make the world look quiescent)
        self.sleep_gen = 0          # incremented by every tick
        self.world = None           # SharedWorld, registered by the first facade
        self.shutdown = False
        # collective state
        self.coll_gen = 0
        self.coll_count = 0
        self.coll_buf = [None] * size
        self.coll_result = None
        self.alive = size
        self.errors = []
        # the schedule: ranks start one after another too (rank r + 1 begins when rank r first blocks or exits), so the
        # scripts' racy start-up code (os.makedirs, logging handlers) runs serially like it does when mpiexec staggers
        # process start-up
        self.state = [READY] * size
        self.baton = 0              # the rank that may run (None: everybody is blocked)
        self.sleeping = set()
        self.colling = set()

    # ---- the baton (call with self.cv held)
    def _pass_baton(self):
        """The caller stopped running: the lowest-ranked READY rank goes next (and only that rank is woken: waking all
        of 50 threads per hand-over made a tick of circle_test.py six times slower)."""
        self.baton = next((r for r in range(self.size) if self.state[r] == READY), None)
        if self.baton is not None:
            self.rank_cv[self.baton].notify()

    def _take_baton(self, r, released=lambda: True):
        """Returns once rank r has been released from whatever it blocked on AND holds the baton (or at shutdown)."""
        while not self.shutdown and not (released() and self.baton == r):
            self.rank_cv[r].wait(0.5)
        self.state[r] = RUNNING      # (after a shutdown everybody just leaves: no order to keep)

    def _shut_down(self):
        self.shutdown = True
        for c in self.rank_cv:
            c.notify()

    def begin(self, r):
        with self.cv:
            self._take_baton(r)

    # ---- quiescence / ticking (call with self.cv held)
    def _maybe_tick(self):
        if self.blocked >= self.alive and self.world is not None and self.world.pending():
            self.world.tick()
            self.sleep_gen += 1
            self.blocked -= self.sleepers
            self.sleepers = 0
            for q in self.sleeping:
                self.state[q] = READY
            self.sleeping.clear()
            if self.max_ticks is not None and self.world.ticks >= self.max_ticks:
                self._shut_down()

    def sleep(self, needs_tick):
        """rospy.sleep: with a latched command, wait for the tick that consumes it."""
        r = rank()
        with self.cv:
            if self.shutdown:
                raise KeyboardInterrupt
            if not needs_tick:
                return
            gen = self.sleep_gen
            self.blocked += 1
            self.sleepers += 1
            self.state[r] = BLOCKED
            self.sleeping.add(r)
            self._maybe_tick()
            self._pass_baton()
            self._take_baton(r, lambda: self.sleep_gen != gen)
            if self.sleep_gen == gen:          # shut down before the tick: un-count ourselves
                self.blocked -= 1
                self.sleepers -= 1
                self.sleeping.discard(r)
                raise KeyboardInterrupt

    def collective(self, rank, value, combine):
        """All-rank rendezvous; ``combine(list_of_values)`` runs once, every rank gets its result."""
        with self.cv:
            if self.shutdown:
                raise KeyboardInterrupt
            gen = self.coll_gen
            self.coll_buf[rank] = value
            self.coll_count += 1
            self.coll_combine = combine
            if self.coll_count == self.alive:
                self._complete_collective()      # the last arriver keeps the baton and carries on
                return self.coll_result
            self.blocked += 1
            self.coll_waiters += 1
            self.state[rank] = BLOCKED
            self.colling.add(rank)
            self._maybe_tick()
            self._pass_baton()
            self._take_baton(rank, lambda: self.coll_gen != gen)
            if self.coll_gen == gen:
                self.blocked -= 1
                self.coll_waiters -= 1
                self.colling.discard(rank)
                raise KeyboardInterrupt
            return self.coll_result

    def _complete_collective(self):
        self.coll_result = self.coll_combine(list(self.coll_buf))
        self.coll_buf = [None] * self.size
        self.coll_count = 0
        self.coll_gen += 1
        self.blocked -= self.coll_waiters
        self.coll_waiters = 0
        for q in self.colling:
            self.state[q] = READY
        self.colling.clear()

    def rank_exit(self, err=None):
        """A rank left its script: the remaining ranks' rendezvous / tick conditions shrink."""
        with self.cv:
            self.alive -= 1
            self.state[rank()] = DONE
            if err is not None:
                self.errors.append(err)
                self._shut_down()
            elif self.alive > 0:
                if self.coll_count and self.coll_count == self.alive:
                    self._complete_collective()
                self._maybe_tick()
            self._pass_baton()


_runtime = None


def runtime():
    if _runtime is None:
        raise RuntimeError("not inside `python -m mrca.spmd`: no SPMD runtime is active")
    return _runtime


def rank():
    return getattr(_tls, "rank", 0)


class Comm:
    """The subset of mpi4py's COMM_WORLD the scripts use (ppo_stage1.py:66-100, ppo_stage2.py:105)."""

    def Get_rank(self):
        return rank()

    def Get_size(self):
        return runtime().size

    def gather(self, obj, root=0):
        out = runtime().collective(rank(), obj, lambda vals: vals)
        return out if rank() == root else None

    def scatter(self, objs, root=0):
        def combine(vals):
            src = vals[root]
            if src is None or len(src) != runtime().size:
                raise ValueError("scatter: root must pass one item per rank")
            return list(src)
        return runtime().collective(rank(), objs, combine)[rank()]

    def bcast(self, obj, root=0):
        return runtime().collective(rank(), obj, lambda vals: vals[root])

    def barrier(self):
        runtime().collective(rank(), None, lambda vals: None)

    Barrier = barrier


def run_script(path, nprocs, max_ticks=None, extra_argv=(), chdir=None):
    """Execute ``path`` as ``nprocs`` rank threads.  Returns the list of per-rank exceptions."""
    global _runtime
    here = os.path.dirname(os.path.abspath(__file__))
    dropin = os.path.join(here, "dropin")
    script_dir = os.path.dirname(os.path.abspath(path))
    old_path, old_argv, old_cwd = list(sys.path), list(sys.argv), os.getcwd()
    old_main = sys.modules.get("__main__")     # runpy installs the script as __main__: put the caller's back afterwards
    sys.path[:0] = [dropin, script_dir]
    for shadowed in ("rospy", "tf", "mpi4py", "mpi4py.MPI", "stage_world1", "stage_world2", "circle_world"):
        sys.modules.pop(shadowed, None)
    for name in list(sys.modules):   # the reference's own package: import it afresh, like a new process would
        if name == "model" or name.startswith("model."):   # (model/ppo.py:10-19 sets up ./log on import)
            sys.modules.pop(name, None)
    sys.argv = [path, *extra_argv]
    import builtins
    import functools
    if not hasattr(builtins, "reduce"):      # model/utils.py:85 calls the Python-2 builtin
        builtins.reduce = functools.reduce
    if chdir:
        os.chdir(chdir)
    _runtime = rt = Runtime(nprocs, max_ticks)

    def body(r):
        _tls.rank = r
        err = None
        rt.begin(r)
        try:
            runpy.run_path(path, run_name="__main__")
        except SystemExit:
            pass
        except KeyboardInterrupt:
            pass
        except BaseException as e:  # noqa: BLE001 -- surface any rank failure to the launcher
            import traceback
            traceback.print_exc()
            err = e
        finally:
            rt.rank_exit(err)

    threads = [threading.Thread(target=body, args=(r,), name=f"rank{r}", daemon=True) for r in range(nprocs)]
    try:
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        sys.path[:] = old_path
        sys.argv = old_argv
        os.chdir(old_cwd)
        if old_main is not None:
            sys.modules["__main__"] = old_main
        _runtime = None
    return rt.errors


def main():
    ap = argparse.ArgumentParser(description="run a reference script as N rank threads on the MI355X env")
    ap.add_argument("-np", "--nprocs", type=int, required=True)
    ap.add_argument("--max-ticks", type=int, default=None, help="stop (like Ctrl-C) after this many simulator ticks")
    ap.add_argument("script")
    ap.add_argument("args", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    errs = run_script(a.script, a.nprocs, a.max_ticks, a.args)
    sys.exit(1 if errs else 0)


if __name__ == "__main__":
    main()
