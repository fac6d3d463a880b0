#!/usr/bin/env python3
"""What a tick could cost if the move launch were OFF a world range's chain: the ray casts of P world ranges
launched back to back, each range on a stream of its own, no move launch between them -- with and without a move chain
running NEXT to them on another env (separate memory: a timing probe must not race on the world it measures).

The product's schedule (mrca_step_many, two ranges half a tick apart) is a chain `move, ray, move, ray ...` per range: a
range's period is move + ray.  If tick k's ray cast and tick k + 1's move launch did not depend on each other (double-buffered
poses), a range's period would be its ray cast alone; this probe measures that bound before anything is built for it.

    python tools/ray_only_probe.py            # 4096 robots, Stage-1 worlds; prints us per "tick" (one ray cast of every range)
"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-collision-avoidance_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as G  # noqa: E402

G.build()
import bench  # noqa: E402
from mrca import scenario as S  # noqa: E402
from mrca.vec_env import VecStageWorld  # noqa: E402

K = 64


def measure(env, other, pool, P, with_move, stagger):
    """One hipGraph of K 'ticks': per tick one mrca_observe_worlds per range on the range's stream; with_move: the other env
    runs K full move launches on a further stream at the same time."""
    W = env.W
    ranges = [(c * W // P, (c + 1) * W // P - c * W // P) for c in range(P)]
    dev = env.device
    streams = [torch.cuda.Stream(device=dev) for _ in range(P + 1)]
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream(device=dev)
    cap.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.graph(g, stream=cap):
        cur = torch.cuda.current_stream(dev)
        for s in streams:
            s.wait_stream(cur)
        prev = None
        for c, (s, r) in enumerate(zip(streams[:P], ranges)):
            with torch.cuda.stream(s):
                if stagger and prev is not None:
                    s.wait_event(prev)
                for k in range(K):
                    env.observe(r)
                    if k == 0:
                        prev = torch.cuda.Event()
                        prev.record(s)
        if with_move:
            with torch.cuda.stream(streams[P]):
                for k in range(K):
                    other.move(pool[k % len(pool)], (0, other.W))
        for s in streams:
            cur.wait_stream(s)
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    return statistics.median(ts) / K, min(ts) / K


def main():
    for name, make in (("stage1 128 x 32", lambda: S.stage1(num_worlds=128, robots_per_world=32, seed=1000)),
                       ("stage2 187 x 44", lambda: S.stage2(num_worlds=187, seed=1000))):
        sc = make()
        env, other = VecStageWorld(sc), VecStageWorld(sc)
        pool = bench.action_pool(sc.num_robots, env.device, 1)
        for e in (env, other):
            e.reset()
            for k in range(3):
                e.step(pool[k])
        torch.cuda.synchronize()
        for P in (1, 2, 3, 4):
            for with_move in (False, True):
                for stagger in ((False, True) if P > 1 else (False,)):
                    med, best = measure(env, other, pool, P, with_move, stagger)
                    print(f"{name}: {P} range(s), ray casts back to back{' + a move chain beside them' if with_move else ''}"
                          f"{', ranges one launch apart' if stagger else ''}: {med:6.2f} us per tick (best {best:6.2f}) = "
                          f"{sc.num_robots / med:6.1f} M agent-steps/s bound", flush=True)
        env.close()
        other.close()


if __name__ == "__main__":
    main()
