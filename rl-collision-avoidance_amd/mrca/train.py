"""Command-line trainer: the `__main__` blocks of ppo_stage1.py:137-204 / ppo_stage2.py:144-215 for the
batched device env.

    python -m mrca.train --stage 1 --worlds 128 --robots-per-world 32 --updates 200
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m mrca.train --stage 2 --worlds 187

Kept from the reference: the three log streams under ./log/<hostname>/ (output.log episode lines,
cal.log episode rewards, ppo.log "policy_loss, value_loss, entropy" per minibatch; ppo_stage1.py:
140-162, model/ppo.py:10-19), checkpoints `policy/Stage1_{n}` / `policy/stage2_{n}.pth` every 20
updates with the reference's state_dict keys (ppo_stage1.py:122-124), resume from
`policy/stage1_2.pth` / `policy/stage2.pth` when present (ppo_stage1.py:185-191).
Added: optimizer / step / RNG state next to each checkpoint (`*.state`, one action-noise generator state PER
RANK) for exact resume, agent-steps/s and success / crash / timeout rates per update, and -- opt-in, for the
large-batch regime of thousands of robots per GPU -- a KL-adaptive learning rate (`--kl-target`), bf16 autocast
for the update / the rollout inference, and a periodic circle-test validation (`--circle-every`) that keeps the
best checkpoint so far as `<policy-dir>/best_circle.pth`.
"""
import argparse
import logging
import os
import socket
import sys
import time

import torch

from . import scenario
from .trainer import HParams, Stage1Trainer


def _loggers(rank):
    host = socket.gethostname()
    d = os.path.join(".", "log", host)
    os.makedirs(d, exist_ok=True)
    fmt = logging.Formatter("%(asctime)s - %(levelname)s - %(message)s")
    out = logging.getLogger("mylogger")
    out.setLevel(logging.INFO)
    if not out.handlers:
        fh = logging.FileHandler(os.path.join(d, "output.log"), mode="a")
        fh.setFormatter(fmt)
        out.addHandler(fh)
        if rank == 0:
            out.addHandler(logging.StreamHandler(sys.stdout))
    cal = logging.getLogger("loggercal")
    cal.setLevel(logging.INFO)
    if not cal.handlers:
        cal.addHandler(logging.FileHandler(os.path.join(d, "cal.log"), mode="a"))
    ppo_log = logging.getLogger("loggerppo")
    ppo_log.setLevel(logging.INFO)
    if not ppo_log.handlers:
        ppo_log.addHandler(logging.FileHandler(os.path.join(d, "ppo.log"), mode="a"))
    return out, cal, ppo_log


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--stage", type=int, default=1, choices=[1, 2])
    ap.add_argument("--worlds", type=int, default=128, help="worlds per GPU")
    ap.add_argument("--robots-per-world", type=int, default=24, help="stage 1 only (NUM_ENV, ppo_stage1.py:32)")
    ap.add_argument("--updates", type=int, default=100)
    ap.add_argument("--batch-size", type=int, default=None)
    ap.add_argument("--policy-dir", default="policy")
    ap.add_argument("--save-every", type=int, default=20)       # ppo_stage1.py:123
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--lr", type=float, default=None, help="default: the reference's 5e-5")
    ap.add_argument("--epoch", type=int, default=None)
    ap.add_argument("--horizon", type=int, default=None)
    ap.add_argument("--coeff-entropy", type=float, default=None)
    ap.add_argument("--kl-target", type=float, default=0.0, help="> 0: KL-adaptive learning rate (ppo.KLAdaptiveLR)")
    ap.add_argument("--lr-max", type=float, default=1e-3)
    ap.add_argument("--bf16-update", action="store_true", help="bf16 autocast in the PPO update (opt-in)")
    ap.add_argument("--bf16-inference", action="store_true", help="bf16 autocast in the rollout policy (opt-in)")
    ap.add_argument("--init", default=None, help="state_dict to start from (overrides the resume file)")
    ap.add_argument("--circle-every", type=int, default=0, help="run the circle test every K updates (rank 0)")
    ap.add_argument("--circle-sizes", default="50:25", help="validation circles as 'robots:radius,...' (50:25 = the reference's "
                                                             "circle test); the score of a checkpoint is the MINIMUM success rate")
    ap.add_argument("--circle-worlds", type=int, default=20)
    ap.add_argument("--circle-perturb", default="", help="'dxy,dth': validate on PERTURBED circles (evaluate.perturbed_start) -- "
                                                          "every validation circle a different scenario, drawn with "
                                                          "--circle-seed (keep it apart from the seeds the final evaluation uses)")
    ap.add_argument("--circle-seed", type=int, default=900001)
    ap.add_argument("--update-path", default="fused", choices=["fused", "stock"],
                    help="PPO update through the HIP forward / backward kernels of the conv front end (fused) or MIOpen (stock)")
    ap.add_argument("--circle-ticks", type=int, default=1500)
    ap.add_argument("--max-seconds", type=float, default=0.0, help="stop after this much wall time (0 = no limit)")
    ap.add_argument("--kl-stop", type=float, default=0.0, help="abandon an update when a minibatch's KL exceeds this x kl-target")
    ap.add_argument("--max-grad-norm", type=float, default=0.0, help="> 0: global-norm gradient clipping (opt-in)")
    ap.add_argument("--logstd-min", type=float, default=None, help="floor of the policy's log std (opt-in; e.g. -1.2)")
    ap.add_argument("--mix-circle", type=int, default=0, help="stage 2: add this many 50-robot circle worlds "
                                                                "(scenario.circle_train) to the training mix")
    ap.add_argument("--mix-circles", default="", help="stage 2: extra circle worlds as 'robots:radius:worlds,...' "
                                                         "(scenario.circle_n, e.g. 10:8:40,20:12:30)")
    ap.add_argument("--stock-policy-path", action="store_true", help="rollout inference through the stock PyTorch layers "
                                                                       "instead of the fused fp32 HIP front end")
    ap.add_argument("--graph", action="store_true", help="replay the rollout tick as one hipGraph (measured: no gain at "
                                                          "4096 robots, the tick is GPU-bound; profiles/r02/r02_e_bench_rollout*.json)")
    ap.add_argument("--no-graph", action="store_true", help="(default) launch the rollout tick kernel by kernel")
    ap.add_argument("--tune-gemms", action="store_true", help="time the library GEMM kernels of shapes the recorded choices "
                                                               "(mrca/gemm_tuning.py) do not list -- other batch sizes -- once")
    ap.add_argument("--no-gemm-choices", action="store_true", help="hipBLASLt's default heuristic for every GEMM")
    ap.add_argument("--log-every", type=int, default=1)
    ap.add_argument("--hold-velocity", action="store_true", help="fidelity: Stage's SetSpeed persistence -- a robot that finished "
                                                                  "keeps driving at its last command until its group is done "
                                                                  "(what ppo_stage2.py's dead robots do under stageros), and the "
                                                                  "speed input survives a reset")
    a = ap.parse_args(argv)

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
        if not a.no_gemm_choices:       # recorded TunableOp choices for the learner's plain library GEMMs
            from . import gemm_tuning
            accepted = gemm_tuning.use_recorded_choices(tune_missing=a.tune_gemms)
            if rank == 0:       # (the record is ignored on another ROCm / hipBLASLt version: say which kernels this run uses)
                print("[train] learner GEMMs: " + ("recorded TunableOp choices (mrca/data/gemm_choices_gfx950_rocm72.csv)"
                                                   if accepted else "library default heuristic (no record accepted)") +
                      ("; timing the shapes the record does not list" if a.tune_gemms else ""), file=sys.stderr)
    dist = None
    if world_size > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MRCA_TRAIN_BACKEND", "nccl")      # "gloo": the CPU tests of the multi-rank path
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    out, cal, ppo_log = _loggers(rank)

    from .vec_env import VecStageWorld
    if a.stage == 1:
        sc = scenario.stage1(num_worlds=a.worlds, robots_per_world=a.robots_per_world, seed=a.seed * 1000 + rank,
                             hold_velocity=a.hold_velocity)
        hp = HParams()                                         # ppo_stage1.py:22-35
        resume, pattern = "stage1_2.pth", "Stage1_{}"
    else:
        sc = scenario.stage2(num_worlds=a.worlds, seed=a.seed * 1000 + rank, hold_velocity=a.hold_velocity)
        hp = HParams(batch_size=512, epoch=4)                  # ppo_stage2.py:28-29
        resume, pattern = "stage2.pth", "stage2_{}.pth"
    for k, v in (("learning_rate", a.lr), ("epoch", a.epoch), ("horizon", a.horizon),
                 ("coeff_entropy", a.coeff_entropy)):
        if v is not None:
            setattr(hp, k, v)
    hp.kl_target, hp.lr_max, hp.kl_stop, hp.max_grad_norm = a.kl_target, a.lr_max, a.kl_stop, a.max_grad_norm
    if a.logstd_min is not None:
        hp.logstd_min = a.logstd_min
    hp.rollout_fused = not a.stock_policy_path and not a.bf16_inference
    hp.graph_tick = bool(a.graph) and not a.no_graph
    hp.update_fused = a.update_path == "fused" and not a.bf16_update and torch.cuda.is_available()
    if a.bf16_update:
        hp.update_dtype = torch.bfloat16
    if a.bf16_inference:
        hp.inference_dtype = torch.bfloat16
    env = VecStageWorld(sc)
    if a.mix_circle > 0 or a.mix_circles:
        from .multi_env import ConcatEnv
        parts = [env]
        if a.mix_circle > 0:
            parts.append(VecStageWorld(scenario.circle_train(num_worlds=a.mix_circle, seed=a.seed * 1000 + rank)))
        for spec in filter(None, a.mix_circles.split(",")):
            r, rad, w = spec.split(":")
            parts.append(VecStageWorld(scenario.circle_n(int(r), float(rad), num_worlds=int(w), seed=a.seed * 1000 + rank,
                                                         train=True)))
        env = ConcatEnv(parts)
        out.info("training mix: %d robots of %s + %d robots in circle worlds (%s)", sc.num_robots, sc.name,
                 env.N - sc.num_robots, ", ".join("%d x %d robots" % (p.W, p.R) for p in parts[1:]))
    if a.batch_size:
        hp.batch_size = a.batch_size
    elif env.N > 64:
        # the reference's 1024 / 512 assume 24 / 44 robots (3 / 6 minibatches per epoch).  Minibatches are drawn
        # from THIS rank's T*N rows (ALL robots of the training mix), so the per-rank batch does not depend on the
        # world size: 32 minibatches per epoch per rank; the global batch of one optimiser step is world_size x this.
        hp.batch_size = max(hp.batch_size, env.N * hp.horizon // 32)
    tr = Stage1Trainer(env, hp=hp, dist=dist, seed=a.seed, stage2=(a.stage == 2))
    out.info("per-rank minibatch %d rows, global batch per optimiser step %d, lr %g, epochs %d, horizon %d",
             hp.batch_size, hp.batch_size * world_size, hp.learning_rate, hp.epoch, hp.horizon)
    os.makedirs(a.policy_dir, exist_ok=True)
    f = a.init or os.path.join(a.policy_dir, resume)
    if os.path.exists(f):
        out.info("############Loading Model########### %s", f)
        tr.policy.load_state_dict(torch.load(f, map_location=env.device))
        st = f + ".state"
        if os.path.exists(st) and not a.init:
            extra = torch.load(st, map_location=env.device)
            tr.optimizer.load_state_dict(extra["optimizer"])
            if a.lr is not None:        # an explicit --lr wins over the (possibly KL-adapted) rate of the saved optimizer
                for grp in tr.optimizer.param_groups:
                    grp["lr"] = a.lr
            else:
                out.info("resuming with the saved learning rate %g", tr.optimizer.param_groups[0]["lr"])
            tr.global_update = extra["global_update"]
            gens = extra.get("generators") or [extra["generator"]]
            if rank < len(gens):
                tr.gen.set_state(gens[rank].cpu())
            else:   # resumed on more ranks than were saved: fresh, distinct noise streams for the new ranks
                tr.gen.manual_seed(a.seed * 1000 + rank + 7919 * (tr.global_update + 1))
    elif a.init:
        raise FileNotFoundError(a.init)
    else:
        out.info("############Start Training###########")
    circle_env = None
    best_circle = -1.0

    def gather_generators():
        g = tr.gen.get_state()
        if dist is None:
            return [g]
        states = [None] * world_size
        dist.all_gather_object(states, g)
        return states

    tr.start()
    n_logged = 0
    t_start = time.perf_counter()
    acc_done = torch.zeros(3, device=env.device)       # terminal events since the last log line
    acc_steps, acc_time = 0, 0.0
    for _ in range(a.updates):
        if a.max_seconds:
            stop = torch.tensor([float(time.perf_counter() - t_start > a.max_seconds)], device=env.device)
            if dist is not None:
                dist.all_reduce(stop, op=dist.ReduceOp.MAX)     # every rank leaves the loop at the same update
            if float(stop) > 0:
                out.info("stopping: --max-seconds %.0f reached", a.max_seconds)
                break
        t0 = time.perf_counter()
        ep_done = torch.zeros(3, device=env.device)
        ep_reward = torch.zeros(env.N, device=env.device)
        last_return = torch.full((env.N,), float("nan"), device=env.device)   # return of each robot's last episode
        for _t in range(hp.horizon):
            was_live = env.live.bool().clone()   # stage 2: a finished robot keeps done=1 until its group restarts
            tr.tick()
            ep_reward += torch.where(was_live, env.reward, torch.zeros_like(env.reward))
            d = env.done.bool() & was_live       # count each terminal event once
            # no host round trip inside the horizon: the statistics stay on the device until the update is over
            last_return = torch.where(d, ep_reward, last_return)
            ep_reward = torch.where(d, torch.zeros_like(ep_reward), ep_reward)
            r = env.result
            ep_done += torch.stack([(d & (r == 1)).sum(), (d & (r == 2)).sum(), (d & (r == 3)).sum()]).float()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            dist.all_reduce(ep_done)
        acc_done += ep_done
        acc_steps += env.N * world_size * hp.horizon
        acc_time += dt
        ep_done = acc_done
        tot = max(float(ep_done.sum()), 1.0)
        for row in tr.loss_log[n_logged:]:
            ppo_log.info("{}, {}, {}".format(*[float(x) for x in row]))
        n_logged = len(tr.loss_log)
        for v in last_return[~torch.isnan(last_return)][:8].tolist():
            cal.info(v)
        kl = "" if tr.last_kl is None else "  kl %.4f  lr %.2e" % (tr.last_kl, tr.optimizer.param_groups[0]["lr"])
        if tr.global_update % a.log_every == 0:
            out.info("update %05d  %.0f agent-steps/s  episodes %d  reach %.3f  crash %.3f  timeout %.3f%s",
                     tr.global_update, acc_steps / acc_time, int(tot), float(ep_done[0]) / tot,
                     float(ep_done[1]) / tot, float(ep_done[2]) / tot, kl)
            acc_done = torch.zeros(3, device=env.device)
            acc_steps, acc_time = 0, 0.0
        if tr.global_update % a.save_every == 0:
            gens = gather_generators()          # collective: every rank takes part
            if rank == 0:
                p = os.path.join(a.policy_dir, pattern.format(tr.global_update))
                torch.save(tr.policy.state_dict(), p)
                torch.save({"optimizer": tr.optimizer.state_dict(), "global_update": tr.global_update,
                            "generator": gens[0], "generators": gens}, p + ".state")
                out.info("########################## model saved when update %d times#########################",
                         tr.global_update)
        if a.circle_every and rank == 0 and tr.global_update % a.circle_every == 0:
            from . import evaluate
            if circle_env is None:
                circle_env = []
                for spec in a.circle_sizes.split(","):
                    r, rad = spec.split(":")
                    sc_c = scenario.circle(num_worlds=a.circle_worlds, seed=a.seed) if (int(r), float(rad)) == (50, 25.0) \
                        else scenario.circle_n(int(r), float(rad), num_worlds=a.circle_worlds, seed=a.seed)
                    circle_env.append((spec, VecStageWorld(sc_c)))
            scores = []
            for spec, ce in circle_env:
                m = evaluate.circle_test(ce, evaluate.cnn_policy_fn(tr.policy), a.circle_ticks,
                                         perturb=tuple(float(v) for v in a.circle_perturb.split(",")) if a.circle_perturb else None,
                                         seed=a.circle_seed)
                scores.append(m["success_rate"])
                out.info("circle %05d  [%s]  success %.3f  crash %.3f  unfinished %.3f  ticks %d", tr.global_update, spec,
                         m["success_rate"], m["crash_rate"], m["unfinished_rate"], m["ticks_run"])
            if min(scores) > best_circle:
                best_circle = min(scores)
                torch.save(tr.policy.state_dict(), os.path.join(a.policy_dir, "best_circle.pth"))
                out.info("circle %05d  new best checkpoint: min success %.3f", tr.global_update, best_circle)
    if tr.global_update % a.save_every != 0:      # the last state of a run that stopped between save points
        gens = gather_generators()
        if rank == 0:
            p = os.path.join(a.policy_dir, pattern.format(tr.global_update))
            torch.save(tr.policy.state_dict(), p)
            torch.save({"optimizer": tr.optimizer.state_dict(), "global_update": tr.global_update,
                        "generator": gens[0], "generators": gens}, p + ".state")
    if rank == 0:
        torch.save(tr.policy.state_dict(), os.path.join(a.policy_dir, "last.pth"))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
