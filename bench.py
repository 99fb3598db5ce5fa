#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): accepted Pr8 steps/s x state dimension,
fp64, 2-D Brusselator N = 2236 (n = 9 999 392) per GPU, inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (N > 1: launched by torch.distributed.run, RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).  With N > 1 every rank
integrates its own independent IVP of the same size in LOCK-STEP: one fp64 RCCL
all-reduce per step for the global error norm, nothing else crosses xGMI (weak
scaling).  A "step" is one accepted 13-stage Pr8 step: 12 fused
stage-accumulate kernels, 13 RHS kernels, 1 fused solution/error-norm kernel.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel class (the
fused stage-accumulate kernels): algorithmic bytes = 8 B * (nnz(A[i,:i]) + 2) * n
per launch, time from HIP events attached to every
such dispatch on the solver's stream, over the timed region itself.
`cpu_baseline` times the NumPy oracle (the restated reference algorithm) on the
host cores of this box, rank 0, N = 1 only, on a bounded sample of the same
workload.

`--config ts5|pr9|rkc` runs the other BASELINE.json configs through the same
harness (for DESIGN.md's table; the driver uses the default).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", type=int, default=None,
                    help="grid size N of the workload (default: BASELINE.json)")
    ap.add_argument("--config", default="pr8", choices=["pr8", "ts5", "pr9", "rkc"],
                    help="pr8 = the BASELINE.json metric config (default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--sample-every", type=int, default=13,
                    help="attach HIP events to every k-th launch of the dominant "
                         "kernel class in the timed region (pseudo-random 1-in-k "
                         "sampling; an event-carrying dispatch costs "
                         "~6 us of queue time)")
    ap.add_argument("--force-lockstep", action="store_true",
                    help="create the RCCL communicator even for one rank")
    ap.add_argument("--replicas", action="store_true",
                    help="N > 1 without the lock-step collective: fully "
                         "independent solvers per GPU (upper bound, SURVEY.md §8e)")
    ap.add_argument("--dry-run", action="store_true",
                    help="exercise only the multi-rank control plane (gloo "
                         "rendezvous, id exchange, barrier, max-over-ranks, JSON) "
                         "with no GPU work -- used by the CPU tests")
    return ap.parse_args()


def gloo_exchange(dist, rank):
    """ncclUniqueId broadcast + shard-size sum over the gloo control group"""
    import torch

    def exchange(make_id, n_local):
        ident = [make_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        total = torch.tensor([int(n_local)], dtype=torch.int64)
        dist.all_reduce(total)
        return ident[0], int(total[0])
    return exchange


def dry_run(args, rank, world, dist):
    """control-plane rehearsal without a GPU (tests/test_bench_cpu.py)"""
    import torch
    n = 1000 + rank
    ident, n_total = gloo_exchange(dist, rank)(lambda: bytes(range(128)), n) \
        if dist is not None else (bytes(range(128)), n)
    assert len(ident) == 128 and n_total == sum(1000 + r for r in range(world))
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "dry-run", "value": n_total / elapsed,
                          "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "max_elapsed": elapsed}),
              flush=True)
    if dist is not None:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------
def make_workload(name, N, rank):
    """returns a dict: device solver factory, oracle factory, byte counts"""
    import extensisq_amd as esq
    from extensisq_amd import workloads as wl
    from extensisq_amd._lib import PROF_RKC, PROF_STAGE
    if name == "pr8":
        N = N or 2236
        rhs, y0, h = wl.pr8_brusselator(N, shard=rank)
        kw = dict(first_step=h, max_step=h, rtol=1e-6, atol=1e-9,
                  nfev_stiff_detect=0)
        return dict(
            label=f"Pr8 (13 stages) on 2-D Brusselator reaction-diffusion N={N}",
            metric="accepted RK steps/s x state-dim (fp64), Pr8 n=1e7",
            cls=esq.Pr8, oracle="Pr8", rhs=rhs, y0=y0, kw=kw, N=N,
            cpu_problem=("bruss2d_rhs", "bruss2d_y0"),
            bytes_per_elt_step=1040.0, klass=PROF_STAGE,
            kernel="stage-accumulate class: k_lincomb / k_*_chain (RHS of the previous "
                   "stage chained in) / k_block_acc")
    if name == "ts5":
        N = N or 1000
        rhs, y0, h = wl.ts5_heat(N, seed=1234 + rank)
        kw = dict(first_step=h, max_step=h, rtol=1e-6, atol=1e-9,
                  nfev_stiff_detect=0)
        return dict(
            label=f"Ts5 (6 stages, FSAL) on 2-D heat equation N={N}",
            metric="accepted RK steps/s x state-dim (fp64), Ts5 n=1e6",
            cls=esq.Ts5, oracle="Ts5", rhs=rhs, y0=y0, kw=kw, N=N,
            cpu_problem=("heat2d_rhs", "heat2d_y0"),
            bytes_per_elt_step=432.0, klass=PROF_STAGE,
            kernel="stage-accumulate class: k_lincomb / k_*_chain (RHS of the previous "
                   "stage chained in) / k_block_acc")
    if name == "pr9":
        N = N or 2236
        rhs = esq.Heat2D(N)
        y0 = wl.heat2d_y0(N, seed=1234 + rank)
        h = 1.0 / rhs.spectral_radius()
        kw = dict(first_step=h, max_step=h, rtol=1e-6, atol=1e-9,
                  nfev_stiff_detect=0)
        return dict(
            label=f"Pr9 (17 stages) on 2-D heat equation N={N}",
            metric="accepted RK steps/s x state-dim (fp64), Pr9 n=5e6 per GPU",
            cls=esq.Pr9, oracle="Pr9", rhs=rhs, y0=y0, kw=kw, N=N,
            cpu_problem=("heat2d_rhs", "heat2d_y0"),
            bytes_per_elt_step=1624.0, klass=PROF_STAGE,
            kernel="stage-accumulate class: k_lincomb / k_*_chain (RHS of the previous "
                   "stage chained in) / k_block_acc")
    N = N or 159
    rhs, y0, h, rho = wl.rkc_diffusion(N, m_target=100)
    kw = dict(first_step=h, max_step=h, rtol=1e-3, atol=1e-3, const_jac=True,
              rho_jac=lambda t, y: rho)     # max_step pins m ~ 100
    return dict(
        label=f"SSV2stab (RKC, m~100 stages/step) on 3-D diffusion N={N}",
        metric="accepted RKC steps/s x state-dim (fp64), SSV2stab n=4e6",
        cls=esq.SSV2stab, oracle="SSV2stab", rhs=rhs, y0=y0, kw=kw, N=N,
        cpu_problem=("diff3d_rhs", "diff3d_y0"),
        bytes_per_elt_step=None, klass=PROF_RKC,
        kernel="k_rkc_stage (three-term Chebyshev recursion)")


def cpu_baseline(w, steps):
    """the oracle (NumPy + OpenBLAS restatement of the reference algorithm) on
    the same workload, a bounded number of steps"""
    from oracle import problems as pb
    from oracle import rk_oracle, rkc_oracle
    cls = (rkc_oracle.SSV2stab if w["oracle"] == "SSV2stab"
           else rk_oracle.METHODS[w["oracle"]])
    fun = getattr(pb, w["cpu_problem"][0])(w["N"])
    y0 = w["y0"]
    s = cls(fun, 0.0, y0, 1.0e9, **w["kw"])
    s.step()                                      # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        s.step()
    dt = time.perf_counter() - t0
    return {"value": y0.size * steps / dt, "unit": "state-dim*steps/s",
            "cores": os.cpu_count(), "kind": "port",
            "sample": f"{steps} accepted steps of the same workload "
                      f"(n={y0.size}) after 1 warm-up, NumPy/OpenBLAS oracle, "
                      f"{dt / steps:.2f} s/step"}


def pmc_traffic(w):
    """HBM bytes per stage-accumulate launch from the committed rocprofv3 PMC
    passes of the default command (profiles/rNN_pmc_traffic.json, written by
    tools/profile_bench.sh + tools/summarize_profiles.py)"""
    if not (w["cls"].__name__ == "Pr8" and w["N"] == 2236):
        return None, None
    pdir = os.path.join(ROOT, "profiles")
    try:
        names = sorted(f for f in os.listdir(pdir) if f.endswith("_pmc_traffic.json"))
        with open(os.path.join(pdir, names[-1])) as fh:
            data = json.load(fh)
        return (data["stage_accumulate"]["hbm_bytes_per_launch"],
                f"profiles/{names[-1]}")
    except Exception:
        return None, None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        sys.exit("launch with torch.distributed.run for --gpus > 1")
    dist = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import torch.distributed as dist_mod      # control plane only (gloo)
        # gloo announces its connections on stdout; stdout is reserved for the
        # one JSON line, so fd 1 points at stderr while the group forms
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist_mod.init_process_group("gloo", rank=rank, world_size=world)
            dist_mod.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
        dist = dist_mod

    if args.dry_run:
        return dry_run(args, rank, world, dist)

    import extensisq_amd as esq
    from extensisq_amd import lockstep
    from extensisq_amd._lib import PROF_RHS, PROF_RKC, PROF_SOLERR, PROF_STAGE

    w = make_workload(args.config, args.grid, rank)
    n = w["y0"].size
    group = None
    if (world > 1 and not args.replicas) or args.force_lockstep:
        group = lockstep.init_lockstep(
            rank, world, local, n,
            exchange=gloo_exchange(dist, rank) if dist is not None else None)
    solver = w["cls"](w["rhs"], 0.0, w["y0"], 1.0e9, device=local, lockstep=group,
                      **w["kw"])
    dev = solver._dev
    klass = w["klass"]

    def barrier():
        dev.synchronize()
        if dist is not None:
            dist.barrier()

    def run(k):
        for _ in range(k):
            msg = solver.step()
            if msg is not None or solver.status != "running":
                raise RuntimeError(f"step failed: {msg}")

    run(args.warmup)
    # ---- timed region: exactly K accepted steps.  Every launch of the dominant
    # kernel class carries a start/stop HIP event pair (dispatch timestamps on
    # the solver's stream, hipExtLaunchKernelGGL): the roofline figure is
    # measured live over the SAME K steps the throughput is quoted on.
    dev.profile_reset()
    dev.profile_enable([klass], every=args.sample_every)
    nfs0, nfev0 = int(esq.NFS[()]), solver.nfev
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    dev.profile_enable(None)
    rejected = int(esq.NFS[()]) - nfs0
    nfev_timed = solver.nfev - nfev0
    prof = {klass: dev.profile_read(klass)}
    moved_bytes = dev.profile_read_moved(klass)
    # ---- the same K steps again without any event: the cost of measuring
    barrier()
    t1 = time.perf_counter()
    run(args.steps)
    barrier()
    elapsed_noprof = time.perf_counter() - t1
    # ---- diagnostic pass (untimed): device time of the other kernel classes
    others = [k for k in (PROF_STAGE, PROF_RHS, PROF_SOLERR, PROF_RKC) if k != klass]
    dev.profile_reset()
    dev.profile_enable(others)
    run(min(args.steps, 10))
    dev.profile_enable(None)
    for k in others:
        prof[k] = dev.profile_read(k)

    if dist is not None:
        import torch
        tt = torch.tensor([elapsed, elapsed_noprof], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, elapsed_noprof = float(tt[0]), float(tt[1])
        rj = torch.tensor([rejected], dtype=torch.int64)
        dist.all_reduce(rj, op=dist.ReduceOp.MAX)
        rejected = int(rj[0])

    if rank == 0:
        ms, cnt, nbytes = prof[klass]
        achieved = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic(w)

        def klass_info(k):
            kms, kcnt, kb = prof[k]
            return {"gbs": kb / (kms * 1e-3) / 1e9 if kms > 0 else None,
                    "avg_launch_us": 1e3 * kms / kcnt if kcnt else None,
                    "launches": kcnt}
        names = {PROF_STAGE: "stage_accumulate", PROF_RHS: "rhs_plugin",
                 PROF_SOLERR: "solution_error", PROF_RKC: "rkc_stage"}
        out = {
            "metric": w["metric"],
            "value": world * n * args.steps / elapsed,
            "unit": "state-dim*steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"{w['label']}, n={n} per GPU, device RHS, state "
                            f"resident in HBM, steps driven by solver.step()",
                "n_per_gpu": n, "global_state_dim": world * n,
                "parallelism": (f"lockstep x{world}: independent IVP per GPU, "
                                "1 fp64 RCCL all-reduce per step")
                if group is not None else
                (f"replicas x{world}: independent solvers, no collective"
                 if world > 1 else "single GPU"),
                "rejected_steps_in_timed_region": rejected,
                "rhs_evaluations_in_timed_region": nfev_timed,
                "ms_per_step_without_events": 1e3 * elapsed_noprof / args.steps,
            },
            "roofline": {
                "bound": "hbm", "kernel": w["kernel"],
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "note": "achieved = SURVEY.md algorithmic bytes / device time; "
                        "blocked accumulation moves fewer bytes than that "
                        "(moved_gbs), so frac may exceed what the fabric carries",
                "traffic": traffic, "traffic_source": traffic_src,
                "launches_timed": cnt, "sampled_every": args.sample_every,
                "avg_launch_us": 1e3 * ms / cnt if cnt else None,
                "algorithmic_bytes_per_launch": nbytes / cnt if cnt else None,
                # bytes the timed launches are designed to move: below the
                # algorithmic count where blocked accumulation reads a K row
                # once for several stages
                "moved_gbs": moved_bytes / (ms * 1e-3) / 1e9 if ms > 0 else None,
                "other_kernels": {names[k]: klass_info(k) for k in others
                                  if prof[k][1]},
                "whole_step_gbs": (w["bytes_per_elt_step"] * n * args.steps
                                   / elapsed / 1e9)
                if w["bytes_per_elt_step"] else None,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, args.cpu_steps)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    if group is not None:
        dev.synchronize()
        lockstep.destroy_lockstep(group)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
