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
per launch, time from HIP events recorded on the
solver's stream around every launch.  `cpu_baseline` times the NumPy oracle
(the restated reference algorithm) on the host cores of this box, rank 0, N = 1
only, on a bounded sample of the same workload.
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
    ap.add_argument("--grid", type=int, default=2236, help="Brusselator N")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2)
    return ap.parse_args()


def cpu_baseline(N, h, steps):
    """the oracle (NumPy + OpenBLAS restatement of the reference algorithm) on
    the same workload, a bounded number of steps"""
    from oracle import problems as pb
    from oracle import rk_oracle
    y0 = pb.bruss2d_y0(N)
    s = rk_oracle.Pr8(pb.bruss2d_rhs(N), 0.0, y0, 1.0, first_step=h, max_step=h,
                      rtol=1e-6, atol=1e-9, nfev_stiff_detect=0)
    s.step()                                      # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        s.step()
    dt = time.perf_counter() - t0
    return {"value": y0.size * steps / dt, "unit": "state-dim*steps/s",
            "cores": os.cpu_count(), "kind": "port",
            "sample": f"{steps} accepted Pr8 steps of the same Brusselator N={N} "
                      f"(n={y0.size}) after 1 warm-up, NumPy/OpenBLAS oracle, "
                      f"{dt / steps:.2f} s/step"}


def pmc_traffic(N):
    """HBM bytes per stage-accumulate launch from the committed rocprofv3 PMC
    passes of this same command (profiles/rNN_pmc_traffic.json, written by
    tools/profile_bench.sh + tools/summarize_profiles.py); None if absent or
    for another problem size"""
    if N != 2236:
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
    if world != args.gpus:
        if rank == 0 and world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run for --gpus > 1")
    dist = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import torch.distributed as dist_mod      # control plane only (gloo)
        dist_mod.init_process_group("gloo", rank=rank, world_size=world)
        dist = dist_mod

    import extensisq_amd as esq
    from extensisq_amd import lockstep, workloads
    from extensisq_amd._lib import PROF_RHS, PROF_SOLERR, PROF_STAGE

    N = args.grid
    rhs, y0, h = workloads.pr8_brusselator(N, shard=rank)
    n = y0.size
    group = None
    if world > 1:
        group = lockstep.init_lockstep(rank, world, local, n)
    solver = esq.Pr8(rhs, 0.0, y0, 1.0e9, first_step=h, max_step=h, rtol=1e-6,
                     atol=1e-9, nfev_stiff_detect=0, device=local,
                     lockstep=group)
    dev = solver._dev

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
    # ---- timed region: exactly K accepted steps.  Every stage-accumulate
    # launch carries a start/stop HIP event pair (dispatch timestamps on the
    # solver's stream, hipExtLaunchKernelGGL): the roofline figure is measured
    # live over the SAME K steps the throughput is quoted on.
    dev.profile_reset()
    dev.profile_enable([PROF_STAGE])
    nfs0 = int(esq.NFS[()])
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    dev.profile_enable(None)
    rejected = int(esq.NFS[()]) - nfs0
    prof = {PROF_STAGE: dev.profile_read(PROF_STAGE)}
    # ---- the same K steps again without any event: the cost of measuring
    barrier()
    t1 = time.perf_counter()
    run(args.steps)
    barrier()
    elapsed_noprof = time.perf_counter() - t1
    # ---- diagnostic pass (untimed): per-class device time of the other kernels
    dev.profile_reset()
    dev.profile_enable([PROF_RHS, PROF_SOLERR])
    run(min(args.steps, 10))
    dev.profile_enable(None)
    prof[PROF_RHS] = dev.profile_read(PROF_RHS)
    prof[PROF_SOLERR] = dev.profile_read(PROF_SOLERR)

    if dist is not None:
        import torch
        tt = torch.tensor([elapsed, elapsed_noprof], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, elapsed_noprof = float(tt[0]), float(tt[1])
        rj = torch.tensor([rejected], dtype=torch.int64)
        dist.all_reduce(rj, op=dist.ReduceOp.MAX)
        rejected = int(rj[0])

    if rank == 0:
        st_ms, st_cnt, st_bytes = prof[PROF_STAGE]
        rh_ms, rh_cnt, rh_bytes = prof[PROF_RHS]
        se_ms, se_cnt, se_bytes = prof[PROF_SOLERR]
        achieved = st_bytes / (st_ms * 1e-3) / 1e9 if st_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic(N)
        out = {
            "metric": "accepted RK steps/s x state-dim (fp64), Pr8 n=1e7",
            "value": world * n * args.steps / elapsed,
            "unit": "state-dim*steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"Pr8 (13 stages) on 2-D Brusselator reaction-"
                            f"diffusion N={N}, n={n} per GPU, h=1/rho (all "
                            f"steps accepted), device RHS, state resident in HBM",
                "n_per_gpu": n, "global_state_dim": world * n,
                "parallelism": (f"lockstep x{world}: independent IVP per GPU, "
                                "1 fp64 RCCL all-reduce per step")
                if world > 1 else "single GPU",
                "rejected_steps_in_timed_region": rejected,
                "ms_per_step_without_events": 1e3 * elapsed_noprof / args.steps,
            },
            "roofline": {
                "bound": "hbm", "kernel": "k_lincomb (fused stage-accumulate)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_src,
                "launches": st_cnt,
                "avg_launch_us": 1e3 * st_ms / st_cnt if st_cnt else None,
                "algorithmic_bytes_per_launch": st_bytes / st_cnt if st_cnt else None,
                "other_kernels": {
                    "rhs_bruss2d": {"gbs": rh_bytes / (rh_ms * 1e-3) / 1e9
                                    if rh_ms > 0 else None,
                                    "avg_launch_us": 1e3 * rh_ms / rh_cnt
                                    if rh_cnt else None, "launches": rh_cnt},
                    "solution_error": {"gbs": se_bytes / (se_ms * 1e-3) / 1e9
                                       if se_ms > 0 else None,
                                       "avg_launch_us": 1e3 * se_ms / se_cnt
                                       if se_cnt else None, "launches": se_cnt},
                },
                "whole_step_gbs": (1040.0 * n * args.steps) / elapsed / 1e9,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, h, args.cpu_steps)
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
