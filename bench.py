#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): accepted Pr8 steps/s x state dimension,
fp64, 2-D Brusselator N = 2236 (n = 9 999 392) per GPU, inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W

One process per GPU.  `python bench.py --gpus N` with N > 1 is self-launching:
the parent (which never touches a GPU) starts N fresh rank processes with RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment and relays rank 0's JSON
line; under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus
N` the launcher has already done that and every process is a rank.  The control
plane (rendezvous, barrier, max-over-ranks time) is a plain TCP star
(extensisq_amd.lockstep.ControlGroup): no PyTorch anywhere.

With N > 1 every rank integrates its own independent IVP of the same size in
LOCK-STEP: one fp64 RCCL all-reduce per step for the global error norm, nothing
else crosses xGMI (weak scaling).  The no-collective variant (`replicas`) is
timed in the same run and reported beside it.

A "step" is one accepted 13-stage Pr8 step: W untimed warm-up steps of the
measured solver, then exactly K timed ones.  Right before the warm-up steps the
DEVICE is loaded for `--device-warmup-ms` (40) with steps of a scratch solver:
an MI355X coming out of idle runs its first ~25 steps up to 15 % slower (power
management, see device_warmup()); `config.cold_start` reports the same W + K
window without that help, `config.sustained` 2 000 steps in one go.  The timed
region carries no events.  Right after it the same K steps are replayed with a HIP event pair on
EVERY kernel launch (dispatch timestamps on the solver's stream); from that
replay come the per-kernel table and `roofline`:
    achieved = the COMPULSORY bytes of the dominant kernel class (every vector a
               launch reads counted once, every output once: `lower_bound_bytes`)
               / its device time
    frac     = achieved / 8 TB/s                                   (<= 1)
    l2_side_gbs = the same with the halo points the marching sweeps' tiles read
               twice included (mostly L2 hits: not memory-interface traffic)
    traffic  = bytes per launch the rocprofv3 PMC counters saw (profiles/); it
               lies between the two byte counts above
    algorithmic_gbs = SURVEY.md §8d bytes / the same time (exceeds the fabric
               rate: blocked accumulation and chaining move fewer bytes)
`cpu_baseline` times the NumPy oracle (the restated reference algorithm) on the
host cores of this box, rank 0, N = 1 only, on a bounded sample of the workload.

`--config ts5|pr9|rkc` runs the other BASELINE.json configs through the same
harness (for DESIGN.md's table; the driver uses the default).
"""
import argparse
import json
import os
import socket
import subprocess
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
    ap.add_argument("--plugin", default=None, choices=["diff3d"],
                    help="run the config's METHOD on another device RHS plugin: diff3d = "
                         "3-D diffusion (default grid 159, n = 4 019 679); not the "
                         "BASELINE.json workload -- for DESIGN.md's plugin table")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=4)
    ap.add_argument("--no-solve-ivp", action="store_true",
                    help="skip the plain solve_ivp(...) figure (PCIe-inclusive)")
    ap.add_argument("--device-warmup-ms", type=float, default=40.0,
                    help="milliseconds of steps of a scratch solver right before the "
                         "measured solver's warm-up steps (the device's power "
                         "management settles; 0: none)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the sustained / generic-plugin / adaptive figures")
    ap.add_argument("--sustained-steps", type=int, default=2000)
    ap.add_argument("--force-lockstep", action="store_true",
                    help="create the RCCL communicator even for one rank")
    ap.add_argument("--replicas", action="store_true",
                    help="N > 1 without the lock-step collective only: fully "
                         "independent solvers per GPU (upper bound, SURVEY.md §8e)")
    ap.add_argument("--timeout", type=float, default=1500.0,
                    help="self-launched ranks are stopped after this many seconds")
    ap.add_argument("--dry-run", action="store_true",
                    help="exercise only the multi-rank control plane (spawn, "
                         "rendezvous, id exchange, barrier, max-over-ranks, JSON) "
                         "with no GPU work -- used by the CPU tests")
    return ap.parse_args()


# ---------------------------------------------------------------------------
# self-launch: the parent never imports the package nor touches a GPU
# ---------------------------------------------------------------------------
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args):
    """start `args.gpus` fresh rank processes of this script, relay rank 0's
    stdout, return non-zero if any rank fails or the time limit passes"""
    world = args.gpus
    port = free_port()
    ctl_port = free_port()
    while ctl_port in range(port, port + 2):      # MASTER_PORT(+1) belong to the launcher
        ctl_port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   ESQ_CTL_PORT=str(ctl_port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
            stdout=subprocess.PIPE if rank == 0 else sys.stderr, cwd=ROOT))
    # rank 0's stdout is drained while the ranks run: a full pipe (a large kernel
    # table, a library banner) must not block it in write()
    import threading
    chunks = []

    def drain():
        for block in iter(lambda: procs[0].stdout.read(65536), b""):
            chunks.append(block)

    reader = threading.Thread(target=drain, daemon=True)
    reader.start()
    deadline = time.time() + args.timeout
    rc = 0
    live = set(range(world))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is not None:
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {r} exited with {code}", file=sys.stderr)
        if rc != 0 or time.time() > deadline:
            if time.time() > deadline and rc == 0:
                rc = 124
                print("bench.py: time limit reached", file=sys.stderr)
            for r in live:                     # exactly the PIDs started above
                procs[r].terminate()
            t_kill = time.time() + 10
            for r in live:
                try:
                    procs[r].wait(timeout=max(0.1, t_kill - time.time()))
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            break
        time.sleep(0.05)
    reader.join(timeout=30)
    out = b"".join(chunks).decode()
    if rc == 0:
        sys.stdout.write(out)
        sys.stdout.flush()
    else:
        sys.stderr.write(out)
    return rc


def pin_to_gpu_numa(local):
    """Before this rank touches its GPU: run on the CPUs of that GPU's NUMA node
    (the pinned result slot, the control plane and Python's controller then sit
    next to the device's PCIe root).  Plain sysfs reads and sched_setaffinity in
    this fresh process -- no GPU call, no re-exec.  The `local`-th AMD display /
    processing-accelerator function in PCI order is taken to be HIP device
    `local`; anything missing or inconclusive (no NUMA information, a restricted
    visibility list) leaves the affinity alone.  Returns what was done."""
    try:
        if os.environ.get("ESQ_BENCH_NO_PIN"):
            return {"pinned": False, "why": "ESQ_BENCH_NO_PIN set"}
        # visibility lists of plain indices are followed (HIP's list indexes into
        # ROCr's); anything else (UUIDs) is not guessed at
        index = local
        for key in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
            val = os.environ.get(key)
            if key == "CUDA_VISIBLE_DEVICES" and os.environ.get("HIP_VISIBLE_DEVICES"):
                continue
            if val:
                items = [v.strip() for v in val.split(",")]
                if not all(v.isdigit() for v in items) or index >= len(items):
                    return {"pinned": False, "why": f"{key}={val!r} not followed"}
                index = int(items[index])
        base = "/sys/bus/pci/devices"
        gpus = []
        for dev in sorted(os.listdir(base)):
            try:
                with open(os.path.join(base, dev, "vendor")) as fh:
                    if fh.read().strip() != "0x1002":
                        continue
                with open(os.path.join(base, dev, "class")) as fh:
                    klass = int(fh.read().strip(), 16) >> 8
                if klass not in (0x0300, 0x0302, 0x0380, 0x1200):
                    continue
                with open(os.path.join(base, dev, "numa_node")) as fh:
                    gpus.append((dev, int(fh.read().strip())))
            except (OSError, ValueError):
                continue
        if index >= len(gpus) or gpus[index][1] < 0:
            return {"pinned": False, "why": "no NUMA node known for this GPU",
                    "gpus_in_sysfs": len(gpus)}
        node = gpus[index][1]
        with open(f"/sys/devices/system/node/node{node}/cpulist") as fh:
            cpus = set()
            for part in fh.read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return {"pinned": False, "why": "node CPUs not in this process's set"}
        os.sched_setaffinity(0, cpus)
        return {"pinned": True, "numa_node": node, "cpus": len(cpus), "pci": gpus[index][0]}
    except Exception as exc:                                   # noqa: BLE001
        return {"pinned": False, "why": repr(exc)}


def repin_to_pci(bdf):
    """affinity of this process -> the CPUs of the NUMA node of PCI function `bdf`"""
    try:
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as fh:
            node = int(fh.read().strip())
        if node < 0:
            return {"repinned": False, "why": "no NUMA node known for " + bdf}
        with open(f"/sys/devices/system/node/node{node}/cpulist") as fh:
            cpus = set()
            for part in fh.read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        os.sched_setaffinity(0, cpus)
        return {"repinned": True, "numa_node": node, "cpus": len(cpus), "pci": bdf,
                "pci_matches": True}
    except Exception as exc:                                   # noqa: BLE001
        return {"repinned": False, "why": repr(exc)}


def dry_run(args, rank, world, ctl):
    """control-plane rehearsal without a GPU (tests/test_bench_cpu.py): spawn,
    rendezvous, id exchange, barrier, min/max-over-ranks timing -- and the SAME
    JSON line the real run prints (`assemble`), filled with placeholder kernel
    rows, so that the shape of the N = 8 line is checked before an 8-GPU node
    ever sees it"""
    n = 1000 + rank
    ident, n_total, offset = ctl.exchange(lambda: bytes(range(128)), n)
    assert len(ident) == 128 and n_total == sum(1000 + r for r in range(world))
    assert offset == sum(1000 + r for r in range(rank))
    ctl.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    mine = time.perf_counter() - t0
    elapsed = ctl.allreduce([mine], "max")[0]
    fastest = ctl.allreduce([mine], "min")[0]
    ctl.barrier()
    if os.environ.get("ESQ_BENCH_DRY_FAIL_RANK") == str(rank):
        sys.exit(3)                       # a failing rank must fail the whole run
    if rank == 0:
        meta = workload_meta(args.config, args.grid, args.plugin)
        table = {"dry-run": {"class": meta["klass"], "launches": args.steps,
                             "total_ms": 1e3 * elapsed, "moved_bytes": 8.0 * n,
                             "floor_bytes": 8.0 * n, "algorithmic_bytes": 8.0 * n}}
        timing = dict(elapsed=elapsed, elapsed_min=fastest, elapsed_prof=elapsed,
                      rejected=0, nfev_timed=0)
        lock = world > 1 and not args.replicas
        # (the real run raises when ncclCommCount disagrees with the world size:
        # rehearsed here with a number from the environment)
        seen = int(os.environ.get("ESQ_BENCH_DRY_RCCL_NRANKS", world))
        if lock and seen != world:
            raise RuntimeError(f"RCCL sees {seen} ranks, expected {world}")
        out = assemble(args, meta, world, 1000, timing, table,
                       lockstep_on=lock, rccl_nranks=world if lock else None,
                       preflight="dry-run" if lock else None,
                       replicas={"value": 1.0, "ms_per_step": 1.0} if lock else None,
                       allreduce_us={"median": 0.0, "p99": 0.0,
                                     "median_without_collective": 0.0,
                                     "collective_median": 0.0, "calls": 0,
                                     "path": "dry-run"} if lock else None,
                       affinity=pin_to_gpu_numa(rank))
        if lock and args.config != "pr9":
            m9 = workload_meta("pr9", None)
            out["config"]["config5_pr9_lockstep"] = {
                "workload": m9["label"] + ", dry-run", "value": 1.0, "ms_per_step": 1.0,
                "ms_per_step_rank_min": 1.0, "rejected_steps_in_timed_region": 0}
        out["metric"] = "dry-run"
        out["value"] = n_total / elapsed
        out["max_elapsed"] = elapsed
        print(json.dumps(out), flush=True)
    ctl.close()


# ---------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------
STAGE_KERNEL = ("stage class: marching chain sweeps (up to 4 RHS evaluations + their "
                "stage arithmetic per launch) / one-stage RHS sweeps with the next "
                "stage's accumulate or the blocked accumulation as epilogue")


def workload_meta(name, N, plugin=None):
    """what the JSON says about a config (no arrays, no GPU)"""
    PROF_STAGE, PROF_RKC = 0, 3               # extensisq_amd._lib.PROF_*
    if plugin == "diff3d" and name != "rkc":
        N = N or 159
        return dict(label=f"{name.capitalize()} on 3-D diffusion N={N} (plugin run, not "
                          f"the BASELINE.json workload)",
                    metric=f"accepted RK steps/s x state-dim (fp64), {name.capitalize()} "
                           f"on the 3-D plugin",
                    bytes_per_elt_step=None, klass=PROF_STAGE, kernel=STAGE_KERNEL)
    if name == "pr8":
        N = N or 2236
        return dict(label=f"Pr8 (13 stages) on 2-D Brusselator reaction-diffusion N={N}",
                    metric="accepted RK steps/s x state-dim (fp64), Pr8 n=1e7",
                    bytes_per_elt_step=1040.0, klass=PROF_STAGE, kernel=STAGE_KERNEL)
    if name == "ts5":
        N = N or 1000
        return dict(label=f"Ts5 (6 stages, FSAL) on 2-D heat equation N={N}",
                    metric="accepted RK steps/s x state-dim (fp64), Ts5 n=1e6",
                    bytes_per_elt_step=432.0, klass=PROF_STAGE, kernel=STAGE_KERNEL)
    if name == "pr9":
        N = N or 2236
        return dict(label=f"Pr9 (17 stages) on 2-D heat equation N={N}",
                    metric="accepted RK steps/s x state-dim (fp64), Pr9 n=5e6 per GPU",
                    bytes_per_elt_step=1624.0, klass=PROF_STAGE, kernel=STAGE_KERNEL)
    N = N or 159
    return dict(label=f"SSV2stab (RKC, m~100 stages/step) on 3-D diffusion N={N}",
                metric="accepted RKC steps/s x state-dim (fp64), SSV2stab n=4e6",
                bytes_per_elt_step=None, klass=PROF_RKC,
                kernel="RKC stage class: rkc_chain<D> (D consecutive Chebyshev stages "
                       "in one marching sweep of the 3-D plugin; -first forms the first "
                       "iterate, -end also evaluates f(t+h, y) and the error estimate) / "
                       "rhs_rkc (one stage: stencil sweep + three-term recursion)")


def make_workload(name, N, rank, plugin=None):
    """returns a dict: device solver factory, oracle factory, byte counts"""
    import extensisq_amd as esq
    from extensisq_amd import workloads as wl
    from extensisq_amd._lib import PROF_RKC, PROF_STAGE
    if plugin == "diff3d" and name != "rkc":
        N = N or 159
        rhs = esq.Diffusion3D(N)
        h = 1.0 / rhs.spectral_radius()
        cls = {"pr8": "Pr8", "ts5": "Ts5", "pr9": "Pr9"}[name]
        meta = workload_meta(name, N, plugin)
        return dict(label=meta["label"], metric=meta["metric"], cls=getattr(esq, cls),
                    oracle=cls, rhs=rhs, y0=wl.diff3d_y0(N),
                    kw=dict(first_step=h, max_step=h, rtol=1e-3, atol=1e-6,
                            nfev_stiff_detect=0),
                    N=N, cpu_problem=("diff3d_rhs", "diff3d_y0"),
                    bytes_per_elt_step=None, klass=PROF_STAGE, kernel=STAGE_KERNEL)
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
            bytes_per_elt_step=1040.0, klass=PROF_STAGE, kernel=STAGE_KERNEL)
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
            bytes_per_elt_step=432.0, klass=PROF_STAGE, kernel=STAGE_KERNEL)
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
            bytes_per_elt_step=1624.0, klass=PROF_STAGE, kernel=STAGE_KERNEL)
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
        kernel="RKC stage class: rkc_chain<D> (D consecutive Chebyshev stages in one "
               "marching sweep of the 3-D plugin; -first forms the first iterate, -end also "
               "evaluates f(t+h, y) and the error estimate) / rhs_rkc (one stage: stencil "
               "sweep + three-term recursion)")


def blas_threads():
    """threads the NumPy BLAS really uses on this box"""
    try:
        from threadpoolctl import threadpool_info
        pools = [p for p in threadpool_info() if p.get("user_api") == "blas"]
        if pools:
            return max(int(p["num_threads"]) for p in pools), pools[0].get(
                "internal_api", "blas")
    except Exception:
        pass
    return None, None


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
    nthr, api = blas_threads()
    return {"value": y0.size * steps / dt, "unit": "state-dim*steps/s",
            "cores": nthr or 1, "host_cpus": os.cpu_count(), "blas": api,
            "kind": "port",
            "sample": f"{steps} accepted steps of the same workload "
                      f"(n={y0.size}) after 1 warm-up, NumPy oracle, "
                      f"{dt / steps:.2f} s/step; BLAS threads = {nthr} (gemv "
                      f"parts), the elementwise NumPy passes and the RHS are "
                      f"single-threaded"}


def pmc_traffic(config):
    """HBM bytes per launch of the dominant kernel class from the committed
    rocprofv3 PMC passes (profiles/rNN_pmc_traffic_<config>.json, written by
    tools/profile_bench.sh + tools/summarize_profiles.py)"""
    pdir = os.path.join(ROOT, "profiles")
    try:
        names = sorted(f for f in os.listdir(pdir)
                       if f.endswith(f"_pmc_traffic_{config}.json"))
        with open(os.path.join(pdir, names[-1])) as fh:
            data = json.load(fh)
        return (data["dominant_class"]["hbm_bytes_per_launch"],
                f"profiles/{names[-1]}")
    except Exception:
        return None, None


def assemble(args, meta, world, n, timing, table, lockstep_on, rccl_nranks,
             preflight, replicas, allreduce_us=None, affinity=None):
    """the ONE JSON line (rank 0).  `table`: per-kernel totals of the profiled
    replay {label: {class, launches, total_ms, moved_bytes, algorithmic_bytes}}"""
    elapsed = timing["elapsed"]
    rows = {}
    for name, r in table.items():
        launches, ms = r["launches"], r["total_ms"]
        rows[name] = {
            "class": r["class"], "launches": launches,
            "avg_us": 1e3 * ms / launches if launches else None,
            "moved_bytes_per_launch": r["moved_bytes"] / launches if launches else None,
            "floor_bytes_per_launch": r["floor_bytes"] / launches if launches else None,
            "algorithmic_bytes_per_launch":
                r["algorithmic_bytes"] / launches if launches else None,
            "gbs": r["floor_bytes"] / (ms * 1e-3) / 1e9 if ms > 0 else None,
            # this kernel's own fraction of the HBM peak on COMPULSORY bytes (VERDICT r05
            # item 3: per kernel, not only for the dominant class)
            "frac": r["floor_bytes"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else None,
            "l2_side_gbs": r["moved_bytes"] / (ms * 1e-3) / 1e9 if ms > 0 else None}
    dom = [r for r in table.values() if r["class"] == meta["klass"]]
    ms = sum(r["total_ms"] for r in dom)
    cnt = sum(r["launches"] for r in dom)
    moved = sum(r["moved_bytes"] for r in dom)
    floor = sum(r["floor_bytes"] for r in dom)
    alg = sum(r["algorithmic_bytes"] for r in dom)
    achieved = floor / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    traffic, traffic_src = pmc_traffic("_".join(
        [args.config] + ([args.plugin] if args.plugin else [])
        + ([str(args.grid)] if args.grid else [])))
    all_moved = sum(r["moved_bytes"] for r in table.values())
    all_floor = sum(r["floor_bytes"] for r in table.values())
    all_ms = sum(r["total_ms"] for r in table.values())
    return {
        "metric": meta["metric"],
        "value": world * n * args.steps / elapsed,
        "unit": "state-dim*steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": f"{meta['label']}, n={n} per GPU, device RHS, state "
                        f"resident in HBM, steps driven by solver.step()",
            "n_per_gpu": n, "global_state_dim": world * n,
            "parallelism": (f"lockstep x{world}: independent IVP per GPU, "
                            "1 fp64 RCCL all-reduce per step")
            if lockstep_on else
            (f"replicas x{world}: independent solvers, no collective"
             if world > 1 else "single GPU"),
            "rccl_nranks": rccl_nranks,
            "rccl_preflight": preflight,
            "replicas_no_collective": replicas,
            # lock-step only: one reduction with / without the RCCL all-reduce
            # (lockstep.time_allreduce), and what lock-step costs per step against
            # fully independent solvers of the same run
            "allreduce_us": allreduce_us,
            "lockstep_minus_replicas_ms": (1e3 * elapsed / args.steps - replicas["ms_per_step"])
            if replicas else None,
            # rank 0's CPU affinity (every rank pins itself to its GPU's NUMA node)
            "cpu_affinity": affinity,
            # skew between the ranks: the timed region of the slowest / fastest
            "ms_per_step_rank_max": 1e3 * elapsed / args.steps,
            "ms_per_step_rank_min": 1e3 * timing["elapsed_min"] / args.steps,
            "rejected_steps_in_timed_region": timing["rejected"],
            "rhs_evaluations_in_timed_region": timing["nfev_timed"],
            "ms_per_step_profiled_replay": 1e3 * timing["elapsed_prof"] / args.steps,
            # the device (not the measured solver) is loaded for this long right
            # before the W warm-up steps; `cold_start`: the same window without it
            "device_warmup_ms": args.device_warmup_ms,
            "cold_start": timing.get("cold"),
            # N = 1 only (rank 0 measures them after the timed region)
            "solve_ivp": None, "sustained": None, "generic_plugin": None,
            "adaptive": None, "rows_kept": None,
            # N > 1, lock-step, any config but pr9: BASELINE.json configs[4] (Pr9, one
            # heat IVP per GPU) measured in the same command after the headline
            "config5_pr9_lockstep": None,
        },
        "roofline": {
            "bound": "hbm", "kernel": meta["kernel"],
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "definition": "compulsory bytes of the class (every vector a launch "
                          "reads counted once, every output written once; the halo "
                          "points its tiles read twice NOT counted) / its device "
                          "time (HIP events on every launch of a K-step replay "
                          "right after the timed region)",
            # the halo-inclusive designed traffic on the same time: what the L2s
            # serve, not what crosses the memory interface
            "l2_side_gbs": moved / (ms * 1e-3) / 1e9 if ms > 0 else None,
            "traffic": traffic, "traffic_source": traffic_src,
            # committed PMC bytes per launch on the live launch time
            "traffic_frac": (traffic / (ms / cnt * 1e-3) / 1e9 / HBM_PEAK_GBS)
            if traffic and cnt and ms > 0 else None,
            # compulsory traffic of the launch sequence actually run, per step
            "lower_bound_bytes": all_floor / args.steps,
            "lower_bound_bytes_per_launch": floor / cnt if cnt else None,
            "launches_timed": cnt,
            "avg_launch_us": 1e3 * ms / cnt if cnt else None,
            "moved_bytes_per_launch": moved / cnt if cnt else None,
            "algorithmic_bytes_per_launch": alg / cnt if cnt else None,
            "algorithmic_gbs": alg / (ms * 1e-3) / 1e9 if ms > 0 else None,
            "device_busy_frac_replay":
                all_ms * 1e-3 / timing["elapsed_prof"] if timing["elapsed_prof"] else None,
            # every kernel of the step: moved bytes / wall time of the
            # TIMED region (launch gaps and the host controller included)
            "whole_step_gbs": all_floor / elapsed / 1e9,
            "whole_step_l2_side_gbs": all_moved / elapsed / 1e9,
            "whole_step_algorithmic_gbs": (
                meta["bytes_per_elt_step"] * n * args.steps / elapsed / 1e9)
            if meta["bytes_per_elt_step"] else None,
            "kernels": rows,
        },
        "cpu_baseline": None,
    }


def raw_table(dev):
    """per-kernel totals of the profiled replay"""
    return {name: {"class": klass, "launches": launches, "total_ms": ms,
                   "moved_bytes": moved, "floor_bytes": floor, "algorithmic_bytes": alg}
            for name, klass, launches, ms, alg, moved, floor in dev.profile_kernels()}


def main():
    args = parse()
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("ESQ_BENCH_ONE_DEVICE"):
        local = 0       # rehearsal of the multi-rank flow on a one-GPU box (--replicas)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

    from extensisq_amd import lockstep            # no GPU call on import
    ctl = lockstep.ControlGroup(rank, world)
    if args.dry_run:
        return dry_run(args, rank, world, ctl)
    affinity = pin_to_gpu_numa(local)              # before the first GPU call

    import extensisq_amd as esq
    from extensisq_amd._lib import (PROF_RHS, PROF_RKC, PROF_SOLERR, PROF_STAGE,
                                    device_count)
    visible = device_count()
    if visible < 1:
        raise RuntimeError("no GPU visible to this rank")
    if local >= visible:
        # the launcher restricted this rank's visibility (one GPU per rank)
        local = local % visible
    if affinity.get("pinned"):
        # was the function pinned to (by position in sysfs, before any GPU call) the
        # device this rank now drives?  HIP's enumeration need not follow PCI order.
        from extensisq_amd._lib import device_pci_bus_id
        got = device_pci_bus_id(local)
        affinity["hip_pci"] = got
        affinity["pci_matches"] = (got == str(affinity.get("pci", "")).lower()) if got else None
        if got and not affinity["pci_matches"]:
            # the guess by position was wrong: move to the CPUs of the node the device
            # really hangs on (the pinned result slot was allocated under the first
            # affinity; everything allocated from here on is local)
            affinity.update(repin_to_pci(got))

    meta = workload_meta(args.config, args.grid, args.plugin)
    w = make_workload(args.config, args.grid, rank, args.plugin)
    n = w["y0"].size
    group = None
    rccl_nranks = None
    preflight = None
    allreduce_us = None
    try:
        if (world > 1 and not args.replicas) or args.force_lockstep:
            group = lockstep.init_lockstep(rank, world, local, n,
                                           exchange=ctl.exchange)
            rccl_nranks = lockstep.comm_size(group)
            if rccl_nranks != world:
                raise RuntimeError(f"RCCL sees {rccl_nranks} ranks, expected {world}")
            # before anything is timed: the collectives the run depends on,
            # checked against values every rank can compute for itself
            preflight = lockstep.preflight(group, rank, world, local, ctl)
            allreduce_us = lockstep.time_allreduce(group, local, ctl)

        def timed(lock_group, w=w):
            solver = w["cls"](w["rhs"], 0.0, w["y0"], 1.0e9, device=local,
                              lockstep=lock_group, **w["kw"])
            dev = solver._dev

            def barrier():
                dev.synchronize()
                ctl.barrier()

            def run(k):
                for _ in range(k):
                    msg = solver.step()
                    if msg is not None or solver.status != "running":
                        raise RuntimeError(f"step failed: {msg}")

            scratch, cold = device_warmup(w, local, args.device_warmup_ms, args.warmup,
                                          args.steps)
            run(args.warmup)
            nfs0, nfev0 = int(esq.NFS[()]), solver.nfev
            # ---- timed region: exactly K accepted steps, no events attached
            barrier()
            t0 = time.perf_counter()
            run(args.steps)
            barrier()
            mine = time.perf_counter() - t0
            rejected = int(esq.NFS[()]) - nfs0
            nfev_timed = solver.nfev - nfev0
            del scratch
            elapsed, rejected = ctl.allreduce([mine, rejected], "max")
            fastest = ctl.allreduce([mine], "min")[0]
            return solver, run, barrier, elapsed, fastest, int(rejected), nfev_timed, cold

        solver, run, barrier, elapsed, fastest, rejected, nfev_timed, cold = timed(group)
        dev = solver._dev
        # ---- profiled replay: the same K steps with an event pair on EVERY
        # launch (all ranks step -- the lock-step collective needs them all)
        dev.profile_reset()
        dev.profile_enable([PROF_STAGE, PROF_RHS, PROF_SOLERR, PROF_RKC], every=1)
        barrier()
        t1 = time.perf_counter()
        run(args.steps)
        barrier()
        elapsed_prof = time.perf_counter() - t1
        dev.profile_enable(None)
        table = raw_table(dev)

        replicas = None
        if world > 1 and group is not None:
            # the no-collective upper bound, same run
            _s2, _r2, _b2, el2, _f2, _rj2, _nf2, _c2 = timed(None)
            replicas = {"value": world * n * args.steps / el2,
                        "ms_per_step": 1e3 * el2 / args.steps}
            del _s2, _r2, _b2
        config5 = None
        if world > 1 and group is not None and args.config != "pr9" and not args.no_extras:
            # BASELINE.json configs[4] proper in the same command: Pr9, one heat IVP of
            # N = 2236 per GPU (seeds 1234 + rank), lock-step over the SAME communicator
            # (a second group object: the norm is taken over this workload's sizes)
            from extensisq_amd.common import LockstepGroup
            w9 = make_workload("pr9", None, rank)
            n9 = w9["y0"].size
            g9 = LockstepGroup(group.comm, world * n9, offset=rank * n9)
            _s9, _r9, _b9, el9, fast9, rej9, _nf9, _c9 = timed(g9, w9)
            config5 = {"workload": w9["label"] + f", n={n9} per GPU, seeds 1234 ... "
                                                 f"{1234 + world - 1}, lock-step",
                       "value": world * n9 * args.steps / el9,
                       "ms_per_step": 1e3 * el9 / args.steps,
                       "ms_per_step_rank_min": 1e3 * fast9 / args.steps,
                       "rejected_steps_in_timed_region": rej9}
            del _s9, _r9, _b9
    except BaseException:
        # a rank that dies must not leave its peers blocked in the all-reduce
        lockstep.abort_lockstep(group)
        raise

    if rank == 0:
        timing = dict(elapsed=elapsed, elapsed_min=fastest, elapsed_prof=elapsed_prof,
                      rejected=rejected, nfev_timed=nfev_timed, cold=cold)
        out = assemble(args, meta, world, n, timing, table,
                       lockstep_on=group is not None, rccl_nranks=rccl_nranks,
                       preflight=preflight, replicas=replicas,
                       allreduce_us=allreduce_us, affinity=affinity)
        out["config"]["config5_pr9_lockstep"] = config5
        if world == 1:
            # further driver-visible figures of the same workload -- none of them
            # the headline
            out["config"]["solve_ivp"] = (None if args.no_solve_ivp
                                          else solve_ivp_figure(w, local))
            out["config"]["sustained"] = (None if args.no_extras
                                          else sustained_figure(solver, run, barrier,
                                                                args.sustained_steps))
            out["config"]["generic_plugin"] = (None if args.no_extras
                                               else generic_plugin_figure(w, local, args))
            out["config"]["adaptive"] = (None if args.no_extras
                                         else adaptive_figure(w, local, esq))
            out["config"]["rows_kept"] = (None if args.no_extras
                                          else rows_kept_figure(w, local, args))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, args.cpu_steps)
        print(json.dumps(out), flush=True)

    if group is not None:
        dev.synchronize()
        ctl.barrier()
        lockstep.destroy_lockstep(group)
    ctl.barrier()
    ctl.close()


def device_warmup(w, device, ms, warmup, steps):
    """Load the DEVICE for `ms` milliseconds with steps of a scratch solver (same
    workload, its own slab) right before the measured solver's W warm-up steps.
    After as little as 50 ms of idling (the construction of a solver is enough) an
    MI355X runs this workload fast for 2-3 steps, 10-15 % slower for the next ~10
    (0.60-0.62 ms) and then relaxes over ~25 ms to its sustained clock (0.50;
    profiles/r03_experiments.md section 17) -- its power management, not the
    solver: the kernels' own durations follow the same curve, and a solver that has
    been stepping shows none of it.  W = 5 warm-up steps of 0.5 ms end before that
    transient does.  The measured solver still takes exactly W untimed and K timed
    steps; the duration is reported in the JSON line (`config.device_warmup_ms`,
    0 switches it off) next to `config.sustained`, which needs no such help."""
    if ms <= 0:
        return None, None
    s = w["cls"](w["rhs"], 0.0, w["y0"], 1.0e9, device=device, **w["kw"])
    t0 = time.perf_counter()
    # the scratch solver's own first W + K steps, timed the same way: what the
    # measured solver would read without this phase (`config.cold_start`)
    for _ in range(warmup):
        s.step()
    s._dev.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        s.step()
    s._dev.synchronize()
    cold = {"ms_per_step": 1e3 * (time.perf_counter() - t1) / steps,
            "note": "the same W warm-up + K timed steps on a solver that starts on "
                    "an idle device (the scratch solver of the device warm-up)"}
    while (time.perf_counter() - t0) * 1e3 < ms:
        if s.step() is not None:
            break
    return s, cold      # kept alive by the caller: freeing 2 GB would be an idle gap


def sustained_figure(solver, run, barrier, steps):
    """the same solver, `steps` more accepted steps in one go (seconds, not
    milliseconds, of GPU work: clocks and caches in their steady state)"""
    barrier()
    t0 = time.perf_counter()
    run(steps)
    barrier()
    dt = time.perf_counter() - t0
    return {"steps": steps, "ms_per_step": 1e3 * dt / steps,
            "value": solver.n * steps / dt}


def generic_plugin_figure(w, device, args):
    """what a plugin that exports ONLY `esq_rhs_fn` gets (first
    example): the same workload with the fused / chain entries switched off
    (ESQ_CHAIN=0), i.e. one RHS launch + one stage kernel per stage"""
    from extensisq_amd._lib import PROF_RHS, PROF_RKC, PROF_SOLERR, PROF_STAGE
    old = os.environ.get("ESQ_CHAIN")
    os.environ["ESQ_CHAIN"] = "0"
    try:
        s = w["cls"](w["rhs"], 0.0, w["y0"], 1.0e9, device=device, **w["kw"])
    finally:
        if old is None:
            del os.environ["ESQ_CHAIN"]
        else:
            os.environ["ESQ_CHAIN"] = old
    dev = s._dev
    for _ in range(args.warmup):
        assert s.step() is None
    dev.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        assert s.step() is None
    dev.synchronize()
    dt = time.perf_counter() - t0
    dev.profile_reset()
    dev.profile_enable([PROF_STAGE, PROF_RHS, PROF_SOLERR, PROF_RKC], every=1)
    for _ in range(args.steps):
        assert s.step() is None
    dev.profile_enable(None)
    tab = raw_table(dev)
    launches = sum(r["launches"] for r in tab.values())
    ms = sum(r["total_ms"] for r in tab.values())
    moved = sum(r["moved_bytes"] for r in tab.values())
    return {"ms_per_step": 1e3 * dt / args.steps, "value": s.n * args.steps / dt,
            "launches_per_step": launches / args.steps + 1,     # + the final sum
            "all_kernels_gbs": moved / (ms * 1e-3) / 1e9 if ms > 0 else None,
            "note": "ESQ_CHAIN=0: RHS plugin launch + stand-alone stage / block / "
                    "solution-error kernels, as for a user plugin without fused entry"}


def rows_kept_figure(w, device, args):
    """the same workload for a caller that reads K after every step (dense output
    at every step, `solver.K`): every row of K is written by the step that forms
    it and the end-point derivative is evaluated when the step is accepted
    (ESQ_LAZY_ROWS=0 ESQ_LAZY_END=0 -- the state a context reaches by itself after
    two such reads, esq_rk_lazy_rows).  By default the rows of a step's last chain
    sweep (read by nothing but that sweep's own solution / error sums) are written
    only on demand, and f(t_new, y_new) is evaluated as stage 0 of the next step's
    first chain sweep."""
    if "rho_jac" in w["kw"]:
        return None                       # SSV2stab config: no K rows
    knobs = ("ESQ_LAZY_ROWS", "ESQ_LAZY_END")
    old = {k: os.environ.get(k) for k in knobs}
    for k in knobs:
        os.environ[k] = "0"
    try:
        s = w["cls"](w["rhs"], 0.0, w["y0"], 1.0e9, device=device, **w["kw"])
    finally:
        for k in knobs:
            if old[k] is None:
                del os.environ[k]
            else:
                os.environ[k] = old[k]
    dev = s._dev
    for _ in range(args.warmup):
        assert s.step() is None
    dev.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        assert s.step() is None
    dev.synchronize()
    dt = time.perf_counter() - t0
    return {"ms_per_step": 1e3 * dt / args.steps, "value": s.n * args.steps / dt,
            "note": "ESQ_LAZY_ROWS=0 ESQ_LAZY_END=0: every K row written by its step, "
                    "end-point derivative evaluated at accept time"}


def adaptive_figure(w, device, esq, span_steps=40):
    """free-running step-size controller (no max_step clamp) at the bench
    tolerances over a fixed t-span of `span_steps` stability-sized steps"""
    kw = dict(w["kw"])
    h = kw.pop("max_step")
    kw["first_step"] = 0.25 * h
    if "rho_jac" in kw:
        return None                       # SSV2stab config: m is pinned by max_step
    s = w["cls"](w["rhs"], 0.0, w["y0"], span_steps * h, device=device, **kw)
    s._dev.synchronize()
    t0 = time.perf_counter()
    accepted = 0
    while s.status == "running":
        msg = s.step()
        if msg is not None and s.status != "finished":
            return {"failed": msg}
        accepted += 1
    s._dev.synchronize()
    dt = time.perf_counter() - t0
    return {"t_span_in_stability_steps": span_steps, "accepted": accepted,
            "rejected": int(esq.NFS[()]), "rhs_evaluations": s.nfev,
            "ms_per_accepted_step": 1e3 * dt / accepted if accepted else None,
            "note": "first_step = h_stab/4, controller free (max_step = inf): the "
                    "step grows to the stability limit, where rejections appear"}


def solve_ivp_figure(w, device, steps=24):
    """the drop-in call itself: `solve_ivp(rhs, (0, steps*h), y0, method=cls)`
    keeps every accepted state on the host, so each step pays an n-vector
    device-to-host copy into a fresh array (PCIe-inclusive; never the headline).
    `ms_per_step` runs from the first step to the return of the call (solver
    construction -- slab allocation, y0 upload -- is reported separately).
    `t_eval_end`: the same call with `t_eval=[t_end]` -- scipy reads `solver.y`
    after every step there too (ivp.py:665) but uses only the interpolant of the
    last one; the deferred mirror (extensisq_amd/lazy.py) then copies nothing."""
    out = _solve_ivp_run(w, device, steps, {})
    # (SSV2stab too since round 6: its cubic interpolant is device-resident as well)
    h = w["kw"]["max_step"]
    ev = _solve_ivp_run(w, device, steps, {"t_eval": [steps * h]})
    out["t_eval_end"] = {k: ev[k] for k in ("ms_per_step", "ms_per_step_mean", "steps",
                                            "value", "assembly_ms")}
    try:
        # the record of the process's download stream (csrc/esq_core.hip, lane_copy)
        from extensisq_amd._lib import copy_lane_info
        out["download_stream"] = copy_lane_info(device)
    except Exception:                                          # noqa: BLE001
        pass
    return out


def _solve_ivp_run(w, device, steps, extra):
    from scipy.integrate import solve_ivp
    h = w["kw"]["max_step"]
    stamps = []

    flags = []

    class Timed(w["cls"]):
        def _step_impl(self):
            stamps.append(time.perf_counter())
            flags.append((getattr(self, "_lazy_eager", None), len(getattr(self, "_lazy_live", []))))
            return super()._step_impl()

    t0 = time.perf_counter()
    res = solve_ivp(w["rhs"], (0.0, steps * h), w["y0"], method=Timed,
                    device=device, **extra, **w["kw"])
    t1 = time.perf_counter()
    n_steps = len(stamps)
    # entry to entry of consecutive steps: the step itself, the download of
    # solver.y and scipy's loop body
    gaps = np.diff(stamps)
    if os.environ.get("ESQ_BENCH_DEBUG"):
        print("[solve_ivp gaps ms]", extra, " ".join(f"{1e3 * g:.2f}" for g in gaps),
              file=sys.stderr)
        print("[solve_ivp lazy flags]", flags, file=sys.stderr)
    per_step = float(np.median(gaps)) if gaps.size else float("nan")
    return {"ms_per_step": 1e3 * per_step, "steps": n_steps,
            "value": w["y0"].size / per_step,
            "ms_per_step_mean": 1e3 * float(gaps.mean()) if gaps.size else None,
            "construction_ms": 1e3 * (stamps[0] - t0),
            "assembly_ms": 1e3 * (t1 - stamps[-1]),
            "note": "median time from one step's start to the next inside "
                    "solve_ivp: the HBM-resident step + scipy's loop; scipy keeps "
                    "every state, so each one is copied into a fresh host array -- "
                    "from the third step on beside the following steps (copy worker, "
                    "second stream); assembly_ms = last step + the copies still under "
                    "way + scipy's final np.vstack of all states (the vstack is "
                    "identical work in the reference)"}


def _watchdog():
    """A default run takes well under a minute.  One that is still going after
    ESQ_BENCH_STACKS_S (300 s) writes every thread's Python stack to stderr -- so
    that a hang on the driver's box leaves a trace -- and after ESQ_BENCH_LIMIT_S
    (1500 s; 0 = never) does so again and exits with code 70 instead of holding the
    GPU box until an outer limit kills it without a word."""
    import faulthandler
    stacks = float(os.environ.get("ESQ_BENCH_STACKS_S", "300"))
    limit = float(os.environ.get("ESQ_BENCH_LIMIT_S", "1500"))
    if stacks > 0:
        faulthandler.dump_traceback_later(stacks, exit=False)
    if limit > 0:
        import threading

        def bail():
            sys.stderr.write("bench.py: still running after %.0f s -- giving up\n" % limit)
            faulthandler.dump_traceback(all_threads=True)
            sys.stderr.flush()
            os._exit(70)

        t = threading.Timer(limit, bail)
        t.daemon = True
        t.start()


if __name__ == "__main__":
    _watchdog()
    main()
