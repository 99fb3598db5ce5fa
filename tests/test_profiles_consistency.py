"""The committed measurement evidence must be self-consistent (CPU check, no GPU):
for every bench config the bench line (profiles/rNN_bench_<config>.json: byte
models / HIP-event time) and the rocprofv3 summaries of the same command
(profiles/rNN_pmc_traffic_<config>.json: PMC bytes / kernel-trace time) agree:

* the bytes the counters saw lie BETWEEN the two byte models of the line -- the
  compulsory bytes (`lower_bound_bytes`: every vector read once, every output
  written once; what `roofline.achieved` / `frac` are computed from) and the
  designed L2-side bytes (halo points the marching sweeps read twice included) --
  two-sided, for every config (no special case);
* the launch durations of the two routes agree;
* the fraction is a fraction."""
import json
import os

import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(__file__), ".."))
P = os.path.join(ROOT, "profiles")
TAG = "r06"


@pytest.mark.parametrize("config", ["pr8", "ts5", "pr9", "rkc", "pr8_7070", "rkc_400", "pr8_diff3d",
                                    "pr8_diff3d_400"])
def test_bench_and_profiles_agree(config):
    with open(os.path.join(P, f"{TAG}_bench_{config}.json")) as fh:
        bench = json.load(fh)
    with open(os.path.join(P, f"{TAG}_pmc_traffic_{config}.json")) as fh:
        prof = json.load(fh)
    r = bench["roofline"]
    dom = prof["dominant_class"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert 0.0 < dom["frac_of_8TBs_pmc"] <= 1.0
    floor, moved = r["lower_bound_bytes_per_launch"], r["moved_bytes_per_launch"]
    assert floor <= moved * (1 + 1e-12)
    # counter bytes between the compulsory and the designed L2-side bytes (vectors of
    # 32-40 MB are partly served by the 32 MB of L2 across launches: down to 0.85 of
    # the floor; halo re-reads that miss L2: up to the designed bytes)
    assert 0.85 * floor <= dom["hbm_bytes_per_launch"] <= 1.10 * moved, (
        floor, dom["hbm_bytes_per_launch"], moved)
    # same box, same command, two processes (one with rocprofv3 attached)
    assert abs(r["avg_launch_us"] * 1e3 - dom["avg_launch_ns_kernel_trace"]) \
        <= 0.12 * dom["avg_launch_ns_kernel_trace"]
    # hence the two fractions bracket each other the same way
    assert r["frac"] <= 1.20 * dom["frac_of_8TBs_pmc"]
    assert dom["frac_of_8TBs_pmc"] <= 1.12 * r["l2_side_gbs"] / r["peak"]
    for name, k in prof["kernels"].items():
        if "bench_designed_bytes" in k and k["bench_designed_bytes"] > 1e6:
            assert k["hbm_bytes"] <= 1.12 * k["bench_designed_bytes"], name
            if k.get("bench_floor_bytes", 0) > 2.56e8:      # streams past every cache
                assert k["hbm_bytes"] >= 0.90 * k["bench_floor_bytes"], name
    assert bench["value"] == pytest.approx(
        bench["config"]["n_per_gpu"] * 1e3 / bench["ms_per_step"], rel=1e-9)


def test_headline_line_has_the_contract_fields():
    with open(os.path.join(P, f"{TAG}_bench_driver_command.json")) as fh:
        b = json.load(fh)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in b, key
    assert b["dtype"] == "f64" and b["data"] == "synthetic" and b["vs_baseline"] is None
    assert b["scaling"] == "weak" and b["higher_is_better"] is True
    assert "workload" in b["config"] and "model" not in b["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "lower_bound_bytes",
                "l2_side_gbs", "traffic_frac"):
        assert key in b["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in b["cpu_baseline"], key
    assert b["steps"] == 20 and b["warmup"] == 5 and b["n_gpus"] == 1
