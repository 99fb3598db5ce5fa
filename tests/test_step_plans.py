"""The step as a program (csrc/esq_step.hip: build_plan / run_plan), checked on the
CPU: `esq_plan_describe` builds the launch plans of a method on a built-in plugin
exactly as esq_rk_stages does on a device -- the plugins answer the library's
side-effect-free queries on the host -- on a detached context that touches no GPU.

* every plan of the 9 tableaux x plugins x chain capabilities 0...15 x lazy rows on /
  off is pinned (launch sequence and designed words per element) in
  tests/golden/step_plans.json (tools/gen_step_plans.py);
* structural invariants hold for every plan: each stage is evaluated exactly once
  and in order, a capability that was not declared is never used, a step that
  leaves K[0] to the next one starts with it;
* the plans of the BASELINE.json configurations are spelled out here.
"""
import json
import os
import re
import sys

import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_step_plans as gsp  # noqa: E402

CAP_FROM_STATE, CAP_SKIP_ROWS, CAP_FROM_ROWS, CAP_SKIP_OUT = 1, 2, 4, 8
CAP_PRE, CAP_ERRNORM = gsp.CAP_PRE, gsp.CAP_ERRNORM


@pytest.fixture(scope="module")
def plans():
    return gsp.table()


def test_plans_match_the_pinned_table(plans, golden_dir):
    with open(os.path.join(golden_dir, "step_plans.json")) as fh:
        want = json.load(fh)
    assert sorted(plans) == sorted(want)
    bad = [k for k in plans if plans[k] != want[k]]
    assert not bad, (bad[:5], plans[bad[0]], want[bad[0]])


TOKEN = re.compile(r"(\w+)\[(\d+)(?:,(\d+),(\d+))?\]([LFS]*)")


def _parse(line):
    label, rest = line.split(":", 1)
    body, tail = rest.split("|")
    steps = [(m.group(1), int(m.group(2)), int(m.group(3) or 0), int(m.group(4) or 0),
              m.group(5)) for m in TOKEN.finditer(body)]
    launches = int(re.search(r"launches=(\d+)", tail).group(1))
    return label.strip(), steps, launches, tail


def test_every_plan_evaluates_each_stage_once_and_uses_only_declared_forms(plans):
    import extensisq_amd as esq
    stages = {m: getattr(esq, m).n_stages for m in gsp.METHODS}
    stages["Heun"] = 2
    for key, lines in plans.items():
        name, plugin, caps, lazy = key.split("/")[:4]
        pre = key.endswith("/pre")        # the whole step of a pair with an early estimate
        caps, lazy, s = int(caps[4:]), int(lazy[4:]), stages[name]
        cls = getattr(esq, name, None)
        P = len(gsp.early_estimate(cls)[0]) if pre else 0
        for line in lines:
            label, steps, launches, tail = _parse(line)
            assert launches == len(steps)
            done = []                     # stages whose derivative this plan evaluates
            estimates = []                # ... and after which stage the estimate ran
            have_arg = label == "prelaunched"
            for op, i, depth, what, flags in steps:
                if op == "chain":
                    # (what 4: the chain's last stage is the end-point evaluation)
                    done += list(range(i, i + depth - (1 if what == 4 else 0)))
                    if i == 0:
                        assert label == "deferred" and caps & CAP_FROM_STATE, (key, line)
                    if "L" in flags:
                        assert lazy and caps & CAP_SKIP_ROWS, (key, line)
                    if "F" in flags:
                        assert caps & CAP_FROM_ROWS, (key, line)
                    if "S" in flags:
                        assert caps & CAP_SKIP_OUT and caps & CAP_FROM_ROWS, (key, line)
                    # an argument must exist unless the chain makes it itself
                    assert have_arg or "F" in flags or i == 0, (key, line)
                    have_arg = what == 0 and "S" not in flags
                    if what == 3:
                        # the estimate rides on the chain that ends right before stage P;
                        # its y_pre is the next argument only for CFMR7osc (calvo.py:257)
                        assert pre and i + depth == P, (key, line)
                        estimates.append(done[-1])
                        have_arg = name == "CFMR7osc" and "S" not in flags
                        if name == "BS5":
                            assert caps & CAP_PRE, (key, line)
                    if what == 4:
                        assert caps & CAP_ERRNORM and i + depth == s + 1, (key, line)
                        assert " ynew solerr" in tail, (key, line)
                elif op == "pre":
                    assert pre and i == P, (key, line)
                    estimates.append(done[-1])
                elif op == "k0":
                    assert label == "deferred" and i == 0
                    done.append(0)
                elif op in ("accum", "lincomb"):
                    assert not have_arg, (key, line)
                    have_arg = True
                elif op == "src":
                    assert i == 1 and not have_arg
                    done.append(1)
                    have_arg = True
                else:                     # stage / block / ynew / solerr / rhs
                    assert have_arg, (key, line)
                    done.append(i)
                    have_arg = op == "stage" or (op == "block" and False)
                    if op == "block":     # the block sweep may or may not form the argument
                        have_arg = None
                if have_arg is None:      # resolved by what follows
                    nxt = steps[steps.index((op, i, depth, what, flags)) + 1][0]
                    have_arg = nxt not in ("lincomb", "accum")
            first = 0 if label == "deferred" else 1
            assert done == list(range(first, s)), (key, line, done)
            # exactly one early estimate, right behind stage P - 1
            assert estimates == ([P - 1] if pre else []), (key, line, estimates)
            # the last launch of a whole step forms y_new where the plugin fuses
            if plugin.startswith(("bruss2d", "heat2d", "diff3d")):
                assert " ynew" in tail, (key, line)


def test_baseline_configuration_plans(plans):
    """the launch sequences bench.py times (BASELINE.json configs 2, 3, 5), spelled out"""
    pr8 = plans["Pr8/bruss2d2236/caps15/lazy1"]
    assert pr8[1].startswith("deferred: chain[0,5,0]L chain[5,4,0]LS chain[9,4,2]LF "
                             "| launches=3 words=16+10 ")
    assert pr8[0].startswith("first: chain[1,5,0]LFS chain[6,3,0]LFS chain[9,4,2]LF ")
    ts5 = plans["Ts5/heat2d1000/caps15/lazy1"]
    assert ts5[0].startswith("first: chain[1,5,1]LF | launches=1 words=2+6 ")
    pr9 = plans["Pr9/heat2d2236/caps15/lazy1"]
    assert pr9[1].startswith("deferred: chain[0,5,0]L chain[5,3,0]L block[8] chain[9,3,0]L "
                             "chain[12,4,0]L solerr[16] | launches=6 ")
    # Pr7: the cheapest sequence is not the longest chains first (30 -> 15 words)
    pr7 = plans["Pr7/bruss2d2236/caps15/lazy1"]
    assert pr7[1].startswith("deferred: chain[0,5,0]L rhs[5] chain[6,4,2]LF "
                             "| launches=3 words=8+7 ")
    # a plugin with esq_rhs_fn only: one library kernel + one RHS launch per stage
    plain = plans["Pr8/plain100/caps0/lazy1"]
    assert plain[0].count("accum[") == 12 and plain[0].count("rhs[") == 12
    # the 3-D plugin (csrc/esq_chain3d.hpp): chains of three and four stages, the end of
    # the step inside the last one; 38 words per element and step (13 fused sweeps: 104)
    d3 = plans["Pr8/diff3d159/caps3/lazy1"]
    assert d3[1].startswith("deferred: chain[0,3,0]L chain[3,3,0]L chain[6,4,0]L chain[10,3,2]L "
                            "| launches=4 words=24+14 ")


def test_more_capabilities_never_cost_more(plans):
    """the optional chain forms only ever widen the planner's choice: the plan with
    every capability costs (in the planner's own units: words x halo factors +
    kernel boundaries -- what it minimises) no more than any plan with a subset"""
    def cost(line):
        return float(re.search(r"cost=([\d.]+)", line).group(1))
    for key, lines in plans.items():
        if "/caps15/" not in key or key.endswith("/pre"):
            continue
        for caps in range(15):
            other = plans[key.replace("/caps15/", f"/caps{caps}/")]
            for a, b in zip(lines, other):
                if a.split(":")[0] == b.split(":")[0]:
                    assert cost(a) <= cost(b) + 0.011, (key, caps, a, b)


def test_round6_chain_forms_only_ever_save_launches_and_words(plans):
    """a chain through the end of an FSAL step / an early estimate on a chain sweep: the
    plans with the new capabilities have no more launches and move no more words than
    the plans without them (the planner's cost of an FSAL plan does not include the
    end-point sweep that follows it, so costs are compared through launches and words:
    the plan through the end of the step has the end-point sweep's 1 launch and
    ne + 2 + 1 words in hand)"""
    import extensisq_amd as esq

    def words(line):
        m = re.search(r"launches=(\d+) words=([\d.]+)\+([\d.]+)", line)
        return int(m.group(1)), float(m.group(2)) + float(m.group(3))
    seen = 0
    for key, lines in plans.items():
        parts = key.split("/")
        if parts[2] != f"caps{15 | CAP_PRE | CAP_ERRNORM}":
            continue
        base = plans[key.replace(parts[2], "caps15")]
        cls = getattr(esq, parts[0], None)
        for a, b in zip(lines, base):
            assert a.split(":")[0] == b.split(":")[0]
            (la, wa), (lb, wb) = words(a), words(b)
            if " solerr" in a and " solerr" not in b:      # (the end-point sweep: above)
                ne = int((cls.E[:cls.n_stages] != 0).sum())
                lb, wb = lb + 1, wb + ne + 3
            assert la <= lb and wa <= wb + 1, (key, a, b)
            seen += 1
    assert seen > 60


def _rkc_plan(m, depth, forms=0, end_slots=5):
    import ctypes as C
    from extensisq_amd import _lib
    lib = _lib.load()
    buf = C.create_string_buffer(1 << 14)
    assert lib.esq_rkc_plan_describe(m, depth | forms, end_slots, buf, len(buf)) == 0
    seq, tail = buf.value.decode().split(" | ")
    launches = seq.split()
    assert tail == f"launches={len(launches)}"
    return launches


def test_chebyshev_step_programs():
    """the launch sequence of an SSV2stab step (esq_rkc_stages_end) for a chain entry
    of a given depth and forms, without a GPU: every stage evaluated exactly once, no
    single stage left at the end of a chained step, the first iterate and the end of
    the step inside the first / last chain where the forms allow"""
    from extensisq_amd._lib import RKC_CHAIN_FIRST as F, RKC_CHAIN_LAST as L

    def stages(launches):
        n = 0
        for k in launches:
            if k.startswith("rkc_chain"):
                n += int(k[len("rkc_chain")])
            elif k == "rhs_rkc":
                n += 1
        return n

    # BASELINE config 3: m = 100 on the 3-D plugin, four stages per sweep
    p = _rkc_plan(100, 4, F | L, 5)
    assert p == ["rkc_chain4-first"] + ["rkc_chain4"] * 23 + ["rkc_chain3-end"]
    # beyond the Infinity Cache: five per sweep, 99 = 19 x 5 + 4
    p = _rkc_plan(100, 5, F | L, 5)
    assert p == ["rkc_chain5-first"] + ["rkc_chain5"] * 18 + ["rkc_chain4-end"]
    # the 2-D plugin at N >= 1500: six per sweep, the end rides with up to five stages
    p = _rkc_plan(100, 6, F | L, 6)
    assert p == ["rkc_chain6-first"] + ["rkc_chain6"] * 15 + ["rkc_chain3-end"]
    # round 3 / a plugin without a chain entry: one launch per stage
    p = _rkc_plan(100, 1)
    assert p == ["k_rkc_first"] + ["rhs_rkc"] * 99 + ["rhs+rkcerr"]
    # no forms declared: first iterate and end of the step are sweeps of their own
    p = _rkc_plan(100, 4)
    assert p == ["k_rkc_first"] + ["rkc_chain4"] * 24 + ["rkc_chain3-last", "rhs+rkcerr"]
    # short steps: 5 = 3 + 2; one chain from start to end (FIRST wins); one stage
    assert _rkc_plan(6, 4, F | L) == ["rkc_chain3-first", "rkc_chain2-end"]
    assert _rkc_plan(4, 4, F | L) == ["rkc_chain3-first", "rhs+rkcerr"]
    assert _rkc_plan(3, 4, F | L) == ["rkc_chain2-first", "rhs+rkcerr"]
    assert _rkc_plan(2, 4, F | L) == ["k_rkc_first", "rhs_rkc", "rhs+rkcerr"]
    assert _rkc_plan(1, 4, F | L) == ["k_rkc_first", "rhs+rkcerr"]
    # an end that does not fit the entry's stage slots stays a sweep of its own
    assert _rkc_plan(9, 4, F | L, 4)[-2:] == ["rkc_chain4-last", "rhs+rkcerr"]
    for depth in range(1, 9):
        for forms in (0, F, L, F | L):
            for slots in (4, 5, 6, 9):
                for m in list(range(1, 40)) + [100, 132, 250]:
                    p = _rkc_plan(m, depth, forms, slots)
                    assert stages(p) == m - 1, (m, depth, forms, p)
                    assert (p.count("k_rkc_first") + sum(k.endswith("-first") for k in p)) == 1
                    assert (p.count("rhs+rkcerr") + sum(k.endswith("-end") for k in p)) == 1
                    if depth >= 3 and m - 1 >= 2:
                        assert "rhs_rkc" not in p, (m, depth, p)
                    elif depth == 2:                  # (an odd count leaves one)
                        assert p.count("rhs_rkc") == (m - 1) % 2, (m, p)


# ---------------------------------------------------------------------------
# whole steps on a detached context (esq_step_dry_run): the host side of the step --
# row maps, the first launch ahead of time with what it saves and restores, rows left
# unwritten, the deferred end-point derivative -- with no GPU behind it
# ---------------------------------------------------------------------------
SCRIPT = [0, 0, 1, 2, 1, 3, 1, 4, 0, 5, 0, 0]
LINE = re.compile(r"(\d+): state_ok=(\d) used=(\d+) dropped=(\d+) missing=(\d+) k0=(\d) "
                  r"fused=(\d+) plain=(\d+)")


PRE_TAIL = re.compile(r" pre=(\d+)/(\d+) fused=(\d+) plain=(\d+)$")


def _dry(cls, plugin, N, script=SCRIPT, caps=15 | gsp.CAP_QUERY, lazy=1, depth=4, src=0,
         pre=None):
    """rows (code, state_ok, used, dropped, missing, k0, end_fused, end_plain); with an
    early estimate registered four more columns: the attempt's estimate number, the
    estimates published so far, how many of them rode on a chain sweep / ran as a pass
    of their own"""
    import ctypes as C
    import numpy as np
    from extensisq_amd import _lib
    lib = _lib.load()
    s = cls.n_stages
    arrs = [np.ascontiguousarray(getattr(cls, k), dtype=float) for k in "ABCE"]
    fsal = int(arrs[3][s] != 0)
    sc = np.asarray(script, dtype=np.int32)
    buf = C.create_string_buffer(1 << 15)
    fuse = gsp.FUSE_ALL | gsp.FUSE_SRC | gsp.FUSE_QUERY
    e, b = ([np.ascontiguousarray(x, dtype=float) for x in pre] if pre is not None
            else (np.zeros(1), np.zeros(1)))
    r = lib.esq_step_dry_run(plugin.encode(), N, s, *[_lib.as_ptr(a) for a in arrs], fsal, caps,
                             fuse, lazy, depth, src, _lib.as_ptr(e), _lib.as_ptr(b),
                             len(e) if pre is not None else 0, sc.ctypes.data_as(C.c_void_p),
                             len(sc), buf, len(buf))
    assert r == 0, r
    rows = []
    for ln in buf.value.decode().strip().split("\n"):
        row = tuple(int(g) for g in LINE.match(ln).groups())
        if pre is not None:
            row += tuple(int(g) for g in PRE_TAIL.search(ln).groups())
        rows.append(row)
    assert [row[0] for row in rows] == list(script)
    return rows


@pytest.mark.parametrize("plugin,N", gsp.PLUGINS)
def test_launch_ahead_leaves_the_step_state_as_it_found_it(plugin, N):
    """every tableau x plugin x capability set: whatever launch_ahead runs on the next
    step's behalf, the context's StepState and row maps are afterwards what they were,
    and the row maps stay permutations disjoint from the spare rows -- through accepted
    and rejected attempts, confirmed and wrong guesses of the next step size, a reader
    in between"""
    import extensisq_amd as esq
    classes = [getattr(esq, m) for m in gsp.METHODS] + [gsp.heun()]
    has_chain = plugin in ("bruss2d", "heat2d", "diff3d")
    for cls in classes:
        for caps in (list(range(16)) + gsp.CAPS_R6 if plugin != "diff3d" and has_chain else
                     [0, 1, 2, 3] if has_chain else [0]):
            for lazy in (0, 1):
                rows = _dry(cls, plugin, N, caps=caps | gsp.CAP_QUERY, lazy=lazy)
                assert all(row[1] == 1 for row in rows), (cls.__name__, caps, lazy, rows)
                used, dropped = [row[2] for row in rows], [row[3] for row in rows]
                assert used == sorted(used) and dropped == sorted(dropped)
                # one launch ahead per attempt at most, each used or dropped once
                assert used[-1] + dropped[-1] <= len(SCRIPT)
                if not has_chain:
                    assert used[-1] == 0 or cls.__name__ in ("Ts5", "BS5", "Heun"), rows


def test_launch_ahead_script_of_the_metric_configuration():
    """Pr8 on the Brusselator plugin, N = 2236, all capabilities: which attempts find
    their first sweep in the queue already, which launches are dropped, what stays
    unwritten, how the end-point derivatives run"""
    import extensisq_amd as esq
    rows = _dry(esq.Pr8, "bruss2d", 2236)
    #        attempt:  0  0  1  2  1  3  1  4  0  5  0  0
    assert [r[2] for r in rows] == [0, 1, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8]      # used
    assert [r[3] for r in rows] == [0, 0, 0, 0, 0, 0, 1, 2, 2, 3, 3, 3]      # dropped
    # K_1 and K_9 .. K_12 unwritten after every attempt but the one a reader followed
    assert [r[4] for r in rows] == [5, 5, 5, 5, 5, 5, 5, 0, 5, 5, 5, 5]
    # every end-point derivative inside a chain sweep (none after the rejected attempt)
    assert [r[6] for r in rows] == [1, 2, 3, 3, 4, 5, 6, 7, 8, 9, 10, 11]
    assert all(r[7] == 0 and r[5] == 0 for r in rows)
    # an FSAL pair: nothing is missing, a reader drops nothing
    rows = _dry(esq.Ts5, "heat2d", 1000, src=1)
    assert [r[2] for r in rows] == [0, 1, 2, 3, 3, 4, 4, 5, 6, 7, 8, 9]
    assert [r[3] for r in rows] == [0, 0, 0, 0, 0, 0, 1, 1, 1, 2, 2, 2]
    assert all(r[4] == 0 and r[5] == 0 and r[6] == 0 for r in rows)


@pytest.mark.parametrize("plugin,N", gsp.PLUGINS)
def test_whole_steps_with_an_early_estimate_on_a_detached_context(plugin, N):
    """BS5 / CFMR7osc with their early estimate registered (esq_rk_set_pre), every
    capability set incl. round 6's: every attempt -- accepted, rejected, launched ahead,
    guessed wrong, with a reader in between -- carries exactly ONE estimate of its own
    (a number no other attempt has), the speculation leaves the context as it found it,
    and with the chain forms declared no estimate runs as a pass of its own"""
    import extensisq_amd as esq
    has_chain = plugin in ("bruss2d", "heat2d", "diff3d")
    all_caps = ([0, 3, 15] + gsp.CAPS_R6 if plugin in ("bruss2d", "heat2d")
                else [0, 3] if plugin == "diff3d" else [0])
    for cls in (esq.BS5, esq.CFMR7osc):
        pre = gsp.early_estimate(cls)
        for caps in all_caps:
            for lazy in (0, 1):
                rows = _dry(cls, plugin, N, caps=caps | gsp.CAP_QUERY, lazy=lazy, pre=pre)
                what = (cls.__name__, plugin, caps, lazy, rows)
                assert all(row[1] == 1 for row in rows), what
                mine = [row[8] for row in rows]
                assert all(m > 0 for m in mine) and len(set(mine)) == len(mine), what
                assert mine == sorted(mine), what
                # (published: this attempt's, the dry run's own probe of launch_ahead,
                # at most two launches ahead)
                assert all(row[8] <= row[9] <= row[8] + 3 for row in rows), what
                fused, plain = rows[-1][10], rows[-1][11]
                assert fused + plain == rows[-1][9], what
                full = caps & 15 == 15 and has_chain
                if full and (cls is esq.CFMR7osc or caps & CAP_PRE):
                    assert plain == 0, what
                if not has_chain:
                    assert fused == 0, what


def test_early_estimate_plans_of_the_method_sweep(plans):
    """the whole-step programs tools/method_sweep.py times at N = 2236, spelled out"""
    caps = 15 | CAP_PRE | CAP_ERRNORM
    bs5 = plans[f"BS5/bruss2d2236/caps{caps}/lazy1/pre"]
    # stages 1-5 from K[0] with the estimate as last target (y_pre not stored), then
    # stage 6 from the rows, y_new, the FSAL stage and the error norm: 2 launches
    assert bs5[0].startswith("first: chain[1,5,3]LF chain[6,2,4]LF | launches=2 words=9+7 ")
    cf = plans[f"CFMR7osc/bruss2d2236/caps{caps}/lazy1/pre"]
    # y_pre of CFMR7osc IS stage 8's argument: stored, the last stage's sweep reads it
    assert cf[1].startswith("deferred: chain[0,4,0]LS chain[4,4,3]LF solerr[8] "
                            "| launches=3 words=13+10 ")
    # without the round-6 forms BS5's estimate is a pass of its own
    old = plans["BS5/bruss2d2236/caps15/lazy1/pre"]
    assert " pre[6] " in old[0]
    # config 2: the whole Ts5 step is one sweep, y and K[0] in, y_new and K[6] out
    ts5 = plans[f"Ts5/heat2d1000/caps{caps}/lazy1"]
    assert ts5[0].startswith("first: chain[1,6,4]LF | launches=1 words=2+2 ")
