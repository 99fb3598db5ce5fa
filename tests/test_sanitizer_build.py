"""The host side of the library under AddressSanitizer + UBSan (`make asan`:
csrc/Makefile -- the host pass only, no device code, so it builds in half a minute
and runs here without a GPU).  The planner, the step programs, the launch ahead of
time with its saved / restored state and the ABI table are host C++ that the CPU
suite already drives through `esq_plan_describe` / `esq_step_dry_run`: the same
tests, in a python that has clang's ASan runtime preloaded and loads the sanitized
build through ESQ_LIB.  Any report of either sanitizer aborts that python."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(__file__), ".."))
CSRC = os.path.join(ROOT, "extensisq_amd", "csrc")
LIB = os.path.join(ROOT, "extensisq_amd", "libextensisq_amd_asan.so")


def _asan_runtime():
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return hits[-1] if hits else None


@pytest.mark.skipif(os.environ.get("ESQ_LIB") is not None, reason="already inside a variant run")
def test_host_side_tests_pass_under_asan_and_ubsan():
    rt = _asan_runtime()
    if rt is None:
        pytest.skip("clang's shared ASan runtime is not installed")
    res = subprocess.run(["make", "-j8", "-C", CSRC, "asan"], capture_output=True, text=True,
                         timeout=1200)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    env = dict(os.environ)
    env.update(LD_PRELOAD=rt, ESQ_LIB=LIB,
               # (python itself leaks by design; everything else is fatal)
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    tests = [os.path.join(ROOT, "tests", t) for t in
             ("test_step_plans.py", "test_abi_symbols.py", "test_host_logic.py")]
    res = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu",
                          "-p", "no:cacheprovider"] + tests,
                         capture_output=True, text=True, timeout=1800, cwd=ROOT, env=env)
    tail = res.stdout[-4000:] + res.stderr[-4000:]
    assert res.returncode == 0, tail
    assert "passed" in res.stdout and "AddressSanitizer" not in tail and \
        "runtime error" not in tail, tail
