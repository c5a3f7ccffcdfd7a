"""CPU: the host side of the library under ThreadSanitizer.

tests/native/tsan_driver.cpp is compiled together with the library's own sources with
`-Xarch_host -fsanitize=thread` (host code only; the device code is compiled as usual and never runs)
and spawns four threads that set different tsp2opt_debug_* knobs and call the planners, the argument
checks and the error-string getter concurrently.  The knobs are per calling thread, so every thread must
keep seeing exactly its own settings, and TSan must report no race."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CSRC = os.path.join(ROOT, "ml4tspbench_amd", "csrc")
EXE = os.path.join(ROOT, "build", "tsan_driver")


def _build():
    srcs = [os.path.join(ROOT, "tests", "native", "tsan_driver.cpp"),
            os.path.join(CSRC, "tsp2opt.hip"), os.path.join(CSRC, "greedy.hip")]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "tsp2opt.h")]
    if os.path.exists(EXE) and os.path.getmtime(EXE) >= max(os.path.getmtime(f) for f in deps):
        return
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    subprocess.check_call([HIPCC, "-O1", "-g", "-std=c++17", "--offload-arch=gfx950",
                           "-Xarch_host", "-fsanitize=thread", "-ffp-contract=off", "-fno-fast-math",
                           "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
                           "-o", EXE, *srcs])


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_host_side_is_race_free_under_tsan():
    _build()
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66 report_signal_unsafe=0")
    r = subprocess.run([EXE], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert "ThreadSanitizer" not in r.stdout, r.stdout[-4000:]
    assert r.returncode == 0 and "0 failures" in r.stdout, r.stdout[-2000:]
