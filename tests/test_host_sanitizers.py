"""The host-only planners -- hzsdr_nco_segments (csrc/hz_host.cpp) and the persistent-pass planner
mm2::plan_call (csrc/hz_firmm2_plan.h) -- built with AddressSanitizer + UndefinedBehaviorSanitizer and
fuzzed (tests/host/plan_fuzz.cpp).  CPU only: the GPU box offers no device sanitizers, this is the
substitute for the reference's `go test -race` (Makefile:35,42) on the code that has fixed-size arrays
and hand-rolled interval arithmetic."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_planners_under_asan_ubsan():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "plan_fuzz")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                               "-I" + os.path.join(ROOT, "go-sdr_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "host", "plan_fuzz.cpp"),
                               os.path.join(ROOT, "go-sdr_amd", "csrc", "hz_host.cpp"), "-o", exe])
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
        out = subprocess.run([exe, "1500"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-4000:]
        assert "plan_fuzz ok" in out.stdout, out.stdout[-2000:]
