"""The oracle under AddressSanitizer + UBSan (CPU only; GPU sanitizers are not available on the pool):
the known-answer tests and the numpy cross-check run in a child interpreter against
oracle/liboracle_asan.so, so an out-of-bounds access or undefined behaviour inside the restatement —
the thing every parity claim rests on — fails the suite instead of silently shaping the expected
values."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_kats_pass_under_asan_and_ubsan():
    oracle = os.path.join(ROOT, "oracle")
    subprocess.run(["make", "-C", oracle, "liboracle_asan.so"], check=True, stdout=subprocess.PIPE,
                   stderr=subprocess.STDOUT)
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], check=True, capture_output=True,
                             text=True).stdout.strip()
    assert os.path.isabs(libasan) and os.path.exists(libasan), libasan
    env = dict(os.environ,
               LD_PRELOAD=libasan,   # the runtime must come first in a process whose main program is not instrumented
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",   # CPython itself "leaks" at exit
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               FLUID_ORACLE_LIB=os.path.join(oracle, "liboracle_asan.so"))
    res = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                          os.path.join(ROOT, "tests", "test_oracle_kat.py"),
                          os.path.join(ROOT, "tests", "test_surface_oracle_kat.py"),
                          os.path.join(ROOT, "tests", "test_golden.py")],
                         env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-4000:] + res.stderr[-4000:]
    assert "passed" in res.stdout and "AddressSanitizer" not in res.stderr
    assert "runtime error" not in res.stderr   # UBSan's report line
