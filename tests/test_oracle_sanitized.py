"""The CPU oracle under AddressSanitizer + UBSan (`make -C oracle sanitize`): the checker every parity claim rests on must not
itself depend on an out-of-bounds access, on signed overflow or on an invalid shift.  Runs the oracle's own tests (KATs, truth
tables at every parameter shape, FFT-vs-exact agreement, the multi-key oracle) in a child interpreter that has the sanitizer
runtimes preloaded and loads the instrumented build through TFHE_ORACLE_SO — and, in the same child, the CPU lane simulator of
the blind-rotate kernel's per-lane code (tests/host_sim: br_core.hpp's index maths with the LDS images as plain arrays, where an
out-of-range rotation index or transposition address is an ASan report instead of a GPU fault).  CPU-only; sanitizers are not
available on the GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    path = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True, check=True).stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


def test_oracle_tests_pass_under_asan_and_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not (asan and ubsan):
        pytest.skip("gcc has no sanitizer runtimes here")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "sanitize"])
    so = os.path.join(ROOT, "oracle", "_san", "libtfhe_oracle_san.so")
    env = dict(os.environ, LD_PRELOAD=f"{asan}:{ubsan}", TFHE_ORACLE_SO=so, TFHE_HOST_SIM_SANITIZE="1", OMP_NUM_THREADS="4",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    probe = subprocess.run([sys.executable, "-c", "import oracle; oracle.lib(); print([l.split()[-1] for l in open('/proc/self/maps') if 'tfhe_oracle' in l][0])"],
                           cwd=ROOT, env=env, capture_output=True, text=True)
    assert probe.returncode == 0 and probe.stdout.strip() == so, probe.stdout + probe.stderr      # the instrumented build is the one in use
    run = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "tests/test_oracle.py", "tests/test_mk.py::test_mk_oracle_nand_decrypts", "tests/test_host_sim.py", "-m", "not gpu"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, (run.stdout[-3000:] + run.stderr[-3000:])
    assert " passed" in run.stdout and "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr
