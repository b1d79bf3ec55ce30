"""The boundary is a C ABI: a plain C program (tests/c_abi/abi_client.c: gcc, dlopen, no Python/C++ types)
drives tfhe_ctx_create / tfhe_load_* / tfhe_gates_batch.  CPU: it compiles against include/tfhe_mi355x.h and fails
loudly without a device.  GPU: its output equals the golden fixtures."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "abi_client.c")
EXE = os.path.join(ROOT, "tests", "c_abi", "abi_client")
GOLD = os.path.join(ROOT, "tests", "golden", "kat_n4.npz")


def _build():
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(SRC):
        subprocess.check_call(["gcc", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-ldl", "-o", EXE])
    return EXE


def _write_case(d):
    kat = np.load(GOLD)
    n, N, k, l, beta, t, g = [int(v) for v in kat["params"]]
    np.array([n, N, k, l, beta, t, g, 1], np.int32).tofile(os.path.join(d, "params.i32"))
    kat["bootstrap_key"].astype(np.int32).tofile(os.path.join(d, "bk.i32"))
    kat["keyswitch_key"].astype(np.int32).tofile(os.path.join(d, "ks.i32"))
    kat["ops"].astype(np.uint8).tofile(os.path.join(d, "ops.u8"))
    for name in ("in0", "in1", "in2"):
        kat[name].astype(np.int32).tofile(os.path.join(d, name + ".i32"))
    return kat


def test_c_client_builds_and_fails_loudly_without_gpu(tfhe, tmp_path):
    exe = _build()
    _write_case(str(tmp_path))
    if tfhe._lib.load().tfhe_device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([exe, tfhe.LIB_PATH, str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 4 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_c_client_reproduces_golden(tfhe, tmp_path):
    exe = _build()
    kat = _write_case(str(tmp_path))
    r = subprocess.run([exe, tfhe.LIB_PATH, str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == f"ok {kat['ops'].size}"
    out = np.fromfile(os.path.join(str(tmp_path), "out.i32"), np.int32).reshape(kat["out"].shape)
    assert np.array_equal(out, kat["out"])


DEMO_SRC = os.path.join(ROOT, "examples", "c", "gates_demo.c")
DEMO_EXE = os.path.join(ROOT, "tests", "c_abi", "gates_demo")


def _build_demo():
    if not os.path.exists(DEMO_EXE) or os.path.getmtime(DEMO_EXE) < os.path.getmtime(DEMO_SRC):
        subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), DEMO_SRC, "-ldl", "-lm", "-o", DEMO_EXE])
    return DEMO_EXE


def test_c_demo_builds_and_fails_loudly_without_gpu(tfhe):
    exe = _build_demo()
    if tfhe._lib.load().tfhe_device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([exe, tfhe.LIB_PATH], capture_output=True, text=True)
    assert r.returncode == 4 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_c_demo_keygen_encrypt_gates_decrypt(tfhe):
    """examples/c/gates_demo.c: secret bits -> cloud key generated on the GPU -> host encryption -> all 13 gate kinds and
    both constants on every input combination -> decryption, through the C ABI alone; every truth table holds."""
    r = subprocess.run([_build_demo(), tfhe.LIB_PATH], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert r.stdout.startswith("ok: 112 gates")
