"""The tan form of the constant twists (csrc/br_core.hpp: load_digits2t + dft8_fwd_tw, the fused rounding of untwist_add2)
against the plain form, on the host: br_core.hpp is host / device code, so a mistyped constant shows up here without a GPU
(one did: a ratio of cosines off in the 6th digit gave a 1e-6 relative error and wrong words on the device)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tan_form_equals_plain_form(tmp_path):
    exe = str(tmp_path / "twist_forms")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "tfhe.jl_amd", "csrc"),
                           "-o", exe, os.path.join(ROOT, "tests", "host", "twist_forms.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    m = re.search(r"forward_max_abs_diff (\S+) forward_max_mag (\S+) untwist_diff_words (\d+)", out)
    assert m, out
    diff, mag, words = float(m.group(1)), float(m.group(2)), int(m.group(3))
    assert mag > 1000                      # digits of up to 10 bits through one radix-8 butterfly
    assert diff < 1e-15 * 64 * mag, out    # a few ulps of the largest value: the two forms round differently, nothing more
    # pre-rounding values 0.05 from an integer at most (the device margin is 0.08 from the HALF-integer): both roundings agree
    assert words == 0, out
