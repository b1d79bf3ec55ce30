"""The compiler's own report of every kernel in the shipped library (tfhe.jl_amd/build/resource_usage.txt, written by the
same hipcc invocation that links libtfhe_mi355x.so): no instantiation the dispatcher can select may spill to scratch, and
the blind-rotate kernels must keep the occupancy their LDS budgets were sized for.  CPU-only: hipcc cross-compiles here."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tfhe.jl_amd", "csrc")
REPORT = os.path.join(ROOT, "tfhe.jl_amd", "build", "resource_usage.txt")

# DIAG instantiations (rounding margin + in-kernel clock; run only under tfhe_set_option("measure_margin", 1)) that may
# spill: the diagnostics add a live double and two 64-bit stamps to a kernel that is register-bound without them.
DIAG_MAY_SPILL = {"void mk_blind_rotate_kernel_w2<4, true, 2>(MkBrArgs)", "void mk_blind_rotate_kernel_w2<4, true, 1>(MkBrArgs)",
                  "void mk_blind_rotate_kernel_g2<4, 5, true, 2, true>(MkGenArgs)", "void mk_blind_rotate_kernel_g2<8, 8, true, 2, false>(MkGenArgs)",
                  "void blind_rotate_kernel_v3<1, 8, true, true, 4>(BrArgs)"}       # (l = 1: no shipped parameter set; 3 dwords)
# Non-DIAG instantiations that keep ONE or TWO spilled dwords (an LDS address / a 64-bit key pointer reloaded once per CMUX
# step of 3 000 - 30 000 instructions): the variants of these kernels that the compiler allocates without any scratch were
# measured SLOWER (N = 2048: 46.5 ms with a scalar wave-half flag and no scratch against 44.6 ms with this one reload;
# profiles/r03/r03g_n2048_wave_modes.txt), so the faster code is shipped and its residue is pinned here: at most this many
# bytes per lane, nothing more.
SMALL_RESIDUE = {r"void blind_rotate_kernel_n2048<[34], false, [124]>\(Br2048Args\)": 12,
                 r"void blind_rotate_kernel_n2048<[234], true, [124]>\(Br2048Args\)": 24,
                 r"void mk_blind_rotate_kernel_g2<(4, 5|8, 8), false, [24], (true|false)>\(MkGenArgs\)": 24,
                 # k = 2 with l = 1: no shipped parameter set; two LDS addresses reloaded once per polynomial
                 r"void blind_rotate_kernel_k2<1, (true|false), [17]>\(BrArgs\)": 16}


def _report():
    subprocess.check_call(["make", "-s", "-C", CSRC])          # no-op when the library is newer than its sources
    txt = open(REPORT).read()
    rows = {}
    for block in re.split(r"remark: Function Name: ", txt)[1:]:
        name = block.split()[0]
        vals = {}
        for key, pat in (("vgpr", r"VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("vgpr_spill", r"VGPRs Spill: (\d+)"), ("sgpr_spill", r"SGPRs Spill: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, block)
            vals[key] = int(m.group(1)) if m else None
        rows[name] = vals
    names = list(rows)
    demangled = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
    return {d: rows[n] for n, d in zip(names, demangled)}


def test_no_selectable_kernel_spills():
    rep = _report()
    assert len(rep) > 60, "resource report looks truncated"
    # (scalar registers "spilled" into lanes of a vector register — v_writelane / v_readlane, no memory — are not counted: the
    #  N = 2048 rotation keeps 96 per-block scalars and overflows the 102 SGPRs by a few)
    def residue_ok(k, v):
        return any(re.fullmatch(pat, k) and v["scratch"] <= cap for pat, cap in SMALL_RESIDUE.items())
    bad = {k: v for k, v in rep.items() if (v["scratch"] or v["vgpr_spill"]) and k not in DIAG_MAY_SPILL and not residue_ok(k, v)}
    assert not bad, f"kernels with scratch / spills: {bad}"
    # the allow-list stays honest: an entry that no longer spills (or no longer exists) must be removed
    for k in DIAG_MAY_SPILL:
        assert k in rep and rep[k]["scratch"] > 0, f"{k} is on the allow-list but does not spill"


def test_blind_rotate_kernels_keep_two_waves_per_simd():
    rep = _report()
    two = [k for k in rep if re.search(r"blind_rotate_kernel_(v3|w2|k2|n2048)<", k)]
    assert len(two) >= 40
    for k in two:
        assert rep[k]["occ"] >= 2 and rep[k]["vgpr"] + rep[k]["agpr"] <= 256, (k, rep[k])
    # the retired instantiations stay out of the shipped library (weak #8 of the round-2 verdict)
    assert not [k for k in rep if "blind_rotate_kernel_h2<4" in k]
    assert not [k for k in rep if re.match(r"void blind_rotate_kernel<\d, 2>", k)]
