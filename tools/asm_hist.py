#!/usr/bin/env python3
"""Instruction histogram of one kernel in tfhe.jl_amd/build/engine_dispatch.s (make -C tfhe.jl_amd/csrc asm).

usage: tools/asm_hist.py <demangled-name substring> [--file x.s] [--ops] [--dump A B]
Prints, for the whole kernel body and for every basic block that ends in a backward branch (a loop body as the
assembler laid it out), the number of VALU FP64 / other VALU / LDS / global / scalar / waitcnt instructions.
"""
import collections
import re
import subprocess
import sys

PATH = "tfhe.jl_amd/build/engine_dispatch.s"      # (multi-key kernels: engine_multikey.s)


def classify(op):
    if op.startswith("v_") and ("_f64" in op):
        return "valu_f64"
    if op.startswith("v_cvt"):
        return "valu_cvt"
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "accvgpr"
    if op.startswith("v_"):
        return "valu_int"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    want = sys.argv[1]
    path = sys.argv[sys.argv.index("--file") + 1] if "--file" in sys.argv else PATH
    lines = open(path).read().split("\n")
    heads = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    names = subprocess.run(["c++filt"], input="\n".join(h[1] for h in heads), capture_output=True, text=True).stdout.split("\n")
    sel = [(i, n) for (i, _), n in zip(heads, names) if want in n]
    if not sel:
        sys.exit("no kernel matches; candidates:\n" + "\n".join(sorted(set(names))))
    for start, name in sel:
        end = next(j for j in range(start, len(lines)) if lines[j].strip().startswith(".section") or lines[j].startswith("\t.amdhsa_kernel") or "s_endpgm" in lines[j])
        # the kernel may have several s_endpgm; extend to .Lfunc_end
        end = next(j for j in range(start, len(lines)) if lines[j].startswith(".Lfunc_end"))
        body = lines[start:end]
        print(f"== {name}  ({end - start} lines)")
        labels = {}
        insts = []
        for l in body:
            m = re.match(r"^(\.LBB\w+):", l)
            if m:
                labels[m.group(1)] = len(insts)
                continue
            s = l.strip()
            if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
                continue
            op = s.split()[0]
            insts.append((op, s))
        tot = collections.Counter(classify(op) for op, _ in insts)
        print("  whole kernel:", dict(tot))
        # loops: backward branches
        loops = []
        for idx, (op, s) in enumerate(insts):
            if op.startswith("s_cbranch") or op == "s_branch":
                tgt = s.split()[-1]
                if tgt in labels and labels[tgt] <= idx:
                    loops.append((labels[tgt], idx, tgt))
        for a, b, tgt in sorted(loops):
            c = collections.Counter(classify(op) for op, _ in insts[a:b + 1])
            print(f"  loop {tgt} [{a}..{b}] {b - a + 1} insts:", dict(c))
            if "--ops" in sys.argv:
                oc = collections.Counter(op for op, _ in insts[a:b + 1])
                print("     ", ", ".join(f"{k}:{v}" for k, v in oc.most_common(60)))
        if "--dump" in sys.argv:
            a, b = int(sys.argv[sys.argv.index("--dump") + 1]), int(sys.argv[sys.argv.index("--dump") + 2])
            for k in range(a, b):
                print(f"{k:6d}  {insts[k][1]}")


if __name__ == "__main__":
    main()
