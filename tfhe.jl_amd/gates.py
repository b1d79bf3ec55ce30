"""Homomorphic gates — mirrors src/gates.jl (same names, argument meaning, results).

Each call is one batch call into the HIP engine (tfhe_gates_batch): a scalar `gate_nand(ck, x, y)`
is a batch of one; LweSampleArray operands are the analogue of Julia's `gate_nand.(ck, xs, ys)`
(docs/src/manual.md:35).  There is no host fallback.
"""
import numpy as np

from ._lib import OPCODES
from .lwe import LweSample, LweSampleArray


def _as_matrix(x, n1):
    if isinstance(x, LweSample):
        return x.flat()[None, :], True
    if isinstance(x, LweSampleArray):
        return x.data, False
    if isinstance(x, (list, tuple)):
        return np.stack([s.flat() for s in x]), False
    a = np.asarray(x, np.int32)
    return (a[None, :], True) if a.ndim == 1 else (a, False)


def _run(ck, op, *operands, device=0):
    eng = ck.engine(device)
    n1 = eng.n + 1
    mats, scalar = [], True
    for x in operands:
        m, s = _as_matrix(x, n1)
        mats.append(m)
        scalar = scalar and s
    B = max(m.shape[0] for m in mats)
    mats = [np.broadcast_to(m, (B, n1)) if m.shape[0] != B else m for m in mats]
    ops = np.full(B, OPCODES[op], np.uint8)
    out = eng.gates(ops, *mats)
    if scalar:
        return LweSample.from_flat(out[0])
    return LweSampleArray(out)


def gate_nand(ck, x, y, device=0):
    """gates.jl:15-18"""
    return _run(ck, "NAND", x, y, device=device)


def gate_or(ck, x, y, device=0):
    """gates.jl:27-30"""
    return _run(ck, "OR", x, y, device=device)


def gate_and(ck, x, y, device=0):
    """gates.jl:39-42"""
    return _run(ck, "AND", x, y, device=device)


def gate_xor(ck, x, y, device=0):
    """gates.jl:51-54"""
    return _run(ck, "XOR", x, y, device=device)


def gate_xnor(ck, x, y, device=0):
    """gates.jl:63-66"""
    return _run(ck, "XNOR", x, y, device=device)


def gate_not(ck, x, device=0):
    """gates.jl:76-79 (no bootstrap)"""
    return _run(ck, "NOT", x, device=device)


def gate_constant(ck, value, device=0):
    """gates.jl:91-93 — a trivial (unencrypted) sample of `value`."""
    eng = ck.engine(device)
    out = eng.gates(np.array([OPCODES["CONST1" if value else "CONST0"]], np.uint8), None)
    return LweSample.from_flat(out[0])


def gate_nor(ck, x, y, device=0):
    """gates.jl:102-105"""
    return _run(ck, "NOR", x, y, device=device)


def gate_andny(ck, x, y, device=0):
    """gates.jl:114-117"""
    return _run(ck, "ANDNY", x, y, device=device)


def gate_andyn(ck, x, y, device=0):
    """gates.jl:126-129"""
    return _run(ck, "ANDYN", x, y, device=device)


def gate_orny(ck, x, y, device=0):
    """gates.jl:138-141"""
    return _run(ck, "ORNY", x, y, device=device)


def gate_oryn(ck, x, y, device=0):
    """gates.jl:150-153"""
    return _run(ck, "ORYN", x, y, device=device)


def gate_mux(ck, x, y, z, device=0):
    """gates.jl:163-177"""
    return _run(ck, "MUX", x, y, z, device=device)


def gates_batch(ck, opcodes, in0, in1=None, in2=None, device=0):
    """Mixed stream of independent gates (BASELINE config 3): opcodes are names or numbers."""
    ops = np.array([OPCODES[o] if isinstance(o, str) else int(o) for o in opcodes], np.uint8)
    eng = ck.engine(device)
    mats = [None if x is None else _as_matrix(x, eng.n + 1)[0] for x in (in0, in1, in2)]
    return LweSampleArray(eng.gates(ops, *mats))
