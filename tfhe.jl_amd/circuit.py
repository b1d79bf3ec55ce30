"""Levelised execution of gate circuits on device-resident ciphertexts.

Real workloads are gate DAGs (examples/tutorial.jl:42-62: a 16-deep XNOR->MUX chain, then 16 parallel
MUXes).  A Circuit records gates with the reference's gate names, assigns every gate the level
1 + max(level of its operands), and runs level by level: one tfhe_gates_level call per level over all the
level's independent gates, ciphertexts staying in the engine's wire table on the GPU — only the inputs go
up and only the requested outputs come down."""
import numpy as np

from ._lib import OPCODES
from .lwe import LweSample, LweSampleArray

_ARITY = {"NOT": 1, "COPY": 1, "CONST0": 0, "CONST1": 0, "MUX": 3}


class Circuit:
    def __init__(self):
        self._n_inputs = 0
        self._gates = []          # (opcode name, a, b, c) with wire ids
        self._level = []          # level per wire (inputs: 0)
        self._outputs = []

    # ---- building -----------------------------------------------------------------------------------
    def input(self):
        if self._gates:
            raise ValueError("declare all inputs before the first gate")
        self._n_inputs += 1
        self._level.append(0)
        return self._n_inputs - 1

    def inputs(self, count):
        return [self.input() for _ in range(count)]

    def gate(self, name, *operands):
        name = name.upper()
        if name not in OPCODES:
            raise ValueError(f"unknown gate {name}")
        arity = _ARITY.get(name, 2)
        if len(operands) != arity:
            raise ValueError(f"gate {name} takes {arity} operand(s)")
        for w in operands:
            if not (0 <= w < len(self._level)):
                raise ValueError(f"operand wire {w} does not exist")
        ops = list(operands) + [-1] * (3 - arity)
        self._gates.append((name, ops[0], ops[1], ops[2]))
        self._level.append(1 + max([self._level[w] for w in operands], default=0))
        return len(self._level) - 1

    # the reference's names (src/gates.jl)
    def nand(self, x, y): return self.gate("NAND", x, y)
    def or_(self, x, y): return self.gate("OR", x, y)
    def and_(self, x, y): return self.gate("AND", x, y)
    def xor(self, x, y): return self.gate("XOR", x, y)
    def xnor(self, x, y): return self.gate("XNOR", x, y)
    def not_(self, x): return self.gate("NOT", x)
    def nor(self, x, y): return self.gate("NOR", x, y)
    def andny(self, x, y): return self.gate("ANDNY", x, y)
    def andyn(self, x, y): return self.gate("ANDYN", x, y)
    def orny(self, x, y): return self.gate("ORNY", x, y)
    def oryn(self, x, y): return self.gate("ORYN", x, y)
    def mux(self, x, y, z): return self.gate("MUX", x, y, z)
    def constant(self, value): return self.gate("CONST1" if value else "CONST0")

    def set_outputs(self, wires):
        self._outputs = list(wires)

    # ---- analysis -----------------------------------------------------------------------------------
    @property
    def num_wires(self):
        return len(self._level)

    def levels(self):
        """List of levels; each level is a list of gate indices (gate g drives wire n_inputs + g)."""
        depth = max(self._level, default=0)
        out = [[] for _ in range(depth)]
        for g in range(len(self._gates)):
            out[self._level[self._n_inputs + g] - 1].append(g)
        return out

    def level_arrays(self):
        """Per level: (opcodes u8, a, b, c, out) index arrays as tfhe_gates_level takes them."""
        res = []
        for gates in self.levels():
            ops = np.array([OPCODES[self._gates[g][0]] for g in gates], np.uint8)
            a = np.array([max(self._gates[g][1], 0) for g in gates], np.int32)
            b = np.array([max(self._gates[g][2], 0) for g in gates], np.int32)
            c = np.array([max(self._gates[g][3], 0) for g in gates], np.int32)
            out = np.array([self._n_inputs + g for g in gates], np.int32)
            res.append((ops, a, b, c, out))
        return res

    # ---- execution ----------------------------------------------------------------------------------
    def run(self, ck, inputs, device=0):
        """inputs: LweSampleArray / list of LweSample / int32 [n_inputs][n+1].  Returns an LweSampleArray
        of the output wires.  Everything between upload and download runs on the GPU."""
        eng = ck.engine(device)
        if isinstance(inputs, LweSampleArray):
            m = inputs.data
        elif isinstance(inputs, (list, tuple)):
            m = np.stack([s.flat() if isinstance(s, LweSample) else np.asarray(s, np.int32) for s in inputs])
        else:
            m = np.asarray(inputs, np.int32)
        if m.shape[0] != self._n_inputs:
            raise ValueError(f"circuit has {self._n_inputs} inputs, got {m.shape[0]}")
        eng.wires_alloc(self.num_wires)
        if self._n_inputs:
            eng.wires_upload(0, m)
        # no per-phase timing events while the levels run: each record keeps the stream's next kernel waiting ~5 us, and a level
        # of a narrow circuit is six short operations around one single-rotation kernel (tutorial circuit: 30.5 -> 30.1 ms)
        timing_before = eng.get_option("timing_events")              # (the engine is shared through ck.engine(): put the caller's setting back)
        eng.set_option("timing_events", 0)
        try:
            for ops, a, b, c, out in self.level_arrays():
                eng.gates_level(ops, a, b, c, out)
            return LweSampleArray(eng.wires_gather(self._outputs))     # one device gather + one copy for all outputs
        finally:
            eng.set_option("timing_events", timing_before)

    def run_batch(self, ck, inputs, device=0):
        """The same circuit on M independent input sets at once: inputs int32 [M][n_inputs][n+1] (or a list of M LweSampleArrays),
        result int32 [M][n_outputs][n+1].  Every level becomes ONE tfhe_gates_level call over the M instances' gates — a level of
        a narrow circuit costs one single-rotation latency whether it holds 1 gate or 256 (one blind rotation per CU), so M <= 256 /
        (rotations of the widest level) instances cost what one does; beyond that the levels run at batch throughput.
        Wire w of instance i is row w * M + i of the wire table: the inputs go up as one block, the outputs come down as one gather."""
        eng = ck.engine(device)
        if isinstance(inputs, (list, tuple)):
            inputs = np.stack([x.data if isinstance(x, LweSampleArray) else np.asarray(x, np.int32) for x in inputs])
        m = np.ascontiguousarray(inputs, dtype=np.int32)
        if m.ndim != 3 or m.shape[1] != self._n_inputs:
            raise ValueError(f"inputs must be [M][{self._n_inputs}][n+1], got {m.shape}")
        M = m.shape[0]
        if M == 0:
            return np.zeros((0, len(self._outputs), m.shape[2]), np.int32)
        eng.wires_alloc(self.num_wires * M)
        if self._n_inputs:
            eng.wires_upload(0, np.ascontiguousarray(m.transpose(1, 0, 2)).reshape(self._n_inputs * M, -1))
        inst = np.arange(M, dtype=np.int32)
        spread = lambda w: (w[:, None] * M + inst[None, :]).reshape(-1).astype(np.int32)      # wire ids -> rows, instance fastest
        timing_before = eng.get_option("timing_events")
        eng.set_option("timing_events", 0)
        try:
            for ops, a, b, c, out in self.level_arrays():
                eng.gates_level(np.repeat(ops, M), spread(a), spread(b), spread(c), spread(out))
            rows = eng.wires_gather(spread(np.asarray(self._outputs, np.int32)))
        finally:
            eng.set_option("timing_events", timing_before)
        return np.ascontiguousarray(rows.reshape(len(self._outputs), M, -1).transpose(1, 0, 2))

