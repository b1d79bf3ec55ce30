"""Levelised circuit execution: host-side levelisation on CPU; on the GPU the tutorial circuit
(examples/tutorial.jl) run level by level on the device-resident wire table, checked word for word against the
oracle evaluating the same gates one at a time."""
import numpy as np
import pytest

from conftest import DEVICE_PAIRS


def tutorial_min_circuit(tfhe, nbits=16):
    """examples/tutorial.jl:42-62: carry = MUX(XNOR(a_i, b_i), carry, a_i) over the bits, then out_i = MUX(carry, b_i, a_i)."""
    c = tfhe.Circuit()
    a, b = c.inputs(nbits), c.inputs(nbits)
    carry = c.constant(False)
    for i in range(nbits):
        carry = c.mux(c.xnor(a[i], b[i]), carry, a[i])
    c.set_outputs([c.mux(carry, b[i], a[i]) for i in range(nbits)])
    return c


def test_levelisation(tfhe):
    c = tutorial_min_circuit(tfhe, 16)
    lv = c.levels()
    # level 1: the constant and the 16 XNORs (all independent); then a 16-deep MUX chain; then 16 parallel MUXes
    assert len(lv) == 18
    assert len(lv[0]) == 17 and all(len(l) == 1 for l in lv[1:17]) and len(lv[17]) == 16
    assert c.num_wires == 32 + 1 + 16 + 16 + 16
    for ops, a, b, cc, out in c.level_arrays():
        assert ops.dtype == np.uint8 and a.shape == b.shape == cc.shape == out.shape == ops.shape
        assert not (set(out.tolist()) & (set(a.tolist()) | set(b.tolist()) | set(cc.tolist())))
    with pytest.raises(ValueError):
        c.gate("NAND", 0)           # wrong arity
    with pytest.raises(ValueError):
        c.gate("NAND", 0, 10**6)    # unknown wire


def _oracle_eval(circuit, oracle_obj, orc, inputs):
    wires = [row for row in inputs]
    for name, a, b, cc in circuit._gates:
        ops = np.array([orc.OPS[name]], np.uint8)
        z = np.zeros((1, inputs.shape[1]), np.int32)
        x = wires[a][None] if a >= 0 else z
        y = wires[b][None] if b >= 0 else z
        w = wires[cc][None] if cc >= 0 else z
        wires.append(oracle_obj.gates(ops, x, y, w)[0])
    return np.stack([wires[w] for w in circuit._outputs])


@pytest.mark.gpu
def test_tutorial_circuit_on_device(tfhe, orc, keys80):
    K = keys80
    c = tutorial_min_circuit(tfhe, 16)
    bits = [(2017 >> i) & 1 == 1 for i in range(16)] + [(42 >> i) & 1 == 1 for i in range(16)]
    enc = tfhe.encrypt(K.rng, K.sk, bits)
    res = c.run(K.ck, enc)
    assert sum(int(v) << i for i, v in enumerate(tfhe.decrypt(K.sk, res))) == 42     # "Answer: 42"
    assert np.array_equal(res.data, _oracle_eval(c, K.oracle, orc, enc.data))
    # error paths of the level API
    eng = K.ck.engine(0)
    with pytest.raises(tfhe.EngineError):   # reads a wire written in the same level
        eng.gates_level(np.array([0, 0], np.uint8), np.array([0, 40], np.int32), np.array([1, 1], np.int32), None, np.array([40, 41], np.int32))
    with pytest.raises(tfhe.EngineError):   # out of range
        eng.gates_level(np.array([0], np.uint8), np.array([0], np.int32), np.array([10**6], np.int32), None, np.array([40], np.int32))


@pytest.mark.gpu
def test_circuit_on_a_batch_of_input_sets(tfhe, keys80):
    """Circuit.run_batch: M instances of the tutorial circuit, every level one call over all instances — word for word what
    Circuit.run gives for each instance alone (every gate is a function of its own operands only), and the right minima."""
    K = keys80
    c = tutorial_min_circuit(tfhe, 8)
    rng = np.random.default_rng(11)
    pairs = [(int(rng.integers(0, 256)), int(rng.integers(0, 256))) for _ in range(5)] + [(7, 7)]
    encs = []
    for x, y in pairs:
        bits = [(x >> i) & 1 == 1 for i in range(8)] + [(y >> i) & 1 == 1 for i in range(8)]
        encs.append(tfhe.encrypt(K.rng, K.sk, bits))
    got = c.run_batch(K.ck, encs)
    assert got.shape == (len(pairs), 8, K.params.lwe_size + 1)
    for i, (x, y) in enumerate(pairs):
        assert np.array_equal(got[i], c.run(K.ck, encs[i]).data), i
        assert sum(int(v) << j for j, v in enumerate(tfhe.decrypt(K.sk, tfhe.LweSampleArray(got[i])))) == min(x, y)
    assert c.run_batch(K.ck, np.zeros((0, 16, K.params.lwe_size + 1), np.int32)).shape == (0, 8, K.params.lwe_size + 1)
    with pytest.raises(ValueError):
        c.run_batch(K.ck, np.zeros((2, 3, K.params.lwe_size + 1), np.int32))


@pytest.mark.gpu
def test_levels_without_timing_events_and_across_streams(tfhe, keys80):
    """`timing_events` = 0 (what Circuit.run sets while its levels run): same words, nothing for tfhe_last_timing_ms to report;
    and the ordering a call owes the previous one when the caller changes streams between calls (the context's own stream records
    its end-of-call event only on demand)."""
    K = keys80
    eng = K.ck.engine(0)
    rng = np.random.default_rng(5)
    B = 40
    x, y = [tfhe.encrypt(K.rng, K.sk, rng.integers(0, 2, B).astype(bool)).data for _ in range(2)]
    ops = np.full(B, tfhe.OPCODES["NAND"], np.uint8)
    want = eng.gates(ops, x, y)
    assert eng.last_timing_ms(0) > 0.0
    idx = np.arange(B, dtype=np.int32)
    eng.wires_alloc(3 * B)
    eng.wires_upload(0, np.concatenate([x, y]))
    eng.set_option("timing_events", 0)
    try:
        eng.gates_level(ops, idx, idx + B, None, idx + 2 * B)
        assert np.array_equal(eng.wires_download(2 * B, B), want)
        with pytest.raises(tfhe.EngineError):
            eng.last_timing_ms(0)
    finally:
        eng.set_option("timing_events", 1)
    # own stream -> a caller's stream -> own stream, each call consuming the previous one's result through the shared workspaces
    # (device buffers and the second stream straight from the HIP runtime the engine already loaded: no second runtime in the process)
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    nbytes = x.nbytes

    def dev(host=None):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
        if host is not None:
            assert hip.hipMemcpy(p, host.ctypes.data_as(C.c_void_p), C.c_size_t(nbytes), 1) == 0      # hipMemcpyHostToDevice
        return p

    def host_of(p):
        out = np.empty_like(x)
        assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, C.c_size_t(nbytes), 2) == 0           # hipMemcpyDeviceToHost
        return out

    dx, dy, d1, d2 = dev(x), dev(y), dev(), dev()
    side = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(side)) == 0
    try:
        eng.gates_dev(ops, dx.value, dy.value, 0, d1.value, B)                          # own stream
        eng.gates_dev(ops, d1.value, dy.value, 0, d2.value, B, stream=side.value)       # the caller's: must wait for the first
        eng.gates_dev(ops, d2.value, dx.value, 0, d1.value, B)                          # own stream again: must wait for the second
        assert hip.hipDeviceSynchronize() == 0
        step2 = eng.gates(ops, want, y)
        assert np.array_equal(host_of(d2), step2)
        assert np.array_equal(host_of(d1), eng.gates(ops, step2, x))
    finally:
        hip.hipStreamDestroy(side)
        for p in (dx, dy, d1, d2):
            hip.hipFree(p)


@pytest.mark.gpu
def test_log_depth_minimum_circuit_on_device(tfhe, orc, keys80):
    """examples/tutorial.py, log_depth=True: the comparator ripple of examples/tutorial.jl:42-56 replaced by a reduction tree
    (7 levels instead of 18); same function, checked against the oracle word for word and against min() on decrypted bits."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from tutorial import encrypted_minimum_circuit
    K = keys80
    c = encrypted_minimum_circuit(16, log_depth=True)
    assert [len(l) for l in c.levels()] == [16, 16, 8, 4, 2, 1, 16]
    rng = np.random.default_rng(5)
    for a, b in [(2017, 42), (42, 2017), (65535, 65535), (0, 1)] + [tuple(int(v) for v in rng.integers(0, 65536, 2)) for _ in range(3)]:
        bits = [(a >> i) & 1 == 1 for i in range(16)] + [(b >> i) & 1 == 1 for i in range(16)]
        enc = tfhe.encrypt(K.rng, K.sk, bits)
        res = c.run(K.ck, enc)
        assert sum(int(v) << i for i, v in enumerate(tfhe.decrypt(K.sk, res))) == min(a, b), (a, b)
    assert np.array_equal(res.data, _oracle_eval(c, K.oracle, orc, enc.data))


@pytest.mark.gpu
@pytest.mark.parametrize("devs", DEVICE_PAIRS)
def test_circuits_on_a_multi_device_context(tfhe, orc, keys80, devs):
    """A multi-device context ({0, 0} on the one-GPU box: two device contexts, two replicas of the wire table) runs narrow
    levels on its first device and shards wide ones (option level_split_min, lowered here so that the 17- and 16-gate levels
    of the tutorial circuit split).  A device fetches the operand rows whose current value another device holds right before it
    reads them — device to device (hipMemcpyPeerAsync; option level_exchange 1) or through pinned host memory (2), both
    forced here, 0 = by peer access — and nothing else travels: same words as one device, every intermediate wire included,
    for the reference's circuit (examples/tutorial.jl:42-62) and for the log-depth variant whose wide levels feed narrow ones."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from tutorial import encrypted_minimum_circuit
    K = keys80
    bits = [(2017 >> i) & 1 == 1 for i in range(16)] + [(42 >> i) & 1 == 1 for i in range(16)]
    enc = tfhe.encrypt(K.rng, K.sk, bits)
    multi = K.ck.engine(devs)
    print(f"devices {devs}: device-to-device copies allowed between {multi.get_option('peer_pairs')} of {len(devs) * (len(devs) - 1)} ordered pairs")
    for circ in (tutorial_min_circuit(tfhe, 16), encrypted_minimum_circuit(16, log_depth=True)):
        want = circ.run(K.ck, enc).data
        want_all = K.ck.engine(0).wires_gather(np.arange(circ.num_wires, dtype=np.int32))
        for exchange in (1, 2, 0):            # device-to-device copies / pinned host staging / by peer access
            multi.set_option("level_exchange", exchange)
            assert multi.get_option("level_exchange") == exchange
            for split in (4096, 8, 2):        # never split at these sizes / only the 16-gate levels / everything but single gates
                multi.set_option("level_split_min", split)
                got = circ.run(K.ck, enc, device=devs).data
                assert np.array_equal(got, want), (exchange, split)
                # the last level of both circuits is 16 parallel MUXes = 32 rotations: sharded over both device contexts iff the
                # threshold allows it (tfhe_last_device_count, ABI v5)
                assert multi.last_device_count() == (1 if split == 4096 else 2), split
                # every intermediate wire, read back through the first device (which fetches what the second one computed)
                assert np.array_equal(multi.wires_gather(np.arange(circ.num_wires, dtype=np.int32)), want_all), (exchange, split)
    # a level of ONE gate is never sharded, whatever the threshold
    multi.set_option("level_split_min", 1)
    multi.wires_alloc(4)
    multi.wires_upload(0, enc.data[:2])
    multi.gates_level(np.array([0], np.uint8), np.array([0], np.int32), np.array([1], np.int32), None, np.array([2], np.int32))
    assert multi.last_device_count() == 1
    assert np.array_equal(multi.wires_download(2, 1), K.ck.engine(0).gates(np.array([0], np.uint8), enc.data[:1], enc.data[1:2]))
    # intermediate wires are coherent too: every wire of the table, read back from the first device
    circ = tutorial_min_circuit(tfhe, 16)
    multi.set_option("level_split_min", 8)
    circ.run(K.ck, enc, device=devs)
    all_multi = multi.wires_gather(np.arange(circ.num_wires, dtype=np.int32))
    circ.run(K.ck, enc)
    assert np.array_equal(all_multi, K.ck.engine(0).wires_gather(np.arange(circ.num_wires, dtype=np.int32)))
    multi.set_option("level_split_min", 4096)


@pytest.mark.gpu
@pytest.mark.parametrize("devs", DEVICE_PAIRS)
def test_streamed_batches_on_a_multi_device_context(tfhe, keys80, orc, devs):
    """tfhe_gates_batch_submit / _wait on a multi-device context: every device takes its shard as a submit of its own (two
    batches in flight per device); results equal the blocking call's and the oracle's on a sample."""
    K = keys80
    multi = K.ck.engine(devs)
    rng = np.random.default_rng(31)
    names = ["NAND", "AND", "OR", "XOR", "MUX"]
    jobs = []
    for B in (600, 64, 3, 900):
        ops = np.array([tfhe.OPCODES[names[i]] for i in rng.integers(0, 5, B)], np.uint8)
        ins = [tfhe.encrypt(K.rng, K.sk, rng.integers(0, 2, B).astype(bool)).data for _ in range(3)]
        jobs.append((ops, ins, multi.gates(ops, *ins)))
    idx = [0, 1, 2]
    assert np.array_equal(jobs[0][2][idx], K.oracle.gates(jobs[0][0][idx], *[a[idx] for a in jobs[0][1]], nthreads=3))
    tickets, outs = [], []
    for ops, ins, _ in jobs:                  # four submits: the third displaces (waits for) the first
        pin = [tfhe.pinned_empty(a.shape) for a in ins]
        for p, a in zip(pin, ins):
            p[:] = a
        t, o = multi.gates_submit(ops, *pin)
        tickets.append(t); outs.append(o)
    for t in tickets:
        multi.gates_wait(t)
    for (ops, ins, want), got in zip(jobs, outs):
        assert np.array_equal(got, want)
