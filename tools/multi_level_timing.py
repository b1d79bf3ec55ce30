#!/usr/bin/env python3
"""A sharded circuit level on a multi-device context: what a level costs the HOST (time inside tfhe_gates_level) and the DEVICES
(wall time per level with everything queued back to back), for the level shapes of examples/tutorial.jl:60-62 — 16 parallel
MUXes — chained so that every level reads what the OTHER device wrote in the level before (the worst case for the exchange of
rows between the replicas of the wire table).  Runs on {0, 0} on a one-GPU box (two device contexts on one GPU: the same calls
as two GPUs; a peer copy within one device is legal) or on --devices 0,1.

  python3 tools/multi_level_timing.py [--devices 0,0] [--levels 200] [--exchange 0|1|2]

Works against the round-4 library as well (its levels exchange through pageable host memory and block the host): the same
script run from a round-4 checkout is the "before" of profiles/r05_multi_level_timing.txt."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tfhe_jl_amd as tfhe  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--devices", default="0,0")
ap.add_argument("--levels", type=int, default=200)
ap.add_argument("--exchange", type=int, default=0)
ap.add_argument("--width", type=int, default=16, help="MUX gates per level")
a = ap.parse_args()
devs = [int(d) for d in a.devices.split(",")]

rng = np.random.default_rng(5)
p = tfhe.tfhe_parameters_80()
sk, ck = tfhe.make_key_pair(rng, p, keygen="device" if hasattr(tfhe.Engine, "keygen_cloud_key") else "host")
W = a.width
bits = rng.integers(0, 2, 3 * W).astype(bool)
enc = tfhe.encrypt(rng, sk, bits).data

res = {"devices": devs, "levels": a.levels, "width": W}
for label, dv in (("one_device", devs[0]), ("multi", devs)):
    eng = ck.engine(dv)
    if label == "multi":
        eng.set_option("level_split_min", 8)
        try:
            eng.set_option("level_exchange", a.exchange)
            res["level_exchange"] = a.exchange
        except tfhe.EngineError:
            res["level_exchange"] = "n/a (round-4 library: pageable host staging, host-synchronous)"
    eng.set_option("timing_events", 0)
    # wires: [0, 3W) inputs; two banks of W outputs written alternately; level t reads bank (t-1) REVERSED as the MUX's second
    # operand, so on two devices every gate's operand was written by the other device's shard of the previous level
    eng.wires_alloc(5 * W)
    eng.wires_upload(0, enc)
    ops = np.full(W, tfhe.OPCODES["MUX"], np.uint8)
    sel = np.arange(W, dtype=np.int32)
    third = np.arange(2 * W, 3 * W, dtype=np.int32)
    banks = [np.arange(3 * W, 4 * W, dtype=np.int32), np.arange(4 * W, 5 * W, dtype=np.int32)]
    prev = np.arange(W, 2 * W, dtype=np.int32)

    def run(levels):
        nonlocal_prev = prev
        host = 0.0
        for t in range(levels):
            out = banks[t & 1]
            t0 = time.perf_counter()
            eng.gates_level(ops, sel, nonlocal_prev[::-1].copy(), third, out)
            host += time.perf_counter() - t0
            nonlocal_prev = out
        final = eng.wires_gather(nonlocal_prev)
        return host, final

    run(20)                                    # warm-up: kernels loaded, buffers sized, clocks up
    # from an idle device: how long the host is held by the first three levels of a burst (the call returns when a level is
    # queued; in the steady state below the host runs up to three levels ahead of the device and is then paced by it)
    t0 = time.perf_counter()
    burst_host, _ = run(3)
    burst_wall = time.perf_counter() - t0
    t0 = time.perf_counter()
    host, final = run(a.levels)
    wall = time.perf_counter() - t0
    res[label] = {"burst_of_3_levels_host_us_inside_gates_level": round(burst_host * 1e6, 1), "burst_of_3_levels_wall_us": round(burst_wall * 1e6, 1),
                  "host_us_per_level_inside_gates_level": round(host / a.levels * 1e6, 1),
                  "wall_us_per_level": round(wall / a.levels * 1e6, 1),
                  "devices_in_last_level": eng.last_device_count() if hasattr(eng, "last_device_count") else None,
                  "final_checksum": int(np.bitwise_xor.reduce(final.view(np.uint32).reshape(-1)))}
assert res["one_device"]["final_checksum"] == res["multi"]["final_checksum"], "multi-device result differs from one device"
print(json.dumps(res))
ck.close()
