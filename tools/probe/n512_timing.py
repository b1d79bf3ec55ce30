"""blind_rotate_kernel_n512 by rotations per workgroup (option n512_rw) and batch size: tfhe_parameters_80 with N = 512."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, tfhe_jl_amd as tfhe
b = tfhe.tfhe_parameters_80()
p = tfhe.SchemeParameters(500, b.lwe_noise_stddev, 512, 1, 2, 10, b.bs_noise_stddev, 8, 2, b.ks_noise_stddev, 1)
rng = np.random.default_rng(1)
sk, ck = tfhe.make_key_pair(rng, p, keygen="device")
eng = ck.engine(0)
eng.set_option("pipeline_min", -1)
for g in (1, 256, 512, 768, 1024, 1536, 2048, 3072, 4096, 8192):
    x = rng.integers(-2**31, 2**31, size=(g, 501), dtype=np.int64).astype(np.int32)
    row = {}
    for rw in (1, 4, 2):
        eng.set_option("n512_w2", 1 if rw == 2 else 0)
        eng.set_option("n512_rw", rw if rw != 2 else 0)
        t = []
        for _ in range(5):
            eng.bootstrap(2**29, x, with_keyswitch=False)
            t.append(eng.last_timing_ms(0))
        row[eng.last_kernel_name()] = round(float(np.median(t)), 3)
    print(g, row, flush=True)
