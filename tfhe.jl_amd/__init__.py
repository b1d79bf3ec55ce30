"""tfhe.jl_amd — MI355X-native TFHE gate bootstrapping behind the reference's gate_*/key API.

Exports mirror src/TFHE.jl:24-61 of nucypher/TFHE.jl.  The hot path (gate_* -> bootstrap ->
keyswitch) runs in hand-written HIP kernels (csrc/, C ABI in include/tfhe_mi355x.h); key generation,
encryption and decryption stay on the host, as in the reference.
"""
from .params import (SchemeParameters, tfhe_parameters_80, tfhe_parameters_128,
                     mktfhe_parameters_2party, mktfhe_parameters_4party, mktfhe_parameters_8party)
from .lwe import LweSample, LweSampleArray
from .keys import SecretKey, CloudKey, make_key_pair, encrypt, decrypt
from .gates import (gate_nand, gate_or, gate_and, gate_xor, gate_xnor, gate_not, gate_constant, gate_nor,
                    gate_andny, gate_andyn, gate_orny, gate_oryn, gate_mux, gates_batch)
from .mk_keys import SharedKey, CloudKeyPart, MKCloudKey, MKLweSample, mk_encrypt, mk_decrypt, mk_gate_nand
from .circuit import Circuit
from .serialize import save_cloud_key, load_cloud_key
from ._lib import Engine, EngineError, OPCODES, LIB_PATH, pinned_empty

__all__ = [
    "make_key_pair", "LweSample", "LweSampleArray", "SecretKey", "CloudKey", "encrypt", "decrypt",
    "tfhe_parameters_80", "tfhe_parameters_128", "SchemeParameters",
    "gate_nand", "gate_or", "gate_and", "gate_xor", "gate_xnor", "gate_not", "gate_constant", "gate_nor",
    "gate_andny", "gate_andyn", "gate_orny", "gate_oryn", "gate_mux", "gates_batch",
    "mktfhe_parameters_2party", "mktfhe_parameters_4party", "mktfhe_parameters_8party",
    "SharedKey", "CloudKeyPart", "MKCloudKey", "MKLweSample", "mk_encrypt", "mk_decrypt", "mk_gate_nand",
    "Circuit", "save_cloud_key", "load_cloud_key", "Engine", "EngineError", "OPCODES", "LIB_PATH", "pinned_empty",
]
