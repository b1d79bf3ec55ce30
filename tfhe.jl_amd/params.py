"""Scheme parameters — mirrors src/api.jl:4-82 and src/mk_api.jl:4-34 of the reference."""
import math
from dataclasses import dataclass


@dataclass(frozen=True)
class SchemeParameters:
    """TFHE scheme parameters (single- or multi-party); field names as api.jl:4-21."""
    lwe_size: int
    lwe_noise_stddev: float
    tlwe_polynomial_degree: int
    tlwe_mask_size: int
    bs_decomp_length: int
    bs_log2_base: int
    bs_noise_stddev: float
    ks_decomp_length: int
    ks_log2_base: int
    ks_noise_stddev: float
    max_parties: int

    # the subset the device context needs (include/tfhe_mi355x.h: tfhe_params)
    def engine_tuple(self):
        return (self.lwe_size, self.tlwe_polynomial_degree, self.tlwe_mask_size, self.bs_decomp_length,
                self.bs_log2_base, self.ks_decomp_length, self.ks_log2_base, self.max_parties)


def tfhe_parameters_80(tlwe_mask_size: int = 1) -> SchemeParameters:
    """~80 bits of security (api.jl:30-45)."""
    return SchemeParameters(
        500, 1 / 2**15 * math.sqrt(2 / math.pi),
        1024, tlwe_mask_size,
        2, 10, 9e-9 * math.sqrt(2 / math.pi),
        8, 2, 1 / 2**15 * math.sqrt(2 / math.pi),
        1)


def tfhe_parameters_128(tlwe_mask_size: int = 1) -> SchemeParameters:
    """~128 bits of security (api.jl:55-69)."""
    return SchemeParameters(
        630, 1 / 2**15,
        1024, tlwe_mask_size,
        3, 7, 1 / 2**25,
        8, 2, 1 / 2**15,
        1)


# mk_api.jl:4-34
mktfhe_parameters_2party = SchemeParameters(500, 0.012467, 1024, 1, 4, 7, 3.29e-10, 8, 2, 2.44e-5, 2)
mktfhe_parameters_4party = SchemeParameters(500, 0.012467, 1024, 1, 5, 6, 3.29e-10, 8, 2, 2.44e-5, 4)
mktfhe_parameters_8party = SchemeParameters(500, 0.012467, 1024, 1, 8, 4, 3.29e-10, 8, 2, 2.44e-5, 8)
