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

    def exactness(self):
        """(exact_domain, log2 of the worst-case pre-rounding magnitude, predicted rounding margin): is this set inside what a
        Float64 transform computes exactly?  The host-side statement of what tfhe_ctx_create decides (csrc/engine_context.hip:
        exactness_class; tfhe_get_option "exact_domain" / "exact_bound_log2_x1000" / "exact_margin_x1e6") — the reference warns
        about the same limit (polynomials.jl:135-144).  2 = exact for ANY Int32 key words, 1 = exact for every real (uniform) key,
        0 = outside.  Depends on (N, k or max_parties, l, beta) only: the engine-independent criterion tests/fuzz_params.py skips by."""
        products = (self.max_parties + 1 if self.max_parties > 1 else self.tlwe_mask_size + 1) * self.bs_decomp_length
        N, beta = self.tlwe_polynomial_degree, self.bs_log2_base
        bound_log2 = math.log2(products * N) + (beta - 1) + 31
        rms = math.sqrt(products * N) * 2.0 ** (beta + 32) / 12.0
        margin = 4.5 * 2.0 ** -53 * rms * max(4.0, math.log2(N / 2))
        ok = margin < 0.25 and 8.0 * rms < 2.0 ** 51
        return (0 if not ok else 2 if bound_log2 < 51.0 else 1), bound_log2, margin


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
