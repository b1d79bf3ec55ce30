// Host-side check of the "tan form" lane code (tfhe.jl_amd/csrc/br_core.hpp compiles for the host as well): the forward
// first pass (load_digits2t + dft8_fwd_tw) against the plain one (load_digits2 + dft8<false>) and the fused untwist / rounding
// against the unfused one, on random digits and spectra.  Prints the largest relative difference of the forward pass and the
// number of differing output words of the untwist; tests/test_twist_forms.py asserts on both.
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <random>
#include "br_core.hpp"
using namespace tfhe;
int main()
{
    std::mt19937_64 rng(2026);
    double worst = 0.0, mag = 0.0;
    long diff_words = 0, near_half = 0;
    for (int it = 0; it < 20000; it++) {
        int32_t temp[16];
        for (int i = 0; i < 16; i++) temp[i] = (int32_t)(uint32_t)rng();
        for (int p = 1; p <= 3; p++) {
            const int beta = p == 3 ? 7 : 10;
            cplx a[8], b[8];
            load_digits2(temp, p, beta, a); dft8<false>(a);
            load_digits2t(temp, p, beta, b); dft8_fwd_tw(b);
            for (int i = 0; i < 8; i++) {
                worst = fmax(worst, fmax(fabs(a[i].x - b[i].x), fabs(a[i].y - b[i].y)));
                mag = fmax(mag, fmax(fabs(a[i].x), fabs(a[i].y)));
            }
        }
        // spectra of the size the inverse transform delivers: integers up to 2^48 plus a small error, as in a blind rotation
        cplx y[8];
        for (int i = 0; i < 8; i++) {
            // choose y so that conj(y) e^{-i pi r/16} = (re, im) with re, im integers + eps
            const double re = (double)((int64_t)(rng() >> 16) - ((int64_t)1 << 47)) + 0.05 * ((double)(rng() % 2001) / 1000.0 - 1.0);
            const double im = (double)((int64_t)(rng() >> 16) - ((int64_t)1 << 47)) + 0.05 * ((double)(rng() % 2001) / 1000.0 - 1.0);
            // conj(y) = (re + i im) e^{+i pi r/16}
            const double c = twc(i), s = tws(i);
            const double cr = re * c - im * s, ci = re * s + im * c;
            y[i].x = cr; y[i].y = -ci;
        }
        int32_t acc1[16] = {0}, acc2[16] = {0};
        untwist_add2<false, true>(y, acc1);
        untwist_add2<false, false>(y, acc2);
        for (int i = 0; i < 16; i++) diff_words += acc1[i] != acc2[i];
        (void)near_half;
    }
    printf("forward_max_abs_diff %.3e forward_max_mag %.3e untwist_diff_words %ld\n", worst, mag, diff_words);
    return 0;
}
