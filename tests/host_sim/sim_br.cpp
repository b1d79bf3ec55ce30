// sim_br.cpp — executes the blind-rotate kernel's per-lane code (tfhe.jl_amd/csrc/br_core.hpp) on the
// host, lane by lane with the LDS modelled as plain arrays, so the index maths, exchanges,
// decomposition, rounding and key layout can be checked against the oracle without a GPU.
// The phase order mirrors blind_rotate_kernel / bk_prepare_kernel in tfhe_engine.hip.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../tfhe.jl_amd/csrc/br_core.hpp"

using namespace tfhe;

namespace {

struct HostTables {
    std::vector<cplx> buf;
    std::vector<cplx> full;   // tw1 | tw2 | twist | tw1f, as the engine builds them (fill_tables)
    Tables T;
    Tables Tfull;
    HostTables()
    {
        full.resize(kTableElems);
        fill_tables<long double>(full.data(), [](long double a) { return cosl(a); }, [](long double a) { return sinl(a); });
        Tfull = tables_from(full.data());
        const long double pi = 3.14159265358979323846264338327950288L;
        buf.resize(512 + 64 + 512);
        for (int q = 0; q < 8; q++)
            for (int t = 0; t < 64; t++) {
                const long double a = -2.0L * pi * (long double)(t * q) / 512.0L;
                buf[q * 64 + t] = mk((double)cosl(a), (double)sinl(a));
            }
        for (int q = 0; q < 8; q++)
            for (int t = 0; t < 8; t++) {
                const long double a = -2.0L * pi * (long double)(t * q) / 64.0L;
                buf[512 + q * 8 + t] = mk((double)cosl(a), (double)sinl(a));
            }
        for (int r = 0; r < 8; r++)
            for (int t = 0; t < 64; t++) {
                const long double a = -pi * (long double)(t + 64 * r) / (long double)kN;
                buf[576 + r * 64 + t] = mk((double)cosl(a), (double)sinl(a));
            }
        T.tw1 = buf.data();
        T.tw2 = buf.data() + 512;
        T.twist = buf.data() + 576;
    }
};

const HostTables &tables()
{
    static HostTables t;
    return t;
}

typedef cplx Regs[64][8];

void forward_fft(Regs &x, const Tables &T)
{
    std::vector<cplx> xch(kXchElems);
    for (int l = 0; l < 64; l++) { fwd_pass_a(l, x[l], T); x1_store_a(l, x[l], xch.data()); }
    for (int l = 0; l < 64; l++) x1_load_b(l, x[l], xch.data());
    for (int l = 0; l < 64; l++) { fwd_pass_b(l, x[l], T); x2_store(l, x[l], xch.data()); }
    for (int l = 0; l < 64; l++) x2_load(l, x[l], xch.data());
    for (int l = 0; l < 64; l++) fwd_pass_c(x[l]);
}

void inverse_fft(Regs &x, const Tables &T)
{
    std::vector<cplx> xch(kXchElems);
    for (int l = 0; l < 64; l++) { inv_pass_c(x[l]); x2_store(l, x[l], xch.data()); }
    for (int l = 0; l < 64; l++) x2_load(l, x[l], xch.data());
    for (int l = 0; l < 64; l++) { inv_pass_b(l, x[l], T); x1_store_b(l, x[l], xch.data()); }
    for (int l = 0; l < 64; l++) x1_load_a(l, x[l], xch.data());
    for (int l = 0; l < 64; l++) inv_pass_a(l, x[l], T);
}

}  // namespace

extern "C" {

// bk_prepare_kernel: Int32 polys [npolys][N] -> engine-order spectra [npolys][8][64] (re,im), scaled 1/M
void sim_bk_prepare(const int32_t *bk_i32, int64_t npolys, double *out)
{
    const Tables &T = tables().T;
    for (int64_t q = 0; q < npolys; q++) {
        Regs x;
        for (int l = 0; l < 64; l++) load_poly(l, bk_i32 + q * kN, T, x[l]);
        forward_fft(x, T);
        for (int l = 0; l < 64; l++)
            for (int k2 = 0; k2 < 8; k2++) {
                out[((q * kM) + k2 * 64 + l) * 2 + 0] = x[l][k2].x / kM;
                out[((q * kM) + k2 * 64 + l) * 2 + 1] = x[l][k2].y / kM;
            }
    }
}

// frequency held at (lane, k2) — lets the test compare sim_bk_prepare with the reference-order spectra
int32_t sim_freq_of(int32_t lane, int32_t k2) { return freq_of(lane, k2); }

// blind_rotate_kernel<L, 2>: bara [n+1] (barb last), bk spectra in engine order, ext out [N+1].
// Returns the largest distance of a pre-round value from an integer over the whole rotation.
double sim_blind_rotate(const int32_t *bara, int32_t n, int32_t L, int32_t log2_base, int32_t mu,
                        const double *bk_spec, int32_t *ext)
{
    const Tables &T = tables().T;
    const Gadget g = make_gadget(L, log2_base);
    const int K1 = 2;
    static int32_t acc[2][64][16];
    std::vector<int32_t> acc_lds(K1 * kN);
    const cplx *bk = reinterpret_cast<const cplx *>(bk_spec);
    double worst = 0.0;

    const int barb = bara[n] & (2 * kN - 1);
    for (int l = 0; l < 64; l++)
        for (int m = 0; m < 16; m++) {
            acc[0][l][m] = 0;
            const int idx = (l + 64 * m + barb) & (2 * kN - 1);
            acc[1][l][m] = (idx & kN) ? (int32_t)(0u - (uint32_t)mu) : mu;
        }
    for (int c = 0; c < K1; c++)
        for (int l = 0; l < 64; l++)
            for (int m = 0; m < 16; m++) acc_lds[c * kN + l + 64 * m] = acc[c][l][m];

    for (int i = 0; i < n; i++) {
        const int a = bara[i] & (2 * kN - 1);
        if (a == 0) continue;
        const cplx *bki = bk + (size_t)i * (L * K1 * K1 * kM);
        static Regs out[2];
        for (int c = 0; c < K1; c++)
            for (int l = 0; l < 64; l++)
                for (int q = 0; q < 8; q++) out[c][l][q] = mk(0.0, 0.0);
        for (int c = 0; c < K1; c++) {
            static int32_t temp[64][16];
            for (int l = 0; l < 64; l++) rotate_sub(l, a, acc_lds.data() + c * kN, acc[c][l], g.offset, temp[l]);
            for (int p = 1; p <= L; p++) {
                Regs x;
                for (int l = 0; l < 64; l++) load_digits(l, temp[l], p, g, T, x[l]);
                forward_fft(x, T);
                for (int l = 0; l < 64; l++) {
                    const cplx *kp = bki + (size_t)((p - 1) * K1 + c) * K1 * kM + l;
                    for (int co = 0; co < K1; co++)
                        for (int k2 = 0; k2 < 8; k2++)
                            out[co][l][k2] = cfma(x[l][k2], kp[(co * 8 + k2) * 64], out[co][l][k2]);
                }
            }
        }
        for (int co = 0; co < K1; co++) {
            inverse_fft(out[co], T);
            for (int l = 0; l < 64; l++) {
                for (int r = 0; r < 8; r++) {   // rounding margin, same formula as untwist_add
                    const cplx w = T.twist[r * 64 + l];
                    const double re = out[co][l][r].x * w.x + out[co][l][r].y * w.y;
                    const double im = out[co][l][r].x * w.y - out[co][l][r].y * w.x;
                    worst = std::fmax(worst, std::fabs(re - std::nearbyint(re)));
                    worst = std::fmax(worst, std::fabs(im - std::nearbyint(im)));
                }
                untwist_add(l, out[co][l], T, acc[co][l]);
            }
            for (int l = 0; l < 64; l++)
                for (int m = 0; m < 16; m++) acc_lds[co * kN + l + 64 * m] = acc[co][l][m];
        }
    }
    for (int l = 0; l < 64; l++)
        for (int m = 0; m < 16; m++) {
            const int j = l + 64 * m;
            if (j == 0) ext[0] = acc[0][l][m];
            else ext[kN - j] = (int32_t)(0u - (uint32_t)acc[0][l][m]);
        }
    ext[kN] = acc[1][0][0];
    return worst;
}

// blind_rotate_kernel_v3's lane code (the shipped path): rotate_sub3 on mirror-extended polynomial images /
// load_digits2t + dft8_fwd_tw (the constant twist in tan form, as the device kernels) / tw1f-folded passes / untwist_add2, lane by lane.  bk spectra in engine order as sim_bk_prepare produces them.
double sim_blind_rotate_v3(const int32_t *bara, int32_t n, int32_t L, int32_t log2_base, int32_t mu,
                           const double *bk_spec, int32_t *ext)
{
    const Tables &T = tables().Tfull;
    const Gadget g = make_gadget(L, log2_base);
    const int32_t xormask = gadget_xor_mask(L, log2_base);
    const int K1 = 2;
    constexpr int kImg = kMir + kN;                 // one polynomial image: mirror | coefficients
    std::vector<int32_t> acc_lds(K1 * kImg);
    const cplx *bk = reinterpret_cast<const cplx *>(bk_spec);
    static LaneTw tw[64];
    for (int l = 0; l < 64; l++) load_lane_tw(l, T, tw[l]);
    double worst = 0.0;
    const int barb = bara[n] & (2 * kN - 1);
    for (int l = 0; l < 64; l++) {
        int32_t z[16], b[16];
        for (int m = 0; m < 16; m++) {
            z[m] = 0;
            const int idx = (l + 64 * m + barb) & (2 * kN - 1);
            b[m] = (idx & kN) ? (int32_t)(0u - (uint32_t)mu) : mu;
        }
        store_cur<16>(l, z, acc_lds.data());
        store_cur<16>(l, b, acc_lds.data() + kImg);
    }
    std::vector<cplx> xch(kXchElems);
    for (int i = 0; i < n; i++) {
        const int a = bara[i] & (2 * kN - 1);          // no skip for a == 0: the step must add exactly zero
        const cplx *bki = bk + (size_t)i * (L * K1 * K1 * kM);
        static Regs out[2];
        for (int c = 0; c < K1; c++)
            for (int l = 0; l < 64; l++)
                for (int q = 0; q < 8; q++) out[c][l][q] = mk(0.0, 0.0);
        for (int c = 0; c < K1; c++) {
            static int32_t temp[64][16];
            for (int l = 0; l < 64; l++) {
                int32_t cur[16];
                load_cur<16>(l, acc_lds.data() + c * kImg, cur);
                rotate_sub3<16>(l, a, acc_lds.data() + c * kImg, cur, g.offset, xormask, temp[l]);
            }
            for (int p = 1; p <= L; p++) {
                Regs x;
                for (int l = 0; l < 64; l++) { load_digits2t(temp[l], p, log2_base, x[l]); dft8_fwd_tw(x[l]); for (int q = 0; q < 8; q++) x[l][q] = cmul(x[l][q], tw[l].tw1f[q]); x1_store_a(l, x[l], xch.data()); }
                for (int l = 0; l < 64; l++) x1_load_b(l, x[l], xch.data());
                for (int l = 0; l < 64; l++) { fwd2_pass_b(x[l], tw[l]); x2_store(l, x[l], xch.data()); }
                for (int l = 0; l < 64; l++) x2_load(l, x[l], xch.data());
                for (int l = 0; l < 64; l++) {
                    fwd_pass_c(x[l]);
                    const cplx *kp = bki + (size_t)((p - 1) * K1 + c) * K1 * kM + l;
                    for (int co = 0; co < K1; co++)
                        for (int k2 = 0; k2 < 8; k2++) out[co][l][k2] = cfma(x[l][k2], kp[(co * 8 + k2) * 64], out[co][l][k2]);
                }
            }
        }
        for (int co = 0; co < K1; co++) {
            for (int l = 0; l < 64; l++) { inv_pass_c(out[co][l]); x2_store(l, out[co][l], xch.data()); }
            for (int l = 0; l < 64; l++) x2_load(l, out[co][l], xch.data());
            for (int l = 0; l < 64; l++) { inv2_pass_b(out[co][l], tw[l]); x1_store_b(l, out[co][l], xch.data()); }
            for (int l = 0; l < 64; l++) x1_load_a(l, out[co][l], xch.data());
            for (int l = 0; l < 64; l++) {
                inv2_pass_a(out[co][l], tw[l]);
                int32_t accr[16];
                load_cur<16>(l, acc_lds.data() + co * kImg, accr);
                untwist_add2<true>(out[co][l], accr, &worst);
                store_cur<16>(l, accr, acc_lds.data() + co * kImg);
            }
        }
    }
    const int32_t *a0 = acc_lds.data() + kMir, *b0 = acc_lds.data() + kImg + kMir;
    ext[0] = a0[0];
    for (int j = 1; j < kN; j++) ext[kN - j] = (int32_t)(0u - (uint32_t)a0[j]);
    ext[kN] = b0[0];
    return worst;
}

}  // extern "C"
