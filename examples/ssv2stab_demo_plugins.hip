// The reference's own two demo problems (docs/Demo_SSV2stab.ipynb: the 3-D heat
// equation with a travelling tanh front, cells "heat problem", and the 3-D combustion
// benchmark of the RKC paper, cells 1-3) as USER plugins of extensisq_amd: two
// pointwise functors on csrc/esq_stencil3d.hpp, compiled with
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
//           -I <repo>/extensisq_amd/csrc examples/ssv2stab_demo_plugins.hip -o libdemo.so
// and bound from Python in examples/ssv2stab_demo_plugins.py.  NumPy twins with the
// same operation order: oracle/problems.py (tanh3d_problem, combustion3d_problem).
#include "esq_stencil3d.hpp"

namespace {

// ---- u_t = lap u + s(x, y, z, t),  exact solution tanh(5x + 10y + 7.5z - 2.5 - 5t),
// Dirichlet data from the exact solution (time dependent), N^3 interior points of a
// grid of (N + 2)^3, mesh width 1 / (N + 1).  Array axes as the notebook's
// np.meshgrid(x, x, x) lays them out: the SECOND index runs along x, the first along y.
struct TanhHeatFn {
    static constexpr bool kZeroOutside = false, kAutonomous = false;
    double inv_h2, step;                          // (N + 1)^2, 1 / (N + 1)
    __device__ double exact(int a, int b, int c, double t) const {     // array indices 0 .. N + 1
        const double X = b * step, Y = a * step, Z = c * step;
        return tanh(((5 * X + 10 * Y) + 7.5 * Z) - (2.5 + 5 * t));
    }
    __device__ double ghost(int, int face, int i, int j, int l, double, double t) const {
        const int a = i + 1 + (face == 0 ? -1 : face == 1 ? 1 : 0);
        const int b = j + 1 + (face == 2 ? -1 : face == 3 ? 1 : 0);
        const int c = l + 1 + (face == 4 ? -1 : face == 5 ? 1 : 0);
        return exact(a, b, c, t);
    }
    __device__ void eval(const esq::Nb3 (&nb)[1], int i, int j, int l, double t,
                         double (&f)[1]) const {
        const esq::Nb3 &u = nb[0];
        const double lap = inv_h2 * ((((((-6 * u.c + u.below) + u.above) + u.up) + u.dn) + u.lf) + u.rt);
        const double s = exact(i + 1, j + 1, l + 1, t);
        const double src = (362.5 * (s - s * s * s) + 5 * (s * s)) - 5;
        f[0] = lap + src;
    }
};

// ---- c_t = lap c - D c exp(-delta / T),   L T_t = lap T + alpha D c exp(-delta / T):
// two fields on N^3 cells, mirror conditions on the three low faces, value 1 on the
// three high faces, mesh width 1 / (N + 1/2); state = [c, T]
struct CombustionFn {
    static constexpr bool kZeroOutside = false, kAutonomous = true;
    double inv_h2, damkohler, delta, alpha, lewis;
    __device__ double ghost(int, int face, int, int, int, double inside, double) const {
        return (face & 1) ? 1.0 : inside;
    }
    __device__ void eval(const esq::Nb3 (&nb)[2], int, int, int, double, double (&f)[2]) const {
        double lap[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const esq::Nb3 &u = nb[q];
            lap[q] = inv_h2 * ((((((-6 * u.c + u.below) + u.above) + u.up) + u.dn) + u.lf) + u.rt);
        }
        const double react = (damkohler * nb[0].c) * exp(-delta / nb[1].c);
        f[0] = lap[0] - react;
        f[1] = (lap[1] + alpha * react) / lewis;
    }
};

// ---- u_t = cx u_xx + cy u_yy + cz u_zz, homogeneous Dirichlet, N^3 interior points of a
// grid of mesh width 1 / (N + 1): NOT one of the notebook's problems -- a one-field functor
// that declares kZeroOutside && kAutonomous and therefore gets the 16-byte sweeps and both
// kinds of chain sweeps (esq_chain3d.hpp, esq_rkc3d.hpp) of the header besides
struct AnisoFn {
    static constexpr bool kZeroOutside = true, kAutonomous = true;
    double ci, cj, cl;                            // coefficient / h^2 along i, j, l
    __device__ double ghost(int, int, int, int, int, double, double) const { return 0.0; }
    __device__ void eval(const esq::Nb3 (&nb)[1], int, int, int, double, double (&f)[1]) const {
        const esq::Nb3 &u = nb[0];
        f[0] = (ci * ((u.below + u.above) - 2.0 * u.c) + cj * ((u.up + u.dn) - 2.0 * u.c)) +
               cl * ((u.lf + u.rt) - 2.0 * u.c);
    }
};

template <class Fn>
struct User {
    int N;
    Fn fn;
};
using Tanh = esq::Stencil3D<1, TanhHeatFn>;
using Comb = esq::Stencil3D<2, CombustionFn>;
using Aniso = esq::Stencil3D<1, AnisoFn>;

}  // namespace

#define DEMO_ENTRIES(PFX, P, FN, NF)                                                          \
    extern "C" int PFX##_rhs(void *user, double t, const double *y, double *f, size_t n,      \
                             void *stream) {                                                  \
        const User<FN> *u = (const User<FN> *)user;                                           \
        if (!u || n != (size_t)NF * P::points(u->N)) return ESQ_EINVAL;                       \
        return P::rhs(u->fn, u->N, t, y, f, stream);                                          \
    }                                                                                         \
    extern "C" int PFX##_fused(void *user, double t, const double *y, double *f,              \
                               const esq_epilogue *epi, size_t n, void *stream, void *e0,     \
                               void *e1) {                                                    \
        const User<FN> *u = (const User<FN> *)user;                                           \
        if (!u || n != (size_t)NF * P::points(u->N)) return ESQ_EINVAL;                       \
        return P::fused(u->fn, u->N, t, y, f, epi, stream, e0, e1);                           \
    }                                                                                         \
    extern "C" int PFX##_rkc(void *user, double t, const double *yjm1, const double *yjm2,    \
                             const double *yn, const double *fn, double mu, double nu,        \
                             double omn, double hmus, double ajm1, double *y_out, size_t n,   \
                             void *stream, void *e0, void *e1) {                              \
        const User<FN> *u = (const User<FN> *)user;                                           \
        if (!u || n != (size_t)NF * P::points(u->N)) return ESQ_EINVAL;                       \
        return P::rkc(u->fn, u->N, t, yjm1, yjm2, yn, fn, mu, nu, omn, hmus, ajm1, y_out,     \
                      stream, e0, e1);                                                        \
    }
DEMO_ENTRIES(tanh3d, Tanh, TanhHeatFn, 1)
DEMO_ENTRIES(comb3d, Comb, CombustionFn, 2)
DEMO_ENTRIES(aniso3d, Aniso, AnisoFn, 1)
// the chain entries of the one-field homogeneous functor (tuning knobs from the
// environment, as the built-in plugin reads them: ESQ_RKC_FORCE / ESQ_RKC_PLANES for tests)
extern "C" int aniso3d_chain(void *user, const double *y_in, const esq_chain *chain, size_t n,
                             void *stream, void *e0, void *e1) {
    const User<AnisoFn> *u = (const User<AnisoFn> *)user;
    if (!u || n != Aniso::points(u->N)) return ESQ_EINVAL;
    return Aniso::chain(u->fn, u->N, y_in, chain, stream, e0, e1, esq::stencil3d_tuning_default());
}
extern "C" int aniso3d_rkc_chain(void *user, const esq_rkc_chain *ch, size_t n, void *stream,
                                 void *e0, void *e1) {
    const User<AnisoFn> *u = (const User<AnisoFn> *)user;
    if (!u || n != Aniso::points(u->N)) return ESQ_EINVAL;
    return Aniso::rkc_chain(u->fn, u->N, ch, stream, e0, e1, esq::stencil3d_tuning_default());
}
