// esq_rhs_diff3d.hpp -- the pointwise functor of the built-in 3-D diffusion plugin, shared
// by its translation units (esq_rhs_diff3d.hip: sweeps; esq_rhs_diff3d_chain.hip: the chain
// sweeps of the explicit pairs; esq_rhs_diff3d_rkc.hip: the Chebyshev chain sweeps -- three
// units so that the template instantiations compile in parallel).
#pragma once
#include "esq_rhs_common.hpp"
#include "esq_stencil3d.hpp"

using namespace esq_rhs;

namespace {

// u_t = (N + 1)^2 * (sum of the six neighbours - 6 u): the expression of the NumPy twin
// (oracle/problems.py: diff3d_rhs), operation for operation
struct Diff3dFn {
    static constexpr bool kZeroOutside = true, kAutonomous = true;
    double c;
    __device__ __forceinline__ double ghost(int, int, int, int, int, double, double) const {
        return 0.0;
    }
    __device__ __forceinline__ void eval(const esq::Nb3 (&nb)[1], int, int, int, double,
                                         double (&f)[1]) const {
        f[0] = c * ((((nb[0].below + nb[0].above) + (nb[0].up + nb[0].dn)) +
                     (nb[0].lf + nb[0].rt)) - 6.0 * nb[0].c);
    }
};
using Diff3d = esq::Stencil3D<1, Diff3dFn>;

inline Diff3dFn fn_of(const Rhs *r) {
    return Diff3dFn{(double)(r->N + 1) * (double)(r->N + 1)};
}
inline esq::Stencil3dTuning tuning_of(const Rhs *r) {
    esq::Stencil3dTuning t;
    t.force = r->rkc_force; t.planes = r->rkc_planes;
    t.march_r = r->diff3d_r;
    return t;
}

}  // namespace
