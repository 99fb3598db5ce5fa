// esq_rhs_common.hpp -- shared by the built-in device RHS plugins (the synthetic
// workloads named in BASELINE.json `configs`).  They stand where
// the user's Python callable `fun(t, y)` stands in the reference
// (common.py:356); their NumPy twins, used by the tests, are in
// oracle/problems.py and use the same operation order (the library is built
// with -ffp-contract=off), so the two agree bit for bit.
//
// All kernels are stencil sweeps: one HBM read + one HBM write per element is
// the floor ("RHS-min" in BASELINE.md); neighbour reuse is served by L1/L2.
// Workgroups are dealt round-robin over the 8 XCDs, so block b is remapped to
// a contiguous band of rows per XCD (blockIdx % 8 = XCD label): the up/down
// neighbour rows then hit the SAME XCD's L2 instead of being fetched twice.
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/extensisq_amd.h"
#include "esq_epilogue.hpp"
#include "esq_plugin.hpp"
#include "esq_stencil2d.hpp"
#include "esq_terms.hpp"

namespace esq_rhs {

constexpr int kBlock = 256;
using esq::kXcd;
using esq::band_block;
using esq::SrcPlain;
using esq::SrcAxpy;
using esq::first_stage_ok;
using esq::kFirstStage;
using esq::axpy_of;

enum Kind { DIAG = 1, HEAT2D = 2, BRUSS2D = 3, DIFF3D = 4, CDIAG = 5 };

struct Rhs {
    int kind;
    int N;
    int device;
    double alpha, a, b;
    double amp;
    double *lam_dev;
    size_t n;
    // tuning / test knobs of THIS plugin object (esq_rhs_set_options; defaults from the
    // process environment when the object is made): RKC_FORCE chain sweeps on grids of
    // any size, RKC_PLANES planes per tile of the 3-D chain sweeps (0 = chosen by
    // geo_rkc3d), DIFF3D_R the 3-D plugin's marching sweep, CHAIN_ROWS the tile height
    // of the 2-D chain sweeps
    int rkc_force, rkc_planes, diff3d_r;
    esq::ChainTuning tune;
};

// the plugin-level switches (esq_options.hpp) of a built-in plugin object
inline void apply_options(Rhs *r, const esq::Options &o) {
    r->rkc_force = o.int_or("RKC_FORCE", 0);
    r->rkc_planes = o.int_or("RKC_PLANES", 0);
    r->diff3d_r = o.int_or("DIFF3D_R", 0);
    r->tune = esq::ChainTuning{};
    if (const char *rows = o.get("CHAIN_ROWS")) {
        r->tune.rows_set = true;
        r->tune.rows = atoi(rows);
    }
}

inline int make(void **out, Rhs proto) {
    if (!out) return ESQ_EINVAL;
    Rhs *r = (Rhs *)malloc(sizeof(Rhs));
    if (!r) return ESQ_ENOMEM;
    *r = proto;
    apply_options(r, esq::Options{});      // the process defaults (ESQ_<KEY>)
    *out = r;
    return 0;
}

using esq::v2d;
using RkcEpi = esq::EpiRkc;

inline RkcEpi make_epi(const double *yjm2, const double *yn, const double *fn,
                       double mu, double nu, double omn, double hmus, double ajm1,
                       double *out) {
    RkcEpi e{};
    e.yjm2 = yjm2; e.yn = yn; e.fn = fn; e.out = out;
    e.mu = mu; e.nu = nu; e.omn = omn; e.hmus = hmus; e.ajm1 = ajm1;
    return e;
}

}  // namespace esq_rhs
