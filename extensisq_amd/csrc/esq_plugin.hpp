// esq_plugin.hpp -- host-side helper for authors of device RHS plugins with a
// fused entry (esq_rhs_fused_fn, include/extensisq_amd.h).
//
// The library describes the epilogue it wants as a plain C struct
// (`esq_epilogue`); the plugin's sweep kernel is a template over the matching
// device-side type of esq_epilogue.hpp.  `esq::dispatch_epilogue` converts one
// into the other and hands it to a generic callable that launches the kernel:
//
//     template <class Epi> __global__ void my_sweep(const double* y, double* f, Epi epi, ...);
//
//     extern "C" int my_rhs_fused(void* user, double t, const double* y, double* f,
//                                 const esq_epilogue* epi, size_t n, void* stream,
//                                 void* start_event, void* stop_event) {
//         const unsigned grid = ...;
//         if (esq::epilogue_reduces(epi)) {            // one partial per workgroup
//             if ((int)grid > epi->partials_cap) return ESQ_ENOTSUP;
//             *epi->partials_used = (int)grid;
//         }
//         const int rc = esq::dispatch_epilogue(epi, [&](auto ep) {
//             hipExtLaunchKernelGGL((my_sweep<decltype(ep)>), dim3(grid), dim3(256), 0,
//                                   (hipStream_t)stream, (hipEvent_t)start_event,
//                                   (hipEvent_t)stop_event, 0, y, f, ep, ...);
//         });
//         return rc ? rc : (int)hipGetLastError();
//     }
//
// Row counts beyond the instantiated ranges (stage / solution rows <= 16, block
// rows <= 8 besides the fresh column, FSAL error rows <= 12) return ESQ_ENOTSUP:
// the library then falls back to esq_rhs_fn + its own kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/extensisq_amd.h"
#include "esq_epilogue.hpp"
#include "esq_chain.hpp"

namespace esq {

inline bool epilogue_reduces(const esq_epilogue *epi) {
    return epi->kind == ESQ_EPI_SOLERR || epi->kind == ESQ_EPI_ERRNORM ||
           epi->kind == ESQ_EPI_RKCERR;
}
inline RedArgs red_of(const esq_epilogue *e) {
    RedArgs r;
    r.atol_vec = e->atol_vec;
    r.atol_s = e->atol_s;
    r.rtol = e->rtol;
    r.n_valid = e->n_valid;
    r.partials = e->partials;
    return r;
}
template <int NT>
EpiStage<NT> make_stage(const esq_epilogue *e) {
    EpiStage<NT> s;
    for (int j = 0; j < kMaxTerms; ++j) {
        s.tm.p[j] = j < e->nt ? e->rows[j] : nullptr;
        s.tm.c[j] = j < e->nt ? e->c[j] : 0.0;
    }
    s.init = e->init; s.y = e->y; s.out = e->out;
    s.c_self = e->c_self; s.h = e->h; s.f_nt = e->f_store_nt;
    s.red = red_of(e);
    return s;
}
template <int NT>
EpiBlock<NT> make_block(const esq_epilogue *e) {
    EpiBlock<NT> s;
    for (int j = 0; j < kMaxTerms; ++j) {
        s.p[j] = j < e->nt ? e->rows[j] : nullptr;
        for (int o = 0; o < kMaxOut; ++o)
            s.w[j][o] = (j < e->nt && o < e->no) ? e->w[j][o] : 0.0;
    }
    for (int o = 0; o < kMaxOut; ++o) {
        s.w_self[o] = o < e->no ? e->w_self[o] : 0.0;
        s.init[o] = o < e->no ? e->init_o[o] : nullptr;
        s.out[o] = o < e->no ? e->out_o[o] : nullptr;
    }
    s.y = e->y; s.h = e->h; s.no = e->no; s.f_nt = e->f_store_nt;
    s.red = red_of(e);
    return s;
}
template <int NT, bool CPLX = false>
EpiSolErr<NT, CPLX> make_solerr(const esq_epilogue *e) {
    EpiSolErr<NT, CPLX> s;
    for (int j = 0; j < kMaxTerms; ++j) {
        s.tm.p[j] = j < e->nt ? e->rows[j] : nullptr;
        s.tm.b[j] = j < e->nt ? e->c[j] : 0.0;
        s.tm.e[j] = j < e->nt ? e->e[j] : 0.0;
    }
    s.b_self = e->c_self; s.e_self = e->e_self;
    s.y = e->y; s.ynew = e->out; s.h = e->h; s.f_nt = e->f_store_nt;
    s.red = red_of(e);
    return s;
}
template <int NT, bool CPLX = false>
EpiErrNorm<NT, CPLX> make_errnorm(const esq_epilogue *e) {
    EpiErrNorm<NT, CPLX> s;
    for (int j = 0; j < kMaxTerms; ++j) {
        s.tm.p[j] = j < e->nt ? e->rows[j] : nullptr;
        s.tm.c[j] = j < e->nt ? e->e[j] : 0.0;
    }
    s.e_self = e->e_self; s.y = e->y; s.h = e->h; s.f_nt = e->f_store_nt;
    s.red = red_of(e);
    return s;
}

// (kind, nt) -> launch(device-side epilogue object).  Returns 0 after the call,
// ESQ_EINVAL for an inconsistent description, ESQ_ENOTSUP outside the
// instantiated ranges.  epi->dry_run: the same answer without the call (the
// library's side-effect-free query, ESQ_FUSE_QUERY).
#define ESQ_EPI_CASE_(MAKE, K) case K: if (!epi->dry_run) launch(MAKE<K>(epi)); return 0;
#define ESQ_EPI_CASES_0_8_(MAKE)                                                 \
    ESQ_EPI_CASE_(MAKE, 0) ESQ_EPI_CASE_(MAKE, 1) ESQ_EPI_CASE_(MAKE, 2)         \
    ESQ_EPI_CASE_(MAKE, 3) ESQ_EPI_CASE_(MAKE, 4) ESQ_EPI_CASE_(MAKE, 5)         \
    ESQ_EPI_CASE_(MAKE, 6) ESQ_EPI_CASE_(MAKE, 7) ESQ_EPI_CASE_(MAKE, 8)
#define ESQ_EPI_CASES_9_12_(MAKE)                                                \
    ESQ_EPI_CASE_(MAKE, 9) ESQ_EPI_CASE_(MAKE, 10) ESQ_EPI_CASE_(MAKE, 11)       \
    ESQ_EPI_CASE_(MAKE, 12)
#define ESQ_EPI_CASES_13_16_(MAKE)                                               \
    ESQ_EPI_CASE_(MAKE, 13) ESQ_EPI_CASE_(MAKE, 14) ESQ_EPI_CASE_(MAKE, 15)      \
    ESQ_EPI_CASE_(MAKE, 16)
inline EpiRkcErr make_rkcerr(const esq_epilogue *e) {
    EpiRkcErr s;
    s.yn = e->rows[0]; s.fn = e->rows[1];
    s.h04 = 0.4 * e->h;
    s.f_nt = e->f_store_nt;
    s.red = red_of(e);
    return s;
}
template <int NT> EpiSolErr<NT, true> make_solerr_c(const esq_epilogue *e) {
    return make_solerr<NT, true>(e);
}
template <int NT> EpiErrNorm<NT, true> make_errnorm_c(const esq_epilogue *e) {
    return make_errnorm<NT, true>(e);
}
// CPLX_OK: the plugin's state may be complex (interleaved re, im): the reducing
// epilogues are then instantiated for complex weights too.  Real-only plugins
// (the default) answer ESQ_ENOTSUP to a complex reduction and the library falls
// back to its stand-alone kernels.
template <bool CPLX_OK = false, class Launch>
int dispatch_epilogue(const esq_epilogue *epi, Launch &&launch) {
    if (!epi || epi->nt < 0) return ESQ_EINVAL;
    if (epi->kind == ESQ_EPI_RKCERR) {
        if (epi->is_complex || !epi->rows[0] || !epi->rows[1] || !epi->partials)
            return ESQ_EINVAL;
        if (!epi->dry_run) launch(make_rkcerr(epi));
        return 0;
    }
    if (epi->is_complex && epilogue_reduces(epi)) {
        if constexpr (CPLX_OK) {
            if (!epi->y || !epi->partials) return ESQ_EINVAL;
            if (epi->kind == ESQ_EPI_SOLERR) {
                if (!epi->out) return ESQ_EINVAL;
                switch (epi->nt) {
                    ESQ_EPI_CASES_0_8_(make_solerr_c) ESQ_EPI_CASES_9_12_(make_solerr_c)
                    ESQ_EPI_CASES_13_16_(make_solerr_c)
                    default: return ESQ_ENOTSUP;
                }
            }
            switch (epi->nt) {
                ESQ_EPI_CASES_0_8_(make_errnorm_c) ESQ_EPI_CASES_9_12_(make_errnorm_c)
                default: return ESQ_ENOTSUP;
            }
        } else {
            return ESQ_ENOTSUP;
        }
    }
    switch (epi->kind) {
        case ESQ_EPI_STAGE:
            if (!epi->out) return ESQ_EINVAL;
            switch (epi->nt) {
                ESQ_EPI_CASES_0_8_(make_stage) ESQ_EPI_CASES_9_12_(make_stage)
                ESQ_EPI_CASES_13_16_(make_stage)
                default: return ESQ_ENOTSUP;
            }
        case ESQ_EPI_BLOCK:
            if (epi->no < 1 || epi->no > ESQ_EPI_MAX_OUT) return ESQ_EINVAL;
            switch (epi->nt) {
                ESQ_EPI_CASES_0_8_(make_block)
                default: return ESQ_ENOTSUP;
            }
        case ESQ_EPI_SOLERR:
            if (!epi->out || !epi->y || !epi->partials) return ESQ_EINVAL;
            switch (epi->nt) {
                ESQ_EPI_CASES_0_8_(make_solerr) ESQ_EPI_CASES_9_12_(make_solerr)
                ESQ_EPI_CASES_13_16_(make_solerr)
                default: return ESQ_ENOTSUP;
            }
        case ESQ_EPI_ERRNORM:
            if (!epi->y || !epi->partials) return ESQ_EINVAL;
            switch (epi->nt) {
                ESQ_EPI_CASES_0_8_(make_errnorm) ESQ_EPI_CASES_9_12_(make_errnorm)
                default: return ESQ_ENOTSUP;
            }
        default: return ESQ_ENOTSUP;
    }
}

// ---- chain entry (esq_rhs_chain_fn): esq_chain -> ChainArgs<D, NU>
template <int D, int NU>
ChainArgs<D, NU> make_chain_args(const esq_chain *c) {
    ChainArgs<D, NU> a;
    for (int j = 0; j < ChainArgs<D, NU>::NUa; ++j) {
        const bool on = j < NU && j < c->nu;
        a.rows[j] = on ? c->rows[j] : nullptr;
        a.eu[j] = on ? c->eu[j] : 0.0;
        for (int e = 0; e < D; ++e) a.cu[e][j] = on ? c->cu[e][j] : 0.0;
    }
    // (umask / kmask of the descriptor: who takes part in which target.  These sweeps do
    // not test them -- a row that takes no part has weight +0.0, esq_chain.hpp)
    for (int e = 0; e < D; ++e) {
        a.ek[e] = c->ek[e];
        a.fk[e] = c->f_out[e];
        for (int k = 0; k < D; ++k) a.ck[e][k] = c->ck[e][k];
    }
    for (int j = 0; j < ChainArgs<D, NU>::NUa; ++j)
        a.c0[j] = (c->from_rows && j < NU && j < c->nu) ? c->c0[j] : 0.0;
    a.y = c->y; a.h = c->h; a.out = c->out; a.f_nt = c->f_store_nt;
    a.ld_nt = (unsigned)c->load_nt;
    a.red.atol_vec = c->atol_vec; a.red.atol_s = c->atol_s; a.red.rtol = c->rtol;
    a.red.n_valid = c->n_valid; a.red.partials = c->partials;
    return a;
}
// (depth, kind_last, nu) -> launch(ChainArgs<D, NU>, integral_constant<kind_last>,
// bool_constant<from_rows>).
// Instantiated: depth 2, 3 and 4 with up to 9 memory rows, 5 and 6 with up to 6.
//   from-rows form: either kind with 1..3 memory rows (the first chain of a step); depth
//   >= 3 with 4..6 rows (a chain in the MIDDLE of a step: the chain before it then leaves
//   its last target unwritten); depth >= 3, solution/error kind, 7+ rows (the chain that
//   ends a step)
//   ESQ_EPI_ERRNORM (FSAL pairs: the end-point stage and the error norm inside the
//   chain): up to 6 memory rows; plain up to depth 4, from rows at every depth (the
//   whole step of a six-stage pair from K[0]; the two sweeps that end a BS5 step)
// MAXD: deepest chain the caller's kernel is instantiated for (compile time grows
// with every depth); MIND: the shallowest (a plugin may spread its depths over several
// translation units that compile in parallel: esq_rhs_bruss2d_chain*.hip)
template <int DD, int K, class Launch>
int dispatch_chain_case(const esq_chain *c, Launch &&launch) {
    using Stage = std::integral_constant<int, ESQ_EPI_STAGE>;
    using SolErr = std::integral_constant<int, ESQ_EPI_SOLERR>;
    using ErrNorm = std::integral_constant<int, ESQ_EPI_ERRNORM>;
    if (c->kind_last == ESQ_EPI_ERRNORM) {
        if constexpr (K <= 6) {
            if (c->from_rows) {
                if constexpr (K >= 1) {
                    if (!c->dry_run) launch(make_chain_args<DD, K>(c), ErrNorm{}, std::true_type{});
                    return 0;
                }
                return ESQ_ENOTSUP;
            }
            if constexpr (DD <= 4) {
                if (!c->dry_run) launch(make_chain_args<DD, K>(c), ErrNorm{}, std::false_type{});
                return 0;
            }
        }
        return ESQ_ENOTSUP;
    }
    if (c->from_rows) {
        if constexpr ((DD >= 3 && K >= 4 && K <= 6) || (K >= 1 && K <= 3)) {
            if (c->dry_run) return 0;
            if (c->kind_last == ESQ_EPI_STAGE)
                launch(make_chain_args<DD, K>(c), Stage{}, std::true_type{});
            else
                launch(make_chain_args<DD, K>(c), SolErr{}, std::true_type{});
            return 0;
        } else if constexpr (DD >= 3 && K >= 7) {
            if (c->kind_last != ESQ_EPI_SOLERR) return ESQ_ENOTSUP;
            if (!c->dry_run) launch(make_chain_args<DD, K>(c), SolErr{}, std::true_type{});
            return 0;
        }
        return ESQ_ENOTSUP;
    }
    if (c->dry_run) return 0;
    if (c->kind_last == ESQ_EPI_STAGE)
        launch(make_chain_args<DD, K>(c), Stage{}, std::false_type{});
    else
        launch(make_chain_args<DD, K>(c), SolErr{}, std::false_type{});
    return 0;
}
template <int DD, int MAXK, class Launch>
int dispatch_chain_rows(const esq_chain *c, Launch &&launch) {
#define ESQ_CHAIN_CASE_(K)                                         \
    case K:                                                        \
        if constexpr (K <= MAXK) return dispatch_chain_case<DD, K>(c, launch); \
        return ESQ_ENOTSUP;
    switch (c->nu) {
        ESQ_CHAIN_CASE_(0) ESQ_CHAIN_CASE_(1) ESQ_CHAIN_CASE_(2) ESQ_CHAIN_CASE_(3)
        ESQ_CHAIN_CASE_(4) ESQ_CHAIN_CASE_(5) ESQ_CHAIN_CASE_(6) ESQ_CHAIN_CASE_(7)
        ESQ_CHAIN_CASE_(8) ESQ_CHAIN_CASE_(9)
        default: return ESQ_ENOTSUP;
    }
#undef ESQ_CHAIN_CASE_
}
template <int MAXD = 4, int MIND = 2, class Launch>
int dispatch_chain(const esq_chain *c, Launch &&launch) {
    if (!c || c->nu < 0) return ESQ_EINVAL;
    // (ESQ_EPI_SOLERR with out == NULL: the target is the scale's partner only)
    if (!c->out && c->kind_last == ESQ_EPI_ERRNORM) return ESQ_EINVAL;
    if (c->from_rows && !c->y) return ESQ_EINVAL;
    if ((c->kind_last == ESQ_EPI_SOLERR || c->kind_last == ESQ_EPI_ERRNORM) &&
        (!c->partials || !c->y))
        return ESQ_EINVAL;
    if (c->kind_last != ESQ_EPI_STAGE && c->kind_last != ESQ_EPI_SOLERR &&
        c->kind_last != ESQ_EPI_ERRNORM)
        return ESQ_ENOTSUP;
    switch (c->depth) {
        case 2:
            if constexpr (MIND <= 2 && MAXD >= 2) return dispatch_chain_rows<2, 9>(c, launch);
            return ESQ_ENOTSUP;
        case 3:
            if constexpr (MIND <= 3 && MAXD >= 3) return dispatch_chain_rows<3, 9>(c, launch);
            return ESQ_ENOTSUP;
        case 4:
            if constexpr (MIND <= 4 && MAXD >= 4) return dispatch_chain_rows<4, 9>(c, launch);
            return ESQ_ENOTSUP;
        case 5:
            if constexpr (MIND <= 5 && MAXD >= 5) return dispatch_chain_rows<5, 6>(c, launch);
            return ESQ_ENOTSUP;
        case 6:
            if constexpr (MIND <= 6 && MAXD >= 6) return dispatch_chain_rows<6, 6>(c, launch);
            return ESQ_ENOTSUP;
        default: return ESQ_ENOTSUP;
    }
}

#undef ESQ_EPI_CASE_
#undef ESQ_EPI_CASES_0_8_
#undef ESQ_EPI_CASES_9_12_
#undef ESQ_EPI_CASES_13_16_

}  // namespace esq
