// esq_rhs_diff3d_chain.hip -- the 3-D diffusion plugin's chain entry for the explicit pairs
// (a translation unit of its own: esq_rhs_diff3d.hpp)
#include "esq_rhs_diff3d.hpp"

extern "C" {

// D consecutive Runge-Kutta stages per launch (esq_rhs_chain_fn, esq_chain3d.hpp)
int esq_rhs_diff3d_chain(void *user, const double *y_in, const esq_chain *chain, size_t n,
                         void *stream, void *start_event, void *stop_event) {
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != DIFF3D || n != r->n || !chain) return ESQ_EINVAL;
    return Diff3d::chain(fn_of(r), r->N, y_in, chain, stream, start_event, stop_event,
                         tuning_of(r));
}

}  // extern "C"
