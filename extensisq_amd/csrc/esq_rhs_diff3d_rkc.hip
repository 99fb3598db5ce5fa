// esq_rhs_diff3d_rkc.hip -- the 3-D diffusion plugin's Chebyshev chain entry (a translation
// unit of its own: esq_rhs_diff3d.hpp)
#include "esq_rhs_diff3d.hpp"

extern "C" {

// D Chebyshev stages per launch (esq_rhs_rkc_chain_fn)
int esq_rhs_diff3d_rkc_chain(void *user, const esq_rkc_chain *ch, size_t n, void *stream,
                             void *start_event, void *stop_event) {
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != DIFF3D || n != r->n || !ch) return ESQ_EINVAL;
    return Diff3d::rkc_chain(fn_of(r), r->N, ch, stream, start_event, stop_event, tuning_of(r));
}

}  // extern "C"
