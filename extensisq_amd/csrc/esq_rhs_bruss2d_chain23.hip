// esq_rhs_bruss2d_chain23.hip -- the Brusselator's chain sweeps of depth 2..3
// (esq_rhs_bruss2d.hpp: why a unit of their own)
#include "esq_rhs_bruss2d.hpp"

namespace esq_rhs {
int bruss2d_chain_d23(Rhs *r, const double *y_in, const esq_chain *chain, void *stream,
                      void *start_event, void *stop_event) {
    return bruss2d_chain_range<2, 3>(r, y_in, chain, stream, start_event, stop_event);
}
}  // namespace esq_rhs
