// esq_rhs_heat2d_chain4.hip -- the heat plugin's chain sweeps of depth 4..4
// (esq_rhs_heat2d.hpp: why a unit of their own)
#include "esq_rhs_heat2d.hpp"

namespace esq_rhs {
int heat2d_chain_d4(Rhs *r, const double *y_in, const esq_chain *chain, void *stream,
                     void *start_event, void *stop_event) {
    return heat2d_chain_range<4, 4>(r, y_in, chain, stream, start_event, stop_event);
}
}  // namespace esq_rhs
