// esq_lincomb.hip -- launchers of k_lincomb<NT, load policy, store policy>
//     out = base + h * (init + sum_j c_j * v_j)       common.py:355-356, 343
// (host-RHS mode, plugins without a fused entry, dense-output stages, the error
// vector) and of its single-workgroup form for the host-slab mode.
#include "esq_internal.hpp"

namespace esqi {

template <int NT, int LDP, int STP>
void launch_lincomb_p(esq_ctx *c, double *out, const double *base,
                      const double *init, const Terms &tm, double h,
                      const Prof *p) {
    hipExtLaunchKernelGGL((k_lincomb<NT, LDP, STP>), dim3(c->grid_stream),
                          dim3(kBlock), 0, c->stream, p ? p->start() : nullptr,
                          p ? p->stop() : nullptr, 0, out, base, init, tm, h,
                          c->len_pad / 2);
}
template <int NT>
void launch_lincomb_n(esq_ctx *c, double *out, const double *base,
                      const double *init, const Terms &tm, double h,
                      const Prof *p) {
    switch (c->stage_policy) {      // ESQ_STAGE_POLICY = <load><store>
        case 1:  launch_lincomb_p<NT, 0, 1>(c, out, base, init, tm, h, p); break;
        case 10: launch_lincomb_p<NT, 1, 0>(c, out, base, init, tm, h, p); break;
        case 11: launch_lincomb_p<NT, 1, 1>(c, out, base, init, tm, h, p); break;
        case 20: launch_lincomb_p<NT, 2, 0>(c, out, base, init, tm, h, p); break;
        case 21: launch_lincomb_p<NT, 2, 1>(c, out, base, init, tm, h, p); break;
        default: launch_lincomb_p<NT, 0, 0>(c, out, base, init, tm, h, p); break;
    }
}
template <int NT>
void launch_lincomb_small(esq_ctx *c, double *out, const double *base,
                          const double *init, const Terms &tm, double h,
                          const Prof *p) {
    const ResultSink rs = next_sink(c, true);
    hipExtLaunchKernelGGL((k_lincomb_small<NT>), dim3(1), dim3(kBlock), 0, c->stream,
                          p ? p->start() : nullptr, p ? p->stop() : nullptr, 0, out,
                          base, init, tm, h, c->len_pad / 2, rs);
    c->self_seq = rs.seq;
    c->self_valid = true;
}
int launch_lincomb(esq_ctx *c, double *out, const double *base, const Terms &tm,
                   int nt, double h, const Prof *p, const double *init) {
    if (c->detached) return nt <= kMaxTerms ? 0 : ESQ_EINVAL;   // host-side dry run
    if (c->host_slab && c->len_pad / 2 <= 4096) {
        // small host-RHS problem: one workgroup, completion signalled in-kernel
#define CASE(N) case N: launch_lincomb_small<N>(c, out, base, init, tm, h, p); break;
        switch (nt) {
            CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
            CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16)
            CASE(17) CASE(18) CASE(19) CASE(20)
            default: return fail(c, ESQ_EINVAL, "too many terms: %d", nt);
        }
#undef CASE
        HIPCHK(c, hipGetLastError());
        return 0;
    }
#define CASE(N) case N: launch_lincomb_n<N>(c, out, base, init, tm, h, p); break;
    switch (nt) {
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
        CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16)
        CASE(17) CASE(18) CASE(19) CASE(20)
        default: return fail(c, ESQ_EINVAL, "too many terms: %d", nt);
    }
#undef CASE
    HIPCHK(c, hipGetLastError());
    return 0;
}

}  // namespace esqi
