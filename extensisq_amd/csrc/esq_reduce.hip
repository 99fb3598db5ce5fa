// esq_reduce.hip -- launchers of the stand-alone reducing kernels (solution +
// error, FSAL error norm, BS5 pre-error) and of the blocked accumulation.
#include "esq_internal.hpp"

namespace esqi {

template <int NT>
void launch_solerr_n(esq_ctx *c, const Terms2 &tm, double h, const Prof &p) {
    const double *av = c->atol_is_vec ? c->atolv : nullptr;
    if (c->cplx)
        hipExtLaunchKernelGGL((k_solution_error<NT, true>), dim3(c->grid_reduce), dim3(kBlock), 0, c->stream, p.start(), p.stop(), 0, c->ynew, c->y, tm, h, av,
                           c->atol_s, c->rtol, c->len_pad / 2, c->n, c->partials);
    else if (c->stage_policy >= 10)
        hipExtLaunchKernelGGL((k_solution_error<NT, false, true>), dim3(c->grid_reduce), dim3(kBlock), 0, c->stream, p.start(), p.stop(), 0, c->ynew, c->y, tm, h, av,
                           c->atol_s, c->rtol, c->len_pad / 2, c->n, c->partials);
    else
        hipExtLaunchKernelGGL((k_solution_error<NT, false>), dim3(c->grid_reduce), dim3(kBlock), 0, c->stream, p.start(), p.stop(), 0, c->ynew, c->y, tm, h, av,
                           c->atol_s, c->rtol, c->len_pad / 2, c->n, c->partials);
}
template <int NT>
void launch_errnorm_n(esq_ctx *c, const Terms &tm, double h, const Prof &p) {
    const double *av = c->atol_is_vec ? c->atolv : nullptr;
    if (c->cplx)
        hipExtLaunchKernelGGL((k_error_norm<NT, true>), dim3(c->grid_reduce), dim3(kBlock), 0, c->stream, p.start(), p.stop(), 0, c->y, c->ynew, tm, h, av,
                           c->atol_s, c->rtol, c->len_pad / 2, c->n, c->partials);
    else
        hipExtLaunchKernelGGL((k_error_norm<NT, false>), dim3(c->grid_reduce), dim3(kBlock), 0, c->stream, p.start(), p.stop(), 0, c->y, c->ynew, tm, h, av,
                           c->atol_s, c->rtol, c->len_pad / 2, c->n, c->partials);
}
template <int NT>
void launch_preerr_n(esq_ctx *c, const Terms2 &tm, double h, const Prof &p) {
    const double *av = c->atol_is_vec ? c->atolv : nullptr;
    if (c->cplx)
        hipExtLaunchKernelGGL((k_pre_error<NT, true>), dim3(c->grid_reduce), dim3(kBlock), 0, c->stream, p.start(), p.stop(), 0, c->y, tm, h, av, c->atol_s,
                           c->rtol, c->len_pad / 2, c->n, c->partials);
    else
        hipExtLaunchKernelGGL((k_pre_error<NT, false>), dim3(c->grid_reduce), dim3(kBlock), 0, c->stream, p.start(), p.stop(), 0, c->y, tm, h, av, c->atol_s,
                           c->rtol, c->len_pad / 2, c->n, c->partials);
}
#define DISPATCH_1_20(FN, nt, ...)                                              \
    switch (nt) {                                                               \
        case 1: FN<1>(__VA_ARGS__); break;   case 2: FN<2>(__VA_ARGS__); break;   \
        case 3: FN<3>(__VA_ARGS__); break;   case 4: FN<4>(__VA_ARGS__); break;   \
        case 5: FN<5>(__VA_ARGS__); break;   case 6: FN<6>(__VA_ARGS__); break;   \
        case 7: FN<7>(__VA_ARGS__); break;   case 8: FN<8>(__VA_ARGS__); break;   \
        case 9: FN<9>(__VA_ARGS__); break;   case 10: FN<10>(__VA_ARGS__); break; \
        case 11: FN<11>(__VA_ARGS__); break; case 12: FN<12>(__VA_ARGS__); break; \
        case 13: FN<13>(__VA_ARGS__); break; case 14: FN<14>(__VA_ARGS__); break; \
        case 15: FN<15>(__VA_ARGS__); break; case 16: FN<16>(__VA_ARGS__); break; \
        case 17: FN<17>(__VA_ARGS__); break; case 18: FN<18>(__VA_ARGS__); break; \
        case 19: FN<19>(__VA_ARGS__); break; case 20: FN<20>(__VA_ARGS__); break; \
        default: return fail(c, ESQ_EINVAL, "bad term count %d", nt);           \
    }

template <int NT>
void launch_block_n(esq_ctx *c, const BlockArgs &a, int no, const Prof &p) {
    hipExtLaunchKernelGGL(k_block_acc<NT>, dim3(c->grid_block), dim3(kBlock), 0,
                          c->stream, p.start(), p.stop(), 0, a, no,
                          c->len_pad / 2);
}

int launch_solerr(esq_ctx *c, const Terms2 &tm, int nt, double h, const Prof &p) {
    if (c->detached) return 0;                  // host-side dry run
    DISPATCH_1_20(launch_solerr_n, nt, c, tm, h, p)
    HIPCHK(c, hipGetLastError());
    return 0;
}
int launch_errnorm(esq_ctx *c, const Terms &tm, int nt, double h, const Prof &p) {
    if (c->detached) return 0;                  // host-side dry run
    DISPATCH_1_20(launch_errnorm_n, nt, c, tm, h, p)
    HIPCHK(c, hipGetLastError());
    return 0;
}
int launch_preerr(esq_ctx *c, const Terms2 &tm, int nt, double h, const Prof &p) {
    if (c->detached) return 0;                  // host-side dry run
    DISPATCH_1_20(launch_preerr_n, nt, c, tm, h, p)
    HIPCHK(c, hipGetLastError());
    return 0;
}
int launch_block(esq_ctx *c, const BlockArgs &a, int nt, int no, const Prof &p) {
    if (c->detached) return 0;                  // host-side dry run
    DISPATCH_1_20(launch_block_n, nt, c, a, no, p)
    HIPCHK(c, hipGetLastError());
    return 0;
}

}  // namespace esqi
