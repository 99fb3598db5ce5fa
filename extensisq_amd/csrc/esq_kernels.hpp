// esq_kernels.hpp -- gfx950 (CDNA4, wave64) device kernels of the explicit
// Runge-Kutta hot path.  Everything here is HBM-bandwidth-bound streaming
// arithmetic in fp64: 16-byte (double2) coalesced accesses, coefficients and row
// pointers in kernel arguments (scalar registers), no re-reads.
//
// Reference expressions replaced (extensisq v0.6.0):
//   k_lincomb         y + h*(K[:i].T @ A[i,:i])            common.py:355-356, 343
//   k_solution_error  y_new, scale, h*(K.T@E)/scale, norm  common.py:341-351, 57-66
//   k_error_norm      same, second pass of FSAL pairs      common.py:335-339
//   k_pre_error       BS5 early estimate                   bogacki.py:340-346
//   k_rkc_*           three-term Chebyshev recursion       sommeijer.py:289, 312-313, 218-220
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "esq_epilogue.hpp"
#include "esq_terms.hpp"

namespace esq {

// Where a reduction's result goes: a device double (input of the lock-step
// all-reduce) and/or a pinned host slot the host polls.  The value is stored
// first, then -- behind a system-scope release -- the sequence number of this
// reduction, so a host that reads seq == expected also reads the value.
struct ResultSink {
    double *dev;                       // may be nullptr
    double *host_value;                // pinned host memory, may be nullptr
    unsigned long long *host_seq;
    unsigned long long seq;
};
__device__ __forceinline__ void publish(const ResultSink &rs, double v) {
    if (rs.dev) *rs.dev = v;
    if (rs.host_value) {
        __hip_atomic_store(rs.host_value, v, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(rs.host_seq, rs.seq, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// ---------------------------------------------------------------------------
// out = base + h * sum_j c_j * v_j        (base may be nullptr -> 0)
// Evaluation order follows the reference: the weighted sum first (ascending j,
// fused multiply-add chain like a BLAS gemv kernel), then *h, then +base, the
// last two rounded separately as NumPy does.
// ---------------------------------------------------------------------------
// LDP: 0 plain loads, 1 nt loads of the K rows, 2 nt loads of everything
// STP: 0 plain store, 1 nt store
// `init` (optional): the value of the SAME FMA chain after its leading terms,
// stored by k_block_acc -- resuming from it is bit-identical to running the
// whole chain here.
template <int NT, int LDP = 0, int STP = 0>
__global__ __launch_bounds__(kBlock) void k_lincomb(
    double *__restrict__ out, const double *__restrict__ base,
    const double *__restrict__ init, Terms tm, double h, size_t n2) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2;
         i += stride) {
        double2 v[NT > 0 ? NT : 1];
#pragma unroll
        for (int j = 0; j < NT; ++j)
            v[j] = LDP >= 1 ? ld2_nt(tm.p[j], i) : ld2(tm.p[j], i);
        double2 yb = make_double2(0.0, 0.0);
        if (base) yb = LDP >= 2 ? ld2_nt(base, i) : ld2(base, i);
        double2 acc = make_double2(0.0, 0.0);
        if (init) acc = LDP >= 1 ? ld2_nt(init, i) : ld2(init, i);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            acc.x = fma(tm.c[j], v[j].x, acc.x);
            acc.y = fma(tm.c[j], v[j].y, acc.y);
        }
        double2 r;
        r.x = __dadd_rn(yb.x, __dmul_rn(h, acc.x));
        r.y = __dadd_rn(yb.y, __dmul_rn(h, acc.y));
        if (STP) st2_nt(out, i, r); else st2(out, i, r);
    }
}

// Single-workgroup form for the host-slab mode (small host-RHS problems): the
// same arithmetic; when the sweep is complete the workgroup itself bumps the
// pinned sequence number the host is spinning on -- no completion kernel, no
// stream synchronisation between the kernel and the host's memcpy of `out`.
template <int NT>
__global__ __launch_bounds__(kBlock) void k_lincomb_small(
    double *__restrict__ out, const double *__restrict__ base,
    const double *__restrict__ init, Terms tm, double h, size_t n2, ResultSink rs) {
    for (size_t i = threadIdx.x; i < n2; i += kBlock) {
        double2 v[NT > 0 ? NT : 1];
#pragma unroll
        for (int j = 0; j < NT; ++j) v[j] = ld2(tm.p[j], i);
        const double2 yb = base ? ld2(base, i) : make_double2(0.0, 0.0);
        double2 acc = init ? ld2(init, i) : make_double2(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            acc.x = fma(tm.c[j], v[j].x, acc.x);
            acc.y = fma(tm.c[j], v[j].y, acc.y);
        }
        st2(out, i, make_double2(__dadd_rn(yb.x, __dmul_rn(h, acc.x)),
                                 __dadd_rn(yb.y, __dmul_rn(h, acc.y))));
    }
    __threadfence_system();            // every thread's stores, before the flag
    __syncthreads();
    if (threadIdx.x == 0) publish(rs, 0.0);
}

// ---------------------------------------------------------------------------
// Blocked accumulation.  Stage i needs sum_{j<i} a_ij K_j; the leading columns
// j < J are needed by ALL later stages, so ONE pass over those K rows forms the
// leading part of every later stage's sum at once (NO outputs) and each later
// stage kernel resumes its chain from the stored value.  Each K row of the
// block is then read once instead of once per later stage (Pr8, J = 7: 93 -> 75
// words per element and step; Pr9, J = 8, 13: 154 -> 109).  The arithmetic is
// the SAME ascending-j FMA chain, cut at J: results are bit-identical.
// ---------------------------------------------------------------------------
struct BlockArgs {
    const double *p[kMaxTerms];        // K rows of the block (non-zero columns)
    double w[kMaxTerms][kMaxOut];      // a_ij for output stage o, 0 = skip
    const double *init[kMaxOut];       // previous-level partial sum or nullptr
    double *out[kMaxOut];
    // output 0 belongs to the boundary stage itself, whose sum is complete here:
    // if y != nullptr its ARGUMENT y + h*sum is written to out[0] instead of the
    // sum (same rounding as k_lincomb), and that stage needs no kernel of its own
    const double *y;
    double h;
};
template <int NT>
__global__ __launch_bounds__(kBlock) void k_block_acc(BlockArgs a, int no,
                                                      size_t n2) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2;
         i += stride) {
        double2 v[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) v[j] = ld2_nt(a.p[j], i);
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o) {
            if (o < no) {                              // uniform
                double2 acc = make_double2(0.0, 0.0);
                if (a.init[o]) acc = ld2_nt(a.init[o], i);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (a.w[j][o] != 0.0) {            // uniform (SGPR weights)
                        acc.x = fma(a.w[j][o], v[j].x, acc.x);
                        acc.y = fma(a.w[j][o], v[j].y, acc.y);
                    }
                }
                if (o == 0 && a.y) {
                    const double2 yb = ld2(a.y, i);
                    acc.x = __dadd_rn(yb.x, __dmul_rn(a.h, acc.x));
                    acc.y = __dadd_rn(yb.y, __dmul_rn(a.h, acc.y));
                    st2(a.out[o], i, acc);      // stage argument: read next, keep cached
                } else {
                    st2_nt(a.out[o], i, acc);   // partial sums: stream past the cache
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Non-FSAL fused pass: y_new = y + h*sum b_j K_j ; err = h*sum e_j K_j ;
// partial sum of |err/scale|^2.  Each K row is read ONCE for both sums.
// ---------------------------------------------------------------------------
template <int NT, bool CPLX, bool NTL = false>
__global__ __launch_bounds__(kBlock) void k_solution_error(
    double *__restrict__ ynew, const double *__restrict__ y, Terms2 tm,
    double h, const double *__restrict__ atol_vec, double atol_s, double rtol,
    size_t n2, size_t n_valid, double *__restrict__ partials) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    double local = 0.0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2;
         i += stride) {
        double2 v[NT > 0 ? NT : 1];
#pragma unroll
        for (int j = 0; j < NT; ++j)
            v[j] = NTL ? ld2_nt(tm.p[j], i) : ld2(tm.p[j], i);
        const double2 yy = ld2(y, i);
        // rows are the union of the two supports; a zero weight contributes
        // fma(0, v, s) == s (and 0*Inf = NaN, exactly like NumPy's gemv)
        double2 sb = make_double2(0.0, 0.0), se = make_double2(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            sb.x = fma(tm.b[j], v[j].x, sb.x);
            sb.y = fma(tm.b[j], v[j].y, sb.y);
            se.x = fma(tm.e[j], v[j].x, se.x);
            se.y = fma(tm.e[j], v[j].y, se.y);
        }
        double2 yn, er;
        yn.x = __dadd_rn(yy.x, __dmul_rn(h, sb.x));
        yn.y = __dadd_rn(yy.y, __dmul_rn(h, sb.y));
        er.x = __dmul_rn(h, se.x);
        er.y = __dmul_rn(h, se.y);
        st2(ynew, i, yn);
        local += ratio_sq<CPLX>(er, yy, yn, atol_vec, atol_s, rtol, i, n_valid);
    }
    block_partial(local, partials);
}

// FSAL second pass (and the public _estimate_error_norm): err from K rows,
// scale from y and y_new already in memory.
template <int NT, bool CPLX>
__global__ __launch_bounds__(kBlock) void k_error_norm(
    const double *__restrict__ y, const double *__restrict__ ynew, Terms tm,
    double h, const double *__restrict__ atol_vec, double atol_s, double rtol,
    size_t n2, size_t n_valid, double *__restrict__ partials) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    double local = 0.0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2;
         i += stride) {
        double2 v[NT > 0 ? NT : 1];
#pragma unroll
        for (int j = 0; j < NT; ++j) v[j] = ld2(tm.p[j], i);
        const double2 ya = ld2(y, i), yb = ld2(ynew, i);
        double2 se = make_double2(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            se.x = fma(tm.c[j], v[j].x, se.x);
            se.y = fma(tm.c[j], v[j].y, se.y);
        }
        double2 er;
        er.x = __dmul_rn(h, se.x);
        er.y = __dmul_rn(h, se.y);
        local += ratio_sq<CPLX>(er, ya, yb, atol_vec, atol_s, rtol, i, n_valid);
    }
    block_partial(local, partials);
}

// BS5 pre-error: y_pre stays in registers.
template <int NT, bool CPLX>
__global__ __launch_bounds__(kBlock) void k_pre_error(
    const double *__restrict__ y, Terms2 tm, double h,
    const double *__restrict__ atol_vec, double atol_s, double rtol, size_t n2,
    size_t n_valid, double *__restrict__ partials) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    double local = 0.0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2;
         i += stride) {
        double2 v[NT > 0 ? NT : 1];
#pragma unroll
        for (int j = 0; j < NT; ++j) v[j] = ld2(tm.p[j], i);
        const double2 yy = ld2(y, i);
        double2 sb = make_double2(0.0, 0.0), se = make_double2(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            sb.x = fma(tm.b[j], v[j].x, sb.x);
            sb.y = fma(tm.b[j], v[j].y, sb.y);
            se.x = fma(tm.e[j], v[j].x, se.x);
            se.y = fma(tm.e[j], v[j].y, se.y);
        }
        double2 yp, er;
        yp.x = __dadd_rn(yy.x, __dmul_rn(h, sb.x));
        yp.y = __dadd_rn(yy.y, __dmul_rn(h, sb.y));
        er.x = __dmul_rn(h, se.x);
        er.y = __dmul_rn(h, se.y);
        local += ratio_sq<CPLX>(er, yy, yp, atol_vec, atol_s, rtol, i, n_valid);
    }
    block_partial(local, partials);
}

}  // namespace esq
