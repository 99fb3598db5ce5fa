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
// Dense-output coefficients in ONE pass over K:  Q_c = scale * sum_j P[j][c] K_j
// for c < np (ref common.py:363 `Q = K.T @ P`, :772 `Q * h`).  Every K row is
// read once; the (row, column) weights sit in the kernel arguments.
// ---------------------------------------------------------------------------
constexpr int kMaxCols = 8;
struct DenseArgs {
    const double *p[kMaxTerms];        // K rows with a non-zero P row
    double w[kMaxTerms][kMaxCols];     // P[j][c]
    double *q[kMaxCols];               // output columns
};
template <int NT>
__global__ __launch_bounds__(kBlock) void k_dense_q(DenseArgs a, int np,
                                                    double scale, size_t n2) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2;
         i += stride) {
        double2 v[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) v[j] = ld2_nt(a.p[j], i);
#pragma unroll
        for (int c = 0; c < kMaxCols; ++c) {
            if (c < np) {                      // uniform
                double2 acc = make_double2(0.0, 0.0);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc.x = fma(a.w[j][c], v[j].x, acc.x);
                    acc.y = fma(a.w[j][c], v[j].y, acc.y);
                }
                acc.x = __dmul_rn(acc.x, scale);
                acc.y = __dmul_rn(acc.y, scale);
                st2(a.q[c], i, acc);
            }
        }
    }
}
// Horner evaluation  out = y0 + x*(q0 + x*(q1 + ... x*q_{np-1}))   (common.py:775-785)
struct HornerArgs {
    const double *q[kMaxCols];
};
__global__ __launch_bounds__(kBlock) void k_horner(double *__restrict__ out,
                                                   const double *__restrict__ y0,
                                                   HornerArgs a, int np, double x,
                                                   size_t n2) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2;
         i += stride) {
        double2 acc = make_double2(0.0, 0.0);
#pragma unroll
        for (int c = kMaxCols - 1; c >= 0; --c) {
            if (c < np) {
                const double2 q = ld2(a.q[c], i);
                if (c == np - 1) {
                    acc.x = __dmul_rn(q.x, x);
                    acc.y = __dmul_rn(q.y, x);
                } else {
                    acc.x = __dmul_rn(__dadd_rn(acc.x, q.x), x);
                    acc.y = __dmul_rn(__dadd_rn(acc.y, q.y), x);
                }
            }
        }
        const double2 b = ld2(y0, i);
        acc.x = __dadd_rn(acc.x, b.x);
        acc.y = __dadd_rn(acc.y, b.y);
        st2(out, i, acc);
    }
}

// final deterministic sum of the per-block partials (one block of 1024):
// thread-strided partial sums, wave64 tree, 16 waves through LDS
__global__ __launch_bounds__(1024) void k_final_sum(
    const double *__restrict__ partials, int count, ResultSink rs) {
    __shared__ double lds[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += 1024) s += partials[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = lds[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) t += lds[w];
        publish(rs, t);
    }
}
// after the lock-step all-reduce: device double -> host slot
__global__ void k_publish(const double *__restrict__ src, ResultSink rs) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        rs.dev = nullptr;
        publish(rs, *src);
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

// ---------------------------------------------------------------------------
// Runge-Kutta-Chebyshev
// ---------------------------------------------------------------------------
// dst = yn + hmus*fn                                  sommeijer.py:289
__global__ __launch_bounds__(kBlock) void k_rkc_first(
    double *__restrict__ dst, const double *__restrict__ yn,
    const double *__restrict__ fn, double hmus, size_t n2) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2;
         i += stride) {
        const double2 a = ld2(yn, i), f = ld2(fn, i);
        double2 r;
        r.x = __dadd_rn(a.x, __dmul_rn(hmus, f.x));
        r.y = __dadd_rn(a.y, __dmul_rn(hmus, f.y));
        st2(dst, i, r);
    }
}
// dst = mu*yjm1 + nu*yjm2 + (1-mu-nu)*yn + hmus*(fy - ajm1*fn)   :312-313
// (left-to-right like the NumPy expression; every product rounded)
// dst may alias fy (the combination overwrites the derivative it consumed)
__global__ __launch_bounds__(kBlock) void k_rkc_stage(
    double *dst, const double *fy,
    const double *__restrict__ yjm1, const double *__restrict__ yjm2,
    const double *__restrict__ yn, const double *__restrict__ fn, double mu,
    double nu, double omn, double hmus, double ajm1, size_t n2) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2;
         i += stride) {
        const double2 a = ld2(yjm1, i), b = ld2(yjm2, i), c = ld2(yn, i);
        const double2 f = ld2(fy, i), g = ld2(fn, i);
        double2 r;
        r.x = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(mu, a.x), __dmul_rn(nu, b.x)),
                                  __dmul_rn(omn, c.x)),
                        __dmul_rn(hmus, __dsub_rn(f.x, __dmul_rn(ajm1, g.x))));
        r.y = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(mu, a.y), __dmul_rn(nu, b.y)),
                                  __dmul_rn(omn, c.y)),
                        __dmul_rn(hmus, __dsub_rn(f.y, __dmul_rn(ajm1, g.y))));
        st2(dst, i, r);
    }
}
// est = 0.8*(yn - y) + 0.4*h*(fn + fy); wt = atol + rtol*max(|y|,|yn|) :218-220
__global__ __launch_bounds__(kBlock) void k_rkc_error(
    const double *__restrict__ y, const double *__restrict__ yn,
    const double *__restrict__ fn, const double *__restrict__ fy, double h,
    const double *__restrict__ atol_vec, double atol_s, double rtol, size_t n2,
    size_t n_valid, double *__restrict__ partials) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    double local = 0.0;
    const double h04 = 0.4 * h;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2;
         i += stride) {
        const double2 a = ld2(y, i), b = ld2(yn, i), f = ld2(fn, i), g = ld2(fy, i);
        double2 er;
        er.x = __dadd_rn(__dmul_rn(0.8, __dsub_rn(b.x, a.x)),
                         __dmul_rn(h04, __dadd_rn(f.x, g.x)));
        er.y = __dadd_rn(__dmul_rn(0.8, __dsub_rn(b.y, a.y)),
                         __dmul_rn(h04, __dadd_rn(f.y, g.y)));
        local += ratio_sq<false>(er, a, b, atol_vec, atol_s, rtol, i, n_valid);
    }
    block_partial(local, partials);
}
// sum (x - y)^2  (y may be nullptr)
__global__ __launch_bounds__(kBlock) void k_sumsq(
    const double *__restrict__ x, const double *__restrict__ y, size_t n2,
    double *__restrict__ partials) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    double local = 0.0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2;
         i += stride) {
        double2 a = ld2(x, i);
        if (y) {
            const double2 b = ld2(y, i);
            a.x -= b.x;
            a.y -= b.y;
        }
        local += a.x * a.x + a.y * a.y;
    }
    block_partial(local, partials);
}
// dst = a + alpha*(b - c)   (a, c optional)
__global__ __launch_bounds__(kBlock) void k_axpbmc(
    double *__restrict__ dst, const double *__restrict__ a, double alpha,
    const double *__restrict__ b, const double *__restrict__ c, size_t n2) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2;
         i += stride) {
        double2 d = ld2(b, i);
        if (c) {
            const double2 cc = ld2(c, i);
            d.x = __dsub_rn(d.x, cc.x);
            d.y = __dsub_rn(d.y, cc.y);
        }
        d.x = __dmul_rn(d.x, alpha);
        d.y = __dmul_rn(d.y, alpha);
        if (a) {
            const double2 aa = ld2(a, i);
            d.x = __dadd_rn(aa.x, d.x);
            d.y = __dadd_rn(aa.y, d.y);
        }
        st2(dst, i, d);
    }
}
// sum |(a - b) / (atol + rtol*|w|)|^2                  sommeijer.py:154-155
__global__ __launch_bounds__(kBlock) void k_wdiff_sumsq(
    const double *__restrict__ a, const double *__restrict__ b,
    const double *__restrict__ w, const double *__restrict__ atol_vec,
    double atol_s, double rtol, size_t n2, size_t n_valid,
    double *__restrict__ partials) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    double local = 0.0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2;
         i += stride) {
        const double2 aa = ld2(a, i), bb = ld2(b, i), ww = ld2(w, i);
        double2 er;
        er.x = __dsub_rn(aa.x, bb.x);
        er.y = __dsub_rn(aa.y, bb.y);
        local += ratio_sq<false>(er, ww, ww, atol_vec, atol_s, rtol, i, n_valid);
    }
    block_partial(local, partials);
}

// ---------------------------------------------------------------------------
// Starting-step helpers (Watts' dhstrt as restated in common.py:519-763)
// ---------------------------------------------------------------------------
// partial sums of log10(atol + rtol*|y|) and partial minima of the same
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void k_log_etol(
    const double *__restrict__ y, const double *__restrict__ atol_vec,
    double atol_s, double rtol, size_t n2, size_t n_valid,
    double *__restrict__ part_sum, double *__restrict__ part_min) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    double s = 0.0, m = INFINITY;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2;
         i += stride) {
        const double2 v = ld2(y, i);
        if (CPLX) {
            if (i < n_valid) {
                const double at = atol_vec ? atol_vec[i] : atol_s;
                const double e = log10(at + rtol * hypot(v.x, v.y));
                s += e;
                m = fmin(m, e);
            }
        } else {
            if (2 * i < n_valid) {
                const double at = atol_vec ? atol_vec[2 * i] : atol_s;
                const double e = log10(at + rtol * fabs(v.x));
                s += e;
                m = fmin(m, e);
            }
            if (2 * i + 1 < n_valid) {
                const double at = atol_vec ? atol_vec[2 * i + 1] : atol_s;
                const double e = log10(at + rtol * fabs(v.y));
                s += e;
                m = fmin(m, e);
            }
        }
    }
    // sum -> part_sum, min -> part_min (same tree, two operators)
    __shared__ double lmin[kBlock / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmin(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0) lmin[threadIdx.x >> 6] = m;
    block_partial(s, part_sum);          // contains the __syncthreads()
    if (threadIdx.x == 0) {
        double t = lmin[0];
#pragma unroll
        for (int w = 1; w < kBlock / 64; ++w) t = fmin(t, lmin[w]);
        part_min[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(1024) void k_final_min(
    const double *__restrict__ partials, int count, ResultSink rs) {
    __shared__ double lds[16];
    double m = INFINITY;
    for (int i = threadIdx.x; i < count; i += 1024) m = fmin(m, partials[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmin(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = lds[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) t = fmin(t, lds[w]);
        publish(rs, t);
    }
}
// next perturbation direction (common.py:700-714):
//   dy  = where(src, src, fill);  spy = where(spy, spy, yp)
//   yp  = where(spy, copysign(dy, spy), dy)      (per real/imag component;
//   the `where` conditions test the whole complex number)
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void k_hs_select(
    double *__restrict__ yp, double *__restrict__ spy,
    const double *__restrict__ src, double fill, size_t n2, size_t n_valid) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2;
         i += stride) {
        double2 d = ld2(src, i), s = ld2(spy, i);
        const double2 y = ld2(yp, i);
        // the zero padding behind the n valid elements must stay zero (it is
        // summed by the norm kernels)
        const size_t first = CPLX ? i : 2 * i;
        if (first >= n_valid) continue;
        const bool second_valid = CPLX || first + 1 < n_valid;
        if (CPLX) {
            if (d.x == 0.0 && d.y == 0.0) { d.x = fill; d.y = 0.0; }
            if (s.x == 0.0 && s.y == 0.0) s = y;
            double2 r = d;
            if (s.x != 0.0 || s.y != 0.0) {
                r.x = copysign(d.x, s.x);
                r.y = copysign(d.y, s.y);
            }
            st2(spy, i, s);
            st2(yp, i, r);
        } else {
            if (d.x == 0.0) d.x = fill;
            if (d.y == 0.0) d.y = fill;
            if (s.x == 0.0) s.x = y.x;
            if (s.y == 0.0) s.y = y.y;
            double2 r;
            r.x = s.x != 0.0 ? copysign(d.x, s.x) : d.x;
            r.y = s.y != 0.0 ? copysign(d.y, s.y) : d.y;
            if (!second_valid) { r.y = 0.0; s.y = 0.0; }
            st2(spy, i, s);
            st2(yp, i, r);
        }
    }
}
// weighted dot product of RKSuite's stiffness check (common.py:413-415, 968,
// 1014):  sum a.b / wt^2,  wt = max(0.5*(|y1| + |y2|), floor).  A complex state
// is treated as the real vector (re, im) with the weight of the complex
// modulus on both parts (common.py:916-924).
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void k_wdot(
    const double *__restrict__ a, const double *__restrict__ b,
    const double *__restrict__ y1, const double *__restrict__ y2, double floor_,
    size_t n2, size_t n_valid, double *__restrict__ partials) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    double local = 0.0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2;
         i += stride) {
        const double2 va = ld2(a, i), vb = ld2(b, i), p = ld2(y1, i), q = ld2(y2, i);
        if (CPLX) {
            if (i < n_valid) {
                const double w = fmax(0.5 * (hypot(p.x, p.y) + hypot(q.x, q.y)), floor_);
                local += (va.x / w) * (vb.x / w) + (va.y / w) * (vb.y / w);
            }
        } else {
            if (2 * i < n_valid) {
                const double w = fmax(0.5 * (fabs(p.x) + fabs(q.x)), floor_);
                local += (va.x / w) * (vb.x / w);
            }
            if (2 * i + 1 < n_valid) {
                const double w = fmax(0.5 * (fabs(p.y) + fabs(q.y)), floor_);
                local += (va.y / w) * (vb.y / w);
            }
        }
    }
    block_partial(local, partials);
}
// dst[0..len) = value (re) / 0 (im) -- padding stays zero
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void k_fill(double *__restrict__ dst,
                                                 double value, double value_im,
                                                 size_t n2, size_t n_valid) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2;
         i += stride) {
        double2 v = make_double2(0.0, 0.0);
        if (CPLX) {
            if (i < n_valid) { v.x = value; v.y = value_im; }
        } else {
            if (2 * i < n_valid) v.x = value;
            if (2 * i + 1 < n_valid) v.y = value;
        }
        st2(dst, i, v);
    }
}

}  // namespace esq
