// esq_aux_kernels.hpp -- gfx950 kernels off the per-step hot loop: dense output,
// Runge-Kutta-Chebyshev stages, norms / axpys of the power iteration and the
// starting-step estimate.  Same conventions as esq_kernels.hpp (fp64, 16-byte
// coalesced accesses, grid-stride, deterministic block reductions).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "esq_epilogue.hpp"
#include "esq_terms.hpp"

namespace esq {

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
