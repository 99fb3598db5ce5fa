// esq_epilogue.hpp -- device side of the fused RHS entry (esq_rhs_fused_fn,
// include/extensisq_amd.h): what a right-hand-side sweep does with the
// derivative it has just computed, while the value is still in registers.
//
// A Runge-Kutta stage is  K_i = f(t_i, ys_i)  followed by pointwise arithmetic
// on K_i and older rows.  The sweep that evaluates f is a stencil pass the RK
// library cannot see into, but everything AFTER it is pointwise, so the plugin
// applies one of the epilogues below per element and the separate streaming
// kernel -- with its re-read of K_i -- disappears:
//
//   EpiNone     store K_i only                                  (plain RHS)
//   EpiStage    next stage's argument  y + h*(init + sum a_j K_j + a_i K_i)
//               (common.py:355-356; also y_new of FSAL pairs, :343, and the
//               end-point evaluation chained with the next step's first stage)
//   EpiBlock    blocked accumulation with K_i as the block's last column
//   EpiSolErr   y_new, error and the partial sum of |err/scale|^2 with K_i as
//               the last stage derivative              (common.py:341-351)
//   EpiErrNorm  FSAL pairs: K_s = f(t+h, y_new) and the error norm in the
//               same sweep                             (common.py:348-351)
//
// Every epilogue runs the SAME ascending-j FMA chain as the stand-alone kernels
// of esq_kernels.hpp, with the fresh derivative entering last (it has the
// largest column index), so stage derivatives and states are bit-identical.
//
// A plugin kernel uses an epilogue `E epi` like this (per thread, per 16-byte
// element index i2 of the state vector; `fresh` = derivative, `centre` = the
// sweep's own input at the same index):
//     typename E::In in;  epi.load(in, i2);        // before the stencil: all
//     ... stencil ...                              // loads in flight together
//     epi.store_f(f, i2, fresh);
//     epi.finish(in, fresh, centre, i2, local);    // stores; adds to `local`
//     ... after the loop, by ALL threads of the block:
//     if (E::kReduce) block_partial(local, epi.red.partials);
//
// Every epilogue also has a ONE-DOUBLE-PER-THREAD twin of the three calls --
// `In1`, load1(in, i), store_f1(f, i, fresh), finish1(in, fresh, centre, i, local),
// i = element index -- for sweeps whose geometry rules out 16-byte accesses (3-D
// grids with an odd edge: csrc/esq_rhs_diff3d.hip).  Same operations, same order.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "esq_terms.hpp"

namespace esq {

constexpr int kMaxOut = 12;

// ---------------------------------------------------------------------------
// block reduction: wave64 shuffle tree -> LDS across the 4 waves -> one
// partial per block (fixed order => bitwise reproducible for a given grid).
// NaN/Inf propagate through plain adds.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ void block_partial(double v, double *partials) {
    __shared__ double lds[kBlock / 64];
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = lds[0];
#pragma unroll
        for (int w = 1; w < kBlock / 64; ++w) s += lds[w];
        partials[blockIdx.x] = s;
    }
}

// the same for a workgroup of NW waves (the split chain sweeps)
template <int NW>
__device__ __forceinline__ void block_partial_w(double v, double *partials) {
    __shared__ double lds[NW];
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = lds[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) s += lds[w];
        partials[blockIdx.x] = s;
    }
}

// np.maximum propagates NaN, fmax drops it: keep NumPy's semantics
__device__ __forceinline__ double pmax(double a, double b) {
    double m = fmax(a, b);
    m = (a != a) ? a : m;
    return (b != b) ? b : m;
}
// weights: scale = atol + rtol * max(|a|, |b|)         common.py:57-61
// CPLX: one double2 is one complex element, |.| = hypot (NumPy's complex abs)
template <bool CPLX>
__device__ __forceinline__ double ratio_sq(double2 err, double2 ya, double2 yb,
                                           const double *atol_vec,
                                           double atol_s, double rtol, size_t i,
                                           size_t n_valid) {
    if (CPLX) {
        if (i >= n_valid) return 0.0;
        const double at = atol_vec ? atol_vec[i] : atol_s;
        const double sc = at + rtol * pmax(hypot(ya.x, ya.y), hypot(yb.x, yb.y));
        const double rx = err.x / sc, ry = err.y / sc;
        return rx * rx + ry * ry;
    } else {
        const size_t e0 = 2 * i;
        double s = 0.0;
        if (e0 < n_valid) {
            const double at = atol_vec ? atol_vec[e0] : atol_s;
            const double sc = at + rtol * pmax(fabs(ya.x), fabs(yb.x));
            const double r = err.x / sc;
            s = r * r;
        }
        if (e0 + 1 < n_valid) {
            const double at = atol_vec ? atol_vec[e0 + 1] : atol_s;
            const double sc = at + rtol * pmax(fabs(ya.y), fabs(yb.y));
            const double r = err.y / sc;
            s += r * r;
        }
        return s;
    }
}

// one real element (the scalar twins of the epilogues)
__device__ __forceinline__ double ratio_sq1(double err, double ya, double yb,
                                            const double *atol_vec, double atol_s,
                                            double rtol, size_t i, size_t n_valid) {
    if (i >= n_valid) return 0.0;
    const double at = atol_vec ? atol_vec[i] : atol_s;
    const double sc = at + rtol * pmax(fabs(ya), fabs(yb));
    const double r = err / sc;
    return r * r;
}
__device__ __forceinline__ double ld1_nt(const double *p, size_t i) {
    return __builtin_nontemporal_load(p + i);
}
__device__ __forceinline__ void st1_nt(double *p, size_t i, double v) {
    __builtin_nontemporal_store(v, p + i);
}

// tolerances + destination of a weighted-RMS reduction (real states)
struct RedArgs {
    const double *atol_vec;
    double atol_s, rtol;
    size_t n_valid;
    double *partials;
};

// ---------------------------------------------------------------------------
struct EpiNone {
    static constexpr bool kReduce = false;
    int f_nt;
    RedArgs red;            // unused
    struct In {};
    __device__ __forceinline__ void load(In &, size_t) const {}
    __device__ __forceinline__ void store_f(double *f, size_t i2, double2 v) const {
        if (f_nt) st2_nt(f, i2, v); else st2(f, i2, v);
    }
    __device__ __forceinline__ void finish(const In &, double2, double2, size_t,
                                           double &) const {}
    struct In1 {};
    __device__ __forceinline__ void load1(In1 &, size_t) const {}
    __device__ __forceinline__ void store_f1(double *f, size_t i, double v) const {
        if (f_nt) st1_nt(f, i, v); else f[i] = v;
    }
    __device__ __forceinline__ void finish1(const In1 &, double, double, size_t,
                                            double &) const {}
};

// out = base + h * (init + sum_j c_j K_j + c_self * fresh);  base = y, or the
// sweep's own input when y == nullptr (end-point evaluation: the new state)
template <int NT>
struct EpiStage {
    static constexpr bool kReduce = false;
    Terms tm;
    const double *init, *y;
    double *out;
    double c_self, h;
    int f_nt;
    RedArgs red;            // unused
    struct In {
        double2 v[NT > 0 ? NT : 1], yb, acc0;
    };
    __device__ __forceinline__ void load(In &in, size_t i2) const {
#pragma unroll
        for (int j = 0; j < NT; ++j) in.v[j] = ld2_nt(tm.p[j], i2);
        in.yb = y ? ld2(y, i2) : make_double2(0.0, 0.0);
        in.acc0 = init ? ld2_nt(init, i2) : make_double2(0.0, 0.0);
    }
    __device__ __forceinline__ void store_f(double *f, size_t i2, double2 v) const {
        if (f_nt) st2_nt(f, i2, v); else st2(f, i2, v);
    }
    __device__ __forceinline__ void finish(const In &in, double2 fresh,
                                           double2 centre, size_t i2,
                                           double &) const {
        double2 acc = in.acc0;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            acc.x = fma(tm.c[j], in.v[j].x, acc.x);
            acc.y = fma(tm.c[j], in.v[j].y, acc.y);
        }
        if (c_self != 0.0) {                          // uniform
            acc.x = fma(c_self, fresh.x, acc.x);
            acc.y = fma(c_self, fresh.y, acc.y);
        }
        const double2 b = y ? in.yb : centre;
        st2(out, i2, make_double2(__dadd_rn(b.x, __dmul_rn(h, acc.x)),
                                  __dadd_rn(b.y, __dmul_rn(h, acc.y))));
    }
    struct In1 {
        double v[NT > 0 ? NT : 1], yb, acc0;
    };
    __device__ __forceinline__ void load1(In1 &in, size_t i) const {
#pragma unroll
        for (int j = 0; j < NT; ++j) in.v[j] = ld1_nt(tm.p[j], i);
        in.yb = y ? y[i] : 0.0;
        in.acc0 = init ? ld1_nt(init, i) : 0.0;
    }
    __device__ __forceinline__ void store_f1(double *f, size_t i, double v) const {
        if (f_nt) st1_nt(f, i, v); else f[i] = v;
    }
    __device__ __forceinline__ void finish1(const In1 &in, double fresh, double centre,
                                            size_t i, double &) const {
        double acc = in.acc0;
#pragma unroll
        for (int j = 0; j < NT; ++j) acc = fma(tm.c[j], in.v[j], acc);
        if (c_self != 0.0) acc = fma(c_self, fresh, acc);          // uniform
        const double b = y ? in.yb : centre;
        out[i] = __dadd_rn(b, __dmul_rn(h, acc));
    }
};

// Blocked accumulation (see k_block_acc) with the fresh derivative as the
// block's LAST column: out_o = init_o + sum_j w[j][o] K_j + w_self[o] * fresh.
// Output 0 may be the boundary stage's argument y + h*sum (y != nullptr).
template <int NT>
struct EpiBlock {
    static constexpr bool kReduce = false;
    const double *p[kMaxTerms];
    double w[kMaxTerms][kMaxOut];
    double w_self[kMaxOut];
    const double *init[kMaxOut];
    double *out[kMaxOut];
    const double *y;
    double h;
    int no, f_nt;
    RedArgs red;            // unused
    struct In {
        double2 v[NT > 0 ? NT : 1];
    };
    __device__ __forceinline__ void load(In &in, size_t i2) const {
#pragma unroll
        for (int j = 0; j < NT; ++j) in.v[j] = ld2_nt(p[j], i2);
    }
    __device__ __forceinline__ void store_f(double *f, size_t i2, double2 v) const {
        if (f_nt) st2_nt(f, i2, v); else st2(f, i2, v);
    }
    __device__ __forceinline__ void finish(const In &in, double2 fresh, double2,
                                           size_t i2, double &) const {
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o) {
            if (o < no) {                              // uniform
                // the output's column of weights is fetched HERE (an index the
                // compiler cannot see through): hoisted to the top of the kernel,
                // the up to 20 x 12 weights of all outputs live in scalar registers
                // at once and spill (EpiBlock<7>: 80-176 spilled SGPRs by
                // instantiation, the heat block sweep 118 vs 147 us)
                int oo = o;
                asm volatile("" : "+s"(oo));
                double2 acc = make_double2(0.0, 0.0);
                if (init[oo]) acc = ld2_nt(init[oo], i2);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const double wj = w[j][oo];
                    if (wj != 0.0) {                   // uniform (SGPR weights)
                        acc.x = fma(wj, in.v[j].x, acc.x);
                        acc.y = fma(wj, in.v[j].y, acc.y);
                    }
                }
                const double ws = w_self[oo];
                if (ws != 0.0) {
                    acc.x = fma(ws, fresh.x, acc.x);
                    acc.y = fma(ws, fresh.y, acc.y);
                }
                if (o == 0 && y) {
                    const double2 yb = ld2(y, i2);
                    acc.x = __dadd_rn(yb.x, __dmul_rn(h, acc.x));
                    acc.y = __dadd_rn(yb.y, __dmul_rn(h, acc.y));
                    st2(out[oo], i2, acc);     // stage argument: read next
                } else {
                    st2_nt(out[oo], i2, acc);  // partial sums: stream out
                }
            }
        }
    }
    struct In1 {
        double v[NT > 0 ? NT : 1];
    };
    __device__ __forceinline__ void load1(In1 &in, size_t i) const {
#pragma unroll
        for (int j = 0; j < NT; ++j) in.v[j] = ld1_nt(p[j], i);
    }
    __device__ __forceinline__ void store_f1(double *f, size_t i, double v) const {
        if (f_nt) st1_nt(f, i, v); else f[i] = v;
    }
    __device__ __forceinline__ void finish1(const In1 &in, double fresh, double,
                                            size_t i, double &) const {
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o) {
            if (o < no) {                              // uniform
                double acc = init[o] ? ld1_nt(init[o], i) : 0.0;
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    if (w[j][o] != 0.0) acc = fma(w[j][o], in.v[j], acc);
                if (w_self[o] != 0.0) acc = fma(w_self[o], fresh, acc);
                if (o == 0 && y) out[o][i] = __dadd_rn(y[i], __dmul_rn(h, acc));
                else st1_nt(out[o], i, acc);
            }
        }
    }
};

// y_new = y + h*(sum b_j K_j + b_self*fresh); err = h*(sum e_j K_j + e_self*fresh);
// partial sum of |err / (atol + rtol*max(|y|, |y_new|))|^2
// CPLX: one double2 is one complex element; the sums act on (re, im) separately
// (real weights), the scale uses the complex modulus (common.py:57-66)
template <int NT, bool CPLX = false>
struct EpiSolErr {
    static constexpr bool kReduce = true;
    Terms2 tm;
    double b_self, e_self;
    const double *y;
    double *ynew;
    double h;
    int f_nt;
    RedArgs red;
    struct In {
        double2 v[NT > 0 ? NT : 1], yb;
    };
    __device__ __forceinline__ void load(In &in, size_t i2) const {
#pragma unroll
        for (int j = 0; j < NT; ++j) in.v[j] = ld2_nt(tm.p[j], i2);
        in.yb = ld2(y, i2);
    }
    __device__ __forceinline__ void store_f(double *f, size_t i2, double2 v) const {
        if (f_nt) st2_nt(f, i2, v); else st2(f, i2, v);
    }
    __device__ __forceinline__ void finish(const In &in, double2 fresh, double2,
                                           size_t i2, double &local) const {
        double2 sb = make_double2(0.0, 0.0), se = make_double2(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            sb.x = fma(tm.b[j], in.v[j].x, sb.x);
            sb.y = fma(tm.b[j], in.v[j].y, sb.y);
            se.x = fma(tm.e[j], in.v[j].x, se.x);
            se.y = fma(tm.e[j], in.v[j].y, se.y);
        }
        // the fresh row is part of the union of the two supports: a zero
        // weight contributes fma(0, v, s) == s, as in k_solution_error
        sb.x = fma(b_self, fresh.x, sb.x);
        sb.y = fma(b_self, fresh.y, sb.y);
        se.x = fma(e_self, fresh.x, se.x);
        se.y = fma(e_self, fresh.y, se.y);
        double2 yn, er;
        yn.x = __dadd_rn(in.yb.x, __dmul_rn(h, sb.x));
        yn.y = __dadd_rn(in.yb.y, __dmul_rn(h, sb.y));
        er.x = __dmul_rn(h, se.x);
        er.y = __dmul_rn(h, se.y);
        st2(ynew, i2, yn);
        local += ratio_sq<CPLX>(er, in.yb, yn, red.atol_vec, red.atol_s, red.rtol,
                                i2, red.n_valid);
    }
    struct In1 {
        double v[NT > 0 ? NT : 1], yb;
    };
    __device__ __forceinline__ void load1(In1 &in, size_t i) const {
#pragma unroll
        for (int j = 0; j < NT; ++j) in.v[j] = ld1_nt(tm.p[j], i);
        in.yb = y[i];
    }
    __device__ __forceinline__ void store_f1(double *f, size_t i, double v) const {
        if (f_nt) st1_nt(f, i, v); else f[i] = v;
    }
    __device__ __forceinline__ void finish1(const In1 &in, double fresh, double,
                                            size_t i, double &local) const {
        static_assert(!CPLX, "the scalar twin is for real states");
        double sb = 0.0, se = 0.0;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            sb = fma(tm.b[j], in.v[j], sb);
            se = fma(tm.e[j], in.v[j], se);
        }
        sb = fma(b_self, fresh, sb);
        se = fma(e_self, fresh, se);
        const double yn = __dadd_rn(in.yb, __dmul_rn(h, sb));
        const double er = __dmul_rn(h, se);
        ynew[i] = yn;
        local += ratio_sq1(er, in.yb, yn, red.atol_vec, red.atol_s, red.rtol, i,
                           red.n_valid);
    }
};

// FSAL: the sweep's input IS y_new, fresh = K_s;
// err = h*(sum e_j K_j + e_self*fresh), scale from y (memory) and y_new (centre)
template <int NT, bool CPLX = false>
struct EpiErrNorm {
    static constexpr bool kReduce = true;
    Terms tm;
    double e_self;
    const double *y;
    double h;
    int f_nt;
    RedArgs red;
    struct In {
        double2 v[NT > 0 ? NT : 1], yb;
    };
    __device__ __forceinline__ void load(In &in, size_t i2) const {
#pragma unroll
        for (int j = 0; j < NT; ++j) in.v[j] = ld2(tm.p[j], i2);
        in.yb = ld2(y, i2);
    }
    __device__ __forceinline__ void store_f(double *f, size_t i2, double2 v) const {
        if (f_nt) st2_nt(f, i2, v); else st2(f, i2, v);
    }
    __device__ __forceinline__ void finish(const In &in, double2 fresh,
                                           double2 centre, size_t i2,
                                           double &local) const {
        double2 se = make_double2(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            se.x = fma(tm.c[j], in.v[j].x, se.x);
            se.y = fma(tm.c[j], in.v[j].y, se.y);
        }
        se.x = fma(e_self, fresh.x, se.x);
        se.y = fma(e_self, fresh.y, se.y);
        double2 er;
        er.x = __dmul_rn(h, se.x);
        er.y = __dmul_rn(h, se.y);
        local += ratio_sq<CPLX>(er, in.yb, centre, red.atol_vec, red.atol_s,
                                red.rtol, i2, red.n_valid);
    }
    struct In1 {
        double v[NT > 0 ? NT : 1], yb;
    };
    __device__ __forceinline__ void load1(In1 &in, size_t i) const {
#pragma unroll
        for (int j = 0; j < NT; ++j) in.v[j] = tm.p[j][i];
        in.yb = y[i];
    }
    __device__ __forceinline__ void store_f1(double *f, size_t i, double v) const {
        if (f_nt) st1_nt(f, i, v); else f[i] = v;
    }
    __device__ __forceinline__ void finish1(const In1 &in, double fresh, double centre,
                                            size_t i, double &local) const {
        static_assert(!CPLX, "the scalar twin is for real states");
        double se = 0.0;
#pragma unroll
        for (int j = 0; j < NT; ++j) se = fma(tm.c[j], in.v[j], se);
        se = fma(e_self, fresh, se);
        const double er = __dmul_rn(h, se);
        local += ratio_sq1(er, in.yb, centre, red.atol_vec, red.atol_s, red.rtol, i,
                           red.n_valid);
    }
};

// Chebyshev recursion (sommeijer.py:312-313), same operation order as
// k_rkc_stage; the derivative itself is not stored
struct EpiRkc {
    static constexpr bool kReduce = false;
    const double *yjm2, *yn, *fn;
    double *out;
    double mu, nu, omn, hmus, ajm1;
    RedArgs red;            // unused
    struct In {
        double2 b, c0, g;
    };
    __device__ __forceinline__ void load(In &in, size_t i2) const {
        in.b = ld2(yjm2, i2);
        in.c0 = ld2(yn, i2);
        in.g = ld2(fn, i2);
    }
    __device__ __forceinline__ void store_f(double *, size_t, double2) const {}
    __device__ __forceinline__ double one(double yjm1, double b, double c0,
                                          double g, double fy) const {
        return __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(mu, yjm1), __dmul_rn(nu, b)),
                                   __dmul_rn(omn, c0)),
                         __dmul_rn(hmus, __dsub_rn(fy, __dmul_rn(ajm1, g))));
    }
    __device__ __forceinline__ void finish(const In &in, double2 fresh,
                                           double2 centre, size_t i2,
                                           double &) const {
        st2(out, i2, make_double2(one(centre.x, in.b.x, in.c0.x, in.g.x, fresh.x),
                                  one(centre.y, in.b.y, in.c0.y, in.g.y, fresh.y)));
    }
    struct In1 {
        double b, c0, g;
    };
    __device__ __forceinline__ void load1(In1 &in, size_t i) const {
        in.b = yjm2[i];
        in.c0 = yn[i];
        in.g = fn[i];
    }
    __device__ __forceinline__ void store_f1(double *, size_t, double) const {}
    __device__ __forceinline__ void finish1(const In1 &in, double fresh, double centre,
                                            size_t i, double &) const {
        out[i] = one(centre, in.b, in.c0, in.g, fresh);
    }
};

// End of a Chebyshev step (sommeijer.py:214-220): the sweep's input is the new
// state y, fresh = f(t + h, y) (stored: it is the next step's f_n);
//   est = 0.8*(yn - y) + 0.4*h*(fn + f);  wt = atol + rtol*max(|y|, |yn|)
// partial sums of |est / wt|^2, same operation order as k_rkc_error
struct EpiRkcErr {
    static constexpr bool kReduce = true;
    const double *yn, *fn;
    double h04;
    int f_nt;
    RedArgs red;
    struct In {
        double2 b, g;
    };
    __device__ __forceinline__ void load(In &in, size_t i2) const {
        in.b = ld2(yn, i2);
        in.g = ld2(fn, i2);
    }
    __device__ __forceinline__ void store_f(double *f, size_t i2, double2 v) const {
        if (f_nt) st2_nt(f, i2, v); else st2(f, i2, v);
    }
    __device__ __forceinline__ double one(double y, double b, double g, double fy) const {
        return __dadd_rn(__dmul_rn(0.8, __dsub_rn(b, y)),
                         __dmul_rn(h04, __dadd_rn(g, fy)));
    }
    __device__ __forceinline__ void finish(const In &in, double2 fresh,
                                           double2 centre, size_t i2,
                                           double &local) const {
        const double2 er = make_double2(one(centre.x, in.b.x, in.g.x, fresh.x),
                                        one(centre.y, in.b.y, in.g.y, fresh.y));
        local += ratio_sq<false>(er, centre, in.b, red.atol_vec, red.atol_s,
                                 red.rtol, i2, red.n_valid);
    }
    struct In1 {
        double b, g;
    };
    __device__ __forceinline__ void load1(In1 &in, size_t i) const {
        in.b = yn[i];
        in.g = fn[i];
    }
    __device__ __forceinline__ void store_f1(double *f, size_t i, double v) const {
        if (f_nt) st1_nt(f, i, v); else f[i] = v;
    }
    __device__ __forceinline__ void finish1(const In1 &in, double fresh, double centre,
                                            size_t i, double &local) const {
        const double er = one(centre, in.b, in.g, fresh);
        local += ratio_sq1(er, centre, in.b, red.atol_vec, red.atol_s, red.rtol, i,
                           red.n_valid);
    }
};

}  // namespace esq
