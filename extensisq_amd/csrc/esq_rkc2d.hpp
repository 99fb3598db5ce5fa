// esq_rkc2d.hpp -- D consecutive stages of the Runge-Kutta-Chebyshev recursion
// (sommeijer.py:309-329) in ONE marching sweep of a one-field 2-D five-point stencil
// plugin (esq_rhs_rkc_chain_fn, include/extensisq_amd.h): the 2-D sibling of
// esq_rkc3d.hpp, on the tile geometry of the marching chain sweeps (esq_chain.hpp).
//
//   Y_{k+1} = mu_k Y_k + nu_k Y_{k-1} + (1 - mu_k - nu_k) y_n
//             + h mu~_k (f(Y_k) - a_k f_n),      k = 0 .. D-1,   Y_0 = y_{j-1}, Y_{-1} = y_{j-2}
//
// One launch per stage moves 5 words per element (y_{j-1}, y_{j-2}, y_n, f_n in, y_j
// out); a chain reads the four inputs once and writes the last two iterates: 6 words
// per D stages.
//
// Geometry.  One wave owns a tile of R grid rows x (64 - 2H) column pairs, lane l
// holds the pair W*ct - H + l (16-byte accesses); stage k runs k rows behind stage 0.
// Y_{k+1} is valid where Y_k was valid one COLUMN and one ROW further out, so the
// tile carries H = ceil(D / 2) halo pairs per side and D - 1 run-in / run-out rows;
// every left / right neighbour is a DPP wave shift (no LDS, no barrier: a wave is a
// tile).  Outside the grid every iterate is zero (Dirichlet) -- masked exactly where
// the one-stage sweep reads a zero neighbour.  Same stencil expression, the same
// left-to-right recursion with every operation rounded: bit-identical to D launches
// of the one-stage entry.
//
// Every access of the marching loop is an unconditional raw-buffer access (masked
// lanes at an out-of-range offset, rows that do not exist through a zero-byte
// resource: esq_rkc3d.hpp has the reasons); the 16-byte stores carry the two wait
// states gfx950 needs behind them (profiles/r04_experiments.md, section 8).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/extensisq_amd.h"
#include "esq_epilogue.hpp"
#include "esq_terms.hpp"

namespace esq {

template <int D>
struct Rkc2dArgs {
    const double *a, *b, *yn, *fn;          // y_{j-1}, y_{j-2}, y_n, f_n
    double *out, *outp;                     // Y_D, Y_{D-1} (outp may be null)
    double mu[D], nu[D], omn[D], hmus[D], ajm1[D];
    double hmus1;                           // FIRST: a = y_n + hmus1 * f_n, b = y_n
    // LAST: stage slot D - 1 is the end of the step (esq_rkc3d.hpp): f = Fn(Y_{D-1}) ->
    // out, Y_{D-1} -> outp, partial sums of the error estimate (ESQ_EPI_RKCERR)
    double h04;
    RedArgs red;
};

// 16-byte buffer accesses (rsrc_t, make_rsrc: esq_terms.hpp)
__device__ __forceinline__ double2 buf_ld2(rsrc_t r, unsigned voff, unsigned soff) {
    using v4u = decltype(__builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, 0));
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    const v2d d = __builtin_bit_cast(v2d, v);
    return make_double2(d.x, d.y);
}
__device__ __forceinline__ void buf_st2(rsrc_t r, unsigned voff, unsigned soff, double2 v) {
    using v4u = decltype(__builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, 0));
    v2d d;
    d.x = v.x;
    d.y = v.y;
    const v4u w = __builtin_bit_cast(v4u, d);
    __builtin_amdgcn_raw_buffer_store_b128(w, r, (int)voff, (int)soff, 0);
    // gfx950: a buffer store of more than 64 bits still reads its data registers for
    // two cycles after issue, also with a scalar register as soffset -- a form the
    // compiler's hazard table exempts.  Two wait states with the data still live:
    asm volatile("s_nop 1" : : "v"(w));
}

// Fn::eval(centres, laplacians) -> derivatives (esq_stencil2d.hpp), one field.
// FIRST: the chain opens a step -- its first input is y_1 = y_n + hmus1 * f_n, formed
// where the window needs it; y_{j-2} = y_n (esq_rkc3d.hpp).
// LAST: the chain ends a step -- D - 1 Chebyshev stages, then f(t + h, y_{n+1}) and the
// error estimate 0.8 (y_n - y) + 0.4 h (f_n + f) as stage slot D - 1 (sommeijer.py:214-220).
template <bool PERIODIC, int D, class Fn, bool FIRST = false, bool LAST = false>
__global__ __launch_bounds__(kBlock) void k_rkc2d_chain(Rkc2dArgs<D> ca, Fn fn, int N, int R,
                                                       unsigned tpr, unsigned ntiles,
                                                       unsigned nblocks, unsigned xcd) {
    constexpr int H = (D + 1) / 2;                 // halo pairs per side
    constexpr int W = 64 - 2 * H;                  // stored pairs per tile
    const unsigned per = (nblocks + xcd - 1) / xcd;
    const unsigned lb = (blockIdx.x % xcd) * per + blockIdx.x / xcd;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned tile = lb * (kBlock / 64) + wave;
    double local = 0.0;                                              // LAST: error sum
    if (lb >= nblocks || tile >= ntiles) {                           // wave-uniform
        if constexpr (LAST) block_partial(local, ca.red.partials);   // (every wave of the block)
        return;
    }
    const int npairs = N / 2;
    const int lane = threadIdx.x & 63;
    const int pc = W * (int)(tile % tpr) - H + lane;
    const bool indom = pc >= 0 && pc < npairs;
    const bool live = PERIODIC ? (pc >= -H && pc < npairs + H) : indom;
    const int pw = PERIODIC ? (pc < 0 ? pc + npairs : (pc >= npairs ? pc - npairs : pc)) : pc;
    const bool store_ok = indom && lane >= H && lane < 64 - H;
    const int r0 = (int)(tile / tpr) * R;
    const int Re = (N - r0) < R ? (N - r0) : R;
    const unsigned vbytes = (unsigned)((size_t)N * (size_t)npairs * 16u);
    const unsigned row_bytes = (unsigned)npairs * 16u;
    // lane parts of the byte offsets (0xfffffff0: masked out, all four dwords out of
    // range): loads / stores
    const unsigned vl = live ? (unsigned)pw * 16u : 0xfffffff0u;
    const unsigned vs = store_ok ? (unsigned)pw * 16u : 0xfffffff0u;
    auto row_ok = [&](int r) { return PERIODIC || (r >= 0 && r < N); };
    auto wrap = [&](int r) { return PERIODIC ? (r < 0 ? r + N : (r >= N ? r - N : r)) : r; };
    auto ld = [&](const double *p, int r) -> double2 {
        const bool ok = row_ok(r);                                  // uniform
        return buf_ld2(make_rsrc(p, ok ? vbytes : 0u), vl, ok ? (unsigned)wrap(r) * row_bytes : 0u);
    };
    auto first = [&](double2 y, double2 f) -> double2 {             // k_rkc_first's rounding
        return make_double2(__dadd_rn(y.x, __dmul_rn(ca.hmus1, f.x)),
                            __dadd_rn(y.y, __dmul_rn(ca.hmus1, f.y)));
    };
    auto one = [&](int k, double yjm1, double bb, double c0, double gg, double fy) -> double {
        return __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(ca.mu[k], yjm1), __dmul_rn(ca.nu[k], bb)),
                                   __dmul_rn(ca.omn[k], c0)),
                         __dmul_rn(ca.hmus[k], __dsub_rn(fy, __dmul_rn(ca.ajm1[k], gg))));
    };
    const double2 zero = make_double2(0.0, 0.0);
    // windows: wm[k], wc[k] = Y_k at rows (centre - 1, centre) of stage k;
    // dy[k], df[k] = y_n, f_n at stage k's centre row (a delay line)
    double2 wm[D], wc[D], dy[D], df[D];
#pragma unroll
    for (int k = 0; k < D; ++k) wm[k] = wc[k] = dy[k] = df[k] = zero;
    const int rbase = r0 - (D - 1);                  // stage 0's first centre row
    double2 cy = zero, cf = zero;                    // FIRST: y_n, f_n at stage 0's centre row
    if constexpr (FIRST) {
        wm[0] = first(ld(ca.yn, rbase - 1), ld(ca.fn, rbase - 1));
        cy = ld(ca.yn, rbase);
        cf = ld(ca.fn, rbase);
        wc[0] = first(cy, cf);
    } else {
        wm[0] = ld(ca.a, rbase - 1);
        wc[0] = ld(ca.a, rbase);
    }
    // operands of stage 0's row, requested ONE ITERATION AHEAD
    double2 pa = zero, pb = zero, py, pf;
#define ESQ_RKC2D_LOAD(IT)                                                  \
    {                                                                       \
        const int i_ = rbase + (IT);                                        \
        if constexpr (FIRST) {                                              \
            py = ld(ca.yn, i_ + 1);                                         \
            pf = ld(ca.fn, i_ + 1);                                         \
        } else {                                                            \
            pa = ld(ca.a, i_ + 1);                                          \
            pb = ld(ca.b, i_);                                              \
            py = ld(ca.yn, i_);                                             \
            pf = ld(ca.fn, i_);                                             \
        }                                                                   \
    }
    ESQ_RKC2D_LOAD(0)
    __builtin_amdgcn_s_waitcnt(0x0f70);              // nothing pending at the loop header
    const int iters = Re + 2 * (D - 1);
    for (int it = 0; it < iters; ++it) {
        const int i0 = rbase + it;
        double2 wp, ykm1;
        if constexpr (FIRST) {
            wp = first(py, pf);
            ykm1 = cy;
            dy[0] = cy;
            df[0] = cf;
            cy = py;
            cf = pf;
        } else {
            wp = pa;
            ykm1 = pb;
            dy[0] = py;
            df[0] = pf;
        }
        ESQ_RKC2D_LOAD(it + 1)
#pragma unroll
        for (int k = 0; k < D; ++k) {
            double2 nw = zero;
            const int ik = i0 - k;
            const bool on = it >= 2 * k;                             // wave-uniform
            if (on) {
                const bool ok = live && row_ok(ik);
                const double lf = lane_left(wc[k].y);
                const double rt = lane_right(wc[k].x);
                double2 cc[1], lap[1], fy[1];
                cc[0] = wc[k];
                lap[0].x = ((wm[k].x + wp.x) + (lf + wc[k].y)) - 4.0 * wc[k].x;
                lap[0].y = ((wm[k].y + wp.y) + (wc[k].x + rt)) - 4.0 * wc[k].y;
                fn.eval(cc, lap, fy);
                if (LAST && k == D - 1) {
                    nw = make_double2(ok ? fy[0].x : 0.0, ok ? fy[0].y : 0.0);
                    if (store_ok) {                                  // a pair this tile stores
                        const double2 er = make_double2(
                            __dadd_rn(__dmul_rn(0.8, __dsub_rn(dy[k].x, wc[k].x)),
                                      __dmul_rn(ca.h04, __dadd_rn(df[k].x, fy[0].x))),
                            __dadd_rn(__dmul_rn(0.8, __dsub_rn(dy[k].y, wc[k].y)),
                                      __dmul_rn(ca.h04, __dadd_rn(df[k].y, fy[0].y))));
                        const size_t i2 = (size_t)wrap(ik) * (size_t)npairs + (size_t)pw;
                        local += ratio_sq<false>(er, wc[k], dy[k], ca.red.atol_vec, ca.red.atol_s,
                                                 ca.red.rtol, i2, ca.red.n_valid);
                    }
                } else {
                    const double vx = one(k, wc[k].x, ykm1.x, dy[k].x, df[k].x, fy[0].x);
                    const double vy = one(k, wc[k].y, ykm1.y, dy[k].y, df[k].y, fy[0].y);
                    nw = make_double2(ok ? vx : 0.0, ok ? vy : 0.0);
                }
            }
            if (k == D - 1) {
                // the last stage is only ever at rows [r0, r0 + Re); before it has
                // started the stores go to a zero-byte resource
                const unsigned so = on ? (unsigned)wrap(ik) * row_bytes : 0u;
                buf_st2(make_rsrc(ca.out, on ? vbytes : 0u), vs, so, nw);
                buf_st2(make_rsrc(ca.outp, (on && ca.outp) ? vbytes : 0u), vs, so, wc[k]);
            }
            // Y_k's window moves one row on; its old lower row is Y_{(k+1)-1} at
            // stage k + 1's centre row, the fresh values are that stage's upper row
            const double2 old = wm[k];
            wm[k] = wc[k];
            wc[k] = wp;
            ykm1 = old;
            wp = nw;
        }
#pragma unroll
        for (int k = D - 1; k >= 1; --k) {
            dy[k] = dy[k - 1];
            df[k] = df[k - 1];
        }
    }
#undef ESQ_RKC2D_LOAD
    if constexpr (LAST) block_partial(local, ca.red.partials);
}

}  // namespace esq
