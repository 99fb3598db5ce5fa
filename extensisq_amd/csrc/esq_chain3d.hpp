// esq_chain3d.hpp -- D consecutive Runge-Kutta stages of a 3-D seven-point stencil
// plugin in ONE marching sweep (esq_rhs_chain_fn, include/extensisq_amd.h; reference
// loop: extensisq/common.py:353-356, the solution and error sums :341-351).
//
// The arithmetic is that of the 2-D chain sweeps (esq_chain.hpp): stage k runs k
// planes behind stage 0; target e + 1's sum for a plane starts when stage 0 visits
// the plane (the K rows read from memory, ONCE for all D targets, ascending column
// order; a leading partial sum of the blocked accumulation is the first such row
// with weight 1) and takes the chain's own derivatives as they appear, one per
// iteration -- the same FMA chain, then *h, then +y, each rounded, as every other
// kernel of the library: K rows and states are bit-identical to D one-stage sweeps.
// A row that takes no part in a target enters it with weight +0.0 instead of being
// skipped under a participation mask (the 2-D sweeps' way): fma(0, v, s) == s for
// every finite v and every s a sum started at +0.0 can hold -- no scalar bit test and
// branch per term, which the 3-D sweeps (two rows per thread, sixteen waves per
// workgroup behind one scalar unit) were bound by; a non-finite v poisons the
// target, as NumPy's K[:i].T @ A[i, :i] over ALL rows does (common.py:355).
// The D - 1 intermediate stage arguments never touch memory; y is read once.
//
// The geometry is that of the Chebyshev chain sweeps (esq_rkc3d.hpp): a workgroup of
// NW waves owns a patch of the (j, l) plane and marches along i; lane <-> l
// (coalesced 512-byte rows, left / right neighbours by DPP wave shifts), a thread
// holds JT consecutive rows j (its j-neighbours are its own registers, the slice's
// edge rows change hands through LDS: one workgroup barrier per plane, off the
// stage-to-stage dependency chain).  T_{k+1} is valid where T_k was valid one point
// further out: D points on every side of the patch are halo, D - 1 planes of run-in
// and run-out per tile.  Every load and store is a raw buffer access whose offset
// is 0xffffffff for a masked lane (no branch in the marching loop).
// Dirichlet 0: every stage argument is masked with the grid.
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <atomic>

#include "../../include/extensisq_amd.h"
#include "esq_chain.hpp"
#include "esq_epilogue.hpp"
#include "esq_plugin.hpp"
#include "esq_rkc3d.hpp"
#include "esq_terms.hpp"

namespace esq {

// cache policy of a raw buffer access: 0 = cacheable, 2 = non-temporal (nt)
template <int AUX>
__device__ __forceinline__ double buf_ld_p(rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(double,
                              __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, AUX));
}
template <int AUX>
__device__ __forceinline__ void buf_st_p(rsrc_t r, unsigned voff, unsigned soff, double v) {
    using v2u = decltype(__builtin_amdgcn_raw_buffer_load_b64(r, 0, 0, 0));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), r, (int)voff, (int)soff, AUX);
}

// St::eval(below, above, up, down, left, right, centre) -> derivative at the point
// (i-1, i+1; j-1, j+1; l-1, l+1); autonomous stencils, zero outside the grid.
// KINDLAST: ESQ_EPI_STAGE (the last target is a stage argument / y_new of an FSAL
// pair) or ESQ_EPI_SOLERR (y_new and the error partial sums).
#ifndef ESQ_CHAIN3D_PIN
#define ESQ_CHAIN3D_PIN 1
#endif
#ifndef ESQ_CHAIN3D_PF
#define ESQ_CHAIN3D_PF 0
#endif
template <int D, int NU, int JT, int NW, int KINDLAST, class St>
__global__ __launch_bounds__(64 * NW) void k_chain3d(const double *__restrict__ ys,
                                                     ChainArgs<D, NU> ca, St st, Geo3d g) {
    constexpr bool SOLERR = KINDLAST == ESQ_EPI_SOLERR;
    constexpr int NUa = ChainArgs<D, NU>::NUa;
    // edge rows of the D centre planes: slot w + 1 belongs to wave w, slots 0 and
    // NW + 1 stay zero (outside the patch), two buffers by iteration parity
    __shared__ double xch[2][D][NW + 2][2][64];
    const unsigned per = g.grid / 8u;
    const unsigned lb = (blockIdx.x % 8u) * per + blockIdx.x / 8u;
    if (lb >= g.ntiles) {                                        // workgroup-uniform
        if constexpr (SOLERR) block_partial_w<NW>(0.0, ca.red.partials);
        return;
    }
    double local = 0.0;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int N = g.N;
    const int tl = (int)(lb % g.TL), tj = (int)((lb / g.TL) % g.TJ);
    const int ti = (int)(lb / (g.TL * g.TJ));
    const int l = tl * g.VL - D + lane;
    const int jb = tj * g.VJ - D + w * JT;
    const int i_lo = ti * g.R;
    const int i_hi = i_lo + g.R < N ? i_lo + g.R : N;
    const int Re = i_hi - i_lo;
    const unsigned plane_bytes = (unsigned)N * (unsigned)N * 8u;
    const size_t vec_bytes = (size_t)plane_bytes * (size_t)N;
    // the chain's derivatives are streamed out (non-temporal) where the library says
    // that nothing reads them soon (esq_chain.f_store_nt)
    const bool f_nt = ca.f_nt != 0;
    // the chain's input T_0; ca.y == nullptr: the chain starts from the state itself
    // (stage 0 = f(t, y)), the base of every target is that input
    const bool own_base = ca.y == nullptr;
    const rsrc_t rin = make_rsrc(ys, vec_bytes);
    const rsrc_t ry = make_rsrc(own_base ? ys : ca.y, own_base ? 0 : vec_bytes);
    const rsrc_t rout = make_rsrc(ca.out, ca.out ? vec_bytes : 0);   // null: stores dropped
    rsrc_t ru[NUa], rk[D];
#pragma unroll
    for (int u = 0; u < NU; ++u) ru[u] = make_rsrc(ca.rows[u], vec_bytes);
#pragma unroll
    for (int k = 0; k < D; ++k) rk[k] = make_rsrc(ca.fk[k], ca.fk[k] ? vec_bytes : 0);
    // every weight as a scalar of its own (esq_chain.hpp: taken straight from the
    // argument struct the compiler re-reads whole load tuples per use)
    double w_cu[D][NUa], w_eu[NUa], w_ck[D][D], w_ek[D];
#if ESQ_CHAIN3D_PIN
#define ESQ_W_(DST, SRC) asm("s_mov_b64 %0, %1" : "=s"(DST) : "s"(SRC))
#else
#define ESQ_W_(DST, SRC) DST = SRC
#endif
#pragma unroll
    for (int e = 0; e < D; ++e) {
#pragma unroll
        for (int u = 0; u < NU; ++u) ESQ_W_(w_cu[e][u], ca.cu[e][u]);
#pragma unroll
        for (int k = 0; k < D; ++k) ESQ_W_(w_ck[e][k], ca.ck[e][k]);
        ESQ_W_(w_ek[e], ca.ek[e]);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) ESQ_W_(w_eu[u], ca.eu[u]);
#undef ESQ_W_
    // in[r]: the point is in the grid AND within D points of the stored patch
    const bool l_in = l >= 0 && l < N && lane < g.VL + 2 * D;
    const bool l_own = l_in && lane >= D && lane < D + g.VL;
    bool in[JT];
    unsigned vo[JT], so[JT];          // byte offsets inside a plane: loads / stores
#pragma unroll
    for (int r = 0; r < JT; ++r) {
        const int j = jb + r;
        in[r] = l_in && j >= 0 && j < N && j < (tj + 1) * g.VJ + D;
        const bool own = in[r] && l_own && j >= tj * g.VJ && j < (tj + 1) * g.VJ;
        const unsigned off = ((unsigned)j * (unsigned)N + (unsigned)l) * 8u;
        vo[r] = in[r] ? off : 0xffffffffu;
        so[r] = own ? off : 0xffffffffu;
    }
    if (threadIdx.x < 64) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int k = 0; k < D; ++k)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    xch[b][k][0][e][lane] = 0.0;
                    xch[b][k][NW + 1][e][lane] = 0.0;
                }
    }
    // (every load cacheable: the library's per-row streaming hint, esq_chain.load_nt, is
    // NOT followed here -- the halo points of neighbouring tiles meet in L2, and
    // non-temporal loads lose those hits: the step's last chain 117 -> 132 us at N = 159,
    // 1.51 -> 1.93 ms at N = 400, profiles/r05_experiments.md §1)
    auto ld = [&](rsrc_t v, int i, int r, unsigned) -> double {
        const bool ok = i >= 0 && i < N;                         // uniform
        return buf_ld(v, ok ? vo[r] : 0xffffffffu, ok ? (unsigned)i * plane_bytes : 0u);
    };
    // windows: wm[k], wc[k] = T_k at planes (centre - 1, centre) of stage k
    double wm[D][JT], wc[D][JT];
    // acc[e][k]: target e + 1's sum for the plane stage k is at (k <= e);
    // yf[k]: the base state at that plane
    double acc[D][D][JT], acce[D][JT], yf[D][JT];
#pragma unroll
    for (int k = 0; k < D; ++k)
#pragma unroll
        for (int r = 0; r < JT; ++r) {
            wm[k][r] = wc[k][r] = yf[k][r] = acce[k][r] = 0.0;
#pragma unroll
            for (int e = 0; e < D; ++e) acc[e][k][r] = 0.0;
        }
    const int ibase = i_lo - (D - 1);                 // stage 0's first centre plane
#pragma unroll
    for (int r = 0; r < JT; ++r) {
        wm[0][r] = ld(rin, ibase - 1, r, 0);
        wc[0][r] = ld(rin, ibase, r, 0);
    }
    // operands of stage 0's plane: the input one plane up, the base state and the
    // memory rows at the plane; requested ONE ITERATION AHEAD -- right behind the
    // sums that consume the previous set, into the same registers
    double pin[JT], py[JT], pu[NUa][JT];
#define ESQ_CHAIN3D_LOAD(IT)                                                \
    {                                                                       \
        const int i_ = ibase + (IT);                                        \
        _Pragma("unroll") for (int r = 0; r < JT; ++r) {                    \
            pin[r] = ld(rin, i_ + 1, r, 0);                                 \
            py[r] = ld(ry, i_, r, 1);                                       \
            _Pragma("unroll") for (int u = 0; u < NU; ++u)                  \
                pu[u][r] = ld(ru[u], i_, r, 8 + u);                         \
        }                                                                   \
    }
    if (ESQ_CHAIN3D_PF) ESQ_CHAIN3D_LOAD(0)
    const int iters = Re + 2 * (D - 1);
    for (int it = 0; it < iters; ++it) {
        const int i0 = ibase + it;
        double wp[JT];
        if (!ESQ_CHAIN3D_PF) ESQ_CHAIN3D_LOAD(it)
#pragma unroll
        for (int r = 0; r < JT; ++r) {
            wp[r] = pin[r];
            yf[0][r] = own_base ? wc[0][r] : py[r];
        }
        // ---- the D targets' sums over the memory rows of plane i0 (requested one
        // iteration ago) ...
#pragma unroll
        for (int e = 0; e < D; ++e)
#pragma unroll
            for (int r = 0; r < JT; ++r) {
                double s_ = 0.0, se_ = 0.0;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    s_ = fma(w_cu[e][u], pu[u][r], s_);
                    if (SOLERR && e == D - 1) se_ = fma(w_eu[u], pu[u][r], se_);
                }
                acc[e][0][r] = s_;
                if (SOLERR && e == D - 1) acce[0][r] = se_;
            }
        // ---- ... and the next plane's operands into the registers they leave: in
        // flight while the stages below run (PF = false: requested where they are used)
        if (ESQ_CHAIN3D_PF && it + 1 < iters) ESQ_CHAIN3D_LOAD(it + 1)
        // the slices' edge rows of all D centre planes change hands
#pragma unroll
        for (int k = 0; k < D; ++k) {
            xch[it & 1][k][w + 1][0][lane] = wc[k][0];
            xch[it & 1][k][w + 1][1][lane] = wc[k][JT - 1];
        }
        __syncthreads();
        double eu[D], ed[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            eu[k] = xch[it & 1][k][w][1][lane];
            ed[k] = xch[it & 1][k][w + 2][0][lane];
        }
        // ---- the stages, each one plane behind its predecessor
#pragma unroll
        for (int k = 0; k < D; ++k) {
            double nw[JT];
            if (it >= 2 * k) {                                   // wave-uniform
                const int ik = i0 - k;
                const bool pl_ok = ik >= 0 && ik < N;
                const bool pl_own = ik >= i_lo && ik < i_hi;
                const unsigned pl = pl_own ? (unsigned)ik * plane_bytes : 0u;
#pragma unroll
                for (int r = 0; r < JT; ++r) {
                    const double up = r > 0 ? wc[k][r > 0 ? r - 1 : 0] : eu[k];
                    const double dn = r < JT - 1 ? wc[k][r < JT - 1 ? r + 1 : r] : ed[k];
                    const double lf = lane_left(wc[k][r]);
                    const double rt = lane_right(wc[k][r]);
                    const double fK = st.eval(wm[k][r], wp[r], up, dn, lf, rt, wc[k][r]);
                    // K_k of the points this tile owns (planes [i_lo, i_hi) only)
                    if (f_nt) buf_st_p<2>(rk[k], pl_own ? so[r] : 0xffffffffu, pl, fK);
                    else buf_st(rk[k], pl_own ? so[r] : 0xffffffffu, pl, fK);
                    // K_k enters the sums of the later targets
#pragma unroll
                    for (int e = k; e < D; ++e) {
                        acc[e][k][r] = fma(w_ck[e][k], fK, acc[e][k][r]);
                        if (SOLERR && e == D - 1) acce[k][r] = fma(w_ek[k], fK, acce[k][r]);
                    }
                    // target k + 1 is complete for this plane
                    const double t = __dadd_rn(yf[k][r], __dmul_rn(ca.h, acc[k][k][r]));
                    nw[r] = (pl_ok && in[r]) ? t : 0.0;
                    if (k == D - 1) {
                        buf_st(rout, pl_own ? so[r] : 0xffffffffu, pl, t);
                        if (SOLERR && pl_own && so[r] != 0xffffffffu) {
                            const double er = __dmul_rn(ca.h, acce[k][r]);
                            const size_t e_ = (size_t)ik * (size_t)N * (size_t)N + so[r] / 8u;
                            local += ratio_sq1(er, yf[k][r], t, ca.red.atol_vec, ca.red.atol_s,
                                               ca.red.rtol, e_, ca.red.n_valid);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < JT; ++r) nw[r] = 0.0;
            }
            // T_k's window moves one plane on; the fresh values are the upper plane
            // of stage k + 1's window
#pragma unroll
            for (int r = 0; r < JT; ++r) {
                wm[k][r] = wc[k][r];
                wc[k][r] = wp[r];
                wp[r] = nw[r];
            }
        }
        // ---- every plane moves one stage on
#pragma unroll
        for (int r = 0; r < JT; ++r) {
#pragma unroll
            for (int k = D - 1; k >= 1; --k) {
                yf[k][r] = yf[k - 1][r];
                acce[k][r] = acce[k - 1][r];
#pragma unroll
                for (int e = k; e < D; ++e) acc[e][k][r] = acc[e][k - 1][r];
            }
        }
    }
#undef ESQ_CHAIN3D_LOAD
    if constexpr (SOLERR) block_partial_w<NW>(local, ca.red.partials);
}

// Shapes (rows per thread x waves per workgroup) by depth: the windows, the targets'
// travelling sums, the base-state delay line and one plane of operands in flight
// are (2 D + D (D + 1) / 2 + D [+ D] + NU + 2) JT doubles per thread.
// (round 6: the 1024-thread workgroup caps these kernels at 128 VGPRs -- the wide ones
// spill, chain3+solerr<9>: 108 scalar and 2 vector registers -- but 4 x 8 and 3 x 8, which
// do not, are 8-12 % slower at N = 159 and 5 % at N = 400: profiles/r06_experiments.md, 7)
template <int D>
struct Chain3dShape {
    static constexpr int JT = 2, NW = 16;
};

// Launch: (depth, kind, rows) -> the instantiation; the register budget per depth
// (rows read from memory) is the caller's ChainCaps3d.  r->rkc_planes > 0 forces
// the planes per tile (tests).
struct ChainCaps3d {
    int stage[ESQ_CHAIN_MAX_DEPTH + 1], solerr[ESQ_CHAIN_MAX_DEPTH + 1];
};
inline ChainCaps3d chain_caps3d() {
    //                      depth: 0  1  2  3  4  5  6 (7)
    ChainCaps3d c = {{0, 0, 9, 9, 6, -1, -1}, {0, 0, 9, 9, 0, -1, -1}};
    return c;
}

template <class St, int MAXD = 4>
int chain3d(const St &st, int N, const double *y_in, const esq_chain *chain, int planes,
            bool force, void *stream, void *start_event, void *stop_event) {
    if (!chain) return ESQ_EINVAL;
    if (chain->from_rows) return ESQ_ENOTSUP;
    // (not in 3-D: the FSAL end-point stage inside the chain, a solution/error target
    // that is not stored -- the plugin does not declare ESQ_CHAIN_CAP_PRE / _ERRNORM)
    if (chain->kind_last == ESQ_EPI_ERRNORM ||
        (chain->kind_last == ESQ_EPI_SOLERR && !chain->out))
        return ESQ_ENOTSUP;
    if (chain->depth < 2 || chain->depth > MAXD) return ESQ_ENOTSUP;
    // (32-bit byte offsets into a vector; grids below 48^3: a tile's run-in planes
    // and halo points outweigh the saving)
    if ((unsigned long long)N * N * N * 8ull > 0xffffffffull - 16ull) return ESQ_ENOTSUP;
    if (N < 2 || (N < 48 && !force)) return ESQ_ENOTSUP;
    const ChainCaps3d caps = chain_caps3d();
    const bool solerr = chain->kind_last == ESQ_EPI_SOLERR;
    if (chain->nu > (solerr ? caps.solerr[chain->depth] : caps.stage[chain->depth]))
        return ESQ_ENOTSUP;
    if (chain->dry_run) {
        // the query: the same answer without the launch -- and what the tile
        // geometry costs (one workgroup per CU: the planner's figure must not
        // depend on an occupancy query)
        if (chain->nu > ESQ_CHAIN_MAX_ROWS - 1 && chain->nu > 9) return ESQ_ENOTSUP;
        if (chain->read_amplification) {
            const int D = chain->depth;
            const int jt = D == 2 ? Chain3dShape<2>::JT : D == 3 ? Chain3dShape<3>::JT
                                                                 : Chain3dShape<4>::JT;
            const int nw = D == 2 ? Chain3dShape<2>::NW : D == 3 ? Chain3dShape<3>::NW
                                                                 : Chain3dShape<4>::NW;
            const Geo3d g = geo_rkc3d(N, D, jt, nw, device_cus(), planes);
            *chain->read_amplification = amp_rkc3d(g, D);
        }
    }
    int rc_launch = 0;
    const int rc = dispatch_chain<MAXD>(chain, [&](auto ca, auto kind, auto from_c) {
        using CA = decltype(ca);
        if constexpr (decltype(from_c)::value || CA::kD > MAXD ||
                      decltype(kind)::value == ESQ_EPI_ERRNORM) {
            rc_launch = ESQ_ENOTSUP;
        } else {
            constexpr int DD = CA::kD, JT = Chain3dShape<DD>::JT, NW = Chain3dShape<DD>::NW;
            auto kern = k_chain3d<DD, CA::kNU, JT, NW, decltype(kind)::value, St>;
            static std::atomic<int> per_cu_cache{0};
            int per_cu = per_cu_cache.load(std::memory_order_relaxed);
            if (per_cu == 0) {
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64 * NW, 0) !=
                        hipSuccess || per_cu < 1)
                    per_cu = 1;
                per_cu_cache.store(per_cu, std::memory_order_relaxed);
            }
            if (NW * JT - 2 * DD < 1) { rc_launch = ESQ_ENOTSUP; return; }
            const Geo3d g = geo_rkc3d(N, DD, JT, NW, device_cus() * per_cu, planes);
            if (decltype(kind)::value == ESQ_EPI_SOLERR) {
                if ((int)g.grid > chain->partials_cap) { rc_launch = ESQ_ENOTSUP; return; }
                if (chain->partials_used) *chain->partials_used = (int)g.grid;
            }
            if (chain->read_amplification) *chain->read_amplification = amp_rkc3d(g, DD);
            hipExtLaunchKernelGGL(kern, dim3(g.grid), dim3(64 * NW), 0, (hipStream_t)stream,
                                  (hipEvent_t)start_event, (hipEvent_t)stop_event, 0, y_in, ca,
                                  st, g);
        }
    });
    if (rc || chain->dry_run) return rc;
    return rc_launch ? rc_launch : (int)hipGetLastError();
}

}  // namespace esq
