// esq_terms.hpp -- types and 16-byte access helpers shared by the RK kernels
// (esq_kernels.hpp) and the RHS plugins (esq_rhs.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace esq {

constexpr int kMaxTerms = 20;     // >= longest coefficient row (Pr9: 16)
constexpr int kBlock = 256;       // 4 waves of 64 lanes
constexpr int kMaxPartials = 8192;

// One linear combination: up to kMaxTerms (pointer, coefficient) pairs, passed
// BY VALUE so that hipcc keeps them in SGPRs (s_load from the kernarg segment):
// the "A-row broadcast" costs no vector memory traffic at all.
struct Terms {
    const double *p[kMaxTerms];
    double c[kMaxTerms];
};
// two coefficient sets over one row list (solution weights b, error weights e)
struct Terms2 {
    const double *p[kMaxTerms];
    double b[kMaxTerms];
    double e[kMaxTerms];
};

__device__ __forceinline__ double2 ld2(const double *p, size_t i) {
    return reinterpret_cast<const double2 *>(p)[i];
}
__device__ __forceinline__ void st2(double *p, size_t i, double2 v) {
    reinterpret_cast<double2 *>(p)[i] = v;
}
// non-temporal (streaming, "nt") 16-byte accesses
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ld2_nt(const double *p, size_t i) {
    const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(p) + i);
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ void st2_nt(double *p, size_t i, double2 v) {
    v2d w;
    w.x = v.x;
    w.y = v.y;
    __builtin_nontemporal_store(w, reinterpret_cast<v2d *>(p) + i);
}

// Raw buffer resources (the whole vector as one buffer, 32-bit byte offsets): a lane
// that is masked out gets an out-of-range offset -- or the whole access a resource
// of zero bytes -- which the hardware's range check turns into "load returns 0,
// store is dropped" with no memory traffic.  Loads and stores of a marching loop
// can then be UNCONDITIONAL: no branch around them (the compiler otherwise wraps
// each load in a branch with an `s_waitcnt vmcnt(0)` behind it), and it counts the
// younger accesses it may leave in flight.  The scalar offset is not range-checked:
// always a valid row / plane.  Vectors of at most 4 GiB - 16 B.
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)(unsigned)bytes, 0x00020000);
}
__device__ __forceinline__ double buf_ld(rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void buf_st(rsrc_t r, unsigned voff, unsigned soff, double v) {
    using v2u = decltype(__builtin_amdgcn_raw_buffer_load_b64(r, 0, 0, 0));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), r, (int)voff, (int)soff, 0);
}

// The neighbouring lane's value: lane l <- lane l - 1 (lane 0 gets 0.0) / lane
// l + 1 (lane 63 gets 0.0) -- one `v_mov_b32_dpp wave_shr:1 / wave_shl:1` per
// half and nothing else.  `__shfl_up / __shfl_down(v, 1, 64)` compile to two
// `ds_bpermute_b32` each: the LDS queue's latency and an `s_waitcnt lgkmcnt` on
// the stencil's dependency chain.  Callers never use what lane 0 (63) receives
// from outside the wave.  ESQ_LANE_DPP=0 at compile time: the shuffles (A/B).
#ifndef ESQ_LANE_DPP
#define ESQ_LANE_DPP 1
#endif
__device__ __forceinline__ double lane_left(double v) {
#if ESQ_LANE_DPP
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
#else
    return __shfl_up(v, 1, 64);
#endif
}
__device__ __forceinline__ double lane_right(double v) {
#if ESQ_LANE_DPP
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x130, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
#else
    return __shfl_down(v, 1, 64);
#endif
}

}  // namespace esq
