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

}  // namespace esq
