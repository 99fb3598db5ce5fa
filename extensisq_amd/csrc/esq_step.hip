// esq_step.hip -- the explicit Runge-Kutta step on the device: tableau and
// blocked-accumulation plan, stage sweeps (RHS plugin + epilogue), solution and
// error norm, accept (extensisq/common.py:222-356).  Host orchestration only;
// the kernels are in esq_kernels.hpp / esq_epilogue.hpp / the RHS plugins.
#include <algorithm>
#include <functional>

#include "esq_internal.hpp"
#include "esq_chain.hpp"

using namespace esqi;

namespace {

// ---- blocked accumulation plan ------------------------------------------------
// words per element and step moved by the stage kernels (+ block kernels) for a
// set of column boundaries; returns -1 if a block needs too many outputs/rows
int plan_words(const std::vector<double> &A, int s, const std::vector<int> &bounds,
               std::vector<esq_ctx::Block> *out) {
    auto nz = [&](int i, int j) { return A[(size_t)i * s + j] != 0.0; };
    std::vector<char> has(s, 0);
    int total = 0, prev = 0;
    if (out) out->clear();
    for (int J : bounds) {
        esq_ctx::Block b;
        b.J = J;
        b.prev = prev;
        std::vector<char> col(s, 0);
        for (int i = J; i < s; ++i) {
            bool any = false;
            for (int j = prev; j < J; ++j)
                if (nz(i, j)) { any = true; col[j] = 1; }
            if (any) b.stages.push_back(i);
        }
        for (int j = prev; j < J; ++j)
            if (col[j]) b.cols.push_back(j);
        if ((int)b.stages.size() > kMaxOut || (int)b.cols.size() > kMaxTerms) return -1;
        if (b.stages.empty()) return -1;
        total += (int)b.cols.size();
        for (int i : b.stages) {
            total += 1 + (has[i] ? 1 : 0);
            has[i] = 1;
        }
        if (out) out->push_back(b);
        prev = J;
    }
    for (int i = 1; i < s; ++i) {
        int last = 0;
        for (int J : bounds)
            if (J <= i) last = J;
        int c = 2;
        for (int j = last; j < i; ++j) c += nz(i, j);
        if (last > 0 && has[i]) c += 1;
        total += c;
    }
    return total;
}

// leading parts of the sums of all later stages, one pass over the block's rows
// returns 0 on success; *made_ystage = true if the block also wrote the
// boundary stage's argument into YSTAGE
static int run_block(esq_ctx *c, const esq_ctx::Block &b, double h,
                     bool *made_ystage) {
    BlockArgs a;
    const int nt = (int)b.cols.size(), no = (int)b.stages.size();
    for (int j = 0; j < kMaxTerms; ++j) {
        a.p[j] = j < nt ? c->krow[c->kmap[b.cols[j]]] : nullptr;
        for (int o = 0; o < kMaxOut; ++o)
            a.w[j][o] = (j < nt && o < no)
                            ? c->A[(size_t)b.stages[o] * c->s + b.cols[j]] : 0.0;
    }
    double reads = nt;
    for (int o = 0; o < kMaxOut; ++o) {
        a.out[o] = o < no ? c->krow[b.out_vec[o]] : nullptr;
        a.init[o] = (o < no && b.in_vec[o] >= 0) ? c->krow[b.in_vec[o]] : nullptr;
        if (a.init[o]) reads += 1;
    }
    // the boundary stage J itself (always output 0 when it uses the block) has
    // no later column to add: write its argument y + h*sum straight to YSTAGE
    a.y = nullptr;
    a.h = h;
    double alg = 0.0;
    *made_ystage = false;
    if (no > 0 && b.stages[0] == b.J && c->stage_init[b.J] == b.out_vec[0] &&
        c->stage_from[b.J] == b.J) {
        a.y = c->y;
        a.out[0] = c->ystage;
        reads += 1;
        int nnz_all = 0;
        for (int j = 0; j < b.J; ++j) nnz_all += c->A[(size_t)b.J * c->s + j] != 0.0;
        alg = 8.0 * (nnz_all + 2) * (double)c->len;   // that stage's booking
        *made_ystage = true;
    }
    // algorithmic bytes: only the folded-in stage (the other partial sums are
    // booked on the stages they serve); moved bytes: its real traffic
    Prof p(c, ESQ_PROF_STAGE, "k_block_acc", nt, alg, false,
           8.0 * (reads + no) * (double)c->len);
    c->self_valid = false;
    return launch_block(c, a, nt, no, p);
}

// coefficient row of stage i as the stage kernels use it: columns below the
// stage's blocked-accumulation boundary are in its stored partial sum, column
// `skip` (if >= 0) comes from registers.  Returns the number of rows to read.
int stage_terms(esq_ctx *c, int i, int skip, Terms &tm, const double **init,
                int *nnz_all, double *c_skip) {
    *init = c->stage_init[i] >= 0 ? c->krow[c->stage_init[i]] : nullptr;
    int nt = 0;
    if (c_skip) *c_skip = 0.0;
    // the per-stage term lists were built once in esq_rk_set_tableau
    for (const Term &term : c->stage_terms[i]) {
        if (term.col == skip) { if (c_skip) *c_skip = term.c; continue; }
        if (nt >= kMaxTerms) return -1;
        tm.p[nt] = c->krow[c->kmap[term.col]];
        tm.c[nt] = term.c;
        ++nt;
    }
    for (int j = nt; j < kMaxTerms; ++j) { tm.p[j] = nullptr; tm.c[j] = 0.0; }
    *nnz_all = c->stage_nnz[i];
    return nt;
}

void epi_common(esq_ctx *c, esq_epilogue &e, int kind) {
    memset(&e, 0, sizeof(e));
    e.kind = kind;
    e.atol_vec = c->atol_is_vec ? c->atolv : nullptr;
    e.atol_s = c->atol_s;
    e.rtol = c->rtol;
    e.n_valid = c->n;
    e.partials = c->partials;
    e.partials_cap = kPartialsCap;
    e.partials_used = &c->red_count;
    e.is_complex = c->cplx ? 1 : 0;
}

// one fused sweep; returns 0, ESQ_ENOTSUP (caller falls back) or an error
int run_fused(esq_ctx *c, double t, const double *y_in, double *f_out,
              const esq_epilogue &e, Prof &p) {
    c->self_valid = false;     // a plugin kernel does not signal its own completion
    esq_epilogue q = e;
    if (c->detached) q.dry_run = 1;     // host-side dry run: the plugin's answer, no launch
    const int r = c->rhs_fused(c->rhs_user, t, y_in, f_out, &q, c->len,
                               (void *)c->stream, (void *)p.start(),
                               (void *)p.stop());
    if (r == ESQ_ENOTSUP) { p.cancel(); return r; }
    if (r != 0) { p.cancel(); return fail(c, ESQ_ERHS, "fused RHS entry returned %d", r); }
    return 0;
}
// the library itself found a fused form not applicable (distinct from a plugin's
// ESQ_ENOTSUP: the plugin was never asked)
constexpr int kNotApplicable = -1000;
bool may_fuse(const esq_ctx *c, int kind) {
    return c->rhs_fused && ((c->fuse_mask >> kind) & 1);
}

// A QUERY instead of a launch (build_plan): the sweep_* builders describe the
// launch exactly as they would make it, ask the plugin whether it would take it
// (esq_epilogue.dry_run / esq_chain.dry_run: nothing is enqueued, nothing written)
// and report the designed words per element.  An entry that has not declared the
// query capability is taken to accept: its refusals are learnt from launches.
struct Dry {
    double reads = 0.0, writes = 0.0;
    bool made = false;                 // block sweep: also wrote the boundary stage's argument
    double amp = 0.0;                  // chain: the plugin's own read amplification (0: not told)
};
int ask_fused(esq_ctx *c, double t, const double *y_in, double *f_out, esq_epilogue &e) {
    if (!(c->fuse_mask & ESQ_FUSE_QUERY)) return 0;
    e.dry_run = 1;
    const int r = c->rhs_fused(c->rhs_user, t, y_in, f_out, &e, c->len, (void *)c->stream,
                               nullptr, nullptr);
    return r == 0 ? 0 : ESQ_ENOTSUP;
}

// RHS sweep of stage i + the accumulate of stage nx = i + 1 (ESQ_EPI_STAGE).
// from_state (stage 1 only): the sweep forms its own input y + h*a_10*K[0] on
// the fly instead of reading YSTAGE (ESQ_FUSE_SRC)
int sweep_next_stage(esq_ctx *c, int i, double t, double h, bool from_state = false,
                     Dry *dry = nullptr) {
    const int nx = i + 1;
    esq_epilogue e;
    epi_common(c, e, ESQ_EPI_STAGE);
    Terms tm;
    int nnz_all = 0;
    const int nt = stage_terms(c, nx, i, tm, &e.init, &nnz_all, &e.c_self);
    if (nt < 0) return dry ? kNotApplicable : fail(c, ESQ_EINVAL, "too many terms");
    e.nt = nt;
    for (int j = 0; j < nt; ++j) { e.rows[j] = tm.p[j]; e.c[j] = tm.c[j]; }
    e.y = c->y;
    e.h = h;
    e.out = c->work;
    e.f_store_nt = c->epi_nt & 1;   // K_i is consumed from registers, not re-read soon
    if (from_state) {
        if (nt > 1 || e.init) return kNotApplicable;   // the library's reason, not the plugin's
        e.in_base = c->y;
        e.in_row = c->krow[c->kmap[0]];
        e.in_c = c->A[(size_t)c->s];                 // A[1][0]
        e.in_h = h;
        if (dry) {
            dry->reads = 2; dry->writes = 2;
            return ask_fused(c, t + c->C[i] * h, nullptr, c->krow[c->kmap[i]], e);
        }
        // booked: stage 1's accumulate (1 + 2 words) + the RHS + stage 2's
        // accumulate; moved: y and K[0] in (stage 2's row K[0] is the same
        // vector), K[1] and the argument of stage 2 out
        Prof p(c, ESQ_PROF_STAGE, "rhs1+stage", nt,
               8.0 * (3 + nnz_all + 4) * (double)c->len, false, 8.0 * 4 * (double)c->len);
        const int r = run_fused(c, t + c->C[i] * h, nullptr, c->krow[c->kmap[i]], e, p);
        if (r == 0) std::swap(c->ystage, c->work);
        return r;
    }
    if (dry) {
        dry->reads = nt + 2 + (e.init ? 1 : 0); dry->writes = 2;
        return ask_fused(c, t + c->C[i] * h, c->ystage, c->krow[c->kmap[i]], e);
    }
    // booked on the stage class: next stage's algorithmic bytes + the RHS's
    // 16 B; moved: ys_in, rows, init, y in; K[i], ys_out out
    Prof p(c, ESQ_PROF_STAGE, "rhs+stage", nt, 8.0 * (nnz_all + 4) * (double)c->len,
           false, 8.0 * (nt + 4 + (e.init ? 1 : 0)) * (double)c->len);
    const int r = run_fused(c, t + c->C[i] * h, c->ystage, c->krow[c->kmap[i]], e, p);
    if (r == 0) std::swap(c->ystage, c->work);   // double buffer
    return r;
}
// `depth` stages in one marching sweep (esq_rhs_chain_fn, esq_chain.hpp): the RHS
// sweeps of stages i .. i + depth - 1; the arguments of the later stages stay in
// registers.  what_last: 0 = the last target is the argument of stage i + depth,
// 1 = y_new of an FSAL pair (weights B), 2 = y_new + error partial sums
// (non-FSAL: weights B and E), 3 = the EARLY error estimate of BS5 / CFMR7osc
// (esq_rk_set_pre: the last target is y_pre with the estimate's weights -- also the
// argument of stage i + depth where the two coincide, else not stored -- and the
// estimate's partial sums; its sum is published right behind the sweep, nobody
// waits), 4 = FSAL pairs: the chain runs THROUGH the end of the step -- stages
// i .. s - 1, y_new, the end-point stage K_s = f(t + h, y_new), the error partial
// sums (depth counts the end-point stage: i + depth == s + 1).
// Returns 0, kNotApplicable / ESQ_ENOTSUP (the
// caller tries a shorter chain or single sweeps) or an error.
int sweep_chain(esq_ctx *c, int i, int depth, double t, double h, int what_last,
                bool lazy_rows = false, bool from_rows = false, bool skip_out = false,
                Dry *dry = nullptr) {
    const int s = c->s;
    if (depth < 2 || depth > ESQ_CHAIN_MAX_DEPTH || s > 62) return kNotApplicable;
    const bool pre = what_last == 3, through = what_last == 4;
    if (pre && (c->pre.rows != i + depth || i + depth >= s)) return kNotApplicable;
    if (through && (!c->fsal || i + depth != s + 1 || i < 1)) return kNotApplicable;
    // the early estimate's y_pre is stored only where it is the next stage's argument
    const bool pre_out = pre && c->pre.b_is_next && !skip_out;
    if (pre && !pre_out && !(c->chain_caps & ESQ_CHAIN_CAP_PRE)) return kNotApplicable;
    if (through && !(c->chain_caps & ESQ_CHAIN_CAP_ERRNORM)) return kNotApplicable;
    esq_chain e;
    memset(&e, 0, sizeof(e));
    e.depth = depth;
    e.kind_last = (what_last == 2 || pre) ? ESQ_EPI_SOLERR
                  : through               ? ESQ_EPI_ERRNORM
                                          : ESQ_EPI_STAGE;
    // per target: weights by COLUMN first (memory rows and chain members alike)
    double cw[ESQ_CHAIN_MAX_DEPTH][64] = {{0}}, ew[64] = {0};
    bool part[ESQ_CHAIN_MAX_DEPTH][64] = {{false}};
    bool use[64] = {false};
    const double *init[ESQ_CHAIN_MAX_DEPTH] = {nullptr};
    double alg = 0.0;
    int n_init = 0;
    for (int q = 0; q < depth; ++q) {            // target q + 1
        const int stage = i + q + 1;             // the stage this target feeds
        if (through && q == depth - 1) {
            // "target depth": the error sum over K_0 .. K_s (weights E where a stage
            // target has its row of A; K_s, the chain's last derivative, always part)
            int nz = 0;
            for (int j = 0; j <= s; ++j) {
                if (c->E[j] == 0.0 && j != s) continue;
                cw[q][j] = c->E[j];
                part[q][j] = true;
                ++nz;
            }
            alg += 8.0 * (nz + 3) * (double)c->len;
        } else if (pre && q == depth - 1) {
            int nz = 0;
            for (int j = 0; j < c->pre.rows; ++j) {
                const double bj = c->pre.b[j], ej = c->pre.e[j];
                const bool self = j == c->pre.rows - 1;      // the fresh row: always part
                if (bj == 0.0 && ej == 0.0 && !self) continue;
                cw[q][j] = bj;
                ew[j] = ej;
                part[q][j] = true;
                ++nz;
            }
            alg += 8.0 * (nz + 1) * (double)c->len;
        } else if (q + 1 < depth - (through ? 1 : 0) || what_last == 0) {
            for (const Term &term : c->stage_terms[stage]) {
                cw[q][term.col] = term.c;
                part[q][term.col] = true;
            }
            if (c->stage_init[stage] >= 0) {
                init[q] = c->krow[c->stage_init[stage]];
                ++n_init;
            }
            alg += 8.0 * (c->stage_nnz[stage] + 4) * (double)c->len;
        } else {
            int nz = 0;
            for (int j = 0; j < s; ++j) {
                // (y_new: the last target, or -- the chain goes through the end of
                // the step -- the argument of the end-point stage)
                const double bj = c->B[j], ej = what_last == 2 ? c->E[j] : 0.0;
                // the fresh last row is always part of the solution/error sums
                // (a zero weight contributes fma(0, v, s), as in EpiSolErr)
                const bool self = what_last == 2 && j == s - 1;
                if (bj == 0.0 && ej == 0.0 && !self) continue;
                cw[q][j] = bj;
                ew[j] = ej;
                part[q][j] = true;
                ++nz;
            }
            alg += 8.0 * (nz + 4) * (double)c->len;
        }
    }
    // columns i .. i + depth - 1 are the chain's own derivatives
    for (int q = 0; q < depth; ++q) {
        for (int k = 0; k < depth; ++k) {
            const int col = i + k;
            if (!part[q][col]) continue;
            if (k > q) return kNotApplicable;    // cannot happen for an explicit method
            e.ck[q][k] = cw[q][col];
            e.kmask[q] |= 1u << k;
            if (q == depth - 1) e.ek[k] = ew[col];
        }
        for (int j = 0; j < s; ++j)
            if (part[q][j] && (j < i || j >= i + depth)) use[j] = true;
    }
    // a chain through the end of the step that forms its own input may read rows for
    // that alone (BS5: a_61 K_1, which no solution or error weight touches): one row
    // more against the launch and the 7 + 1 words of the argument's own kernel
    if (from_rows && through && i >= 1)
        for (const Term &term : c->stage_terms[i]) use[term.col] = true;
    // memory rows: the leading partial sums first (each the start of its target's
    // chain: weight 1, fma(1, p, 0) == p), then the K rows by ascending column
    int nu = 0;
    for (int q = 0; q < depth; ++q) {
        if (!init[q]) continue;
        if (nu >= ESQ_CHAIN_MAX_ROWS) return kNotApplicable;
        e.rows[nu] = init[q];
        e.cu[q][nu] = 1.0;
        e.umask[q] |= 1u << nu;
        ++nu;
    }
    for (int j = 0; j < s; ++j) {
        if (!use[j]) continue;
        if (nu >= ESQ_CHAIN_MAX_ROWS) return kNotApplicable;
        e.rows[nu] = c->krow[c->kmap[j]];
        for (int q = 0; q < depth; ++q) {
            e.cu[q][nu] = cw[q][j];
            if (part[q][j]) e.umask[q] |= 1u << nu;
        }
        e.eu[nu] = ew[j];
        ++nu;
    }
    e.nu = nu;
    if (from_rows) {
        // the argument of stage i from the rows this chain reads anyway: only where it
        // needs no other row (no partial sum either) -- a pure saving
        if (i < 1 || c->stage_init[i] >= 0 || c->stage_from[i] != 0) return kNotApplicable;
        e.from_rows = 1;
        for (const Term &term : c->stage_terms[i]) {
            if (!use[term.col]) return kNotApplicable;
            int u = 0;
            for (; u < nu; ++u)
                if (e.rows[u] == c->krow[c->kmap[term.col]]) break;
            if (u == nu) return kNotApplicable;
            e.c0[u] = term.c;
            e.umask0 |= 1u << u;
        }
    }
    // i == 0: the chain starts from the state itself (stage 0 = f(t, y))
    e.y = i == 0 ? nullptr : c->y;
    e.h = h;
    unsigned long long skipped = 0;
    int n_stored = 0;
    for (int k = 0; k < depth; ++k) {
        e.t[k] = i + k < s ? t + c->C[i + k] * h : t + h;
        // lazy_rows: a derivative that nothing after this sweep reads -- no later
        // stage's row of A (the stage right behind the chain gets its whole
        // argument from the chain), no solution / error weight outside this sweep
        // -- is not written (restore_rows re-evaluates it for whoever asks)
        const int col = i + k;
        bool needed = !lazy_rows || col == s;        // (K_s is the next step's K_0)
        for (int st = i + depth + (what_last == 0 ? 1 : 0); st < s && !needed; ++st)
            needed = c->A[(size_t)st * s + col] != 0.0;
        if (!needed && what_last != 2 && !through)
            needed = c->B[col] != 0.0 || c->E[col] != 0.0;
        e.f_out[k] = needed ? c->krow[c->kmap[col]] : nullptr;
        if (!needed) skipped |= 1ull << col;
        n_stored += needed;
    }
    // skip_out: the next chain forms its own input (from_rows)
    e.out = what_last == 0 ? (skip_out ? nullptr : c->work)
            : pre          ? (pre_out ? c->work : nullptr)
                           : c->ynew;
    e.f_store_nt = c->epi_nt & 1;
    // loads: a row (and y) that nothing after this sweep reads again is streamed,
    // the others stay cacheable -- the next chain finds them in the Infinity Cache
    // (Pr8, n = 1e7: K_0, K_2..K_4 cacheable in the middle chain: 0.527 -> 0.509 ms)
    {
        unsigned m = (what_last == 2 || through) ? 2u : 0u;   // y: last read of the step
        for (int u = 0; u < nu; ++u) {
            int col = -1;
            for (int j = 0; j <= s && col < 0; ++j)
                if (j < c->n_rows && e.rows[u] == c->krow[c->kmap[j]]) col = j;
            bool later = false;                            // (a partial sum: read once)
            if (col >= 0 && col < s) {
                for (int st = i + depth + (what_last == 0 ? 1 : 0); st < s && !later; ++st)
                    later = c->A[(size_t)st * s + col] != 0.0;
                if (!later && what_last != 2 && !through)
                    later = c->B[col] != 0.0 || c->E[col] != 0.0;
            }
            if (!later) m |= 1u << (8 + u);
        }
        if (c->chain_ld_nt_set) {                          // ESQ_CHAIN_LDNT (tuning)
            const unsigned o = c->chain_ld_nt[i == 0 ? 0 : (what_last == 2 || through) ? 2 : 1];
            m = (o & 3u) | ((o & 4u) ? 0xffffff00u : 0u);
        }
        e.load_nt = (int)m;
    }
    e.atol_vec = c->atol_is_vec ? c->atolv : nullptr;
    e.atol_s = c->atol_s;
    e.rtol = c->rtol;
    e.n_valid = c->n;
    e.partials = c->partials;
    e.partials_cap = kPartialsCap;
    e.partials_used = &c->red_count;
    // booked: what the one-stage sweeps book; moved: input, y, rows, inits in
    // (times the tile geometry's read amplification); the chain's K rows and the
    // last target out
    // nu counts the partial sums too
    const double reads = ((i == 0 || from_rows) ? 1 : 2) + nu,
                 writes = n_stored + (e.out ? 1 : 0);
    (void)n_init;
    if (dry) {
        dry->reads = reads; dry->writes = writes;
        if (!(c->chain_caps & ESQ_CHAIN_CAP_QUERY)) return 0;
        int used = 0;
        double amp_q = 0.0;
        e.partials_used = &used;
        e.read_amplification = &amp_q;     // (a plugin may price its tile geometry itself)
        e.dry_run = 1;
        const int rq = c->rhs_chain(c->rhs_user, i == 0 ? c->y : c->ystage, &e, c->len,
                                    (void *)c->stream, nullptr, nullptr);
        dry->amp = amp_q;
        return rq == 0 ? 0 : ESQ_ENOTSUP;
    }
    double amp = 1.0;
    e.read_amplification = &amp;
    char label[24];
    snprintf(label, sizeof(label), "chain%d%s%s", depth,
             what_last == 2 ? "+solerr" : pre ? "+pre" : through ? "+errnorm" : "",
             n_stored == 0 ? "-K" : "");
    Prof p(c, ESQ_PROF_STAGE, label, nu, alg, false, 8.0 * (reads + writes) * (double)c->len);
    c->self_valid = false;
    if (c->detached) e.dry_run = 1;     // host-side dry run: the plugin's answer, no launch
    const int r = c->rhs_chain(c->rhs_user, i == 0 ? c->y : c->ystage, &e, c->len,
                               (void *)c->stream, (void *)p.start(), (void *)p.stop());
    // designed traffic incl. the halo rows / columns the plugin's tiles re-read
    if (p.on) {
        p.ev.moved = 8.0 * (reads * amp + writes) * (double)c->len;
        p.ev.floor = 8.0 * (reads + writes) * (double)c->len;
    }
    if (r == ESQ_ENOTSUP) { p.cancel(); return r; }
    if (r != 0) { p.cancel(); return fail(c, ESQ_ERHS, "chain RHS entry returned %d", r); }
    if ((what_last == 0 && !skip_out) || pre_out) std::swap(c->ystage, c->work);
    if (skipped) {
        c->missing_rows |= skipped;
        c->tail_missing = true;
        c->tail_accepted = false;
        c->tail_t = t;
        c->tail_h = h;
    }
    return 0;
}

// the early estimate by the library's own pass over the rows (k_pre_error: y_pre in
// registers); its sum goes to the estimate's slot, nobody waits
int pre_kernel(esq_ctx *c, double h, Dry *dry = nullptr) {
    int nz = 0;
    for (int j = 0; j < c->pre.rows; ++j) nz += c->pre.b[j] != 0.0 || c->pre.e[j] != 0.0;
    if (dry) { dry->reads = nz + 1; dry->writes = 0; return 0; }
    // (rows this step's chains left unwritten: only if the estimate reads one of them)
    bool need = c->k0_missing && (c->pre.b[0] != 0.0 || c->pre.e[0] != 0.0);
    for (int j = 0; j < c->pre.rows && c->tail_missing; ++j)
        need = need || (((c->missing_rows >> j) & 1ull) &&
                        (c->pre.b[j] != 0.0 || c->pre.e[j] != 0.0));
    if (need) {
        const int rr = esqi::restore_rows(c);
        if (rr) return rr;
    }
    Terms2 tm;
    const int nt = build_row_terms2(c, c->pre.b.data(), c->pre.rows, c->pre.e.data(),
                                    c->pre.rows, tm, c->kmap);
    if (nt < 1) return fail(c, ESQ_EINVAL, "bad weights");
    {
        Prof p(c, ESQ_PROF_SOLERR, "k_pre_error", nt, 8.0 * (nt + 1) * (double)c->len);
        const int r = launch_preerr(c, tm, nt, h, p);
        if (r) return r;
    }
    ++c->pre.plain;
    return publish_pre(c, c->partials, (int)c->grid_reduce);
}

bool may_use_src(const esq_ctx *c) {
    return c->src_pays && may_fuse(c, ESQ_EPI_STAGE) && (c->fuse_mask & ESQ_FUSE_SRC);
}

// RHS sweep of stage i = J - 1 + the blocked accumulation at boundary J with
// K_i as the block's last column (ESQ_EPI_BLOCK)
int sweep_block(esq_ctx *c, const esq_ctx::Block &b, int i, double t, double h,
                bool *made_ystage, Dry *dry = nullptr) {
    esq_epilogue e;
    epi_common(c, e, ESQ_EPI_BLOCK);
    const int no = (int)b.stages.size();
    int nt = 0;
    for (int col : b.cols) {
        if (col == i) continue;
        if (nt >= ESQ_EPI_MAX_ROWS) return ESQ_ENOTSUP;
        e.rows[nt] = c->krow[c->kmap[col]];
        for (int o = 0; o < no; ++o)
            e.w[nt][o] = c->A[(size_t)b.stages[o] * c->s + col];
        ++nt;
    }
    e.nt = nt;
    e.no = no;
    double reads = nt + 1;                       // rows + the sweep's input
    for (int o = 0; o < no; ++o) {
        e.w_self[o] = c->A[(size_t)b.stages[o] * c->s + i];
        e.out_o[o] = c->krow[b.out_vec[o]];
        e.init_o[o] = b.in_vec[o] >= 0 ? c->krow[b.in_vec[o]] : nullptr;
        if (e.init_o[o]) reads += 1;
    }
    e.h = h;
    double alg = 16.0 * (double)c->len;          // the RHS itself
    *made_ystage = false;
    if (no > 0 && b.stages[0] == b.J && c->stage_init[b.J] == b.out_vec[0] &&
        c->stage_from[b.J] == b.J) {
        e.y = c->y;
        e.out_o[0] = c->work;
        reads += 1;
        int nnz_all = 0;
        for (int j = 0; j < b.J; ++j) nnz_all += c->A[(size_t)b.J * c->s + j] != 0.0;
        alg += 8.0 * (nnz_all + 2) * (double)c->len;   // the boundary stage's booking
        *made_ystage = true;
    }
    e.f_store_nt = (c->epi_nt >> 1) & 1;
    if (dry) {
        dry->reads = reads; dry->writes = no + 1; dry->made = *made_ystage;
        return ask_fused(c, t + c->C[i] * h, c->ystage, c->krow[c->kmap[i]], e);
    }
    Prof p(c, ESQ_PROF_STAGE, "rhs+block", nt, alg, false,
           8.0 * (reads + no + 1) * (double)c->len);
    const int r = run_fused(c, t + c->C[i] * h, c->ystage, c->krow[c->kmap[i]], e, p);
    if (r == 0 && *made_ystage) std::swap(c->ystage, c->work);
    if (r != 0) *made_ystage = false;
    return r;
}

// FSAL pairs: RHS sweep of the last stage also forms y_new (ESQ_EPI_STAGE)
int sweep_ynew(esq_ctx *c, int i, double t, double h, Dry *dry = nullptr) {
    esq_epilogue e;
    epi_common(c, e, ESQ_EPI_STAGE);
    int nt = 0, nnz_all = 0;
    for (int j = 0; j < c->s; ++j) {
        if (c->B[j] == 0.0) continue;
        ++nnz_all;
        if (j == i) { e.c_self = c->B[j]; continue; }
        if (nt >= ESQ_EPI_MAX_ROWS) return ESQ_ENOTSUP;
        e.rows[nt] = c->krow[c->kmap[j]];
        e.c[nt] = c->B[j];
        ++nt;
    }
    e.nt = nt;
    e.y = c->y;
    e.h = h;
    e.out = c->ynew;
    e.f_store_nt = c->epi_nt & 1;
    if (dry) {
        dry->reads = nt + 2; dry->writes = 2;
        return ask_fused(c, t + c->C[i] * h, c->ystage, c->krow[c->kmap[i]], e);
    }
    Prof p(c, ESQ_PROF_STAGE, "rhs+stage", nt, 8.0 * (nnz_all + 4) * (double)c->len,
           false, 8.0 * (nt + 4) * (double)c->len);
    return run_fused(c, t + c->C[i] * h, c->ystage, c->krow[c->kmap[i]], e, p);
}

// non-FSAL pairs: RHS sweep of the last stage + y_new + error partial sums
int sweep_solerr(esq_ctx *c, int i, double t, double h, Dry *dry = nullptr) {
    esq_epilogue e;
    epi_common(c, e, ESQ_EPI_SOLERR);
    int nt = 0;
    for (int j = 0; j < c->s; ++j) {
        const double bj = c->B[j], ej = c->E[j];
        if (j == i) { e.c_self = bj; e.e_self = ej; continue; }
        if (bj == 0.0 && ej == 0.0) continue;
        if (nt >= ESQ_EPI_MAX_ROWS) return ESQ_ENOTSUP;
        e.rows[nt] = c->krow[c->kmap[j]];
        e.c[nt] = bj;
        e.e[nt] = ej;
        ++nt;
    }
    e.nt = nt;
    e.y = c->y;
    e.h = h;
    e.out = c->ynew;
    e.f_store_nt = (c->epi_nt >> 2) & 1;   // K_{s-1}: next read by the dense output
    if (dry) {
        int used = 0;
        e.partials_used = &used;
        dry->reads = nt + 2; dry->writes = 2;
        return ask_fused(c, t + c->C[i] * h, c->ystage, c->krow[c->kmap[i]], e);
    }
    // booked: the RHS's 16 B + the fused solution/error pass (rows incl. the
    // fresh one + y + y_new); moved: ys_in, rows, y in; K_i, y_new out
    Prof p(c, ESQ_PROF_SOLERR, "rhs+solerr", nt, 8.0 * (nt + 1 + 2 + 2) * (double)c->len,
           false, 8.0 * (nt + 4) * (double)c->len);
    return run_fused(c, t + c->C[i] * h, c->ystage, c->krow[c->kmap[i]], e, p);
}

// ---- cost of a plan in memory-side request units (one read word = 1, one
// written word = 2: profiles/r02_experiments.md) when the RHS plugin runs
// marching chain sweeps (esq_chain.hpp).  Mirrors the decomposition
// esq_rk_stages makes: the longest chain that crosses no boundary, else the
// block / stage / solution-error sweep.  Read words of a chain carry its halo
// factor; wide chains (many memory rows) run below the request-rate ceiling.
double plan_units_chained(const esq_ctx *c, const std::vector<int> &bounds) {
    const int s = c->s, depth = c->chain_depth;
    const std::vector<double> &A = c->A;
    auto nz = [&](int i, int j) { return A[(size_t)i * s + j] != 0.0; };
    static const double kHalo[7] = {1.0, 1.0, 1.10, 1.20, 1.31, 1.38, 1.46};
    // cost of a written word in read words; chains with more than six memory rows run
    // below the request-rate ceiling (profiles/r03_experiments.md §18)
    const double kW = 2.0, kWide = 1.12;
    std::vector<int> from(s, 0);
    std::vector<char> has_init(s, 0), cur(s, 0);
    struct Blk { int J, nt, no, ninit; std::vector<int> cols; };
    std::vector<Blk> blocks;
    int prev = 0;
    for (int J : bounds) {
        Blk b{J, 0, 0, 0, {}};
        std::vector<char> col(s, 0);
        for (int i = J; i < s; ++i) {
            bool any = false;
            for (int j = prev; j < J; ++j)
                if (nz(i, j)) { any = true; col[j] = 1; }
            if (any) { ++b.no; b.ninit += cur[i]; cur[i] = 1; }
        }
        for (int j = prev; j < J; ++j)
            if (col[j]) b.cols.push_back(j);
        // more than 8 rows besides the fresh column: the block epilogue is not
        // instantiated, the boundary would cost an unfused RHS + block kernel
        if (b.no == 0 || b.no > kMaxOut || (int)b.cols.size() > 9) return -1.0;
        blocks.push_back(b);
        prev = J;
    }
    for (int i = 1; i < s; ++i) {
        int last = 0;
        for (int J : bounds)
            if (J <= i) last = J;
        if (cur[i] && last > 0) { from[i] = last; has_init[i] = 1; }
    }
    auto rows_of = [&](int stage, int below, std::vector<char> &u) {
        for (int j = from[stage]; j < stage && j < below; ++j)
            if (nz(stage, j)) u[j] = 1;
    };
    double total = c->fsal ? 0.0 : 1.0 + 2.0 * kW;   // end-point sweep: 1 read, 2 writes
    int i = 1;
    while (i < s) {
        const Blk *bnext = nullptr;
        for (const auto &b : blocks)
            if (b.J == i + 1) bnext = &b;
        bool done = false;
        if (!bnext) {
            for (int D = depth; D >= 2 && !done; --D) {
                if (i + D > s) continue;
                bool crosses = false;
                for (const auto &b : blocks) crosses |= (b.J > i && b.J <= i + D);
                if (crosses) continue;
                const bool last_sol = i + D == s;
                std::vector<char> u(s, 0);
                int ninit = 0;
                for (int q = 0; q < D; ++q) {
                    const int st = i + q + 1;
                    if (q + 1 < D || !last_sol) {
                        rows_of(st, i, u);
                        ninit += has_init[st];
                    } else {
                        for (int j = 0; j < i; ++j)
                            if (c->B[j] != 0.0 || (!c->fsal && c->E[j] != 0.0)) u[j] = 1;
                    }
                }
                int nu = 0;
                for (int j = 0; j < s; ++j) nu += u[j];
                // what the built-in two-field plugin accepts (register budget);
                // the leading partial sums are memory rows of the chain too
                if (nu + ninit > ESQ_CHAIN_MAX_ROWS - 1 ||
                    !chain_within_caps(D, last_sol, nu + ninit))
                    continue;
                double units = (2 + nu + ninit) * kHalo[D] + kW * (D + 1);
                if (nu > 6) units *= kWide;
                // a two-stage chain moves its words a quarter slower than the deeper
                // ones (heat, n = 5e6: chain2<6> 8.4 us per 40 MB word, chain3<4> /
                // chain3<7> / chain4<4> 6.4-6.7): as many rows to walk, less to do
                // per row.  Pr9 then takes its boundary at J = 9 (0.550 ms/step)
                // instead of J = 10 (0.579)
                if (D == 2) units *= 1.25;
                total += units;
                i += D;
                done = true;
            }
        }
        if (done) continue;
        if (bnext) {
            int nt = 0;
            for (int col : bnext->cols) nt += col != i;
            total += (2 + nt + bnext->ninit) + kW * (1 + bnext->no);
        } else if (i == s - 1) {
            int nu = 0;
            for (int j = 0; j < s; ++j)
                nu += j != i && (c->B[j] != 0.0 || (!c->fsal && c->E[j] != 0.0));
            total += 2 + nu + 2.0 * kW;
        } else {
            int nu = 0;
            for (int j = from[i + 1]; j < i; ++j) nu += nz(i + 1, j);
            total += 2 + nu + has_init[i + 1] + 2.0 * kW;
        }
        ++i;
    }
    return total;
}

// ---------------------------------------------------------------------------
// THE STEP AS A PROGRAM.  esq_rk_stages(i_from, i_to) runs a launch list that is
// built once per key -- (stage range, is stage 1's argument in YSTAGE already, is
// K[0] still to be evaluated, may rows be left unwritten) -- and replayed every
// step.  build_plan walks the stages once and, for every launch it considers,
// asks the plugin's entry a side-effect-free question (Dry: the launch described
// exactly as it would be made, esq_chain.dry_run / esq_epilogue.dry_run).  The
// preferences, in order, per stage i:
//   * the longest marching chain from i that crosses no blocked-accumulation
//     boundary (i == 0: the deferred end-point derivative in front of the first
//     chain); a chain that ends the step, or starts it, in the form that makes its
//     own input from the rows it reads (the launch before it then does not write
//     that argument);
//   * stage 1 from the state (ESQ_FUSE_SRC) where the working set is cache-resident;
//   * the RHS sweep with the next stage's argument / the blocked accumulation /
//     y_new (+ error sums) as its epilogue;
//   * the plain RHS launch + the library's own kernels.
// A plugin entry WITHOUT the query capability is taken to accept; if a launch of
// it is refused at run time, the refusal is remembered (c->refused), the step is
// finished the plain way and the plans are rebuilt without that launch.
// ---------------------------------------------------------------------------
unsigned plan_key(int i_from, int i_to, bool ready, bool k0_missing, bool lazy) {
    return (unsigned)i_from | ((unsigned)i_to << 8) | (ready ? 1u << 16 : 0u) |
           (k0_missing ? 1u << 17 : 0u) | (lazy ? 1u << 18 : 0u);
}
// stage the early estimate is tested BEFORE (esq_rk_set_pre), for a whole step; 0: none
int pre_at(const esq_ctx *c, int i_from, int i_to) {
    return (c->pre.rows && i_from == 1 && i_to == c->s) ? c->pre.rows : 0;
}
unsigned long long step_signature(const PlanStep &st) {
    return (unsigned long long)st.op | ((unsigned long long)(unsigned char)st.i << 8) |
           ((unsigned long long)(unsigned char)st.depth << 16) |
           ((unsigned long long)(unsigned char)st.what << 24) |
           ((unsigned long long)st.from_rows << 32) | ((unsigned long long)st.skip_out << 33) |
           ((unsigned long long)st.lazy << 34);
}
const esq_ctx::Block *block_at(const esq_ctx *c, int J) {
    for (const auto &b : c->blocks)
        if (b.J == J) return &b;
    return nullptr;
}

// Cost of a launch in the planner's units: one 8-byte word per element read by a
// streaming kernel = 1.  Measured rates behind the constants (profiles/
// r03_experiments.md, r04_experiments.md): the chain sweeps read their halo rows
// and columns twice (kHalo by depth) and move a word ~15 % slower than the
// library's streaming kernels; a written word costs about a quarter more than a
// read one; two-stage chains walk as many rows for less work (+25 %); every launch
// costs a kernel boundary (~2 us, expressed in words of this context's size).
double step_cost(const esq_ctx *c, const PlanStep &st) {
    static const double kHalo[8] = {1.0, 1.0, 1.10, 1.20, 1.31, 1.38, 1.46, 1.55};
    const double kW = 1.25;
    const double word_us = (double)c->len_pad * 8.0 / 5.5e6;          // one word at 5.5 TB/s
    const double launch = 2.0 / (word_us > 1e-3 ? word_us : 1e-3);
    double u = st.reads + kW * st.writes;
    if (st.op == OP_CHAIN) {
        // (a plugin that prices its own tile geometry -- the 3-D sweeps, whose halo
        // grows with the surface of a tile -- is taken at its word)
        const double halo = st.amp > 0.0f ? (double)st.amp : kHalo[st.depth < 8 ? st.depth : 7];
        u = 1.15 * (st.reads * halo + kW * st.writes);
        if (st.depth == 2) u *= 1.25;
    }
    return u + launch;
}

// One way to evaluate stage i (and, as a chain, more): the launches, where the
// walk stands afterwards, what it cost.
struct PlanOption {
    std::vector<PlanStep> steps;
    int next = 0;
    bool ready = false, block_done = false, ynew = false, solerr = false;
    double cost = 0.0;
};

Plan build_plan(esq_ctx *c, int i_from, int i_to, bool ready0, bool k0_missing, bool lazy_ok) {
    const int s = c->s;
    const double t = 0.0, h = 1.0;             // queries do not depend on them
    const bool chains = c->rhs_chain && c->rhs_fused && !c->cplx && c->chain_depth >= 2;
    // whole steps of the pairs with an early estimate: it is tested before stage P
    const int P = pre_at(c, i_from, i_to);
    auto crosses = [&](int lo, int hi) {       // a boundary J with lo < J <= hi
        for (const auto &b : c->blocks)
            if (b.J > lo && b.J <= hi) return true;
        // (a chain of stages lo .. hi - 1 forms stage hi's argument: it may END at the
        // estimate -- hi == P, the estimate its last target -- not reach over it)
        if (P && lo < P && hi > P) return true;
        return false;
    };
    auto mk = [&](PlanOp op, int i, const Dry &d, int depth = 0, int what = 0,
                  bool lazy = false, bool from_rows = false, bool skip_out = false) {
        return PlanStep{(unsigned char)op, (signed char)i, (signed char)depth,
                        (signed char)what, lazy, from_rows, skip_out, (float)d.reads,
                        (float)d.writes, (float)d.amp};
    };
    auto refused = [&](const PlanStep &st) { return c->refused.count(step_signature(st)) != 0; };
    std::map<unsigned long long, std::pair<int, Dry>> asked;     // chain queries, memoised
    auto ask_chain = [&](int i, int D, int what, bool lazy, bool from_rows, bool skip_out,
                         Dry &d) {
        Dry none;
        const PlanStep st = mk(OP_CHAIN, i, none, D, what, lazy, from_rows, skip_out);
        if (refused(st)) return false;
        const unsigned long long sig = step_signature(st);
        auto it = asked.find(sig);
        if (it == asked.end()) {
            Dry q;
            const int r = sweep_chain(c, i, D, t, h, what, lazy, from_rows, skip_out, &q);
            if (c->plan_debug)
                fprintf(stderr, "[esq plan] chain(i=%d, D=%d, what=%d%s%s%s) -> %d  words %g+%g\n",
                        i, D, what, lazy ? ", lazy" : "", from_rows ? ", from rows" : "",
                        skip_out ? ", no out" : "", r, q.reads, q.writes);
            it = asked.emplace(sig, std::make_pair(r, q)).first;
        }
        d = it->second.second;
        return it->second.first == 0;
    };
    // what a chain of depth D from stage i ends in: -1 = no such chain
    auto last_kind = [&](int i, int D) {
        if (i + D == s && i_to == s)
            return c->fsal ? (may_fuse(c, ESQ_EPI_STAGE) ? 1 : -1)
                           : (may_fuse(c, ESQ_EPI_SOLERR) ? 2 : -1);
        // (the chain that evaluates the stage before the early estimate carries it)
        if (P && i + D == P) return may_fuse(c, ESQ_EPI_SOLERR) ? 3 : -1;
        return (i + D < i_to && may_fuse(c, ESQ_EPI_STAGE)) ? 0 : -1;
    };
    // FSAL pairs: a chain may run THROUGH the end of the step (y_new, the end-point
    // stage, the error sums; what = 4, depth counts the end-point stage)
    const bool through_cap = c->fsal && i_to == s && (c->chain_caps & ESQ_CHAIN_CAP_ERRNORM) &&
                             may_fuse(c, ESQ_EPI_ERRNORM) && may_fuse(c, ESQ_EPI_STAGE);
    const bool lazy = lazy_ok && c->rhs && i_to == s && (c->chain_caps & ESQ_CHAIN_CAP_SKIP_ROWS);
    const bool from_cap = c->chain_from_rows && (c->chain_caps & ESQ_CHAIN_CAP_FROM_ROWS);
    const bool skip_cap = from_cap && (c->chain_caps & ESQ_CHAIN_CAP_SKIP_OUT) && i_to == s;
    // the argument of stage i by the library's own kernel (after a block sweep that
    // left only the partial sums: the stage kernel alone)
    auto argument = [&](int i, bool block_done) {
        Dry d;
        const double *init = nullptr;
        Terms tm;
        int nnz_all = 0;
        const int nt = stage_terms(c, i, -1, tm, &init, &nnz_all, nullptr);
        d.reads = (nt < 0 ? 0 : nt) + 1 + (init ? 1 : 0);
        d.writes = 1;
        // (a blocked-accumulation boundary at i: the block kernel's rows and outputs)
        if (!block_done)
            if (const esq_ctx::Block *b = block_at(c, i)) {
                d.reads += (double)b->cols.size();
                d.writes += (double)b->stages.size();
            }
        return mk(block_done ? OP_LINCOMB : OP_ACCUM, i, d);
    };
    // ---- every way to go on from stage i, in order of preference
    auto options = [&](int i, bool ready, bool block_done, bool first) {
        std::vector<PlanOption> out;
        const esq_ctx::Block *bnext = block_at(c, i + 1);
        auto push = [&](std::vector<PlanStep> steps, int next, bool rdy, bool bd = false,
                        bool yn = false, bool se = false) {
            PlanOption o;
            // stage P - 1 is done: the early estimate, by the library's own pass unless
            // the chain carried it
            if (P && i < P && next == P &&
                !(steps.back().op == OP_CHAIN && steps.back().what == 3)) {
                Dry dp;
                pre_kernel(c, h, &dp);
                steps.push_back(mk(OP_PRE_KERNEL, P, dp));
            }
            o.steps = std::move(steps);
            o.next = next; o.ready = rdy; o.block_done = bd; o.ynew = yn; o.solerr = se;
            for (const PlanStep &st : o.steps) o.cost += step_cost(c, st);
            out.push_back(std::move(o));
        };
        // the first chain of a step can start from the state: stage 1's argument
        // y + h*a_10*K_0 from the K_0 it reads anyway (one stage more than the
        // plan's depth, as with the fused end-point stage)
        const bool first_from = first && i == 1 && !ready && chains && from_cap && i_to == s &&
                                !block_at(c, 2);
        if (chains && !bnext && through_cap && i >= (P ? P : 1)) {
            // through the end of the step: stages i .. s - 1 and the end-point stage,
            // one stage deeper than the plan's depth (as with the deferred end-point
            // derivative in front of a first chain)
            const int D = s + 1 - i;
            int d_max = c->chain_depth + 1 + (first_from ? 1 : 0);
            if (d_max > ESQ_CHAIN_MAX_DEPTH) d_max = ESQ_CHAIN_MAX_DEPTH;
            if (D >= 2 && D <= d_max && !crosses(i, s)) {
                Dry d;
                if (from_cap && ask_chain(i, D, 4, lazy, true, false, d))
                    push({mk(OP_CHAIN, i, d, D, 4, lazy, true, false)}, s, false, false, true,
                         true);
                if (D <= c->chain_depth + 1 && ask_chain(i, D, 4, lazy, false, false, d)) {
                    std::vector<PlanStep> st;
                    if (!ready) st.push_back(argument(i, block_done));
                    st.push_back(mk(OP_CHAIN, i, d, D, 4, lazy, false, false));
                    push(std::move(st), s, false, false, true, true);
                }
            }
        }
        if (chains && i + 1 < i_to && !bnext) {
            const int d_top = first_from && c->chain_depth < ESQ_CHAIN_MAX_DEPTH
                                  ? c->chain_depth + 1 : c->chain_depth;
            for (int D = d_top; D >= 2; --D) {
                if (i + D > i_to || crosses(i, i + D)) continue;
                const int what = last_kind(i, D);
                if (what < 0) continue;
                // a chain that hands over to one that forms its own input need not
                // write its last target
                const bool hands_on = what == 0 || (what == 3 && c->pre.b_is_next);
                for (int skip = (hands_on && skip_cap) ? 1 : 0; skip >= 0; --skip) {
                    // a chain that ends the step, or the first one of a step: its
                    // input from the rows it reads anyway
                    // ... any chain whose first argument is a combination of rows it
                    // reads anyway (sweep_chain decides; the launch before it can then
                    // leave that argument unwritten)
                    Dry d;
                    if (from_cap && (i_to == s || first_from) &&
                        ask_chain(i, D, what, lazy, true, skip != 0, d))
                        push({mk(OP_CHAIN, i, d, D, what, lazy, true, skip != 0)}, i + D,
                             hands_on && !skip, false, what == 1 || what == 2, what == 2);
                    if (D > c->chain_depth) continue;      // (that depth: from rows only)
                    if (ask_chain(i, D, what, lazy, false, skip != 0, d)) {
                        std::vector<PlanStep> st;
                        if (!ready) st.push_back(argument(i, block_done));
                        st.push_back(mk(OP_CHAIN, i, d, D, what, lazy, false, skip != 0));
                        push(std::move(st), i + D, hands_on && !skip, false,
                             what == 1 || what == 2, what == 2);
                    }
                }
            }
        }
        // stage 1 from the state (no stage-1 kernel, no argument in memory)
        if (i == 1 && !ready && i + 1 < i_to && may_use_src(c) && !block_at(c, 2)) {
            Dry d;
            const PlanStep st = mk(OP_SRC_STAGE, 1, d);
            if (!refused(st) && sweep_next_stage(c, 1, t, h, true, &d) == 0) {
                push({mk(OP_SRC_STAGE, 1, d)}, 2, true);
            }
        }
        auto with_arg = [&](PlanStep st) {
            std::vector<PlanStep> v;
            if (!ready) v.push_back(argument(i, block_done));
            v.push_back(st);
            return v;
        };
        if (i + 1 < i_to && !bnext && may_fuse(c, ESQ_EPI_STAGE)) {
            // this stage's RHS sweep also forms the NEXT stage's argument
            Dry d;
            if (!refused(mk(OP_STAGE_SWEEP, i, d)) && sweep_next_stage(c, i, t, h, false, &d) == 0)
                push(with_arg(mk(OP_STAGE_SWEEP, i, d)), i + 1, true);
        }
        if (i + 1 < i_to && bnext && may_fuse(c, ESQ_EPI_BLOCK)) {
            // ... or runs the blocked accumulation at the column boundary
            Dry d;
            bool made = false;
            if (!refused(mk(OP_BLOCK_SWEEP, i, d)) &&
                sweep_block(c, *bnext, i, t, h, &made, &d) == 0)
                push(with_arg(mk(OP_BLOCK_SWEEP, i, d)), i + 1, d.made, !d.made);
        }
        if (i == s - 1 && i_to == s) {
            Dry d;
            if (c->fsal && may_fuse(c, ESQ_EPI_STAGE) && !refused(mk(OP_YNEW_SWEEP, i, d)) &&
                sweep_ynew(c, i, t, h, &d) == 0)         // FSAL: ... also forms y_new
                push(with_arg(mk(OP_YNEW_SWEEP, i, d)), i + 1, false, false, true);
            if (!c->fsal && may_fuse(c, ESQ_EPI_SOLERR) && !refused(mk(OP_SOLERR_SWEEP, i, d)) &&
                sweep_solerr(c, i, t, h, &d) == 0)       // others: y_new + error sums
                push(with_arg(mk(OP_SOLERR_SWEEP, i, d)), i + 1, false, false, true, true);
        }
        {
            Dry d;
            d.reads = 1; d.writes = 1;
            push(with_arg(mk(OP_RHS, i, d)), i + 1, false);
        }
        return out;
    };
    // ---- what esq_rk_solution_error still has to do after the last launch
    auto tail_cost = [&](bool ynew, bool solerr) {
        if (i_to != s) return 0.0;
        int nb = 0, ne = 0;
        for (int j = 0; j < s; ++j) { nb += c->B[j] != 0.0; ne += (c->B[j] != 0.0 || c->E[j] != 0.0); }
        Dry d;
        if (c->fsal) { d.reads = ynew ? 0 : nb + 1; d.writes = ynew ? 0 : 1; }
        else { d.reads = solerr ? 0 : ne + 1; d.writes = solerr ? 0 : 1; }
        double cost = d.reads > 0 ? step_cost(c, mk(OP_ACCUM, s, d)) : 0.0;
        if (c->fsal && !solerr && through_cap) {
            // the end-point sweep with the error norm (esq_rk_solution_error): the same
            // for every plan but those that run through the end of the step
            Dry de;
            for (int j = 0; j < s; ++j) de.reads += c->E[j] != 0.0;
            de.reads += 2; de.writes = 1;
            cost += step_cost(c, mk(OP_ACCUM, s, de));
        }
        return cost;
    };
    // ---- cheapest sequence from (i, ready, block_done) to the end of the range
    struct Best { double cost = -1.0; int pick = -1; };
    std::map<unsigned, Best> memo;
    std::map<unsigned, std::vector<PlanOption>> opts;
    auto key_of = [](int i, bool ready, bool bd, bool first) {
        return (unsigned)i | (ready ? 256u : 0u) | (bd ? 512u : 0u) | (first ? 1024u : 0u);
    };
    std::function<double(int, bool, bool, bool, bool, bool)> solve =
        [&](int i, bool ready, bool bd, bool first, bool yn, bool se) -> double {
        if (i >= i_to) return tail_cost(yn, se);
        const unsigned k = key_of(i, ready, bd, first);
        auto it = memo.find(k);
        if (it != memo.end()) return it->second.cost;
        std::vector<PlanOption> &o = opts[k];
        o = options(i, ready, bd, first);
        Best best;
        for (int q = 0; q < (int)o.size(); ++q) {
            const double cst = o[q].cost +
                               solve(o[q].next, o[q].ready, o[q].block_done, false, o[q].ynew,
                                     o[q].solerr);
            if (best.pick < 0 || cst < best.cost) { best.cost = cst; best.pick = q; }
        }
        memo[k] = best;
        return best.cost;
    };
    Plan plan;
    auto walk = [&](int i, bool ready, bool bd, bool first) {
        while (i < i_to) {
            const unsigned k = key_of(i, ready, bd, first);
            const PlanOption &o = opts[k][memo[k].pick];
            for (const PlanStep &st : o.steps) plan.steps.push_back(st);
            if (o.ynew) plan.ynew_ready = true;
            if (o.solerr) plan.solerr_ready = true;
            i = o.next; ready = o.ready; bd = o.block_done; first = false;
        }
    };
    if (!k0_missing) {
        solve(i_from, ready0, false, true, false, false);
        walk(i_from, ready0, false, true);
        return plan;
    }
    // f(t, y) was left to this step (esq_rk_accept): one stage more in front of the
    // first chain -- stage 0 reads the state itself, its argument never existed,
    // K[0] is written once and not read back -- or a launch of its own
    Dry dk;
    dk.reads = 1; dk.writes = 1;
    const PlanStep k0 = mk(OP_RHS_K0, 0, dk);
    double best = step_cost(c, k0) + solve(i_from, false, false, true, false, false);
    int best_D = 0, best_what = 0;
    bool best_skip = false;
    Dry best_d;
    if (i_from == 1 && chains && may_fuse(c, ESQ_EPI_STAGE) &&
        (c->chain_caps & ESQ_CHAIN_CAP_FROM_STATE)) {
        for (int D = c->chain_depth; D >= 2; --D) {
            if (1 + D >= i_to || crosses(0, 1 + D) || D + 1 > ESQ_CHAIN_MAX_DEPTH) continue;
            const int what0 = (P && 1 + D == P) ? 3 : 0;
            if (what0 == 3 && !may_fuse(c, ESQ_EPI_SOLERR)) continue;
            const bool hands_on = what0 == 0 || c->pre.b_is_next;
            // (its last target unwritten where the chain behind it forms its own input)
            for (int skip = (hands_on && skip_cap) ? 1 : 0; skip >= 0; --skip) {
                Dry d;
                if (!ask_chain(0, D + 1, what0, lazy, false, skip != 0, d)) continue;
                const double cst = step_cost(c, mk(OP_CHAIN, 0, d, D + 1, what0, lazy, false, skip != 0)) +
                                   solve(D + 1, hands_on && !skip, false, false, false, false);
                if (cst < best) {
                    best = cst; best_D = D; best_d = d; best_skip = skip != 0; best_what = what0;
                }
            }
        }
    }
    if (best_D) {
        plan.steps.push_back(mk(OP_CHAIN, 0, best_d, best_D + 1, best_what, lazy, false, best_skip));
        walk(best_D + 1, (best_what == 0 || c->pre.b_is_next) && !best_skip, false, false);
    } else {
        plan.steps.push_back(k0);
        walk(i_from, false, false, true);
    }
    return plan;
}

const Plan &get_plan(esq_ctx *c, int i_from, int i_to, bool ready, bool k0_missing) {
    const bool lazy_ok = c->lazy_rows && !c->keep_rows;
    const unsigned key = plan_key(i_from, i_to, ready, k0_missing, lazy_ok);
    auto it = c->plans.find(key);
    if (it == c->plans.end())
        it = c->plans.emplace(key, build_plan(c, i_from, i_to, ready, k0_missing, lazy_ok)).first;
    return it->second;
}

// argument of stage i from the rows in memory (after a block sweep that left only
// the partial sums): the stage kernel alone
int lincomb_stage(esq_ctx *c, int i, double h) {
    const double *init = nullptr;
    Terms tm;
    int nnz_all = 0;
    const int nt = stage_terms(c, i, -1, tm, &init, &nnz_all, nullptr);
    if (nt < 0) return fail(c, ESQ_EINVAL, "too many terms");
    Prof p(c, ESQ_PROF_STAGE, "k_lincomb", nt, 8.0 * (nnz_all + 2) * (double)c->len, false,
           8.0 * (nt + 2 + (init ? 1 : 0)) * (double)c->len);
    return launch_lincomb(c, c->ystage, c->y, tm, nt, h, &p, init);
}

// one entry of a plan; ESQ_ENOTSUP / kNotApplicable: the plugin refused at run time
int run_step(esq_ctx *c, const PlanStep &st, double t, double h) {
    const int i = st.i;
    switch (st.op) {
        case OP_RHS_K0: {
            const int r = call_rhs(c, t, c->y, c->krow[c->kmap[0]]);
            if (r == 0) { ++c->end_plain; c->k0_missing = false; }
            return r;
        }
        case OP_CHAIN: {
            int r = sweep_chain(c, i, st.depth, t, h, st.what, st.lazy, st.from_rows,
                                st.skip_out);
            if (r == 0 && i == 0) { ++c->end_fused; c->k0_missing = false; }
            // the early estimate rode on the sweep: its sum to the estimate's slot
            if (r == 0 && st.what == 3) {
                ++c->pre.fused;
                r = publish_pre(c, c->partials, c->red_count);
            }
            return r;
        }
        case OP_PRE_KERNEL: return pre_kernel(c, h);
        case OP_SRC_STAGE: return sweep_next_stage(c, 1, t, h, true);
        case OP_ACCUM: return esq_rk_stage_accumulate(c, i, h);
        case OP_LINCOMB: return lincomb_stage(c, i, h);
        case OP_STAGE_SWEEP: return sweep_next_stage(c, i, t, h);
        case OP_BLOCK_SWEEP: {
            bool made = false;
            const esq_ctx::Block *b = block_at(c, i + 1);
            return b ? sweep_block(c, *b, i, t, h, &made) : kNotApplicable;
        }
        case OP_YNEW_SWEEP: return sweep_ynew(c, i, t, h);
        case OP_SOLERR_SWEEP: return sweep_solerr(c, i, t, h);
        default: return call_rhs(c, t + c->C[i] * h, c->ystage, c->krow[c->kmap[i]]);
    }
}

// would esq_rk_accept leave f(t_new, y_new) to the next step's first chain?
bool defers_end_point(esq_ctx *c) {
    if (c->fsal || !c->lazy_end || !c->rhs || !c->rhs_chain || !c->rhs_fused || c->cplx ||
        !(c->chain_caps & ESQ_CHAIN_CAP_FROM_STATE) || c->chain_depth < 2 ||
        !may_fuse(c, ESQ_EPI_STAGE))
        return false;
    const Plan &next = get_plan(c, 1, c->s, false, true);
    return !next.steps.empty() && next.steps[0].op == OP_CHAIN && next.steps[0].i == 0;
}

// The first launch of the step that FOLLOWS the one in flight, enqueued now: the
// context is put into the state esq_rk_accept would leave it in (y <-> y_new, the
// row map rotated, the rows the sweep writes mapped to spare physical rows), the
// first entry of that step's program runs, and the context is put back.  What
// the launch produced is remembered in c->ahead; esq_rk_accept(t_new, ., h) with
// the same (t_new, h) makes it real.  Only a chain sweep is launched ahead (it
// writes nothing but its own K rows and the stage-argument buffer).
void launch_ahead(esq_ctx *c, double t_new, double h) {
    c->ahead.valid = c->ahead.committed = false;
    if (!c->ahead_on || !c->have_tab || !c->rhs || c->cplx || c->host_slab || h == 0.0) return;
    const bool k0_next = defers_end_point(c);
    // (a pair that is not FSAL and does not leave f(t_new, y_new) to the chain has
    // that derivative evaluated by esq_rk_accept: nothing can run ahead of it)
    if (!c->fsal && !k0_next) return;
    const bool lazy_ok = c->lazy_rows && !c->keep_rows;
    const Plan &plan = get_plan(c, 1, c->s, false, k0_next);
    // a chain sweep, or stage 1 from the state: launches that write nothing but K
    // rows of the new step and the stage-argument buffer
    if (plan.steps.empty() ||
        (plan.steps[0].op != OP_CHAIN && plan.steps[0].op != OP_SRC_STAGE))
        return;
    const PlanStep st = plan.steps[0];
    const int n_cols = st.op == OP_CHAIN ? st.depth : 1;
    if ((int)c->spare_rows.size() < ESQ_CHAIN_MAX_DEPTH || !c->spare_vec) {
        int first = 0;
        if (esq_aux_rows(c, ESQ_CHAIN_MAX_DEPTH + 1, &first) != 0) return;
        c->spare_rows.clear();
        for (int k = 0; k < ESQ_CHAIN_MAX_DEPTH; ++k) c->spare_rows.push_back(first + k);
        c->spare_vec = c->krow[first + ESQ_CHAIN_MAX_DEPTH];
        c->ahead.valid = c->ahead.committed = false;        // (esq_aux_rows went through ENTER)
    }
    // ---- the state after an accept
    esq_ctx::Ahead &a = c->ahead;
    const std::vector<int> kmap_now = c->kmap;
    a.kmap = c->kmap;
    std::swap(a.kmap[0], a.kmap[c->s]);
    a.spares.clear();
    int used = 0;
    for (int k = 0; k < n_cols; ++k) {
        const int col = st.i + k;
        if (col == 0) continue;              // K[0] of the new step: the free K[s] slot
        a.spares.push_back(a.kmap[col]);     // (this step's row: a spare once accepted)
        a.kmap[col] = c->spare_rows[used++];
    }
    for (size_t q = (size_t)used; q < c->spare_rows.size(); ++q) a.spares.push_back(c->spare_rows[q]);
    // everything a launch may change besides device memory, by assignment
    const StepState sv = *c;
    c->kmap = a.kmap;
    // (the new step's y_new buffer: NOT the old state -- the attempt may be rejected,
    // and after an accept the old state is what the interpolant starts from)
    c->y = sv.ynew;
    c->ynew = c->spare_vec;
    c->tail_missing = false;
    c->missing_rows = 0;
    c->k0_missing = k0_next;
    const int r = run_step(c, st, t_new, h);
    // ---- what it left behind, and the context as it was
    a.ystage = c->ystage; a.work = c->work;
    a.tail_missing = c->tail_missing; a.missing_rows = c->missing_rows;
    a.k0_done = k0_next && !c->k0_missing;
    a.wrote_ynew = st.op == OP_CHAIN && (st.what == 1 || st.what == 2 || st.what == 4);
    // (a first launch that IS the step -- Ts5's chain through the end -- leaves the
    // error partial sums behind; one that carries the early estimate has published it)
    a.red_count = c->red_count;
    a.pre_seq = (st.op == OP_CHAIN && st.what == 3) ? c->pre_last_seq : 0;
    c->kmap = kmap_now;
    static_cast<StepState &>(*c) = sv;
    if (r != 0) {                        // refused at run time: the plans learn it
        if (r == ESQ_ENOTSUP || r == kNotApplicable) {
            c->refused.insert(step_signature(st));
            c->plans.clear();
        }
        return;
    }
    a.valid = true;
    a.t = t_new; a.h = h;
    a.key = plan_key(1, c->s, false, k0_next, lazy_ok);
    a.k0_next = k0_next;
}

}  // namespace

void esqi::drop_plans(esq_ctx *c) {
    c->plans.clear();
    c->refused.clear();
    c->ahead.valid = c->ahead.committed = false;
}

bool esqi::launch_ahead_if_asked(esq_ctx *c) {
    const double h = c->ahead_ask_h, t = c->ahead_ask_t;
    c->ahead_ask_h = 0.0;
    if (h != 0.0) launch_ahead(c, t, h);
    const bool rkc = rkc_launch_ahead_if_asked(c);
    return (h != 0.0 && c->ahead.valid) || rkc;
}

// The rows `missing_rows` of the step in flight (or of the step just accepted) exist
// only as terms of the sums their own sweep formed.  Re-evaluate
// them the plain way, stage by stage: a_i. K in ascending column order from the
// rows in memory (what k_lincomb does when no partial sum is stored: the same FMA
// chain as the blocked / chained sweeps), then the RHS.  A context that is asked
// again within a few steps (dense output on every step) keeps its rows from then on.
int esqi::restore_rows(esq_ctx *c) {
    if (!c->tail_missing && !c->k0_missing) return 0;
    (void)hipSetDevice(c->device);
    c->idle = false;
    c->self_valid = false;
    // (WORK is scratch below: a first launch made ahead of time has its result there)
    if (c->ahead.valid || c->ahead.committed) ++c->ahead_dropped;
    c->ahead.valid = c->ahead.committed = false;
    // (a flag is cleared only once its rows are in memory: after a failure the
    // next reader tries again instead of reading what is not there)
    if (c->k0_missing) {
        // f(t, y) of the current state (= K[s] of the step just accepted)
        const int r = call_rhs(c, c->k0_t, c->y, c->krow[c->kmap[0]]);
        if (r) return r;
        c->k0_missing = false;
        ++c->end_plain;
    }
    if (!c->tail_missing) return 0;
    const std::vector<int> &map = c->tail_accepted ? c->kmap_last : c->kmap;
    // after esq_rk_accept the pre-step state is in the YNEW slot
    const double *base = c->tail_accepted ? c->ynew : c->y;
    for (int st = 1; st < c->s; ++st) {             // ascending: a row may need an earlier one
        if (!((c->missing_rows >> st) & 1ull)) continue;
        Terms tm;
        const int nt = build_row_terms(c, &c->A[(size_t)st * c->s], st, tm, map);
        if (nt < 0) return fail(c, ESQ_EINVAL, "too many terms");
        int r = launch_lincomb(c, c->work, base, tm, nt, c->tail_h);
        if (r) return r;
        r = call_rhs(c, c->tail_t + c->C[st] * c->tail_h, c->work, c->krow[map[st]]);
        if (r) return r;
    }
    c->tail_missing = false;
    c->missing_rows = 0;
    ++c->restores;
    if (c->accepted_steps - c->last_restore_at <= 4) c->keep_rows = true;
    c->last_restore_at = c->accepted_steps;
    return 0;
}

// is the early estimate's y_pre the argument of the stage that follows it (CFMR7osc:
// calvo.py:257)?  Depends on the blocked-accumulation plan: with a boundary, stage
// `rows` resumes a stored partial sum, the estimate's y_pre runs the whole chain --
// bit-identical values, but not the same rows, so the two are kept apart then
static void update_pre(esq_ctx *c) {
    c->pre.b_is_next = false;
    const int rows = c->pre.rows;
    if (!rows || !c->have_tab || rows >= c->s || (int)c->stage_init.size() <= rows) return;
    bool same = true;
    for (int j = 0; j < rows; ++j) same = same && c->pre.b[j] == c->A[(size_t)rows * c->s + j];
    c->pre.b_is_next = same && c->stage_init[rows] < 0;
}

extern "C" {

// (re)build the blocked-accumulation plan and the per-stage term lists for the
// current tableau: called by esq_rk_set_tableau and again when a chain entry is
// registered (the cost model depends on how the plugin sweeps)
int esq_replan(esq_ctx *c) {
    if (!c || !c->have_tab) return ESQ_EINVAL;
    const int s = c->s;
    drop_plans(c);
    const bool chained = c->rhs_chain && c->chain_depth >= 2;
    auto cost = [&](const std::vector<int> &bounds) -> double {
        if (chained) return plan_units_chained(c, bounds);
        return (double)plan_words(c->A, s, bounds, nullptr);
    };
    // ---- blocked accumulation plan (up to 3 column boundaries, exhaustive)
    c->blocks.clear();
    c->stage_init.assign(s, -1);
    c->stage_from.assign(s, 0);
    if (c->block_acc && s >= 4) {
        std::vector<int> best;
        double best_words = cost(best);
        // fewest boundaries first: a plan with more boundaries must be strictly
        // better (every boundary is one more launch)
        const bool dbg = c->plan_debug;
        if (dbg) fprintf(stderr, "[esq plan] no boundary: %.2f\n", best_words);
        for (int b1 = 2; b1 < s; ++b1) {
            const double w = cost({b1});
            if (dbg) fprintf(stderr, "[esq plan] J = %d: %.2f\n", b1, w);
            if (w >= 0 && w < best_words) { best_words = w; best = {b1}; }
        }
        for (int b1 = 2; b1 < s; ++b1)
            for (int b2 = b1 + 1; b2 < s; ++b2) {
                const double w = cost({b1, b2});
                if (w >= 0 && w < best_words) { best_words = w; best = {b1, b2}; }
            }
        if (!chained)
            for (int b1 = 2; b1 < s; ++b1)
                for (int b2 = b1 + 1; b2 < s; ++b2)
                    for (int b3 = b2 + 1; b3 < s; ++b3) {
                        const double w = cost({b1, b2, b3});
                        if (w >= 0 && w < best_words) { best_words = w; best = {b1, b2, b3}; }
                    }
        if (!best.empty()) {
            std::vector<esq_ctx::Block> blocks;
            plan_words(c->A, s, best, &blocks);
            int count = 0;
            for (auto &bl : blocks) count += (int)bl.stages.size();
            // partial-sum rows: those of an earlier plan on this context are
            // reused (a second esq_rk_set_tableau must not leak a slab)
            int first = c->block_rows_first;
            if (count > c->block_rows_count) {
                int r = esq_aux_rows(c, count, &first);
                if (r) return r;
                c->block_rows_first = first;
                c->block_rows_count = count;
            }
            std::vector<int> cur(s, -1);
            for (auto &bl : blocks) {
                for (int i : bl.stages) {
                    bl.in_vec.push_back(cur[i]);
                    bl.out_vec.push_back(first);
                    cur[i] = first++;
                }
            }
            c->blocks = blocks;
            for (int i = 1; i < s; ++i) {
                int last = 0;
                for (int J : best)
                    if (J <= i) last = J;
                c->stage_from[i] = cur[i] >= 0 ? last : 0;
                c->stage_init[i] = cur[i] >= 0 && last > 0 ? cur[i] : -1;
                if (c->stage_init[i] < 0) c->stage_from[i] = 0;
            }
        }
    }
    // what each stage kernel still has to add: the non-zero columns at or beyond
    // its blocked-accumulation boundary (built once; the step allocates nothing)
    c->stage_terms.assign(s, {});
    c->stage_nnz.assign(s, 0);
    for (int i = 1; i < s; ++i)
        for (int j = 0; j < i; ++j) {
            const double a = c->A[(size_t)i * s + j];
            if (a == 0.0) continue;
            ++c->stage_nnz[i];
            if (j >= c->stage_from[i]) c->stage_terms[i].push_back(Term{j, a});
        }
    update_pre(c);
    return 0;
}

int esq_rk_set_tableau(esq_ctx *c, int s, const double *A, const double *B,
                       const double *C, const double *E, int fsal) {
    if (!c || !A || !B || !C || !E || s < 1) return ESQ_EINVAL;
    ENTER(c);
    if (s + 1 > c->n_rows)
        return fail(c, ESQ_EINVAL, "tableau needs %d rows, context has %d", s + 1,
                    c->n_rows);
    for (int i = 0; i < s; ++i) {
        int nz = 0;
        for (int j = 0; j < s; ++j) {
            if (j >= i && A[i * s + j] != 0.0)
                return fail(c, ESQ_EINVAL, "A must be strictly lower triangular");
            nz += A[i * s + j] != 0.0;
        }
        if (nz > kMaxTerms)
            return fail(c, ESQ_EINVAL, "row %d of A has %d > %d nonzeros", i, nz, kMaxTerms);
    }
    {
        // the solution / error kernels read the union of the supports of B and E
        int nz = 0;
        for (int j = 0; j < s; ++j) nz += (B[j] != 0.0 || E[j] != 0.0);
        if (nz + (E[s] != 0.0) > kMaxTerms)
            return fail(c, ESQ_EINVAL, "B and E together have %d > %d nonzero weights",
                        nz + (E[s] != 0.0), kMaxTerms);
    }
    c->s = s;
    c->fsal = fsal ? 1 : 0;
    c->A.assign(A, A + (size_t)s * s);
    c->B.assign(B, B + s);
    c->C.assign(C, C + s);
    c->E.assign(E, E + s + 1);
    c->have_tab = true;
    if (c->pre.rows > s - 1) c->pre.rows = 0;      // (an estimate of another tableau)
    return esq_replan(c);
}

int esq_rk_stage_accumulate(esq_ctx *c, int i, double h) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    if (!c->have_tab) return fail(c, ESQ_ESTATE, "no tableau set");
    if (i < 1 || i >= c->s) return fail(c, ESQ_EINVAL, "stage %d out of range", i);
    if (i > 1) ENSURE_ROWS(c);
    for (const auto &b : c->blocks)
        if (b.J == i) {
            bool made = false;
            const int r = run_block(c, b, h, &made);
            if (r) return r;
            if (made) return 0;        // YSTAGE already holds this stage's argument
        }
    // columns [stage_from, i): the chain resumes from the stored partial sum
    const double *init = nullptr;
    Terms tm;
    int nnz_all = 0;
    const int nt = stage_terms(c, i, -1, tm, &init, &nnz_all, nullptr);
    if (nt < 0) return fail(c, ESQ_EINVAL, "too many terms");
    Prof p(c, ESQ_PROF_STAGE, "k_lincomb", nt, 8.0 * (nnz_all + 2) * (double)c->len,
           false, 8.0 * (nt + 2 + (init ? 1 : 0)) * (double)c->len);
    return launch_lincomb(c, c->ystage, c->y, tm, nt, h, &p, init);
}

int esq_rk_block_plan(esq_ctx *c, int *boundaries, int max_boundaries,
                      int *words_plain, int *words_blocked) {
    if (!c || !c->have_tab) return ESQ_EINVAL;
    std::vector<int> b;
    for (const auto &bl : c->blocks) b.push_back(bl.J);
    if (words_plain) *words_plain = plan_words(c->A, c->s, {}, nullptr);
    if (words_blocked) *words_blocked = plan_words(c->A, c->s, b, nullptr);
    for (int k = 0; k < (int)b.size() && k < max_boundaries; ++k)
        if (boundaries) boundaries[k] = b[k];
    return (int)b.size();
}

int esq_rk_eval_rhs(esq_ctx *c, int dst_row, double t, int src_slot, int src_row) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    // logical row 0 from a vector that is not a K row: this call IS the evaluation
    // the deferred end-point derivative was waiting for -- but the flag goes only
    // once the row is in memory (restore_rows' rule: after a failure the next
    // reader tries again instead of reading what is not there)
    const bool is_k0 = dst_row == 0 && src_slot != ESQ_SLOT_K;
    if (!is_k0) ENSURE_ROWS(c);
    double *dst = slot_ptr(c, ESQ_SLOT_K, dst_row);
    double *src = slot_ptr(c, src_slot, src_row);
    if (!dst || !src) return fail(c, ESQ_EINVAL, "bad row/slot");
    const int r = call_rhs(c, t, src, dst);
    if (r == 0 && is_k0) c->k0_missing = false;
    return r;
}

int esq_rk_stages(esq_ctx *c, int i_from, int i_to, double t, double h) {
    if (!c) return ESQ_EINVAL;
    // YSTAGE may already hold the first stage's argument (esq_rk_accept)
    const bool ready = i_from == 1 && c->pre_valid && c->pre_h == h;
    // ... or the whole first launch of this step may have run already (launch_ahead)
    struct { bool committed, tail_missing, k0_next; double t, h; unsigned key;
             unsigned long long missing_rows; } ahead{c->ahead.committed, c->ahead.tail_missing,
                                                     c->ahead.k0_next, c->ahead.t, c->ahead.h,
                                                     c->ahead.key, c->ahead.missing_rows};
    bool skip_first = ahead.committed && i_from == 1 && i_to == c->s && ahead.t == t &&
                      ahead.h == h;
    if (c->ahead.valid) ++c->ahead_dropped;     // launched for an attempt that was rejected
    ENTER(c);
    if (!c->have_tab) return fail(c, ESQ_ESTATE, "no tableau set");
    if (i_from < 1 || i_to > c->s || i_from > i_to)
        return fail(c, ESQ_EINVAL, "bad stage range [%d, %d)", i_from, i_to);
    if (i_from > 1) ENSURE_ROWS(c);
    c->tail_missing = false;       // whatever was missing is overwritten from here on
    c->missing_rows = 0;
    c->ynew_ready = false;
    c->solerr_ready = false;
    c->pre_last_seq = 0;
    // (the plan that launch belongs to: K[0] was still to come then)
    const bool k0_then = skip_first ? ahead.k0_next : c->k0_missing;
    if (skip_first && plan_key(i_from, i_to, false, k0_then, c->lazy_rows && !c->keep_rows) !=
                          ahead.key)
        skip_first = false;
    const Plan &plan = get_plan(c, i_from, i_to, skip_first ? false : ready,
                                skip_first ? k0_then : c->k0_missing);
    if (skip_first) {
        c->tail_missing = ahead.tail_missing;
        c->missing_rows = ahead.missing_rows;
        c->tail_accepted = false;
        c->tail_t = t;
        c->tail_h = h;
        // the launch ended in y_new: that buffer is this step's YNEW now (the old one
        // -- the state before the last step, dead from here on -- is the spare)
        if (c->ahead.wrote_ynew) std::swap(c->ynew, c->spare_vec);
        c->red_count = c->ahead.red_count;
        c->pre_last_seq = c->ahead.pre_seq;
        ++c->ahead_used;
    } else if (ahead.committed) {
        ++c->ahead_dropped;
    }
    for (size_t q = skip_first ? 1 : 0; q < plan.steps.size(); ++q) {
        const PlanStep &st = plan.steps[q];
        const int r = run_step(c, st, t, h);
        if (r == 0) continue;
        if (r != ESQ_ENOTSUP && r != kNotApplicable) return r;
        // an entry without the query capability refused a launch the plan took for
        // granted: remember it, finish THIS step the plain way (an accumulate and
        // an RHS launch per stage; rows left unwritten so far are restored by the
        // accumulate), and plan again without that launch
        const PlanStep failed = st;
        c->plans.clear();
        c->refused.insert(step_signature(failed));
        int i0 = failed.i;
        if (failed.op == OP_CHAIN && failed.i == 0) {
            const int r0 = call_rhs(c, t, c->y, c->krow[c->kmap[0]]);
            if (r0) return r0;
            ++c->end_plain;
            c->k0_missing = false;
            i0 = 1;
        }
        const int P = pre_at(c, i_from, i_to);
        for (int i = i0; i < i_to; ++i) {
            if (P && i == P && i0 < P) {
                // (the estimate has not been published by this attempt yet)
                const int rq = pre_kernel(c, h);
                if (rq) return rq;
            }
            int rp = esq_rk_stage_accumulate(c, i, h);
            if (rp) return rp;
            rp = call_rhs(c, t + c->C[i] * h, c->ystage, c->krow[c->kmap[i]]);
            if (rp) return rp;
        }
        return 0;
    }
    c->ynew_ready = plan.ynew_ready;
    c->solerr_ready = plan.solerr_ready;
    return 0;
}

// ---- the step programs as text, without a GPU (tests/test_step_plans.py) ---------
// A DETACHED context (no device, no slab: the vectors are distinct fake addresses
// nothing dereferences) gets the tableau and one of the built-in plugins' entries;
// the plans are built exactly as esq_rk_stages builds them -- the plugins answer
// the queries on the host -- and printed one per line:
//   <key>: <op>[<i>[,<depth>,<what>]][flags] ...  | launches=<n> words=<read>+<written>
// keys: first = a step that starts from K[0] in memory (the first step, a retry
// after a rejection), deferred = after an accepted step that left f(t, y) to this
// one, prelaunched = stage 1's argument formed at accept time.
// a detached context with one of the built-in plugins' entries and the tableau set
static int make_detached(esq_ctx *c, void **user_out, const char *plugin, int N, int s,
                         const double *A, const double *B, const double *C, const double *E,
                         int fsal, int chain_caps, int fuse_mask, int lazy_rows,
                         int chain_depth, int src_pays, int extra_rows = 0) {
    c->detached = true;
    c->device = -1;
    void *user = nullptr;
    size_t n = 0;
    const std::string name(plugin);
    if (name == "bruss2d") {
        if (esq_rhs_bruss2d_create(&user, N, 0.1, 1.0, 3.4)) return ESQ_EINVAL;
        n = 2 * (size_t)N * N;
        c->rhs = esq_rhs_bruss2d; c->rhs_fused = esq_rhs_bruss2d_fused;
        c->rhs_chain = esq_rhs_bruss2d_chain;
    } else if (name == "heat2d") {
        if (esq_rhs_heat2d_create(&user, N)) return ESQ_EINVAL;
        n = (size_t)N * N;
        c->rhs = esq_rhs_heat2d; c->rhs_fused = esq_rhs_heat2d_fused;
        c->rhs_chain = esq_rhs_heat2d_chain;
    } else if (name == "diff3d") {
        if (esq_rhs_diff3d_create(&user, N)) return ESQ_EINVAL;
        n = (size_t)N * N * N;
        c->rhs = esq_rhs_diff3d; c->rhs_fused = esq_rhs_diff3d_fused;
        c->rhs_chain = esq_rhs_diff3d_chain;
    } else if (name == "plain") {                 // an esq_rhs_fn-only plugin
        if (esq_rhs_heat2d_create(&user, N)) return ESQ_EINVAL;
        n = (size_t)N * N;
        c->rhs = esq_rhs_heat2d;
    } else {
        return ESQ_EINVAL;
    }
    *user_out = user;
    c->rhs_user = user;
    c->n = c->len = n;
    c->len_pad = ((n + kPadDoubles - 1) / kPadDoubles) * kPadDoubles;
    c->stride = c->len_pad;
    c->n_rows = s + 1 + extra_rows;
    double *fake = reinterpret_cast<double *>((uintptr_t)1 << 40);
    c->krow.resize(c->n_rows);
    c->kmap.resize(c->n_rows);
    for (int r = 0; r < c->n_rows; ++r) { c->krow[r] = fake + (size_t)r * c->stride; c->kmap[r] = r; }
    c->kmap_last = c->kmap;
    double *base = fake + (size_t)c->n_rows * c->stride;
    c->y = base; c->ynew = base + c->stride; c->ystage = base + 2 * c->stride;
    c->atolv = base + 3 * c->stride; c->work = base + 4 * c->stride;
    c->partials = base + 5 * c->stride;
    c->fuse_mask = c->rhs_fused ? fuse_mask : 0;
    c->chain_caps = c->rhs_chain ? chain_caps : 0;
    c->lazy_rows = lazy_rows != 0;
    c->lazy_end = true;
    c->chain_depth = chain_depth;
    c->src_pays = src_pays != 0;
    return esq_rk_set_tableau(c, s, A, B, C, E, fsal);
}

int esq_plan_describe(const char *plugin, int N, int s, const double *A, const double *B,
                          const double *C, const double *E, int fsal, int chain_caps,
                          int fuse_mask, int lazy_rows, int chain_depth, int src_pays,
                          const double *e_pre, const double *b_pre, int pre_rows, char *buf,
                          size_t buflen) {
    if (!plugin || !A || !B || !C || !E || !buf || buflen < 2 || s < 1 || N < 1) return ESQ_EINVAL;
    esq_ctx ctx;
    esq_ctx *c = &ctx;
    void *user = nullptr;
    int r = make_detached(c, &user, plugin, N, s, A, B, C, E, fsal, chain_caps, fuse_mask,
                          lazy_rows, chain_depth, src_pays);
    if (r && !user) return r;
    if (!r && pre_rows) r = esq_rk_set_pre(c, e_pre, b_pre, pre_rows);
    size_t used = 0;
    buf[0] = 0;
    static const char *kOp[] = {"k0", "chain", "src", "accum", "lincomb", "stage", "block",
                                "ynew", "solerr", "rhs", "pre"};
    const struct { const char *label; bool ready, k0; } keys[] = {
        {"first", false, false}, {"deferred", false, true}, {"prelaunched", true, false}};
    for (const auto &k : keys) {
        if (r) break;
        if (k.k0 && fsal) continue;
        const Plan plan = build_plan(c, 1, s, k.ready, k.k0, c->lazy_rows);
        std::string line = std::string(k.label) + ":";
        double rd = 0, wr = 0, cost = 0;
        for (const PlanStep &st : plan.steps) {
            char tok[64];
            if (st.op == OP_CHAIN)
                snprintf(tok, sizeof(tok), " chain[%d,%d,%d]%s%s%s", st.i, st.depth, st.what,
                         st.lazy ? "L" : "", st.from_rows ? "F" : "", st.skip_out ? "S" : "");
            else
                snprintf(tok, sizeof(tok), " %s[%d]", kOp[st.op], st.i);
            line += tok;
            rd += st.reads; wr += st.writes;
            cost += step_cost(c, st);
        }
        char tail[128];
        snprintf(tail, sizeof(tail), " | launches=%zu words=%g+%g cost=%.2f%s%s\n",
                 plan.steps.size(), rd, wr, cost, plan.ynew_ready ? " ynew" : "",
                 plan.solerr_ready ? " solerr" : "");
        line += tail;
        if (used + line.size() + 1 > buflen) { r = ESQ_EINVAL; break; }
        memcpy(buf + used, line.c_str(), line.size() + 1);
        used += line.size();
    }
    esq_rhs_free(user);
    return r;
}

// ---- whole steps on a detached context: the HOST side of the step -- plans, row
// maps, the launch ahead of time and what it saves and restores, the lazy rows, the
// deferred end-point derivative -- run without a GPU (every launch is the plugin's
// answer to the query instead; reductions return 0).  `script` holds one code per
// attempt, as RungeKutta._step_impl would make the calls:
//   0  accepted, next step size named BEFORE the norm is known (a run at max_step:
//      esq_rk_solution_error_ahead) and confirmed by the accept
//   1  accepted, next step size named at accept time only
//   2  rejected (the attempt is repeated with half the step)
//   3  accepted with a next step size that the following attempt does NOT take
//   4  accepted; a reader asks for rows of K before the next step (esq_rk_row_id)
//   5  accepted with the guess of code 0, but the accept names another step size
// One line per attempt goes to buf:
//   <code>: state_ok=<0|1> used=<n> dropped=<n> missing=<rows> k0=<0|1> fused=<n> plain=<n>
// state_ok: every launch_ahead of the attempt left StepState and the row map as it
// found them.  tests/test_step_plans.py (also under the sanitizer build).
namespace {
bool same_state(const StepState &a, const StepState &b) {
    static_assert(sizeof(StepState) == 112, "a new field of StepState: compare it below");
    return a.y == b.y && a.ynew == b.ynew && a.ystage == b.ystage && a.work == b.work &&
           a.ynew_ready == b.ynew_ready && a.solerr_ready == b.solerr_ready &&
           a.red_count == b.red_count && a.tail_missing == b.tail_missing &&
           a.tail_accepted == b.tail_accepted && a.missing_rows == b.missing_rows &&
           a.tail_t == b.tail_t && a.tail_h == b.tail_h && a.k0_missing == b.k0_missing &&
           a.k0_t == b.k0_t && a.end_fused == b.end_fused && a.end_plain == b.end_plain &&
           a.pre_last_seq == b.pre_last_seq;
}
}  // namespace
int esq_step_dry_run(const char *plugin, int N, int s, const double *A, const double *B,
                         const double *C, const double *E, int fsal, int chain_caps,
                         int fuse_mask, int lazy_rows, int chain_depth, int src_pays,
                         const double *e_pre, const double *b_pre, int pre_rows,
                         const int *script, int n_attempts, char *buf, size_t buflen) {
    if (!plugin || !A || !B || !C || !E || !buf || buflen < 2 || s < 1 || N < 1 || !script ||
        n_attempts < 1)
        return ESQ_EINVAL;
    esq_ctx ctx;
    esq_ctx *c = &ctx;
    void *user = nullptr;
    // (spare rows of the launch ahead of time come from esq_aux_rows: fake addresses)
    int r = make_detached(c, &user, plugin, N, s, A, B, C, E, fsal, chain_caps, fuse_mask,
                          lazy_rows, chain_depth, src_pays);
    if (r && !user) return r;
    if (!r && pre_rows) r = esq_rk_set_pre(c, e_pre, b_pre, pre_rows);
    c->ahead_on = true;
    size_t used = 0;
    buf[0] = 0;
    double t = 0.0, h = 1.0 / 64;
    double sumsq = 0.0;
    for (int q = 0; q < n_attempts && r == 0; ++q) {
        const int code = script[q];
        bool ok = true;
        r = esq_rk_stages(c, 1, s, t, h);
        if (r) break;
        {
            // the speculation by itself: whatever the launch does to the context on the
            // next step's behalf, the context is afterwards what it was
            const StepState before = *c;
            // (the first launch adds the spare rows to the context: compare the
            // rows of the method, logical 0 .. s)
            auto head = [&](const std::vector<int> &m) {
                return std::vector<int>(m.begin(), m.begin() + s + 1);
            };
            const std::vector<int> kmap = head(c->kmap), kmap_last = head(c->kmap_last);
            launch_ahead(c, t + h, h);
            ok = same_state(before, *c) && kmap == head(c->kmap) &&
                 kmap_last == head(c->kmap_last);
            c->ahead.valid = c->ahead.committed = false;
        }
        // ... and as the calls of a step make it
        const double h_guess = (code == 0 || code == 5) ? h : 0.0;
        r = esq_rk_solution_error_ahead(c, t, h, h_guess, &sumsq);
        if (r) break;
        unsigned long long pre_seq = 0;
        if (c->pre.rows) {
            // every whole-step attempt carries exactly one early estimate of its own
            double pre_sum = 0.0;
            r = esq_rk_pre_result(c, &pre_sum);
            if (r) break;
            pre_seq = c->pre_last_seq;
        }
        if (code == 2) {                                   // rejected
            h *= 0.5;
        } else {
            const double h_next = code == 5 ? 0.75 * h : h;
            r = esq_rk_accept(c, t + h, fsal ? 0 : 1, h_next);
            if (r) break;
            t += h;
            h = code == 3 ? 0.875 * h_next : h_next;
            if (code == 4) {
                const int id = esq_rk_row_id(c, s > 2 ? s - 1 : 0, 1);
                if (id < 0) { r = id; break; }
            }
        }
        // invariants of the row maps: permutations of distinct physical rows, the
        // spares disjoint from both
        {
            std::set<int> seen(c->kmap.begin(), c->kmap.begin() + s + 1);
            ok = ok && (int)seen.size() == s + 1;
            for (int sp : c->spare_rows) ok = ok && !seen.count(sp);
            std::set<int> sp(c->spare_rows.begin(), c->spare_rows.end());
            ok = ok && sp.size() == c->spare_rows.size();
        }
        char line[240];
        int w = snprintf(line, sizeof(line),
                 "%d: state_ok=%d used=%ld dropped=%ld missing=%d k0=%d fused=%ld plain=%ld",
                 code, ok ? 1 : 0, c->ahead_used, c->ahead_dropped,
                 c->tail_missing ? __builtin_popcountll(c->missing_rows) : 0,
                 c->k0_missing ? 1 : 0, c->end_fused, c->end_plain);
        if (c->pre.rows)
            w += snprintf(line + w, sizeof(line) - (size_t)w, " pre=%llu/%llu fused=%ld plain=%ld",
                          pre_seq, c->pre.seq, c->pre.fused, c->pre.plain);
        snprintf(line + w, sizeof(line) - (size_t)w, "\n");
        const size_t len = strlen(line);
        if (used + len + 1 > buflen) { r = ESQ_EINVAL; break; }
        memcpy(buf + used, line, len + 1);
        used += len;
    }
    esq_rhs_free(user);
    return r;
}

int esq_rk_solution(esq_ctx *c, double h) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    if (!c->have_tab) return fail(c, ESQ_ESTATE, "no tableau set");
    ENSURE_ROWS(c);
    Terms tm;
    const int nt = build_row_terms(c, c->B.data(), c->s, tm, c->kmap);
    if (nt < 0) return fail(c, ESQ_EINVAL, "too many terms");
    Prof p(c, ESQ_PROF_SOLERR, "k_lincomb", nt, 8.0 * (nt + 2) * (double)c->len);
    return launch_lincomb(c, c->ynew, c->y, tm, nt, h, &p);
}

int esq_rk_error_norm(esq_ctx *c, double h, double *sumsq_out) {
    if (!c || !sumsq_out) return ESQ_EINVAL;
    ENTER(c);
    if (!c->have_tab) return fail(c, ESQ_ESTATE, "no tableau set");
    ENSURE_ROWS(c);
    Terms tm;
    const int nt = build_row_terms(c, c->E.data(), c->s + c->fsal, tm, c->kmap);
    if (nt < 1) return fail(c, ESQ_EINVAL, "error weights are all zero");
    {
        Prof p(c, ESQ_PROF_SOLERR, "k_error_norm", nt,
               8.0 * (nt + 2) * (double)c->len);
        const int r = launch_errnorm(c, tm, nt, h, p);
        if (r) return r;
    }
    return finish_reduction(c, sumsq_out);
}

int esq_rk_solution_error_ahead(esq_ctx *c, double t, double h, double h_next,
                                 double *sumsq_out) {
    if (!c || !sumsq_out) return ESQ_EINVAL;
    // the request is taken up by finish_reduction, between the enqueue of the final
    // sum and the wait for it
    c->ahead_ask_t = t + h;
    c->ahead_ask_h = h_next;
    const int r = esq_rk_solution_error(c, t, h, sumsq_out);
    c->ahead_ask_h = 0.0;
    return r;
}
// one call for a whole attempt of a device-RHS step: esq_rk_stages(1, s, t, h), then
// esq_rk_solution_error_ahead(t, h, h_next) -- on small grids a step is bound by the
// host's share (config 2: 23 us of kernel, 33 us of step), and every call through the
// binding is a microsecond of it.  pre_sumsq_out (may be NULL): the early estimate of
// the attempt where one is registered (esq_rk_set_pre), else untouched.
int esq_rk_attempt(esq_ctx *c, double t, double h, double h_next, double *sumsq_out,
                   double *pre_sumsq_out) {
    if (!c || !sumsq_out) return ESQ_EINVAL;
    int r = esq_rk_stages(c, 1, c->s, t, h);
    if (r) return r;
    r = esq_rk_solution_error_ahead(c, t, h, h_next, sumsq_out);
    if (r) return r;
    if (pre_sumsq_out && c->pre.rows && c->pre_last_seq) r = esq_rk_pre_result(c, pre_sumsq_out);
    return r;
}
int esq_rk_solution_error(esq_ctx *c, double t, double h, double *sumsq_out) {
    if (!c || !sumsq_out) return ESQ_EINVAL;
    const bool ynew_ready = c->ynew_ready, solerr_ready = c->solerr_ready;
    ENTER(c);
    c->ynew_ready = c->solerr_ready = false;
    if (!c->have_tab) return fail(c, ESQ_ESTATE, "no tableau set");
    if (c->fsal && solerr_ready)       // the chain ran through the end of the step
        return finish_reduction(c, sumsq_out, false, c->partials, c->red_count);
    if (c->fsal) {
        int r = 0;
        if (!ynew_ready) r = esq_rk_solution(c, h);
        if (r) return r;
        if (may_fuse(c, ESQ_EPI_ERRNORM)) {
            // K[s] = f(t + h, y_new) and the error norm in ONE sweep
            esq_epilogue e;
            epi_common(c, e, ESQ_EPI_ERRNORM);
            int nt = 0;
            bool ok = true;
            for (int j = 0; j < c->s; ++j) {
                if (c->E[j] == 0.0) continue;
                if (nt >= ESQ_EPI_MAX_ROWS) { ok = false; break; }
                e.rows[nt] = c->krow[c->kmap[j]];
                e.e[nt] = c->E[j];
                ++nt;
            }
            if (ok) {
                e.nt = nt;
                e.e_self = c->E[c->s];
                e.y = c->y;
                e.h = h;
                e.f_store_nt = (c->epi_nt >> 4) & 1;   // K[s] is the next step's K[0]
                // booked: RHS 16 B + error pass (rows incl. K[s], y, y_new);
                // moved: y_new, rows, y in; K[s] out
                Prof p(c, ESQ_PROF_SOLERR, "rhs+errnorm", nt,
                       8.0 * (nt + 1 + 2 + 2) * (double)c->len, false,
                       8.0 * (nt + 3) * (double)c->len);
                r = run_fused(c, t + h, c->ynew, c->krow[c->kmap[c->s]], e, p);
                if (r == 0) return finish_reduction(c, sumsq_out, false, c->partials,
                                                    c->red_count);
                if (r != ESQ_ENOTSUP) return r;
            }
        }
        r = call_rhs(c, t + h, c->ynew, c->krow[c->kmap[c->s]]);
        if (r) return r;
        return esq_rk_error_norm(c, h, sumsq_out);
    }
    if (solerr_ready)
        return finish_reduction(c, sumsq_out, false, c->partials, c->red_count);
    ENSURE_ROWS(c);
    Terms2 tm;
    const int nt = build_row_terms2(c, c->B.data(), c->s, c->E.data(), c->s, tm, c->kmap);
    if (nt < 1) return fail(c, ESQ_EINVAL, "bad weights");
    {
        Prof p(c, ESQ_PROF_SOLERR, "k_solution_error", nt,
               8.0 * (nt + 2) * (double)c->len);
        const int r = launch_solerr(c, tm, nt, h, p);
        if (r) return r;
    }
    return finish_reduction(c, sumsq_out);
}

int esq_rk_pre_error(esq_ctx *c, double h, const double *e_pre,
                     const double *b_scale_pre, int rows, double *sumsq_out) {
    if (!c || !e_pre || !b_scale_pre || !sumsq_out) return ESQ_EINVAL;
    ENTER(c);
    if (rows < 1 || rows > c->n_rows) return fail(c, ESQ_EINVAL, "bad rows %d", rows);
    ENSURE_ROWS(c);
    Terms2 tm;
    const int nt = build_row_terms2(c, b_scale_pre, rows, e_pre, rows, tm, c->kmap);
    if (nt < 1) return fail(c, ESQ_EINVAL, "bad weights");
    {
        Prof p(c, ESQ_PROF_SOLERR, "k_pre_error", nt,
               8.0 * (nt + 1) * (double)c->len);
        const int r = launch_preerr(c, tm, nt, h, p);
        if (r) return r;
    }
    return finish_reduction(c, sumsq_out);
}

int esq_rk_set_pre(esq_ctx *c, const double *e_pre, const double *b_scale_pre, int rows) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    drop_plans(c);
    c->pre.rows = 0;
    c->pre.e.clear();
    c->pre.b.clear();
    c->pre.b_is_next = false;
    if (rows == 0) return 0;
    if (!c->have_tab) return fail(c, ESQ_ESTATE, "no tableau set");
    if (!e_pre || !b_scale_pre || rows < 2 || rows > c->s - 1)
        return fail(c, ESQ_EINVAL, "early estimate over %d rows of a %d-stage pair", rows, c->s);
    c->pre.e.assign(e_pre, e_pre + rows);
    c->pre.b.assign(b_scale_pre, b_scale_pre + rows);
    c->pre.rows = rows;
    update_pre(c);
    return 0;
}
int esq_rk_pre_result(esq_ctx *c, double *sumsq_out) {
    if (!c || !sumsq_out) return ESQ_EINVAL;
    ENTER_KEEP(c);
    const unsigned long long seq = c->pre_last_seq;
    if (!seq) return fail(c, ESQ_ESTATE, "no early estimate in flight (esq_rk_set_pre, then "
                          "esq_rk_stages over the whole step)");
    if (c->detached) { *sumsq_out = 0.0; return 0; }
    // the attempt's final reduction has been waited for: the estimate, published by an
    // earlier kernel of the same stream, is there -- the spin is for its visibility only
    const auto &slot = c->h_slot->pre[seq % kPreSlots];
    for (unsigned long spins = 1;; ++spins) {
        if (__atomic_load_n(&slot.seq, __ATOMIC_ACQUIRE) == seq) break;
        if ((spins & 0xfff) == 0) {
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) {
                if (__atomic_load_n(&slot.seq, __ATOMIC_ACQUIRE) == seq) break;
                return fail(c, ESQ_ESTATE, "early estimate %llu finished without a result", seq);
            }
            if (q != hipErrorNotReady)
                return fail(c, (int)q, "stream failed while waiting for the early estimate: %s",
                            hipGetErrorString(q));
        }
    }
    *sumsq_out = slot.value;
    return 0;
}

int esq_rk_custom_sol_err(esq_ctx *c, double h, const double *b, const double *e,
                          int rows, int store_ynew, double *sumsq_out) {
    if (!store_ynew) return esq_rk_pre_error(c, h, e, b, rows, sumsq_out);
    if (!c || !b || !e || !sumsq_out) return ESQ_EINVAL;
    ENTER(c);
    if (rows < 1 || rows > c->n_rows) return fail(c, ESQ_EINVAL, "bad rows %d", rows);
    ENSURE_ROWS(c);
    Terms2 tm;
    const int nt = build_row_terms2(c, b, rows, e, rows, tm, c->kmap);
    if (nt < 1) return fail(c, ESQ_EINVAL, "bad weights");
    {
        Prof p(c, ESQ_PROF_SOLERR, "k_solution_error", nt,
               8.0 * (nt + 2) * (double)c->len);
        const int r = launch_solerr(c, tm, nt, h, p);
        if (r) return r;
    }
    return finish_reduction(c, sumsq_out);
}

int esq_rk_accept(esq_ctx *c, double t_new, int with_end_eval, double h_next) {
    if (!c) return ESQ_EINVAL;
    // the next step's first launch may be in the queue already (launch_ahead, asked
    // for by esq_rk_solution_error_ahead): it is this accept that makes it real
    bool commit = c->ahead.valid && h_next != 0.0 && c->ahead.h == h_next &&
                  c->ahead.t == t_new && c->have_tab && (with_end_eval || c->fsal);
    if (c->ahead.valid && !commit) ++c->ahead_dropped;
    ENTER(c);                      // (clears the flags of c->ahead, not what it holds)
    if (!c->have_tab) return fail(c, ESQ_ESTATE, "no tableau set");
    // ... or is made now that the step size is known: it runs while the host
    // returns from this step and enters the next
    if (!commit && h_next != 0.0 && (with_end_eval || c->fsal)) {
        launch_ahead(c, t_new, h_next);
        commit = c->ahead.valid;
        c->ahead.valid = false;
    }
    // the next step's first stage argument can be formed now: stage 1 reads
    // nothing but y and K[0]
    const bool want_pre = !commit && h_next != 0.0 && c->s >= 2 && c->rhs != nullptr &&
                          !(may_use_src(c) && c->s >= 3);
    bool pre_done = false;
    // the end-point derivative can wait for the next step's first chain sweep (if
    // the next whole step's program starts with that chain: a query, not a guess)
    const bool defer = !commit && with_end_eval && defers_end_point(c);
    if (commit) {
        // (the launch ahead evaluated f(t_new, y_new) as its stage 0, or the pair is FSAL)
    } else if (defer) {
        c->k0_missing = true;
        c->k0_t = t_new;
    } else if (!c->fsal && with_end_eval) {
        int r = ESQ_ENOTSUP;
        if (want_pre && may_fuse(c, ESQ_EPI_STAGE)) {
            // K[s] = f(t_new, y_new) and YSTAGE = y_new + h_next*a_10*K[s] in
            // ONE sweep (K[s] becomes K[0], y_new becomes y below)
            esq_epilogue e;
            epi_common(c, e, ESQ_EPI_STAGE);
            e.nt = 0;
            e.c_self = c->A[(size_t)c->s];          // A[1][0]
            e.y = nullptr;                          // base = the sweep's input
            e.h = h_next;
            e.out = c->ystage;
            e.f_store_nt = (c->epi_nt >> 3) & 1;    // K[0] of the next step
            const int nnz = e.c_self != 0.0 ? 1 : 0;
            Prof p(c, ESQ_PROF_STAGE, "rhs+stage", 0, 8.0 * (nnz + 4) * (double)c->len,
                   false, 8.0 * 3 * (double)c->len);
            r = run_fused(c, t_new, c->ynew, c->krow[c->kmap[c->s]], e, p);
            if (r == 0) pre_done = true;
            else if (r != ESQ_ENOTSUP) return r;
        }
        if (r == ESQ_ENOTSUP) {
            r = call_rhs(c, t_new, c->ynew, c->krow[c->kmap[c->s]]);
            if (r) return r;
        }
    }
    c->kmap_last = c->kmap;
    c->tail_accepted = true;       // a restore now works from kmap_last and YNEW
    ++c->accepted_steps;
    std::swap(c->y, c->ynew);
    if (commit) {
        // the rows the launch wrote take the place of those logical rows; the
        // physical rows they replace (still this step's, for its readers) are the
        // spares of the next launch ahead
        c->kmap = c->ahead.kmap;
        c->spare_rows = c->ahead.spares;
        c->ystage = c->ahead.ystage;
        c->work = c->ahead.work;
        c->k0_missing = false;
        if (c->ahead.k0_done) ++c->end_fused;
        c->ahead.committed = true;
        return 0;
    }
    std::swap(c->kmap[0], c->kmap[c->s]);
    if (want_pre && !pre_done && !defer) {
        // stage 1's accumulate, launched now: it runs while the host controller
        // is between steps
        const int r = esq_rk_stage_accumulate(c, 1, h_next);
        if (r) return r;
        pre_done = true;
    }
    c->pre_valid = pre_done;
    c->pre_h = h_next;
    return 0;
}

int esq_rk_error_vector(esq_ctx *c, double h, int last_step) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    if (!c->have_tab) return fail(c, ESQ_ESTATE, "no tableau set");
    ENSURE_ROWS(c);
    Terms tm;
    const int nt = build_row_terms(c, c->E.data(), c->s + c->fsal, tm,
                                   last_step ? c->kmap_last : c->kmap);
    if (nt < 0) return fail(c, ESQ_EINVAL, "too many terms");
    return launch_lincomb(c, c->work, nullptr, tm, nt, h);
}

int esq_rk_row_id(esq_ctx *c, int logical_row, int last_step) {
    if (!c || logical_row < 0 || logical_row >= c->n_rows) return ESQ_EINVAL;
    ENTER_KEEP(c);
    if (c->tail_missing || c->k0_missing) {        // the caller is about to read that row
        const int rr = esqi::restore_rows(c);
        if (rr) return rr < 0 ? rr : ESQ_ESTATE;   // (ids are >= 0: never a positive HIP code)
    }
    return last_step ? c->kmap_last[logical_row] : c->kmap[logical_row];
}
int esq_rk_download_last_K(esq_ctx *c, int row, double *host) {
    if (!c || !host) return ESQ_EINVAL;
    ENTER_KEEP(c);
    if (row < 0 || row >= c->n_rows) return fail(c, ESQ_EINVAL, "bad row %d", row);
    ENSURE_ROWS(c);
    return d2h(c, host, c->krow[c->kmap_last[row]], c->len * sizeof(double), c->idle);
}

int esq_rk_lazy_rows(esq_ctx *c, int *missing_out, int *keeps_out, long *restores_out,
                     long *end_fused_out, long *end_plain_out) {
    if (!c) return ESQ_EINVAL;
    if (missing_out)
        *missing_out = (c->tail_missing ? __builtin_popcountll(c->missing_rows) : 0) +
                       (c->k0_missing ? 1 : 0);
    if (keeps_out) *keeps_out = (c->keep_rows || !c->lazy_rows) ? 1 : 0;
    if (restores_out) *restores_out = c->restores;
    if (end_fused_out) *end_fused_out = c->end_fused;
    if (end_plain_out) *end_plain_out = c->end_plain;
    return 0;
}

int esq_rk_set_launch_ahead(esq_ctx *c, int on) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    c->ahead_on = on != 0;
    return 0;
}
int esq_rk_launch_ahead_stats(esq_ctx *c, long *used_out, long *dropped_out) {
    if (!c) return ESQ_EINVAL;
    if (used_out) *used_out = c->ahead_used;
    if (dropped_out) *dropped_out = c->ahead_dropped;
    return 0;
}

int esq_rk_dense_stage(esq_ctx *c, int row, const double *a, int count, double h) {
    if (!c || !a) return ESQ_EINVAL;
    ENTER(c);
    if (row < 1 || row >= c->n_rows || count < 0 || count > row)
        return fail(c, ESQ_EINVAL, "bad row/count %d/%d", row, count);
    ENSURE_ROWS(c);
    Terms tm;
    const int nt = build_row_terms(c, a, count, tm, c->kmap_last);
    if (nt < 0) return fail(c, ESQ_EINVAL, "too many terms");
    // after esq_rk_accept the pre-step state is in the YNEW slot
    return launch_lincomb(c, c->ystage, c->ynew, tm, nt, h);
}
int esq_rk_dense_eval(esq_ctx *c, int row, double t) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    if (row < 0 || row >= c->n_rows) return fail(c, ESQ_EINVAL, "bad row %d", row);
    return call_rhs(c, t, c->ystage, c->krow[c->kmap_last[row]]);
}
int esq_rk_upload_last_K(esq_ctx *c, int row, const double *host) {
    if (!c || !host) return ESQ_EINVAL;
    ENSURE_ROWS(c);
    const bool was_idle = c->idle;
    ENTER(c);
    if (row < 0 || row >= c->n_rows) return fail(c, ESQ_EINVAL, "bad row %d", row);
    return h2d(c, c->krow[c->kmap_last[row]], host, c->len * sizeof(double), was_idle);
}

}  // extern "C"
