// esq_options.hpp -- the library's tuning switches, per OBJECT.
//
// Until round 5 every switch was an environment variable, and the Python classes'
// `esq_options=` keyword worked by writing os.environ for the duration of a
// constructor: switches read later had no effect, process-wide tables leaked from
// one solver to the next, and setenv raced with getenv in worker threads (ADVICE
// r05).  Now a switch is a (key, value) pair of the object it steers:
//
//   esq_create2(..., "chain_depth=3;lazy_rows=0")        the context's switches
//   esq_rhs_set_options(user, "chain_rows=12")           a built-in plugin object's
//
// and the process environment (ESQ_<KEY>) is only the DEFAULT of a key the caller did
// not give -- read through env_get() below, the one getenv of the library, on the
// constructing thread.  Unknown keys are refused (esq_option_level).
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <utility>
#include <vector>

namespace esq {

// THE place the library reads the process environment: ESQ_<KEY>
inline const char *env_get(const char *key) {
    char name[80];
    snprintf(name, sizeof(name), "ESQ_%s", key);
    return getenv(name);
}

// which object a key steers: 1 = a context (esq_create2), 2 = a plugin object
// (esq_rhs_set_options), 0 = not a switch of the library
constexpr int kOptContext = 1, kOptPlugin = 2;
struct OptionKey {
    const char *key;
    int level;
};
inline const OptionKey *option_keys() {
    static const OptionKey keys[] = {
        // the step's program
        {"CHAIN_DEPTH", kOptContext},      // stages per marching sweep (1: one sweep per stage)
        {"CHAIN_FROM_ROWS", kOptContext},  // 0: no chain forms its own input
        {"LAZY_ROWS", kOptContext},        // 0: every K row written by its step
        {"LAZY_END", kOptContext},         // 0: end-point derivative at accept time
        {"BLOCK_ACC", kOptContext},        // 0: no blocked accumulation
        {"SRC", kOptContext},              // 0 | 1: first sweep from the state
        {"PLAN_DEBUG", kOptContext},       // the planner's queries on stderr
        // cache policies and grids of the library's own kernels
        {"EPI_NT", kOptContext},
        {"STAGE_POLICY", kOptContext},
        {"BLOCKS_PER_CU", kOptContext},
        {"CHAIN_LDNT", kOptContext},
        // Chebyshev steps
        {"RKC_DEPTH", kOptContext},
        {"RKC_FIRST", kOptContext},
        {"RKC_LAST", kOptContext},
        // lock-step
        {"COMM_TIMEOUT_S", kOptContext},
        // plugin objects: tile geometry of the chain sweeps (tests force small tiles)
        {"CHAIN_ROWS", kOptPlugin},
        {"RKC_FORCE", kOptPlugin},
        {"RKC_PLANES", kOptPlugin},
        {"DIFF3D_R", kOptPlugin},
        {nullptr, 0}};
    return keys;
}
// key in any case, with or without the ESQ_ prefix -> canonical upper case, no prefix
inline std::string option_canonical(const char *key) {
    std::string k(key ? key : "");
    for (char &ch : k)
        if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 'a' + 'A');
    if (k.rfind("ESQ_", 0) == 0) k.erase(0, 4);
    return k;
}
inline int option_level(const char *key) {
    const std::string k = option_canonical(key);
    for (const OptionKey *o = option_keys(); o->key; ++o)
        if (k == o->key) return o->level;
    return 0;
}

class Options {
    std::vector<std::pair<std::string, std::string>> kv_;

  public:
    // "key=value;key=value" (';' or whitespace between pairs); keys of a level not in
    // `level_mask` and unknown keys: -1 and the offender in *bad.  An empty value is kept
    // (it says "as if unset").
    int parse(const char *text, int level_mask, std::string *bad) {
        kv_.clear();
        if (!text) return 0;
        std::string cur;
        auto flush = [&]() -> int {
            if (cur.empty()) return 0;
            const size_t eq = cur.find('=');
            const std::string key = option_canonical(cur.substr(0, eq).c_str());
            const std::string val = eq == std::string::npos ? "1" : cur.substr(eq + 1);
            cur.clear();
            if (!(option_level(key.c_str()) & level_mask)) {
                if (bad) *bad = key;
                return -1;
            }
            kv_.emplace_back(key, val);
            return 0;
        };
        for (const char *p = text; *p; ++p) {
            if (*p == ';' || *p == '\n' || *p == ' ' || *p == '\t') {
                if (flush()) return -1;
            } else {
                cur.push_back(*p);
            }
        }
        return flush();
    }
    // the explicit value, else the process default ESQ_<KEY>, else nullptr
    const char *get(const char *key) const {
        for (auto it = kv_.rbegin(); it != kv_.rend(); ++it)
            if (it->first == key) return it->second.empty() ? nullptr : it->second.c_str();
        return env_get(key);
    }
    bool has(const char *key) const { return get(key) != nullptr; }
    unsigned uint_or(const char *key, unsigned dflt) const {
        const char *s = get(key);
        if (!s || !*s) return dflt;
        char *end = nullptr;
        const long v = strtol(s, &end, 10);
        return (end != s && v >= 0) ? (unsigned)v : dflt;
    }
    int int_or(const char *key, int dflt) const {
        const char *s = get(key);
        if (!s || !*s) return dflt;
        char *end = nullptr;
        const long v = strtol(s, &end, 10);
        return end != s ? (int)v : dflt;
    }
};

}  // namespace esq
