// SLIMM_FORCE: the ONE environment variable by which tests and tuning runs make the library take a path it would not pick
// by itself -- the layout fallbacks of the tile kernels, the 32-byte lineage rows, a pair set that overflows, a grouping plan.
//   SLIMM_FORCE="key=value,key,key=value"      (a key without a value reads as 1)
// Read at every use (a test varies it between two contexts of one process).  Every key has a `-m gpu` test that names it:
//   direct_atomics, two_level=0|1, fused_scan=0, matrix=0|2, wide_tiles=0|1, tile_shift=13|14, wide_rows, pair_cap=N,
//   scatter_big=0|1                                   tests/test_gpu_parity.py, test_gpu_layouts.py, test_gpu_group.py
//   group_bits=N, group_width=N, group_passes=N, group_grid=N, group_staged=0|1      tests/test_gpu_group_by_ident.py
//   group_collectives=rccl|copy                                                       tests/test_gpu_group.py
#pragma once
#include <cstdlib>
#include <cstring>

namespace slimm {

// true when SLIMM_FORCE names `key`; *text = its value's first character position (nullptr when it has none)
inline bool forced_text(const char* key, const char** text) {
    const char* e = getenv("SLIMM_FORCE");
    if (text) *text = nullptr;
    if (!e) return false;
    const size_t n = strlen(key);
    for (const char* p = e; *p;) {
        const char* end = strchr(p, ',');
        const size_t len = end ? static_cast<size_t>(end - p) : strlen(p);
        if (len >= n && memcmp(p, key, n) == 0 && (len == n || p[n] == '=')) {
            if (text && len > n) *text = p + n + 1;
            return true;
        }
        if (!end) break;
        p = end + 1;
    }
    return false;
}
inline bool forced(const char* key, long* value = nullptr) {
    const char* t = nullptr;
    if (!forced_text(key, &t)) return false;
    if (value) *value = t ? atol(t) : 1;
    return true;
}

// SLIMM_TRACE="cli,host,push" (or "all" / "1"): diagnostics on stderr -- cli: the command's stage marks (host/slimm_main.cpp),
// host: the library's host steps between the device phases (context.h: HostTrace), push: the window pipeline's events
// (windows.hip).  Nothing is computed differently with it.
inline bool traced(const char* what) {
    const char* e = getenv("SLIMM_TRACE");
    if (!e || !*e) return false;
    if (strcmp(e, "1") == 0 || strcmp(e, "all") == 0) return true;
    const size_t n = strlen(what);
    for (const char* p = e; *p;) {
        const char* end = strchr(p, ',');
        const size_t len = end ? static_cast<size_t>(end - p) : strlen(p);
        if (len == n && memcmp(p, what, n) == 0) return true;
        if (!end) break;
        p = end + 1;
    }
    return false;
}

}  // namespace slimm
