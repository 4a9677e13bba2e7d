#include "sldb.hpp"

#include <cstdio>
#include <cstring>

namespace slimm {

namespace {
struct Reader {
    FILE* f;
    bool ok = true;
    template <typename T>
    T get() {
        T v{};
        if (ok && fread(&v, sizeof(T), 1, f) != 1) ok = false;
        return v;
    }
    std::string str(uint64_t n) {
        std::string s;
        if (!ok) return s;
        if (n > (1ull << 32)) {
            ok = false;
            return s;
        }
        s.resize(n);
        if (n && fread(&s[0], 1, n, f) != n) ok = false;
        return s;
    }
};
}  // namespace

bool load_slimm_database(const std::string& path, SlimmDatabase& db, std::string& err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        err = "cannot open database " + path;
        return false;
    }
    Reader r{f};
    db.ac_taxid.clear();
    db.taxid_name.clear();
    uint64_t n1 = r.get<uint64_t>();
    for (uint64_t i = 0; r.ok && i < n1; ++i) {
        uint64_t len = r.get<uint64_t>();
        std::string acc = r.str(len);
        uint64_t cnt = r.get<uint64_t>();
        if (cnt > 64) r.ok = false;
        std::vector<uint32_t> lin(r.ok ? cnt : 0);
        for (uint64_t k = 0; r.ok && k < cnt; ++k) lin[k] = r.get<uint32_t>();
        if (r.ok) db.ac_taxid[acc] = std::move(lin);
    }
    uint64_t n2 = r.get<uint64_t>();
    for (uint64_t i = 0; r.ok && i < n2; ++i) {
        uint32_t taxid = r.get<uint32_t>();
        uint32_t rank = r.get<uint32_t>();
        uint64_t len = r.get<uint64_t>();
        std::string name = r.str(len);
        if (r.ok) db.taxid_name[taxid] = std::make_pair(rank, std::move(name));
    }
    fclose(f);
    if (!r.ok) {
        err = "truncated or malformed database " + path;
        return false;
    }
    return true;
}

bool save_slimm_database(const std::string& path, const SlimmDatabase& db, std::string& err) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) {
        err = "cannot write database " + path;
        return false;
    }
    auto put64 = [&](uint64_t v) { fwrite(&v, 8, 1, f); };
    auto put32 = [&](uint32_t v) { fwrite(&v, 4, 1, f); };
    put64(db.ac_taxid.size());
    for (auto& kv : db.ac_taxid) {
        put64(kv.first.size());
        fwrite(kv.first.data(), 1, kv.first.size(), f);
        put64(kv.second.size());
        for (uint32_t t : kv.second) put32(t);
    }
    put64(db.taxid_name.size());
    for (auto& kv : db.taxid_name) {
        put32(kv.first);
        put32(kv.second.first);
        put64(kv.second.second.size());
        fwrite(kv.second.second.data(), 1, kv.second.second.size(), f);
    }
    bool ok = fclose(f) == 0;
    if (!ok) err = "write error on " + path;
    return ok;
}

}  // namespace slimm
