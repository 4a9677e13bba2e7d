// sanitizer driver: readers + host profile on CPU
#include "../../slimm_amd/csrc/host/alignment_file.hpp"
#include "../../slimm_amd/csrc/host/sldb.hpp"
#include <cstdio>
using namespace slimm;
int main(int argc, char** argv) {
    for (int i = 1; i < argc; ++i) {
        std::string p = argv[i];
        if (p.size() > 5 && p.substr(p.size() - 5) == ".sldb") {
            SlimmDatabase db; std::string err;
            bool ok = load_slimm_database(p, db, err);
            printf("%s: %d acc=%zu tax=%zu %s\n", p.c_str(), ok, db.ac_taxid.size(), db.taxid_name.size(), err.c_str());
            if (ok) { save_slimm_database(p + ".copy", db, err); }
            continue;
        }
        AlignmentFile f;
        if (!f.open(p)) { printf("%s: %s\n", p.c_str(), f.error().c_str()); continue; }
        RecordBatch b; long n, tot = 0;
        while ((n = f.read_batch(b, 1 << 20, true)) > 0) { tot += n; b.clear(); }  // large requests: the parallel record walk
        printf("%s: refs=%zu records=%ld rc=%ld %s\n", p.c_str(), f.ref_names().size(), tot, n, f.error().c_str());
    }
}
