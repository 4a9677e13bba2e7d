// sanitizer driver: readers + host profile on CPU
#include "../../slimm_amd/csrc/host/alignment_file.hpp"
#include "../../slimm_amd/csrc/host/sldb.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace slimm;
int main(int argc, char** argv) {
    if (const char* t = getenv("SAN_READER_THREADS")) AlignmentFile::settings().threads = static_cast<unsigned>(atoi(t));   // (this driver's own knob)
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
        if (p.size() > 4 && p.substr(p.size() - 4) == ".bam") {
            // read_raw (the windows `slimm` hands to slimm_push_bam_bytes): small and large windows, the compressed bytes
            // from a mapping of the file and through buffered reads, a few records taken by read_batch first
            for (int mode = 0; mode < 6; ++mode) {
                if (mode & 1) setenv("SLIMM_NO_MMAP", "1", 1); else unsetenv("SLIMM_NO_MMAP");
                const size_t cap = mode < 2 ? (1u << 20) : mode < 4 ? (3u << 20) + 12345u : (64u << 20);
                AlignmentFile g;
                if (!g.open(p)) continue;
                if (mode >= 4) { RecordBatch h; (void)g.read_batch(h, 100, false); }
                std::vector<uint8_t> buf(cap);
                long k; unsigned long long bytes = 0, sum = 0; int windows = 0;
                while ((k = g.read_raw(buf.data(), cap)) > 0) {
                    bytes += static_cast<unsigned long long>(k); ++windows;
                    for (long i = 0; i < k; i += 509) sum += buf[static_cast<size_t>(i)];
                    if (g.raw_exhausted()) break;
                }
                printf("%s: read_raw mode %d cap %zu: %llu bytes in %d windows (sum %llu) rc=%ld %s\n", p.c_str(), mode, cap, bytes,
                       windows, sum, k, g.error().c_str());
            }
            unsetenv("SLIMM_NO_MMAP");
            // read_blocks (the windows `slimm` hands to slimm_push_bgzf_blocks: whole BGZF blocks by pread), alternating with
            // read_raw, small and large buffers, an inflated-size limit that ends windows early
            for (int mode = 0; mode < 3; ++mode) {
                const size_t cap = mode == 0 ? (1u << 20) : mode == 1 ? (2u << 20) + 777u : (32u << 20);
                const size_t lim = mode == 1 ? (3u << 20) : (1900u << 20);
                AlignmentFile g;
                if (!g.open(p)) continue;
                std::vector<uint8_t> buf(cap);
                long k = 1; unsigned long long cbytes = 0, ibytes = 0; int windows = 0;
                while (k > 0) {
                    if (g.can_read_blocks() && (windows % 3) != 2) {
                        size_t inf = 0;
                        k = g.read_blocks(buf.data(), cap, lim, &inf);
                        if (k > 0) { cbytes += static_cast<unsigned long long>(k); ibytes += inf; }
                    } else {
                        k = g.read_raw(buf.data(), cap);
                        if (k > 0) ibytes += static_cast<unsigned long long>(k);
                        if (k > 0 && g.raw_exhausted()) break;
                    }
                    ++windows;
                }
                printf("%s: read_blocks mode %d: %llu compressed + inflated to %llu bytes in %d windows rc=%ld %s\n", p.c_str(), mode, cbytes,
                       ibytes, windows, k, g.error().c_str());
            }
        }
        if (p.size() > 4 && p.substr(p.size() - 4) == ".sam") {   // read_text (slimm_push_sam_bytes): the text behind the header
            for (size_t cap : {size_t(1) << 16, (size_t(1) << 20) + 13u, size_t(64) << 20}) {
                AlignmentFile g;
                if (!g.open(p)) continue;
                std::vector<uint8_t> buf(cap);
                long k; unsigned long long bytes = 0, lines = 0; int first = -1;
                while ((k = g.read_text(buf.data(), cap)) > 0) {
                    if (first < 0) first = buf[0];
                    bytes += static_cast<unsigned long long>(k);
                    for (long i = 0; i < k; ++i) lines += buf[static_cast<size_t>(i)] == 10;
                }
                printf("%s: read_text cap %zu: %llu bytes, %llu lines, first byte %d, rc=%ld %s\n", p.c_str(), cap, bytes, lines, first, k,
                       g.error().c_str());
            }
        }
    }
}
