// Reader / writer of the SLIMM database file (.sldb), SURVEY.md section 8 f2.
//
// The reference saves `slimm_database` with cereal's BinaryOutputArchive (reference src/misc.hpp:77-100, 178-195).
// cereal is not part of the reference checkout; the layout below is cereal's portable-less binary archive for these
// types (native little-endian, 64-bit size tags, enums as their underlying 32-bit type) as recorded in SURVEY.md
// Appendix B.  It is self-consistent with tests/sldb_io.py; it has never seen a real upstream .sldb (none exists here).
//
//   u64 n1;  n1 x { u64 len; char accession[len];  u64 cnt;  u32 lineage[cnt]; }                 // ac__taxid
//   u64 n2;  n2 x { u32 taxid;  u32 rank;  u64 len;  char name[len]; }                           // taxid__name
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace slimm {

struct SlimmDatabase {
    std::unordered_map<std::string, std::vector<uint32_t>> ac_taxid;                  // accession -> 8 taxids
    std::unordered_map<uint32_t, std::pair<uint32_t, std::string>> taxid_name;       // taxid -> (rank, name)
};

bool load_slimm_database(const std::string& path, SlimmDatabase& db, std::string& err);
bool save_slimm_database(const std::string& path, const SlimmDatabase& db, std::string& err);

}  // namespace slimm
