// Name helpers shared by the two command-line tools.
#pragma once
#include <cstdint>
#include <string>

namespace slimm {

// get_accession_id, reference src/misc.hpp:415-422: a sequence name up to its first '.', '|' or white space
inline std::string get_accession_id(const std::string& name) {
    size_t i = 0;
    while (i < name.size()) {
        unsigned char c = static_cast<unsigned char>(name[i]);
        if (c == '.' || c == '|' || c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f') break;
        ++i;
    }
    return name.substr(0, i);
}

// to_taxa_ranks, reference src/misc.hpp:37-48 (strain 0 ... superkingdom 7, anything else "intermidiate" 8)
inline uint32_t to_taxa_ranks(const std::string& s) {
    static const char* const names[] = {"strain", "species", "genus", "family", "order", "class", "phylum", "superkingdom"};
    for (uint32_t i = 0; i < 8; ++i)
        if (s == names[i]) return i;
    return 8;
}

}  // namespace slimm
