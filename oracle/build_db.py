"""TEST INFRASTRUCTURE ONLY -- NOT PART OF THE PRODUCT.

A plain-Python restatement of the reference's database builder (reference src/slimm_build.cpp:151-346), the checker
for slimm_amd/csrc/host/slimm_build_main.cpp.  Line handling follows the reference's stringstream calls literally,
including what they do on lines they were not written for (a header line, a short line).  PARITY UNPINNED: the
reference holds no fixture for its builder and cannot be built here (SeqAn and cereal are not vendored).
"""
import gzip

RANKS = ["strain", "species", "genus", "family", "order", "class", "phylum", "superkingdom"]
_WS = " \t\n\r\v\f"


def to_taxa_ranks(s):  # reference src/misc.hpp:37-48
    return RANKS.index(s) if s in RANKS else 8


def get_accession_id(name):  # reference src/misc.hpp:415-422
    for i, c in enumerate(name):
        if c in ".|" or c in _WS:
            return name[:i]
    return name


def _open(path):
    with open(path, "rb") as f:
        magic = f.read(2)
    return gzip.open(path, "rt") if magic == b"\x1f\x8b" else open(path, "r")


class _Stream:
    """The few std::stringstream operations the reference uses."""

    def __init__(self, s):
        self.s, self.i, self.ok = s, 0, True

    def getline(self, old, delim="\t"):
        if not self.ok:
            return old  # a failed stream leaves the string alone
        if self.i >= len(self.s):
            self.ok = False  # nothing extracted: failbit; the string has been erased
            return ""
        j = self.s.find(delim, self.i)
        if j < 0:
            out, self.i = self.s[self.i:], len(self.s)
        else:
            out, self.i = self.s[self.i:j], j + 1
        return out

    def u32(self, old):
        if not self.ok:
            return old
        i = self.i
        while i < len(self.s) and self.s[i] in _WS:
            i += 1
        j = i + 1 if i < len(self.s) and self.s[i] == "+" else i
        k = j
        while k < len(self.s) and self.s[k].isdigit():
            k += 1
        if k == j:
            self.ok = False
            # the sentry fails (value untouched) when only white space is left; num_get stores 0 when what follows is
            # not a number (C++11)
            return old if i >= len(self.s) else 0
        self.i = k
        return min(int(self.s[j:k]), 0xFFFFFFFF)


def accessions_of_fasta(path):  # get_accession_numbers, reference src/slimm_build.cpp:151-170
    out = set()
    with _open(path) as f:
        for line in f:
            if line.startswith(">"):
                out.add(get_accession_id(line[1:].rstrip("\r\n")))
    return out


def build(fasta, acc2taxid_paths, nodes_path, names_path, batch=1000000):
    """Returns (ac__taxid: {accession: [8 taxids]}, taxid__name: {taxid: (rank, name)}, missed: sorted list)."""
    accessions = accessions_of_fasta(fasta)
    ac_taxid = {}
    for path in acc2taxid_paths:  # get_taxid_from_accession, reference src/slimm_build.cpp:223-278
        if not accessions:
            break
        with _open(path) as f:
            lines = iter(f)
            while True:
                mapping, n = {}, 0  # get_batch_mappings_ac__taxid, :175-195
                taxid, ac, ignore = 0, "", ""
                for line in lines:
                    st = _Stream(line.rstrip("\n"))
                    ac = st.getline(ac)
                    ignore = st.getline(ignore)
                    taxid = st.u32(taxid)
                    mapping[ac] = taxid
                    n += 1
                    if n >= batch:
                        break
                if n == 0:
                    break
                for a in sorted(accessions):
                    if a in mapping:
                        ac_taxid[a] = [mapping[a]] + [0] * 7
                        accessions.discard(a)
    missed = sorted(accessions)

    parent, names = {}, {}  # fill_name_taxid_linage, reference src/slimm_build.cpp:283-346
    taxid = par = 0
    rank = ignore = name = ""
    with _open(nodes_path) as f:
        for line in f:
            st = _Stream(line.rstrip("\n"))
            taxid = st.u32(taxid)
            ignore = st.getline(ignore)
            ignore = st.getline(ignore)
            par = st.u32(par)
            ignore = st.getline(ignore)
            ignore = st.getline(ignore)
            rank = st.getline(rank)
            parent[taxid] = (to_taxa_ranks(rank), par)
    with _open(names_path) as f:
        for line in f:
            line = line.rstrip("\n")
            if "scientific name" in line:
                st = _Stream(line)
                taxid = st.u32(taxid)
                ignore = st.getline(ignore)
                ignore = st.getline(ignore)
                name = st.getline(name)
                names[taxid] = name
    taxid_name = {}
    for a, lin in ac_taxid.items():
        tid = lin[0]
        taxid_name[tid] = (0, names.get(tid, ""))
        while tid != 1:
            if tid not in parent:
                break
            rk, up = parent[tid]
            if 1 <= rk <= 7:
                lin[rk] = tid
                taxid_name[tid] = (rk, names.get(tid, ""))
            tid = up
    return ac_taxid, taxid_name, missed
