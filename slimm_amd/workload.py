"""Host-side containers for one SLIMM input: header contigs, taxonomy database and the
decoded alignment-record stream, in the shape the C ABI (include/slimm_hip.h) takes them.

Mirrors what the reference holds after `read_bam_file` + `load_slimm_database`
(reference src/slimm.hpp:399-445, src/misc.hpp:77-100): contig names/lengths in header
order (= BAM refID), `ac__taxid` (accession -> 8 taxids) and `taxid__name` (taxid -> rank, name).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

LINEAGE_LEN = 8  # reference src/misc.hpp:4
RANKS = ["strain", "species", "genus", "family", "order", "class", "phylum", "superkingdom"]  # misc.hpp:24-35

FLAG_UNMAPPED = 0x4
FLAG_FIRST = 0x40
FLAG_LAST = 0x80


def accession_of(name: str) -> str:
    """Text before the first whitespace, '.' or '|' (reference src/misc.hpp:415-422)."""
    for i, c in enumerate(name):
        if c in ".| \t\n\r\v\f":
            return name[:i]
    return name


def read_keys_from_names(names: Sequence[str]) -> np.ndarray:
    """62-bit keys for qNames (equal names <=> equal keys up to hash collisions).

    The C ABI folds the mate number into the two low bits, so only 62 bits are significant.
    """
    import xxhash

    out = np.empty(len(names), dtype=np.uint64)
    for i, n in enumerate(names):
        out[i] = xxhash.xxh64_intdigest(n) >> 2
    return out


def canonical_identity(names: Sequence[str], flags) -> "tuple[List[str], np.ndarray]":
    """(base names, flags carrying the base's mate bit) of records as they stand in a file (quirk Q18).

    The reference keys a read by the string qName + ".1" / ".2" / "" (src/slimm.hpp:204-208): an unflagged record of a
    read named "N.1" is the same read as a first-in-pair record of "N".  Mirrors slimm_host_canonical_read_name
    (include/slimm_hip.h; csrc/read_identity.h), which the library's own readers apply.
    """
    flags = np.array(flags, dtype=np.uint16)
    bases = list(names)
    for i, n in enumerate(names):
        if flags[i] & 0xC0:
            continue
        if len(n) >= 2 and n[-2] == "." and n[-1] in "12":
            bases[i] = n[:-2]
            flags[i] |= 0x40 if n[-1] == "1" else 0x80
    return bases, flags


@dataclass
class Taxonomy:
    """The slimm_database (reference src/misc.hpp:77-100)."""

    accessions: List[str]
    lineage: np.ndarray  # [n_acc, 8] uint32: own, species, genus, family, order, class, phylum, superkingdom
    tax_id: np.ndarray  # [n_tax] uint32
    tax_rank: np.ndarray  # [n_tax] uint32 (0..8)
    tax_name: List[str]

    def __post_init__(self):
        self.lineage = np.ascontiguousarray(self.lineage, dtype=np.uint32).reshape(-1, LINEAGE_LEN)
        self.tax_id = np.ascontiguousarray(self.tax_id, dtype=np.uint32)
        self.tax_rank = np.ascontiguousarray(self.tax_rank, dtype=np.uint32)
        assert len(self.accessions) == self.lineage.shape[0]
        assert len(self.tax_name) == self.tax_id.shape[0] == self.tax_rank.shape[0]

    def lineage_for_header(self, ref_names: Sequence[str]) -> np.ndarray:
        """Dense [R, 8] lineage table in header order.

        A contig whose accession is missing from the database gets an all-zero row
        (reference src/slimm.hpp:433-442, quirk Q13).
        """
        index = {a: i for i, a in enumerate(self.accessions)}
        out = np.zeros((len(ref_names), LINEAGE_LEN), dtype=np.uint32)
        for i, n in enumerate(ref_names):
            j = index.get(accession_of(n))
            if j is not None:
                out[i] = self.lineage[j]
        return out


@dataclass
class Records:
    """Decoded alignment records in file order (reference src/slimm.hpp:194-213 reads exactly these fields)."""

    read_key: np.ndarray  # uint64, identity of qName (62 significant bits)
    flag: np.ndarray  # uint16 SAM flag (with the mate bit of the canonical identity: canonical_identity above)
    ref_id: np.ndarray  # int32 BAM refID (-1 = none)
    begin_pos: np.ndarray  # int32 0-based position
    qname: Optional[List[str]] = None  # only for small cases / the oracle: the names as they stand in the file
    file_flag: Optional[np.ndarray] = None  # the flags as they stand in the file, where they differ from `flag` (Q18)

    def __post_init__(self):
        self.read_key = np.ascontiguousarray(self.read_key, dtype=np.uint64)
        self.flag = np.ascontiguousarray(self.flag, dtype=np.uint16)
        self.ref_id = np.ascontiguousarray(self.ref_id, dtype=np.int32)
        self.begin_pos = np.ascontiguousarray(self.begin_pos, dtype=np.int32)
        n = self.read_key.shape[0]
        assert self.flag.shape[0] == n and self.ref_id.shape[0] == n and self.begin_pos.shape[0] == n

    def __len__(self) -> int:
        return int(self.read_key.shape[0])

    def take(self, idx) -> "Records":
        q = [self.qname[i] for i in idx] if self.qname is not None else None
        ff = self.file_flag[idx] if self.file_flag is not None else None
        return Records(self.read_key[idx], self.flag[idx], self.ref_id[idx], self.begin_pos[idx], q, ff)

    def flags_in_file(self) -> np.ndarray:
        return self.flag if self.file_flag is None else self.file_flag


@dataclass
class Options:
    """arg_options (reference src/slimm.hpp:49-87) restricted to what the hot path reads."""

    bin_width: int = 0
    min_reads: int = 0
    cov_cut_off: float = 0.95
    abundance_cut_off: float = 0.01
    rank: str = "species"


@dataclass
class Workload:
    ref_names: List[str]
    ref_len: np.ndarray  # uint32 [R]
    taxonomy: Taxonomy
    records: Records
    avg_read_len: int
    options: Options = field(default_factory=Options)
    name: str = ""
    grouped: bool = True  # records of one read are contiguous (mapper order)

    def __post_init__(self):
        self.ref_len = np.ascontiguousarray(self.ref_len, dtype=np.uint32)
        assert len(self.ref_names) == self.ref_len.shape[0]

    @property
    def n_refs(self) -> int:
        return len(self.ref_names)

    def lineage(self) -> np.ndarray:
        return self.taxonomy.lineage_for_header(self.ref_names)
