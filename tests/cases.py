"""Hand-made micro-cases for the alignment-to-profile path.

`tiny` and `holes` are the two known-answer cases of SURVEY.md Appendix C.1 / C.2: their
expected outputs were observed from the reference's own code by the surveyor and are the only
reference-produced vectors that exist for this path (the reference ships no tests).
The remaining cases exercise single quirks of SURVEY.md Appendix A; their expected values
are derived by hand in the tests that use them.
"""
from __future__ import annotations

import json
import os
import random
from typing import Dict, List, Tuple

import numpy as np

from slimm_amd.workload import Options, Records, Taxonomy, Workload, canonical_identity, read_keys_from_names

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

RANK_NAMES = ["strain", "species", "genus", "family", "order", "class", "phylum", "superkingdom"]


def taxonomy_from_lineages(acc_lineage: Dict[str, List[int]], extra_rank: Dict[int, int] = None,
                           unnamed: Tuple[int, ...] = ()) -> Taxonomy:
    """Database where every taxid in a lineage has a name and the rank implied by its column.

    Like slimm_build (reference src/slimm_build.cpp:329-342): the accession's own taxid is a strain unless
    the same taxid is also its species (then it is recorded as a species).
    """
    rank_of: Dict[int, int] = {}
    for lin in acc_lineage.values():
        for lv in range(7, -1, -1):  # lower columns win, except own==species which stays species
            t = lin[lv]
            if t == 0:
                continue
            if lv == 0 and lin[1] == t:
                continue
            rank_of[t] = lv
    if extra_rank:
        rank_of.update(extra_rank)
    tids = sorted(rank_of)
    names = ["" if t in unnamed else f"{RANK_NAMES[rank_of[t]]}_{t}" for t in tids]
    accs = list(acc_lineage)
    return Taxonomy(accs, np.array([acc_lineage[a] for a in accs], dtype=np.uint32), np.array(tids, dtype=np.uint32),
                    np.array([rank_of[t] for t in tids], dtype=np.uint32), names)


def records_from_sam(rows: List[Tuple[str, int, str, int]], ref_names: List[str]) -> Records:
    """rows = (qname, flag, rname or '*', 1-based POS) as they would stand in a SAM file."""
    idx = {n: i for i, n in enumerate(ref_names)}
    q = [r[0] for r in rows]
    raw = np.array([r[1] for r in rows], dtype=np.uint16)
    base, flag = canonical_identity(q, raw)  # Q18: what a producer hands the C ABI; the oracle gets names + flags as written
    return Records(read_keys_from_names(base), flag, np.array([idx.get(r[2], -1) for r in rows], dtype=np.int32),
                   np.array([r[3] - 1 for r in rows], dtype=np.int32), q, raw if (raw != flag).any() else None)


def tiny_case() -> Workload:
    """SURVEY.md Appendix C.1."""
    lin = {
        "ACC_A": [101, 11, 21, 31, 41, 51, 61, 2],
        "ACC_B": [102, 11, 21, 31, 41, 51, 61, 2],
        "ACC_C": [12, 12, 21, 31, 41, 51, 61, 2],
        "ACC_D": [103, 13, 22, 32, 42, 52, 62, 2157],
        "ACC_E": [104, 14, 22, 32, 42, 52, 62, 2157],
    }
    names = ["ACC_A.1", "ACC_B.1", "ACC_C.1", "ACC_D.1", "ACC_E.1"]
    lens = [1000, 1500, 2000, 1200, 900]
    A, B, Cc, D, E = names
    rows = [("r1", 0, A, 10), ("r1", 256, A, 700), ("r2", 0, A, 100), ("r2", 256, B, 200), ("r3", 0, A, 300),
            ("r3", 256, Cc, 400), ("r4", 0, A, 500), ("r4", 256, D, 600), ("r5", 65, B, 50), ("r5", 129, B, 900),
            ("r6", 4, "*", 0)]
    rng = random.Random(7)
    for i in range(60):
        ref = rng.choice([B, Cc, D])
        pos = rng.randint(1, 850)
        rows.append((f"u{i}", 0, ref, pos))
    rows += [("r7", 0, E, 880), ("r7", 256, D, 5), ("r8", 0, Cc, 1990)]
    return Workload(names, np.array(lens, dtype=np.uint32), taxonomy_from_lineages(lin), records_from_sam(rows, names),
                    avg_read_len=50, options=Options(bin_width=100), name="tiny")


def holes_case() -> Workload:
    """SURVEY.md Appendix C.2."""
    lin = {
        "H1": [301, 0, 21, 31, 41, 51, 61, 2],
        "H2": [302, 0, 21, 31, 41, 51, 61, 2],
        "S1": [201, 15, 21, 31, 41, 51, 61, 2],
        "S2": [201, 15, 21, 31, 41, 51, 61, 2],
    }
    names = ["H1.1", "H2.1", "S1.1", "S2.1", "NODB.1"]
    rows = []
    for i in range(10):
        for q, r in zip("abcde", names):
            rows.append((f"{q}{i}", 0, r, 1 + 90 * i))
    H1, H2, S1, S2, NODB = names
    rows += [("m1", 0, H1, 500), ("m1", 256, H2, 500), ("m2", 0, S1, 500), ("m2", 256, S2, 500), ("m3", 0, S1, 960),
             ("m3", 256, NODB, 990), ("edge", 0, H1, 990)]
    return Workload(names, np.full(5, 1000, dtype=np.uint32), taxonomy_from_lineages(lin),
                    records_from_sam(rows, names), avg_read_len=50, options=Options(bin_width=100), name="holes")


def q18_case(order=None) -> Workload:
    """Quirk Q18: the reference's read key is the STRING qName + ".1" / ".2" / "" (src/slimm.hpp:204-208), so an unflagged
    record of a read literally named "N.1" and a first-in-pair record of "N" are ONE read.  Records of one key string are
    adjacent here (name-grouped input in the canonical sense); order = a permutation for the any-order path."""
    lin = {"X": [101, 11, 21, 31, 41, 51, 61, 2], "Y": [102, 12, 21, 31, 41, 51, 61, 2]}
    names = ["X.1", "Y.1"]
    X, Y = names
    rows = [
        ("N", 0x41, X, 10), ("N.1", 0, Y, 10),            # one read "N.1" on X and Y
        ("M", 0x81, X, 300), ("M.2", 0x100, X, 500),      # one read "M.2"; X counted once, at its first bin (Q1)
        ("K.1", 0x41, Y, 100),                             # "K.1.1": a read of its own
        ("K.1", 0, Y, 600), ("K", 0x41, X, 700),           # both "K.1": one read on Y and X
        ("Q.1", 0x81, X, 800),                             # "Q.1.2"
        ("Q", 0x41, Y, 850),                               # "Q.1": apart from "Q.1.2"
        ("T", 0xC1, X, 400), ("T.1", 0, X, 900),           # both flags: first wins -> "T.1" twice, one read
        ("P.3", 0, Y, 200), ("P.12", 0, Y, 250),           # no suffix rule applies
        ("W.2", 0, X, 50), ("W", 0x41, X, 150), ("W", 0x81, Y, 50),   # "W.2", "W.1", "W.2": two reads
        ("U.1", 0x4, "*", 0), ("U", 0x41, Y, 950),         # unmapped record of "U.1" + a mapped one
    ]
    if order is not None:
        rows = [rows[i] for i in order]
    return Workload(names, np.array([1000, 1000], dtype=np.uint32), taxonomy_from_lineages(lin), records_from_sam(rows, names),
                    avg_read_len=50, options=Options(bin_width=100, cov_cut_off=0.99), name="q18")


# by hand: 18 records, 17 mapped; reads = N.1 {X,Y}, M.2 {X}, K.1.1 {Y}, K.1 {Y,X}, Q.1.2 {X}, Q.1 {Y}, T.1 {X}, P.3 {Y},
# P.12 {Y}, W.2 {X,Y}, W.1 {X}, U.1 {Y} = 12 reads, 9 of them on one reference
Q18_EXPECTED = {"hits": 17, "matches": 12, "uniq_matches": 9}


def q18_apart_case(tail=("r.1", "r.2")) -> Workload:
    """Q18 in a file that IS grouped by QNAME while the reference's key strings are not adjacent (the round-5 judge's file):
    `r`/0x40 + `r`/0x80 on ACC_A, fifty other reads, then the unflagged `r.1` on ACC_B and `r.2` on ACC_C.  The reference
    (src/slimm.hpp:204-211: a hash map keyed by qName + ".1" / ".2") holds the reads "r.1" = {ACC_A, ACC_B} and
    "r.2" = {ACC_A, ACC_C}: 52 reads, 50 of them on one reference -- a reader that compares adjacent names only sees 54."""
    t = tiny_case()
    A, B, Cc, D, _E = t.ref_names
    rows = [("r", 0x41, A, 10), ("r", 0x81, A, 300)]
    rng = random.Random(11)
    for i in range(50):
        rows.append((f"u{i}", 0, rng.choice([B, Cc, D]), rng.randint(1, 850)))
    refs = {"r.1": (B, 20), "r.2": (Cc, 500)}
    for q in tail:
        rows.append((q, 0, refs[q][0], refs[q][1]))
    return Workload(t.ref_names, t.ref_len, t.taxonomy, records_from_sam(rows, t.ref_names), avg_read_len=50,
                    options=Options(bin_width=100), name="q18-apart", grouped=False)


Q18_APART_EXPECTED = {"hits": 54, "matches": 52, "uniq_matches": 50}


# Expected outputs transcribed from SURVEY.md Appendix C (reference-observed).
TINY_EXPECTED = {
    "hits": 73, "matches": 68, "uniq_matches": 64, "uniq_matches2": 65, "n_valid": 4,
    "cutoffs": [0.363636, 0.0909091],
    # per ref: reads_count, uniq1, uniq2, bins, bins>0, uniq1 bins>0, uniq2 bins>0
    "refs": {"ACC_A": [4, 1, 1, 11, 4, 1, 1], "ACC_B": [24, 23, 23, 16, 10, 10, 10], "ACC_C": [18, 17, 17, 21, 8, 8, 8],
             "ACC_D": [25, 23, 24, 13, 8, 8, 8], "ACC_E": [1, 0, 0, 10, 1, 0, 0]},
    "invalid": ["ACC_E"],
    "profile": {"11": [36.7647, 25], "12": [25.0, 17], "13": [35.2941, 24], "21*": [1.47059, 1], "0*": [1.47059, 1]},
}
HOLES_EXPECTED = {
    "hits": 57, "matches": 54, "uniq_matches": 51, "uniq_matches2": 51, "n_valid": 5,
    "cutoffs": [0.818182, 0.818182],
    "refs3": {"H1": [12, 11, 11], "H2": [11, 10, 10], "S1": [12, 10, 10], "S2": [11, 10, 10], "NODB": [11, 10, 10]},
    "H1_cov": [1, 1, 2, 1, 1, 2, 1, 1, 1, 0, 1],
    "profile": {"15": [38.8889, 21], "21*": [42.5926, 23], "0*": [18.5185, 10]},
}


def workload_to_json(w: Workload) -> dict:
    t = w.taxonomy
    return {
        "name": w.name, "ref_names": w.ref_names, "ref_len": w.ref_len.tolist(), "avg_read_len": w.avg_read_len,
        "options": vars(w.options),
        "db": {"accessions": t.accessions, "lineage": t.lineage.tolist(), "tax_id": t.tax_id.tolist(),
               "tax_rank": t.tax_rank.tolist(), "tax_name": t.tax_name},
        "records": {"qname": w.records.qname, "flag": w.records.flags_in_file().tolist(), "ref_id": w.records.ref_id.tolist(),
                    "begin_pos": w.records.begin_pos.tolist()},
    }


def workload_from_json(d: dict) -> Workload:
    db = d["db"]
    t = Taxonomy(db["accessions"], np.array(db["lineage"], dtype=np.uint32), np.array(db["tax_id"], dtype=np.uint32),
                 np.array(db["tax_rank"], dtype=np.uint32), db["tax_name"])
    r = d["records"]
    raw = np.array(r["flag"], dtype=np.uint16)
    base, flag = canonical_identity(r["qname"], raw)
    rec = Records(read_keys_from_names(base), flag, np.array(r["ref_id"], dtype=np.int32),
                  np.array(r["begin_pos"], dtype=np.int32), r["qname"], raw if (raw != flag).any() else None)
    return Workload(d["ref_names"], np.array(d["ref_len"], dtype=np.uint32), t, rec, d["avg_read_len"],
                    Options(**d["options"]), d["name"])


def load_golden(name: str):
    with open(os.path.join(GOLDEN_DIR, name + ".json")) as f:
        d = json.load(f)
    return workload_from_json(d["input"]), d["expected"], d.get("provenance", "")
