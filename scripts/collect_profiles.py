#!/usr/bin/env python3
"""Merge several `<sample>_profile.tsv` files (reference src/slimm.hpp:733-843 / this repo's `slimm` command line) into
one table: one row per (taxa_level, taxa_id, linage), one abundance and one read-count column per sample.

    python scripts/collect_profiles.py [-o merged_profile.tsv] a_profile.tsv b_profile.tsv ...

Same job as the reference's `collect_profiles.py`.  That script predates the current
five-column profile (it keys rows on column 3 and copies column 5, newline included, as the sample value); this one
reads the five columns by name, fills samples that lack a taxon with 0, and sorts by level and then by the first
sample's abundance, descending, which is what the reference's sort amounts to.  No pandas needed.
"""
import argparse
import os
import sys


def sample_name(path):
    base = os.path.basename(path)
    base = base[:base.rfind(".")] if "." in base else base
    return base[:-len("_profile")] if base.endswith("_profile") else base


def read_profile(path):
    rows = {}
    with open(path) as f:
        header = f.readline().rstrip("\n").split("\t")
        want = ["taxa_level", "taxa_id", "linage", "abundance", "read_count"]
        if header != want:
            raise SystemExit(f"{path}: not a slimm profile (header {header})")
        for line in f:
            line = line.rstrip("\n")
            if not line:
                continue
            level, taxid, linage, ab, reads = line.split("\t")
            rows[(level, taxid, linage)] = (float(ab), int(reads))
    return rows


def merge(paths):
    names = [sample_name(p) for p in paths]
    tables = [read_profile(p) for p in paths]
    keys = set()
    for t in tables:
        keys.update(t)
    rank_order = {r: i for i, r in enumerate(["superkingdom", "phylum", "class", "order", "family", "genus", "species",
                                              "strain"])}
    def sort_key(k):
        return (rank_order.get(k[0], 99), [-t.get(k, (0.0, 0))[0] for t in tables], k[1], k[2])
    out = ["\t".join(["taxa_level", "taxa_id", "linage"] + [f"{n}_abundance" for n in names] + [f"{n}_read_count" for n in names])]
    for k in sorted(keys, key=sort_key):
        ab = ["%.6g" % t.get(k, (0.0, 0))[0] for t in tables]
        rd = [str(t.get(k, (0.0, 0))[1]) for t in tables]
        out.append("\t".join(list(k) + ab + rd))
    return "\n".join(out) + "\n"


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("-o", "--output", default="merged_profile.tsv")
    ap.add_argument("profiles", nargs="+")
    a = ap.parse_args(argv)
    text = merge(a.profiles)
    if a.output == "-":
        sys.stdout.write(text)
    else:
        with open(a.output, "w") as f:
            f.write(text)
    return 0


if __name__ == "__main__":
    sys.exit(main())
