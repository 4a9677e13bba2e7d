"""Writers of NCBI-style inputs for slimm_build (nodes.dmp, names.dmp, accession2taxid, FASTA) derived from a synthetic
Taxonomy, so that building a database from them must give that taxonomy back.  Test infrastructure only."""
import gzip

import numpy as np

RANKS = ["strain", "species", "genus", "family", "order", "class", "phylum", "superkingdom"]


def write_dumps(tmp, tax, *, n_extra_nodes=50, seed=3, gz_fasta=False, split_acc_files=True, versioned_ids=True):
    """Returns dict(fasta=, nodes=, names=, acc=[paths]).  The tree: every non-zero lineage slot is a node of that rank;
    between a node and its parent sits a 'no rank' / 'clade' node now and then (they must be skipped); the accession's
    own taxid is a 'no rank' strain node below the species (or IS the species node when lineage[0] == lineage[1])."""
    rng = np.random.default_rng(seed)
    parent, rank = {1: 1}, {1: "no rank"}
    next_mid = [3_000_000_000]

    def link(child, par, rk):
        if child in parent:
            return
        if rng.random() < 0.3:  # an unranked node in between
            mid = next_mid[0]
            next_mid[0] += 1
            parent[mid], rank[mid] = par, ("clade" if rng.random() < 0.5 else "no rank")
            par = mid
        parent[child], rank[child] = par, rk

    lin = tax.lineage.astype(np.int64)
    for row in lin:
        up = 1
        for lv in range(7, 0, -1):
            if row[lv]:
                link(int(row[lv]), up, RANKS[lv])
                up = int(row[lv])
        if row[0] and int(row[0]) not in parent:
            link(int(row[0]), up, "no rank")
    for k in range(n_extra_nodes):  # nodes nobody asks for
        t = 4_000_000_000 + k
        parent[t], rank[t] = 1, "species"
    ids = list(parent)
    rng.shuffle(ids)
    nodes = str(tmp / "nodes.dmp")
    with open(nodes, "w") as f:
        for t in ids:
            f.write(f"{t}\t|\t{parent[t]}\t|\t{rank[t]}\t|\tXX\t|\t0\t|\t1\t|\t11\t|\t1\t|\t0\t|\t1\t|\t1\t|\t0\t|\t\t|\n")
    name_of = dict(zip(tax.tax_id.tolist(), tax.tax_name))
    names = str(tmp / "names.dmp")
    with open(names, "w") as f:
        for t in ids:
            nm = name_of.get(t, f"node {t}")
            f.write(f"{t}\t|\told name of {t}\t|\t\t|\tsynonym\t|\n")
            f.write(f"{t}\t|\t{nm}\t|\t\t|\tscientific name\t|\n")
            f.write(f"{t}\t|\tcommon {t}\t|\t\t|\tgenbank common name\t|\n")
    fasta = str(tmp / ("refs.fa.gz" if gz_fasta else "refs.fa"))
    op = gzip.open if gz_fasta else open
    with op(fasta, "wt") as f:
        for a in tax.accessions:
            head = f"{a}.1 some organism, complete genome" if versioned_ids else a
            f.write(f">{head}\nACGTACGTAC\nGGTTAACC\n")
    order = rng.permutation(len(tax.accessions))
    rows = [(tax.accessions[i], int(lin[i, 0])) for i in order]
    decoys = [(f"ZZ{k:07d}", 4_000_000_000 + (k % max(1, n_extra_nodes))) for k in range(len(rows))]
    cut = len(rows) // 2 if split_acc_files else len(rows)
    paths = []
    for k, part in enumerate([rows[:cut], rows[cut:]]):
        if not part and k:
            break
        p = str(tmp / f"part{k}.accession2taxid")
        mixed = part + decoys[k::2]
        rng.shuffle(mixed)
        with open(p, "w") as f:
            f.write("accession\taccession.version\ttaxid\tgi\n")
            for a, t in mixed:
                f.write(f"{a}\t{a}.1\t{t}\t{abs(hash(a)) % 10**9}\n")
        paths.append(p)
    return {"fasta": fasta, "nodes": nodes, "names": names, "acc": paths}
