"""Independent Python writers of SAM, BAM (BGZF) and .sldb files for testing the C++ readers.

Written from the SAM/BAM specification and SURVEY.md Appendix B; they share no code with
slimm_amd/csrc/host/alignment_file.cpp or sldb.cpp.  Test infrastructure only.
"""
import struct
import zlib

import numpy as np


def qnames_of(records):
    if records.qname is not None:
        return list(records.qname)
    return [f"r{int(k)}" for k in records.read_key]


def _file_flags(records):
    """The flags as they stand in the file (Records.file_flag where the canonical identity changed them: Q18)."""
    ff = getattr(records, "file_flag", None)
    return records.flag if ff is None else ff


def sam_header(ref_names, ref_len, hd="@HD\tVN:1.6\tSO:unsorted\tGO:query"):
    lines = [hd] if hd else []
    lines += [f"@SQ\tSN:{n}\tLN:{int(l)}" for n, l in zip(ref_names, ref_len)]
    return "\n".join(lines) + "\n"


def write_sam(path, ref_names, ref_len, records, read_len=100, hd="@HD\tVN:1.6\tSO:unsorted\tGO:query"):
    q = qnames_of(records)
    fflag = _file_flags(records)
    seq = "A" * read_len
    with open(path, "w") as f:
        f.write(sam_header(ref_names, ref_len, hd))
        for i in range(len(records)):
            r = int(records.ref_id[i])
            rn = ref_names[r] if r >= 0 else "*"
            f.write(f"{q[i]}\t{int(fflag[i])}\t{rn}\t{int(records.begin_pos[i]) + 1}\t255\t{read_len}M\t*\t0\t0\t{seq}\t*\n")


def _bgzf_block(data: bytes) -> bytes:
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25  # 12 header + 6 extra + 8 trailer - 1
    return (b"\x1f\x8b\x08\x04" + b"\x00\x00\x00\x00" + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize)
            + comp + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


def bam_record_bytes(records, read_len=100, irregular_seed=None, l_seq_of=None) -> bytes:
    """The alignment records of a BAM file as they follow its header (SAM/BAM specification 4.2), concatenated:
    block_size | refID | pos | l_read_name | mapq | bin | n_cigar_op | flag | l_seq | next_refID | next_pos | tlen |
    read_name | cigar | seq | qual (| tags).  irregular_seed: record sizes vary (sequence lengths 0 .. 3 x read_len, one to
    four CIGAR operations, a few optional tag bytes), so that record boundaries fall anywhere.  l_seq_of: {record index:
    sequence length} for single records of another size."""
    q = qnames_of(records)
    fflag = _file_flags(records)
    rng = np.random.default_rng(irregular_seed) if irregular_seed is not None else None
    out = bytearray()
    seq = bytes([0x11] * ((read_len + 1) // 2))
    qual = bytes([0xff] * read_len)
    cigar = struct.pack("<I", (read_len << 4) | 0)
    for i in range(len(records)):
        name = q[i].encode() + b"\x00"
        if l_seq_of and i in l_seq_of:
            l_seq = int(l_seq_of[i])
            cg, sq, ql, tags = struct.pack("<I", (l_seq << 4) | 0), bytes([0x11]) * ((l_seq + 1) // 2), bytes([0x28]) * l_seq, b""
        elif rng is None:
            l_seq, cg, sq, ql, tags = read_len, cigar, seq, qual, b""
        else:
            l_seq = int(rng.integers(0, 3 * read_len + 1))
            n_op = int(rng.integers(1, 5))
            cg = b"".join(struct.pack("<I", (int(rng.integers(1, 200)) << 4) | int(rng.integers(0, 5))) for _ in range(n_op))
            sq = bytes(rng.integers(0, 256, size=(l_seq + 1) // 2, dtype=np.uint8))
            ql = bytes(rng.integers(0, 42, size=l_seq, dtype=np.uint8))
            tags = b"NMC" + bytes([int(rng.integers(0, 200))]) if rng.random() < 0.5 else b""
        body = struct.pack("<iiBBHHHIiii", int(records.ref_id[i]), int(records.begin_pos[i]), len(name), 255, 4680, len(cg) // 4,
                           int(fflag[i]), l_seq, -1, -1, 0) + name + cg + sq + ql + tags
        out += struct.pack("<i", len(body)) + body
    return bytes(out)


def write_bam(path, ref_names, ref_len, records, read_len=100, hd="@HD\tVN:1.6\tSO:unsorted\tGO:query", irregular_seed=None,
              l_seq_of=None):
    text = sam_header(ref_names, ref_len, hd).encode()
    out = bytearray()
    out += b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(ref_names))
    for n, l in zip(ref_names, ref_len):
        nb = n.encode() + b"\x00"
        out += struct.pack("<i", len(nb)) + nb + struct.pack("<i", int(l))
    out += bam_record_bytes(records, read_len, irregular_seed, l_seq_of)
    with open(path, "wb") as f:
        for s in range(0, len(out), 0xff00):
            f.write(_bgzf_block(bytes(out[s:s + 0xff00])))
        f.write(_bgzf_block(b""))  # EOF marker


def write_sldb(path, taxonomy):
    """cereal binary layout of slimm_database."""
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(taxonomy.accessions)))
        for acc, lin in zip(taxonomy.accessions, taxonomy.lineage):
            a = acc.encode()
            f.write(struct.pack("<Q", len(a)) + a + struct.pack("<Q", 8) + np.asarray(lin, dtype="<u4").tobytes())
        f.write(struct.pack("<Q", len(taxonomy.tax_name)))
        for t, r, n in zip(taxonomy.tax_id, taxonomy.tax_rank, taxonomy.tax_name):
            b = n.encode()
            f.write(struct.pack("<IIQ", int(t), int(r), len(b)) + b)


def read_sldb(path):
    """Inverse of write_sldb: ({accession: [taxids]}, {taxid: (rank, name)}), entries in file order."""
    with open(path, "rb") as f:
        b = f.read()
    o = 0

    def u64():
        nonlocal o
        v = struct.unpack_from("<Q", b, o)[0]
        o += 8
        return v

    ac, tn = {}, {}
    for _ in range(u64()):
        n = u64()
        a = b[o:o + n].decode()
        o += n
        c = u64()
        ac[a] = list(struct.unpack_from("<%dI" % c, b, o))
        o += 4 * c
    for _ in range(u64()):
        t, r = struct.unpack_from("<II", b, o)
        o += 8
        n = u64()
        tn[t] = (r, b[o:o + n].decode())
        o += n
    assert o == len(b)
    return ac, tn
