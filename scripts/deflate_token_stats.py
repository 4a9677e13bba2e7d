"""What the resolve phase of the device inflate (bgzf_tokens.hip) has to work with: literals, matches, match lengths and
distances of the BGZF blocks of a synthetic BAM, counted by a small pure-Python DEFLATE reader (a few blocks take seconds).
    python scripts/deflate_token_stats.py [records] [easy|realistic] [blocks]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from slimm_amd.synth import CONFIGS, make_workload
from slimm_amd.synth_bam import write_synthetic_bam

LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEXT = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DBASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
DEXT = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]


class Bits:
    def __init__(self, data):
        self.d, self.pos = data, 0
    def get(self, n):
        v = 0
        for k in range(n):
            v |= ((self.d[self.pos >> 3] >> (self.pos & 7)) & 1) << k
            self.pos += 1
        return v


def build(lengths):
    count = [0] * 16
    for l in lengths:
        count[l] += 1
    count[0] = 0
    offs, sym = [0] * 16, [0] * len(lengths)
    for l in range(1, 16):
        offs[l] = offs[l - 1] + count[l - 1]
    for s, l in enumerate(lengths):
        if l:
            sym[offs[l]] = s
            offs[l] += 1
    return count, sym


def decode(br, h):
    count, sym = h
    code = first = index = 0
    for l in range(1, 16):
        code |= br.get(1)
        c = count[l]
        if code - c < first:
            return sym[index + (code - first)]
        index += c
        first = (first + c) << 1
        code <<= 1
    raise ValueError("bad code")


def tokens(payload):
    """[(literal run, match length, distance)] of one raw DEFLATE stream"""
    br, out, run = Bits(payload), [], 0
    global N_DEFLATE_BLOCKS
    N_DEFLATE_BLOCKS = 0
    while True:
        last, typ = br.get(1), br.get(2)
        N_DEFLATE_BLOCKS += 1
        if typ == 0:
            br.pos = (br.pos + 7) & ~7
            n = br.get(16)
            br.get(16)
            run += n
            br.pos += 8 * n
        else:
            if typ == 1:
                lh = build([8] * 144 + [9] * 112 + [7] * 24 + [8] * 8)
                dh = build([5] * 30)
            else:
                nl, nd, nc = br.get(5) + 257, br.get(5) + 1, br.get(4) + 4
                cl = [0] * 19
                for k in range(nc):
                    cl[[16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15][k]] = br.get(3)
                ch, ls = build(cl), []
                while len(ls) < nl + nd:
                    s = decode(br, ch)
                    if s < 16:
                        ls.append(s)
                    elif s == 16:
                        ls += [ls[-1]] * (3 + br.get(2))
                    elif s == 17:
                        ls += [0] * (3 + br.get(3))
                    else:
                        ls += [0] * (11 + br.get(7))
                lh, dh = build(ls[:nl]), build(ls[nl:])
            while True:
                s = decode(br, lh)
                if s < 256:
                    run += 1
                elif s == 256:
                    break
                else:
                    ln = LBASE[s - 257] + br.get(LEXT[s - 257])
                    ds = decode(br, dh)
                    out.append((run, ln, DBASE[ds] + br.get(DEXT[ds])))
                    run = 0
        if last:
            break
    if run:
        out.append((run, 0, 0))
    return out


n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
kind = sys.argv[2] if len(sys.argv) > 2 else "realistic"
n_blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 6
w = make_workload(CONFIGS["config3"], seed=1, n_records=n)
tmp = tempfile.mkdtemp(prefix="slimm_tok_")
bam = os.path.join(tmp, kind + ".bam")
info = write_synthetic_bam(bam, w.ref_names, w.ref_len, w.records, read_len=100, realistic=(kind == "realistic"))
blob = np.fromfile(bam, dtype=np.uint8).tobytes()
os.unlink(bam)
print(f"{kind}: {info['raw_bytes'] / info['compressed_bytes']:.2f} x ({info['deflate']})")
p, k = 0, 0
while p + 18 <= len(blob) and k < n_blocks + 2:
    bs = blob[p + 16] + (blob[p + 17] << 8) + 1
    if k >= 2:   # (the first blocks hold the header)
        isize = int.from_bytes(blob[p + bs - 4:p + bs], "little")
        t = tokens(blob[p + 18:p + bs - 8])
        lit = sum(x[0] for x in t)
        m = [x for x in t if x[1]]
        ml = np.array([x[1] for x in m])
        md = np.array([x[2] for x in m])
        runs = np.array([x[0] for x in t])
        over = int(np.sum(md < ml))
        # chunks as k_inflate_resolve cuts them: tokens whose spans end inside 4096 bytes, at most 256 (512) tokens
        def chunks(max_tok):
            c, i = 0, 0
            spans = [x[0] + x[1] for x in t]
            while i < len(spans):
                s, j = 0, i
                while j < len(spans) and j - i < max_tok and s + spans[j] <= 4096:
                    s += spans[j]
                    j += 1
                i = max(j, i + 1)
                c += 1
            return c
        print(f"block {k}: {isize} B, {lit} literals ({100 * lit / isize:.0f} %), {len(m)} matches, mean length {ml.mean():.1f} (max {ml.max()}), "
              f"median distance {int(np.median(md))}, {over} overlap themselves, longest literal run {runs.max()}, "
              f"{N_DEFLATE_BLOCKS} DEFLATE blocks, tokens per 4 KB {len(t) * 4096 / isize:.0f}, chunks with 256 tokens {chunks(256)}, with 512 {chunks(512)}, with 1024 {chunks(1024)}")
    p += bs
    k += 1
