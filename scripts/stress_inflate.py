"""Random DEFLATE streams through the device inflate (slimm_bgzf_inflate_with: the two-phase kernels, and the lane-per-block
kernel) against zlib: every payload kind x level x strategy, blocks with several DEFLATE blocks, and CORRUPT blocks (bit flips
in the payload, the CRC, the ISIZE) -- the device must give zlib's bytes whenever zlib accepts a stream with a matching
trailer, and an error otherwise; never bytes of its own.
    python scripts/stress_inflate.py [seeds] [first seed]        (GPU box, or SLIMM_EMU=1 for the host emulator)"""
import ctypes as C, os, struct, sys, zlib
sys.path.insert(0, ".")
import numpy as np
from slimm_amd import capi
if os.environ.get("SLIMM_EMU") == "1":
    capi.LIB_PATH = os.path.join("tests", "native", "libslimm_emu.so"); capi._lib = None
L = capi.lib()


def wrap(data, raw, crc=None, isize=None):
    bsize = 12 + 6 + len(raw) + 8
    assert bsize <= 65536
    head = b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, (bsize - 1) & 0xffff)
    return head + raw + struct.pack("<II", (zlib.crc32(data) & 0xffffffff) if crc is None else crc, len(data) if isize is None else isize)


def inflate(blob, how):
    src = np.frombuffer(blob, dtype=np.uint8)
    out = np.zeros(1 << 24, dtype=np.uint8)
    n, ms, err, lanes = C.c_uint64(), C.c_double(), C.create_string_buffer(256), C.c_uint32()
    rc = L.slimm_bgzf_inflate_with(0, src.ctypes.data_as(C.c_void_p), C.c_uint64(len(blob)), out.ctypes.data_as(C.c_void_p), C.c_uint64(out.size),
                                   C.byref(n), C.byref(ms), err, C.c_uint64(256), how, C.byref(lanes))
    return rc, bytes(out[:n.value]) if rc == 0 else b"", lanes.value


def payload(rng):
    kind = int(rng.integers(0, 9))
    n = int(rng.choice([1, 2, 7, 100, 3000, 20000, 65000, 65280]))
    if kind == 0: return bytes(rng.integers(0, 256, n, dtype=np.uint8))
    if kind == 1: return bytes(rng.integers(0, int(rng.integers(2, 9)), n, dtype=np.uint8))
    if kind == 2: return bytes(n)
    if kind == 3: return (bytes(rng.integers(97, 123, int(rng.integers(1, 40)), dtype=np.uint8)) * (n // 3 + 1))[:n]
    if kind == 4:
        rec = b"".join(struct.pack("<iiiBBHHHIiii", 200 + k % 9, k % 300, 913 * k, 9, 30, 4680, 1, 0, 100, -1, -1, 0) + b"read%06d\0" % (k // 4)
                       + bytes(rng.integers(0, 256, 50, dtype=np.uint8) & 0x33) + bytes(rng.choice(np.array([2, 12, 23, 37], dtype=np.uint8), 100)) for k in range(n // 200 + 1))
        return rec[:n]
    if kind == 5: return b"".join(bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 400)) for _ in range(n // 100 + 1))[:n]
    if kind == 6: return (b"ACGT" * (n // 4 + 1))[:n]
    if kind == 7: return bytes(rng.integers(0, 256, n // 2, dtype=np.uint8)) + bytes(n - n // 2)
    a = bytes(rng.integers(0, 256, min(n, 300), dtype=np.uint8))
    return (a * (n // len(a) + 1))[:n]


n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
first = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
bad = 0
for seed in range(first, first + n_seeds):
    rng = np.random.default_rng(seed)
    blocks, want = [], []
    for _ in range(int(rng.integers(1, 90))):
        data = payload(rng)
        level = int(rng.choice([0, 1, 3, 6, 9]))
        strat = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
        c = zlib.compressobj(level, zlib.DEFLATED, -15, int(rng.choice([1, 8, 9])), strat)
        if rng.random() < 0.3 and len(data) > 10:   # several DEFLATE blocks: flushes in the middle
            cut = int(rng.integers(1, len(data)))
            raw = c.compress(data[:cut]) + c.flush(int(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_BLOCK]))) + c.compress(data[cut:]) + c.flush()
        else:
            raw = c.compress(data) + c.flush()
        if 12 + 6 + len(raw) + 8 > 65536:
            continue
        blocks.append(wrap(data, raw))
        want.append(data)
    blob, plain = b"".join(blocks), b"".join(want)
    for how in (0, 1):
        rc, got, lanes = inflate(blob, how)
        if rc != 0 or got != plain:
            print(f"seed {seed} how {how}: VALID input: rc {rc}, equal {got == plain}")
            bad += 1
    # corrupt blocks, one at a time between two good ones
    good = blocks[0] if blocks else wrap(b"abc", zlib.compress(b"abc")[2:-4])
    for _ in range(25):
        k = int(rng.integers(0, len(blocks)))
        blk = bytearray(blocks[k])
        where = int(rng.integers(18, len(blk)))
        blk[where] ^= 1 << int(rng.integers(0, 8))
        raw = bytes(blk[18:-8])
        crc, isize = struct.unpack("<II", bytes(blk[-8:]))
        try:
            d = zlib.decompressobj(-15)
            ref = d.decompress(raw)
            ok = d.eof and (zlib.crc32(ref) & 0xffffffff) == crc and len(ref) == isize and isize <= 65536
        except zlib.error:
            ok, ref = False, b""
        for how in (0, 1):
            rc, got, lanes = inflate(good + bytes(blk) + good, how)
            if ok:
                if rc != 0 or got != want[0] + ref + want[0]:
                    print(f"seed {seed} how {how}: zlib accepts the flipped block {k} (byte {where}) but the device says rc {rc}")
                    bad += 1
            elif rc == 0:
                print(f"seed {seed} how {how}: the device ACCEPTED a corrupt block {k} (byte {where} of {len(blk)})")
                bad += 1
print(f"{n_seeds} seeds from {first}: {bad} failures")
sys.exit(1 if bad else 0)
