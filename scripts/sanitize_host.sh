#!/bin/bash
# AddressSanitizer + UBSan over the host code that does not need a GPU: the SAM/BAM/.sldb readers (good, truncated and
# corrupted inputs, 1 and 8 decode threads) and the host profile stages.  GPU sanitizers are not available on the pool,
# so this is the CPU-build sanitizer pass.  Usage: scripts/sanitize_host.sh [workdir]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
W="${1:-/tmp/slimm_sanitize}"
mkdir -p "$W"
FLAGS="-std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer"
g++ $FLAGS "$ROOT/tests/native/san_readers.cpp" "$ROOT/slimm_amd/csrc/host/alignment_file.cpp" \
    "$ROOT/slimm_amd/csrc/host/sldb.cpp" -lz -lpthread -ldl -o "$W/san_readers"
g++ $FLAGS "$ROOT/tests/native/host_profile_bench.cpp" "$ROOT/slimm_amd/csrc/host_profile.cpp" -o "$W/san_profile"
g++ $FLAGS "$ROOT/slimm_amd/csrc/host/slimm_build_main.cpp" "$ROOT/slimm_amd/csrc/host/sldb.cpp" -lz -o "$W/san_build"
# ThreadSanitizer over the parallel BGZF inflate / record decode
g++ -std=c++17 -g -O1 -fsanitize=thread "$ROOT/tests/native/san_readers.cpp" "$ROOT/slimm_amd/csrc/host/alignment_file.cpp" \
    "$ROOT/slimm_amd/csrc/host/sldb.cpp" -lz -lpthread -ldl -o "$W/tsan_readers"
g++ -std=c++17 -g -O1 -fsanitize=thread -I"$ROOT/include" "$ROOT/slimm_amd/csrc/host/slimm_main.cpp" "$ROOT/slimm_amd/csrc/host/alignment_file.cpp" \
    "$ROOT/slimm_amd/csrc/host/sldb.cpp" -lz -lpthread -ldl -o "$W/tsan_slimm"
cd "$ROOT"
python - "$W" <<'PY'
import os, subprocess, sys
sys.path.insert(0, os.getcwd())
from slimm_amd.synth import CONFIGS, make_workload
from tests.bam_io import write_bam, write_sam, write_sldb
from tests.cases import tiny_case, holes_case
d = sys.argv[1]
files = []
cases = [tiny_case(), holes_case(), make_workload(CONFIGS["config1"], seed=3),
         make_workload(CONFIGS["config2"], seed=4, n_records=150_000)]
for i, w in enumerate(cases):
    for fmt, wr in (("sam", write_sam), ("bam", write_bam)):
        p = f"{d}/c{i}.{fmt}"
        wr(p, w.ref_names, w.ref_len, w.records, read_len=w.avg_read_len)
        files.append(p)
    p = f"{d}/c{i}.sldb"
    write_sldb(p, w.taxonomy)
    files.append(p)
# (a BAM that hardly compresses: more than the reader's own first windows hold, so that read_raw reaches the part of the file it
# reads in place and read_blocks takes over)
write_bam(f"{d}/c4.bam", cases[3].ref_names, cases[3].ref_len, cases[3].records, read_len=cases[3].avg_read_len, irregular_seed=5)
files.append(f"{d}/c4.bam")
data = open(f"{d}/c2.bam", "rb").read()
open(f"{d}/trunc.bam", "wb").write(data[:len(data) // 3])
bad = bytearray(data); bad[200] ^= 0xff
open(f"{d}/flip.bam", "wb").write(bytes(bad))
open(f"{d}/short.sldb", "wb").write(open(f"{d}/c2.sldb", "rb").read()[:100])
open(f"{d}/garbage.sam", "w").write("@SQ\tSN:x\tLN:10\nr1\t0\tx\n")
files += [f"{d}/trunc.bam", f"{d}/flip.bam", f"{d}/short.sldb", f"{d}/garbage.sam"]
bad = 0
for thr in ("1", "8"):
    r = subprocess.run([f"{d}/san_readers"] + files, capture_output=True, text=True,
                       env=dict(os.environ, SAN_READER_THREADS=thr))
    print(r.stdout)
    if r.returncode or r.stderr.strip():
        bad += 1; print("SANITIZER OUTPUT:\n" + r.stderr)
r = subprocess.run([f"{d}/tsan_readers", f"{d}/c3.bam", f"{d}/c4.bam", f"{d}/c2.bam", f"{d}/trunc.bam"], capture_output=True, text=True,
                   env=dict(os.environ, SAN_READER_THREADS="8"))
print(r.stdout)
if r.returncode or r.stderr.strip():
    bad += 1; print("THREAD SANITIZER OUTPUT:\n" + r.stderr)
r = subprocess.run([f"{d}/san_profile"], capture_output=True, text=True)
print(r.stdout)
if r.returncode or r.stderr.strip():
    bad += 1; print("SANITIZER OUTPUT:\n" + r.stderr)
# slimm_build: well-formed dumps (plain and gzip'ed FASTA, tiny batches), then cut and garbled ones
import pathlib
from slimm_amd.synth import synth_taxonomy
from tests.ncbi_dumps import write_dumps
for k, (gz, batch) in enumerate([(False, "1000000"), (True, "7")]):
    t = pathlib.Path(d) / f"dumps{k}"
    t.mkdir(exist_ok=True)
    dd = write_dumps(t, synth_taxonomy(400, strain_level=bool(k), hole_every=3), gz_fasta=gz)
    runs = [dd]
    if k == 0:
        cut = dict(dd)
        for key in ("nodes", "names"):
            b = open(dd[key], "rb").read()
            open(dd[key] + ".cut", "wb").write(b[:len(b) // 2 + 3])
            cut[key] = dd[key] + ".cut"
        open(f"{t}/junk.a2t", "wb").write(b"\t\t\t\n\n\xff\xfe\t1\nACC000001\nACC000002\t\t99999999999999999999\n" + os.urandom(4096))
        cut["acc"] = [f"{t}/junk.a2t"] + dd["acc"]
        open(f"{t}/junk.fa", "wb").write(b">\n>>|\n@x\n" + os.urandom(2048) + b"\n>ACC000003.1 z")
        runs += [cut, dict(cut, fasta=f"{t}/junk.fa")]
    for rr in runs:
        r = subprocess.run([f"{d}/san_build", "-nm", rr["names"], "-nd", rr["nodes"], "-o", f"{t}/o.sldb", "-b", batch, rr["fasta"]]
                           + rr["acc"], capture_output=True, text=True, errors="replace")
        noise = [l for l in r.stderr.splitlines() if not l.startswith(("[MSG]", "[WARNING!]", "[VERBOSE"))]
        if r.returncode or noise:
            bad += 1; print("SANITIZER OUTPUT (slimm_build):\n" + r.stderr)
print("slimm_build under sanitizers: done")
# The `slimm` command itself under ThreadSanitizer -- its reader thread, the inflater and the pusher of the raw windows
# (RecordPump), the fallback to the host decoder -- with the HIP library replaced by the host emulator of tests/native (the
# command dlopen()s what SLIMM_HIP_LIB names).  Skipped when that library has not been built (make -C tests/native).
emu = os.path.join(os.getcwd(), "tests", "native", "libslimm_emu.so")
if os.path.exists(emu) and os.path.exists(f"{d}/tsan_slimm"):
    for extra in (["--window-mb", "1"], ["--window-mb", "3", "--no-mmap"], ["--host-decode"]):
        for inp in ("c3.bam", "c2.sam"):
            r = subprocess.run([f"{d}/tsan_slimm"] + extra + ["-w", "1000", "-o", f"{d}/cli_", f"{d}/{inp[:2]}.sldb", f"{d}/{inp}"], capture_output=True,
                               text=True, errors="replace", env=dict(os.environ, SLIMM_HIP_LIB=emu))
            if r.returncode or "ThreadSanitizer" in r.stderr:
                bad += 1; print("SANITIZER OUTPUT (slimm under TSan):\n" + r.stderr[-4000:])
    print("slimm (command, emulated device) under ThreadSanitizer: done")
print("sanitizer findings:", bad)
sys.exit(1 if bad else 0)
PY
