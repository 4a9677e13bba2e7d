#!/bin/bash
# rocprofv3 kernel trace of `slimm DB realistic.bam` (100 M records unless $1 says otherwise) -> gpurun_out/prof_cli/
set -e
N=${1:-100000000}
OUT=${2:-gpurun_out/prof_cli}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python3 - "$N" <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from slimm_amd.synth import CONFIGS, make_workload
from slimm_amd.synth_bam import write_synthetic_bam
from tests.bam_io import write_sldb
n = int(sys.argv[1])
w = make_workload(CONFIGS["config3"], seed=1, n_records=n)
os.makedirs("/tmp/slimm_prof/out", exist_ok=True)
write_sldb("/tmp/slimm_prof/db.sldb", w.taxonomy)
print(write_synthetic_bam("/tmp/slimm_prof/realistic.bam", w.ref_names, w.ref_len, w.records, read_len=100, realistic=True))
PY
./slimm_amd/slimm -w 1000 -o /tmp/slimm_prof/out/ /tmp/slimm_prof/db.sldb /tmp/slimm_prof/realistic.bam > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d "$OUT" -o cli -- ./slimm_amd/slimm -w 1000 -o /tmp/slimm_prof/out/ /tmp/slimm_prof/db.sldb /tmp/slimm_prof/realistic.bam > "$OUT.log" 2>&1 || true
SLIMM_TRACE=cli ./slimm_amd/slimm -w 1000 -o /tmp/slimm_prof/out/ /tmp/slimm_prof/db.sldb /tmp/slimm_prof/realistic.bam 2>&1 | grep trace | cut -c1-300
rm -rf /tmp/slimm_prof
