#!/bin/bash
# SLIMM_TRACE=push of `slimm DB realistic.bam`: the window pipeline's events with their times
set -e
N=${1:-100000000}
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
write_synthetic_bam("/tmp/slimm_prof/realistic.bam", w.ref_names, w.ref_len, w.records, read_len=100, realistic=True)
PY
./slimm_amd/slimm -w 1000 -o /tmp/slimm_prof/out/ /tmp/slimm_prof/db.sldb /tmp/slimm_prof/realistic.bam > /dev/null 2>&1
SLIMM_TRACE=cli,push ./slimm_amd/slimm -w 1000 -o /tmp/slimm_prof/out/ /tmp/slimm_prof/db.sldb /tmp/slimm_prof/realistic.bam 2>&1 | grep -E "\[push|trace\] device|reached" | cut -c1-200
rm -rf /tmp/slimm_prof
