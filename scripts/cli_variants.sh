#!/bin/bash
# `slimm DB realistic.bam` (100 M records) with library variants (scripts/build_variant.sh): scripts/cli_variants.sh NAME ...
cd "$GRAFT_REPO_ROOT"
python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from slimm_amd.synth import CONFIGS, make_workload
from slimm_amd.synth_bam import write_synthetic_bam
from tests.bam_io import write_sldb
w = make_workload(CONFIGS["config3"], seed=1, n_records=100_000_000)
os.makedirs("/tmp/slimm_var/out", exist_ok=True)
write_sldb("/tmp/slimm_var/db.sldb", w.taxonomy)
write_synthetic_bam("/tmp/slimm_var/realistic.bam", w.ref_names, w.ref_len, w.records, read_len=100, realistic=True)
PY
for rep in 1 2 3 4; do
for v in "$@"; do
  lib=$PWD/slimm_amd/libslimm_hip.so; [ "$v" != base ] && lib=$PWD/build/var/$v/libslimm_hip.so
  s=$(date +%s%N)
  SLIMM_HIP_LIB=$lib SLIMM_TRACE=cli ./slimm_amd/slimm -w 1000 -o /tmp/slimm_var/out/ /tmp/slimm_var/db.sldb /tmp/slimm_var/realistic.bam 2> /tmp/slimm_var/err.txt > /dev/null
  e=$(date +%s%N)
  echo "$v: $(( (e - s) / 1000000 )) ms; $(grep -o 'slimm_push_bam_bytes [0-9.]* ms' /tmp/slimm_var/err.txt | head -1); $(grep -o 'slimm_create *[0-9.]* ms' /tmp/slimm_var/err.txt)"
done; done
rm -rf /tmp/slimm_var
