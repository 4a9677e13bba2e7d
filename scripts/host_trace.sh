#!/bin/bash
# The host's share of a step, piece by piece (SLIMM_TRACE=host: "[host] <call>: <piece> <us>" lines of the library), for the
# headline stream and config 2: the last step's lines.   scripts/host_trace.sh TAG
TAG=${1:-host_trace}; O=gpurun_out/$TAG; mkdir -p $O
for c in config4 config2; do
  SLIMM_TRACE=host python3 bench.py --quick --engines 1 --config $c --steps 3 --warmup 1 --no-any-order > $O/$c.json 2> $O/$c.err
  grep "^\[host\]" $O/$c.err | awk '/analyze_alignments: set device/ {buf=""} {buf=buf $0 "\n"} END {printf "%s", buf}' > $O/host_trace_$c.txt
  echo "== $c"; cat $O/host_trace_$c.txt
done
