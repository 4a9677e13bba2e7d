#!/bin/bash
# bucket scatter: rounds ordered by tile in LDS (k_tile_scatter_big) against the direct rounds, per-kernel lines at config 4
# (1 B records) and at the configs named in CONFIGS (default: config3 config5)
for big in 1 0; do
  export SLIMM_FORCE=scatter_big=$big
  echo "== SLIMM_FORCE=scatter_big=$big"
  python bench.py --quick --breakdown --steps 5 --warmup 2 2>&1 >/dev/null | grep -E "^# (k_tile|device)"
  for c in ${CONFIGS:-config3 config5}; do python bench.py --config $c --quick --breakdown --steps 10 2>&1 >/dev/null | grep -E "^# (k_tile|device)"; done
done
