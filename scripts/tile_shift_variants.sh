#!/bin/bash
# bench --breakdown lines of the tile kernels with the tile size the layout picks and with either one forced:
#   scripts/tile_shift_variants.sh [shift ...]   (default: auto 13 14)
for v in ${@:-auto 13 14}; do
  if [ $v = auto ]; then unset SLIMM_FORCE; else export SLIMM_FORCE=tile_shift=$v; fi
  echo "== tile shift $v"
  python bench.py --quick --breakdown --steps 5 --warmup 2 2>&1 >/dev/null | grep -E "^# (k_tile|device)"
  for c in config2 config3 config5; do python bench.py --config $c --quick --breakdown --steps 10 2>&1 >/dev/null | grep -E "^# (k_tile|device)"; done
done
