#!/bin/bash
# sweep_tile_cells.sh OUTFILE K "TC1 TC2 ..." [bench flags]: ms/step of the tiled launch against the tile size
# (option "tile_cells"); one line per run: tile_cells ms_per_step ntiles lane_slots
OUT=$1; K=$2; TCS=$3; shift 3
for tc in $TCS; do
  python3 bench.py --no-cpu-baseline --k $K --tile-cells $tc "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
t = d.get('config', {}).get('tiling') or d.get('tiling') or {}
print($tc, round(d['ms_per_step'], 5), t.get('ntiles'), t.get('cells_per_tile'), t.get('lane_slots'))" >> $OUT || echo "$tc failed" >> $OUT
done
