#!/bin/bash
# Round-6 call 8: timeline of the C2 step (idle time, overlap, gaps) from a rocprofv3 kernel trace; decode-step skinny GEMM with an XCD-aware unit order (lab patch, not kept: no change in FETCH_SIZE -- the column-tile counts are multiples of 8, so the row tiles of a column tile already share an XCD).
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
{
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
OUT=gpurun_out/prof_tl; rm -rf $OUT; mkdir -p $OUT
( rocprofv3 --kernel-trace -d $OUT -o t -- python bench.py --legs none --no-cpu --no-gen --no-pmc --steps 12 --warmup 4 ) > $OUT/run.log 2>&1 || tail -5 $OUT/run.log
DB=$(find $OUT -name "*_results.db" | head -1)
python profiles/timeline_rocpd.py "$DB" gpurun_out/r6_timeline.txt
rm -rf $OUT/*.db
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "generate or skinny or decode" 2>&1 | grep -v "^W2026\|^E2026" | tail -4
for x in 1 0; do
  P5_DEC_XCD=$x timeout 300 python bench.py --legs none --no-cpu --steps 5 --warmup 2 2>/dev/null | grep '^{' | python -c "
import sys, json
l = json.loads(sys.stdin.read()); g = l['roofline_generation']
print('dec_xcd=$x', 'ms/step', l['ms_per_step'], 'gen items/s', l['beam10_items_per_sec'], 'decode step ms', g['ms_per_step'], 'traffic MB', round((g.get('traffic') or 0) / 1e6, 1), 'stale', g.get('traffic_stale'), 'plain', l['generation_plain_bf16'].get('items_per_s'))
"
done
} 2>&1 | tee gpurun_out/r6_call8.txt
