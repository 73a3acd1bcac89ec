#!/bin/bash
# rocprofv3 kernel trace of one command on the GPU box, summarised into profiles/<name>.md (per kernel) and
# profiles/<name>_by_shape.md (per kernel x launch grid).   usage: profiles/profile.sh <name> <command...>
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; shift
OUT="$ROOT/gpurun_out/prof_$NAME"
mkdir -p "$OUT"
export TMPDIR=/tmp
( cd "$ROOT" && rocprofv3 --kernel-trace --stats -d "$OUT" -o t -- "$@" ) > "$OUT/run.log" 2>&1 || { tail -20 "$OUT/run.log"; exit 1; }
DB=$(find "$OUT" -name "*_results.db" | head -1)
python "$ROOT/profiles/summarize_rocpd.py" "$DB" "$ROOT/gpurun_out/${NAME}.md" --in-step "$ROOT/gpurun_out/${NAME}_in_step.json" > /dev/null
python "$ROOT/profiles/summarize_rocpd.py" "$DB" "$ROOT/gpurun_out/${NAME}_by_shape.md" --by-grid > /dev/null
grep -v "^W2026\|^I2026" "$OUT/run.log" | tail -5
rm -rf "$OUT"/*.db 2>/dev/null || true
