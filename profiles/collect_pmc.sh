#!/bin/bash
# HBM traffic of bench.py's roofline kernels from the PMC counters, collected exactly as MI355X_MICROARCH.md (HBM / rocprofv3
# sections) prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one pass), kernel trace only (no
# sys/hip/hsa trace next to --pmc), FETCH_SIZE doubled on gfx950 (it tallies 128-byte requests of wide coalesced reads at
# 64 bytes).  Writes profiles/pmc_traffic.json (read by bench.py -> roofline.traffic) and gpurun_out/pmc_raw.txt.
#   usage (GPU box, repo root):  bash profiles/collect_pmc.sh
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
export TMPDIR=/tmp
OUT="$ROOT/gpurun_out/pmc"
rm -rf "$OUT"; mkdir -p "$OUT"
: > "$ROOT/gpurun_out/pmc_raw.txt"
# key (bench.py: pmc_traffic(name, shape)) | shape | kernel-name pattern | command
CASES=(
  "fwd_wide|8192x2048x512|p5_gemm5_kernelILb0E|python tools/gemm_one.py 8192 2048 512"
  "wgrad_group2|8192x512x2048|p5_gemm5_kernelILb1E|python tools/gemm_group_one.py"
)
for c in "${CASES[@]}"; do
  IFS='|' read -r NAME SHAPE PAT CMD <<< "$c"
  for CTR in FETCH_SIZE WRITE_SIZE; do
    D="$OUT/${CTR}_$(echo "$SHAPE" | tr 'x' '_')"
    rocprofv3 --pmc $CTR --kernel-trace -d "$D" -o g -- $CMD > "$D.log" 2>&1 || { tail -5 "$D.log"; exit 1; }
    DB=$(find "$D" -name "*_results.db" | head -1)
    echo "## $NAME $SHAPE $CTR" >> "$ROOT/gpurun_out/pmc_raw.txt"
    python profiles/pmc_dump.py "$DB" "$PAT" >> "$ROOT/gpurun_out/pmc_raw.txt"
  done
done
python - "$ROOT" <<'PY'
import json, re, sys
root = sys.argv[1]
tab, cur = {}, None
for line in open(f"{root}/gpurun_out/pmc_raw.txt"):
    m = re.match(r"## (\S+) (\S+) (\S+)", line)
    if m:
        cur = (m.group(1), m.group(2), m.group(3))
        continue
    f = line.split()
    if cur and len(f) >= 2 and f[0] in ("FETCH_SIZE", "WRITE_SIZE"):
        ent = tab.setdefault(f"{cur[0]}|{cur[1]}", {})
        ent[f[0] + "_raw_kb"] = float(f[1])
for k, e in tab.items():
    if "FETCH_SIZE_raw_kb" in e and "WRITE_SIZE_raw_kb" in e:
        e["read_bytes"] = e["FETCH_SIZE_raw_kb"] * 1024.0 * 2.0      # gfx950 correction (MI355X_MICROARCH.md, HBM section)
        e["write_bytes"] = e["WRITE_SIZE_raw_kb"] * 1024.0
        e["traffic_bytes"] = e["read_bytes"] + e["write_bytes"]
        e["note"] = "per launch; FETCH_SIZE x2 (gfx950), separate --pmc passes, averaged over the launches of tools/gemm_one.py / tools/gemm_group_one.py"
json.dump(tab, open(f"{root}/profiles/pmc_traffic.json", "w"), indent=1, sort_keys=True)
json.dump(tab, open(f"{root}/gpurun_out/pmc_traffic.json", "w"), indent=1, sort_keys=True)   # gpurun merges gpurun_out/ back
print(json.dumps(tab, indent=1, sort_keys=True))
PY
