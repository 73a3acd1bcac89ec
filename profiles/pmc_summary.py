"""Per-kernel summary of two rocprofv3 --pmc passes (SQ_* activity counters; LDS conflict counters) dumped by pmc_dump.py.
usage: python profiles/pmc_summary.py <sq_dump.txt> <lds_dump.txt> > table.md
MFMA % = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 8 x SQ_BUSY_CYCLES) -- the normalisation that reproduces bench.py's roofline fraction
for the forward GEMM; wait % / issue % = SQ_WAIT_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES; LDS conflict % =
SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS (cycles the LDS pipe spent on conflicts per cycle it was active)."""
import collections, re, sys


def load(fn):
    d = collections.defaultdict(dict)
    for line in open(fn):
        m = re.match(r"(\S+)\s+([\d.]+)\s+\(n=(\d+)\)\s+(.*)", line)
        if m:
            d[m.group(4).strip()][m.group(1)] = (float(m.group(2)), int(m.group(3)))
    return d


sq, lds = load(sys.argv[1]), load(sys.argv[2])
rows = []
for k, v in sq.items():
    if "SQ_BUSY_CYCLES" not in v:
        continue
    busy, n = v["SQ_BUSY_CYCLES"]
    wc = v["SQ_WAVE_CYCLES"][0]
    lc = lds.get(k, {})
    rows.append((busy * n, k, n, busy, v.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[0], wc, v["SQ_WAIT_ANY"][0], v["SQ_ACTIVE_INST_ANY"][0],
                 lc.get("SQ_LDS_BANK_CONFLICT", (0, 0))[0], lc.get("SQ_ACTIVE_INST_LDS", (0, 0))[0]))
print("| kernel | samples | busy cycles | MFMA % | wait % | issue % | LDS conflict % |\n|---|---|---|---|---|---|---|")
for tot, k, n, busy, mf, wc, wait, act, conf, lact in sorted(rows, reverse=True):
    print(f"| `{k[:90]}` | {n} | {busy:.0f} | {100 * mf / max(1, 32 * busy):.1f} | {100 * wait / max(1, wc):.1f} | {100 * act / max(1, wc):.1f} | "
          f"{100 * conf / max(1, lact):.1f} |")
