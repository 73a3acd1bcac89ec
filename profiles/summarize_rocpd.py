"""Turn a rocprofv3 (rocpd sqlite) kernel trace into a per-kernel stats table (what `--stats` prints as CSV).
usage: python profiles/summarize_rocpd.py gpurun_out/prof/x_results.db [out.md] [--by-grid] [--in-step out.json]
--by-grid keys the table by (kernel, grid) so that every shape a kernel runs at gets its own row (per-shape durations).
--in-step writes what bench.py reports as `roofline.in_step_us`: average durations of the roofline kernels INSIDE the training steps
(launches up to the last optimizer update; the stand-alone roofline timing launches of bench.py come after it and are excluded),
plus launches and kernel time per step."""
import json
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    suf = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace("rocpd_kernel_dispatch", "")
    rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x, s.arch_vgpr_count, s.accum_vgpr_count, d.group_segment_size "
                       f"from rocpd_kernel_dispatch{suf} d join rocpd_info_kernel_symbol{suf} s on d.kernel_id = s.id order by d.start").fetchall()
    stats = {}
    for name, st, en, gx, gy, gz, wx, vg, ag, lds in rows:
        k = short(name)
        if "--by-grid" in sys.argv:
            k = f"{k} grid=({gx // max(1, wx)},{gy},{gz})"
        a = stats.setdefault(k, [0, 0, 1 << 62, 0, vg, ag, lds])
        dur = en - st
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    total = sum(a[1] for a in stats.values())
    span = rows[-1][2] - rows[0][1] if rows else 0
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | lds |", "|---|---|---|---|---|---|---|---|---|---|"]
    for k, a in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| {k} | {a[0]} | {a[1]/1e6:.3f} | {a[1]/a[0]/1e3:.2f} | {a[2]/1e3:.2f} | {a[3]/1e3:.2f} | {100*a[1]/total:.1f} | {a[4]} | {a[5]} | {a[6]} |")
    lines.append("")
    lines.append(f"dispatches: {len(rows)}; sum of kernel time {total/1e6:.3f} ms; first-start to last-end {span/1e6:.3f} ms")
    if "--in-step" in sys.argv:
        path = sys.argv[sys.argv.index("--in-step") + 1]
        adam = [i for i, r in enumerate(rows) if "p5_adamw" in r[0] and "segments" not in r[0]]      # (flat kernel or the tile-wise one: one per optimizer step)
        res = {}
        if adam:
            inside = rows[:adam[-1] + 1]
            steps = len(adam)

            def avg(pred):
                d = [(r[2] - r[1]) / 1e3 for r in inside if pred(r)]
                return (sum(d) / len(d), len(d)) if d else (None, 0)
            a, n = avg(lambda r: "p5_gemm5_kernelILb1" in r[0] and r[3] // max(1, r[6]) == 192)
            res["wgrad_group2"], res["wgrad_group2_launches_per_step"] = a, n / steps
            a, n = avg(lambda r: "p5_gemm5_kernelILb0" in r[0])
            res["fwd_wide"], res["fwd_wide_launches_per_step"] = a, n / steps
            res["fwd_wide_note"] = "all K-contiguous launches of p5_gemm5_kernel in the steps (FFN up-projection, fused q/k/v, cross K/V, head, dH)"
            res["steps"] = steps
            res["launches_per_step"] = len(inside) / steps
            res["kernel_ms_per_step"] = sum(r[2] - r[1] for r in inside) / 1e6 / steps
        json.dump(res, open(path, "w"), indent=1, sort_keys=True)
    out = "\n".join(lines)
    if len(sys.argv) > 2 and not sys.argv[2].startswith("--"):
        open(sys.argv[2], "w").write(out + "\n")
    print(out)


if __name__ == "__main__":
    main()
