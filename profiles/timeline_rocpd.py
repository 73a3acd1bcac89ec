"""Timeline of the training steps in a rocprofv3 (rocpd sqlite) kernel trace: per step (AdamW launch to AdamW launch) the wall time, the
time during which NO kernel runs, the time during which exactly one / more than one kernel runs, per-queue busy time, and the kernels
that follow the largest idle gaps.   usage: python profiles/timeline_rocpd.py x_results.db [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    suf = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace("rocpd_kernel_dispatch", "")
    cols = [r[1] for r in cur.execute(f"pragma table_info(rocpd_kernel_dispatch{suf})")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = cur.execute(f"select s.kernel_name, d.start, d.end, {('d.' + qcol) if qcol else '0'} from rocpd_kernel_dispatch{suf} d "
                       f"join rocpd_info_kernel_symbol{suf} s on d.kernel_id = s.id order by d.start").fetchall()
    adam = [i for i, r in enumerate(rows) if "p5_adamw" in r[0] and "segments" not in r[0]]
    out = [f"dispatches {len(rows)}, optimizer launches {len(adam)}, queue column {qcol}"]
    if len(adam) < 4:
        print("\n".join(out)); return
    steps = []
    for a, b in zip(adam[2:-1], adam[3:]):          # steady-state steps: from the end of one AdamW to the end of the next
        seg = rows[a + 1:b + 1]
        t0, t1 = rows[a][2], rows[b][2]
        ev = []
        for name, st, en, q in seg:
            ev.append((max(st, t0), 1)); ev.append((min(en, t1), -1))
        ev.sort()
        depth, last, idle, one, multi = 0, t0, 0, 0, 0
        for t, dlt in ev:
            span = t - last
            if depth == 0: idle += span
            elif depth == 1: one += span
            else: multi += span
            depth += dlt; last = t
        idle += max(0, t1 - last)
        busy_q = {}
        for name, st, en, q in seg:
            busy_q[q] = busy_q.get(q, 0) + (en - st)
        # gaps: idle intervals followed by which kernel
        gaps = []
        cur_end = t0
        for name, st, en, q in sorted(seg, key=lambda r: r[1]):
            if st > cur_end: gaps.append((st - cur_end, short(name)))
            cur_end = max(cur_end, en)
        steps.append(dict(wall=t1 - t0, idle=idle, one=one, multi=multi, busy_q=busy_q, n=len(seg), gaps=gaps))
    n = len(steps)
    avg = lambda k: sum(s[k] for s in steps) / n / 1e3
    out.append(f"steady-state steps analysed: {n}; launches per step {sum(s['n'] for s in steps) / n:.0f}")
    out.append(f"wall {avg('wall'):.0f} us = idle (no kernel running) {avg('idle'):.0f} + one kernel {avg('one'):.0f} + two or more {avg('multi'):.0f}")
    qs = sorted({q for s in steps for q in s["busy_q"]})
    out.append("busy us per queue: " + ", ".join(f"{q}: {sum(s['busy_q'].get(q, 0) for s in steps) / n / 1e3:.0f}" for q in qs))
    agg = {}
    for s in steps:
        for g, name in s["gaps"]:
            a = agg.setdefault(name, [0, 0]); a[0] += g; a[1] += 1
    out.append("idle time in front of (us per step, count per step, avg gap us):")
    for name, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
        out.append(f"  {t / n / 1e3:7.1f}  {c / n:5.1f}  {t / c / 1e3:6.2f}  {name}")
    text = "\n".join(out)
    if len(sys.argv) > 2: open(sys.argv[2], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
