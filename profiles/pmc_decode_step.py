"""HBM-side traffic of ONE decode step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/gen_bench.py in the plain bf16
mode: sums the counter over every dispatch of the decode-step kernels and divides by the number of steps (= dispatches of
p5_beam_step_kernel).  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950.
usage: python profiles/pmc_decode_step.py <fetch_db> <write_db> <out.json>"""
import json, sqlite3, sys

STEP_KERNELS = ("p5_skinny_gemm_kernel", "p5_dec_self_attn2_kernel", "p5_dec_cross_attn3_kernel", "p5_head_lse_kernel", "p5_dec_score2_kernel",
                "p5_beam_step_kernel", "p5_rmsnorm_f32in_kernel")


def total(dbfile, counter):
    db = sqlite3.connect(dbfile); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    ev = [t for t in tabs if 'pmc_event' in t][0]; info = [t for t in tabs if 'info_pmc' in t][0]
    disp = [t for t in tabs if 'kernel_dispatch' in t][0]; sym = [t for t in tabs if 'info_kernel_symbol' in t][0]
    rows = cur.execute(f"select s.kernel_name, sum(e.value), count(*) from {ev} e join {info} i on e.pmc_id=i.id join {disp} d on e.event_id=d.event_id "
                       f"join {sym} s on d.kernel_id=s.id where i.name = '{counter}' group by s.kernel_name").fetchall()
    by = {}
    steps = 0
    for name, v, n in rows:
        for k in STEP_KERNELS:
            if k in name:
                by[k] = by.get(k, 0.0) + v
                if k == "p5_beam_step_kernel":
                    steps += n
    return by, steps


f, steps = total(sys.argv[1], "FETCH_SIZE")
w, steps_w = total(sys.argv[2], "WRITE_SIZE")
assert steps > 0 and steps == steps_w, (steps, steps_w)
read = sum(f.values()) * 1024.0 * 2.0 / steps
write = sum(w.values()) * 1024.0 / steps
out = {"decode_steps": steps, "read_bytes_per_step": read, "write_bytes_per_step": write, "traffic_bytes_per_step": read + write,
       "read_kb_raw_by_kernel_per_step": {k: v / steps for k, v in sorted(f.items())}, "write_kb_raw_by_kernel_per_step": {k: v / steps for k, v in sorted(w.items())},
       "note": "FETCH_SIZE x2 (gfx950), separate --pmc passes over tools/gen_bench.py (plain bf16 search, 20 users, beam 10); per decode step = per p5_beam_step_kernel dispatch; "
               "the counters sit on the fabric side of the L2 and count Infinity-Cache hits (MI355X_MICROARCH.md): L2-miss traffic, an upper bound of HBM traffic"}
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
