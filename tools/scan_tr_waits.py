"""Static scan of the device ISA (hipcc -S --cuda-device-only of the three translation units) for the instruction pattern of the one
hardware-only wrong-result failure of round 5 (profiles/r05_call21_keep_mask_bisect.txt): an MFMA that consumes
ds_read_b64_tr_b16 results behind a PARTIAL s_waitcnt lgkmcnt(n > 0), the read's destination registers written by a VALU instruction
shortly before.  Usage:  for tu in p5_attn_tu p5_gemm_tu p5_lib; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -I openp5_amd/csrc -S
--cuda-device-only -o /tmp/$tu.s openp5_amd/csrc/$tu.hip; done; python tools/scan_tr_waits.py /tmp/p5_*.s   (output: profiles/r05_tr_read_wait_scan.txt)"""
import re, sys, collections
# for every kernel: MFMAs whose A/B operand registers were last written by ds_read_b64_tr_b16 and that issue behind a PARTIAL lgkmcnt wait
# (some LDS reads still outstanding); of those, the ones whose tr-read destination had been written by a VALU instruction within the
# previous W instructions (the pattern of the mask-storing forward that failed on the hardware)
W = 24
def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()
for f in sys.argv[1:]:
    s = open(f).read()
    for km in re.finditer(r"^(_Z\S+):.*?\n(.*?)\.end_amdhsa_kernel", s, re.S | re.M):
        name, body = km.group(1), km.group(2).split("\n")
        ins = []
        for l in body:
            t = l.strip()
            if not t or t.startswith((";", ".")) or t.endswith(":"): continue
            ins.append(t)
        last_writer = {}      # reg -> (index, kind)
        outstanding = 0       # LDS reads in flight after the last wait (upper bound)
        part = risky = total = 0
        for i, t in enumerate(ins):
            op = t.split()[0]
            args = [a.strip() for a in t[len(op):].split(",")]
            if op.startswith("s_waitcnt"):
                m = re.search(r"lgkmcnt\((\d+)\)", t)
                if m: outstanding = min(outstanding, int(m.group(1)))
                continue
            if op.startswith("ds_read") or op.startswith("ds_bpermute"):
                outstanding += 1
                for r in regs(args[0]): last_writer[r] = (i, "tr" if "_tr_" in op else "lds", last_writer.get(r))
                continue
            if op.startswith("v_mfma"):
                srcs = regs(args[1]) | regs(args[2])
                trs = [last_writer[r] for r in srcs if r in last_writer and last_writer[r][1] == "tr"]
                if trs:
                    total += 1
                    if outstanding > 0:
                        part += 1
                        # was any of those tr destinations VALU-written shortly before the tr read?
                        if any(w[2] is not None and w[2][1] == "valu" and w[0] - w[2][0] <= W for w in trs): risky += 1
                for r in regs(args[0]): last_writer[r] = (i, "mfma", None)
                continue
            if op.startswith("v_") and args and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                for r in regs(args[0]): last_writer[r] = (i, "valu", None)
        if total:
            print(f"{name[:60]:60s} mfma<-tr {total:4d}  behind a partial wait {part:4d}  of those tr dst VALU-written <= {W} instrs before {risky:4d}")
