"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (dev helper)."""
import re, sys
t = open(sys.argv[1]).read()
seen = set()
for b in t.split("Function Name: ")[1:]:
    name = b.split("\n")[0]
    if name in seen:
        continue
    seen.add(name)
    def g(k):
        m = re.search(re.escape(k) + r": (\d+)", b)
        return m.group(1) if m else "?"
    short = re.sub(r"^_Z\d+", "", name)[:70]
    print("%-70s vgpr=%4s agpr=%4s spill=%s scratch=%s occ=%s lds=%s" % (
        short, g("VGPRs"), g("AGPRs"), g("VGPRs Spill"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"), g("LDS Size [bytes/block]")))
