"""Per-kernel resource usage (VGPR/AGPR/SGPR/LDS/scratch) from a hipcc -save-temps .s file (dev tool)."""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
md = s[s.rfind('amdhsa.kernels'):]
for blk in md.split('  - .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    if pat not in name:
        continue
    g = lambda k: re.search(k + r':\s+(\d+)', blk).group(1)
    print(name[:110], 'agpr', blk.split()[0], 'vgpr', g('.vgpr_count'), 'spill', g('.vgpr_spill_count'), 'scratch', g('.private_segment_fixed_size'),
          'lds', g('.group_segment_fixed_size'), 'sgpr', g('.sgpr_count'))
