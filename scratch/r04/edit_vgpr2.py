"""r04 assembly edit: next_free_vgpr and accum_offset of ONE kernel set separately.  edit_vgpr2.py KERNEL NEXT_FREE ACCUM file.s"""
import re, sys
(kernel, nfree, accum, path) = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
lines = open(path).read().split('\n')
inside = False
for (i, line) in enumerate(lines):
    if line.strip().startswith('.amdhsa_kernel'):
        inside = kernel in line
    if line.strip().startswith('.end_amdhsa_kernel'):
        inside = False
    if inside and re.match(r'\s*\.amdhsa_next_free_vgpr\s', line):
        lines[i] = re.sub(r'\d+\s*$', str(nfree), line)
    if inside and re.match(r'\s*\.amdhsa_accum_offset\s', line):
        lines[i] = re.sub(r'\d+\s*$', str(accum), line)
for n in [i for (i, l) in enumerate(lines) if l.strip().startswith('.name:') and kernel in l]:
    for j in range(n, min(n + 40, len(lines))):
        if lines[j].strip().startswith('.vgpr_count:'):
            lines[j] = re.sub(r'\d+\s*$', str(nfree), lines[j]); break
open(path, 'w').write('\n'.join(lines))
