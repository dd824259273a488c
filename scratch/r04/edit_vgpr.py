"""r04 assembly edit: the VGPR allocation of ONE kernel (descriptor + metadata), nothing else.  edit_vgpr.py KERNEL_SUBSTRING COUNT file.s"""
import re, sys
(kernel, count, path) = (sys.argv[1], int(sys.argv[2]), sys.argv[3])
lines = open(path).read().split('\n')
inside = False
changed = 0
for (i, line) in enumerate(lines):
    if line.strip().startswith('.amdhsa_kernel'):
        inside = kernel in line
    if line.strip().startswith('.end_amdhsa_kernel'):
        inside = False
    if inside and re.match(r'\s*\.amdhsa_(next_free_vgpr|accum_offset)\s', line):
        lines[i] = re.sub(r'\d+\s*$', str(count), line); changed += 1
# metadata (YAML): .name: <kernel> ... .vgpr_count: N  within the same mapping
name_at = [i for (i, l) in enumerate(lines) if l.strip().startswith('.name:') and kernel in l]
for n in name_at:
    for j in range(n, min(n + 40, len(lines))):
        if lines[j].strip().startswith('.vgpr_count:'):
            lines[j] = re.sub(r'\d+\s*$', str(count), lines[j]); changed += 1
            break
open(path, 'w').write('\n'.join(lines))
print('edit_vgpr: %d lines changed' % changed)
