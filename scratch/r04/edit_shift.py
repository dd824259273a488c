"""r04 assembly edit on the failing decoder core (allocation 40 kept): the 64-bit shift of the stream window takes its shift amount from
v15 (a copy) instead of v39, the LAST register of the wave's allocation.  edit_shift.py file.s"""
import re, sys
path = sys.argv[1]
text = open(path).read()
start = text.index('_ZN12_GLOBAL__N_122bac_decode_core_kernelENS_10SimdParamsE:')
end = text.index('.end_amdhsa_kernel', start)
body = text[start:end]
(body, n) = re.subn(r'^\tv_lshlrev_b64 v\[12:13\], v39, v\[12:13\]$', '\tv_mov_b32_e32 v15, v39\n\tv_lshlrev_b64 v[12:13], v15, v[12:13]', body, flags=re.M)
print('edit_shift: %d shifts rewritten' % n)
open(path, 'w').write(text[:start] + body + text[end:])
