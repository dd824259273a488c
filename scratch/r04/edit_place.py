"""r04 assembly edit: the failing decoder core, code and VGPR allocation untouched, reports WHERE it ran: HW_ID / GPR_ALLOC / LDS_ALLOC /
s_memtime captured into SGPRs above the compiler's (next_free_sgpr 60 -> 80) at entry and stored by lane 0 into the wavefront's stage
words right before s_endpgm.  edit_place.py file.s"""
import re, sys
path = sys.argv[1]
text = open(path).read()
start = text.index('_ZN12_GLOBAL__N_122bac_decode_core_kernelENS_10SimdParamsE:')
end = text.index('.end_amdhsa_kernel', start)
body = text[start:end]
head = '''_ZN12_GLOBAL__N_122bac_decode_core_kernelENS_10SimdParamsE:
	s_load_dwordx2 s[64:65], s[0:1], 0x60
	s_getreg_b32 s66, hwreg(HW_REG_HW_ID)
	s_getreg_b32 s67, hwreg(HW_REG_GPR_ALLOC)
	s_getreg_b32 s68, hwreg(HW_REG_LDS_ALLOC)
	s_mov_b32 s69, s2
	s_memtime s[70:71]
'''
tail = '''	s_waitcnt vmcnt(0) lgkmcnt(0)
	s_memtime s[72:73]
	s_getreg_b32 s75, hwreg(HW_REG_HW_ID)
	s_mov_b64 exec, 1
	s_lshl_b32 s74, s69, 8
	v_mov_b32_e32 v1, s74
	s_waitcnt lgkmcnt(0)
	v_mov_b32_e32 v2, s66
	global_store_dword v1, v2, s[64:65]
	v_mov_b32_e32 v3, s75
	global_store_dword v1, v3, s[64:65] offset:4
	v_mov_b32_e32 v4, s68
	global_store_dword v1, v4, s[64:65] offset:8
	v_mov_b32_e32 v5, s67
	global_store_dword v1, v5, s[64:65] offset:12
	v_mov_b32_e32 v6, 0
	global_store_dword v1, v6, s[64:65] offset:16
	v_mov_b32_e32 v7, s70
	global_store_dword v1, v7, s[64:65] offset:20
	v_mov_b32_e32 v8, s71
	global_store_dword v1, v8, s[64:65] offset:24
	v_mov_b32_e32 v9, s72
	global_store_dword v1, v9, s[64:65] offset:28
	v_mov_b32_e32 v10, s73
	global_store_dword v1, v10, s[64:65] offset:32
	s_endpgm
'''
assert body.count('\ts_endpgm\n') == 1
(body, nhead) = re.subn(r'^_ZN12_GLOBAL__N_122bac_decode_core_kernelENS_10SimdParamsE:[^\n]*\n', head, body, count=1, flags=re.M)
assert nhead == 1
body = body.replace('\ts_endpgm\n', tail, 1)
body = re.sub(r'(\.amdhsa_next_free_sgpr\s+)\d+', r'\g<1>80', body)
open(path, 'w').write(text[:start] + body + text[end:])
