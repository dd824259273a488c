"""r04 assembly edit: like edit_dump.py, but the state is dumped when the wave ENTERS the decode loop (everything the prologue set up:
all VGPRs, s0..s81, the lane's 32 ring words and its 11 probability rows in LDS) and the kernel ends there. The prologue is
deterministic, so a run next to other kernels must dump what a run alone dumps.  edit_dump_early.py file.s"""
import re, sys
path = sys.argv[-1]
marker = sys.argv[1] if len(sys.argv) > 2 else None      # a label of the loop: dump when the wave first gets there
text = open(path).read()
start = text.index('_ZN12_GLOBAL__N_122bac_decode_core_kernelENS_10SimdParamsE:')
end = text.index('.end_amdhsa_kernel', start)
body = text[start:end]
head = '''_ZN12_GLOBAL__N_122bac_decode_core_kernelENS_10SimdParamsE:
	s_load_dwordx4 s[76:79], s[0:1], 0x38
	s_mov_b32 s69, s2
	s_getreg_b32 s66, hwreg(HW_REG_HW_ID)
	s_getreg_b32 s67, hwreg(HW_REG_GPR_ALLOC)
	s_getreg_b32 s68, hwreg(HW_REG_LDS_ALLOC)
	s_memtime s[70:71]
'''
dump = ['\ts_mov_b64 s[80:81], exec', '\ts_mov_b64 exec, -1', '\ts_waitcnt vmcnt(0) lgkmcnt(0)', '\ts_memtime s[72:73]',
        '\tv_mbcnt_lo_u32_b32 v8, -1, 0', '\tv_mbcnt_hi_u32_b32 v8, -1, v8', '\tv_lshl_or_b32 v8, s69, 6, v8',
        '\tv_mul_lo_u32 v8, v8, s78', '\tv_add_u32_e32 v8, 0x1000, v8']
for k in range(40):
    if k != 8:
        dump.append('\tglobal_store_dword v8, v%d, s[76:77] offset:%d' % (k, 4*k))
dump.append('\ts_waitcnt vmcnt(0) lgkmcnt(0)')
for j in range(82):
    dump.append('\tv_mov_b32_e32 v9, s%d' % j)
    dump.append('\tglobal_store_dword v8, v9, s[76:77] offset:%d' % (256 + 4*j))
    if j % 8 == 7:
        dump.append('\ts_waitcnt vmcnt(0)')
# LDS: v3 = ring base of the lane - 512 (row r at v3 + 512 + 256 r); v18 = 8 * lane (probability row k at v18 + 512 k)
for r in range(32):
    dump.append('\tds_read_b32 v9, v3 offset:%d' % (512 + 256*r))
    dump.append('\ts_waitcnt lgkmcnt(0)')
    dump.append('\tglobal_store_dword v8, v9, s[76:77] offset:%d' % (1024 + 4*r))
for k in range(11):
    dump.append('\tds_read_b64 v[10:11], v18 offset:%d' % (512*k))
    dump.append('\ts_waitcnt lgkmcnt(0)')
    dump.append('\tglobal_store_dwordx2 v8, v[10:11], s[76:77] offset:%d' % (1024 + 128 + 8*k))
dump += ['\ts_waitcnt vmcnt(0)', '\ts_endpgm']
(body, nhead) = re.subn(r'^_ZN12_GLOBAL__N_122bac_decode_core_kernelENS_10SimdParamsE:[^\n]*\n', head, body, count=1, flags=re.M)
assert nhead == 1
if marker is None:
    (body, nl) = re.subn(r'^\ts_branch \.LBB3_43$', '\n'.join(dump), body, count=1, flags=re.M)
else:
    (body, nl) = re.subn(r'^(\.%s:[^\n]*)$' % re.escape(marker), lambda m: m.group(1) + '\n' + '\n'.join(dump), body, count=1, flags=re.M)
assert nl == 1
body = re.sub(r'(\.amdhsa_next_free_sgpr\s+)\d+', r'\g<1>88', body)
open(path, 'w').write(text[:start] + body + text[end:])
