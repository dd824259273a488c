#!/usr/bin/env python
"""ISA guard for two gfx950 hazards the compiler does not know about (run by `__graft_entry__.build()` on the shipped code objects).

(1) A STORE HAZARD.

Round 2 met it ("a trap met on the way"; scratch/dropped/README.md): a 16-byte MUBUF store
(`buffer_store_dwordx4`, also x3) whose soffset operand is a REGISTER, issued right behind the vector instruction that wrote
one of its data registers, stores stale lanes (lanes 12-15 of every row of 16: the writer's last pass had not landed). LLVM's
hazard recogniser treats MUBUF stores with an SGPR soffset as exempt from the ">64-bit store data" hazard, in both directions
(vector write -> store, store -> vector overwrite of its data), so nothing separates the two. A build with the fault passed
all 390 GPU tests until a scheduling fence went; only the full-size parity tests showed it. This script makes the pattern a
BUILD error instead:

    for every buffer_store_dwordx3 / x4 whose soffset is a register (sN, m0, ttmpN -- not a literal / inline constant):
      no VALU instruction (v_*, including MFMA / accumulator moves) that writes one of its data VGPRs may sit within
      WINDOW wait states before it, nor within WINDOW wait states after it            (s_nop N counts N + 1; any other
                                                                                      instruction counts 1)

(2) A 64-BIT SHIFT WHOSE SHIFT AMOUNT SITS IN THE LAST REGISTER OF THE WAVE'S VGPR ALLOCATION.
Round 3 met it as "a decoder core that derails only next to other kernels" (DESIGN.md section 5); round 4 reduced it to this
(scratch/r04/probe_shift64.hip, probe_last_vgpr.hip; logs under profiles/r04_decoder_fault/): `v_lshlrev_b64`, `v_lshrrev_b64` and
`v_ashrrev_i64` take a 32-bit shift amount in src0, but with src0 = vK and K the LAST register the wave owns (K + 1 == the
allocation: next_free_vgpr rounded up to the granule of 8, AGPRs included) the result is wrong whenever other waves run on the same
SIMD -- 13 % of the shifts next to MFMA waves, 0.02 % next to plain vector waves, never alone on the SIMD, never with K + 1 inside
the allocation -- as if the operand were fetched as the pair (vK, vK+1), the upper half belonging to a neighbour. The register
allocator hands out the highest register last, so the pattern appears exactly when a kernel's pressure peaks at a multiple of 8
and disappears with any unrelated edit (the first decoder core of coder_simd.hip used 40 of 40 registers). Rule:

    no v_lshlrev_b64 / v_lshrrev_b64 / v_ashrrev_i64 (and, untested but of the same operand shape, v_lshl_add_u64 with a register shift, v_trig_preop_f64 and
    v_cmp[x]_class_f64 for their 32-bit src1) may read its 32-bit VGPR operand from v(A - 1), A = the kernel's VGPR allocation

It works on the device code actually shipped: the gfx950 code objects are cut out of the `.hip_fatbin` section of
libeae_hip.so (clang offload bundles), disassembled with llvm-objdump, and scanned function by function. Branch targets end a
window conservatively (a label between the two instructions does not excuse them: the fall-through path still runs).

Usage: isa_guard.py <libeae_hip.so | file.s> [...]; exit code 1 and one line per finding when the pattern is present.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM_BIN = os.environ.get('EAE_LLVM_BIN', '/opt/rocm/lib/llvm/bin')
WINDOW = 3          # wait states on either side of the store that must be free of vector writes to its data registers
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'

_STORE = re.compile(r'^\s*buffer_store_dwordx([34])\s+([va])\[(\d+):(\d+)\]\s*,\s*([^,]+),\s*s\[\d+:\d+\]\s*,\s*([^\s,]+)')
_VREG = re.compile(r'\b[va](\d+)\b|\b[va]\[(\d+):(\d+)\]')
_SHIFT64 = re.compile(r'^(v_lshlrev_b64|v_lshrrev_b64|v_ashrrev_i64)(?:_e64)?$')
_SRC1_32 = re.compile(r'^(v_trig_preop_f64|v_cmpx?_class_f64|v_lshl_add_u64)(?:_e32|_e64)?$')
GRANULE = 8         # VGPR allocation granule of gfx950 (wave64), MI355X_MICROARCH.md
_INSN = re.compile(r'^\s+([a-z_][a-z0-9_]*)\s*(.*)$')


def _tool(name):
    path = os.path.join(LLVM_BIN, name)
    if not os.path.isfile(path):
        raise RuntimeError('isa_guard: {0} not found (set EAE_LLVM_BIN to the directory that holds llvm-objcopy, llvm-objdump and '
                           'llvm-readelf); the guard cannot vouch for the shipped code objects without them'.format(path))
    return path


def code_objects(path):
    """The gfx950 code objects (bytes) bundled into a HIP shared library."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, 'fat.bin')
        subprocess.run([_tool('llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, path], check=True)
        with open(fat, 'rb') as f:
            data = f.read()
    out = []
    pos = data.find(MAGIC)
    while pos >= 0:
        (count,) = struct.unpack_from('<Q', data, pos + len(MAGIC))
        cursor = pos + len(MAGIC) + 8
        for _ in range(count):
            (offset, size, triple_len) = struct.unpack_from('<QQQ', data, cursor)
            triple = data[cursor + 24:cursor + 24 + triple_len].decode()
            cursor += 24 + triple_len
            if 'gfx950' in triple and size:
                out.append(data[pos + offset:pos + offset + size])
        pos = data.find(MAGIC, pos + len(MAGIC))
    return out


def disassemble(blob):
    with tempfile.NamedTemporaryFile(suffix='.co') as f:
        f.write(blob)
        f.flush()
        done = subprocess.run([_tool('llvm-objdump'), '-d', '--no-show-raw-insn', f.name], check=True,
                              stdout=subprocess.PIPE, universal_newlines=True)
    return done.stdout


def allocations(blob):
    """{kernel name: VGPR allocation of a wave} from the code object's metadata: `.vgpr_count` (on gfx90a and later the whole
    unified budget, ArchVGPRs up to accum_offset plus AccVGPRs) rounded up to the allocation granule. A kernel with AccVGPRs is
    left out: its last physical registers are accumulators, which no vector-ALU operand of rule (2) can name."""
    with tempfile.NamedTemporaryFile(suffix='.co') as f:
        f.write(blob)
        f.flush()
        notes = subprocess.run([_tool('llvm-readelf'), '--notes', f.name], check=True, stdout=subprocess.PIPE,
                               universal_newlines=True).stdout
    return _allocations_from_metadata(notes)


def _allocations_from_metadata(text):
    """The kernel entries of amdhsa.kernels list their keys alphabetically: `.agpr_count` opens an entry, `.symbol` (<name>.kd)
    names it, `.vgpr_count` closes it."""
    out = {}
    agpr = name = None
    for line in text.splitlines():
        m = re.match(r'^\s*-?\s*\.(agpr_count|symbol|vgpr_count):\s*(\S+)', line)
        if not m:
            continue
        (key, value) = (m.group(1), m.group(2).strip("'\""))
        if key == 'agpr_count':
            (agpr, name) = (int(value), None)
        elif key == 'symbol':
            name = value[:-3] if value.endswith('.kd') else value
        elif key == 'vgpr_count' and name is not None:
            if not agpr:
                out[name] = max(GRANULE, (int(value) + GRANULE - 1)//GRANULE*GRANULE)
            (agpr, name) = (None, None)
    return out


def _regs(file_, token):
    """Register indices of one operand token for the register file 'v' or 'a'."""
    m = re.match(r'^%s(\d+)$' % file_, token)
    if m:
        return {int(m.group(1))}
    m = re.match(r'^%s\[(\d+):(\d+)\]$' % file_, token)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def _written(mnemonic, operands):
    """(file, index) of the registers a vector-ALU instruction writes: its first operand, a VGPR or (MFMA, v_accvgpr_write) an
    AccVGPR. v_cmp* write SGPRs / VCC; v_readlane & co. write SGPRs."""
    if not mnemonic.startswith('v_') or mnemonic.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane', 'v_nop')):
        return set()
    first = operands.split(',')[0].strip()
    return {('v', i) for i in _regs('v', first)} | {('a', i) for i in _regs('a', first)}


def _written_vgprs(mnemonic, operands):
    return {i for (f, i) in _written(mnemonic, operands) if f == 'v'}


def _wait_states(mnemonic, operands):
    if mnemonic == 's_nop':
        try:
            return int(operands.split()[0], 0) + 1
        except (ValueError, IndexError):
            return 1
    return 1


def scan(text, origin, alloc=None):
    """Findings in one disassembly / assembly listing: list of strings. `alloc`: {function: VGPR allocation} for rule (2); for an
    assembly file it is read from the `.amdhsa_next_free_vgpr` directives of the file itself."""
    findings = []
    function = '?'
    insns = []          # (mnemonic, operands, line) of the current function
    if alloc is None:
        alloc = {}
        for m in re.finditer(r'\.amdhsa_kernel\s+(\S+)(.*?)\.end_amdhsa_kernel', text, flags=re.S):
            n = re.search(r'\.amdhsa_next_free_vgpr\s+(\d+)', m.group(2))
            if n:
                alloc[m.group(1)] = max(GRANULE, (int(n.group(1)) + GRANULE - 1)//GRANULE*GRANULE)

    def flush():
        last = alloc.get(function)
        for (i, (mn, ops, line)) in enumerate(insns):
            # ---- rule (2): a 32-bit operand of a 64-bit instruction read from the last register of the allocation
            toks = [t.strip() for t in ops.split(',')]
            src = None
            if _SHIFT64.match(mn) and len(toks) >= 2:
                src = toks[1]
            elif _SRC1_32.match(mn) and len(toks) >= 3:
                src = toks[2]
            if src is not None and last is not None and _regs('v', src) == {last - 1}:
                findings.append('{0}: {1}: `{2}` reads its 32-bit operand from v{3}, the last register of the wave\'s allocation of {4} '
                                '(rule 2: wrong results next to other waves)'.format(origin, function, line.strip(), last - 1, last))
            # ---- rule (1): 16-byte buffer store with a register soffset next to a vector write of its data
            m = _STORE.match(' ' + mn + ' ' + ops)
            if not m:
                continue
            soffset = m.group(6)
            if not re.match(r'^(s\d+|m0|ttmp\d+|vcc_lo|vcc_hi)$', soffset):
                continue
            data = {(m.group(2), r) for r in range(int(m.group(3)), int(m.group(4)) + 1)}
            for direction in (-1, 1):
                (gap, j) = (0, i + direction)
                while 0 <= j < len(insns) and gap < WINDOW:
                    (mn2, ops2, line2) = insns[j]
                    hit = _written(mn2, ops2) & data
                    if hit:
                        findings.append('{0}: {1}: `{2}` {3} `{4}` ({5} wait states apart, {6}{7}) (rule 1)'.format(
                            origin, function, line.strip(), 'behind' if direction < 0 else 'ahead of', line2.strip(), gap,
                            sorted(hit)[0][0], sorted(hit)[0][1]))
                        break
                    gap += _wait_states(mn2, ops2)
                    j += direction
        del insns[:]

    for raw in text.splitlines():
        line = raw.split(';')[0].split('//')[0].rstrip()
        if not line:
            continue
        label = re.match(r'^(?:[0-9a-f]+ )?<?([A-Za-z_.$][\w.$]*)>?:\s*$', line)
        if label:
            name = label.group(1)
            if not name.startswith(('.L', 'L')):       # a function symbol: windows never cross functions
                flush()
                function = name
            continue
        m = _INSN.match(line)
        if m:
            insns.append((m.group(1), m.group(2), line))
    flush()
    return findings


def check(paths):
    findings = []
    for path in paths:
        if path.endswith('.s'):
            with open(path) as f:
                findings += scan(f.read(), os.path.basename(path))
            continue
        blobs = code_objects(path)
        if not blobs:
            raise RuntimeError('no gfx950 code object in {}'.format(path))
        for (k, blob) in enumerate(blobs):
            findings += scan(disassemble(blob), '{0}#{1}'.format(os.path.basename(path), k), allocations(blob))
    return findings


def main(argv):
    if not argv:
        sys.stderr.write(__doc__)
        return 2
    findings = check(argv)
    for line in findings:
        print(line)
    if findings:
        print('isa_guard: {} finding(s): rule 1 = a 16-byte buffer store with a register soffset next to a vector write of its data, '
              'rule 2 = a 64-bit shift (or v_trig_preop_f64 / v_cmp_class_f64) fed from the last VGPR of the allocation'.format(len(findings)))
        return 1
    print('isa_guard: clean ({})'.format(', '.join(os.path.basename(p) for p in argv)))
    return 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
