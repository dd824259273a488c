#!/usr/bin/env python
"""ISA guard for a store hazard the compiler does not keep apart on gfx950 (run by `__graft_entry__.build()`).

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

_STORE = re.compile(r'^\s*buffer_store_dwordx([34])\s+v\[(\d+):(\d+)\]\s*,\s*([^,]+),\s*s\[\d+:\d+\]\s*,\s*([^\s,]+)')
_VREG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')
_INSN = re.compile(r'^\s+([a-z_][a-z0-9_]*)\s*(.*)$')


def code_objects(path):
    """The gfx950 code objects (bytes) bundled into a HIP shared library."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, 'fat.bin')
        subprocess.run([os.path.join(LLVM_BIN, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, path], check=True)
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
        done = subprocess.run([os.path.join(LLVM_BIN, 'llvm-objdump'), '-d', '--no-show-raw-insn', f.name], check=True,
                              stdout=subprocess.PIPE, universal_newlines=True)
    return done.stdout


def _written_vgprs(mnemonic, operands):
    """VGPRs a vector-ALU instruction writes: its first operand (v_cmp* write SGPRs / VCC; v_readlane & co. write SGPRs)."""
    if not mnemonic.startswith('v_') or mnemonic.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane', 'v_nop')):
        return set()
    first = operands.split(',')[0]
    m = _VREG.search(first)
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def _wait_states(mnemonic, operands):
    if mnemonic == 's_nop':
        try:
            return int(operands.split()[0], 0) + 1
        except (ValueError, IndexError):
            return 1
    return 1


def scan(text, origin):
    """Findings in one disassembly / assembly listing: list of strings."""
    findings = []
    function = '?'
    insns = []          # (mnemonic, operands, line) of the current function

    def flush():
        for (i, (mn, ops, line)) in enumerate(insns):
            m = _STORE.match(' ' + mn + ' ' + ops)
            if not m:
                continue
            soffset = m.group(5)
            if not re.match(r'^(s\d+|m0|ttmp\d+|vcc_lo|vcc_hi)$', soffset):
                continue
            data = set(range(int(m.group(2)), int(m.group(3)) + 1))
            for direction in (-1, 1):
                (gap, j) = (0, i + direction)
                while 0 <= j < len(insns) and gap < WINDOW:
                    (mn2, ops2, line2) = insns[j]
                    hit = _written_vgprs(mn2, ops2) & data
                    if hit:
                        findings.append('{0}: {1}: `{2}` {3} `{4}` ({5} wait states apart, v{6})'.format(
                            origin, function, line.strip(), 'behind' if direction < 0 else 'ahead of', line2.strip(), gap,
                            sorted(hit)[0]))
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
            findings += scan(disassemble(blob), '{0}#{1}'.format(os.path.basename(path), k))
    return findings


def main(argv):
    if not argv:
        sys.stderr.write(__doc__)
        return 2
    findings = check(argv)
    for line in findings:
        print(line)
    if findings:
        print('isa_guard: {} 16-byte buffer store(s) with a register soffset next to a vector write of their data'.format(len(findings)))
        return 1
    print('isa_guard: clean ({})'.format(', '.join(os.path.basename(p) for p in argv)))
    return 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
