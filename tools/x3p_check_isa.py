#!/usr/bin/env python3
"""Static check of conv_x3p_kernel's ISA (csrc/conv_x3.hip).  The activation loads of the K loop are inline-asm
global_load_dwordx4 whose results stay IN FLIGHT across a barrier and the loop's back edge; hipcc does not know that, so a
register copy (v_mov / v_accvgpr_write / spill) of such a register placed between the load and its s_waitcnt would read stale
data.  The source ties the registers through "+v" so that no copy is needed; this script verifies it per instantiation:
  * every VGPR that is the destination of a plain global_load_dwordx4 is written by nothing else except a constant move, and
  * is read only by arithmetic (the bf16 split), never by a move, an AGPR write or a scratch store;
  * the kernels use no scratch.
usage: python tools/x3p_check_isa.py [file.s]   (default: compiles csrc/conv_x3.hip to assembly).  Exit status 1 on a finding."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COPY = ('v_mov_b32', 'v_mov_b64', 'v_accvgpr_write', 'scratch_store', 'buffer_store', 'v_pk_mov', 'v_swap')


def regs(tok):
    """v12 -> {12}; v[4:7] -> {4..7}"""
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', tok)
    return {int(m.group(1))} if m else set()


def assembly(extra=()):
    out = os.path.join(tempfile.mkdtemp(prefix='x3p_isa_'), 'conv_x3.s')
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-fhip-fp32-correctly-rounded-divide-sqrt',
           '--cuda-device-only', '-S', os.path.join(ROOT, 'htd_amd/csrc/conv_x3.hip'), '-o', out, *extra]
    subprocess.run(cmd, check=True, capture_output=True)
    return open(out).read()


def check(text):
    findings, n = [], 0
    for m in re.finditer(r'^(_ZN\S*conv_x3p_kernel\S*):[^\n]*\n(.*?)^\.Lfunc_end', text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        short = re.search(r'conv_x3p_kernel(\w+?)EEv', name).group(1)
        n += 1
        ins, in_asm = [], False                  # (mnemonic, operands, written by inline asm)
        for line in body.splitlines():
            if '#ASMSTART' in line:
                in_asm = True
            if '#ASMEND' in line:
                in_asm = False
            line = line.split(';')[0].strip()
            if not line or line.endswith(':') or line.startswith('.'):
                continue
            op, _, rest = line.partition(' ')
            ins.append((op, [t.strip() for t in rest.replace('\t', ' ').split(',')], in_asm))
        loads = [i for i, (op, args, a) in enumerate(ins) if a and op == 'global_load_dwordx4']
        waits = [i for i, (op, args, a) in enumerate(ins) if a and op == 's_waitcnt' and 'vmcnt' in args[0]]
        if not loads or not waits:
            findings.append(f'{short}: inline-asm loads / waits not found (parser out of date?)')
            continue
        dest = set()
        for i in loads:
            dest |= regs(ins[i][1][0])
        # After the prologue's drain (the first asm wait) a staging register is "armed" from the first load into it: from
        # there to the K loop's last counted wait a load into it may be outstanding at any point (text order approximates
        # the loop), so nothing but a load may write it and no move / AGPR write / store may read it.
        armed = set()
        for op, args, a in ins[waits[0] + 1:waits[-1] + 1]:
            if a and op == 'global_load_dwordx4':
                armed |= regs(args[0])
                continue
            if not args or op.startswith('s_'):
                continue
            stores = op.startswith(('global_store', 'ds_write', 'ds_store', 'scratch_store', 'buffer_store'))
            wr = set() if stores else regs(args[0])
            rd = set()
            for t in (args if stores else args[1:]):
                rd |= regs(t.split(' ')[0])
            if wr & armed:
                findings.append(f'{short}: {op} {", ".join(args)} writes a staging register inside the K loop')
            if rd & armed and op.startswith(COPY):
                findings.append(f'{short}: {op} {", ".join(args)} copies a staging register')
        if 'scratch_' in body:
            findings.append(f'{short}: uses scratch')
    return n, findings


if __name__ == '__main__':
    text = open(sys.argv[1]).read() if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else assembly(sys.argv[1:])
    n, findings = check(text)
    print(f'{n} conv_x3p_kernel instantiations checked, {len(findings)} findings')
    for f in findings:
        print('  ' + f)
    sys.exit(1 if findings or n == 0 else 0)
