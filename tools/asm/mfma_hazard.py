"""Static check for the MFMA source-operand overwrite found in round 3 (DESIGN.md "MFMA operand hazard").

What it looks for.  A `v_mfma_*` reads SrcA / SrcB (and SrcC) from VGPRs.  MFMAs that issue back to back queue in front of the
matrix pipe; on gfx950 the ones at the back of such a queue were observed to pick up A / B values written by VALU instructions that
issued a few slots AFTER them (wrong dS / dW_l at H = 8, hd = 32; twelve idle issue slots after the group cured it, eight did not).
Neither hipcc's hazard recogniser nor the hardware orders that pair, so this checker does it on the shipped code object: for
every MFMA it walks the following WINDOW issue slots along every control-flow path (fall-through and both sides of each branch)
and reports any VALU instruction whose destination registers overlap the MFMA's SrcA / SrcB registers.  `s_nop k` counts as k + 1
slots, every other instruction as one; the walk stops early at `s_endpgm`.

Asynchronous writers (ds_read*, global/buffer/scratch loads) are reported separately and are NOT violations: their data returns
tens to hundreds of cycles after issue (MI355X_MICROARCH.md: ds_read latency >= ~50 cycles on top of the issue queue), which is why
the GEMM main loops — fragments reloaded right behind the MFMA cluster that read them, as in the guide's verified 256^2 template —
never showed the problem.

Input: assembly text, either hipcc -save-temps `.s` or `llvm-objdump -d` of the device code object.
    python tools/asm/mfma_hazard.py <file> [kernel-name filter] [--window N] [--show K]
Library use: tests/test_mfma_hazard_cpu.py."""
import re
import subprocess
import sys

WINDOW = 12

_REG = re.compile(r'\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b')
_LABEL = re.compile(r'^([.\w$]+):')
_OBJ_LABEL = re.compile(r'^[0-9a-f]+ <([^>]+)>:')
_OBJ_INSN = re.compile(r'^\s+([a-z_][\w.]*\b.*?)\s*//\s*([0-9A-Fa-f]+):')

VALU_PREFIX = ('v_',)
NOT_VALU = ('v_mfma', 'v_smfmac', 'v_nop')
ASYNC_PREFIX = ('ds_read', 'ds_load', 'global_load', 'buffer_load', 'scratch_load', 'flat_load', 'ds_bpermute', 'ds_permute', 'ds_swizzle')


def _regs(tok):
    """VGPR / AGPR set named by one operand token ('v[4:7]', 'v12', 'a[0:3]'); AGPRs are offset by 1000"""
    out = set()
    for m in _REG.finditer(tok):
        if m.group(1):
            base = 1000 if m.group(1) == 'a' else 0
            out.update(range(base + int(m.group(2)), base + int(m.group(3)) + 1))
        else:
            base = 1000 if m.group(4) == 'a' else 0
            out.add(base + int(m.group(5)))
    return out


class Insn:
    __slots__ = ('text', 'op', 'dst', 'srcs', 'line', 'addr')

    def __init__(self, text, line, addr=None):
        self.text, self.line, self.addr = text, line, addr
        parts = text.split(None, 1)
        self.op = parts[0]
        ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
        self.dst = _regs(ops[0]) if ops else set()
        self.srcs = [_regs(o) for o in ops[1:]]


def parse(text):
    """-> {kernel name: (insns, labels {name: index})}.  Accepts -save-temps .s and llvm-objdump -d output."""
    kernels, cur, name = {}, None, None
    objdump = bool(re.search(r'^[0-9a-f]{8,16} <', text, re.M))
    for ln, raw in enumerate(text.split('\n'), 1):
        if objdump:
            m = _OBJ_LABEL.match(raw)
            if m:
                lab = m.group(1)
                if not lab.startswith('L') and not lab.startswith('.L') and '$local' not in lab:
                    name = lab
                    cur = kernels.setdefault(name, ([], {}))
                elif cur is not None:
                    cur[1][lab.split('$')[0]] = len(cur[0])
                continue
            m = _OBJ_INSN.match(raw)
            if m and cur is not None:
                cur[0].append(Insn(re.sub(r'\s+', ' ', m.group(1).strip()), ln, int(m.group(2), 16)))
            continue
        line = raw.split(';')[0].rstrip() if not raw.lstrip().startswith(';;#') else ''
        if not line.strip():
            continue
        m = _LABEL.match(line.strip())
        if m and not line.startswith('\t'):
            lab = m.group(1)
            if lab.startswith('.L') or lab.startswith('L'):
                if cur is not None:
                    cur[1][lab] = len(cur[0])
            elif not lab.startswith('.') and not lab.startswith('__hip') and not lab.startswith('_ZTS'):
                name = lab
                cur = kernels.setdefault(name, ([], {}))
            continue
        st = line.strip()
        if st.startswith('.') or cur is None:
            if st.startswith('.end_amdhsa_kernel') or st.startswith('.Lfunc_end'):
                pass
            continue
        if re.match(r'^[a-z_]', st):
            cur[0].append(Insn(re.sub(r'\s+', ' ', st), ln))
    return {k: v for k, v in kernels.items() if any(i.op.startswith('v_mfma') for i in v[0])}


def _slots(ins):
    if ins.op == 's_nop':
        try:
            return int(ins.text.split()[1], 0) + 1
        except (IndexError, ValueError):
            return 1
    return 1


def _is_valu_writer(ins):
    return ins.op.startswith(VALU_PREFIX) and not ins.op.startswith(NOT_VALU) and not ins.op.startswith('v_cmp') and bool(ins.dst)


def _branch_target(ins):
    if ins.op.startswith('s_cbranch') or ins.op == 's_branch':
        t = ins.text.split()[-1]
        return t
    return None


def operand_origin(insns, i, reg, back=200):
    """who produced register `reg` as an operand of MFMA i (last writer in program text before it): 'valu' (packed / converted by
    the VALU: the operands the round-3 failure was about), 'async' (LDS / memory return: GEMM-style fragments), 'acc', or 'unknown'"""
    for k in range(i - 1, max(-1, i - back), -1):
        if reg in insns[k].dst:
            op = insns[k].op
            if op.startswith(ASYNC_PREFIX): return 'async'
            if op.startswith('v_accvgpr') or op.startswith('v_mfma'): return 'acc'
            if op.startswith('v_'): return 'valu'
            return 'unknown'
    return 'unknown'


def run_position(insns, i, gap=4):
    """1-based position of MFMA i in its run: MFMAs in program text with at most `gap` other issue slots between neighbours"""
    k, slots = 1, 0
    for j in range(i - 1, -1, -1):
        if insns[j].op.startswith('v_mfma'):
            k += 1
            slots = 0
        else:
            slots += _slots(insns[j])
            if slots > gap:
                break
    return k


def check_kernel(insns, labels, window=WINDOW):
    """-> (violations, async_hits): each a list of (mfma index, writer index, slots between, overlapping registers)"""
    viol, asyn = [], []
    addr_index = {x.addr: k for k, x in enumerate(insns) if x.addr is not None}
    for i, ins in enumerate(insns):
        if not ins.op.startswith('v_mfma') and not ins.op.startswith('v_smfmac'):
            continue
        ab = set()
        for s in ins.srcs[:2]:
            ab |= s
        if not ab:
            continue
        # walk: (index, slots used so far)
        seen = {}
        stack = [(i + 1, 0)]
        while stack:
            j, used = stack.pop()
            while j < len(insns) and used < window:
                if seen.get(j, window + 1) <= used:
                    break
                seen[j] = used
                w = insns[j]
                if w.op == 's_endpgm':
                    break
                hit = w.dst & ab
                if hit:
                    if _is_valu_writer(w):
                        viol.append((i, j, used, sorted(hit)))
                    elif w.op.startswith(ASYNC_PREFIX):
                        asyn.append((i, j, used, sorted(hit)))
                tgt = _branch_target(w)
                used += _slots(w)
                if tgt is not None:
                    t = labels.get(tgt, labels.get(tgt.split('$')[0]))
                    if t is None and w.addr is not None and re.fullmatch(r'-?\d+', tgt):
                        # llvm-objdump prints the relative target: dwords from the next instruction (simm16)
                        off = int(tgt)
                        off = off - 65536 if off >= 32768 else off
                        t = addr_index.get(w.addr + 4 + 4 * off)
                    if t is not None:
                        stack.append((t, used))
                    if w.op == 's_branch':
                        break
                j += 1
    return viol, asyn


def check_text(text, name_filter='', window=WINDOW):
    """-> {kernel: dict(mfma=count, violations=[...], asynchronous=count, insns=list)}"""
    out = {}
    for name, (insns, labels) in parse(text).items():
        if name_filter and name_filter not in name and name_filter not in demangle(name):
            continue
        v, a = check_kernel(insns, labels, window)
        built = [x for x in v if operand_origin(insns, x[0], x[3][0]) == 'valu']
        out[name] = dict(mfma=sum(1 for x in insns if x.op.startswith('v_mfma')), violations=v, valu_built=built, asynchronous=len(a), insns=insns)
    return out


def demangle(n):
    try:
        return subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip() or n
    except OSError:
        return n


def describe(insns, v, context=0):
    i, j, used, regs = v
    lines = [f'  MFMA   @{insns[i].line}: {insns[i].text}']
    if context:
        for k in range(i + 1, min(j, i + 1 + context)):
            lines.append(f'         @{insns[k].line}: {insns[k].text}')
    lines.append(f'  writer @{insns[j].line}: {insns[j].text}    <- {used} issue slot(s) after the MFMA, overwrites v{regs[0]}..v{regs[-1]}')
    return '\n'.join(lines)


def disassemble_so(path):
    """every gfx950 device code object of a HIP shared library (one offload bundle per translation unit, concatenated in .hip_fatbin)
    -> llvm-objdump -d text"""
    import os, tempfile
    llvm = '/opt/rocm/lib/llvm/bin'
    out = []
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, 'fat.bin')
        subprocess.run([os.path.join(llvm, 'llvm-objcopy'), '--dump-section', f'.hip_fatbin={fat}', path, os.path.join(td, 'x.so')], check=True)
        blob = open(fat, 'rb').read()
        magic = b'__CLANG_OFFLOAD_BUNDLE__'
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for k, a in enumerate(starts):
            b = starts[k + 1] if k + 1 < len(starts) else len(blob)
            one, co = os.path.join(td, f'b{k}.bin'), os.path.join(td, f'b{k}.co')
            open(one, 'wb').write(blob[a:b])
            subprocess.run([os.path.join(llvm, 'clang-offload-bundler'), '--unbundle', '--type=o', f'--input={one}',
                            '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--output={co}'], check=True)
            if os.path.getsize(co):
                out.append(subprocess.run([os.path.join(llvm, 'llvm-objdump'), '-d', '--no-show-raw-insn', co], capture_output=True,
                                          text=True, check=True).stdout)
    return '\n'.join(out)


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    window = WINDOW
    show = 3
    for k, a in enumerate(sys.argv):
        if a == '--window': window = int(sys.argv[k + 1])
        if a == '--show': show = int(sys.argv[k + 1])
    args = [a for a in args if not a.isdigit()]
    path = args[0]
    text = disassemble_so(path) if path.endswith('.so') else open(path).read()
    res = check_text(text, args[1] if len(args) > 1 else '', window)
    bad = 0
    for name, r in sorted(res.items()):
        nv = len(r['violations'])
        bad += nv
        print(f'{demangle(name)[:110]:110s} mfma {r["mfma"]:5d}  VALU overwrites of A/B within {window} slots: {nv:4d}, of VALU-built operands: '
              f'{len(r["valu_built"]):4d}   (async reloads: {r["asynchronous"]})')
        for v in (r['valu_built'] + r['violations'])[:show]:
            print(describe(r['insns'], v, context=14))
    sys.exit(1 if bad else 0)
