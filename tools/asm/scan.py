"""Static scan of one kernel family in a hipcc -save-temps .s file: per kernel, MFMA count, barriers, and what sits INSIDE the main loop
(first .. last MFMA): compiler-placed `s_waitcnt vmcnt` (outside ;;#ASMSTART blocks), scratch traffic (spills).
    python tools/asm/scan.py <file.s> [name filter]"""
import re, sys, subprocess

def kernels(path):
    s = open(path).read()
    for f in re.split(r'\n\t\.globl\t', s)[1:]:
        name = f.split('\n', 1)[0].split()[0]
        body = f.split('\n')
        end = next((i for i, l in enumerate(body) if l.startswith('.Lfunc_end')), len(body))
        yield name, body[:end]

def demangle(n):
    return subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()

if __name__ == '__main__':
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    for name, body in kernels(sys.argv[1]):
        dn = demangle(name)
        if flt not in dn:
            continue
        mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
        if not mf:
            continue
        inasm, vm, sc = False, [], []
        for i, l in enumerate(body):
            if 'ASMSTART' in l: inasm = True
            elif 'ASMEND' in l: inasm = False
            if mf[0] <= i <= mf[-1]:
                if not inasm and 's_waitcnt' in l and 'vmcnt' in l: vm.append((i, l.strip()))
                if 'scratch_' in l: sc.append(i)
        bars = sum(1 for l in body if 's_barrier' in l)
        print(f'{dn[:90]:90s} mfma {len(mf):4d} barriers {bars:2d} | in main loop: compiler vmcnt waits {len(vm)} {[v[1] for v in vm[:3]]} scratch ops {len(sc)}')
