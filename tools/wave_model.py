"""A one-wave, in-order issue model of a gfx950 kernel's ISA (hipcc -S output): how long does ONE wave alone on its SIMD need
for a stretch of straight-line code?  Used to compare schedules of the K5-K7 backward tile (one wave per SIMD: nothing hides a
wait).  Model: 4-clock issue cadence; the matrix pipe runs one MFMA at a time (16x16x4 f32: 32 clocks, 4x4x1: 8); a VALU / DS /
VMEM instruction that reads a register written by an MFMA waits for it; DS results arrive LDS_LAT clocks after issue, in order
(s_waitcnt lgkmcnt(n)); global loads arrive VM_LAT clocks after issue, in order (vmcnt(n)); stores never block.
    python tools/wave_model.py file.s <mangled kernel name substring> [first_label last_label]"""
import re
import sys

LDS_LAT = {'b32': 64, 'b64': 80, 'b128': 128, 'other': 64}
VM_LAT = 900
REG = re.compile(r'\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b')


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def model(lines):
    t, mf = 0, 0
    ready = {}                     # register -> clock its value is available (MFMA results only)
    lgkm, vm = [], []              # completion clocks of outstanding DS / VMEM-load ops, in issue order
    stall_lgkm = stall_vm = stall_mfma_dep = n_mfma = 0
    for l in lines:
        op, _, rest = l.partition(' ')
        ops = [x.strip() for x in rest.split(',')] if rest else []
        t += 4
        if op.startswith('s_waitcnt'):
            for kind, q in (('lgkmcnt', lgkm), ('vmcnt', vm)):
                m = re.search(kind + r'\((\d+)\)', l)
                if m:
                    n = int(m.group(1))
                    done = q[:len(q) - n] if n else q[:]
                    if done:
                        w = max(done)
                        if w > t:
                            if kind == 'lgkmcnt': stall_lgkm += w - t
                            else: stall_vm += w - t
                            t = w
                    del q[:len(q) - n if n else len(q)]
            continue
        if op == 's_nop':
            t += 4 * int(ops[0]) if ops else 0
            continue
        if op.startswith('v_mfma'):
            n_mfma += 1
            dur = 8 if '4x4x1' in op else 32
            src = set().union(*(regs(x) for x in ops[1:])) if len(ops) > 1 else set()
            dep = max([ready.get(r, 0) for r in src] + [0])
            start = max(t, mf, dep)
            if dep > max(t, mf): stall_mfma_dep += dep - max(t, mf)
            mf = start + dur
            for r in regs(ops[0]): ready[r] = start + dur + 8
            t = start
            continue
        src = set().union(*(regs(x) for x in ops[1:])) if len(ops) > 1 else set()
        if op.startswith(('ds_write', 'global_store', 'global_atomic', 'buffer_store')):
            src |= regs(ops[0]) if ops else set()
        dep = max([ready.get(r, 0) for r in src] + [0])
        if dep > t:
            stall_mfma_dep += dep - t
            t = dep
        if op.startswith('ds_'):
            kind = 'b128' if 'b128' in op else 'b64' if ('b64' in op or 'read2' in op) else 'b32'
            lgkm.append(t + LDS_LAT[kind])
        elif op.startswith(('global_load', 'buffer_load')):
            vm.append(t + VM_LAT)
        elif op.startswith('s_load'):
            lgkm.append(t + 200)
        if ops and not op.startswith(('ds_write', 'global_store', 'global_atomic', 's_')):
            for r in regs(ops[0]): ready.pop(r, None)
    return dict(clocks=t, mfma=n_mfma, stall_lgkm=stall_lgkm, stall_vm=stall_vm, stall_mfma_dep=stall_mfma_dep, instr=len(lines))


def body(path, name, first=None, last=None):
    s = open(path).read()
    a = s.index(name)
    a = s.index('\n', s.index(':', a))
    b = s.index('.Lfunc_end', a)
    out, on = [], first is None
    for l in s[a:b].split('\n'):
        l = l.split(';')[0].strip()
        if not l or l.startswith('.') and not l.endswith(':'):
            continue
        if l.endswith(':'):
            if first and l[:-1] == first: on = True
            if last and l[:-1] == last: break
            continue
        if on:
            out.append(l)
    return out


if __name__ == '__main__':
    ls = body(*sys.argv[1:])
    print(model(ls))
