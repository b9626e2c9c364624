"""Static check: no inline-asm instruction reads (or overwrites) a VGPR that a matrix-core instruction has just written.

On CDNA the result of a v_mfma_* is not interlocked: software must leave wait states between the MFMA and the first instruction that
reads (or writes) its destination registers.  hipcc's hazard recogniser inserts the s_nop for every instruction it can see -- and it
cannot see inside inline asm.  The point kernels use inline asm for LDS reads, counted waits and a few lane operations; if the scheduler
ever makes one of those the FIRST toucher of a fresh accumulator, it reads stale registers (this happened with an inline-asm v_max_f32
ReLU: 3 % errors that came and went with unrelated code changes).  This tool scans the device assembly of the kernels in program order
and reports every instruction between ;;#ASMSTART and ;;#ASMEND that names a register written by an MFMA fewer than WAIT wait states
earlier (s_nop N counts N + 1, every other instruction 1; 8-pass 32x32x16 MFMAs need 11, the check asks for 12).

    python tools/mfma_hazard_check.py /tmp/k1.s dpn_fwd_kernel dpn_bwd_kernel      # flags of deepphysinet_amd/build.py UNITS, see lds_hazard_check.py
"""
import re
import sys

REG = re.compile(r'\b([va])(?:(\d+)|\[(\d+):(\d+)\])')
WAIT = 12


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        kind = m.group(1)
        if m.group(2) is not None:
            out.add((kind, int(m.group(2))))
        else:
            out.update((kind, i) for i in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


def check(lines, name):
    fresh = {}                       # register -> wait states since the MFMA that wrote it
    hazards, n_asm, in_asm = 0, 0, False
    for no, raw in lines:
        if ';;#ASMSTART' in raw:
            in_asm = True
            continue
        if ';;#ASMEND' in raw:
            in_asm = False
            continue
        ins = raw.split(';')[0].strip()
        if not ins or ins.endswith(':') or ins.startswith('.'):
            continue
        op = ins.split()[0]
        body = ins[len(op):]
        if in_asm:
            n_asm += 1
            bad = sorted(r for r in regs_of(body) if r in fresh and fresh[r] < WAIT)
            if bad:
                hazards += 1
                print(f'{name}: line {no}: inline asm `{ins}` touches {bad[:4]} {fresh[bad[0]]} wait states after the MFMA that wrote it')
        step = int(ins.split()[1]) + 1 if op == 's_nop' else 1
        for r in list(fresh):
            fresh[r] += step
            if fresh[r] >= WAIT:
                del fresh[r]
        if op.startswith('v_mfma') or op.startswith('v_smfmac'):
            for r in regs_of(body.split(',')[0]):
                fresh[r] = 0
    print(f'{name}: {n_asm} inline-asm instructions, {hazards} of them touch a fresh MFMA result')
    return hazards


def main():
    path, wanted = sys.argv[1], sys.argv[2:]
    text = open(path).read().split('\n')
    starts = [(i, l.split(':')[0]) for i, l in enumerate(text) if re.match(r'^_Z\w+:', l)]
    total = 0
    for i, sym in starts:
        if wanted and not any(w in sym for w in wanted):
            continue
        end = next((j for j in range(i, len(text)) if text[j].startswith('.Lfunc_end')), len(text))
        total += check([(j + 1, text[j]) for j in range(i + 1, end)], sym)
    sys.exit(1 if total else 0)


if __name__ == '__main__':
    main()
