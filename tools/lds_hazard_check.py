"""Static check of the inline-asm LDS reads in the point kernels.

The forward / backward / weight-gradient kernels read their LDS operands with
inline-asm `ds_read_b128` and counted `s_waitcnt lgkmcnt(N)` (hipcc would drain
every LDS-DMA in flight before a read it can see).  The compiler does not know
those destination registers are pending, so nothing but the source order stops
it from touching one (a copy to an AGPR, a spill) before the wait that covers
it.  This tool scans the device assembly of a kernel in program order, models
the LGKM counter (LDS operations retire in order) and reports every instruction
that names a register whose ds_read has not been waited for.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -Iinclude -DDPN_TU=1 -mllvm -amdgpu-mfma-vgpr-form \
        deepphysinet_amd/csrc/dpn_kernels.hip -o /tmp/k1.s          # the flags of deepphysinet_amd/build.py UNITS
    python tools/lds_hazard_check.py /tmp/k1.s dpn_fwd_kernel dpn_bwd_kernel
    (DPN_TU=2 without the -mllvm flag for dpn_wgrad_kernel; tests/test_capi_cpu.py runs both)

Control flow is ignored (the scan is linear), which is exact for the unrolled
pipelines of these kernels: no inline-asm ds_read is pending across a branch.
Only the reads inside ;;#ASMSTART ... ;;#ASMEND are tracked by register: an LDS
operation the compiler emitted itself (ds_bpermute of a shuffle, a visible
ds_read) takes a slot of the counter, and its destination is the compiler's to
wait for -- in the branchy epilogues a linear scan would pair such a read with
code of another path.
"""
import re
import sys

REG = re.compile(r'\b([va])(?:(\d+)|\[(\d+):(\d+)\])')
LGKM = re.compile(r'^(ds_|s_load|s_buffer_load|s_sendmsg|s_memtime|s_memrealtime)')
WAIT = re.compile(r'lgkmcnt\((\d+)\)')


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
    pending = []                     # one entry per outstanding LGKM operation: set of destination registers (empty for writes)
    hazards = 0
    n_reads = 0
    max_pending = 0
    in_asm = False
    for no, raw in lines:
        if ';;#ASMSTART' in raw:
            in_asm = True
        elif ';;#ASMEND' in raw:
            in_asm = False
        ins = raw.split(';')[0].strip()
        if not ins or ins.endswith(':') or ins.startswith('.'):
            continue
        op = ins.split()[0]
        touched = regs_of(ins[len(op):])
        live = set().union(*pending) if pending else set()
        bad = touched & live
        if bad:
            hazards += 1
            print(f'{name}: line {no}: `{ins}` touches pending {sorted(bad)[:4]}')
        if op == 's_waitcnt':
            m = WAIT.search(ins)
            if m:
                n = int(m.group(1))
                pending = pending[len(pending) - n:] if n else []
            elif 'lgkmcnt' not in ins and re.fullmatch(r's_waitcnt\s+(0x[0-9a-f]+|\d+)', ins):
                pending = []         # raw immediate: treat as a full wait only if the lgkm field is 0 (not emitted by this code base)
        elif LGKM.match(op):
            if op.startswith('s_load') or op.startswith('s_buffer_load'):
                # scalar loads share the counter and return OUT of order: a counted lgkmcnt(N) wait behind one is no longer a statement
                # about the LDS reads in front of it
                if any(pending):
                    hazards += 1
                    print(f'{name}: line {no}: `{ins}` is issued with inline-asm LDS reads outstanding (counted waits assume in-order returns)')
            dst = set()
            if in_asm and op.startswith('ds_read'):
                first = ins[len(op):].split(',')[0]
                dst = regs_of(first)
                n_reads += 1
            pending.append(dst)
            max_pending = max(max_pending, len(pending))
    print(f'{name}: {n_reads} ds_read, at most {max_pending} LGKM operations in flight (the counter holds 15: issue stalls beyond), {hazards} hazards')
    return hazards


def main():
    path, wanted = sys.argv[1], sys.argv[2:]
    text = open(path).read().split('\n')
    starts = [(i, l.split(':')[0]) for i, l in enumerate(text) if re.match(r'^_Z\w+:', l)]
    total = 0
    for idx, (i, sym) in enumerate(starts):
        if wanted and not any(w in sym for w in wanted):
            continue
        end = next((j for j in range(i, len(text)) if text[j].strip().startswith('s_endpgm')), len(text))
        # kernels with several exits: run to the .Lfunc_end label
        end = next((j for j in range(i, len(text)) if text[j].startswith('.Lfunc_end')), end)
        total += check([(j + 1, text[j]) for j in range(i + 1, end)], sym)
    sys.exit(1 if total else 0)


if __name__ == '__main__':
    main()
