"""Check (CPU, on a `hipcc -S` listing): the decode GEMV's hand-issued loads.  Between an asm `global_load_dwordx4 vD, ...` and a wait that covers it nothing
may read or write vD -- the register allocator copying such a register (seen with the 4-output forms of the kernel) reads stale data and races the load.
Model: loads return in order; `s_waitcnt vmcnt(N)` leaves the N youngest outstanding.  Only the hand-issued loads are tracked (the compiler's own loads in
between can only make a wait cover MORE than the model assumes: the check errs on the side of reporting).  Each kernel body is scanned twice in a row, so the
state at a loop's back edge meets the loop's first instructions.
usage: python tools/isa_inflight_check.py file.s gemv1_nt   -> exit status 1 and the offending lines if any"""
import re, sys


def regs_of(tok):
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', tok)
    return {int(m.group(1))} if m else set()


def check(listing, pattern):
    bad, kernels = [], 0
    for name in [m for m in re.findall(r'^(\S+):\s', listing, re.M) if pattern in m and not m.startswith('.')]:
        body = listing[listing.index(name + ':'):]
        body = body[:body.index('.end_amdhsa_kernel')].split('\n')
        hand = [i for i, l in enumerate(body) if l.strip().startswith('global_load_dword') and i > 0 and 'ASMSTART' in body[i - 1]]
        if not hand:
            continue
        kernels += 1
        fifo, in_asm = [], False                       # outstanding hand-issued loads, oldest first: sets of destination registers
        for rep in range(2):
            for n, l in enumerate(body[hand[0] - 1:], hand[0] - 1):
                x = l.strip()
                if x.startswith(';;#ASMSTART'):
                    in_asm = True; continue
                if x.startswith(';;#ASMEND'):
                    in_asm = False; continue
                x = x.split(';')[0].strip()
                if not x or x.startswith('.') or x.endswith(':'):
                    continue
                toks = re.findall(r'v\[\d+:\d+\]|v\d+', x)
                touched = set().union(*[regs_of(t) for t in toks]) if toks else set()
                flying = set().union(*fifo) if fifo else set()
                if x.startswith('s_waitcnt') and 'vmcnt' in x:
                    cnt = int(re.search(r'vmcnt\((\d+)\)', x).group(1))
                    fifo = fifo[len(fifo) - cnt:] if cnt < len(fifo) else fifo
                    if cnt == 0:
                        fifo = []
                    continue
                if in_asm and x.startswith('global_load_dword'):
                    addr = set().union(*[regs_of(t) for t in toks[1:]])
                    if addr & flying:
                        bad.append((name, n, x))
                    fifo.append(regs_of(toks[0]))
                    continue
                if touched & flying:
                    bad.append((name, n, x))
    return bad, kernels


if __name__ == "__main__":
    bad, kernels = check(open(sys.argv[1]).read(), sys.argv[2])
    print(f"{kernels} kernels with hand-issued loads checked, {len(bad)} suspicious lines")
    for b in bad[:20]:
        print(b)
    sys.exit(1 if bad else 0)
