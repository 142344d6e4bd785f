"""Check (CPU, on a `hipcc -S` listing): the vector-memory instructions of every GEMM epilogue form, path by path.

The ping-pong GEMM (csrc/gemm256_kernel.h) leaves the previous tile's epilogue stores IN FLIGHT while the next tile's first K-tile is consumed: the wait at
the head of the K loop is `s_waitcnt vmcnt(L + NSTx)` with NSTx = the number of store instructions the epilogue issued behind the next tile's prologue DMA
(`vmcnt` is one in-order counter).  If a toolchain merged, split or predicated those stores the count would be off and the first K-tile would read LDS
that has not landed -- silently.  Every epilogue form brackets its body with `; UFV_EPI_BEGIN <form>` / `; UFV_EPI_END <form>` comment lines (UFV_EPI_MARK);
this walks the control-flow graph of each kernel between a BEGIN and the ENDs it reaches and returns, per region, the (min, max) number of vector-memory
STORES and LOADS over all paths.

expected (MT = MA0 + MA1 accumulator rows, NT = 2 + NB1 n-tiles, from the kernel's template arguments in its mangled name):
  wide         stores == 2 MT on EVERY path (NSTW), no loads
  swiglu_wide  stores == MT on every path (NSTS), no loads
  wide_mx      stores == 2 MT + MT / 4 on every path (NSTM: e4m3 codes per (row, run) + the scale bytes of four rows per store), no loads
  swiglu_mx    stores == MT + MT / 4 on every path, no loads
  rope         stores == 2 MT on every path (NSTW), no loads (the cos / sin rows are fetched before the next tile's prologue DMA, outside the region)
  resid        stores == MT NT on every path (NST), loads == MT NT (all retired by counted waits inside the form)
  plain        stores <= NST = (SWIGLU ? 2 MT : MT NT) with equality on the all-in-range path (the form relaxes on interior tiles only)
usage: python tools/isa_epilogue_stores.py file.s"""
import re
import sys

VM_STORE = re.compile(r'^(buffer_store|global_store|flat_store|scratch_store)')
VM_LOAD = re.compile(r'^(buffer_load|global_load|flat_load|scratch_load)')
KERNEL = re.compile(r'gemm_nt_256ILb(\d)ELb(\d)ELb(\d)ELb(\d)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELb(\d)ELi(\d)ELb(\d)E')


def kernels(listing):
    """-> [(mangled name, template args dict, body lines)]"""
    lines = listing.split('\n')
    starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w*gemm_nt_256\w*:', l)]
    out = []
    for k, (i, name) in enumerate(starts):
        body = lines[i + 1:]
        for j, l in enumerate(body):
            if l.startswith('.Lfunc_end') or l.strip().startswith('.end_amdhsa_kernel'):      # (code may follow an early s_endpgm)
                body = body[:j + 1]
                break
        m = KERNEL.search(name)
        t = dict(zip(("OUT_F32", "SWIGLU", "FP8", "SKT", "MA0", "MA1", "NB1", "PH2", "KSPL", "ROPE", "MX", "HALF"), (int(x) for x in m.groups())))
        out.append((name, t, body))
    return out


def blocks(body):
    """basic blocks: [(label, [instruction | marker strings], [successor labels], falls_through)]; markers are kept as 'MARK BEGIN wide' pseudo-instructions"""
    bl, cur, label = [], [], "entry"
    for l in body:
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            bl.append([label, cur]); label, cur = m.group(1), []
            continue
        x = l.strip()
        mk = re.match(r'^;\s*UFV_EPI_(BEGIN|END)\s+(\w+)', x)
        if mk:
            cur.append(f"MARK {mk.group(1)} {mk.group(2)}")
            continue
        x = x.split(';')[0].strip()
        if not x or x.startswith('.') or x.endswith(':'):
            continue
        cur.append(x)
    bl.append([label, cur])
    return bl


def region_counts(body):
    """-> [(form, (min_stores, max_stores), (min_loads, max_loads))] for every BEGIN marker of the kernel body"""
    bl = blocks(body)
    index = {b[0]: i for i, b in enumerate(bl)}
    # flatten to instruction positions (block, offset) and walk: from a BEGIN, follow fall-through and branches until an END
    sys.setrecursionlimit(100000)
    results = []

    def succ(bi, oi):
        """positions that follow instruction (bi, oi)"""
        ins = bl[bi][1][oi]
        nxt = (bi, oi + 1) if oi + 1 < len(bl[bi][1]) else first_of(bi + 1)
        if ins.startswith('s_branch'):
            return [first_of(index[ins.split()[1]])]
        if ins.startswith('s_cbranch'):
            return [p for p in (first_of(index[ins.split()[1]]), nxt) if p is not None]
        if ins.startswith('s_setpc'):                  # a long branch: s_getpc / s_add_u32 sN, sN, (.LBBx_y-.Lpost_getpcK)&4294967295 / s_addc_u32 / s_setpc
            for back in range(oi - 1, max(oi - 8, -1), -1):
                m = re.search(r'\((\.LBB\d+_\d+)-\.Lpost_getpc', bl[bi][1][back])
                if m:
                    return [first_of(index[m.group(1)])]
            raise RuntimeError(f"s_setpc without a recognisable target in {bl[bi][0]}")
        if ins.startswith('s_endpgm') or ins.startswith('s_swappc'):
            return []
        return [nxt] if nxt is not None else []

    def first_of(bi):
        while bi < len(bl) and not bl[bi][1]:
            bi += 1
        return (bi, 0) if bi < len(bl) else None

    for bi, (label, inss) in enumerate(bl):
        for oi, ins in enumerate(inss):
            if not ins.startswith("MARK BEGIN"):
                continue
            form = ins.split()[2]
            memo, onstack = {}, set()

            def walk(pos):
                """(min stores, max stores, min loads, max loads) from pos to the region's END; None = a path that never reaches an END"""
                if pos in memo:
                    return memo[pos]
                if pos in onstack:
                    raise RuntimeError(f"cycle inside epilogue region {form} at {bl[pos[0]][0]}")
                b, o = pos
                x = bl[b][1][o]
                if x.startswith("MARK END"):
                    if x.split()[2] != form:
                        raise RuntimeError(f"region {form} runs into END {x.split()[2]}")
                    memo[pos] = (0, 0, 0, 0)
                    return memo[pos]
                if x.startswith("MARK BEGIN") and pos != (bi, oi):
                    raise RuntimeError(f"region {form} runs into another BEGIN")
                st = 1 if VM_STORE.match(x) else 0
                ld = 1 if VM_LOAD.match(x) else 0
                onstack.add(pos)
                subs = [walk(p) for p in succ(b, o)]
                onstack.discard(pos)
                subs = [s for s in subs if s is not None]
                if not subs:
                    memo[pos] = None
                    return None
                r = (st + min(s[0] for s in subs), st + max(s[1] for s in subs), ld + min(s[2] for s in subs), ld + max(s[3] for s in subs))
                memo[pos] = r
                return r
            r = walk((bi, oi))
            if r is None:
                raise RuntimeError(f"region {form} never reaches its END")
            results.append((form, (r[0], r[1]), (r[2], r[3])))
    return results


def expected(form, t):
    MT, NT = t["MA0"] + t["MA1"], 2 + t["NB1"]
    if form == "wide":
        return dict(stores=(2 * MT, 2 * MT), loads=(0, 0))
    if form == "swiglu_wide":
        return dict(stores=(MT, MT), loads=(0, 0))
    if form == "wide_mx":                          # MX-emitting epilogue: a code store per (row, run) + one scale store per four rows (NSTM)
        return dict(stores=(2 * MT + MT // 4, 2 * MT + MT // 4), loads=(0, 0))
    if form == "swiglu_mx":
        return dict(stores=(MT + MT // 4, MT + MT // 4), loads=(0, 0))
    if form == "rope":                             # the fused QKV + RoPE + KV-append epilogue: (MT / 2 row pairs) x (2 heads) x (2 n-tiles) = 2 MT = NSTW
        return dict(stores=(2 * MT, 2 * MT), loads=(0, 0))
    if form == "resid":
        return dict(stores=(MT * NT, MT * NT), loads=(MT * NT, MT * NT))
    if form == "plain":
        nst = 2 * MT if t["SWIGLU"] else MT * NT
        return dict(stores=(None, nst), loads=None)
    raise KeyError(form)


def relax_ok(t):
    """RELAX_OK of gemm256_kernel.h: the kernels whose K loop may leave epilogue stores in flight (the others wait strictly: their counts do not matter)"""
    MT, NT = t["MA0"] + t["MA1"], 2 + t["NB1"]
    la0, la1 = (2 if t["MA0"] > 2 else 1), (2 if t["MA1"] > 2 else 1)
    l_all = la0 + la1 + 2 + t["NB1"]
    if t.get("MX", 0) & 2:
        return bool(t["PH2"]) and l_all + (MT if t["SWIGLU"] else 2 * MT) + MT // 4 <= 63
    nst = 2 * MT if t["SWIGLU"] else MT * NT
    return bool(t["PH2"]) and not t["SKT"] and not t["KSPL"] and l_all + nst <= 63


def check(listing):
    """-> (list of violations, number of regions checked, forms seen); kernels that never relax a wait are skipped"""
    bad, n, forms = [], 0, set()
    for name, t, body in kernels(listing):
        if not relax_ok(t):
            continue
        for form, st, ld in region_counts(body):
            e = expected(form, t)
            n += 1
            forms.add(form)
            lo, hi = e["stores"]
            if (lo is not None and st[0] != lo) or st[1] != hi:
                bad.append((name, form, "stores", st, e["stores"]))
            if e["loads"] is not None and ld != e["loads"]:
                bad.append((name, form, "loads", ld, e["loads"]))
    return bad, n, forms


if __name__ == "__main__":
    bad, n, forms = check(open(sys.argv[1]).read())
    print(f"{n} epilogue regions checked ({sorted(forms)}), {len(bad)} violations")
    for b in bad[:20]:
        print("  ", b)
    sys.exit(1 if bad else 0)
