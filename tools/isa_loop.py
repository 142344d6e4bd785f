"""Diagnostic: condensed view of a kernel's ISA (run-length encoded key instructions)."""
import re, sys
src, name = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(name) and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
mf = [i for i, l in enumerate(body) if "v_mfma" in l]
lo, hi = max(0, mf[0] - 40), min(len(body), mf[-1] + 30)
pat = re.compile(r"ds_read|ds_write|v_mfma|s_waitcnt|s_barrier|global_load|s_nop|s_cbranch|^\.LBB|scratch|v_exp|s_setprio")
prev, cnt, arg = None, 0, ""
for l in body[lo:hi]:
    t = l.strip()
    if not pat.search(t): continue
    k = t.split()[0]
    a = " ".join(t.split()[1:4]) if k in ("s_waitcnt", "s_nop", "s_cbranch_scc1", "s_cbranch_vccz", "s_cbranch_vccnz", "s_cbranch_scc0") or k.startswith(".LBB") else ""
    key = (k, a)
    if key == prev: cnt += 1
    else:
        if prev: print(f"{cnt:3d} x {prev[0]} {prev[1]}")
        prev, cnt = key, 1
if prev: print(f"{cnt:3d} x {prev[0]} {prev[1]}")
print("total lines", len(body), "mfma", len(mf))
