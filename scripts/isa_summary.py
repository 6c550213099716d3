"""Developer aid: run-length summary of the instruction kinds of one kernel in a hipcc -save-temps .s file.
usage: isa_summary.py file.s kernel-name-substring [first-label [last-label]]"""
import re, sys
s = open(sys.argv[1]).read()
i = s.index("\n" + [l for l in s.split("\n") if sys.argv[2] in l and l.split(";")[0].strip().endswith(":")][0])
k = s[i:s.index("s_endpgm", i)]
lines = k.split("\n")
def kind(l):
    l = l.strip()
    if l.startswith("v_mfma"): return "MFMA"
    if l.startswith("global_load") or l.startswith("buffer_load"): return "GLOAD"
    if l.startswith("global_store") or l.startswith("buffer_store"): return "GSTORE"
    if l.startswith("scratch_"): return "SCRATCH"
    if l.startswith("ds_read") or l.startswith("ds_load"): return "DSR"
    if l.startswith("ds_write") or l.startswith("ds_store"): return "DSW"
    if l.startswith("ds_bpermute") or l.startswith("ds_swizzle"): return "BPERM"
    if l.startswith("s_waitcnt"): return "WAIT " + l.split(None, 1)[1]
    if l.startswith("s_barrier"): return "BARRIER"
    if l.startswith("s_cbranch") or l.startswith("s_branch"): return "BR " + l.split()[-1]
    if re.match(r"^\.LBB\d+_\d+:", l): return "LABEL " + l.split(":")[0]
    if l.startswith("v_exp") or l.startswith("v_rcp") or l.startswith("v_rsq") or l.startswith("v_sqrt"): return "TRANS"
    if l.startswith("v_"): return "VALU"
    if l.startswith("s_"): return "SALU"
    return None
out, last, cnt = [], None, 0
for l in lines:
    kd = kind(l)
    if kd is None: continue
    if kd == last and not kd.startswith(("WAIT", "LABEL", "BR")): cnt += 1
    else:
        if last: out.append(f"{last} x{cnt}" if cnt > 1 else last)
        last, cnt = kd, 1
out.append(f"{last} x{cnt}")
lo = hi = None
if len(sys.argv) > 3:
    lo = next((n for n, o in enumerate(out) if o.startswith("LABEL " + sys.argv[3])), None)
if len(sys.argv) > 4:
    hi = next((n for n, o in enumerate(out) if o.startswith("LABEL " + sys.argv[4])), None)
print(" | ".join(out[lo:hi]))
