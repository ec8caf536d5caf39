"""idle time between consecutive kernels of a rocprofv3 kernel trace (csv): python gap_analysis.py <kernel_trace.csv> [name-filter]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# the last 20 Lloyd iterations of the fit: from the 20th-from-last km_assign_dma launch to the km_status that follows the last one
ia = [i for i, e in enumerate(ev) if e[2].startswith("void km_assign_dma")]
ist = [i for i, e in enumerate(ev) if e[2].startswith("km_status")]
hi = ist[-1]                                            # the last full iteration ends with its km_status
ia = [i for i in ia if i < hi]
lo = ia[-20] - 1
win = ev[lo + 1: hi + 1]
busy = sum(e[1] - e[0] for e in win)
span = win[-1][1] - win[0][0]
gaps, dur = {}, {}
prev = None
for e in win:
    if prev is not None:
        gaps.setdefault(e[2][:44], []).append(e[0] - prev[1])
    dur.setdefault(e[2][:44], []).append(e[1] - e[0])
    prev = e
print(f"20 iterations: span {span/20e3:.1f} us/iter, busy {busy/20e3:.1f} us/iter, idle {(span-busy)/20e3:.1f} us/iter, {len(win)/20:.1f} kernels/iter")
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    g = gaps.get(k, [0])
    print(f"  {k:46s} n={len(dur[k]):4d} dur {sum(dur[k])/20e3:7.2f} us/iter   gap before: {sum(g)/20e3:6.2f} us/iter")
