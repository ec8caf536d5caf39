"""Where one training step's wall time goes, from a rocprofv3 kernel trace of bench.py (csv): busy time per queue, time the GPU
runs nothing at all, gaps between consecutive kernels of the main queue.   python scripts/step_timeline.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows)
# the last full training step: between the last two launches of the stem conversion kernel
marks = [i for i, e in enumerate(ev) if e[2].startswith("ncdhw_to_ndhwc")]
lo, hi = marks[-3], marks[-2]
win = ev[lo:hi]
t0, t1 = win[0][0], ev[hi][0]
span = t1 - t0
# union of busy intervals (any queue)
cur_e, busy = t0, 0
for s, e, _, _ in sorted(win):
    s = max(s, cur_e)
    if e > s:
        busy += e - s
        cur_e = e
byq = collections.defaultdict(list)
for e in win:
    byq[e[3]].append(e)
print(f"step span {span/1e6:.2f} ms, GPU busy (any queue) {busy/1e6:.2f} ms, idle {100*(span-busy)/span:.1f} %, {len(win)} kernels")
for q, lst in byq.items():
    b = sum(e[1] - e[0] for e in lst)
    gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
    small = [g for g in gaps if 0 < g < 20000]
    print(f"  queue {q}: {len(lst)} kernels, busy {b/1e6:.2f} ms; gaps < 20 us between consecutive kernels: {len(small)} totalling {sum(small)/1e6:.2f} ms "
          f"(median {sorted(small)[len(small)//2]/1e3 if small else 0:.1f} us)")
# biggest idle stretches of the whole GPU
idle = []
cur_e = t0
for s, e, n, _ in sorted(win):
    if s > cur_e:
        idle.append((s - cur_e, n))
    cur_e = max(cur_e, e)
idle.sort(reverse=True)
print("  longest idle stretches (us, next kernel):", [(round(g / 1e3, 1), n[:40]) for g, n in idle[:6]])
# waits of the main queue (gaps >= 20 us between its consecutive kernels): where it stands still for the side queue
mainq = max(byq, key=lambda q: len(byq[q]))
lst = byq[mainq]
waits = [(lst[i + 1][0] - lst[i][1], lst[i][2][:36], lst[i + 1][2][:36]) for i in range(len(lst) - 1) if lst[i + 1][0] - lst[i][1] >= 20000]
print(f"  main queue waits >= 20 us: {len(waits)} totalling {sum(w[0] for w in waits)/1e6:.2f} ms:", [(round(w[0] / 1e3), w[1], w[2]) for w in waits])
# per phase: forward = up to the NT-Xent kernel, backward after it
nt = [i for i, e in enumerate(win) if e[2].startswith("ntxent_fwd")]
if nt:
    tf = win[nt[0]][0]
    print(f"  forward {(tf - t0)/1e6:.2f} ms, backward + optimizer {(t1 - tf)/1e6:.2f} ms")
