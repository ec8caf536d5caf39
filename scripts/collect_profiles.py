"""Post-process the rocprofv3 outputs of scripts/collect_profiles.sh into the files kept under profiles/.
usage: collect_profiles.py <scratch dir> <out dir> <round tag>"""
import csv, glob, json, os, sys

scratch, out, tag = sys.argv[1], sys.argv[2], sys.argv[3]
# round 6: the dominant kernel is the PERSISTENT form — 256 workgroups walk layer1's 3136 tile blocks — specialised on the channel count, so the
# kernel NAME tells the layers apart (the one-block kernel's grid size did): <false, 64> = layer1's four forward launches per step (training
# forward: BatchNorm statistics, no optional operand), <true, 64> = its four data-gradient launches (mask / addend / BatchNorm-backward sums)
KERNEL = "conv_wino2p_kernel<false, 64>"
KERNEL_DG = "conv_wino2p_kernel<true, 64>"
GRID = 256 * 512


def find(sub, pat):
    f = sorted(glob.glob(os.path.join(scratch, sub, "**", pat), recursive=True), key=os.path.getmtime)
    return f[-1] if f else None      # the newest: gpurun merges a new collection INTO the local scratch directory


def counter_mean(sub, name):
    f = find(sub, "*counter_collection.csv")
    per = {}
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == name and int(r.get("Grid_Size", r.get("Grid_Size_X", 0))) == GRID:
            per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    v = list(per.values())
    return sum(v) / len(v), len(v)


def trace_mean(sub):
    """average duration of the dominant kernel's FORWARD launches at the layer1 grid: per step it runs eight times at that grid, four
    forward launches then four data-gradient launches (which share the GPU with the side stream's weight gradients) — bench.py's
    roofline object times the forward ones, so the trace is cut the same way.  Returns (forward ms, n, data-gradient ms)"""
    f = find(sub, "*kernel_trace.csv")
    allr = list(csv.DictReader(open(f)))
    fwd = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in allr if KERNEL in r["Kernel_Name"] and int(r["Grid_Size_X"]) == GRID]
    dg = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in allr if KERNEL_DG in r["Kernel_Name"] and int(r["Grid_Size_X"]) == GRID]
    return sum(fwd) / len(fwd) / 1e6, len(fwd), (sum(dg) / len(dg) / 1e6 if dg else None)


def json_line(path):
    for line in open(path):
        if line.startswith("{"):
            return json.loads(line)
    return None


os.makedirs(out, exist_ok=True)
d = json_line(os.path.join(scratch, "default.log"))
json.dump(d, open(os.path.join(out, f"{tag}_bench_default_run.json"), "w"), indent=1)
p = json_line(os.path.join(scratch, "stats.log"))
json.dump(p, open(os.path.join(out, f"{tag}_bench_under_rocprof.json"), "w"), indent=1)
ks = find("stats", "*kernel_stats.csv")
open(os.path.join(out, f"{tag}_kernel_stats_bench_steps5.csv"), "w").write(open(ks).read())
fetch, nf = counter_mean("pmc_fetch", "FETCH_SIZE")
write, nw = counter_mean("pmc_write", "WRITE_SIZE")
tr, nt, tr_dg = trace_mean("stats")
sq = {}
if find("pmc_sq", "*counter_collection.csv"):
    for name in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                 "GRBM_GUI_ACTIVE"):
        try:
            sq[name] = counter_mean("pmc_sq", name)[0]
        except Exception:
            pass
res = {
    "FETCH_SIZE_KB_mean": fetch, "FETCH_SIZE_launches": nf, "WRITE_SIZE_KB_mean": write, "WRITE_SIZE_launches": nw,
    "note": f"{KERNEL} (the persistent form of the Winograd F(4,3) x F(2,3) kernel) at the layer1 shape (256 workgroups of 512 threads walk 3136 tile "
            "blocks: M=1605632 rows = 200704 tiles of 2 x 4 outputs, N=64, K=1728; B=32): the 4 FORWARD launches per step (fused BN statistics); the "
            "4 data-gradient launches are conv_wino2p_kernel<true, 64>. Separate rocprofv3 --pmc passes (FETCH_SIZE, "
            "WRITE_SIZE; each with --kernel-trace only) over `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary`. "
            "Units KiB. Per MI355X_MICROARCH.md §HBM, gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide (16 B/lane) coalesced read, so "
            "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (the pixel gather reads 16-byte pieces: the factor 2 is an upper "
            "bound there). Algorithmic minimum: 0.82 GB (forward) / 1.64 GB (data gradient with mask and z). The kernel executes 118.4 "
            "GFLOP on the matrix pipe per launch (28.9 M v_mfma_f32_32x32x2_f32) for 355.1 GFLOP of the direct form.",
    "hbm_bytes_per_launch": (2 * fetch + write) * 1024,
    "rocprof_trace_avg_ms": tr, "rocprof_trace_launches": nt, "rocprof_trace_avg_ms_dgrad_overlapped": tr_dg,
    "hip_event_avg_ms": p["roofline"]["ms_per_launch"],
    "sq_counters_per_launch": sq,
}
if sq.get("SQ_VALU_MFMA_BUSY_CYCLES") and sq.get("GRBM_GUI_ACTIVE"):
    # MFMA pipe busy cycles summed over the 1024 SIMDs / (elapsed cycles x 1024); GRBM_GUI_ACTIVE comes summed over the 8
    # XCDs, so elapsed = GUI_ACTIVE / 8.  (The busy count equals 64 cycles x the launch's 28.9 M v_mfma_f32_32x32x2_f32.)
    res["mfma_busy_fraction"] = sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (sq["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    if sq.get("SQ_WAVE_CYCLES"):
        res["wave_cycle_split"] = {k: sq[k] / sq["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in sq}
json.dump(res, open(os.path.join(out, f"{tag}_pmc_conv_wino.json"), "w"), indent=1)
for sub, name in (("stats_kmeans", "kmeans"), ("stats_rows", "rows"), ("stats_fit", "fit_cluster")):
    f = find(sub, "*kernel_stats.csv")
    if f:
        open(os.path.join(out, f"{tag}_kernel_stats_{name}.csv"), "w").write(open(f).read())
other = {}
for log in ("rows.log", "topk_k.log", "conv_shapes.log", "wino_ablations.log"):
    pth = os.path.join(scratch, log)
    if os.path.exists(pth):
        other[log[:-4]] = [l.rstrip() for l in open(pth) if l.strip() and "amdgpu.ids" not in l]
json.dump(other, open(os.path.join(out, f"{tag}_other_rows.json"), "w"), indent=1)
# round 5: plain-text rows and the one-rank distributed line (its fields make a multi-GPU run self-verifying)
for log, name in (("topk_kernels.log", "topk_kernels.txt"), ("b8_kernels.log", "small_batch_b8_kernels.txt"), ("planar_diag.log", "planar_layout_diagnostic.txt"),
                  ("persist_ab.log", "persistent_kernel_ab.txt"), ("ddp_ab.log", "ddp_one_rank_ab.txt"), ("ddp_seq.log", "ddp_one_rank_step_kernels.txt")):
    pth = os.path.join(scratch, log)
    if os.path.exists(pth):
        open(os.path.join(out, f"{tag}_{name}"), "w").write("".join(l for l in open(pth) if "amdgpu.ids" not in l and "warning" not in l))
fd = os.path.join(scratch, "force_dist.log")
if os.path.exists(fd):
    j = json_line(fd)
    if j:
        one = os.path.join(scratch, "force_dist_oneshot.json")
        if os.path.exists(one):
            for line in open(one):
                if line.startswith("{"):
                    j["kmeans_oneshot_row_from_stderr"] = json.loads(line)
        json.dump(j, open(os.path.join(out, f"{tag}_bench_force_dist_one_rank.json"), "w"), indent=1)
print(json.dumps({k: res[k] for k in ("hbm_bytes_per_launch", "rocprof_trace_avg_ms", "hip_event_avg_ms")}), d["value"], p["value"])
