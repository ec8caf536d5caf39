"""per-kernel register / LDS / occupancy table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage):
    python scripts/kernel_resources.py video_similarity_search_amd/csrc/conv.hip [name-filter]"""
import re, subprocess, sys, os
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
d = os.path.dirname(os.path.abspath(src))
extra = ["-ffp-contract=off"] if os.path.basename(src) == "kmeans.hip" else []
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I" + os.path.join(d, "../../include"),
                      "-c", src, "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name:
        continue
    print(f"{name[:100]:100s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>4s} sgpr {r.get('TotalSGPRs','?'):>4s} spill {r.get('VGPR Spill','?'):>3s} "
          f"occ {r.get('Occupancy [waves/SIMD]','?'):>2s} lds {r.get('LDS Size [bytes/block]','?'):>6s}")
