"""Print VGPR / AGPR / scratch / LDS / occupancy of every kernel (tuning aid, not a test)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "contour-context_amd", "csrc", "cont2_amd.hip")
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-fPIC", "-c", src,
                    "-o", "/tmp/_kres.o", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark: (?:Function Name: )?(\S+)? ?\[-Rpass", line)
    m2 = re.search(r"remark: Function Name: (\S+)", line)
    if m2:
        cur = m2.group(1)
        dm = subprocess.run(["c++filt", cur], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", dm).replace("void ", "") or cur
        rows[cur] = {}
        continue
    m3 = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m3 and cur:
        rows[cur][m3.group(1).strip()] = int(m3.group(2))
for k, v in rows.items():
    print("%-44s VGPR %3d AGPR %3d SGPR %3d scratch %5d LDS %6d occupancy %d" % (
        k[:44], v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("TotalSGPRs", -1), v.get("ScratchSize", -1),
        v.get("LDS Size", -1), v.get("Occupancy", -1)))
