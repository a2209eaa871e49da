"""Condense rocprofv3 output into the files kept under profiles/ (run on the GPU box, see README.md).

  python profiles/summarize.py <round-tag> <out-dir> <kernel_stats.csv> <kernel_trace.csv> <pmc_fetch.csv> <pmc_write.csv> <batch>
                               [<db_scans> <workload> <git_head>]

Writes <tag>_kernel_stats_cc_kernels.csv (rows of this repo's kernels), <tag>_kernel_trace_timed_launches.csv
(per-kernel mean duration over the launches whose grid equals the timed <batch>-scan step), the per-dispatch
FETCH_SIZE / WRITE_SIZE rows of this repo's kernels and <tag>_pmc_summary.json (HBM bytes per launch of the
<batch>-scan launches; FETCH_SIZE x2 for the wide float4 stream of cc_k_rasterize, MI355X_MICROARCH.md HBM section).
"""
import collections
import csv
import json
import os
import sys

tag, out, f_stats, f_trace, f_fetch, f_write, batch = sys.argv[1:8]
batch = int(batch)
db_scans = int(sys.argv[8]) if len(sys.argv) > 8 else None
workload = sys.argv[9] if len(sys.argv) > 9 else "sparse"
git_head = sys.argv[10] if len(sys.argv) > 10 else None
os.makedirs(out, exist_ok=True)


def short(name):
    n = name.split("(")[0]
    if n.startswith("void "):
        n = n[5:]
    return n.split("<")[0]


# The launches of a timed step are the LARGEST launches of each kernel (the DB build ingests 128-scan chunks; a step
# ingests <batch> scans at once and queries them in chunks of <= 512): they are picked by grid size.
_max_grid = {}


def note_grid(k, g):
    _max_grid[k] = max(_max_grid.get(k, 0), g)


def step_grid(k):
    return _max_grid.get(k)


for fn in (f_trace, f_fetch, f_write):
    if os.path.exists(fn):
        for r in csv.DictReader(open(fn)):
            k = short(r["Kernel_Name"])
            if k.startswith("cc_k_"):
                note_grid(k, int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]))

if os.path.exists(f_stats):
    rows = [r for r in csv.DictReader(open(f_stats)) if short(r["Name"]).startswith("cc_k_")]
    with open(os.path.join(out, tag + "_kernel_stats_cc_kernels.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows:
            r["Name"] = short(r["Name"])
            w.writerow(r)

if os.path.exists(f_trace):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f_trace)):
        k = short(r["Kernel_Name"])
        if not k.startswith("cc_k_"):
            continue
        g = step_grid(k)
        gs = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"])
        if g is None or gs == g or k in ("cc_k_ksort_new", "cc_k_ksort_merge", "cc_k_ksort_act", "cc_k_extract"):
            acc[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(os.path.join(out, tag + "_kernel_trace_timed_launches.csv"), "w") as f:
        f.write("kernel,launches,mean_us,min_us,max_us\n")
        for k, v in sorted(acc.items()):
            f.write("%s,%d,%.1f,%.1f,%.1f\n" % (k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))

summ = {"note": "rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected in separate runs); counter unit = KB. FETCH_SIZE on gfx950 "
                "reports half of the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM section): fetch_bytes_corrected "
                "doubles it for cc_k_rasterize (16 B/lane float4 stream); other kernels' access widths are uncalibrated: raw value given.",
        "batch_scans": batch, "db_scans": db_scans, "workload": workload, "git_head": git_head, "kernels": {}}
for fn, key, oname in ((f_fetch, "fetch_bytes_raw", "_pmc_fetch_cc_kernels.csv"), (f_write, "write_bytes", "_pmc_write_cc_kernels.csv")):
    if not os.path.exists(fn):
        continue
    rows = [r for r in csv.DictReader(open(fn)) if short(r["Kernel_Name"]).startswith("cc_k_")]
    keep = ["Dispatch_Id", "Grid_Size", "Kernel_Name", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Counter_Name",
            "Counter_Value", "Start_Timestamp", "End_Timestamp"]
    with open(os.path.join(out, tag + oname), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=keep)
        w.writeheader()
        for r in rows:
            r2 = {k: r[k] for k in keep}
            r2["Kernel_Name"] = short(r["Kernel_Name"])
            w.writerow(r2)
    acc = collections.defaultdict(list)
    for r in rows:
        k = short(r["Kernel_Name"])
        if step_grid(k) is not None and int(r["Grid_Size"]) == step_grid(k):
            acc[k].append(float(r["Counter_Value"]) * 1024.0)
    for k, v in acc.items():
        summ["kernels"].setdefault(k, {"grid_size": step_grid(k)})[key] = sum(v) / len(v)
for k, d in summ["kernels"].items():
    fr, wr = d.get("fetch_bytes_raw"), d.get("write_bytes")
    d["fetch_bytes_corrected"] = fr * 2 if (k == "cc_k_rasterize" and fr is not None) else None
    if fr is not None and wr is not None:
        d["hbm_bytes_per_launch"] = (d["fetch_bytes_corrected"] or fr) + wr
json.dump(summ, open(os.path.join(out, tag + "_pmc_summary.json"), "w"), indent=1)
print("wrote", sorted(os.listdir(out)))
