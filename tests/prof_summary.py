"""Print this repo's kernels from a rocprofv3 *_kernel_stats.csv (tuning aid)."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith("cc_k_"):
        print("%-16s calls %4s avg %10.1f us  min %10.1f  max %10.1f  total %8.2f ms" % (
            r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
            float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
