"""Static instruction census per kernel (build container, no GPU): hipcc -S of the product's translation unit, instructions
classified by mnemonic.  usage: python profiles/r4/census.py > profiles/r4/f_static_instruction_census.txt"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(ROOT, "contour-context_amd", "csrc", "cont2_amd.hip")
asm = "/tmp/_census.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
                       "-Wno-unused-value", src, "-o", asm], stderr=subprocess.DEVNULL)
rows, cur = {}, None
for line in open(asm):
    m = re.match(r"^(_Z\w+|cc_k_\w+):\s*(;.*)?$", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        cur = rows.setdefault(name, dict(total=0, valu=0, f64=0, salu=0, lds=0, vmem=0, wait=0)) if name.startswith("cc_k_") else None
        continue
    if cur is None:
        continue
    t = line.strip()
    if t.startswith("s_endpgm"):
        cur = None
        continue
    op = t.split()[0] if t and not t.startswith((";", ".", "//")) and not t.endswith(":") else None
    if not op or not re.match(r"^[a-z_0-9]+$", op):
        continue
    cur["total"] += 1
    if op.startswith("s_waitcnt"):
        cur["wait"] += 1
    elif op.startswith("s_"):
        cur["salu"] += 1
    elif op.startswith("v_"):
        cur["valu"] += 1
        cur["f64"] += 1 if "f64" in op else 0
    elif op.startswith("ds_"):
        cur["lds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        cur["vmem"] += 1
print("# static instruction counts per kernel (hipcc -O3 --offload-arch=gfx950 -S, profiles/r4/census.py, end of round 4);")
print("# f64 = vector instructions with an f64 operand; salu includes the exec-mask bookkeeping of divergent branches")
for k in sorted(rows):
    v = rows[k]
    if v["total"]:
        print("%-34s total %5d  valu %5d f64 %4d salu %5d (%2d %%) lds %4d vmem %4d waitcnt %4d" % (
            k[:34], v["total"], v["valu"], v["f64"], v["salu"], round(100.0 * v["salu"] / v["total"]), v["lds"], v["vmem"], v["wait"]))
