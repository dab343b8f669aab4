#!/usr/bin/env python3
"""Turn one round's rocprofv3 output under gpurun_out/prof/<tag>_kt (kernel trace + stats of bench.py) and the
PMC log of tools/pmc.sh into the files kept under profiles/ (kernel stats, per-launch trace of the scan kernel,
roofline_traffic.json).  Usage: tools/collect_profiles.py <tag> <pmc-log> [round-prefix]"""
import csv, glob, json, os, re, statistics, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, pmc_log = sys.argv[1], sys.argv[2]
pre = sys.argv[3] if len(sys.argv) > 3 else "r01"
kt_dir = os.path.join(ROOT, "gpurun_out", "prof", f"{tag}_kt")
# gpurun merges every call's output into the same local directory: take the newest run
newest = lambda pat: max(glob.glob(os.path.join(kt_dir, "**", pat), recursive=True), key=os.path.getmtime)
stats = newest("*kernel_stats.csv")
trace = newest("*kernel_trace.csv")
out_stats = os.path.join(ROOT, "profiles", f"{pre}_kernel_stats.csv")
open(out_stats, "w").write(open(stats).read())

def short(name):
    m = re.search(r"(kmp_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "").replace(" ", "")) if m else name

rows = []
for r in csv.DictReader(open(trace)):
    if "kmp_scan" in r["Kernel_Name"]:
        rows.append((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("VGPR_Count", r.get("Arch_VGPR_Count", "")),
                     r.get("SGPR_Count", ""), r.get("LDS_Block_Size", ""), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))))
rows.sort()
with open(os.path.join(ROOT, "profiles", f"{pre}_kernel_trace_scan.csv"), "w") as f:
    f.write("dispatch_id,kernel,duration_ns,vgpr,sgpr,lds,grid_x,workgroup_x\n")
    for r in rows:
        f.write(",".join(f'"{x}"' if "," in str(x) else str(x) for x in r) + "\n")
dur = [r[2] for r in rows]
kernel = rows[-1][1]
fetch = tcc = None
for line in open(pmc_log):
    p = line.split()
    if len(p) >= 3 and p[0] == "FETCH_SIZE": fetch = float(p[-1].split("=")[-1])
    if len(p) >= 3 and p[0] == "TCC_EA0_RDREQ_sum": tcc = float(p[-1].split("=")[-1])
hbm = int(round(fetch * 1024 * 2))
json.dump({
    "kernel": kernel,
    "workload": "1,000,000 x 1500 B synthetic payloads, one 16-byte pattern (bench.py default)",
    "hbm_bytes_per_launch": hbm,
    "derivation": f"rocprofv3 --pmc FETCH_SIZE (own pass, profiles/{pre}_pmc_scan_flat.txt): mean {fetch:.1f} KiB per launch; gfx950 counts 128-B requests "
                  f"at 64 B for wide coalesced streaming reads, so x2 (MI355X_MICROARCH.md, HBM); cross-check TCC_EA0_RDREQ_sum {tcc:.0f} x 128 B = "
                  f"{tcc * 128:.4e} B; the arena is 1,000,000 x 1504 B = 1.504e9 B: wavefront ranges start on 128-byte lines, no line is fetched twice. "
                  "WRITE_SIZE is negligible (1024 x 8 B partial counts).",
    "algorithmic_bytes_per_launch": 1500000000,
    "rocprof_kernel_trace_avg_ns": round(statistics.mean(dur)),
    "rocprof_kernel_trace_median_ns": round(statistics.median(dur)),
    "rocprof_kernel_trace_avg_ns_after_first_100": round(statistics.mean(dur[100:])) if len(dur) > 100 else None,
    "launches": len(dur),
}, open(os.path.join(ROOT, "profiles", "roofline_traffic.json"), "w"), indent=1)
print(f"{kernel}: {len(dur)} launches, mean {statistics.mean(dur)/1e3:.1f} us, median {statistics.median(dur)/1e3:.1f} us, "
      f"after the first 100: {statistics.mean(dur[100:])/1e3:.1f} us; HBM bytes/launch {hbm}")
