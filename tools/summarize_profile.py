"""dev helper: gpurun_out/prof_<tag>/ (written by tools/collect_profile.sh on the GPU box) -> profiles/<tag>/
    python tools/summarize_profile.py r01"""
import csv, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = f"gpurun_out/prof_{tag}", f"profiles/{tag}"
os.makedirs(dst, exist_ok=True)
bench = json.loads([l for l in open(f"{src}/bench.json").read().splitlines() if l.startswith("{")][-1])
stats = list(csv.DictReader(open(f"{src}/kernel_stats.csv")))
dom = max(stats, key=lambda r: float(r["TotalDurationNs"]))
assert "interp_planned_kernel" in dom["Name"], dom["Name"]
def per_launch(counter):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f"{src}/pmc_{counter}.csv")) if "interp_planned_kernel" in r["Kernel_Name"]]
    return sum(v) / len(v), len(v)
fetch, nf = per_launch("FETCH_SIZE")
write, nw = per_launch("WRITE_SIZE")
summary = {
    "kernel": dom["Name"], "calls": int(dom["Calls"]), "avg_ms_rocprof": float(dom["AverageNs"]) / 1e6,
    "FETCH_SIZE_KB_per_launch": fetch, "FETCH_SIZE_launches": nf, "WRITE_SIZE_KB_per_launch": write, "WRITE_SIZE_launches": nw,
    "bench_kernel_ms_hip_events": bench["roofline"]["kernel_ms"] if "kernel_ms" in bench["roofline"] else bench["ms_per_step"],
    # gfx950: FETCH_SIZE tallies 128-B line requests at 64 B (MI355X_MICROARCH.md, HBM section) -> x2
    "traffic_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
    "algorithmic_bytes": bench["roofline"].get("algorithmic_bytes"),
    "workload": bench["config"]["workload"],
}
json.dump(summary, open(f"{dst}/summary.json", "w"), indent=1)
shutil.copy(f"{src}/bench.json", f"{dst}/bench_{tag}.json")
shutil.copy(f"{src}/kernel_stats.csv", f"{dst}/bench_kernel_stats.csv")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    shutil.copy(f"{src}/pmc_{c}.csv", f"{dst}/pmc_{c}.csv")
if os.path.exists(f"{src}/pmc_extra.csv"):          # tools/collect_counters.sh
    shutil.copy(f"{src}/pmc_extra.csv", f"{dst}/pmc_extra.csv")
print(json.dumps(summary, indent=1))
