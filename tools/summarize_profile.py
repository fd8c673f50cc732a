"""dev helper: gpurun_out/prof_<tag>/<name>/ (written by tools/collect_profile.sh on the GPU box) -> profiles/<tag>/<name>/
    python tools/summarize_profile.py r02 cylinder3D"""
import csv, json, os, shutil, sys
tag, name = sys.argv[1], sys.argv[2]
src, dst = f"gpurun_out/prof_{tag}/{name}", f"profiles/{tag}/{name}"
os.makedirs(dst, exist_ok=True)
bench = json.loads([l for l in open(f"{src}/bench.json").read().splitlines() if l.startswith("{")][-1])
stats = list(csv.DictReader(open(f"{src}/kernel_stats.csv")))
cand = [r for r in stats if "interp_planned" in r["Name"] and "permute" not in r["Name"]]
dom = max(cand, key=lambda r: float(r["TotalDurationNs"]))
short = dom["Name"].split("(")[0].split("::")[-1]


def per_launch(counter):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f"{src}/pmc_{counter}.csv")) if short.split("<")[0] in r["Kernel_Name"]]
    return sum(v) / len(v), len(v)


fetch, nf = per_launch("FETCH_SIZE")
write, nw = per_launch("WRITE_SIZE")
cfg = bench["config"]
# the bench line printed inside the rocprofv3 --stats run: HIP events and rocprofv3 see the same launches there
prof_line = [l for l in open(f"{src}/stats.log").read().splitlines() if l.startswith("{")]
events_in_profiled_run = json.loads(prof_line[-1])["roofline"]["kernel_ms"] if prof_line else None
summary = {
    "kernel": dom["Name"], "calls": int(dom["Calls"]), "avg_ms_rocprof": float(dom["AverageNs"]) / 1e6,
    "FETCH_SIZE_KB_per_launch": fetch, "FETCH_SIZE_launches": nf, "WRITE_SIZE_KB_per_launch": write, "WRITE_SIZE_launches": nw,
    "hip_events_ms_in_the_profiled_process": events_in_profiled_run,
    "bench_kernel_ms_hip_events": bench["roofline"]["kernel_ms"],
    # gfx950: FETCH_SIZE tallies 128-B line requests at 64 B (MI355X_MICROARCH.md, HBM section) -> x2
    "traffic_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
    "algorithmic_bytes": bench["roofline"].get("algorithmic_bytes"),
    "traffic_over_algorithmic": (2.0 * fetch + write) * 1024.0 / bench["roofline"]["algorithmic_bytes"],
    "frac_of_peak_rocprof": bench["roofline"]["algorithmic_bytes"] / (float(dom["AverageNs"]) * 1e-9) / 8e12,
    "workload": cfg["workload"], "workload_key": f"{cfg['workload'].split(' ')[0]}/{cfg.get('shape_key', 'T%d' % cfg['t_batch'])}",
    "device": bench.get("device"), "kernel_ms_hip_events_stats": {k: v for k, v in bench["roofline"].items() if k.startswith("kernel_ms")},
}
json.dump(summary, open(f"{dst}/summary.json", "w"), indent=1)
shutil.copy(f"{src}/bench.json", f"{dst}/bench.json")
shutil.copy(f"{src}/kernel_stats.csv", f"{dst}/kernel_stats.csv")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    shutil.copy(f"{src}/pmc_{c}.csv", f"{dst}/pmc_{c}.csv")
print(json.dumps(summary, indent=1))
