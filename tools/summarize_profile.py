"""dev helper: gpurun_out/prof_<tag>/<name>[/lease<n>]/ (written by tools/collect_profile.sh on the GPU box, one lease per
gpurun call) -> profiles/<tag>/<name>/: summary.json lists EVERY lease (device identity, rocprofv3 average, HIP-event
min / median / max of the un-profiled and of the profiled process, FETCH_SIZE / WRITE_SIZE) so that a driver-run timing can
be matched to a profile taken on a comparable box; the per-lease raw CSVs are kept beside it.
    python tools/summarize_profile.py r03 cylinder3D"""
import csv, glob, json, os, shutil, sys
tag, name = sys.argv[1], sys.argv[2]
src, dst = f"gpurun_out/prof_{tag}/{name}", f"profiles/{tag}/{name}"
os.makedirs(dst, exist_ok=True)
leases = sorted((d for d in glob.glob(f"{src}/lease*") if os.path.isdir(d)), key=lambda d: int(d.rsplit("lease", 1)[1])) or [src]


def one(lease_dir):
    bench = json.loads([l for l in open(f"{lease_dir}/bench.json").read().splitlines() if l.startswith("{")][-1])
    stats = list(csv.DictReader(open(f"{lease_dir}/kernel_stats.csv")))
    # the headline's kernel (bench.json names it; the profiled run has no sub-records, so the largest total is the same one)
    want = bench["roofline"].get("kernel", "").split("<")[0]
    cand = [r for r in stats if "interp_planned" in r["Name"] and "permute" not in r["Name"]]
    named = [r for r in cand if want and want in r["Name"]]
    dom = max(named or cand, key=lambda r: float(r["TotalDurationNs"]))
    short = dom["Name"].split("(")[0].split("::")[-1]

    def per_launch(counter):
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f"{lease_dir}/pmc_{counter}.csv")) if short.split("<")[0] in r["Kernel_Name"]]
        return sum(v) / len(v), len(v)

    fetch, nf = per_launch("FETCH_SIZE")
    write, nw = per_launch("WRITE_SIZE")
    cfg = bench["config"]
    prof_line = [l for l in open(f"{lease_dir}/stats.log").read().splitlines() if l.startswith("{")]
    in_prof = json.loads(prof_line[-1])["roofline"] if prof_line else {}
    b_alg = bench["roofline"]["algorithmic_bytes"]
    events = {k: v for k, v in bench["roofline"].items() if k.startswith("kernel_ms")}
    return {
        "lease": os.path.basename(lease_dir) if lease_dir != src else "lease1",
        "device": bench.get("device"),
        "kernel": dom["Name"], "calls": int(dom["Calls"]), "avg_ms_rocprof": float(dom["AverageNs"]) / 1e6,
        "min_ms_rocprof": float(dom["MinNs"]) / 1e6, "max_ms_rocprof": float(dom["MaxNs"]) / 1e6,
        "hip_events_ms_unprofiled_process": events,
        "hip_events_ms_in_the_profiled_process": {k: v for k, v in in_prof.items() if k.startswith("kernel_ms")},
        "FETCH_SIZE_KB_per_launch": fetch, "FETCH_SIZE_launches": nf, "WRITE_SIZE_KB_per_launch": write, "WRITE_SIZE_launches": nw,
        # gfx950: FETCH_SIZE tallies 128-B line requests at 64 B (MI355X_MICROARCH.md, HBM section) -> x2
        "traffic_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
        "algorithmic_bytes": b_alg, "traffic_over_algorithmic": (2.0 * fetch + write) * 1024.0 / b_alg,
        "frac_of_peak_rocprof": b_alg / (float(dom["AverageNs"]) * 1e-9) / 8e12,
        "frac_of_peak_hip_events_unprofiled": b_alg / (events["kernel_ms"] * 1e-3) / 8e12,
        "workload": cfg["workload"],
        "workload_key": f"{cfg['workload'].split(' ')[0]}/{cfg.get('shape_key', 'T%d' % cfg['t_batch'])}" + ("/inplace" if "in place" in cfg.get("input", "") else ""),
        "code_sha": bench.get("code_sha"),
    }, bench


records = []
for i, d in enumerate(leases):
    rec, bench = one(d)
    records.append(rec)
    sub = f"{dst}/{rec['lease']}"
    os.makedirs(sub, exist_ok=True)
    for f in ("bench.json", "kernel_stats.csv", "pmc_FETCH_SIZE.csv", "pmc_WRITE_SIZE.csv"):
        shutil.copy(f"{d}/{f}", f"{sub}/{f}")
fr = [r["frac_of_peak_rocprof"] for r in records]
summary = {
    "workload_key": records[0]["workload_key"], "workload": records[0]["workload"], "kernel": records[0]["kernel"],
    # fingerprint of csrc/ + include/ the collection ran on (bench.py: code_sha; `roofline.traffic_stale` compares it with the
    # tree it runs in) and the commit the dev container stood at when the summary was made
    "code_sha": records[-1]["code_sha"], "code_sha_all_leases_equal": len({r["code_sha"] for r in records}) == 1,
    "git_head_at_summary": os.popen("git rev-parse HEAD 2>/dev/null").read().strip() or None,
    "algorithmic_bytes": records[0]["algorithmic_bytes"], "n_leases": len(records),
    "frac_of_peak_rocprof_min_median_max": [min(fr), sorted(fr)[len(fr) // 2], max(fr)],
    "avg_ms_rocprof_per_lease": [r["avg_ms_rocprof"] for r in records],
    # what bench.py reports as `roofline.traffic` (labelled recorded): the mean over the leases
    "traffic_bytes_per_launch": sum(r["traffic_bytes_per_launch"] for r in records) / len(records),
    "traffic_over_algorithmic": sum(r["traffic_over_algorithmic"] for r in records) / len(records),
    "leases": records,
}
json.dump(summary, open(f"{dst}/summary.json", "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "leases"}, indent=1))
