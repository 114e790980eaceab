#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/profile_round.sh (gpurun_out/prof_<tag>_*) into small tracked files under
profiles/: the per-kernel stats table, and the HBM traffic per launch of the probe+gather kernel from the PMC passes
(FETCH_SIZE doubled, as /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes for gfx950 wide coalesced reads)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("ROUND", "r04")
OUR = ("probe_gather_kernel", "miss_fill_kernel", "scatter_rows_kernel", "route_", "sample_insert_kernel", "scan_assign_kernel", "relabel_clear_kernel",
       "bucket_", "mean_aggregate")  # the product's kernels (torch has kernels in anonymous namespaces too)


def find(tag, kind, suffix):
    hits = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{kind}", "**", f"*{suffix}"), recursive=True)
    return hits[0] if hits else None


def bench_line(tag, kind):
    p = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{kind}.log")
    try:
        for line in reversed(open(p).read().strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
    except Exception:
        pass
    return None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "default"
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    summary = {"round": ROUND, "tag": tag}
    stats = find(tag, "trace", "kernel_stats.csv")
    if stats:
        rows = list(csv.DictReader(open(stats)))
        keep = [r for r in rows if "coala" in r["Name"].lower() or "anonymous namespace" in r["Name"]]
        with open(os.path.join(out_dir, f"{ROUND}_{tag}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            for r in keep + [r for r in rows if r not in keep][:6]:
                r = dict(r)
                r["Name"] = r["Name"][:160]
                w.writerow(r)
    trace = find(tag, "trace", "kernel_trace.csv")
    line = bench_line(tag, "trace")
    steps = line["steps"] if line else 200
    if trace:
        allrows = list(csv.DictReader(open(trace)))
        per = {}
        for r in allrows:
            if "anonymous namespace" in r["Kernel_Name"]:
                name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
                if name.startswith(OUR):
                    per.setdefault(name, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        summary["avg_us_rocprof_timed_region"] = {k: round(sum(v[-steps:]) / len(v[-steps:]) / 1e3, 2) for k, v in per.items()}
        rows = [r for r in allrows if "probe_gather_kernel" in r["Kernel_Name"]]
        rows = rows[-steps:]
        durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
        summary["probe_gather_avg_us_rocprof_timed_region"] = round(sum(durs) / max(len(durs), 1) / 1e3, 2)
        summary["probe_gather_launches_timed_region"] = len(durs)
        r0 = rows[-1] if rows else {}
        summary["probe_gather_vgpr"] = r0.get("VGPR_Count") or r0.get("Arch_VGPR_Count")
        summary["probe_gather_grid_block"] = [r0.get("Grid_Size_X") or r0.get("Grid_Size"), r0.get("Workgroup_Size_X") or r0.get("Workgroup_Size")]
    if line:
        summary["bench_roofline_from_hipEvents"] = line.get("roofline")
        summary["bench_value"] = line.get("value")
        summary["bench_ms_per_step"] = line.get("ms_per_step")
        summary["bench_config"] = line.get("config")
    # PMC passes: HBM bytes per launch of every cache kernel over the last <steps> dispatches of each (the timed region)
    per_kernel = {}
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        f = find(tag, kind, "counter_collection.csv")
        if not f:
            continue
        by = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or "anonymous namespace" not in r["Kernel_Name"]:
                continue
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            if name.startswith(OUR):
                by.setdefault(name, []).append(float(r["Counter_Value"]))
        for name, vals in by.items():
            vals = vals[-steps:]
            per_kernel.setdefault(name, {})[counter] = sum(vals) / len(vals)
    pmc_tab = {}
    for name, c in per_kernel.items():
        # counters are in KiB; FETCH_SIZE reads exactly half of a wide coalesced stream on gfx950 -> doubled
        fetch = c.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0
        write = c.get("WRITE_SIZE", 0.0) * 1024.0
        pmc_tab[name] = {"FETCH_SIZE_KiB_per_launch_raw": c.get("FETCH_SIZE"), "WRITE_SIZE_KiB_per_launch_raw": c.get("WRITE_SIZE"),
                         "fetch_bytes_per_launch_corrected_x2": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write}
    if pmc_tab:
        summary["pmc_per_kernel"] = pmc_tab
        summary["pmc_note"] = ("separate --pmc FETCH_SIZE / WRITE_SIZE passes, last <steps> dispatches of each kernel; FETCH_SIZE x2 (gfx950: 128-B "
                               "requests tallied at 64 B); reads of the pinned-host cold tier do not pass the HBM counters")
        k1 = [v for k, v in pmc_tab.items() if k.startswith("probe_gather_kernel")]
        if k1:
            summary["pmc"] = k1[0]
            if tag in ("default", "papers100m"):
                # keyed by workload; bench.py quotes an entry only while coala_cache.hip is the version it was measured on
                import hashlib
                path = os.path.join(out_dir, "pmc_probe_gather.json")
                try:
                    table = json.load(open(path))
                    if "hbm_bytes_per_launch" in table:      # the round-2 single-entry form
                        table = {}
                except Exception:
                    table = {}
                sha = hashlib.sha256(open(os.path.join(ROOT, "coala-gnn_amd", "csrc", "coala_cache.hip"), "rb").read()).hexdigest()[:16]
                alg = (line or {}).get("roofline", {}).get("alg_bytes_per_launch")
                table[tag] = {"hbm_bytes_per_launch": int(k1[0]["hbm_bytes_per_launch"]), "alg_bytes_per_launch_of_that_run": alg,
                              "ratio_to_algorithmic": round(k1[0]["hbm_bytes_per_launch"] / alg, 4) if alg else None,
                              "kernel_source_sha16": sha, "source": f"profiles/{ROUND}_{tag}_summary.json"}
                with open(path, "w") as f:
                    json.dump(table, f, indent=1)
    with open(os.path.join(out_dir, f"{ROUND}_{tag}_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    # the GPU box only sends gpurun_out/ back (<= 64 MiB): mirror the small summaries there and drop the raw traces
    if os.environ.get("GRAFT_REPO_ROOT"):
        import shutil
        mirror = os.path.join(ROOT, "gpurun_out", "profiles_out")
        os.makedirs(mirror, exist_ok=True)
        for name in os.listdir(out_dir):
            if name.startswith(f"{ROUND}_{tag}_") or name == "pmc_probe_gather.json":
                shutil.copy(os.path.join(out_dir, name), os.path.join(mirror, name))
        for kind in ("trace", "fetch", "write"):
            shutil.rmtree(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{kind}"), ignore_errors=True)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
