"""Condense gpurun_out/prof (rocprofv3 csv) into a text summary for profiles/."""
import csv, glob, collections, os, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
print("== rocprofv3 --kernel-trace --stats -- " + os.environ.get("PROF_CMD", "python3 bench.py --no-cpu-baseline") + " ==")
for f in glob.glob(root + "/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:110]:110s} calls={r['Calls']:>4s} avg_ns={float(r['AverageNs']):12.0f} pct={r['Percentage']}")
# per-dispatch durations of k_pass (launches differ in size: 1 instrumented frame, a warm-up remainder, full launches)
for f in glob.glob(root + "/trace/*/*_kernel_trace.csv"):
    durs = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if "k_pass" in r["Kernel_Name"])
    if durs:
        print("k_pass dispatch durations, ms: " + " ".join(f"{d:.3f}" for d in durs))
print("\n== PMC (mean per dispatch) ==")
for d in ("pmc1", "pmc2", "pmc3", "pmc4"):
    for f in glob.glob(f"{root}/{d}/*/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = "k_pass" if "k_pass" in r["Kernel_Name"] else None
            if k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {x: r[x] for x in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Scratch_Size")}
                agg[k]["_meta"] = meta
        for k, cs in agg.items():
            for name, v in cs.items():
                if name == "_meta":
                    continue
                print(f"{k:11s} {name:26s} mean/dispatch {sum(v)/len(v):.5g}  dispatches {len(v)}")
            if d == "pmc1":
                print(f"{k:11s} launch {cs['_meta']}")
