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
summary = {}
print("\n== PMC (mean per dispatch) ==")
for d in ("pmc1", "pmc2", "pmc3", "pmc4", "pmc5", "pmc6"):
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
            tot = {n: sum(v) for n, v in cs.items() if n != "_meta"}
            summary.update(tot)
            summary["_dispatches"] = max(summary.get("_dispatches", 0), max(len(v) for n, v in cs.items() if n != "_meta"))
# derived figures over ALL k_pass dispatches of the run (sums, so that short and long dispatches weigh by their work)
print("\n== derived (all k_pass dispatches) ==")
g = summary.get
if g("SQ_ACTIVE_INST_VALU") and g("SQ_THREAD_CYCLES_VALU"):
    print(f"lanes active per VALU instruction = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = {g('SQ_THREAD_CYCLES_VALU') / g('SQ_ACTIVE_INST_VALU'):.1f} of 64")
if g("SQ_INSTS_VALU") and g("SQ_INSTS_SALU"):
    print(f"SALU : VALU = 1 : {g('SQ_INSTS_VALU') / g('SQ_INSTS_SALU'):.2f}; branches per VALU instruction = 1 : {g('SQ_INSTS_VALU') / max(g('SQ_INSTS_BRANCH', 0), 1):.1f}")
if g("SQ_WAVE_CYCLES"):
    wc = g("SQ_WAVE_CYCLES")
    print(f"wave time: issuing {g('SQ_ACTIVE_INST_ANY', 0) / wc:.1%}, parked in s_waitcnt {g('SQ_WAIT_ANY', 0) / wc:.1%}, issue-stalled {g('SQ_WAIT_INST_ANY', 0) / wc:.1%}")
n_frames = int(os.environ.get("PROF_FRAMES", "0") or 0)
n_rays = float(os.environ.get("PROF_RAYS_PER_FRAME", "0") or 0)
n_exec = float(os.environ.get("PROF_RAYS_EXECUTED_PER_FRAME", "0") or 0)
if n_frames:
    print(f"{n_frames} frames over the {summary.get('_dispatches', 0)} dispatches")
    if g("SQ_INSTS_VALU") and n_rays:
        print(f"VALU wave-instructions per ray = {g('SQ_INSTS_VALU') / (n_frames * n_rays):.2f} (all {n_rays:.0f} World::ray_intersection calls of a frame)"
              + (f"; per ray whose shape tests ran ({n_exec:.0f} a frame) = {g('SQ_INSTS_VALU') / (n_frames * n_exec):.2f}" if n_exec else ""))
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        print(f"HBM traffic per frame = (2 x FETCH_SIZE + WRITE_SIZE) KB / frames = {(2 * g('FETCH_SIZE') + g('WRITE_SIZE')) * 1000 / n_frames / 1e6:.2f} MB (gfx950: FETCH_SIZE reports half the bytes of wide reads)")
