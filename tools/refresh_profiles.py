#!/usr/bin/env python3
"""Turn gpurun_out/prof/ (made by tools/refresh_profiles.sh on the GPU box) into the tracked summaries under profiles/.
usage: python tools/refresh_profiles.py r03"""
import csv, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
for f in ("bench_default", "bench_forced_dist", "bench_steps20"):
    shutil.copy(os.path.join(src, f + ".json"), os.path.join(dst, f"{tag}_{f}.json"))
shutil.copy(os.path.join(src, "bench_stats", "b_kernel_stats.csv"), os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
shutil.copy(os.path.join(src, "toppra_stats", "t_kernel_stats.csv"), os.path.join(dst, f"{tag}_toppra_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "smoothing_stats", "s_kernel_stats.csv")):   # tools/smoothing_one.py under rocprofv3 --kernel-trace --stats
    shutil.copy(os.path.join(src, "smoothing_stats", "s_kernel_stats.csv"), os.path.join(dst, f"{tag}_smoothing_kernel_stats.csv"))
for d in sorted(os.listdir(src)):
    m = re.match(r"edt_stats_(\d+)_(\d+)_(\w+)$", d)
    if m and os.path.isdir(os.path.join(src, d)):
        name = f"{tag}_edt_{m.group(1)}_{m.group(3)}" + ("" if (m.group(1), m.group(2)) in (("1024", "64"), ("4096", "4")) else f"_x{m.group(2)}")
        shutil.copy(os.path.join(src, d, "e_kernel_stats.csv"), os.path.join(dst, name + "_kernel_stats.csv"))
for f in ("min3_mb.log", "lds_mb.log"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_microbench_{f}"))


def per_dispatch(path, counter, kernel_sub):
    """counter value per dispatch (summed over the XCDs / instances of a dispatch) of kernels whose name contains kernel_sub"""
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or kernel_sub not in r["Kernel_Name"]:
            continue
        acc[r["Dispatch_Id"]] = acc.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    return list(acc.values())


# ---- EDT traffic (roofline.traffic): FETCH_SIZE / WRITE_SIZE in separate passes, KB per dispatch ----
def edt_traffic(size, kernels):
    raw = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        path = os.path.join(src, f"edt_{size}_{counter}", "e_counter_collection.csv")
        for short, sub in kernels:
            v = per_dispatch(path, counter, sub)
            raw[f"{short}_{counter}"] = sum(v[3:]) / max(1, len(v[3:]))      # the first three dispatches are warm-up
    return raw


traffic = {}
raw = edt_traffic(1024, (("colbits", "edt_colbits"), ("band", "edt_band")))
total = int((2 * raw["colbits_FETCH_SIZE"] + raw["colbits_WRITE_SIZE"] + raw["band_FETCH_SIZE"] + raw["band_WRITE_SIZE"]) * 1024)
traffic["salt20"] = {
    "hbm_bytes_per_launch": total, "raw_KB": raw,
    "corrections": "FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc passes, KB per dispatch. colbits FETCH_SIZE x2 "
                   "(gfx950 reports 1/2 of 16 B/lane streaming reads; 2 x raw = the 64 MiB the kernel reads). band FETCH_SIZE taken raw "
                   "(4 B/lane reads of the column words, width uncalibrated; includes per-XCD L2 re-fetches of neighbouring bands and "
                   "Infinity-Cache hits). WRITE_SIZE exact (16-byte streaming stores).",
    "algorithmic_bytes_per_launch": 5 * 64 * 1024 * 1024,
    "workload": "EDT of 64 x 1024x1024 salt20 grids (tools/edt_variants.py libsea_current_hip.so 1024 64 salt20)"}
raw4 = edt_traffic(4096, (("colbits", "edt_colbits"), ("updown", "edt_updown"), ("band", "edt_band_wide")))
total4 = int((2 * raw4["colbits_FETCH_SIZE"] + raw4["colbits_WRITE_SIZE"] + raw4["updown_FETCH_SIZE"] + raw4["updown_WRITE_SIZE"] +
              raw4["band_FETCH_SIZE"] + raw4["band_WRITE_SIZE"]) * 1024)
traffic["salt20_4096"] = {
    "hbm_bytes_per_launch": total4, "raw_KB": raw4,
    "corrections": "as above; the updown and band kernels read 4- and 8-byte words (taken raw)",
    "algorithmic_bytes_per_launch": 5 * 4 * 4096 * 4096,
    "workload": "EDT of 4 x 4096x4096 salt20 grids (tools/edt_variants.py libsea_current_hip.so 4096 4 salt20)"}
json.dump(traffic, open(os.path.join(dst, "edt_traffic.json"), "w"), indent=1)
print("EDT traffic bytes per launch:", total, total4)

# ---- what bounds the band kernels: counters per launch on block-type and salt maps ----
bound = {}
for size, fam, sub, cells in ((1024, "blocks", "edt_band_k16", 64 << 20), (4096, "blocks", "edt_band_wide", 64 << 20),
                              (1024, "salt20", "edt_band_k16", 64 << 20), (4096, "salt20", "edt_band_wide", 64 << 20)):
    c = {}
    for d in sorted(os.listdir(src)):
        if d.startswith(f"edtpmc_{size}_{fam}_") and os.path.isdir(os.path.join(src, d)):
            path = os.path.join(src, d, "e_counter_collection.csv")
            for r in csv.DictReader(open(path)):
                if sub in r["Kernel_Name"]:
                    c.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
                    c[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    m = {k: (lambda v: sum(v[3:]) / max(1, len(v[3:])))(list(v.values())) for k, v in c.items()}
    if not m:
        continue
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8                      # summed over the 8 XCDs
    simds = 1024
    e = {"kernel": sub, "counters_per_launch": m, "gpu_cycles_per_launch": cyc,
         "VALU_instructions_per_pixel": 64 * m.get("SQ_INSTS_VALU", 0) / cells,
         "VALU_busy_fraction": (4 * m.get("SQ_ACTIVE_INST_VALU", 0) / simds) / cyc if cyc else None,
         "LDS_busy_fraction": (m.get("SQ_LDS_IDX_ACTIVE", 0) / 256) / cyc if cyc else None,
         "wave_time_split": {k: m.get(k, 0) / m["SQ_WAVE_CYCLES"] for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY") if "SQ_WAVE_CYCLES" in m},
         "note": "VALU_busy = SQ_ACTIVE_INST_VALU quad-cycles x 4 / 1024 SIMDs over the launch's GPU cycles (GRBM_GUI_ACTIVE / 8 XCDs); "
                 "LDS_busy = SQ_LDS_IDX_ACTIVE / 256 CUs over the same cycles"}
    bound[f"{size}_{fam}"] = e
json.dump(bound, open(os.path.join(dst, f"{tag}_edt_band_bound_pmc.json"), "w"), indent=1)
print("band kernels:", json.dumps({k: {kk: v[kk] for kk in ("VALU_instructions_per_pixel", "VALU_busy_fraction", "LDS_busy_fraction")} for k, v in bound.items()}, indent=1))

# ---- the longest search of the headline batch alone on the chip ----
lg = json.load(open(os.path.join(src, "astar_longest.json")))
cl = {}
for d in sorted(os.listdir(src)):
    if d.startswith("astarlong_") and os.path.isdir(os.path.join(src, d)):
        for r in csv.DictReader(open(os.path.join(src, d, "a_counter_collection.csv"))):
            if "astar_kernel_dual" in r["Kernel_Name"]:
                cl.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
                cl[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
ml = {k: sum(list(v.values())[-10:]) / 10 for k, v in cl.items()}          # the last ten dispatches are the single-query launches
lg["counters_per_launch_both_wavefronts"] = ml
lg["per_step_both_wavefronts"] = {k: ml[k] / lg["steps"] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR") if k in ml}
if "SQ_WAVE_CYCLES" in ml:
    lg["wave_time_split"] = {k: ml.get(k, 0) / ml["SQ_WAVE_CYCLES"] for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")}
lg["note"] = ("one query, two wavefronts, alone on the chip: cycles_per_step is wavefront 0's critical path (kilocycles / steps); the counters "
              "are sums over both wavefronts; SQ_WAIT_ANY includes wavefront 1's sleeps while it has nothing to do")
json.dump(lg, open(os.path.join(dst, f"{tag}_astar_longest_query_pmc.json"), "w"), indent=1)
print("A* longest:", json.dumps({k: lg[k] for k in ("ms_alone", "steps", "cycles_per_step", "per_step_both_wavefronts")}, indent=1))

# ---- A* at saturation: 6144 copies of one query in one launch ----
log = open(os.path.join(src, "astar_FETCH_SIZE.log")).read()
m = re.search(r"x6144: ([\d.]+) ms, (\d+) expansions / (\d+) popped / (\d+) steps each -> ([\d.]+) G", log)
ms, ex, pop, steps, gexp = float(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), float(m.group(5))
copies = 6144
astar = {"workload": "6144 copies of one salt20 query (tools/astar_saturation.py salt20 6144) in one launch of the throughput build of astar_kernel_dual (3072 resident slots of two wavefronts: two rounds, no tail); per_step = per frontier step of wavefront 0, both wavefronts' instructions counted",
         "expansions_per_query": ex, "popped_per_query": pop, "steps_per_query": steps, "launch_ms_under_profiler": ms,
         "G_expansions_per_s_under_profiler": gexp}


def top3(d, counter):
    v = sorted(per_dispatch(os.path.join(src, d, "a_counter_collection.csv"), counter, "astar_kernel"), reverse=True)[:3]
    return sum(v) / len(v)


c = {}
for d, names in (("astar_SQ_INSTS_VALU_SQ_INSTS_SALU_SQ_WAVE_CYCLES", ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES")),
                 ("astar_SQ_INSTS_LDS_SQ_INSTS_VMEM_RD_SQ_INSTS_VMEM_WR", ("SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")),
                 ("astar_TCC_ATOMIC_sum", ("TCC_ATOMIC_sum",)), ("astar_TCC_HIT_sum_TCC_MISS_sum", ("TCC_HIT_sum", "TCC_MISS_sum")),
                 ("astar_FETCH_SIZE", ("FETCH_SIZE",)), ("astar_WRITE_SIZE", ("WRITE_SIZE",))):
    for n in names:
        c[n] = top3(d, n)
nsteps = copies * steps
nexp = copies * ex
astar["counters_per_launch"] = c
astar["per_step"] = {k: c[k] / nsteps for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")}
astar["per_step"]["wave_cycles_x4"] = 4 * c["SQ_WAVE_CYCLES"] / nsteps
astar["L2_atomic_requests_per_expansion"] = c["TCC_ATOMIC_sum"] / nexp
astar["L2_miss_fraction"] = c["TCC_MISS_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
astar["HBM_GBps_fetch_plus_write_raw"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / (ms * 1e-3) / 1e9
astar["round2_same_workload"] = {"per_step": {"SQ_INSTS_VALU": 208, "SQ_INSTS_SALU": 180, "SQ_INSTS_LDS": 25}, "G_expansions_per_s": 30, "source": "profiles/r02_astar_saturation_pmc.json"}
astar["round1_same_workload"] = {"L2_atomic_requests_per_expansion": 1.28, "L2_miss_fraction": 0.56, "cycles_per_step_alone": 2830,
                                 "G_expansions_per_s": "8.5-9.7", "source": "profiles/r01_astar_saturation_pmc.json"}
json.dump(astar, open(os.path.join(dst, f"{tag}_astar_saturation_pmc.json"), "w"), indent=1)
print("A*:", json.dumps({k: v for k, v in astar.items() if k not in ("counters_per_launch", "workload")}, indent=1))

# ---- TOPP-RA instruction counts ----
tp = {}
path = os.path.join(src, "toppra_pmc", "t_counter_collection.csv")
for kern in ("toppra_fast_kernel", "toppra_sample_kernel"):
    tp[kern] = {}
    for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES"):
        v = per_dispatch(path, n, kern)
        tp[kern][n + "_per_launch"] = sum(v) / len(v)
    tp[kern]["per_plan"] = {n: tp[kern][n + "_per_launch"] / 1024 for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU")}
    tp[kern]["wave_cycles_x4_per_plan"] = 4 * tp[kern]["SQ_WAVE_CYCLES_per_launch"] / 1024
tp["workload"] = ("1024 plans, 6 joints, 200 stages (tools/toppra_one.py); sweep kernel: three wavefronts per plan (chains | pair elimination / "
                  "forward coefficients | slots / backward coefficients), counts are sums over the three; a plan is 400 dependent stages")
tp["toppra_fast_kernel"]["VALU_per_stage"] = tp["toppra_fast_kernel"]["per_plan"]["SQ_INSTS_VALU"] / 400
tp["round2_first_version"] = {"VALU_per_stage": 226, "ms_sweep": 0.522, "ms_sample": 0.180, "note": "one wavefront per plan, exact divisions on the chains"}
json.dump(tp, open(os.path.join(dst, f"{tag}_toppra_pmc.json"), "w"), indent=1)
print("TOPP-RA:", json.dumps(tp, indent=1))
for f in sorted(os.listdir(dst)):
    if f.startswith(tag) and f.endswith("kernel_stats.csv") and "edt" in f:
        for r in csv.DictReader(open(os.path.join(dst, f))):
            if "edt" in r["Name"]:
                print(f, r["Name"][:48], r["Calls"], r["AverageNs"])
