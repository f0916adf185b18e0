#!/usr/bin/env python3
"""Turn gpurun_out/prof/ (made by tools/refresh_profiles.sh on the GPU box) into the tracked summaries under profiles/."""
import csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(dst, f"{tag}_bench_default.json"))
shutil.copy(os.path.join(src, "bench_stats", "b_kernel_stats.csv"), os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
shutil.copy(os.path.join(src, "edt_stats", "e_kernel_stats.csv"), os.path.join(dst, f"{tag}_edt64_kernel_stats.csv"))


def per_kernel(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = (r["Kernel_Name"].split("(")[0], r["Dispatch_Id"])
        acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
    out = {}
    for (name, _), v in acc.items():
        out.setdefault(name, []).append(v)
    return {k: (sum(v) / len(v), len(v)) for k, v in out.items()}


rows = []
raw = {}
for counter, d in (("FETCH_SIZE", "edt_fetch"), ("WRITE_SIZE", "edt_write")):
    pk = per_kernel(os.path.join(src, d, "e_counter_collection.csv"), counter)
    with open(os.path.join(dst, f"{tag}_edt64_pmc_{counter.lower()}.csv"), "w") as f:
        f.write("kernel,counter,mean_KB_per_dispatch,dispatches\n")
        for k, (m, n) in sorted(pk.items()):
            f.write(f"{k},{counter},{m:.3f},{n}\n")
            short = "colbits" if "colbits" in k else "band" if "band" in k else None
            if short:
                raw[f"{short}_{counter}"] = m
total = int((2 * raw["colbits_FETCH_SIZE"] + raw["colbits_WRITE_SIZE"] + raw["band_FETCH_SIZE"] + raw["band_WRITE_SIZE"]) * 1024)
json.dump({"salt20": {
    "hbm_bytes_per_launch": total, "raw_KB": raw,
    "corrections": "FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc passes, KB per dispatch. colbits FETCH_SIZE x2 "
                   "(gfx950 reports 1/2 of 16 B/lane streaming reads; calibrated here: 2 x raw = the 64 MiB the kernel reads). band "
                   "FETCH_SIZE taken raw (4 B/lane reads of the column words, width uncalibrated; includes per-XCD L2 re-fetches of "
                   "neighbouring bands and Infinity-Cache hits). WRITE_SIZE exact (8 MiB and 256 MiB).",
    "algorithmic_bytes_per_launch": 5 * 64 * 1024 * 1024,
    "workload": "EDT of 64 x 1024x1024 salt20 grids (tools/edt_variants.py)"}}, open(os.path.join(dst, "edt_traffic.json"), "w"), indent=1)
print("traffic bytes per launch:", total, raw)
for r in csv.DictReader(open(os.path.join(dst, f"{tag}_edt64_kernel_stats.csv"))):
    print(r["Name"][:60], r["Calls"], r["AverageNs"])
