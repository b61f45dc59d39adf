#!/usr/bin/env python3
"""Per-kernel means of the FETCH_SIZE / WRITE_SIZE passes (tools/gpu_pmc2.sh) -> profiles/r01/pmc_fetch_write_summary.json,
profiles/scan_traffic.json, profiles/chain_traffic.json.  gfx950: FETCH_SIZE counts 64 B per 128 B request -> x2
(MI355X_MICROARCH.md, HBM section; calibrated on pack_kernel: 1.0 GB of ASCII read -> 0.52 GB reported)."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmcf_" + c, "*", "*counter_collection.csv")), key=os.path.getmtime)
    if not files:
        sys.exit("no counter_collection.csv for " + c)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(files[-1])):
        if r["Counter_Name"] == c:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    out[c] = {k: {"dispatches": len(v), "mean": sum(v) / len(v)} for k, v in acc.items()}
json.dump(out, open(os.path.join(ROOT, "profiles", "r01", "pmc_fetch_write_summary.json"), "w"), indent=1)
src = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python3 bench.py --steps 8 "
       "--warmup 4 --cpu-rounds 0 --index-steps 30 (tools/gpu_pmc2.sh; per-kernel means in profiles/r01/pmc_fetch_write_summary.json)")
corr = "gfx950: FETCH_SIZE counts 64 B per 128 B request (MI355X_MICROARCH.md, HBM section) -> x2; calibrated on pack_kernel"
for name, key, fn, alg in (("scan_kernel<0,2> (count pass)", "void scan_kernel<0, 2>", "scan_traffic.json", 258472108), ("chain_kernel", "chain_kernel", "chain_traffic.json", None)):
    if key not in out["FETCH_SIZE"]:
        print("kernel not in this run:", key)
        continue
    f, w = out["FETCH_SIZE"][key]["mean"], out["WRITE_SIZE"][key]["mean"]
    d = {"kernel": name, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "dispatches": out["FETCH_SIZE"][key]["dispatches"], "correction": corr,
         "hbm_bytes_per_launch": (2 * f + w) * 1024, "source": src}
    if alg:
        d["algorithmic_bytes_per_launch"] = alg
    json.dump(d, open(os.path.join(ROOT, "profiles", fn), "w"), indent=1)
    print(fn, d["hbm_bytes_per_launch"], d["dispatches"])
print("pack_kernel calibration: FETCH_SIZE KiB", out["FETCH_SIZE"].get("pack_kernel"))
