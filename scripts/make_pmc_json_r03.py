"""Rebuild profiles/r03_pmc_traffic.json (and its copy bench_pmc_r03.json, which bench.py reads on the GPU box, where
profiles/ does not travel) from the rocprofv3 passes of scripts/profile_r03.sh under gpurun_out/prof: FETCH_SIZE,
WRITE_SIZE and the MFMA-busy pass, averaged per launch of each kernel; notes of the previous file are kept.
usage: python scripts/make_pmc_json.py [gpurun_out/prof]"""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict

here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "gpurun_out", "prof")


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def counters(pass_name):
    acc = defaultdict(lambda: [0.0, set()])
    for f in glob.glob(os.path.join(prof, pass_name, "**", "*_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = (short(row["Kernel_Name"]), row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"])
            acc[k][1].add(row["Dispatch_Id"])
    return {k: (v[0] / max(1, len(v[1])), len(v[1]), v[0]) for k, v in acc.items()}


def stats(pass_name):
    out = {}
    for f in glob.glob(os.path.join(prof, pass_name, "**", "*_kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            out[short(row["Name"])] = (float(row["AverageNs"]), int(row["Calls"]))
    return out


fetch, write, mfma, st = counters("pmc_fetch"), counters("pmc_write"), counters("pmc_mfma"), stats("stats_infer")
path = os.path.join(here, "profiles", "r03_pmc_traffic.json")
doc = json.load(open(path))


def find(prefix, table):   # kernel names in the notes are prefixes of the demangled names
    for k in table:
        name = k[0] if isinstance(k, tuple) else k
        if name.startswith(prefix) or prefix.startswith(name):
            return name
    return None


def busy(name):
    b, g = mfma.get((name, "SQ_VALU_MFMA_BUSY_CYCLES")), mfma.get((name, "GRBM_GUI_ACTIVE"))
    return round(8.0 * b[2] / (g[2] * 1024.0), 4) if b and g and g[2] > 0 else None


main = find(doc["rocprof_kernel_name"], fetch)
doc["kernel_source_sha256"] = hashlib.sha256(open(os.path.join(here, doc["kernel_source"]), "rb").read()).hexdigest()
doc["launches"] = fetch[(main, "FETCH_SIZE")][1]
doc["FETCH_SIZE_kb_per_launch_raw"] = round(fetch[(main, "FETCH_SIZE")][0], 1)
doc["WRITE_SIZE_kb_per_launch"] = round(write[(main, "WRITE_SIZE")][0], 1)
doc["traffic_bytes_per_launch"] = int(round((2 * fetch[(main, "FETCH_SIZE")][0] + write[(main, "WRITE_SIZE")][0]) * 1024))
doc["rocprof_avg_launch_ns"] = int(round(st[find(doc["rocprof_kernel_name"], st)][0]))
doc["mfma_pipe_busy_fraction"] = busy(main)
for k, d in doc.get("other_kernels", {}).items():
    name = find(k.replace("conv1_pool_kernel<true>", "_ZN12_GLOBAL__N_117conv1_pool_kernelILb1E"), fetch)
    if name is None:
        continue
    if "FETCH_SIZE_kb_raw" in d and (name, "FETCH_SIZE") in fetch:
        d["FETCH_SIZE_kb_raw"] = round(fetch[(name, "FETCH_SIZE")][0], 1)
    if "WRITE_SIZE_kb" in d and (name, "WRITE_SIZE") in write:
        d["WRITE_SIZE_kb"] = round(write[(name, "WRITE_SIZE")][0], 1)
    sname = find(name, st)
    if "rocprof_avg_launch_ns" in d and sname:
        d["rocprof_avg_launch_ns"] = int(round(st[sname][0]))
    if "mfma_pipe_busy_fraction" in d and busy(name) is not None:
        d["mfma_pipe_busy_fraction"] = busy(name)
for p in (path, os.path.join(here, "bench_pmc_r03.json")):
    json.dump(doc, open(p, "w"), indent=1)
    open(p, "a").write("\n")
print(json.dumps({k: v for k, v in doc.items() if k != "other_kernels"}, indent=1))
