#!/usr/bin/env python
"""gpurun_out/r01/* (written by tools/collect_profiles.sh) -> small committed summaries under profiles/."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"
SRC = os.path.join(ROOT, "gpurun_out", TAG)
DST = os.environ.get("PROFILES_DST") or os.path.join(ROOT, "profiles")      # (PROFILES_DST: summarise on the GPU box, into gpurun_out)


def pmc_summary(dirs, out):
    agg = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(os.path.join(SRC, d, "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if "rnnt::" in r["Kernel_Name"]:
                    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                    agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
        for (k, c), v in sorted(agg.items()):
            w.writerow([k, c, len(v), round(sum(v) / len(v), 1)])
    return agg


def main():
    os.makedirs(DST, exist_ok=True)
    for c in ("c2", "c3", "c4", "c5"):
        src = os.path.join(SRC, f"bench_{c}.json")
        if os.path.exists(src) and os.path.getsize(src):
            shutil.copy(src, os.path.join(DST, f"{TAG}_bench_{c}.json"))
    for c in ("c2", "c3", "c4"):
        db = os.path.join(SRC, f"stats_{c}", f"{c}_results.db")
        if os.path.exists(db):
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "rocprof_summary.py"), db,
                                   os.path.join(DST, f"{TAG}_rocprof_{c}_kernel_stats.csv")])
    agg = pmc_summary(["pmc_fetch", "pmc_write"], os.path.join(DST, f"{TAG}_rocprof_c4_pmc_hbm.csv"))
    pmc_summary(["pmc_sq1", "pmc_sq2"], os.path.join(DST, f"{TAG}_rocprof_c4_pmc_sq.csv"))
    if glob.glob(os.path.join(SRC, "pmc_gather1", "*_counter_collection.csv")):
        gagg = pmc_summary([f"pmc_gather{i}" for i in range(1, 8)], os.path.join(DST, f"{TAG}_rocprof_c4_pmc_gather_path.csv"))
    else:
        gagg = {}
    # calibration of FETCH_SIZE on a known access pattern: one dword per STRIDE bytes over the c4 tensor
    pagg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(SRC, "pmc_probe", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "k_probe" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                pagg[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
    if pagg:
        with open(os.path.join(DST, f"{TAG}_rocprof_probe_fetch_size.csv"), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["kernel", "grid_size", "dispatches", "mean_FETCH_SIZE_KiB"])
            for (k, g), v in sorted(pagg.items()):
                w.writerow([k, g, len(v), round(sum(v) / len(v), 1)])
    for f in glob.glob(os.path.join(SRC, "table_*.md")):
        shutil.copy(f, os.path.join(DST, f"{TAG}_" + os.path.basename(f)))
    for name in ("parity_errors.json", "lattice_probe.txt", "lattice_routes.txt", "cabi_probe.txt", "wd_trace_c4.txt",
                 "shape_map.md", "bench_c4_blocks_of_8.json", "host_overhead.txt",
                 "bench_c4_logdomain_lattice.json", "bench_c4_rccl_group.json", "bench_c4_cold_start.json",
                 "graph_probe.txt", "lsm_rate_by_size.txt", "bench_c4_n128.json", "ubench_copy_rate.txt",
                 "compact_host_probe.txt", "ubench_gather_variants.txt"):
        if os.path.exists(os.path.join(SRC, name)):
            shutil.copy(os.path.join(SRC, name), os.path.join(DST, f"{TAG}_" + name))
    # HBM traffic of the dominant kernel, per launch (FETCH_SIZE doubled for wide coalesced reads)
    agg3 = pmc_summary(["pmc_fetch_c3", "pmc_write_c3"], os.path.join(DST, f"{TAG}_rocprof_c3_pmc_hbm.csv"))
    doc = {"_about": "HBM bytes per launch of the log-softmax kernel from rocprofv3 --pmc FETCH_SIZE / "
                     "WRITE_SIZE (separate passes, profiles/%s_rocprof_c{3,4}_pmc_hbm.csv); FETCH_SIZE doubled for "
                     "wide coalesced reads as MI355X_MICROARCH.md (HBM) prescribes" % TAG}
    for cfg, table, pats in (("c4", agg, ("k_lsm_regs<", "k_lsm_small<4,0,")), ("c3", agg3, ("k_lsm_large<0,",)),
                             ("c4_lattice_wd", agg, ("::k_lattice_wd<",))):
        def pick(counter):
            ks = [k for k in table if any(k[0].replace(" ", "").find(pat) >= 0 for pat in pats) and k[1] == counter]
            return (ks[0][0], sum(table[ks[0]]) / len(table[ks[0]])) if ks else (None, None)
        (kname, f_kib), (_, w_kib) = pick("FETCH_SIZE"), pick("WRITE_SIZE")
        if f_kib is not None and w_kib is not None:
            doc[cfg] = {"kernel": kname, "fetch_size_kib": f_kib, "write_size_kib": w_kib,
                        "traffic_bytes": f_kib * 2048 + w_kib * 1024,
                        "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/collect_profiles.sh {TAG} "
                                  f"(profiles/{TAG}_rocprof_{cfg[:2]}_pmc_hbm.csv)"}
    # the gather kernel of the loss entry (sparse dword reads: the counter tallies 64 B per request while the memory
    # system moves the whole 128-byte line -- profiles/<tag>_rocprof_probe_fetch_size.csv, the stride probes of
    # tools/ubench/gather_variants.hip run as long as a full read -- so FETCH_SIZE is doubled here too)
    def gpick(counter):
        ks = [k for k in gagg if "k_to_diagonal<true" in k[0].replace(" ", "") and k[1] == counter]
        return sum(gagg[ks[0]]) / len(gagg[ks[0]]) if ks else None
    gf, gw = gpick("FETCH_SIZE"), gpick("WRITE_SIZE")
    if gf is not None and gw is not None:
        doc["c4_gather"] = {"kernel": "rnnt::k_to_diagonal<true>", "fetch_size_kib": gf, "write_size_kib": gw,
                            "traffic_bytes": gf * 2048 + gw * 1024,
                            "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/gather_probe.py, "
                                      f"tools/collect_profiles.sh {TAG} (profiles/{TAG}_rocprof_c4_pmc_gather_path.csv)"}
    json.dump(doc, open(os.path.join(DST, "hbm_traffic.json"), "w"), indent=1)
    # bench.py reads hbm_traffic.json at run time, i.e. the file of the PREVIOUS collection; fill a missing
    # `roofline.traffic` from the PMC passes of this same collection
    for cfg in ("c3", "c4"):
        bj = os.path.join(DST, f"{TAG}_bench_{cfg}.json")
        if cfg in doc and os.path.exists(bj):
            d = json.load(open(bj))
            if d.get("roofline", {}).get("traffic") is None:
                d["roofline"]["traffic"] = doc[cfg]["traffic_bytes"]
                json.dump(d, open(bj, "w"))
    print(sorted(os.listdir(DST)))


if __name__ == "__main__":
    main()
