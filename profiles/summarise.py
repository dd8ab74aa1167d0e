#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of profiles/collect.sh (merged into gpurun_out/) into the tracked summaries:
profiles/<tag>_full_kernel_stats.csv, <tag>_pmc_hbm.md, hbm_traffic.json.  Usage: summarise.py r02a"""
import csv, collections, json, os, shutil, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go = os.path.join(root, "gpurun_out")
shutil.copy(os.path.join(go, "prof_%s" % tag, "%s_kernel_stats.csv" % tag), os.path.join(root, "profiles", "%s_full_kernel_stats.csv" % tag))
with open(os.path.join(go, "prof_%s.log" % tag)) as f:
    lines = [l for l in f if l.startswith('{"metric"')]
open(os.path.join(root, "profiles", "%s_full_bench_under_rocprof.json" % tag), "w").write(lines[-1])
bj = os.path.join(go, "bench_%s.json" % tag)
if os.path.exists(bj):
    shutil.copy(bj, os.path.join(root, "profiles", "%s_full_bench.json" % tag))

def agg(path):
    acc = collections.defaultdict(float); n = collections.Counter(); seen = set()
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('edsx::', '').split('<')[0]
        acc[k] += float(r['Counter_Value'])
        key = (k, r['Dispatch_Id'])
        if key not in seen:
            seen.add(key); n[k] += 1
    return {k: acc[k] / n[k] for k in acc}
f = agg(os.path.join(go, "pmc_%s_fetch" % tag, "p_counter_collection.csv"))
w = agg(os.path.join(go, "pmc_%s_write" % tag, "p_counter_collection.csv"))
out = ["# HBM traffic per launch from rocprofv3 PMC counters (build %s, 1000 x 100 Mb workload)" % tag, "",
       "Collected in two separate passes (`rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`,",
       "`python3 bench.py --steps 1 --warmup 1 --cpu-baseline-mb 0`, script `profiles/collect.sh`), averaged over the",
       "launches of each kernel.  FETCH_SIZE/WRITE_SIZE are reported in KB.  On gfx950 FETCH_SIZE counts a 128-B request",
       "as 64 B for wide coalesced streaming reads (MI355X_MICROARCH.md, HBM section), so the read side is doubled;",
       "WRITE_SIZE is exact.", "",
       "| kernel | FETCH_SIZE raw (GB) | read, corrected x2 (GB) | WRITE_SIZE (GB) | total (GB) |", "|---|---|---|---|---|"]
tj = {}
for k in sorted(f, key=lambda k: -(f[k] + w.get(k, 0))):
    if k.startswith('k_synth'):
        continue
    fr = f[k] * 1024 / 1e9; wr = w.get(k, 0) * 1024 / 1e9
    if fr + wr < 0.02:
        continue
    out.append("| %s | %.2f | %.2f | %.2f | %.2f |" % (k, fr, 2 * fr, wr, 2 * fr + wr))
    tj[k] = round((2 * f[k] + w.get(k, 0)) * 1024)
out += ["", "k_scan_extract: algorithmic bytes = 1000 x 1e8 cells = 100.0 GB read once; the counters say %.1f GB read +" % (2 * f['k_scan_extract'] * 1024 / 1e9),
        "%.1f GB written (grouping records of the runs the scan groups itself + the variant columns it hands on).  No re-reads." % (w['k_scan_extract'] * 1024 / 1e9)]
open(os.path.join(root, "profiles", "%s_pmc_hbm.md" % tag), "w").write("\n".join(out) + "\n")
json.dump({"1000x100000000": dict(tj, source="profiles/%s_pmc_hbm.md" % tag)}, open(os.path.join(root, "profiles", "hbm_traffic.json"), "w"), indent=1)
print("\n".join(out))

# ---- SQ wave counters (where the waves of each kernel spend their time)
sqp = os.path.join(go, "pmc_%s_sq" % tag, "p_counter_collection.csv")
if os.path.exists(sqp):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(sqp)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('edsx::', '')
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    o2 = ["# SQ wave counters per kernel (build %s, 1000 x 100 Mb workload, summed over the launches of one bench run)" % tag, "",
          "`rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU",
          "SQ_INSTS_VALU SQ_ACTIVE_INST_LDS` (a pass of its own).  WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = ready but not",
          "issued, ACTIVE_INST_ANY = issuing; the three add up to WAVE_CYCLES (quad-cycle units).", "",
          "| kernel | wave cycles | parked (WAIT_ANY) | issue stall (WAIT_INST_ANY) | issuing (ACTIVE_INST_ANY) | of it VALU | of it LDS | VALU instructions |",
          "|---|---|---|---|---|---|---|---|"]
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        wc = v.get("SQ_WAVE_CYCLES", 0)
        if wc < 1e8 or k.startswith("k_synth"):
            continue
        pc = lambda n: "%.0f %%" % (100.0 * v.get(n, 0) / wc)
        o2.append("| %s | %.3g | %s | %s | %s | %s | %s | %.3g |" % (k, wc, pc("SQ_WAIT_ANY"), pc("SQ_WAIT_INST_ANY"), pc("SQ_ACTIVE_INST_ANY"),
                                                                  pc("SQ_ACTIVE_INST_VALU"), pc("SQ_ACTIVE_INST_LDS"), v.get("SQ_INSTS_VALU", 0)))
    open(os.path.join(root, "profiles", "%s_pmc_sq.md" % tag), "w").write("\n".join(o2) + "\n")
    print("\n".join(o2))
