#!/usr/bin/env python3
"""Turn one profiling session (bench JSON + rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE counter passes) into the
files kept under profiles/.  Usage:
  python tools/profile_summary.py <session dir> <tag>      e.g.  gpurun_out/prof_c r01_c
The session dir holds bench_default.json, stats/**/_kernel_stats.csv, fetch/**/_counter_collection.csv and
write/**/_counter_collection.csv as written by the commands quoted in the generated summary."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(pattern):
    hits = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    if not hits:
        raise SystemExit(f"nothing matches {pattern}")
    return hits[-1]          # the newest: gpurun merges every session into the same local directory


def short(name):
    name = name.replace("void ", "")
    cut = name.find("(")
    return (name if cut < 0 else name[:cut])[:64]


def counter_mb(path, counter):
    """mean Counter_Value (KB on gfx950 for FETCH_SIZE / WRITE_SIZE) per launch, by kernel name -> MB"""
    acc = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                a = acc[short(r["Kernel_Name"])]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    return {k: (n, v / n / 1024.0) for k, (n, v) in acc.items()}


def main():
    sess, tag = sys.argv[1], sys.argv[2]
    out = os.path.join(ROOT, "profiles")
    bench = json.loads(open(os.path.join(sess, "bench_default.json")).read().strip().splitlines()[-1])
    stats_csv = one(os.path.join(sess, "stats", "**", "*_kernel_stats.csv"))
    shutil.copy(stats_csv, os.path.join(out, f"{tag}_kernel_stats.csv"))
    with open(os.path.join(out, f"{tag}_bench.json"), "w") as f:
        f.write(json.dumps(bench) + "\n")
    rows = list(csv.DictReader(open(stats_csv)))
    steps = 13
    total_ns = sum(float(r["TotalDurationNs"]) for r in rows)
    fetch = counter_mb(one(os.path.join(sess, "fetch", "**", "*_counter_collection.csv")), "FETCH_SIZE")
    write = counter_mb(one(os.path.join(sess, "write", "**", "*_counter_collection.csv")), "WRITE_SIZE")

    L = []
    rf, cb = bench["roofline"], bench["cpu_baseline"]
    L.append(f"# Profile {tag} (1x MI355X, {bench['config']['workload']}, per-GPU batch {bench['config']['per_gpu_batch']})\n")
    L.append(f"Headline (`python bench.py`, `{tag}_bench.json`): **{bench['value']:.0f} {bench['unit']}**, {bench['ms_per_step']} ms/step, "
             f"{bench['model_tflops_per_gpu']} TFLOP/s of model FLOPs ({100 * bench['mfma_frac_of_peak_step']:.1f} % of the 2516 TFLOP/s bf16 MFMA peak over the whole step); "
             f"CPU oracle {cb['value']} {cb['unit']} on {cb['cores']} host cores ({cb['sample']}).\n")
    L.append(f"`roofline`: `{rf['kernel']}` — {rf['achieved']} {rf['unit']} algorithmic = {100 * rf['frac']:.1f} % of peak, average launch {rf['avg_launch_us']} us by HIP events.\n")
    L.append("## Kernel time\n")
    L.append("`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-clock-probe --profile-steps 0 --single-stream` "
             f"({steps} steps; `--single-stream` merges the two modality streams so a kernel's duration is not stretched by a concurrently running one; raw CSV: `{tag}_kernel_stats.csv`).\n")
    L.append("| kernel | calls | avg us | total ms | % of GPU time |\n|---|---|---|---|---|")
    for r in rows[:26]:
        L.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['TotalDurationNs']) / 1e6:.2f} | {r['Percentage']} |")
    L.append(f"\nSum of kernel time {total_ns / 1e6 / steps:.2f} ms/step.\n")
    L.append("## Per-kernel pricing from the bench line (HIP events)\n")
    L.append("| family | launches/step | avg us | share | achieved | of peak |\n|---|---|---|---|---|---|")
    for k, v in bench["kernels"].items():
        L.append(f"| {k} | {v['launches_per_step']} | {v['avg_us']} | {100 * v['share_of_kernel_time']:.1f} % | {v['achieved']} {v['unit']} | {100 * v['frac']:.1f} % |")
    L.append("\n## HBM traffic per launch (PMC)\n")
    L.append("Separate `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes over `bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-clock-probe --profile-steps 0`; "
             "counter unit KB; FETCH_SIZE doubled (gfx950 tallies the 128-B requests of 16-B/lane streams as 64 B, MI355X_MICROARCH.md §HBM); per launch, "
             "averaged over all shapes of the kernel in a step.\n")
    L.append("| kernel | launches sampled | 2 x FETCH (MB) | WRITE (MB) | HBM MB / launch |\n|---|---|---|---|---|")
    names = sorted(set(fetch) & set(write), key=lambda k: -(2 * fetch[k][1] + write[k][1]) * fetch[k][0])
    for k in names[:14]:
        n, fm = fetch[k]
        _, wm = write[k]
        L.append(f"| `{k}` | {n} | {2 * fm:.1f} | {wm:.1f} | {2 * fm + wm:.1f} |")
    L.append("")
    # per bench family (bench.py prices families, rocprofv3 lists kernel symbols): HBM bytes per launch, launch-weighted
    fam_of = (("gemm_big_kernel<false, false", "gemm_big_nt"), ("gemm_big_kernel<false, true", "gemm_big_nn"), ("gemm_big_kernel<true, true", "gemm_big_tn+splitk"),
              ("attn_fwd_kernel", "attn_fwd"), ("ln_fwd_kernel", "layernorm_fwd"), ("ln_bwd_kernel", "layernorm_bwd"), ("cast_kernel", "cast_f32_bf16"),
              ("patchify_kernel", "patchify"))
    fam = defaultdict(lambda: [0, 0.0])
    for k in set(fetch) & set(write):
        for pat, name in fam_of:
            if pat in k:
                n = fetch[k][0]
                fam[name][0] += n
                fam[name][1] += n * (2 * fetch[k][1] + write[k][1]) * 1e6
    traffic = {name: round(v[1] / v[0]) for name, v in fam.items() if v[0]}
    with open(os.path.join(out, f"{tag}_traffic.json"), "w") as f:
        json.dump({"unit": "HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc, launch-weighted over the family's kernels)",
                   "per_gpu_batch": bench["config"]["per_gpu_batch"], "traffic": traffic}, f, indent=1)
        f.write("\n")
    L.append(f"Per bench family (`{tag}_traffic.json`, read by bench.py for `roofline.traffic`): " + ", ".join(f"{k} {v / 1e6:.0f} MB" for k, v in sorted(traffic.items())) + "\n")
    mf = sorted(glob.glob(os.path.join(sess, "mfma", "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if mf:
        acc, cnt, dur, seen = defaultdict(lambda: defaultdict(float)), defaultdict(int), defaultdict(float), set()
        with open(mf[-1]) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"])
                    cnt[k] += 1
                    dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        L.append("## MFMA-pipe utilisation and wave states (PMC)\n")
        L.append("`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU` over "
                 "`bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-clock-probe --profile-steps 0 --single-stream`.  MFMA utilisation = "
                 "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs): the gfx94x `MfmaUtil` formula with GRBM_GUI_ACTIVE, which rocprofv3 reports "
                 "summed over the 8 XCDs, brought back to one clock domain (the implied clock is 2.2-2.3 GHz).  It counts EXECUTED matrix work at the clock the kernel "
                 "actually ran at: padding tiles (N = 513 in 64 / 128-row tiles) and the S / dP recomputation of the two-kernel attention backward are included, which is why it "
                 "sits above the algorithmic fractions in the table above.  `parked` = SQ_WAIT_ANY / SQ_WAVE_CYCLES (s_waitcnt, barriers), `issue-stalled` = "
                 "SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, `VALU` = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES.\n")
        L.append("| kernel | launches | avg us | MFMA-pipe utilisation | waves parked | issue-stalled | VALU active |\n|---|---|---|---|---|---|---|")
        for k in sorted(acc, key=lambda k: -dur[k])[:9]:
            a = acc[k]
            gui, wc = a.get("GRBM_GUI_ACTIVE", 0.0), a.get("SQ_WAVE_CYCLES", 0.0)
            if gui <= 0 or wc <= 0:
                continue
            util = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 1024.0)
            L.append(f"| `{k}` | {cnt[k]} | {dur[k] / cnt[k] / 1e3:.1f} | {100 * util:.1f} % | {100 * a.get('SQ_WAIT_ANY', 0) / wc:.0f} % | "
                     f"{100 * a.get('SQ_WAIT_INST_ANY', 0) / wc:.0f} % | {100 * a.get('SQ_ACTIVE_INST_VALU', 0) / wc:.0f} % |")
        L.append("")
    with open(os.path.join(out, f"{tag}_summary.md"), "w") as f:
        f.write("\n".join(L))
    print("\n".join(L))


if __name__ == "__main__":
    main()
