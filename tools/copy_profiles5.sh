#!/bin/bash
# after tools/gpu_round5.sh <tag> (under gpurun) has merged its output into gpurun_out/<tag>: copy what is judged into profiles/r05_*
# usage: bash tools/copy_profiles5.sh <tag>
set -euo pipefail
tag=$1; G=gpurun_out/$tag; P=profiles; r=r05
declare -A NAME=( [default]=default_exact_b128 [fast]=fast_b128 [batch1]=exact_batch1 [c3fir8]=c3_fir8_fast [c5]=c5_exact )
for c in "${!NAME[@]}"; do
  d=$G/$c; n=${NAME[$c]}
  [ -d $d ] || continue
  grep '^{' $d/bench.json > $P/${r}_bench_$n.json
  grep '^{' $d/bench_under_rocprof.json > $P/${r}_bench_under_rocprof_$n.json
  cp $d/kernel_stats.csv $P/${r}_rocprofv3_kernel_stats_$n.csv
  cp $d/kernel_stats_warm.csv $P/${r}_rocprofv3_kernel_stats_warm_$n.csv
  cp $d/pmc_summary.txt $P/${r}_pmc_summary_$n.txt
done
python3 - "$P" "$r" "$tag" <<'PY'
import json, re, sys
P, r, tag = sys.argv[1:4]
def parse(path):
    d, kernels = {}, ""
    for line in open(path):
        m = re.match(r"(\w+)\s+([\d.]+)\s+\(avg", line)
        if m:
            d[m.group(1)] = float(m.group(2))
        if line.startswith("# kernels:"):
            kernels = line[len("# kernels:"):].strip()
    return d, kernels
B_ALG = 33976320
out = {"source": f"tools/gpu_round5.sh (rocprofv3 --pmc, one counter group per pass, counters only, MI355X), round 5: the same box and the same gpurun call as the bench lines beside it (gpurun_out/{tag})",
       "workload": "headline: 256 mics x 128x128 x 256",
       "note": "traffic = 128 B x TCC_MISS_sum + 1024 B x WRITE_SIZE (calibration: profiles/r03_fetch_calibration.txt)",
       "measurements": []}
for name, frames, math in (("default_exact_b128", 128, "exact"), ("fast_b128", 128, "fast"), ("exact_batch1", 1, "exact")):
    c, kernels = parse(f"{P}/{r}_pmc_summary_{name}.txt")
    out["measurements"].append({"frames_per_step": frames, "math": math, "kernel": kernels,
                                "traffic_bytes_per_launch": int(128 * c["TCC_MISS_sum"] + 1024 * c["WRITE_SIZE"]),
                                "algorithmic_bytes_per_launch": B_ALG * frames, "tcc_miss": c["TCC_MISS_sum"], "tcc_hit": c["TCC_HIT_sum"],
                                "write_size_kb": c["WRITE_SIZE"], "fetch_size_kb": c["FETCH_SIZE"]})
json.dump(out, open(f"{P}/{r}_hbm_traffic.json", "w"), indent=1)
# the table of profiles/README.md: per configuration, B_alg x frames / warm-median kernel time / 8 TB/s beside the bench line's own fraction
import csv
ARGS = {"default_exact_b128": "the driver's command", "fast_b128": "`--math fast`", "exact_batch1": "`--batch 1 --steps 200 --warmup 20`",
        "c3_fir8_fast": "`--workload c3 --interp fir8 --math fast`", "c5_exact": "`--workload c5`, one rank's slab, 1024 frames in flight"}
rows_out = ["| config (files `r05_*_<config>.*`) | math | frames / launch | frames/s (bench, not profiled) | ms/step | kernel ms (HIP events, bench) | kernel ms (rocprofv3, warm median) | `B_alg`·frames ÷ warm median ÷ 8 TB/s | bench `roofline.frac` | `valu.frac` | VALU instr. / launch | HBM-side traffic ÷ algorithmic bytes | dominant kernel (+ its pre-pass, warm mean) |",
            "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for name in ("default_exact_b128", "fast_b128", "exact_batch1", "c3_fir8_fast", "c5_exact"):
    b = json.load(open(f"{P}/{r}_bench_{name}.json"))
    rows = list(csv.DictReader(open(f"{P}/{r}_rocprofv3_kernel_stats_warm_{name}.csv")))
    k = next(x for x in rows if "das_" in x["Name"])
    pre = [x for x in rows if "pack_" in x["Name"]]
    c, _ = parse(f"{P}/{r}_pmc_summary_{name}.txt")
    alg = b["roofline"]["achieved"] * 1e9 * b["roofline"]["kernel_ms"] * 1e-3  # the line's own algorithmic bytes per launch
    med = float(k["WarmMedianNs"]) * 1e-9
    frames = b["config"].get("frames_per_step", b["config"].get("batch", 0))
    valu = c.get("SQ_INSTS_VALU", 0)
    kname = re.sub(r"^void awpu::|\(.*$", "", k["Name"])
    prepass = f" (+ `{re.sub(r'^void awpu::|^awpu::|[(].*$', '', pre[0]['Name'])}` {float(pre[0]['WarmAverageNs']) / 1e3:.1f} µs)" if pre else ""
    rows_out.append(f"| `{name}` ({ARGS[name]}) | {b['config'].get('math')} | {frames} | {b['value']:,.0f} | {b['ms_per_step']:.4f} | {b['roofline']['kernel_ms']:.4f} | {med * 1e3:.4f} | "
                    f"**{alg / med / 8e12:.4f}** | {b['roofline']['frac']:.4f} | {b.get('valu', {}).get('frac', float('nan')):.3f} | "
                    f"{valu / 1e9:.2f} G | {(128 * c['TCC_MISS_sum'] + 1024 * c['WRITE_SIZE']) / 1e9:.3g} / {alg / 1e9:.3g} GB | `{kname}`{prepass} |")
table = "\n".join(rows_out)
print(table)
readme = open(f"{P}/README.md").read()
begin, end = "<!-- r05 table begin -->", "<!-- r05 table end -->"
if begin in readme:
    readme = readme[:readme.index(begin) + len(begin)] + "\n" + table + "\n" + readme[readme.index(end):]
    open(f"{P}/README.md", "w").write(readme)
PY
