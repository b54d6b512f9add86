#!/bin/bash
# after tools/gpu_round.sh <tag> (under gpurun) has merged its output into gpurun_out/: copy what is judged into profiles/
# usage: bash tools/copy_profiles.sh <tag> [round prefix, default r04]
set -euo pipefail
tag=$1; r=${2:-r04}; G=gpurun_out; P=profiles
for pair in "b128:" "b1:_batch1" "exact:_exact"; do
  t=${pair%%:*}; sfx=${pair##*:}
  d=$G/${tag}_headline_$t
  [ -d $d ] || continue
  cp $d/kernel_stats.csv $P/${r}_rocprofv3_kernel_stats$sfx.csv
  cp $d/kernel_stats_warm.csv $P/${r}_rocprofv3_kernel_stats_warm$sfx.csv
  grep '^{' $d/bench_under_rocprof.json > $P/${r}_bench_under_rocprof$sfx.json
  cp $d/pmc_sq_summary.txt $P/${r}_pmc_sq_summary${sfx:-_b128}.txt
  cp $d/pmc_hbm_summary.txt $P/${r}_pmc_hbm_summary${sfx:-_b128}.txt
done
for n in headline c1 c2 c3 c4 c5 c3_fir8 headline_exact; do
  [ -f $G/${tag}_round/bench_$n.json ] && grep '^{' $G/${tag}_round/bench_$n.json > $P/${r}_bench_$n.json || true
done
[ -f $G/${tag}_round/bench_rehearsal2.json ] && grep '^{' $G/${tag}_round/bench_rehearsal2.json > $P/${r}_bench_rehearsal_2ranks_one_gpu_gloo.json || true
python3 - "$P" "$r" "$tag" <<'PY'
import json, os, re, sys
P, r, tag = sys.argv[1:4]
def parse(path):
    d = {}
    for line in open(path):
        m = re.match(r"(\w+)\s+([\d.]+)\s+\(avg", line)
        if m:
            d[m.group(1)] = float(m.group(2))
    return d
cases = [("b128", 128, "awpu::das_quad_kernel<false, 0>", f"gpurun_out/{tag}_headline_b128 (bench.py defaults)", "fast"),
         ("batch1", 1, "awpu::das_quadh_kernel<1, false> (+ pack_halves_kernel, not in the figure)", f"gpurun_out/{tag}_headline_b1 (--batch 1)", "fast"),
         ("exact", 128, "awpu::das_exact_quad_kernel", f"gpurun_out/{tag}_headline_exact (--math exact)", "exact")]
out = {"source": "tools/gpu_profile.sh -> tools/pmc_hbm.sh (rocprofv3 --pmc, one counter group per pass, counters only, MI355X), round 4",
       "workload": "headline: 256 mics x 128x128 x 256",
       "note": "traffic = 128 B x TCC_MISS_sum + 1024 B x WRITE_SIZE: calibrated on the kernels' own access patterns (profiles/r03_fetch_calibration.txt, "
               "tools/microbench/fetch_calib.hip): 64-byte scalar table requests are counted exactly by FETCH_SIZE, 16-byte-per-lane LDS-DMA at half; "
               "one missed 128-byte line is one TCC_MISS in both.",
       "measurements": []}
for name, frames, kernel, run, math in cases:
    path = f"{P}/{r}_pmc_hbm_summary_{name}.txt"
    if not os.path.exists(path):
        continue
    c = parse(path)
    out["measurements"].append({
        "frames_per_step": frames, "math": math, "kernel": kernel, "run": run,
        "fetch_size_kb_raw": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"], "tcc_hit": c["TCC_HIT_sum"], "tcc_miss": c["TCC_MISS_sum"],
        "tcc_ea0_rdreq": c["TCC_EA0_RDREQ_sum"], "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
        "traffic_bytes_per_launch": int(128 * c["TCC_MISS_sum"] + 1024 * c["WRITE_SIZE"]),
        "two_x_fetch_plus_write_bytes": int(2048 * c["FETCH_SIZE"] + 1024 * c["WRITE_SIZE"])})
    print(name, out["measurements"][-1]["traffic_bytes_per_launch"], round(out["measurements"][-1]["l2_hit_rate"], 4))
json.dump(out, open(f"{P}/{r}_hbm_traffic.json", "w"), indent=2)
for n in ("headline", "c1", "c2", "c3", "c4", "c5", "c3_fir8", "headline_exact"):
    path = f"{P}/{r}_bench_{n}.json"
    if os.path.exists(path):
        d = json.load(open(path))
        print(n, round(d["value"]), round(d["roofline"]["kernel_ms"], 3), round(d["valu"]["frac"], 3), d["parity"]["ok"], d["parity"]["max_rel_unfloored"])
PY
