#!/bin/bash
# after tools/gpu_round_c.sh <tag> (under gpurun) has merged its output into gpurun_out/: copy what is judged into profiles/
# usage: bash tools/copy_profiles.sh <tag> [round prefix, default r03]
set -euo pipefail
tag=$1; r=${2:-r03}; G=gpurun_out; P=profiles
cp $G/${tag}_headline_b128/kernel_stats.csv $P/${r}_rocprofv3_kernel_stats.csv
grep '^{' $G/${tag}_headline_b128/bench_under_rocprof.json > $P/${r}_bench_under_rocprof.json
cp $G/${tag}_headline_b128/pmc_sq_summary.txt $P/${r}_pmc_sq_summary_b128.txt
cp $G/${tag}_headline_b128/pmc_hbm_summary.txt $P/${r}_pmc_hbm_summary_b128.txt
cp $G/${tag}_headline_b1/kernel_stats.csv $P/${r}_rocprofv3_kernel_stats_batch1.csv
grep '^{' $G/${tag}_headline_b1/bench_under_rocprof.json > $P/${r}_bench_under_rocprof_batch1.json
cp $G/${tag}_headline_b1/pmc_sq_summary.txt $P/${r}_pmc_sq_summary_batch1.txt
cp $G/${tag}_headline_b1/pmc_hbm_summary.txt $P/${r}_pmc_hbm_summary_batch1.txt
for n in headline c1 c2 c3 c4 c5 c3_fir8; do grep '^{' $G/${tag}_round/bench_$n.json > $P/${r}_bench_$n.json; done
grep '^{' $G/${tag}_round/bench_rehearsal2.json > $P/${r}_bench_rehearsal_2ranks_one_gpu_gloo.json
python3 - "$P" "$r" "$tag" <<'PY'
import json, re, sys
P, r, tag = sys.argv[1:4]
def parse(path):
    d = {}
    for line in open(path):
        m = re.match(r"(\w+)\s+([\d.]+)\s+\(avg", line)
        if m:
            d[m.group(1)] = float(m.group(2))
    return d
j = json.load(open(f"{P}/{r}_hbm_traffic.json"))
for idx, name in enumerate(("b128", "batch1")):
    c = parse(f"{P}/{r}_pmc_hbm_summary_{name}.txt")
    m = j["measurements"][idx]
    m["run"] = re.sub(r"gpurun_out/\w+?_headline", f"gpurun_out/{tag}_headline", m["run"])
    m.update(fetch_size_kb_raw=c["FETCH_SIZE"], write_size_kb=c["WRITE_SIZE"], tcc_hit=c["TCC_HIT_sum"], tcc_miss=c["TCC_MISS_sum"],
             tcc_ea0_rdreq=c["TCC_EA0_RDREQ_sum"], l2_hit_rate=c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
             traffic_bytes_per_launch=int(128 * c["TCC_MISS_sum"] + 1024 * c["WRITE_SIZE"]))
    if "two_x_fetch_plus_write_bytes" in m:
        m["two_x_fetch_plus_write_bytes"] = int(2048 * c["FETCH_SIZE"] + 1024 * c["WRITE_SIZE"])
    print(name, m["traffic_bytes_per_launch"], round(m["l2_hit_rate"], 4))
json.dump(j, open(f"{P}/{r}_hbm_traffic.json", "w"), indent=2)
for n in ("headline", "c1", "c2", "c3", "c4", "c5", "c3_fir8"):
    d = json.load(open(f"{P}/{r}_bench_{n}.json"))
    print(n, round(d["value"]), round(d["roofline"]["kernel_ms"], 3), round(d["valu"]["frac"], 3), d["parity"]["ok"])
PY
