#!/bin/bash
# Copy the artifacts of a record session (tools/gpu_session.sh <tag> tests smoke pmc sh:cp bench trace profile bench:... in the order
# of profiles/r06_final_record_run_summary.log) from gpurun_out/ into profiles/<prefix>_*:  tools/collect_record.sh <tag> <prefix>
set -eu
T=$1; P=profiles/$2; G=gpurun_out
tail -n 1 $G/bench_$T.log > ${P}_bench.json
i=0
for name in b16 v2_b4 sdxl_base sd15_control streams3 split; do
  f=$(ls $G/bench_${T}_*.log | sort -V | sed -n "$((i + 1))p"); i=$((i + 1))
  tail -n 1 $f > ${P}_bench_$name.json
done
cp $G/hbm_traffic_$T.json ${P}_hbm_traffic.json
cp $G/step_timeline_$T.txt ${P}_step_timeline_original.txt
cp $G/kernel_stats_$T.csv ${P}_bench_kernel_stats.csv
cp $G/op_profile_$T.txt ${P}_op_profile.txt
tail -n 25 $G/pytest_$T.log > ${P}_pytest_tail.log
cp $G/summary_$T.log ${P}_record_run_summary.log
python - "$P" <<'PY'
import json, sys
p = sys.argv[1]
d = json.load(open(p + "_bench.json"))
r = d["roofline"]
print("bench:", d["ms_per_step"], "ms", d["value"], d["unit"], "frac", r["frac"], "lib", d.get("library", d.get("config", {}).get("library")))
for n in ["b16", "v2_b4", "sdxl_base", "sd15_control", "streams3", "split"]:
    e = json.load(open(f"{p}_bench_{n}.json"))
    print(n, e["ms_per_step"], "ms", e["value"], e["unit"], "frac", e["roofline"]["frac"])
PY
