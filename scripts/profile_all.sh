#!/bin/bash
# Run ON THE GPU BOX (one gpurun call): scripts/profile_gpu.sh for every workload of profiles/hbm_traffic.json.
#   scripts/profile_all.sh r06      -> gpurun_out/prof_r05_{c2,sparse,t4,fullwin,c3,c3t4,c1,c1_sparse}/
R=${1:-r06}
cd "$(dirname "$0")/.."
run() { local tag=$1; shift; local t0=$(date +%s); scripts/profile_gpu.sh ${R}_$tag "$@" > /dev/null 2>&1; echo "$tag $(( $(date +%s) - t0 ))s" | tee -a gpurun_out/profile_all.log; }
rm -f gpurun_out/profile_all.log
run c2
run fullwin --carrier-window 0 -1
run sparse --mix sparse
run c1 --config c1
run c1_sparse --config c1 --mix sparse
run t4 --templates 4 --batch 32768
run c3 --config c3
run c3t4 --config c3 --templates 4 --batch 4096
