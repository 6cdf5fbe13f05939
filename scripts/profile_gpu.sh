#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats + PMC passes of bench.py.
# Usage: scripts/profile_gpu.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
# PMC passes are separate runs with --kernel-trace only (never with sys/hip traces).
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
# raw rocprofv3 output (tens of MB per pass) stays on the box; only the summaries, the kernel
# stats CSV and the bench logs go to gpurun_out/ (which is merged back, 64 MiB limit)
O=/tmp/prof_$TAG
K=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O $K
export TMPDIR=/tmp
cd $R
# kernel-trace pass: bench.py's sustained protocol (>= 1 s of steps), so that clocks settle and the per-kernel averages
# are the ones bench.py's HIP events see; PMC passes: 8 steps (counter collection serialises
# every dispatch, data generation included -- a full-length run takes tens of minutes)
# (every pass under its own timeout and logged as it ends: a pass that hangs costs its limit, not the
# whole call, and what finished before it still comes home; PMC passes keep few blocks resident --
# the synthetic-data kernels are serialised and counted too)
BENCH_STATS="python bench.py --streams 1 --cpu-seconds 0 --profile-kernels 0 --legs none --min-seconds 1 $*"
BENCH="python bench.py --streams 1 --steps 8 --warmup 1 --cpu-seconds 0 --profile-kernels 0 --legs none --min-seconds 0 --resident-blocks 1 $*"
T0=$(date +%s)
timeout ${PROF_STATS_LIMIT:-420} rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- $BENCH_STATS > $O/bench_stats.log 2>&1
echo "stats rc=$? $(( $(date +%s) - T0 ))s" >> $K/progress.log; cp $O/bench_stats.log $K/ 2>/dev/null
pass() {  # name counters...
  local name=$1; shift
  local t1=$(date +%s)
  timeout ${PROF_PASS_LIMIT:-300} rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -o $name -- $BENCH > $O/bench_$name.log 2>&1
  echo "$name rc=$? $(( $(date +%s) - t1 ))s" >> $K/progress.log; cp $O/bench_$name.log $K/ 2>/dev/null
}
pass pmc_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass pmc_sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT
pass pmc_grbm GRBM_GUI_ACTIVE GRBM_COUNT
pass pmc_fetch FETCH_SIZE
pass pmc_write WRITE_SIZE
python scripts/summarize_profile.py $O > $O/summary.md 2>&1
cp $O/summary.md $O/hbm_traffic.json $O/bench_*.log $K/ 2>/dev/null
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $K/kernel_stats.csv 2>/dev/null
cat $O/summary.md
