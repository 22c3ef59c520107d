#!/bin/bash
# The short form of r06_final.sh after a change that leaves the headline kernel alone: suite + smoke, PMC of the headline (so that the
# bench line written afterwards carries pmc.stale = false), the final bench line, the five data kinds on one box, the full-ranking
# measures by tick on the data kinds.
cd "$GRAFT_REPO_ROOT"; TAG=${1:-r06}
bash tools/ab/gpu_suite.sh
cp gpurun_out/suite/gputest.log gpurun_out/${TAG}_gputest.log
bash tools/pmc_bench.sh gpurun_out/pmc_bench $TAG > gpurun_out/pmc_bench.log 2>&1; tail -3 gpurun_out/pmc_bench.log
python tools/merge_pmc.py gpurun_out/pmc_bench/summary.json
cp profiles/hbm_traffic.json gpurun_out/hbm_traffic.json
mkdir -p gpurun_out/r04b
python bench.py --steps 20 --warmup 5 > gpurun_out/r04b/bench_final.json 2> /dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r04b/bench_final.json').read().strip().splitlines()[-1]); print('final', d['value'], d['value_runs_min_median_max'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['pmc']['stale'], d['limiter']['frac'], d['roofline']['hbm_frac_measured'], d['cpu_baseline']['min_median_max'], d.get('side'))"
tools/ab/r05_kinds.sh ${TAG}_kinds 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_kinds.txt; cat gpurun_out/${TAG}_kinds.txt
bash tools/ab/r06_fullrank_by_tick.sh $TAG
rm -rf gpurun_out/pmc_bench/sq gpurun_out/pmc_bench/mix gpurun_out/pmc_bench/fetch gpurun_out/pmc_bench/write
find gpurun_out -name "*.csv" -size +2M -delete
du -sh gpurun_out
