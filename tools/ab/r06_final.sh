#!/bin/bash
# round 6 (round 5s script, retagged): the evidence on the current sources -- suite + smoke, PMC of the headline and of config 5 (so that the bench lines
# written afterwards carry pmc.stale = false), every profile of tools/refresh_profiles.sh, the five data kinds on one box,
# the launch forms of bench.py, the audit and the fuzzers.   usage: tools/ab/r06_final.sh [tag]
cd "$GRAFT_REPO_ROOT"; TAG=${1:-r06}
bash tools/ab/gpu_suite.sh
cp gpurun_out/suite/gputest.log gpurun_out/${TAG}_gputest.log
bash tools/pmc_bench.sh gpurun_out/pmc_bench $TAG > gpurun_out/pmc_bench.log 2>&1; tail -3 gpurun_out/pmc_bench.log
python tools/merge_pmc.py gpurun_out/pmc_bench/summary.json
bash tools/pmc_trees.sh gpurun_out/pmc_trees $TAG > gpurun_out/pmc_trees.log 2>&1; tail -3 gpurun_out/pmc_trees.log
cp profiles/hbm_traffic.json gpurun_out/hbm_traffic.json
bash tools/refresh_profiles.sh $TAG 2>&1 | tail -2
FR_LS_PIPELINE=0 bash tools/pmc_kernels.sh gpurun_out/${TAG}_pmc_verify linesearch_verify_kernel -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-e2e --repeats 0 > gpurun_out/${TAG}_pmc_verify.log 2>&1
tools/ab/r05_kinds.sh ${TAG}_kinds 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_kinds.txt; cat gpurun_out/${TAG}_kinds.txt
bash tools/ab/r04_bench.sh 2>&1 | tail -14
python bench.py --steps 20 --warmup 5 > gpurun_out/r04b/bench_final.json 2> /dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r04b/bench_final.json').read().strip().splitlines()[-1]); print('final', d['value'], d['value_runs_min_median_max'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['pmc']['stale'], d['limiter']['frac'], d['roofline']['hbm_frac_measured'], d['cpu_baseline']['min_median_max'])"
bash tools/ab/r06_fullrank_by_tick.sh $TAG > /dev/null
# what travels back is capped at 64 MiB: keep the summaries, drop the raw rocprofv3 output they were made from
rm -rf gpurun_out/pmc_bench/sq gpurun_out/pmc_bench/mix gpurun_out/pmc_bench/fetch gpurun_out/pmc_bench/write \
       gpurun_out/pmc_trees/sq1 gpurun_out/pmc_trees/sq2 gpurun_out/pmc_trees/fetch gpurun_out/pmc_trees/write \
       gpurun_out/${TAG}_pmc_verify/sq1 gpurun_out/${TAG}_pmc_verify/sq2 gpurun_out/${TAG}_pmc_verify/sq3 gpurun_out/${TAG}_pmc_verify/sq4 \
       gpurun_out/${TAG}_pmc_verify/ic gpurun_out/${TAG}_pmc_verify/mem1 gpurun_out/${TAG}_pmc_verify/mem2
find gpurun_out -name "*.csv" -size +2M -delete
du -sh gpurun_out
