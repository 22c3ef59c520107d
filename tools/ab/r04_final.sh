#!/bin/bash
# round 4: the evidence on the final sources -- suite, profiles, PMC of the headline and of config 5, every launch form of bench.py
cd "$GRAFT_REPO_ROOT"
bash tools/ab/gpu_suite.sh
# (the PMC captures first: the bench lines written afterwards then carry pmc.stale = false)
bash tools/pmc_bench.sh gpurun_out/pmc_bench r04 > gpurun_out/pmc_bench.log 2>&1; tail -3 gpurun_out/pmc_bench.log
python tools/merge_pmc.py gpurun_out/pmc_bench/summary.json
bash tools/pmc_trees.sh gpurun_out/pmc_trees r04 > gpurun_out/pmc_trees.log 2>&1; tail -3 gpurun_out/pmc_trees.log
cp profiles/hbm_traffic.json gpurun_out/hbm_traffic.json
bash tools/refresh_profiles.sh r04 2>&1 | tail -2
FR_LS_PIPELINE=0 bash tools/pmc_kernels.sh gpurun_out/r04_pmc_verify linesearch_verify_kernel -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-e2e > gpurun_out/r04_pmc_verify.log 2>&1
bash tools/ab/r04_bench.sh 2>&1 | tail -14
python bench.py --steps 20 --warmup 5 > gpurun_out/r04b/bench_final.json 2> /dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r04b/bench_final.json').read().strip().splitlines()[-1]); print('final', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['pmc']['stale'], d['limiter']['frac'], d['roofline']['hbm_frac_measured'])"
python bench.py --measure trees --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('trees', d['value'], d['pmc']['stale'], d['roofline']['traffic'], d['limiter'])"
