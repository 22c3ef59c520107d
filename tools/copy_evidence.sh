#!/bin/bash
# After `gpurun -- tools/ab/r05_final.sh <tag>`: copy what the run left under gpurun_out/ into profiles/ (the judged copies).
# usage (repo root, this container): bash tools/copy_evidence.sh r05
set -e
TAG=${1:-r06}
cp gpurun_out/profiles_new/${TAG}_* profiles/
cp gpurun_out/r04b/bench_final.json profiles/${TAG}_bench.json
cp gpurun_out/r04b/bench.json profiles/${TAG}_bench_driver_form.json
cp gpurun_out/r04b/bench_threads2.json profiles/${TAG}_bench_threads_2ctx_1gpu.json
cp gpurun_out/r04b/bench_threads8.json profiles/${TAG}_bench_threads_8ctx_1gpu.json
cp gpurun_out/r04b/bench_torch2_gloo.json profiles/${TAG}_bench_selflaunched_torch_2rank_1gpu_gloo.json
cp gpurun_out/r04b/bench_torch2_nccl.json profiles/${TAG}_bench_torchrun_2rank_1gpu_rccl_refused_gloo.json
cp gpurun_out/r04b/bench_trees.json profiles/${TAG}_bench_trees.json
for k in mslr hard ties tiesmix hardties; do cp gpurun_out/${TAG}_kinds/bench_$k.json profiles/${TAG}_kinds_bench_$k.json; done
cp gpurun_out/${TAG}_kinds.txt profiles/${TAG}_kinds.txt
cp gpurun_out/${TAG}_gputest.log profiles/${TAG}_gputest.log
cp gpurun_out/hbm_traffic.json profiles/hbm_traffic.json
cp gpurun_out/pmc_bench/summary.json profiles/${TAG}_pmc_bench_summary.json
cp gpurun_out/pmc_trees/summary.json profiles/${TAG}_pmc_trees_summary.json
cp gpurun_out/${TAG}_pmc_verify/summary.txt profiles/${TAG}_pmc_verify_summary.txt
cp gpurun_out/${TAG}_fullrank_by_tick.txt profiles/${TAG}_fullrank_by_tick.txt
