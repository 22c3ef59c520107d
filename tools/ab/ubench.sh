#!/bin/bash
# VALU issue-rate microbenchmark (tools/ubench/valu_rate.hip; build here first: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate)
cd "$GRAFT_REPO_ROOT/tools/ubench" && timeout 300 ./valu_rate | tee ../../gpurun_out/valu_rate.txt
