#!/bin/bash
# PMC passes (profiles/collect_pmc.sh) for the three workloads profiles/r05_pmc_traffic.json holds; summarise each with
# python profiles/pmc_summarize.py gpurun_out/pmc_<key> <key> profiles/r05_pmc_traffic.json
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 bash profiles/collect_pmc.sh H_f32 --config H
timeout -k 10 600 bash profiles/collect_pmc.sh H_f64 --config H --dtype f64
timeout -k 10 400 bash profiles/collect_pmc.sh C3_f32 --config C3
timeout -k 10 500 bash profiles/collect_pmc.sh C5_f32 --config C5
for k in H_f32 H_f64 C3_f32 C5_f32; do python3 profiles/pmc_summarize.py gpurun_out/pmc_$k $k gpurun_out/r05_pmc_traffic.json > /dev/null; rm -rf gpurun_out/pmc_$k/*/*/*.db; done
echo ALLDONE
