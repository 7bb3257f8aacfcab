#!/bin/bash
# PMC passes (profiles/collect_pmc.sh) for the three workloads profiles/r02_pmc_traffic.json holds; summarise each with
# python profiles/pmc_summarize.py gpurun_out/pmc_<key> <key> profiles/r02_pmc_traffic.json
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 bash profiles/collect_pmc.sh H_f32 --config H
timeout -k 10 400 bash profiles/collect_pmc.sh C3_f32 --config C3
timeout -k 10 500 bash profiles/collect_pmc.sh C5_f32 --config C5
echo ALLDONE
