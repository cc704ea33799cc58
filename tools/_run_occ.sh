#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05ae
timeout 120 ./tools/_lds_occ > gpurun_out/r05ae/occ.txt 2>&1
cat gpurun_out/r05ae/occ.txt
