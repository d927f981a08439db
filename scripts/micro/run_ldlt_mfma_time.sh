#!/bin/bash
# builds and runs scripts/micro/ldlt_mfma_time.hip on the GPU box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I ucoslam-cv3_amd/csrc scripts/micro/ldlt_mfma_time.hip ucoslam-cv3_amd/csrc/ctx.hip -o /tmp/ldlt_mfma_time 2>&1 | grep -E "error" ; /tmp/ldlt_mfma_time
