#!/bin/bash
# builds and runs scripts/micro/ldlt_time.hip on the GPU box; LDLT_PRINT_CLK=1 prints wave 0's phase stamps of the 8-camera case
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I ucoslam-cv3_amd/csrc ${LDLT_PRINT_CLK:+-DLDLT_PRINT_CLK=1} scripts/micro/ldlt_time.hip ucoslam-cv3_amd/csrc/ctx.hip -o /tmp/ldlt_time 2>&1 | grep -E "error" ; /tmp/ldlt_time
