#!/bin/bash
# every PMC summary profiles/ keeps, in one GPU call: headline k = 16, configs[4]'s K = 32, radius, ICP step, RANSAC loop
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tools/pmc_kernel.sh knn16 "knn_union_ix_kernel<16" python3 tools/prof_knn.py 1e6 1e6 16 3 > gpurun_out/pmc_knn16.out 2>&1 && echo knn16 ok && \
tools/pmc_kernel.sh knn32 "knn_union_ix_kernel<32" python3 tools/prof_knn.py 1e7 1.25e7 32 2 > gpurun_out/pmc_knn32.out 2>&1 && echo knn32 ok && \
tools/pmc_kernel.sh radius "radius_union_kernel,prov_sort_kernel" python3 tools/prof_radius.py 1e6 1e6 3 > gpurun_out/pmc_radius.out 2>&1 && echo radius ok && \
tools/pmc_kernel.sh icp "icp_step_kernel" python3 tools/prof_icp.py 10 > gpurun_out/pmc_icp.out 2>&1 && echo icp ok && \
tools/pmc_kernel.sh ransac "icp_step_kernel,ransac_words_draw_kernel,ransac_draw_loop_kernel,ransac_first_kernel" python3 tools/prof_icp_ransac.py > gpurun_out/pmc_ransac.out 2>&1 && echo ransac ok
