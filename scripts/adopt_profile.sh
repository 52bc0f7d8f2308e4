# Adopt gpurun_out/<tag>/ (written by gpu_profile_round.sh) as the committed profile set: profiles/r02_final_* + pmc_latest.json.
TAG=${1:-r03_final}
cd gpurun_out/$TAG && cp bench_kernel_stats.csv ../../profiles/${TAG}_bench_kernel_stats.csv && cp bench_line.json ../../profiles/${TAG}_bench_line.json \
 && cp bench_line_under_rocprof.json ../../profiles/${TAG}_bench_line_under_rocprof.json && cp pmc_summary.json ../../profiles/${TAG}_pmc_by_kernel.json \
 && cp pmc_stage_summary.json ../../profiles/pmc_latest.json && cd ../.. && python3 -c "
import sys; sys.path.insert(0,'.')
import bench; print('PMC stamp matches the checkout:', bool(bench._pmc_traffic()))"
