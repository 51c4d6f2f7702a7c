#!/bin/bash
# round 4, first GPU call: data for two decisions — (1) cached lower bounds vs the ball test, (2) the rasteriser at a trained scene's tile-instance count
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python3 scripts/cache_bound_probe.py 20 50 24 > gpurun_out/r04a_cache_probe.log 2>&1
timeout 300 python3 bench.py --config raster > gpurun_out/r04a_raster_line.json 2> gpurun_out/r04a_raster.err
timeout 300 python3 bench.py --config raster --log-scale -3.6 > gpurun_out/r04a_raster_line_m36.json 2>> gpurun_out/r04a_raster.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04a_raster_stats -- python3 $R/bench.py --config raster > $R/gpurun_out/r04a_raster_prof.log 2>&1
cd $R
find gpurun_out/r04a_raster_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04a_raster_kernel_stats.csv
find gpurun_out/r04a_raster_stats -name "*.db" -delete; find gpurun_out/r04a_raster_stats -name "*kernel_trace.csv" -delete
tail -3 gpurun_out/r04a_cache_probe.log; cut -c1-600 gpurun_out/r04a_raster_line.json; head -12 gpurun_out/r04a_raster_kernel_stats.csv
