#!/bin/bash
# Round profile of bench.py on the GPU box:  scripts/profile_round.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of the bench command (timed region + stand-alone stages; the 600-iteration late phase and the side lines are switched off so
#      that the per-kernel averages are those of the headline regime)                              -> gpurun_out/<tag>_stats/
#   2. separate --pmc passes (kernel-trace only, one counter group each)                           -> gpurun_out/<tag>_pmc_<counter>/
#   3. steady-state traffic passes (12 timed steps after 6 warm-up steps)                          -> gpurun_out/<tag>_pmc_FETCH_SIZE / WRITE_SIZE (scripts/traffic_pass.sh)
# Copy the summaries to profiles/ afterwards (scripts/collect_profile.py <tag>; gpurun_out/ is scratch).
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
FLAGS="--late 0 --whole-run 0 --no-side-lines"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- python3 $R/bench.py --steps 20 --warmup 5 $FLAGS > $R/gpurun_out/${tag}_bench.log 2>&1
for grp in "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  t=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${tag}_pmc_$t -- python3 $R/bench.py --steps 12 --warmup 6 --no-cpu-baseline $FLAGS > $R/gpurun_out/${tag}_pmc_$t.log 2>&1
done
for grp in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${tag}_pmc_$grp -- python3 $R/bench.py --steps 12 --warmup 6 --no-cpu-baseline $FLAGS > $R/gpurun_out/${tag}_pmc_$grp.log 2>&1
done
# 4. the LATE phase of a run (round 4: VERDICT r03 item 1 — `frac` of the late phase must reproduce from a profile taken there): the same command with 900 warm-up
#    iterations; scripts/collect_profile.py lists every launch of the dominant kernel, the last 20 are the timed steps
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}late_stats -- python3 $R/bench.py --steps 20 --warmup 900 --no-cpu-baseline $FLAGS > $R/gpurun_out/${tag}late_bench.log 2>&1
# 5. the rasteriser at a trained scene's tile-instance count
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}raster_stats -- python3 $R/bench.py --config raster > $R/gpurun_out/${tag}raster_bench.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
  t=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${tag}raster_pmc_$t -- python3 $R/bench.py --config raster > $R/gpurun_out/${tag}raster_pmc_$t.log 2>&1
done
find $R/gpurun_out -name "*.db" -delete
# collect HERE (the raw kernel traces are far beyond the 64 MiB gpurun copies back), then drop the raw directories: copy gpurun_out/<tag>_profiles/* to profiles/ afterwards
for t in ${tag} ${tag}late ${tag}raster; do python3 $R/scripts/collect_profile.py $t --out $R/gpurun_out/${tag}_profiles > $R/gpurun_out/${tag}_profiles_$t.log 2>&1; done
rm -rf $R/gpurun_out/${tag}_stats $R/gpurun_out/${tag}late_stats $R/gpurun_out/${tag}raster_stats $R/gpurun_out/${tag}_pmc_* $R/gpurun_out/${tag}raster_pmc_*
tail -1 $R/gpurun_out/${tag}_bench.log | cut -c1-400
