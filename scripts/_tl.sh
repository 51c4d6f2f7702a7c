cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --no-side-lines --no-cpu-baseline --whole-run 0 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/scripts/timeline.py /tmp/tl 0.0 3 > $GRAFT_REPO_ROOT/gpurun_out/r06as_timeline_all.log
ls /tmp/tl/*/ > $GRAFT_REPO_ROOT/gpurun_out/r06as_ls.log
cd $GRAFT_REPO_ROOT && python bench.py > gpurun_out/r06as_bench.json 2> gpurun_out/r06as_bench.err
