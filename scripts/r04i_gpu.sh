#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 1800 python3 -m pytest tests -q -m gpu > gpurun_out/r04i_gpu_suite.log 2>&1
tail -6 gpurun_out/r04i_gpu_suite.log
timeout 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
