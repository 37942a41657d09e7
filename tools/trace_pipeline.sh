#!/bin/bash
# kernel timeline of the pipelined bench loop: rocprofv3 --kernel-trace of a short run, then tools/pipeline_timeline.py
#   gpurun -- 'bash tools/trace_pipeline.sh [extra bench args]'
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/trace_pipe
rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --no-secondary --steps 400 "$@" > $out/bench.log 2>&1
grep -h '^{"metric"' $out/bench.log | cut -c1-300
python3 tools/pipeline_timeline.py $out/trace 200 100 300 > $out/timeline.txt 2>&1
cat $out/timeline.txt
rm -rf $out/trace
