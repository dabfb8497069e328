#!/bin/bash
# Kernel timeline of the last blocks of a pipelined CPM link:   bash tools/link_timeline.sh <out name> [link_pipe_run.py flags]
name=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/tl_$name
rocprofv3 --kernel-trace --output-format csv -d $out -o kt -- python3 tools/link_pipe_run.py "$@" > $out.log 2> $out.err
kt=$(find $out -name '*kernel_trace.csv' | head -1)
python3 tools/timeline.py $kt 40 > gpurun_out/timeline_$name.txt
cat $out.log >> gpurun_out/timeline_$name.txt
rm -rf $out
