#!/bin/bash
# usage: tools/trace_step.sh <tag> <bench args...>: rocprofv3 kernel trace of a bench line -> per-step busy / idle (tools/gap_trace.py)
R=${GRAFT_REPO_ROOT:-$PWD}
tag=$1; shift
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tr_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -o x -- python $R/bench.py --no-cpu-baseline "$@" > /tmp/tr_$tag.log 2>&1
f=$(find /tmp/tr_$tag -name '*kernel_trace.csv' | head -1)
if [ -z "$f" ]; then echo "no trace"; tail -5 /tmp/tr_$tag.log; exit 1; fi
echo "== $tag: $*"
python $R/tools/gap_trace.py $f ${GAPS:-14} ${SEQ:-}
