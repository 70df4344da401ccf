#!/bin/bash
# usage: tools/kstats.sh <tag> <command...>   -- rocprofv3 kernel stats of a command, top kernels printed and the
# csv kept under gpurun_out/kstats_<tag>.csv (run through gpurun from the repo root)
R=${GRAFT_REPO_ROOT:-$PWD}
tag=$1; shift
mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ks_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -o x -- "$@" > /tmp/ks_$tag.log 2>&1
f=$(find /tmp/ks_$tag -name '*kernel_stats.csv' | head -1)
if [ -z "$f" ]; then echo "no stats produced"; tail -20 /tmp/ks_$tag.log; exit 1; fi
python $R/tools/summarize_profile.py $f $R/gpurun_out/kstats_$tag.csv "$tag: $*"
head -${KSTATS_TOP:-14} $R/gpurun_out/kstats_$tag.csv
