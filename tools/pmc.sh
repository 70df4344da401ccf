#!/bin/bash
# usage: tools/pmc.sh <kernel-name-substring> "<CTR1 CTR2 ...>" <command...>
# One rocprofv3 --pmc pass (counters only + kernel trace) of a command; prints the per-launch sums of the kernels
# whose name contains the substring.  Run through gpurun from the repo root; repeat per counter group.
R=${GRAFT_REPO_ROOT:-$PWD}
pat=$1; ctrs=$2; shift 2
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pmc_run
rocprofv3 --pmc $ctrs --kernel-trace -d /tmp/pmc_run -o pmc -- "$@" > /tmp/pmc_run.log 2>&1
db=$(find /tmp/pmc_run -name '*.db' | head -1)
if [ -z "$db" ]; then echo "no db"; tail -5 /tmp/pmc_run.log; exit 1; fi
python $R/tools/pmc_read.py $db "$pat"
