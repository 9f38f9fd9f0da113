#!/bin/sh
# Round 6: HIP / kernel trace of the exchange step through a one-rank RCCL communicator (profiles/exchange_trace.py); no counters.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/xtrace
NCCL_DEBUG=WARN rocprofv3 --kernel-trace --hip-trace --rccl-trace --memory-copy-trace --output-format csv -d $O/xtrace -o x -- python3 $R/profiles/exchange_trace.py run > $O/r06_exchange_trace_run.txt 2>&1
python3 $R/profiles/exchange_trace.py report $O/xtrace > $O/r06_exchange_trace.txt 2>&1
grep -v "^[EW]2026\|NCCL WARN\|^$" $O/r06_exchange_trace_run.txt >> $O/r06_exchange_trace.txt
rm -rf $O/xtrace
cat $O/r06_exchange_trace.txt
