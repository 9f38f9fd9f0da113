#!/bin/sh
# Round 5: the exact matcher's cross-block merge -- partials polled by the last split's block (epoch-tagged words) against the ticket
# scheme of rounds 1-4 (SFM_MATCH_MERGE=ticket), same box, alternating; sfm_match_soa on descriptor arrays.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
OUT=${1:-$O/r05_match_merge_ab.txt}
: > $OUT
for rep in 1 2 3; do
  for mode in poll ticket; do
    SFM_MATCH_MERGE=$mode MATCH_NO_CPU=1 MATCH_SIZES=512,1024,1500,2048,2500,3000,4096 python3 profiles/match_bench.py 2>/dev/null | python3 -c "
import json,sys
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
print('$mode', ' '.join('%d: %.2f us' % (r['n'], 1e3*r['ms_exact']) for r in rows))" >> $OUT
  done
done
cat $OUT
