#!/bin/sh
# Round 6: counters of the scoring kernel, packed scan (product) against round 5's v_alignbit scan (lab bench, reserved[3] = 5)
sh profiles/pmc_pf.sh r06_pack > /dev/null 2>&1
sh profiles/pmc_pf.sh r06_alignbit --reserved 0 0 0 5 > /dev/null 2>&1
