#!/bin/bash
# A/B of library variants built by tools/build_variants.sh on the first-call probe: tools/ab_first_call.sh early noearly -- 250 1000
libs=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do libs+=("$1"); shift; done; shift
for rep in 1 2; do for d in "${libs[@]}"; do echo "== $d (round $rep)"; NBODY_HIP_SO=$PWD/tools/exp/$d/libnbody_hip.so python tools/first_call_probe.py "$@" | cut -c1-86; done; done
