#!/bin/bash
# usage: tools/run_sweep.sh N "v,k,w v,k,w ..." variant-dirs...
n=$1; shapes=$2; shift 2
for d in "$@"; do
  NBODY_HIP_SO=$PWD/tools/exp/$d/libnbody_hip.so timeout -k 10 300 python tools/sweep.py $n $shapes || exit 1
done
