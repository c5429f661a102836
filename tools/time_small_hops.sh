#!/bin/bash
# das at the JACK periods below 512 (the headline batch's samples): BF_DAS_INTERLEAVE=1 (64-lane), 2 (half-wavefront), 0 (generic)
for il in 1 3; do
  for hop in 256 128 64; do
    echo -n "il=$il hop=$hop: "
    BF_DAS_INTERLEAVE=$il python tools/run_das.py --hop $hop --frames $((65536*512/hop)) --iters 20 2>/dev/null | tail -1
  done
done
