#!/bin/bash
# das at the 1024-frame JACK period (the headline batch's samples): BF_DAS_SPLIT2048 = 3 (a wavefront per frame), 2 (64-lane split), 0 (generic)
for sp in 3 2 0; do for lay in planar interleaved; do echo -n "split=$sp $lay: "; BF_DAS_SPLIT2048=$sp python tools/run_das.py --hop 1024 --frames 32768 --iters 20 --layout $lay 2>/dev/null | tail -1; done; done
echo -n "split=3 M=4: "; python tools/run_das.py --hop 1024 --frames 32768 --mics 4 --iters 20 2>/dev/null | tail -1
echo -n "split=3 M=16: "; python tools/run_das.py --hop 1024 --frames 32768 --mics 16 --iters 10 2>/dev/null | tail -1
