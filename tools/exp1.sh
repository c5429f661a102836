for mt in 32 64 128 256; do echo "BF_MVDR_TILE=$mt"; BF_MVDR_TILE=$mt python tools/run_das.py --algo mvdr --iters 10 | tail -1; done
for pt in 1024 2048 4096 8192; do for mt in 32 128; do echo "BF_PIPE_TILE=$pt BF_MVDR_TILE=$mt"; BF_PIPE_TILE=$pt BF_MVDR_TILE=$mt python tools/run_das.py --algo mvdr --iters 10 | tail -1; done; done
for pt in 0 1024 2048 4096; do echo "phase BF_PIPE_TILE=$pt"; BF_PIPE_TILE=$pt python tools/run_das.py --algo phase --iters 10 | tail -1; done
