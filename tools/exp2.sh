for v in 0 1 2; do for i64 in 1 0; do echo "mvdr BF_STFT_VARIANT=$v BF_ISTFT_F64=$i64"; BF_MVDR_TILE=128 BF_STFT_VARIANT=$v BF_ISTFT_F64=$i64 python tools/run_das.py --algo mvdr --iters 10 | tail -1; done; done
for v in 0 2; do echo "phase BF_STFT_VARIANT=$v"; BF_STFT_VARIANT=$v python tools/run_das.py --algo phase --iters 10 | tail -1; done
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
