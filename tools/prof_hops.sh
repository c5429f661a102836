export TMPDIR=/tmp
for cfg in "mvdr 256 131072" "phase 256 131072" "phase 1024 32768" "mvdr 1024 32768"; do
  set -- $cfg
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ph_$1_$2 -- python tools/run_das.py --algo $1 --hop $2 --frames $3 --iters 5 --warmup 3 --settle-ms 0 > gpurun_out/ph_$1_$2.log 2>&1
  echo "== $1 hop $2"; for f in $(find gpurun_out/ph_$1_$2 -name "*kernel_stats*"); do head -6 $f | cut -d, -f1-4 | cut -c1-170; done
done
