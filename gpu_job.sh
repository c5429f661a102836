cd /root/repo
python bench.py --steps 40 --warmup 5 > gpurun_out/bench1.json 2> gpurun_out/bench1.err; cut -c1-300 gpurun_out/bench1.json; python -c "
import json; d=json.load(open('gpurun_out/bench1.json')); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d.get('extra'), d['cpu_baseline']['value'])"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu --no-extra 2>&1 | tail -1 | cut -c1-200
