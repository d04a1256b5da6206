#!/bin/bash
cd $GRAFT_REPO_ROOT
for N in 4 8; do
  for sc in weak strong; do
    timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500+N)) bench.py --gpus $N --steps 4 --warmup 1 --repeats 2 --dist-backend gloo --scaling $sc --verify > gpurun_out/reh_${N}_$sc.json 2> gpurun_out/reh_${N}_$sc.err
    echo "N=$N $sc rc=$? $(grep -c 'verify ok' gpurun_out/reh_${N}_$sc.err) $(grep -o 'verify ok.*' gpurun_out/reh_${N}_$sc.err | head -1) $(python -c "import json;d=json.load(open('gpurun_out/reh_${N}_$sc.json'));print(d['n_gpus'],d['scaling'],round(d['ms_per_step'],3))" 2>&1 | tail -1)"
  done
done
