# Rehearsal of the N>1 code path on a one-GPU box: 2 ranks share GPU 0, gloo instead of RCCL.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in c2 c5; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
     bench.py --gpus 2 --steps 2 --warmup 1 --config $cfg --backend gloo --share-device > gpurun_out/rehearse_$cfg.json 2> gpurun_out/rehearse_$cfg.err || { tail -20 gpurun_out/rehearse_$cfg.err; exit 1; }
  python - <<PY
import json
j=json.loads(open("gpurun_out/rehearse_$cfg.json").read().strip().splitlines()[-1]); print("$cfg", j["n_gpus"], j["value"], j["config"]["M_total"], j.get("top1") or j.get("last_batch"))
PY
done
