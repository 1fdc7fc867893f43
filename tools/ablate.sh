for a in 0 16 8 24 4 28; do
  ABO_ABLATE=$a python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/abl_$a.log 2>&1
  python - <<PY
import json
j=json.loads(open("gpurun_out/abl_$a.log").read().strip().splitlines()[-1]); print("ablate=$a", j["roofline"]["avg_launch_ms"], j["roofline"]["achieved"])
PY
done
