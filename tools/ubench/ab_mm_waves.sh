cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/mmw
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "msda" > gpurun_out/mmw/t.log 2>&1; tail -2 gpurun_out/mmw/t.log
ARGS="--no-cpu-baseline --no-fp32 --no-h2d --steps 20 --warmup 5"
show() { python - $1 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).readline())
print(sys.argv[1], d['ms_per_step'], '; '.join(f"{k['name']} {k['avg_us']:.0f}" for k in d['kernels'] if 'msda' in k['name']))
PY
}
python bench.py $ARGS > gpurun_out/mmw/w2.json 2>/dev/null; show gpurun_out/mmw/w2.json
bash gedepth_amd/csrc/build.sh -DMM_WAVES=3 > /dev/null 2>&1
python bench.py $ARGS > gpurun_out/mmw/w3.json 2>/dev/null; show gpurun_out/mmw/w3.json
bash gedepth_amd/csrc/build.sh > /dev/null 2>&1
python bench.py $ARGS > gpurun_out/mmw/w2b.json 2>/dev/null; show gpurun_out/mmw/w2b.json
