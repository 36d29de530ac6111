mkdir -p gpurun_out/r03e
timeout -k 10 600 python -u -m pytest tests/test_gpu_upload.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -15 || exit 1
RTA_WIDE_VERBOSE=1 timeout -k 10 300 python bench.py --workload soup10m --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r03e/soup.json 2> gpurun_out/r03e/soup.err || { tail gpurun_out/r03e/soup.err; exit 2; }
grep -i "upload\|wide" gpurun_out/r03e/soup.err
python - <<PY
import json
j=json.load(open("gpurun_out/r03e/soup.json"))
print(j["value"], j["scene"])
for k,b in j["batches"].items(): print(k, {x:(round(b[x],2) if isinstance(b[x],float) else b[x]) for x in ("order_ms","walk_ms","replay_ms","mrays_per_s","retraced_rays")})
PY
