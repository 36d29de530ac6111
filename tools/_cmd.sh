mkdir -p gpurun_out/r03d
timeout -k 10 300 python bench.py --workload soup10m --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r03d/soup.json 2> gpurun_out/r03d/soup.err || { tail gpurun_out/r03d/soup.err; exit 2; }
python - <<PY
import json
j=json.load(open("gpurun_out/r03d/soup.json"))
print(j["value"])
for k,b in j["batches"].items(): print(k, {x:(round(b[x],2) if isinstance(b[x],float) else b[x]) for x in ("order_ms","walk_ms","replay_ms","mrays_per_s","retraced_rays")})
PY
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r03d/dragon.json 2> gpurun_out/r03d/dragon.err || exit 3
python - <<PY
import json
j=json.load(open("gpurun_out/r03d/dragon.json"))
r=j["roofline"]
print(j["ms_per_step"], j["cold_ms_first_step"], r["exclusive_pass"]["intersect_ms"], r["exclusive_pass"]["advance_ms"])
PY

