# every workload's bench line (no profiler), printed against profiles/pmc_latest.json as committed.   usage: tools/bench_lines.sh TAG
TAG=${1:-rXX}; mkdir -p gpurun_out/$TAG
run() { N=$1; shift; timeout 900 python bench.py "$@" > gpurun_out/$TAG/bench$N.json 2> gpurun_out/$TAG/bench$N.err; echo "bench$N rc=$?"; python - <<PY
import json
d=json.load(open("gpurun_out/$TAG/bench$N.json")); print("  ", d["value"], d["unit"], d["ms_per_step"], (d.get("parity") or {}).get("ok"), d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("traffic_source"))
PY
}
run "" 
run _camera_k20 --steps 20 --warmup 5
run _camera_mesh --steps 100 --warmup 20 --with-mesh
run _lidar --workload lidar --steps 100 --warmup 10
run _decay --workload decay --steps 120 --warmup 24
run _multicam --workload multicam --steps 100 --warmup 20 --cameras 4
run _multicam8 --workload multicam --steps 100 --warmup 20 --cameras 8
run _node --workload node
python -c "import __graft_entry__ as g; g.smoke()"
