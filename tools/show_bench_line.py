import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launches_timed"], d["roofline"]["stage"])
print(json.dumps(d["steady_state"])[:1800])
print(d["multi_gpu_workload_on_1_gpu"]["value"], d["pcie_inclusive"]["value"], d["cpu_baseline"]["value"])
print(json.dumps(d["roofline_extra"]["marching_cubes"]))
