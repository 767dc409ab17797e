import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_step
print(json.dumps(bench_step.run_graphed(2, torch.device("cuda", 0), steps=5, warmup=2), indent=1))
