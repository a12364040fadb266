"""Developer tool: the fixed-work measurement of bench.py alone (one GPU)."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
sys.argv = [sys.argv[0]] + sys.argv[1:]
args = bench.parse()
files, label = bench.model_files(args, 0)
print(json.dumps(bench.fixed_work(files, args.width, args.height, args, 0, 0.07)))
