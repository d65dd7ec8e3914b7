import sys, os, importlib, json
sys.path.insert(0, '/root/repo')
os.chdir('/root/repo')
import torch
import bench
rt = importlib.import_module("vg-renderer_amd.runtime")
ctx = rt.Context(0)
print(json.dumps(bench.frame_leg(rt, torch, ctx, 0), indent=1))
