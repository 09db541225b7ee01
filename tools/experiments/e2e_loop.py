"""Runs the end-to-end training step (bench.py's end_to_end leg) N times -- the program to put under rocprofv3."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
print(bench.end_to_end_bench(torch.device("cuda:0"), steps=n))
