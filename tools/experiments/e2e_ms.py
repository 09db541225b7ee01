"""The end-to-end (batch-1) training step of bench.py's `end_to_end` leg, three times: ms per video."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
for i in range(3):
    print(bench.end_to_end_bench(dev)["ms_per_video"])
