"""The pooling kernel alone at a chip-filling size (2048 (observation, camera) units), the launch bench.py's
roofline_pool.saturated times: run under `rocprofv3 --kernel-trace --stats` to get the kernel's own duration."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    # (under a counter pass the profiler serialises the launches: the per-launch byte counters are what that pass is for)
    io16 = "--fp32-edges" not in sys.argv
    print(json.dumps(bench.pool_saturated(dev, torch.float16, units=2048, reps=20, io16=io16)))
