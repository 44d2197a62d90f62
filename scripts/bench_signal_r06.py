#!/usr/bin/env python3
"""The signal sub-result of bench.py alone (C5 size + signal_long), as one JSON object."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(json.dumps(bench.signal_subresult(torch.device("cuda:0"), False), indent=1))
