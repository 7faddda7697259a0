#!/usr/bin/env python3
"""The config-4 pipeline + corpus-encode block of bench.py on its own (progress on stderr, one JSON object per line on stdout)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from fusion_amd import encoders
if "--no-gemm-tuning" not in sys.argv:
    encoders.enable_gemm_tuning()
torch.cuda.set_device(0)
for rec in bench.measure_pipeline4(torch.device("cuda", 0)):
    print(json.dumps(rec), flush=True)
