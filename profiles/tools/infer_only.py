"""The sliding-window inference leg of bench.py alone (for rocprofv3): python3 profiles/tools/infer_only.py [volume] [dtype]"""
import json, sys, pathlib, argparse
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import bench

a = argparse.Namespace(inference_size=int(sys.argv[1]) if len(sys.argv) > 1 else 512, size=128)
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
print(json.dumps(bench.inference_leg(a, torch.device("cuda:0"), dt)))
