"""bench.py's `pcie_inclusive` block alone (uploads beside compute: fp32 and uint8 crops; copy / pass / both apart).
usage: python tools/pcie_diag.py"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
pkg = importlib.import_module("simple-hrnet_amd")
dev = torch.device("cuda", 0)
net = pkg.NativeHRNet(48, 17, (384, 288), "bf16", max_batch=256, device=0).load_state_dict(pkg.synth_state_dict(48, 17, 0))
print(json.dumps(bench.pcie_measure(pkg, net, 256, dev)))
