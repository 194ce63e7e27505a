#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r4_10; mkdir -p $out
python - > $out/bits.txt 2>&1 <<'PY'
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, '.')
pkg = importlib.import_module("simple-hrnet_amd")
outs = []
for ts in ("0", "0.06"):
    os.environ["HRN_TAIL_SMALL"] = ts
    net = pkg.NativeHRNet(48, 17, (384, 288), "bf16", max_batch=256, device=0).load_state_dict(pkg.synth_state_dict(48, 17, 0))
    crops = torch.from_numpy(pkg.synth_crops(256, 384, 288, seed=61)).cuda()
    hm = net(crops).cpu().numpy(); net.close(); outs.append(hm)
print("tail-small bit-identical:", np.array_equal(outs[0], outs[1]), float(np.abs(outs[0]-outs[1]).max()))
PY
cat $out/bits.txt | tail -2
tools/envsweep.sh $out/sweep "HRN_TAIL_SMALL=0" "HRN_TAIL_SMALL=0.03" "HRN_TAIL_SMALL=0.06" "HRN_TAIL_SMALL=0.1" "HRN_TAIL_SMALL=0" "HRN_TAIL_SMALL=0.03" "HRN_TAIL_SMALL=0.06" "HRN_TAIL_SMALL=0.1"
