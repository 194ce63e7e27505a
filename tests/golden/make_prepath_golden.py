"""Golden vectors of the crop pre-path (SimpleHRNet.py:236-278), generated with the reference's own resampler.

torchvision is not installed here, so the reference's ``self.transform`` is rebuilt from what its four transforms do
(oracle/prepath_oracle.py) around the REAL ``PIL.Image.resize`` -- the only non-trivial arithmetic in it.  Run in the
build container:  python tests/golden/make_prepath_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import prepath_oracle as P  # noqa: E402


def main():
    import PIL

    rng = np.random.default_rng(20240924)
    # a frame with structure (gradients + noise) so that resampling errors are visible
    hf, wf = 240, 320
    yy, xx = np.mgrid[0:hf, 0:wf]
    frame = np.stack([(xx * 255 // wf), (yy * 255 // hf), ((xx + yy) * 255 // (hf + wf))], -1).astype(np.int32)
    frame = np.clip(frame + rng.integers(-40, 41, frame.shape), 0, 255).astype(np.uint8)
    dets = np.array([
        [30.4, 20.5, 110.5, 200.49, 0.9, 0.9, 0],    # tall: pad x
        [100.5, 50.5, 300.2, 120.7, 0.8, 0.8, 0],    # wide: pad y (ties round to even)
        [10.0, 10.0, 74.0, 106.0, 0.7, 0.7, 0],      # exact 3:2 = the target itself: no padding, no resampling (96x64)
        [0.2, 0.3, 319.6, 239.7, 0.6, 0.6, 0],       # the whole frame
        [200.0, 100.0, 212.0, 190.0, 0.5, 0.5, 0],   # very thin: large pad, strong downscale in y
    ], dtype=np.float32)
    h, w = 96, 64
    images, boxes = P.prepath(frame, dets, h, w, resize=P.pil_resize)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "prepath_240x320_to_96x64.npz")
    np.savez_compressed(out, frame=frame, dets=dets, images=images, boxes=boxes, h=h, w=w,
                        pillow_version=np.array(PIL.__version__))
    print("wrote", out, images.shape, boxes.tolist())


if __name__ == "__main__":
    main()
