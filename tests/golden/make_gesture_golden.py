#!/usr/bin/env python3
"""Regenerates tests/golden/gesture_bridge.json in the BUILD container (reads /root/reference, which never travels):
the "this"/"that" point annotations of the reference's four bundled examples (__assets__/Bridge_example/*/data.txt,
the inputs of test_code/inference.py) and checksums of the gesture maps this repo's cv2-free rasteriser
(this_and_that_vdm_amd/gesture_map.py) produces from them at the reference resolution (256 x 384, 14 frames, dilate on).
cv2 is absent here, so the expected values pin the rasteriser against regressions on the reference's real inputs; they are
not a cv2 bit-parity claim.   python tests/golden/make_gesture_golden.py"""
import hashlib
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from this_and_that_vdm_amd import gesture_map as gm  # noqa: E402

SRC = "/root/reference/__assets__/Bridge_example"
H, W, FRAMES = 256, 384, 14


def summarise(cond):
    u8 = np.clip(cond * 255.0, 0, 255).round().astype(np.uint8)
    return dict(sha256_u8=hashlib.sha256(u8.tobytes()).hexdigest(), sum=float(cond.astype(np.float64).sum()),
                min=float(cond.min()), max=float(cond.max()))


def main():
    import PIL.Image
    out = {"height": H, "width": W, "frames": FRAMES, "examples": {}}
    for name in sorted(os.listdir(SRC)):
        d = os.path.join(SRC, name)
        pts = gm.read_points_file(os.path.join(d, "data.txt"))
        with PIL.Image.open(os.path.join(d, "im_0.jpg")) as im:
            ow, oh = im.size
        cond, frames, coords = gm.rasterise_points(pts, (oh, ow), H, W, FRAMES, dilate=True, flip=False)
        flipped, _, _ = gm.rasterise_points(pts, (oh, ow), H, W, FRAMES, dilate=True, flip=True)
        out["examples"][name] = dict(points=[list(p) for p in pts], org_hw=[oh, ow], frames=frames,
                                     coords=[list(c) for c in coords], map=summarise(cond), map_flipped=summarise(flipped),
                                     per_frame_sum={str(f): float(cond[f].astype(np.float64).sum()) for f in frames})
    with open(os.path.join(REPO, "tests", "golden", "gesture_bridge.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote gesture_bridge.json:", {k: v["map"]["sha256_u8"][:12] for k, v in out["examples"].items()})


if __name__ == "__main__":
    main()
