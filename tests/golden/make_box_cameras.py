"""Writes tests/golden/box_cameras.json: camera poses (nerfstudio camera-to-world, OpenGL axes) and pinhole
intrinsics of four frames of the reference's bundled dataset description
``src/pogs/data/utils/datasets/box/transforms.json`` -- data only (numbers read with the json module), so that GPU
tests can drive the host glue on the reference's own camera convention without the reference tree.
usage: python tests/golden/make_box_cameras.py /root/reference"""
import json
import os
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
src = os.path.join(ref, "src/pogs/data/utils/datasets/box/transforms.json")
frames = json.load(open(src))["frames"]
pick = [0, 7, 14, 21]
out = {"source": "src/pogs/data/utils/datasets/box/transforms.json, frames %s" % pick,
       "frames": [{k: frames[i][k] for k in ("fl_x", "fl_y", "cx", "cy", "w", "h", "transform_matrix")} for i in pick]}
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "box_cameras.json")
json.dump(out, open(dst, "w"), indent=1)
print(dst, len(out["frames"]), "frames")
