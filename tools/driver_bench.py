#!/usr/bin/env python3
"""Write a folder of synthetic 768x568 BMP files and run ia_driver on it in both flows."""
import json, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bmp_util, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
d = tempfile.mkdtemp(prefix="iadrv")
os.makedirs(d + "/in"); os.makedirs(d + "/out")
base = synth.rgb_batch("C", 768, 568, 8)
for i in range(n):
    bmp_util.write_bmp24(f"{d}/in/cyl{i:04d}.BMP", np.roll(base[i % 8], i, axis=2))
G16 = [1 / 16, 2 / 16, 1 / 16, 2 / 16, 4 / 16, 2 / 16, 1 / 16, 2 / 16, 1 / 16]
cfg = {"image_folder": d + "/in", "input_image_filter": "", "saveIntermediateImages": False, "saveFinalImages": True,
       "calc_MSQE": False, "output_dir": d + "/out", "extract_channel": "grey",
       "operations": [{"type": "histogram-equalization"}, {"type": "gaussian-noise", "std_dev": -1, "mean": -1},
                      {"type": "salt-and-pepper", "intensity": 50},
                      {"type": "linear-filter", "kernel": {"width": 3, "height": 3, "values": G16}},
                      {"type": "median-filter", "kernel": {"width": 3, "height": 3, "values": [1] * 9}},
                      {"type": "sobel-filter"}]}
json.dump(cfg, open(d + "/cfg.json", "w"))
exe = os.path.join(ROOT, "imageanalysis_amd", "lib", "ia_driver")
for mode in ["--per-image", "--batch"]:
    t0 = time.time()
    r = subprocess.run([exe, d + "/cfg.json", mode], capture_output=True, text=True)
    print(mode, "wall %.2f s" % (time.time() - t0))
    print("\n".join(l for l in r.stdout.splitlines() if "entire batch" in l or "reading" in l or "running all" in l or "copying back" in l or "grayscale" in l or "Sobel" in l or "median" in l))
