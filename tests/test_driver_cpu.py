"""The driver's host side without a GPU: config parsing (comments included), folder listing
with the filter, BMP decoding of the layouts a data folder can hold."""
import json
import os
import subprocess

import numpy as np

import bmp_util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "imageanalysis_amd", "lib", "ia_driver")

CONFIG = """{
  "image_folder": %(inp)s,
  // regex pattern of types to match.  For example 'cyl|inter' will process all files starting with cyl or inter
  "input_image_filter": "cyl|inter",
  "saveIntermediateImages": false,
  "saveFinalImages": true, /* block comment */
  "calc_MSQE": false,
  "output_dir": %(out)s,
  // red, blue, green, or grey
  "extract_channel": "grey",
  "operations": [
    { "type": "gaussian-noise",
      // set to -1 to calculate it from the image
      "std_dev": -1, "mean": -1 },
    { "type": "linear-filter", "kernel": { "width": 3, "height": 3, "values": [1, 1, 1, 10, 1, 1, 1, 1, 1] } },
    { "type": "quantization", "levels": [ { "min": 10, "max": 20, "val": 15 }, { "min": 30, "max": 250, "val": 77 } ] },
    { "type": "made-up-operation" },
    { "type": "feature-extract", "output": %(csv)s },
  ],
}
"""


def test_check_config_lists_and_decodes(tmp_path):
    inp, out = tmp_path / "in", tmp_path / "out"
    inp.mkdir()
    out.mkdir()
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, (3, 9, 13), dtype=np.uint8)        # odd width: padded rows
    bmp_util.write_bmp24(str(inp / "cyl01.BMP"), rgb)
    bmp_util.write_bmp24(str(inp / "cyl02.BMP"), rgb, top_down=True)
    bmp_util.write_bmp8(str(inp / "inter01.BMP"), rgb[0])
    bmp_util.write_bmp24(str(inp / "skipped01.BMP"), rgb)          # not matched by the filter
    (inp / "inter02.BMP").write_bytes(b"BMnot really")
    cfg = tmp_path / "cfg.json"
    cfg.write_text(CONFIG % {"inp": json.dumps(str(inp)), "out": json.dumps(str(out)),
                             "csv": json.dumps(str(out / "features.csv"))})
    r = subprocess.run([DRIVER, str(cfg), "--check-config"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert "cyl01.BMP: 13x9" in lines and "cyl02.BMP: 13x9" in lines and "inter01.BMP: 13x9" in lines
    assert any(l.startswith("inter02.BMP: ") and "13x9" not in l for l in lines)
    assert not any("skipped01" in l for l in lines)
    assert "operation made-up-operation (unsupported)" in lines
    assert lines[-1] == "3 of 4 files readable, 5 operations"
    assert list(out.iterdir()) == []


def test_bad_config_is_reported(tmp_path):
    cfg = tmp_path / "bad.json"
    cfg.write_text('{ "image_folder": "x", "operations": [ { "type": "linear-filter" ')
    r = subprocess.run([DRIVER, str(cfg), "--check-config"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "JSON" in r.stderr
