"""The reference's own driver configurations (test/configs/*.json, kept as data under
tests/golden/reference_configs): the driver's parser reads every one of them, and every operation
list gives, through the batched executor, what the oracle gives."""
import glob
import json
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "imageanalysis_amd", "lib", "ia_driver")
CONFIGS = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "reference_configs", "*.json")))


def load_config(path):
    """JSON with // and /* */ comments and trailing commas, as jsoncpp accepts it."""
    text = open(path).read()
    out, i, in_str = [], 0, False
    while i < len(text):
        c = text[i]
        if in_str:
            out.append(c)
            if c == "\\":
                out.append(text[i + 1])
                i += 1
            elif c == '"':
                in_str = False
        elif c == '"':
            in_str = True
            out.append(c)
        elif text.startswith("//", i):
            while i < len(text) and text[i] != "\n":
                i += 1
            continue
        elif text.startswith("/*", i):
            i = text.index("*/", i) + 2
            continue
        else:
            out.append(c)
        i += 1
    return json.loads(re.sub(r",(\s*[}\]])", r"\1", "".join(out)))


def test_there_are_configs():
    assert len(CONFIGS) >= 16


@pytest.mark.parametrize("path", CONFIGS, ids=[os.path.basename(p) for p in CONFIGS])
def test_driver_parses_reference_config(path, tmp_path):
    cfg = load_config(path)
    # same text, with the folders pointed at an (empty) directory that exists
    text = open(path).read()
    folder = tmp_path / "images"
    folder.mkdir()
    for key in ("image_folder", "output_dir"):
        if key in cfg:
            text = text.replace(json.dumps(cfg[key]), json.dumps(str(folder)))
    local = tmp_path / "cfg.json"
    local.write_text(text)
    r = subprocess.run([DRIVER, str(local), "--check-config"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert lines[-1] == f"0 of 0 files readable, {len(cfg.get('operations', []))} operations"
    assert not any("(unsupported)" in l for l in lines), lines


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("path", CONFIGS, ids=[os.path.basename(p) for p in CONFIGS])
def test_reference_config_through_the_executor(path, fused):
    import torch
    import oracle_lib as O
    import synth
    from imageanalysis_amd.pipeline import Pipeline, ops_from_config
    ops = ops_from_config(load_config(path).get("operations", []))
    W, H, B = 64, 48, 3
    rgb = synth.rgb_batch("C", W, H, B)
    x0, x1 = O.xorwow_planes(W * H)
    nk = sum(1 for o in ops if o["type"] == O.OP_KMEANS)
    cent = O.libc_random_0_256(2 * nk * B).reshape(B, nk, 2) if nk else None
    want, want_hist, want_feats = O.pipeline_ex(rgb, ops, x0, x1, centroids=cent)
    p = Pipeline(W, H, B, ops, fused=fused)
    feats = torch.zeros((B, 7), dtype=torch.float32, device="cuda")
    p.bind_features(feats)
    hs = torch.zeros(256, dtype=torch.int64, device="cuda")
    got = p.run(torch.from_numpy(rgb).cuda(), hist_sum=hs).cpu().numpy()
    assert np.array_equal(got, want)
    assert np.array_equal(hs.cpu().numpy(), want_hist)
    if any(o["type"] == O.OP_FEATURES for o in ops):
        assert np.array_equal(feats.cpu().numpy(), want_feats, equal_nan=True)
    p.close()
