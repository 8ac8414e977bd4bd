#!/usr/bin/env python3
"""Writes tests/golden/ref_host.npz: inputs and the outputs the REFERENCE'S OWN host-only
functions give for them.  Those functions are compiled, unmodified, from /root/reference by
`make -C oracle ref` (oracle/_ref/libref_host.so); this script only runs where that library
exists (the authoring container).  The .npz is data -- inputs and expected outputs -- and is
what travels; the library does not.

    python tests/golden/make_ref_host_golden.py
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402

i32p = np.ctypeslib.ndpointer(np.int32, flags="C")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C")


def load():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"])
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_host.so"))
    L.ref_random.argtypes = [C.c_int, C.c_int]
    L.ref_equalizeHistogram.argtypes = [i32p, i32p, C.c_int]
    L.ref_centroidsHaveChanged.argtypes = [u8p, C.c_int]
    L.ref_arraySum.argtypes = [i32p, C.c_int]
    for n in ("ref_calcMode", "ref_calculateCellPerimeter"):
        getattr(L, n).argtypes = [i32p]
    L.ref_calculateCellArea.argtypes = [i32p, C.c_int]
    L.ref_calcMinAndMaxOfRange.argtypes = [i32p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ref_confusion_metrics.argtypes = [i32p, f32p, C.c_int, C.c_int, i32p, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                        C.POINTER(C.c_float)]
    return L


def histograms(rs):
    """Histograms as the pipeline produces them (bin 0 empty) from the three synthetic image
    kinds at several sizes, plus arbitrary ones (bin 0 filled, sparse, single bin, empty)."""
    hs, ns = [], []
    for kind, W, H in [("C", 64, 48), ("U", 37, 23), ("K", 32, 16), ("C", 768, 568), ("U", 768, 568), ("C", 200, 100)]:
        for idx in range(2):
            g = synth.gray_image(kind, W, H, idx)
            h = np.bincount(g.reshape(-1), minlength=256).astype(np.int32)
            h[0] = 0
            hs.append(h)
            ns.append(W * H)
    for _ in range(8):
        h = rs.integers(0, 5000, 256).astype(np.int32)
        h[rs.random(256) < rs.random()] = 0
        hs.append(h)
        ns.append(int(h.sum()))
    one = np.zeros(256, np.int32)
    one[77] = 4096
    hs.append(one)
    ns.append(4096)
    top = np.zeros(256, np.int32)
    top[255] = 10
    top[1] = 10
    hs.append(top)
    ns.append(40)            # half of the pixels are zeros the histogram does not hold
    hs.append(np.zeros(256, np.int32))
    ns.append(512)           # an all-zero image
    kat = np.zeros(256, np.int32)
    kat[10], kat[20], kat[200] = 4, 4, 8
    hs.append(kat)
    ns.append(16)
    return np.stack(hs), np.array(ns, np.int32)


def main():
    L = load()
    rs = np.random.default_rng(20261111)
    libc = C.CDLL("libc.so.6")
    out = {}

    # equalizeHistogram, imageAnalysis.cu:155-176
    hs, ns = histograms(rs)
    maps = np.zeros_like(hs)
    for i in range(len(hs)):
        m = np.zeros(256, np.int32)
        L.ref_equalizeHistogram(hs[i].copy(), m, int(ns[i]))
        maps[i] = m
    out.update(eq_hist=hs, eq_num_pixels=ns, eq_map=maps)

    # random(), imageAnalysis.cu:39-41, on the rand() stream of a program that never calls srand
    libc.srand(1)
    out["random_0_256_unseeded"] = np.array([L.ref_random(0, 256) for _ in range(4000)], np.int32)
    libc.srand(20261111)
    out["random_3_17_seed_20261111"] = np.array([L.ref_random(3, 17) for _ in range(500)], np.int32)

    # centroidsHaveChanged, imageAnalysis.cu:281-293
    cin, cout, cchg, ck = [], [], [], []
    for k in (2, 2, 2, 3, 5):
        for _ in range(60):
            c = rs.integers(0, 256, 2 * 5).astype(np.uint8)
            if rs.random() < 0.4:
                c[k:2 * k] = c[:k]           # nothing changed
            if rs.random() < 0.3:
                c[k] = c[0]                  # one of them unchanged
            work = c.copy()
            chg = L.ref_centroidsHaveChanged(work, k)
            cin.append(c)
            cout.append(work)
            cchg.append(chg)
            ck.append(k)
    out.update(cent_in=np.stack(cin), cent_out=np.stack(cout), cent_changed=np.array(cchg, np.int32), cent_k=np.array(ck, np.int32))

    # arraySum, imageAnalysis.cu:682-688
    arrs = [rs.integers(0, 9, n).astype(np.int32) for n in (9, 25, 49, 121, 1)] + [np.ones(9, np.int32)]
    out["sum_in"] = np.array([np.pad(a, (0, 121 - len(a))) for a in arrs], np.int32)
    out["sum_len"] = np.array([len(a) for a in arrs], np.int32)
    out["sum_out"] = np.array([L.ref_arraySum(a.copy(), len(a)) for a in arrs], np.int32)

    # calcMode / calculateCellArea / calculateCellPerimeter / calcMinAndMaxOfRange, imageMachineLearning.cu:341-391
    fh, fn = histograms(rs)
    tie = np.zeros(256, np.int32)
    tie[[5, 90, 250]] = 33                   # equal maxima: the first wins
    fh = np.concatenate([fh, tie[None]])
    fn = np.concatenate([fn, np.array([200], np.int32)])
    mode, area, peri, lo, hi = [], [], [], [], []
    for h, n in zip(fh, fn):
        a, b = C.c_int(0), C.c_int(0)
        mode.append(L.ref_calcMode(h.copy()))
        area.append(L.ref_calculateCellArea(h.copy(), int(n)))
        peri.append(L.ref_calculateCellPerimeter(h.copy()))
        L.ref_calcMinAndMaxOfRange(h.copy(), C.byref(a), C.byref(b))
        lo.append(a.value)
        hi.append(b.value)
    out.update(feat_hist=fh, feat_num_pixels=fn, feat_mode=np.array(mode, np.int32), feat_area=np.array(area, np.int32),
               feat_perimeter=np.array(peri, np.int32), feat_range_min=np.array(lo, np.int32),
               feat_range_max=np.array(hi, np.int32))

    # computeConfusionMatrix / computeAccuracy / computePrecisionAndRecall, imageMachineLearning.cu:214-255
    A = 8
    mp, md, mn, mc, macc, mprec, mrec = [], [], [], [], [], [], []
    for n in (1, 7, 20, 50, 50, 203):
        rows = rs.normal(0, 1, (n, A)).astype(np.float32)
        true = rs.integers(0, 6, n)
        if n == 50 and not mn.count(50):
            true = rs.integers(0, 3, n)       # classes never seen: empty rows and columns
        rows[:, A - 1] = true
        pred = np.where(rs.random(n) < 0.7, true, rs.integers(0, 6, n)).astype(np.int32)
        cm = np.zeros(36, np.int32)
        acc, prec, rec = C.c_float(), C.c_float(), C.c_float()
        L.ref_confusion_metrics(pred.copy(), rows.copy(), A, n, cm, C.byref(acc), C.byref(prec), C.byref(rec))
        mp.append(np.pad(pred, (0, 203 - n)))
        md.append(np.pad(rows, ((0, 203 - n), (0, 0))))
        mn.append(n)
        mc.append(cm)
        macc.append(acc.value)
        mprec.append(prec.value)
        mrec.append(rec.value)
    out.update(cm_pred=np.stack(mp), cm_rows=np.stack(md), cm_n=np.array(mn, np.int32), cm_confusion=np.stack(mc),
               cm_accuracy=np.array(macc, np.float32), cm_precision=np.array(mprec, np.float32),
               cm_recall=np.array(mrec, np.float32), cm_num_classes=np.array([L.ref_num_classes()], np.int32))

    path = os.path.join(HERE, "ref_host.npz")
    np.savez_compressed(path, **out)
    print(path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
