"""debug: k_overlapg (array-level seam) on the goldens, errors per slot"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import archnemesis_dist_amd as pkg
eng = pkg.AnsfmEngine(0)
gd = os.path.join(ROOT, "tests", "golden")
for name in sorted(os.listdir(gd)):
    if not name.startswith("ck_"): continue
    z = np.load(os.path.join(gd, name))
    if "dk" not in z.files: continue
    taug, dk = eng.k_overlapg(z["DELG"], z["kg"], z["dkdT"], z["amount"])
    ref = z["dk"]
    print(name, "kg", z["kg"].shape, "dk", ref.shape, "tau err", np.max(np.abs(taug - z["taug"]) / (np.abs(z["taug"]) + 1e-300)))
    scale = np.abs(ref).max(axis=1, keepdims=True) + 1e-300
    err = np.abs(dk - ref) / scale
    # per slot (last axis)
    print("   per-slot max err:", ["%.1e" % e for e in err.reshape(-1, ref.shape[-1]).max(axis=0)])
    if err.max() > 1e-6:
        idx = np.unravel_index(np.argmax(err), err.shape)
        print("   worst at", idx, dk[idx], ref[idx])
        w, g, l, s = idx
        print("   got", dk[w, :, l, s]); print("   ref", ref[w, :, l, s])
