"""How many front-end points differ from the reference's golden points after the float32 cast, and where."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bnv_fusion_amd.frontend import depth_to_input_pts
z = np.load("tests/golden/frontend_120.npz")
for name, src in (("f64", torch.from_numpy(z["depth"])), ("f32", torch.from_numpy(z["depth"].astype(np.float32))),
                  ("u16", torch.from_numpy(np.round(z["depth"] * 1000).astype(np.uint16)))):
    pts = depth_to_input_pts(src.cuda(), z["intr"], z["T_wc"], max_depth=float(z["max_depth"]))[0].cpu().numpy()
    ref = z["pts_w"].astype(np.float32)
    if pts.shape[0] != ref.shape[0]:
        print(name, "row count differs", pts.shape, ref.shape); continue
    bad = pts[:, :3] != ref
    print(name, "mismatching values:", int(bad.sum()), "of", bad.size, "per axis", bad.sum(0).tolist(),
          "max abs", float(np.abs(pts[:, :3] - ref).max()))
    for r, c in np.argwhere(bad)[:5]:
        print("   row", r, "axis", c, repr(float(pts[r, c])), repr(float(ref[r, c])), "ref f64", repr(float(z["pts_w"][r, c])))
