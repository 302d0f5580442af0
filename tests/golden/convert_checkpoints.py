"""Convert the reference's Lightning checkpoints to plain .npz tensor files.

Run in the build container only (needs /root/reference/pretrained):
    python tests/golden/convert_checkpoints.py
Weights are data: only the hot-path tensors are kept (point encoder + SDF
decoder; the colour head nerf.color_layer*/fc_rgb is never evaluated on this
path, modules.py:919), under their original state_dict key names so that
``LitFusionPointNet.load_state_dict`` accepts either source.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(__file__))
import ref_shims  # noqa: E402

OUT = os.path.join(os.path.dirname(__file__), "..", "..", "bnv_fusion_amd", "weights")


def main():
    os.makedirs(OUT, exist_ok=True)
    sd = ref_shims.load_checkpoint_state_dict("/root/reference/pretrained/pointnet.ckpt")
    keep = {k: v.numpy() for k, v in sd.items()
            if k.startswith("pointnet_backbone.") or k.startswith("nerf.geo_layer")
            or k.startswith("nerf.fc_alpha")}
    np.savez(os.path.join(OUT, "pointnet_fp32.npz"), **keep)
    print("fp32:", {k: v.shape for k, v in keep.items()})
    sd = ref_shims.load_checkpoint_state_dict("/root/reference/pretrained/pointnet_tcnn.ckpt")
    keep = {k: v.numpy() for k, v in sd.items()}
    np.savez(os.path.join(OUT, "pointnet_tcnn.npz"), **keep)
    print("tcnn:", {k: v.shape for k, v in keep.items()})


if __name__ == "__main__":
    main()
