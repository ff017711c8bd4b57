import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from gyroflow_amd import abi, synthetic as S, warp
import _oracle as O
from test_gpu_lens_models import DIGITAL
w, h = 3840, 2160
lens = S.gopro_style_lens(w, h); lens["digital"] = "gopro_superview"
fr = S.SyntheticFrame("YUV422P16LE", 640, 360, seed=37, lens=dict(S.gopro_style_lens(640, 360), digital="gopro_superview"), fov=1.1)
ref = O.run_frame(fr); got = warp.run_frame(fr)
print("digital:", warp.last_backend(), all(np.array_equal(a, b) for a, b in zip(ref, got)))
fr = S.SyntheticFrame("YUV422P16LE", 640, 360, seed=38, fov=1.1, base_overrides={"light_refraction_coefficient": 1.33})
ref = O.run_frame(fr); got = warp.run_frame(fr)
print("refraction:", warp.last_backend(), all(np.array_equal(a, b) for a, b in zip(ref, got)))
