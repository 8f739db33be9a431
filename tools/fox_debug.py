import importlib, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch; torch.zeros(1, device="cuda")
from PIL import Image
PKG = "surface-irradiance-estimation-from-neural-radiance-fields_amd"
pyngp = importlib.import_module(PKG + ".build").import_pyngp()
FOX = os.path.join(ROOT, "tests", "golden", "fox")
t = pyngp.Testbed(); t.root_dir = FOX
t.load_training_data(os.path.join(FOX, "transforms_train.json"))
t.shall_train = True
while t.frame():
    if t.training_step >= 2000: break
print("loss", t.loss, "step", t.training_step)
t.shall_train = False
t.background_color = [0, 0, 0, 1]; t.snap_to_pixel_centers = True
def srgb(x): return np.where(x < 0.0031308, 12.92 * x, 1.055 * np.power(np.maximum(x, 1e-12), 0.41666) - 0.055)
def save(name, img): Image.fromarray((np.clip(srgb(img[..., :3]), 0, 1) * 255).astype(np.uint8)).save(os.path.join(ROOT, "gpurun_out", "fox", name))
res = t.nerf.training.dataset.metadata[0].resolution
print("res", res, "fov", t.fov, "fov_axis", t.fov_axis)
t.render_ground_truth = True; t.set_camera_to_training_view(0)
ref = t.render(res[0], res[1], 1, True); t.render_ground_truth = False
img = t.render(res[0], res[1], 4, True)
print("ref", ref.shape, ref[..., :3].mean(), "img", img[..., :3].mean(), img[..., 3].mean(), "fov", t.fov, t.fov_axis, "screen_center", t.screen_center)
save("dbg_ref.png", ref); save("dbg_out.png", img)
