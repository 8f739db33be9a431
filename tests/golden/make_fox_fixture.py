"""Builds tests/golden/fox/ -- the reference's own sample scene as DATA: its 50 photographs (data/nerf/fox/images/*.jpg, 1080 x 1920)
downscaled by 2 and re-encoded as baseline JPEGs, and its transforms.json with the pixel-unit intrinsics scaled to match (the
distortion coefficients are in normalised coordinates and stay). Every 8th photograph is held out: transforms_train.json /
transforms_test.json list the two sets (same header), transforms.json lists all frames as the reference's file does -- including the
17 frames whose images the reference does not ship, which its loader (src/nerf_loader.cu:364-388) and this build's skip.
Run once in the build container (needs /root/reference and Pillow); the GPU box only reads the result."""
import json, os, sys
from PIL import Image, JpegImagePlugin

SRC = "/root/reference/data/nerf/fox"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fox")
os.makedirs(os.path.join(DST, "images"), exist_ok=True)
t = json.load(open(os.path.join(SRC, "transforms.json")))
for k in ("fl_x", "fl_y", "cx", "cy", "w", "h"):
    t[k] = t[k] * 0.5
kept = []
for fr in t["frames"]:
    p = os.path.join(SRC, fr["file_path"])
    if not os.path.exists(p):
        continue
    im = Image.open(p)
    assert im.size == (1080, 1920)
    small = im.resize((540, 960), Image.LANCZOS)
    small.save(os.path.join(DST, fr["file_path"]), "JPEG", quality=92, progressive=False, subsampling=JpegImagePlugin.get_sampling(im))
    kept.append(fr)
json.dump(t, open(os.path.join(DST, "transforms.json"), "w"), indent=1)
test = [fr for i, fr in enumerate(kept) if i % 8 == 4]
train = [fr for i, fr in enumerate(kept) if i % 8 != 4]
for name, frames in (("transforms_train.json", train), ("transforms_test.json", test)):
    d = dict(t)
    d["frames"] = frames
    json.dump(d, open(os.path.join(DST, name), "w"), indent=1)
print(f"{len(kept)} photographs ({len(train)} train / {len(test)} held out), {sum(os.path.getsize(os.path.join(DST, f['file_path'])) for f in kept) / 1e6:.1f} MB")
