import cProfile, os, pstats, shutil, sys, tempfile
import numpy as np
from PIL import Image
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tmp = tempfile.mkdtemp(prefix="e2e_")
rs = np.random.RandomState(0)
base = rs.randint(0, 256, (512, 512, 3), dtype=np.uint8)
for i in range(4096):
    Image.fromarray(np.roll(base, i * 7, axis=1)).save(os.path.join(tmp, f"{i:06d}.jpg"), quality=90)
import torch
from clip_assisted_data_labeling_amd import embed_driver
from clip_assisted_data_labeling_amd.embedder import CLIP_Encoder
enc = CLIP_Encoder("ViT-L-14/seed0", None, device="cuda")
ds = embed_driver.Feature_Dataset(tmp, "ViT-L-14/seed0", 256, shuffle_filenames=False, num_workers=32, encoder=enc, device="cuda",
                                  gpu_preprocess=True, packed_store=os.path.join(tmp, "_s"), force_reencode=True)
pr = cProfile.Profile(); pr.enable()
ds.process()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
shutil.rmtree(tmp)
