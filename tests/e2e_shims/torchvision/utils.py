import numpy as np
from PIL import Image


def save_image(tensor, path, **kw):
    a = (tensor.detach().clamp(0, 1).cpu().numpy() * 255.0 + 0.5).astype(np.uint8)
    if a.ndim == 3:
        a = a.transpose(1, 2, 0)
    Image.fromarray(a.squeeze()).save(path)
