"""PIL restatement of the reference's NYU validation transform (parity yard-stick).  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/dataloaders/nyu.py:48-59 (val_transform) with transforms.Resize / CenterCrop
(/root/reference/dataloaders/transforms.py:311-341, 344-392).  The reference's Resize calls scipy.misc.imresize, which was
removed from SciPy; its documented implementation (scipy 1.2 `misc/pilutil.py`) is restated here on the PIL that is installed:
    imresize(arr, size, interp='nearest', mode):  im = toimage(arr, mode);  size = (array(im.size) * size).astype(int) for a float,
    (size[1], size[0]) for a tuple;  return fromimage(im.resize(size, resample=NEAREST))
with mode None for the uint8 RGB frame and 'F' (float32, no rescaling) for the depth map."""
import numpy as np
from PIL import Image


def _imresize(arr, size, mode=None):
    im = Image.fromarray(arr.astype(np.float32), mode="F") if mode == "F" else Image.fromarray(arr)
    if isinstance(size, float):
        size = tuple((np.array(im.size) * size).astype(int))
    else:
        size = (size[1], size[0])
    return np.asarray(im.resize(size, resample=Image.NEAREST))


def _resize(img, size):                                        # transforms.Resize.__call__ (:329-341)
    return _imresize(img, size) if img.ndim == 3 else _imresize(img, size, "F")


def _center_crop(img, size):                                   # transforms.CenterCrop (:360-392)
    th, tw = size
    h, w = img.shape[0], img.shape[1]
    i, j = int(round((h - th) / 2.)), int(round((w - tw) / 2.))
    return img[i:i + th, j:j + tw]


def val_transform(rgb_u8, depth, output_size=(224, 224), iheight=480):
    """rgb_u8 [H, W, 3] uint8, depth [H, W] float -> (rgb [oh, ow, 3] float64 in [0, 1], depth [oh, ow] float32): nyu.py:48-59."""
    def chain(img):
        return _resize(_center_crop(_resize(img, 250.0 / iheight), (228, 304)), tuple(output_size))
    return np.asarray(chain(rgb_u8), dtype=np.float64) / 255, chain(depth)
