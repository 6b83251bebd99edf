"""NYU-Depth-v2 validation transform on the GPU.

Reference: dataloaders/nyu.py:5 (raw frames are 480 x 640), :48-59 `val_transform`
    Resize(250.0 / iheight) -> CenterCrop((228, 304)) -> Resize(output_size)      (transforms.py:311-341, 344-392)
followed by `/ 255` for the colour image; every Resize is scipy.misc.imresize(..., 'nearest'), i.e. PIL's NEAREST resize.
Nearest-neighbour resizing and cropping are index maps, so the whole chain is one row table and one column table; the tables
reproduce PIL's arithmetic exactly (its affine scaler accumulates the source coordinate by repeated addition in double and
truncates: x_src[i] = int(s/2 + s + s + ...), pinned against PIL itself by tests/test_oracle.py).  `GpuValTransform` uploads
the tables once and calls `fd_val_transform`, which gathers raw uint8 HWC frames (+ raw depth) into the network's NCHW float
input (+ the depth target) -- no per-frame CPU work.
"""
import numpy as np
import torch

IHEIGHT, IWIDTH = 480, 640           # raw NYU frame (reference nyu.py:5)


def _nearest_table(n_in, n_out):
    """Source index of every output index for PIL's NEAREST resize of n_in samples to n_out."""
    scale = n_in / n_out
    pos, tab = scale * 0.5, np.empty(n_out, np.int32)
    for i in range(n_out):
        tab[i] = int(pos)
        pos += scale
    return np.minimum(tab, n_in - 1)


def val_index_maps(output_size=(224, 224), iheight=IHEIGHT, iwidth=IWIDTH):
    """(ymap[out_h], xmap[out_w]): raw-frame row / column read by each output row / column."""
    f = 250.0 / iheight
    w1, h1 = int(iwidth * f), int(iheight * f)                 # imresize with a float: size = (array(im.size) * f).astype(int)
    y1, x1 = _nearest_table(iheight, h1), _nearest_table(iwidth, w1)
    th, tw = 228, 304
    i, j = int(round((h1 - th) / 2.)), int(round((w1 - tw) / 2.))     # CenterCrop.get_params (transforms.py:373-374)
    if i < 0 or j < 0:
        raise ValueError("frame too small for the 228 x 304 centre crop")
    oh, ow = output_size
    y2, x2 = _nearest_table(th, oh), _nearest_table(tw, ow)
    return y1[i + y2].astype(np.int32), x1[j + x2].astype(np.int32)


class GpuValTransform:
    """rgb [n, H, W, 3] uint8 (GPU), depth [n, H, W] float32 (GPU, optional) -> x [n, 3, oh, ow] float32, depth [n, 1, oh, ow]."""

    def __init__(self, output_size=(224, 224), device="cuda", iheight=IHEIGHT, iwidth=IWIDTH):
        self.output_size, self.raw = tuple(output_size), (iheight, iwidth)
        ymap, xmap = val_index_maps(output_size, iheight, iwidth)
        self.ymap, self.xmap = torch.from_numpy(ymap).to(device), torch.from_numpy(xmap).to(device)

    def __call__(self, rgb, depth=None):
        from fastdepth_hip import capi
        from fastdepth_hip.engine import lib
        if not rgb.is_cuda or rgb.dtype != torch.uint8 or rgb.dim() != 4 or rgb.shape[-1] != 3 or tuple(rgb.shape[1:3]) != self.raw:
            raise RuntimeError("expected a uint8 [n, %d, %d, 3] GPU tensor, got %s %s on %s" % (self.raw + (tuple(rgb.shape), rgb.dtype, rgb.device)))
        rgb = rgb.contiguous()
        n, (oh, ow) = rgb.shape[0], self.output_size
        x = torch.empty((n, 3, oh, ow), dtype=torch.float32, device=rgb.device)
        d = dp = None
        if depth is not None:
            if not depth.is_cuda or depth.dtype != torch.float32 or tuple(depth.shape) != (n,) + self.raw:
                raise RuntimeError("expected a float32 [n, %d, %d] GPU depth tensor" % self.raw)
            depth = depth.contiguous()
            d = torch.empty((n, 1, oh, ow), dtype=torch.float32, device=rgb.device)
            dp = depth.data_ptr()
        L = lib()
        with torch.cuda.device(rgb.device):
            capi.check(L, L.fd_val_transform(rgb.data_ptr(), dp, n, self.raw[0], self.raw[1], oh, ow, self.ymap.data_ptr(), self.xmap.data_ptr(),
                                             x.data_ptr(), d.data_ptr() if d is not None else None,
                                             torch.cuda.current_stream(rgb.device).cuda_stream), "fd_val_transform")
        return (x, d) if depth is not None else x
