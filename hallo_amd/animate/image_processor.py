"""`VaeImageProcessor.preprocess` of diffusers 0.27.2 as the two Hallo pipelines use it (third-party in the reference:
hallo/animate/face_animate.py:119-121, 333; hallo/animate/face_animate_static.py:117-125, 390-405)."""
import numpy as np
import torch


def preprocess_image(image, height, width, normalize):
    """diffusers VaeImageProcessor.preprocess (0.27.2) for the two processors StaticPipeline builds (:117-125):
    ref_image_processor (do_convert_rgb, do_normalize) and cond_image_processor (do_convert_rgb, no normalisation).
    PIL -> RGB, resize (lanczos) to (width, height), float32 / 255, NCHW, optional 2x - 1.  Tensors (n, 3, H, W) are resized with nearest
    sampling (diffusers' tensor branch) and normalised only if `normalize` and no value is negative (diffusers' own rule)."""
    if isinstance(image, torch.Tensor):
        x = image if image.dim() == 4 else image.unsqueeze(0)
        x = x.float()
        if tuple(x.shape[-2:]) != (height, width):
            # diffusers 0.27.2 resizes tensors with F.interpolate(size=...) = nearest: src index floor(dst * in / out)
            iy = (torch.arange(height, device=x.device) * x.shape[-2]) // height
            ix = (torch.arange(width, device=x.device) * x.shape[-1]) // width
            x = x[..., iy[:, None], ix[None, :]]
        if normalize and not bool(x.min() < 0):
            x = 2.0 * x - 1.0
        return x
    from PIL import Image
    imgs = image if isinstance(image, (list, tuple)) else [image]
    out = []
    for im in imgs:
        if not isinstance(im, Image.Image):
            raise TypeError(f"expected a PIL image or a tensor, got {type(im)}")
        im = im.convert("RGB")
        if im.size != (width, height):
            im = im.resize((width, height), resample=Image.LANCZOS)
        out.append(np.asarray(im, dtype=np.float32) / 255.0)
    x = torch.from_numpy(np.stack(out, axis=0)).permute(0, 3, 1, 2).contiguous()
    return 2.0 * x - 1.0 if normalize else x
