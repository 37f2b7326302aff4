"""``inference_detector`` under the reference's name (``mmdet/apis/inference.py:88-158``), for loaded images.

The reference composes ``cfg.data.test.pipeline`` per image on the CPU (mmcv / OpenCV), collates, scatters and calls
``model(return_loss=False, rescale=True, **data)``.  Here the same pipeline block drives ``FusedTestPipeline`` (one
launch per image, parity unpinned -- see ``preprocess.py``) and the detector's fused ``simple_test``.  Image files are
not read: the build has no image decoder (the reference reads them with ``mmcv.imread``)."""
import numpy as np
import torch

from .preprocess import FusedTestPipeline


def inference_detector(model, imgs, test_pipeline=None):
    """``imgs``: an (h, w, 3) uint8 array (BGR, as ``mmcv.imread`` returns) or a list / tuple of them.  Returns the
    per-image result (a list of per-class (k, 5) arrays) or, for a list, the list of them -- like the reference.
    ``test_pipeline`` defaults to ``model.cfg.data.test.pipeline``."""
    is_batch = isinstance(imgs, (list, tuple))
    if not is_batch:
        imgs = [imgs]
    if any(not isinstance(i, np.ndarray) for i in imgs):
        raise NotImplementedError('inference_detector takes loaded images (numpy arrays): no image decoder in this build')
    if test_pipeline is None:
        cfg = getattr(model, 'cfg', None)
        if cfg is None:
            raise ValueError('pass test_pipeline= or attach the config as model.cfg (init_detector does in the reference)')
        test_pipeline = cfg.data.test.pipeline
    device = next(model.parameters()).device
    pipe = test_pipeline if callable(test_pipeline) else FusedTestPipeline.from_config(test_pipeline, device=device)
    batch, metas = pipe(list(imgs))
    with torch.no_grad():
        results = model.simple_test(batch, metas, rescale=True)
    return results if is_batch else results[0]
