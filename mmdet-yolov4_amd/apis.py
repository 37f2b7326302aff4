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


def _unwrap(field):
    """A test-time batch carries one entry per augmentation (``img=[tensor]``, ``img_metas=[[meta, ...]]``); the path
    has one (``BaseDetector.forward_test`` -> ``simple_test``, mmdet/models/detectors/base.py:128-166)."""
    if isinstance(field, (list, tuple)) and len(field) == 1 and isinstance(field[0], (list, tuple, torch.Tensor)):
        return field[0]
    return field


def single_gpu_test(model, data_loader):
    """Run the detector over ``data_loader`` and return the list of per-image results
    (``mmdet/apis/test.py:16-68`` without the visualisation branch).  Each item is ``dict(img=..., img_metas=...)``
    as the test pipeline's collate produces; ``rescale=True`` like the reference's test loop."""
    model.eval()
    device = next(model.parameters()).device
    results = []
    with torch.no_grad():
        for data in data_loader:
            img, metas = _unwrap(data['img']), _unwrap(data['img_metas'])
            results.extend(model.simple_test(img.to(device, non_blocking=True), metas, rescale=True))
    return results


def multi_gpu_test(model, data_loader, size=None, gpu_collect=True):
    """Every rank runs its ``sampler_indices`` share, then rank 0 receives the merged, dataset-ordered list and the
    others ``None`` (``mmdet/apis/test.py:71-113``).  ``size`` defaults to ``len(data_loader.dataset)``.  With
    ``gpu_collect`` the byte buffers of the gather live on the rank's GPU (RCCL); otherwise on the host (gloo) --
    the reference's other branch goes through a shared temp directory instead."""
    from . import dist as D
    results = single_gpu_test(model, data_loader)
    if size is None:
        size = len(data_loader.dataset)
    device = next(model.parameters()).device if gpu_collect else 'cpu'
    return D.collect_results(results, size, device=device)
