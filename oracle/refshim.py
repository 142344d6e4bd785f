"""Import shims that let the reference's hot-path modules load in the BUILD container.

Test infrastructure only (used by oracle/gen_fixtures.py).  The reference lives at
/root/reference, which exists only in the build container: nothing that runs on the GPU
box may call this.  Recipe follows SURVEY.md §8c:

  1. bypass ufvideo/__init__.py and ufvideo/model/__init__.py (they import timm/cv2 eagerly)
     by registering empty package shells whose __path__ points into the reference;
  2. transformers 5.x dropped TRANSFORMERS_CACHE (ufvideo/model/projector.py:24 imports it);
  3. placeholder timm RegStage/LayerNorm2d (timm is absent offline — RegStage stays unpinned);
  4. stub the I/O-only imports of ufvideo/mm_utils.py (cv2, decord, imageio, moviepy,
     torchvision.transforms.functional, pycocotools).
"""
import sys
import types

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    import transformers
    import torch.nn as nn

    transformers.TRANSFORMERS_CACHE = "/tmp/_no_cache"
    pk = types.ModuleType("ufvideo"); pk.__path__ = [REF + "/ufvideo"]; sys.modules["ufvideo"] = pk
    pm = types.ModuleType("ufvideo.model"); pm.__path__ = [REF + "/ufvideo/model"]; sys.modules["ufvideo.model"] = pm

    class _NoRegStage(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, x):
            raise RuntimeError("timm RegStage is not available offline")

    _stub("timm"); _stub("timm.models")
    _stub("timm.models.regnet", RegStage=_NoRegStage)
    _stub("timm.models.layers", LayerNorm=nn.LayerNorm, LayerNorm2d=nn.LayerNorm)
    for n in ["cv2", "imageio", "decord", "moviepy", "moviepy.editor", "torchvision", "torchvision.transforms",
              "torchvision.transforms.functional", "pycocotools", "pycocotools.mask"]:
        if n not in sys.modules:
            _stub(n)
    sys.modules["decord"].VideoReader = object
    sys.modules["decord"].cpu = lambda *a: None
    sys.modules["moviepy.editor"].VideoFileClip = object
    sys.modules["torchvision.transforms.functional"].resize = None
    sys.modules["torchvision.transforms.functional"].to_pil_image = None
    sys.modules["pycocotools"].mask = sys.modules["pycocotools.mask"]
