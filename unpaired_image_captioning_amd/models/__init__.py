"""`models.setup(opt)` with the reference's dispatch (P/models/__init__.py:22-58), restricted to
the architectures on the hot path."""
from .AttModel import TopDownModel  # noqa: F401
from .CaptionModel import CaptionModel  # noqa: F401
from .FCModel_NMT import FCModel_NMT  # noqa: F401
from . import NMT_Models  # noqa: F401
from .Discriminator import SentenceDiscriminator  # noqa: F401
from .GCN import SceneGraphEncoder  # noqa: F401


def setup(opt):
    if opt.caption_model == 'fc':
        return FCModel_NMT(opt)          # P/models/__init__.py:24-26
    if opt.caption_model == 'topdown':
        return TopDownModel(opt)
    raise Exception("Caption model not supported by the MI355X hot path: {}".format(opt.caption_model))
