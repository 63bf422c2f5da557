from .lovasz_softmax import Lovasz_softmax  # noqa: F401
from .focal_softmax import FocalSoftmaxLoss  # noqa: F401
from .contrast_pixel_loss import ContrastMEMLoss  # noqa: F401
