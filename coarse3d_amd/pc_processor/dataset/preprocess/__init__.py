from .augmentor import AugmentParams, Augmentor  # noqa: F401
from .projection import RangeProjection  # noqa: F401
