"""Mirror of the part of ``pc_processor.dataset`` that sits directly in front of the training step
(reference pc_processor/dataset/preprocess/__init__.py:1-2): point augmentation and the spherical
range projection, on the device."""
from . import preprocess  # noqa: F401
