"""Mirror of the reference ``pc_processor`` package surface that the training hot path uses
(reference pc_processor/__init__.py, models/__init__.py:1-5, loss/__init__.py:1-3,
metrics/__init__.py:1)."""
from . import dataset, loss, metrics, models  # noqa: F401
