"""Mirror of the reference ``pc_processor`` package surface that the training hot path uses
(reference pc_processor/__init__.py, models/__init__.py:1-5, loss/__init__.py:1-3,
metrics/__init__.py:1, dataset/preprocess/__init__.py:1-2, postproc/knn.py)."""
from . import dataset, loss, metrics, models, postproc  # noqa: F401
