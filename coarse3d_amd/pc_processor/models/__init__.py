from .salsanext_proto import SalsaNextProto  # noqa: F401
from .rangenet_proto import RangeNetProto  # noqa: F401
from .squeezesegv3_proto import SqueezeSegV3Proto  # noqa: F401
from .sinkhorn import distributed_sinkhorn  # noqa: F401
from .projector import ProjectionV1  # noqa: F401
