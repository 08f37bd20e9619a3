from .fpn import FPN  # noqa: F401
from .multi_stage_merging import MultiStageMerging  # noqa: F401
from .chain import NeckChain  # noqa: F401
