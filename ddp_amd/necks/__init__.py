from .multi_stage_merging import MultiStageMerging  # noqa: F401
