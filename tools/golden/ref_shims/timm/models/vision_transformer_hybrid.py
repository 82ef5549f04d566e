"""HybridEmbed is never instantiated (hybrid_backbone: null in every shipped config)."""


class HybridEmbed:
    def __init__(self, *a, **k):
        raise NotImplementedError('HybridEmbed is out of scope (SURVEY.md §2 row 7)')
