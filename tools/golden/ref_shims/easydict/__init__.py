class EasyDict(dict):
    """attribute-access dict; only imported by the reference's _irpe.py (dead: rpe_config is null)."""
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v
