"""Minimal stand-in for `yacs.config.CfgNode`, only so that the golden-vector
generator can import the reference package in the build container (yacs is not
installed there).  Not used by the product or the tests."""


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v
