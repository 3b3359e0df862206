"""Empty stand-in: the reference imports torchvision unconditionally from
model/methods/Interp_Parts.py (out of scope); nothing on the hot path uses it."""
