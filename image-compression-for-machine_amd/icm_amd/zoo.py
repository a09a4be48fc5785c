"""compressai/zoo mirror restricted to the hot path (zoo/__init__.py:23-43)."""
from .models import WACNN, SymmetricalTransFormer, SymmetricalTransFormer3

models = {"cnn": WACNN, "stf": SymmetricalTransFormer, "stf6": SymmetricalTransFormer3}
