"""Alias so that `import mimamo_net_amd` resolves to the hyphenated package directory."""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
sys.modules[__name__] = importlib.import_module("mimamo-net_amd")
