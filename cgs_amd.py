"""Import alias: ``import cgs_amd`` yields the package living in the directory
``critic-guided-segmentation-of-rewarding-objects-in-first-person-views_amd/`` (whose name is not a valid
Python identifier).  Submodules are attributes of the package: ``from cgs_amd import nets, engine``."""
import importlib
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
PACKAGE_NAME = "critic-guided-segmentation-of-rewarding-objects-in-first-person-views_amd"
_pkg = importlib.import_module(PACKAGE_NAME)
sys.modules[__name__] = _pkg
