"""Import alias: the package directory is ``mmdet-yolov4_amd/`` (not an identifier), so
``import mmdet_yolov4_amd`` loads it from there under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mmdet-yolov4_amd')
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
