"""Import alias for the package directory ``tf-faster-rcnn_amd/`` (a hyphen is not importable).

``import tf_faster_rcnn_amd`` loads ``tf-faster-rcnn_amd/__init__.py`` under this module's name, with
the directory as the package search path, so sub-modules import normally.
"""

import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tf-faster-rcnn_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_module = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _module
_spec.loader.exec_module(_module)
