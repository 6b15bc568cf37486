"""tf-faster-rcnn_amd -- MI355X-native proposal/RoI hot path of FurkanOM/tf-faster-rcnn.

The directory name carries a hyphen (the project's name); import it as ``tf_faster_rcnn_amd``
(the alias module at the repository root) -- ``tf_faster_rcnn_amd.models.faster_rcnn``,
``tf_faster_rcnn_amd.utils.bbox_utils`` and ``tf_faster_rcnn_amd.utils.train_utils`` mirror the
reference's ``models/faster_rcnn.py``, ``utils/bbox_utils.py`` and ``utils/train_utils.py``.
"""

from . import _lib, sharding  # noqa: F401
from .models import faster_rcnn  # noqa: F401
from .utils import bbox_utils, eval_utils, train_utils  # noqa: F401
from . import pipeline  # noqa: F401,E402

__version__ = "0.1.0"


def library_path() -> str:
    """Path of the C-ABI shared library the package binds to."""
    return _lib.LIB_PATH
