"""Tensor-level front end of the HIP C-ABI (include/hd_ops.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every op below
hands raw pointers to libhd_ops.so.  There is NO CPU path: tensors must live on a HIP device
and the library must have been built (``__graft_entry__.build()``), otherwise these raise.

The op set mirrors what a detector framework of the reference's kind plugs into its
heads/assigners/post-processing (reference README.md:8-9,19,65): anchors, pairwise IoU +
assignment, delta codec, NMS, RPN proposals, RoIAlign, YOLOv5 decode / build-targets / loss,
CIoU, per-class NMS, WBF.

Split by op family (round 5): `_core` (plumbing), `detect` (the detection-specific ops of SURVEY.md section 8a), `backbone`
(BatchNorm / convolution / pooling glue of the backbone and neck), `graph` (hipGraph hygiene).  Everything is re-exported
here, so `from heltondetection_amd import ops; ops.yolo_loss(...)` is unchanged.
"""
from ._core import *  # noqa: F401,F403
from .backbone import *  # noqa: F401,F403
from .detect import *  # noqa: F401,F403
from .graph import *  # noqa: F401,F403
