"""The reference's post-processing script under its own name (rpnToRegion.py:18-164): anchor decode + rotated NMS of one
sample's RPN maps, on the GPU (lisec_amd.boxes)."""
import numpy as np

from .boxes import rpnToRegion as _rpn_to_region


def rpnToRegion(labelsClass, labelsRegress):
    """rpnToRegion(labelsClass (100,200,2), labelsRegress (100,200,14)) (rpnToRegion.py:116-164) with the reference's
    fixed maxBoxes=20, overlapThresh=0.; returns what nonMaxSuppressionFast returns there: (boxes (k,7), probs (k,)),
    boxes = x, y, z, l, w, h, yaw.  A leading sample axis of length 1 (predictMain's files) is accepted."""
    cls, reg = np.asarray(labelsClass), np.asarray(labelsRegress)
    if cls.ndim == 4:
        cls, reg = cls[0], reg[0]
    return _rpn_to_region(cls, reg, maxBoxes=20, overlapThresh=0.)
