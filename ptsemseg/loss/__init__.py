"""Import-path shim (`from ptsemseg.loss import get_loss_function`, train.py:19) -> multiagentperception_amd.loss."""
from multiagentperception_amd.loss import (bootstrapped_cross_entropy2d, cross_entropy2d, get_loss_function,  # noqa: F401
                                            key2loss, multi_scale_cross_entropy2d)
