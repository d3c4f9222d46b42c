"""Import-path shim (`from ptsemseg.loss.loss import cross_entropy2d`) -> multiagentperception_amd.loss."""
from multiagentperception_amd.loss import (bootstrapped_cross_entropy2d, cross_entropy2d,  # noqa: F401
                                            multi_scale_cross_entropy2d)
