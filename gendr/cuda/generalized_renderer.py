"""``gendr.cuda.generalized_renderer`` of the reference (pybind11 module) -> C-ABI backed functions."""
from gendr_amd.cuda.generalized_renderer import (forward_render, backward_render, sigmoid_forward,       # noqa: F401
                                                 sigmoid_backward, t_conorm_forward, t_conorm_backward)
